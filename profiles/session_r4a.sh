#!/bin/bash
# round 4, session a: the reference's shipped workloads on the round-3 binary (parity first; timings = the "before" column)
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r4a; mkdir -p $OUT
run() { name=$1; shift; echo "== $name: $*" ; ( time timeout ${TMO:-900} "$@" ) > $OUT/$name.log 2>&1; echo "   rc=$? $(tail -n 3 $OUT/$name.log | tr '\n' ' ' | cut -c1-300)"; }
run tests python -m pytest -m gpu -q --maxfail=10 -p no:cacheprovider --durations=15 tests/test_gpu_rollout.py tests/test_gpu_device_mode.py tests/test_gpu_plans_full_size.py -k "obs18 or pop350 or pop80x5 or stock"
run stock python profiles/stock_workloads.py --sweep-r --generic
grep -v '^\[' $OUT/stock.log | sed -n '/^{/,$p' | sed '/^real/,$d' > $OUT/stock_workloads_before.json
echo done
