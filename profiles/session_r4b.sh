#!/bin/bash
# round 4, session b: hidden-static instances (KSpec::HID_STATIC) -- bitwise vs generic, stock workloads parity, timings per R
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r4b; mkdir -p $OUT
run() { name=$1; shift; echo "== $name: $*" ; ( time timeout ${TMO:-900} "$@" ) > $OUT/$name.log 2>&1; echo "   rc=$? $(tail -n 3 $OUT/$name.log | tr '\n' ' ' | cut -c1-300)"; }
run tests python -m pytest -m gpu -q --maxfail=10 -p no:cacheprovider --durations=10 tests/test_gpu_rollout.py tests/test_gpu_device_mode.py tests/test_gpu_plans_full_size.py -k "obs18 or pop350 or pop80x5 or stock or hidden_static or shape_specialised or persistent_and_per_step"
run stock python profiles/stock_workloads.py --sweep-r --generic
sed -n '/^{/,$p' $OUT/stock.log | sed '/^real/,$d' > $OUT/stock_workloads.json
echo done
