#!/bin/bash
# round 3, GPU session A: the new / changed parity tests + a short bench line.  bash profiles/session_r3a.sh
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r3a
mkdir -p $OUT
run() { name=$1; shift; echo "== $name: $*" ; ( time timeout ${TMO:-900} "$@" ) > $OUT/$name.log 2>&1; echo "   rc=$? $(tail -n 4 $OUT/$name.log | tr '\n' ' ' | cut -c1-400)"; }
run tests_new python -m pytest -m gpu -q --maxfail=8 -p no:cacheprovider tests/test_gpu_closed_forms.py tests/test_stock_configs.py tests/test_gpu_device_mode.py tests/test_gpu_dist.py tests/test_gpu_rollout.py
run tests_world python -m pytest -m gpu -q --maxfail=8 -p no:cacheprovider tests/test_gpu_sharded_world.py
run tests_icem python -m pytest -m gpu -q --maxfail=8 -p no:cacheprovider tests/test_gpu_plans_full_size.py -k "icem"
run bench python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras
grep -h '"metric"' $OUT/bench.log | tail -1 > $OUT/bench_line.json
echo done
