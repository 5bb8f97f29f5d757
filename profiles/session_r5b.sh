#!/bin/bash
# round 5, second GPU session: fragments two chunks ahead in the k-split instances (HIPETS_KS_TRIPLE) against the one-chunk form and
# against no k-split on the same box; pets_hopper's fused DEVICE-mode instances (tests + timings); the sliced member-schedule kernel
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r5b; mkdir -p $OUT
run() { name=$1; shift; echo "== $name: $*" ; ( time timeout ${TMO:-1500} "$@" ) > $OUT/$name.log 2>&1; echo "   rc=$? $(tail -n 3 $OUT/$name.log | tr '\n' ' ' | cut -c1-300)"; }
run smoke python -c "import __graft_entry__ as g; g.smoke()"
run small_default python profiles/small_batch_probe.py
HIPETS_LIB=$PWD/profiles/variants/ks2.so run small_ks2 python profiles/small_batch_probe.py
HIPETS_LIB=$PWD/profiles/variants/noks.so run small_noks python profiles/small_batch_probe.py
run hopper python profiles/stock_workloads.py --only stock_hopper --sweep-r
HIPETS_ORACLE_CACHE_OUT=$PWD/gpurun_out/oracle_cache run tests python -m pytest tests/test_gpu_rollout.py tests/test_gpu_closed_forms.py tests/test_gpu_device_mode.py tests/test_gpu_planning.py tests/test_gpu_batched_plans.py tests/test_gpu_cost_model.py -m gpu -q --maxfail=40 -p no:cacheprovider
run bench python bench.py --no-cpu-baseline
grep -h '"metric"' $OUT/bench.log | tail -1 > $OUT/bench_line.json
echo done
