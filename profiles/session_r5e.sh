#!/bin/bash
# round 5: tail sharing of the output layer (HIPETS_TAIL_SHARE) + Philox on 64-bit products -- cfg2 per row-tile count against the
# -DHIPETS_TAIL_SHARE=0 variant on the same box, stock workloads, the tests that compare instances / replay plans
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r5e; mkdir -p $OUT
run() { name=$1; shift; echo "== $name: $*" ; ( time timeout ${TMO:-1500} "$@" ) > $OUT/$name.log 2>&1; echo "   rc=$? $(tail -n 3 $OUT/$name.log | tr '\n' ' ' | cut -c1-300)"; }
run smoke python -c "import __graft_entry__ as g; g.smoke()"
run variants_default python profiles/kernel_variants.py
HIPETS_LIB=$PWD/profiles/variants/nots.so run variants_nots python profiles/kernel_variants.py
run stock python profiles/stock_workloads.py --no-plans
HIPETS_LIB=$PWD/profiles/variants/nots.so run stock_nots python profiles/stock_workloads.py --no-plans
HIPETS_ORACLE_CACHE_OUT=$PWD/gpurun_out/oracle_cache run tests python -m pytest tests/test_gpu_rollout.py tests/test_gpu_closed_forms.py tests/test_gpu_device_mode.py tests/test_gpu_planning.py tests/test_gpu_plans_full_size.py -m gpu -q --maxfail=40 -p no:cacheprovider
run bench python bench.py --no-cpu-baseline
grep -h '"metric"' $OUT/bench.log | tail -1 > $OUT/bench_line.json
echo done
