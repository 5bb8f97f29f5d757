#!/bin/bash
# round 6, session v: MPPI kernels -- the update's weighted sum with its population staged through LDS tiles by all sixteen waves, the sampler
# with whole candidates staged in LDS (draws element-parallel, recurrence in LDS, coalesced stores): parity of everything that plans (memo on:
# every replayed plan must keep its bits), kernel statistics of the cfg5 MPPI plan and of the shipped pets_mppi_halfcheetah workload
set -u
export TMPDIR=/tmp
OUT=gpurun_out/${SESSION:-r6v}; mkdir -p $OUT
run() { name=$1; shift; echo "== $name: $*" ; ( time timeout ${TMO:-900} "$@" ) > $OUT/$name.log 2>&1; echo "   rc=$? $(tail -n 3 $OUT/$name.log | tr '\n' ' ' | cut -c1-300)"; }
run tests_plans python -m pytest tests/test_gpu_planning.py tests/test_gpu_plans_full_size.py tests/test_gpu_batched_plans.py tests/test_gpu_sharded_world.py tests/test_gpu_c_abi.py tests/test_stock_configs.py -m gpu -q -p no:cacheprovider --durations=8
TMO=300 run stats_cfg5 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cfg_cfg5_mppi_plan -o t -- python profiles/other_configs.py --only cfg5_mppi_plan --mode device --reps 5
for d in $OUT/cfg_*; do f=$(find $d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -8 "$f" | cut -c1-200 | tee $d.top.csv; done
find $OUT -name "*.csv" -size +1M -delete
echo done
