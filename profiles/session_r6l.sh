#!/bin/bash
# round 6, session l: optimizer kernels (coloured-noise sampler with its coefficients in registers, refit dealt over workgroups with the
# elites' values held in registers, MPPI update with chunked loads): parity of everything that plans, kernel statistics of the plans.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/${SESSION:-r6l}; mkdir -p $OUT
run() { name=$1; shift; echo "== $name: $*" ; ( time timeout ${TMO:-900} "$@" ) > $OUT/$name.log 2>&1; echo "   rc=$? $(tail -n 3 $OUT/$name.log | tr '\n' ' ' | cut -c1-300)"; }
run tests_plans python -m pytest tests/test_gpu_planning.py tests/test_gpu_plans_full_size.py tests/test_gpu_batched_plans.py tests/test_gpu_planet.py tests/test_gpu_sharded_world.py tests/test_gpu_closed_loop.py tests/test_gpu_c_abi.py tests/test_gpu_persistent_collective.py -q -p no:cacheprovider --durations=8
for c in cfg1_cem_plan cfg4_icem_plan cfg5_mppi_plan planet; do
  TMO=300 run stats_$c rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cfg_$c -o t -- python profiles/other_configs.py --only $c --mode device --reps 5
done
TMO=300 run stats_bench rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_device -o bench -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras
run bench python bench.py --no-cpu-baseline
grep -h '"metric"' $OUT/bench.log | tail -1 > $OUT/bench_line.json
for d in $OUT/cfg_* $OUT/trace_device; do f=$(find $d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -8 "$f" | cut -c1-200 > $d.top.csv; done
find $OUT -name "*.csv" -size +1M -delete
echo done
