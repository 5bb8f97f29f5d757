#!/bin/bash
# round 6, session t: the refit with its particle totals in one round trip and the rank counting four keys to an LDS read --
# the planning / full-size plan / batched suites (memo on: the plans must keep their bits) and the kernel statistics of the bench command
set -u
export TMPDIR=/tmp
OUT=gpurun_out/${SESSION:-r6t}; mkdir -p $OUT
run() { name=$1; shift; echo "== $name: $*" ; ( time timeout ${TMO:-1500} "$@" ) > $OUT/$name.log 2>&1; echo "   rc=$? $(tail -n 3 $OUT/$name.log | tr '\n' ' ' | cut -c1-300)"; }
run tests python -m pytest tests/test_gpu_planning.py tests/test_gpu_plans_full_size.py tests/test_gpu_batched_plans.py tests/test_gpu_planet.py -m gpu -q -p no:cacheprovider
TMO=200 run stats rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o t -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras
f=$(find $OUT/stats -name "*kernel_stats.csv" | head -1); cut -c1-200 "$f" | head -12
find $OUT -name "*.csv" -size +1M -delete
run bench python bench.py --no-cpu-baseline
grep -h '"metric"' $OUT/bench.log | tail -1 | cut -c1-600
echo done
