#!/bin/bash
# round 6, session p: kernel statistics of the plans only (optimizer kernels), no tests
set -u
export TMPDIR=/tmp
OUT=gpurun_out/${SESSION:-r6p}; mkdir -p $OUT
run() { name=$1; shift; echo "== $name: $*" ; ( time timeout ${TMO:-900} "$@" ) > $OUT/$name.log 2>&1; echo "   rc=$? $(tail -n 3 $OUT/$name.log | tr '\n' ' ' | cut -c1-300)"; }
for c in cfg1_cem_plan cfg4_icem_plan cfg5_mppi_plan planet; do
  TMO=300 run stats_$c rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cfg_$c -o t -- python profiles/other_configs.py --only $c --mode device --reps 5
done
TMO=300 run stats_bench rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_device -o bench -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras
run bench python bench.py --no-cpu-baseline --no-extras
grep -h '"metric"' $OUT/bench.log | tail -1 > $OUT/bench_line.json
find $OUT -name "*.csv" -size +1M -delete
echo done
