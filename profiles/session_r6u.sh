#!/bin/bash
# round 6, session u: which of the two refit changes of session t costs time -- kernel statistics of the bench command on the four builds
set -u
export TMPDIR=/tmp
OUT=gpurun_out/${SESSION:-r6u}; mkdir -p $OUT
run() { name=$1; shift; echo "== $name: $*" ; ( time timeout ${TMO:-1500} "$@" ) > $OUT/$name.log 2>&1; echo "   rc=$? $(tail -n 3 $OUT/$name.log | tr '\n' ' ' | cut -c1-300)"; }
for rep in 1 2; do
for v in m0r0 m1r0 m0r1 m1r1; do
  if [ $v = m1r1 ]; then unset HIPETS_LIB; else export HIPETS_LIB=$PWD/profiles/variants/$v.so; fi
  TMO=200 run ${v}_$rep rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${v}_$rep -o t -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras
  f=$(find $OUT/${v}_$rep -name "*kernel_stats.csv" | head -1); grep -h "cem_refit" "$f" | cut -c100-170
done
done
unset HIPETS_LIB
find $OUT -name "*.csv" -size +1M -delete
echo done
