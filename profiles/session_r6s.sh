#!/bin/bash
# round 6, session s: (1) where the refit's time goes at cfg2 -- timing-only variants without the particle means / the selection / the sweeps
# (-DHIPETS_REFIT_SKIP=1 / 2 / 4 / 7), kernel statistics of the bench command; (2) the whole GPU suite with the oracle memo OFF
set -u
export TMPDIR=/tmp
OUT=gpurun_out/${SESSION:-r6s}; mkdir -p $OUT
run() { name=$1; shift; echo "== $name: $*" ; ( time timeout ${TMO:-2400} "$@" ) > $OUT/$name.log 2>&1; echo "   rc=$? $(tail -n 3 $OUT/$name.log | tr '\n' ' ' | cut -c1-300)"; }
for v in 0 1 2 4 7; do
  if [ $v = 0 ]; then unset HIPETS_LIB; else export HIPETS_LIB=$PWD/profiles/variants/refitskip$v.so; fi
  TMO=200 run refit_$v rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/refit_$v -o t -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras
  f=$(find $OUT/refit_$v -name "*kernel_stats.csv" | head -1); grep -h "cem_refit" "$f" | cut -c1-160
done
unset HIPETS_LIB
find $OUT -name "*.csv" -size +1M -delete
HIPETS_ORACLE_CACHE=0 run tests_nocache python -m pytest tests -m gpu -q -p no:cacheprovider --durations=10 --ignore=tests/test_oracle_memo_pinned.py
echo done
