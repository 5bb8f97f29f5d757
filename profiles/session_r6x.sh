#!/bin/bash
# round 6, session x: where the narrow PlaNet form's step goes -- timing-only variants (no weight stream / a quarter of the vector work / both)
set -u
export TMPDIR=/tmp
OUT=gpurun_out/${SESSION:-r6x}; mkdir -p $OUT
run() { name=$1; shift; echo "== $name: $*" ; ( time timeout ${TMO:-300} "$@" ) > $OUT/$name.log 2>&1; echo "   rc=$? $(tail -n 3 $OUT/$name.log | tr '\n' ' ' | cut -c1-200)"; }
for v in shipped narrow_noload narrow_nofma narrow_neither; do
  if [ $v = shipped ]; then unset HIPETS_LIB; else export HIPETS_LIB=$PWD/profiles/variants/$v.so; fi
  HIPETS_PLANET_NARROW=1 run $v python profiles/other_configs.py --only planet --reps 6
  grep -h "rollout_ms" $OUT/$v.log | tr -d '\n' | cut -c1-200; echo
done
echo done
