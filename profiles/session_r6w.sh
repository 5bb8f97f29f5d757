#!/bin/bash
# round 6, session w: the narrow form of the PlaNet rollout (planet_narrow.hpp: four rows per workgroup, linear ops as packed FMAs) --
# parity (oracle, reference goldens, the 16-row form), timing of the shipped planner's batch in both forms, kernel statistics of the plan
set -u
export TMPDIR=/tmp
OUT=gpurun_out/${SESSION:-r6w}; mkdir -p $OUT
run() { name=$1; shift; echo "== $name: $*" ; ( time timeout ${TMO:-900} "$@" ) > $OUT/$name.log 2>&1; echo "   rc=$? $(tail -n 3 $OUT/$name.log | tr '\n' ' ' | cut -c1-300)"; }
run tests_planet python -m pytest tests/test_gpu_planet.py -q -p no:cacheprovider --durations=5
for form in 1 0; do
  HIPETS_PLANET_NARROW=$form TMO=300 run planet_form$form python profiles/other_configs.py --only planet --reps 6
  grep -h "rollout_ms\|ms_per_plan\|frac" $OUT/planet_form$form.log | tr -d '\n' | cut -c1-400; echo
done
TMO=300 run stats_planet rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cfg_planet -o t -- python profiles/other_configs.py --only planet --mode device --reps 5
f=$(find $OUT/cfg_planet -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -6 "$f" | cut -c1-200
find $OUT -name "*.csv" -size +1M -delete
echo done
