#!/bin/bash
# round 6, session b: (1) the new bitwise matrix (every shipped DEVICE instance, persistent == per-step at one and two workgroups per CU),
# the debug / ASan build (line tables only: 17 MB each), the PlaNet tests on the v6 opts struct; (2) the phase profile of the SHIPPED
# one-tile instances from a -DHIPETS_LEAN_PROF=1 build (profiles/variants/leanprof.so, built by profiles/build_variant.py);
# (3) kernel traces of a cfg1 CEM plan and of the PlaNet plan: how much of a small plan's wall time lies BETWEEN its kernels.
#     bash profiles/session_r6b.sh
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r6b; mkdir -p $OUT
run() { name=$1; shift; echo "== $name: $*" ; ( time timeout ${TMO:-1200} "$@" ) > $OUT/$name.log 2>&1; echo "   rc=$? $(tail -n 3 $OUT/$name.log | tr '\n' ' ' | cut -c1-300)"; }
run smoke python -c "import __graft_entry__ as g; g.smoke()"
run tests python -m pytest tests/test_gpu_device_mode.py tests/test_gpu_debug_build.py tests/test_gpu_planet.py tests/test_gpu_c_abi.py -q -p no:cacheprovider --durations=8
HIPETS_LIB=$PWD/profiles/variants/leanprof.so run phase_profile python profiles/one_tile_phase_profile.py
grep -h '^{"lib"' $OUT/phase_profile.log | tail -1 > $OUT/one_tile_phase_profile.json
run small_batches python profiles/small_batch_probe.py
grep -h '^{"lib"' $OUT/small_batches.log | tail -1 > $OUT/small_batches.json
run plan_gaps python profiles/plan_gap_probe.py
for c in cfg1_cem_plan planet; do
  for MODE in device fast; do
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_${c}_$MODE -o t -- python profiles/other_configs.py --only $c --mode $MODE --reps 20 > $OUT/trace_${c}_$MODE.log 2>&1
  done
done
find $OUT -name "*_kernel_stats.csv" | head
find $OUT -name "*.csv" -size +2M -delete
echo done
