#!/bin/bash
# round 6, session j: the ragged last turn of the WIDE two-tile instance (one-tile logical workgroups, R = 1 op bodies): parity,
# and the five cfg4' population sizes with / without it (HIPETS_RAGGED_LAST_TURN=0) on the same box.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/${SESSION:-r6j}; mkdir -p $OUT
run() { name=$1; shift; echo "== $name: $*" ; ( time timeout ${TMO:-600} "$@" ) > $OUT/$name.log 2>&1; echo "   rc=$? $(tail -n 3 $OUT/$name.log | tr '\n' ' ' | cut -c1-300)"; }
run tests_wide python -m pytest tests/test_gpu_rollout.py tests/test_gpu_device_mode.py tests/test_gpu_plans_full_size.py -q -p no:cacheprovider --durations=5 -k "wide or 376 or cfg4p or humanoid or persistent or ragged"
PROBE_RS=0 run cfg4p_iterations python profiles/cfg4p_iteration_probe.py
grep -h '^{"obs"' $OUT/cfg4p_iterations.log | tail -1 > $OUT/cfg4p_iterations.json
HIPETS_RAGGED_LAST_TURN=0 PROBE_RS=0 run cfg4p_iterations_two_tile_turns python profiles/cfg4p_iteration_probe.py
grep -h '^{"obs"' $OUT/cfg4p_iterations_two_tile_turns.log | tail -1 > $OUT/cfg4p_iterations_two_tile_turns.json
PROBE_RS=0 PROBE_POPS=497,520,609,1001,1024 run ragged_sizes python profiles/cfg4p_iteration_probe.py
HIPETS_RAGGED_LAST_TURN=0 PROBE_RS=0 PROBE_POPS=497,520,609,1001,1024 run ragged_sizes_two_tile_turns python profiles/cfg4p_iteration_probe.py
echo done
