#!/bin/bash
# round 6, session h: the WIDE instances hand rows over as plain rows + one tagged record per row ("row flags"): parity (bitwise against
# the generic kernel, persistent == per-step incl. turn-based sizes, oracle replays, the fused iCEM plan replay, the time-out path),
# turn trace, per population size against the tagged-pair build (-DHIPETS_WIDE_ROWFLAGS=0) on the same box.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/${SESSION:-r6h}; mkdir -p $OUT
run() { name=$1; shift; echo "== $name: $*" ; ( time timeout ${TMO:-600} "$@" ) > $OUT/$name.log 2>&1; echo "   rc=$? $(tail -n 3 $OUT/$name.log | tr '\n' ' ' | cut -c1-300)"; }
run tests_wide python -m pytest tests/test_gpu_rollout.py tests/test_gpu_device_mode.py tests/test_gpu_plans_full_size.py -q -p no:cacheprovider --durations=5 -k "wide or 376 or cfg4p or humanoid or persistent or timed_out"
HIPETS_LIB=$PWD/profiles/variants/steptrace.so run turn_trace python profiles/turn_trace.py
run cfg4p_iterations python profiles/cfg4p_iteration_probe.py
grep -h '^{"obs"' $OUT/cfg4p_iterations.log | tail -1 > $OUT/cfg4p_iterations.json
HIPETS_LIB=$PWD/profiles/variants/norowflags.so run cfg4p_iterations_norowflags python profiles/cfg4p_iteration_probe.py
grep -h '^{"obs"' $OUT/cfg4p_iterations_norowflags.log | tail -1 > $OUT/cfg4p_iterations_norowflags.json
echo done
