#!/bin/bash
# round 6, session c: the WIDE instances collect a turn's rows by LDS-DMA (rollout.hpp dma_collect).  Parity of everything that runs
# them (bitwise against the generic kernel and persistent == per-step incl. the turn-based sizes, oracle replays at cfg4' size, the
# fused iCEM plan replay), then the measurements: per population size of the cfg4' plan, the turn trace, the bench's other_configs;
# and the phase profile of the shipped one-tile instances (LEAN_PROF variant; PlaNet's accumulators moved to dynamic LDS).
#     bash profiles/session_r6c.sh
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r6c; mkdir -p $OUT
run() { name=$1; shift; echo "== $name: $*" ; ( time timeout ${TMO:-1500} "$@" ) > $OUT/$name.log 2>&1; echo "   rc=$? $(tail -n 3 $OUT/$name.log | tr '\n' ' ' | cut -c1-300)"; }
run smoke python -c "import __graft_entry__ as g; g.smoke()"
run tests_wide python -m pytest tests/test_gpu_rollout.py tests/test_gpu_device_mode.py tests/test_gpu_plans_full_size.py -q -p no:cacheprovider --durations=8 -k "wide or 376 or cfg4p or humanoid or persistent"
HIPETS_LIB=$PWD/profiles/variants/leanprof.so run phase_profile python profiles/one_tile_phase_profile.py
grep -h '^{"lib"' $OUT/phase_profile.log | tail -1 > $OUT/one_tile_phase_profile.json
run cfg4p_iterations python profiles/cfg4p_iteration_probe.py
grep -h '^{"obs"' $OUT/cfg4p_iterations.log | tail -1 > $OUT/cfg4p_iterations.json
# the same probe on a build WITHOUT the DMA collect (-DHIPETS_DMA_COLLECT=0: the round-5 register path), same box
HIPETS_LIB=$PWD/profiles/variants/nodma.so run cfg4p_iterations_nodma python profiles/cfg4p_iteration_probe.py
grep -h '^{"obs"' $OUT/cfg4p_iterations_nodma.log | tail -1 > $OUT/cfg4p_iterations_nodma.json
HIPETS_LIB=$PWD/profiles/variants/steptrace.so run turn_trace python profiles/turn_trace.py
run bench python bench.py
grep -h '"metric"' $OUT/bench.log | tail -1 > $OUT/bench_line.json
echo done
