#!/bin/bash
# round 6, session d: measurements that decide carried items.
#  (1) turn trace of the turn-based DEVICE form (cfg4', cfg4) with and without the LDS-DMA collect (STEP_TRACE builds incl. hipets.hip)
#  (2) headline: cfg2 rollout kernel, shipped library vs the timing-only -DHIPETS_TIMING_NO_DRAWS=1 build (upper bound of what taking
#      the Philox draws off the step's critical path can gain), five blocks each
#  (3) profiles/microbench/pair_exchange: one tagged exchange between two workgroups, same / other XCD
#  (4) the phase profile of the shipped one-tile instances (kernel-only durations)
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r6d; mkdir -p $OUT
run() { name=$1; shift; echo "== $name: $*" ; ( time timeout ${TMO:-900} "$@" ) > $OUT/$name.log 2>&1; echo "   rc=$? $(tail -n 3 $OUT/$name.log | tr '\n' ' ' | cut -c1-300)"; }
HIPETS_LIB=$PWD/profiles/variants/steptrace.so run turn_trace_dma python profiles/turn_trace.py
HIPETS_LIB=$PWD/profiles/variants/steptrace_nodma.so run turn_trace_nodma python profiles/turn_trace.py
for i in 1 2 3; do
run headline_$i python profiles/headline_probe.py
HIPETS_LIB=$PWD/profiles/variants/nodraws.so run headline_nodraws_$i python profiles/headline_probe.py
done
run pair_exchange profiles/microbench/pair_exchange
HIPETS_LIB=$PWD/profiles/variants/leanprof.so run phase_profile python profiles/one_tile_phase_profile.py
grep -h '^{"lib"' $OUT/phase_profile.log | tail -1 > $OUT/one_tile_phase_profile.json
echo done
