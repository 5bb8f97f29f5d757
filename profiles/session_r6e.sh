#!/bin/bash
# round 6, session e: the LDS-DMA collect rewritten (row slots dealt to the waves, no per-chunk row lookup): parity, turn trace, per
# population size against the no-DMA build; the pair-exchange microbenchmark (bounded).
set -u
export TMPDIR=/tmp
OUT=gpurun_out/${SESSION:-r6e}; mkdir -p $OUT
run() { name=$1; shift; echo "== $name: $*" ; ( time timeout ${TMO:-600} "$@" ) > $OUT/$name.log 2>&1; echo "   rc=$? $(tail -n 3 $OUT/$name.log | tr '\n' ' ' | cut -c1-300)"; }
run tests_wide python -m pytest tests/test_gpu_rollout.py tests/test_gpu_device_mode.py tests/test_gpu_plans_full_size.py -q -p no:cacheprovider --durations=5 -k "wide or 376 or cfg4p or humanoid or persistent"
HIPETS_LIB=$PWD/profiles/variants/steptrace.so run turn_trace_dma python profiles/turn_trace.py
run cfg4p_iterations python profiles/cfg4p_iteration_probe.py
grep -h '^{"obs"' $OUT/cfg4p_iterations.log | tail -1 > $OUT/cfg4p_iterations.json
# the same probe without the by-column input pass of the WIDE instances (-DHIPETS_INPUT_BY_COLUMN=0), same box
HIPETS_LIB=$PWD/profiles/variants/nocols.so run cfg4p_iterations_nocols python profiles/cfg4p_iteration_probe.py
grep -h '^{"obs"' $OUT/cfg4p_iterations_nocols.log | tail -1 > $OUT/cfg4p_iterations_nocols.json
TMO=60 run pair_exchange profiles/microbench/pair_exchange
echo done
