#!/bin/bash
# round 6, session k: step trace of the ragged last turn (cfg4' pop 497: a two-tile turn + a one-tile turn per step) against two-tile turns
set -u
export TMPDIR=/tmp
OUT=gpurun_out/${SESSION:-r6k}; mkdir -p $OUT
run() { name=$1; shift; echo "== $name: $*" ; ( time timeout ${TMO:-600} "$@" ) > $OUT/$name.log 2>&1; echo "   rc=$? $(tail -n 3 $OUT/$name.log | tr '\n' ' ' | cut -c1-300)"; }
export HIPETS_LIB=$PWD/profiles/variants/steptrace.so
TRACE_CASES=cfg4p_pop497:376:2:2:497,cfg4p_pop1001:376:2:3:1001 run turn_trace_ragged python profiles/turn_trace.py
HIPETS_RAGGED_LAST_TURN=0 TRACE_CASES=cfg4p_pop497:376:2:2:497,cfg4p_pop1001:376:2:3:1001 run turn_trace_two_tile python profiles/turn_trace.py
echo done
