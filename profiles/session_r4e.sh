#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r4e; mkdir -p $OUT
run() { name=$1; shift; echo "== $name: $*" ; ( time timeout ${TMO:-900} "$@" ) > $OUT/$name.log 2>&1; echo "   rc=$? $(tail -n 3 $OUT/$name.log | tr '\n' ' ' | cut -c1-300)"; }
run probe python profiles/plan_gap_probe.py
HIPETS_NO_STREAM_SCOPE=1 run probe_noscope python profiles/plan_gap_probe.py
run debug_tests python -m pytest -m gpu -q -p no:cacheprovider tests/test_gpu_debug_build.py
echo done
