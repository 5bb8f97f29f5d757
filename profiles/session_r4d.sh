#!/bin/bash
# round 4, session d: the whole GPU suite with per-test durations; the oracle memo's new entries are written for committing; bench line
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r4d; mkdir -p $OUT
export HIPETS_ORACLE_CACHE_OUT=$PWD/gpurun_out/oracle_cache
run() { name=$1; shift; echo "== $name: $*" ; ( time timeout ${TMO:-1500} "$@" ) > $OUT/$name.log 2>&1; echo "   rc=$? $(tail -n 3 $OUT/$name.log | tr '\n' ' ' | cut -c1-300)"; }
run gpu_tests python -m pytest tests -m gpu -q -p no:cacheprovider --durations=60
run bench python bench.py
grep -h '^{' $OUT/bench.log | tail -1 > $OUT/bench_line.json
echo done
