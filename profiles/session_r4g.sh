#!/bin/bash
# round 4, final verification: smoke, the whole GPU suite (oracle memo committed), the default bench line
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r4g; mkdir -p $OUT
run() { name=$1; shift; echo "== $name: $*" ; ( time timeout ${TMO:-1500} "$@" ) > $OUT/$name.log 2>&1; echo "   rc=$? $(tail -n 3 $OUT/$name.log | tr '\n' ' ' | cut -c1-300)"; }
run smoke python -c "import __graft_entry__ as g; g.smoke()"
HIPETS_ORACLE_CACHE_OUT=$PWD/gpurun_out/oracle_cache run tests python -m pytest tests -m gpu -q -x -p no:cacheprovider --durations=12
run bench python bench.py
grep -h '"metric"' $OUT/bench.log | tail -1 > $OUT/bench_line.json
echo done
