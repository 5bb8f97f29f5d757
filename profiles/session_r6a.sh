#!/bin/bash
# round 6, session a: the library with (1) mode='device' as the default of the Python layer, (2) FAST member draws made by every
# workgroup in its own prologue (common.hpp fast_member: a keyed bijection of the workgroup indices; no schedule kernel any more),
# (3) hipets_step in FAST mode through the per-step launch form, (4) batched DEVICE-mode plans, (5) ABI v6.  Every FAST entry of the
# oracle memo and EVERY plans entry (new key: the row -> member map; new pin records) is recomputed here and written to
# gpurun_out/oracle_cache for committing.      bash profiles/session_r6a.sh
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r6a; mkdir -p $OUT
run() { name=$1; shift; echo "== $name: $*" ; ( time timeout ${TMO:-2400} "$@" ) > $OUT/$name.log 2>&1; echo "   rc=$? $(tail -n 3 $OUT/$name.log | tr '\n' ' ' | cut -c1-300)"; }
run smoke python -c "import __graft_entry__ as g; g.smoke()"
HIPETS_ORACLE_CACHE_OUT=$PWD/gpurun_out/oracle_cache HIPETS_ORACLE_CACHE_PRUNE=1 run tests python -m pytest tests -m gpu -q --maxfail=40 -p no:cacheprovider --durations=15 ${PYTEST_EXTRA:-}
run bench python bench.py
grep -h '"metric"' $OUT/bench.log | tail -1 > $OUT/bench_line.json
echo done
