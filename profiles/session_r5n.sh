#!/bin/bash
# round 5, FAST rows dealt as one run + WIDE cost model: the whole GPU suite (memo on: the FAST entries miss and are re-recorded), the cfg4'
# per-iteration probe, the shipped workloads with every forced R, the bench line.   bash profiles/session_r5n.sh
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r5n; mkdir -p $OUT
run() { name=$1; shift; echo "== $name: $*" ; ( time timeout ${TMO:-1500} "$@" ) > $OUT/$name.log 2>&1; echo "   rc=$? $(tail -n 3 $OUT/$name.log | tr '\n' ' ' | cut -c1-300)"; }
run smoke python -c "import __graft_entry__ as g; g.smoke()"
HIPETS_ORACLE_CACHE_PRUNE=1 HIPETS_ORACLE_CACHE_OUT=$PWD/gpurun_out/oracle_cache run tests python -m pytest tests -m gpu -q --maxfail=30 -p no:cacheprovider --durations=10
run cfg4p python profiles/cfg4p_iteration_probe.py
run stock python profiles/stock_workloads.py --sweep-r
run bench python bench.py
grep -h '"metric"' $OUT/bench.log | tail -1 > $OUT/bench_line.json
echo done
