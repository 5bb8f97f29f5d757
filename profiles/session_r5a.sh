#!/bin/bash
# round 5, first GPU session: the k-split of one-tile workgroups (KSpec::KSPLIT) -- whole GPU suite (memo on; new entries written out),
# the one-tile probe on the shipped library and on the -DHIPETS_KSPLIT=0 variant (same box), the default bench line
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r5a; mkdir -p $OUT
run() { name=$1; shift; echo "== $name: $*" ; ( time timeout ${TMO:-1500} "$@" ) > $OUT/$name.log 2>&1; echo "   rc=$? $(tail -n 3 $OUT/$name.log | tr '\n' ' ' | cut -c1-300)"; }
run smoke python -c "import __graft_entry__ as g; g.smoke()"
run small_ks python profiles/small_batch_probe.py
HIPETS_LIB=$PWD/profiles/variants/noks.so run small_noks python profiles/small_batch_probe.py
HIPETS_ORACLE_CACHE_OUT=$PWD/gpurun_out/oracle_cache run tests python -m pytest tests -m gpu -q --maxfail=80 -p no:cacheprovider --durations=12
run bench python bench.py
grep -h '"metric"' $OUT/bench.log | tail -1 > $OUT/bench_line.json
echo done
