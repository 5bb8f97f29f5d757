#!/bin/bash
# round 5, evidence session on the shipped library: smoke, the WHOLE GPU suite with the oracle memo OFF (every full-size replay runs
# the oracle itself: HIPETS_ORACLE_CACHE=0 -- the memo's entries are rewritten from this run), the default bench line, the N = 2 path on
# one GPU (gloo), the rocprofv3 collection.   bash profiles/session_r5_evidence.sh
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r5_evidence; mkdir -p $OUT
run() { name=$1; shift; echo "== $name: $*" ; ( time timeout ${TMO:-1800} "$@" ) > $OUT/$name.log 2>&1; echo "   rc=$? $(tail -n 3 $OUT/$name.log | tr '\n' ' ' | cut -c1-300)"; }
run smoke python -c "import __graft_entry__ as g; g.smoke()"
HIPETS_ORACLE_CACHE=0 HIPETS_ORACLE_CACHE_OUT=$PWD/gpurun_out/oracle_cache run tests_nocache python -m pytest tests -m gpu -q --maxfail=30 -p no:cacheprovider --durations=15
run bench python bench.py
grep -h '"metric"' $OUT/bench.log | tail -1 > $OUT/bench_line.json
HIPETS_DIST_BACKEND=gloo run bench_gloo2 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 10 --warmup 2
grep -h '"metric"' $OUT/bench_gloo2.log | tail -1 > $OUT/bench_line_gloo2.json
run planet python profiles/planet_probe.py
run collect bash profiles/collect.sh r5
echo done
