#!/bin/bash
# round 6, closing evidence session on the library as shipped: smoke, the WHOLE GPU suite (oracle memo on; new entries, if any, to
# gpurun_out/oracle_cache), the default bench line, the N = 2 route on one GPU (gloo), the phase profile of the shipped one-tile
# instances (LEAN_PROF variant), the turn trace (STEP_TRACE variant), probes, the rocprofv3 / PMC collection.
#   bash profiles/session_r6_final.sh          then, on the build box:  bash profiles/finalize_r6.sh   (summaries + the CPU suite)
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r6_final; mkdir -p $OUT
run() { name=$1; shift; echo "== $name: $*" ; ( time timeout ${TMO:-1800} "$@" ) > $OUT/$name.log 2>&1; echo "   rc=$? $(tail -n 3 $OUT/$name.log | tr '\n' ' ' | cut -c1-300)"; }
run smoke python -c "import __graft_entry__ as g; g.smoke()"
HIPETS_ORACLE_CACHE_OUT=$PWD/gpurun_out/oracle_cache run tests python -m pytest tests -m gpu -q --maxfail=30 -p no:cacheprovider --durations=10
run bench python bench.py
grep -h '"metric"' $OUT/bench.log | tail -1 > $OUT/bench_line.json
HIPETS_DIST_BACKEND=gloo run bench_gloo2 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 10 --warmup 2
grep -h '"metric"' $OUT/bench_gloo2.log | tail -1 > $OUT/bench_line_gloo2.json
HIPETS_LIB=$PWD/profiles/variants/leanprof.so run phase_profile python profiles/one_tile_phase_profile.py
grep -h '^{"lib"' $OUT/phase_profile.log | tail -1 > $OUT/one_tile_phase_profile.json
HIPETS_LIB=$PWD/profiles/variants/steptrace.so run turn_trace python profiles/turn_trace.py
HIPETS_LIB=$PWD/profiles/variants/steptrace.so TRACE_CASES=cfg4p_pop497:376:2:2:497,cfg4p_pop1001:376:2:3:1001 run turn_trace_ragged python profiles/turn_trace.py
run cfg4p_iterations python profiles/cfg4p_iteration_probe.py
grep -h '^{"obs"' $OUT/cfg4p_iterations.log | tail -1 > $OUT/cfg4p_iterations.json
run small_batches python profiles/small_batch_probe.py
for i in 1 2 3 4 5; do run headline_$i python profiles/headline_probe.py; done
TMO=60 run pair_exchange profiles/microbench/pair_exchange
run collect bash profiles/collect.sh r6
echo done
