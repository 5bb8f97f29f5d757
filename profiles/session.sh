#!/bin/bash
# One GPU-box session: smoke, GPU tests (new ones first), kernel-variant timings, microbenchmarks, a short bench line.
# Usage (from the repo root on the box): bash profiles/session.sh <tag> [steps...]; everything lands in gpurun_out/<tag>/
set -u
TAG=${1:-s1}
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p $OUT
run() { name=$1; shift; echo "== $name: $*" ; ( time timeout ${TMO:-900} "$@" ) > $OUT/$name.log 2>&1; echo "   rc=$? $(tail -n 3 $OUT/$name.log | tr '\n' ' ' | cut -c1-300)"; }
run smoke python -c "import __graft_entry__ as g; g.smoke()"
run tests python -m pytest tests -m gpu -q --maxfail=30 -p no:cacheprovider
run variants_default python profiles/kernel_variants.py
run bench python bench.py --steps 20 --warmup 3 --cpu-budget 15
HIPETS_DIST_BACKEND=gloo run bench_gloo2 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 1
run other_configs python profiles/other_configs.py --reps 5
for c in cfg1_cartpole cfg4_humanoid_truncated_obs cfg4_humanoid_v4_obs376 cfg5_cheetah_run planet; do
  run trace_$c rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$c -o t -- python profiles/other_configs.py --only $c --reps 5
done
grep -h '"metric"' $OUT/bench.log | tail -1 > $OUT/bench_line.json
echo done
