#!/bin/bash
# round 5, closing session on the shipped library (FAST rows dealt as one run): the ten plan-replay memo entries whose KEY survived the
# change of the row dealing (pets_halfcheetah FAST: workgroup count and schedule coincide, the row -> workgroup map does not) recomputed
# and merged, then the whole GPU suite with the memo on, the bench line, the N = 2 route on one GPU, probes, the rocprofv3 collection.
#   bash profiles/session_r5_final.sh
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r5_final; mkdir -p $OUT
run() { name=$1; shift; echo "== $name: $*" ; ( time timeout ${TMO:-1500} "$@" ) > $OUT/$name.log 2>&1; echo "   rc=$? $(tail -n 3 $OUT/$name.log | tr '\n' ' ' | cut -c1-300)"; }
run smoke python -c "import __graft_entry__ as g; g.smoke()"
if [ -z "${SKIP_MEMO_FIX:-}" ]; then  # (done once, in the session of commit c54828b; later runs of this script: SKIP_MEMO_FIX=1)
HIPETS_ORACLE_CACHE=0 HIPETS_ORACLE_CACHE_OUT=$PWD/gpurun_out/oracle_cache_fix run memo_fix python -m pytest tests/test_gpu_plans_full_size.py -q -p no:cacheprovider -k "test_fused_cem_plan_cfg2_replayed_through_oracle and stock_halfcheetah and fast"
python - <<'PY'
import numpy as np, os
dst = "tests/golden/oracle_cache/plans_full_size.npz"
fix = "gpurun_out/oracle_cache_fix/plans_full_size.npz"
with np.load(dst) as z:
    data = {k: z[k] for k in z.files}
with np.load(fix) as z:
    new = {k: z[k] for k in z.files}
changed = [k for k in new if k in data and not np.array_equal(new[k], data[k])]
print(f"[memo_fix] {len(new)} recomputed entries, {len(changed)} replace a stored value, {len([k for k in new if k not in data])} new keys")
data.update(new)
os.makedirs("gpurun_out/oracle_cache", exist_ok=True)
for p in (dst, "gpurun_out/oracle_cache/plans_full_size.npz"):
    np.savez_compressed(p, **data)
PY
fi
HIPETS_ORACLE_CACHE_OUT=$PWD/gpurun_out/oracle_cache run tests python -m pytest tests -m gpu -q --maxfail=30 -p no:cacheprovider --durations=10
run bench python bench.py
grep -h '"metric"' $OUT/bench.log | tail -1 > $OUT/bench_line.json
HIPETS_DIST_BACKEND=gloo run bench_gloo2 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 10 --warmup 2
grep -h '"metric"' $OUT/bench_gloo2.log | tail -1 > $OUT/bench_line_gloo2.json
run planet python profiles/planet_probe.py
run collect bash profiles/collect.sh r5
echo done
