#!/bin/bash
# pets_inv_pendulum's fused instance (learned reward + a termination function over every state dim, obs 4): the new GPU tests and its
# rollout timings (default, forced generic kernel, every R).  bash profiles/session_r4l.sh
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r4l
mkdir -p $OUT
( time HIPETS_ORACLE_CACHE_OUT=$PWD/gpurun_out/oracle_cache timeout 600 python -m pytest tests/test_gpu_rollout.py -m gpu -q -x -p no:cacheprovider \
    -k "obs4_pop480 or obs4_pop96 or obs3_pop96 or shipped or kernel_class" --durations=8 ) > $OUT/tests.log 2>&1
echo "tests rc=$? $(tail -n 3 $OUT/tests.log | tr '\n' ' ')"
timeout 200 python profiles/stock_workloads.py --only stock_inv_pendulum --sweep-r --generic > $OUT/stock_inv_pendulum.json 2> $OUT/stock_inv_pendulum.err
echo "stock_inv_pendulum rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r4l/stock_inv_pendulum.json"))["stock_inv_pendulum"]
print(d["kernel_class"], {m: {k: (round(v["rollout_kernel_ms"], 4), round(v["frac_of_fp32_peak"], 3)) if "error" not in v else "err" for k, v in d[m].items()} for m in ("device", "fast")},
      {m: (round(v["ms_per_plan"], 3), round(v["kernel_frac_of_fp32_peak"], 3)) for m, v in d.get("cem_plan", {}).items()})
PY
