#!/bin/bash
# Learned-reward fused instances + hipets_kernel_class: the new GPU tests, then the rollout / plan timings of the shipped
# learned-reward workloads (default instance, forced generic kernel, every R).  bash profiles/session_r4k.sh
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r4k
mkdir -p $OUT
( time HIPETS_ORACLE_CACHE_OUT=$PWD/gpurun_out/oracle_cache timeout 600 python -m pytest tests/test_gpu_rollout.py tests/test_gpu_plans_full_size.py -m gpu -q -x -p no:cacheprovider \
    -k "obs20 or obs18 or obs17_pop64 or obs16 or obs23 or shipped or kernel_class or stock_pusher or stock_mppi or obs17_pop40" --durations=8 ) > $OUT/tests.log 2>&1
echo "tests rc=$? $(tail -n 3 $OUT/tests.log | tr '\n' ' ')"
for W in stock_pusher stock_reacher stock_mppi_halfcheetah_model; do
    timeout 200 python profiles/stock_workloads.py --only $W --sweep-r --generic > $OUT/$W.json 2> $OUT/$W.err
    echo "$W rc=$?"
done
python - <<'PY'
import json
for w in ("stock_pusher", "stock_reacher", "stock_mppi_halfcheetah_model"):
    try:
        d = json.load(open(f"gpurun_out/r4k/{w}.json"))[w]
    except Exception as exc:
        print(w, "unreadable", exc); continue
    print(w, d["kernel_class"], {m: {k: (round(v["rollout_kernel_ms"], 4), round(v["frac_of_fp32_peak"], 3)) if "error" not in v else "err" for k, v in d[m].items()} for m in ("device", "fast")},
          {m: (round(v["ms_per_plan"], 3), round(v["kernel_frac_of_fp32_peak"], 3)) for m, v in d.get("cem_plan", {}).items()})
PY
