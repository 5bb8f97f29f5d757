#!/bin/bash
# pets_hopper's FAST fused instance (all-dims termination through per-row LDS flags): closed-form tests incl. the new fused combos, the
# bitwise cases, the class table; rollout timings.  bash profiles/session_r4s.sh
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r4s
mkdir -p $OUT
( time timeout 500 python -m pytest tests/test_gpu_closed_forms.py tests/test_gpu_rollout.py -m gpu -q --maxfail=10 -p no:cacheprovider \
    -k "closed_forms or obs11 or obs10 or shipped or kernel_class or obs4_pop96 or obs3_pop96" --durations=5 ) > $OUT/tests.log 2>&1
echo "tests rc=$? $(tail -n 3 $OUT/tests.log | tr '\n' ' ')"
grep -n "FAILED\|^E  " $OUT/tests.log | head -20
timeout 200 python profiles/stock_workloads.py --only stock_hopper --sweep-r --generic > $OUT/stock_hopper.json 2> $OUT/stock_hopper.err
echo "stock_hopper rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r4s/stock_hopper.json"))["stock_hopper"]
print(d["kernel_class"], {m: {k: (round(v["rollout_kernel_ms"], 4), round(v["frac_of_fp32_peak"], 3)) if "error" not in v else "err" for k, v in d[m].items()} for m in ("device", "fast")},
      {m: (round(v["ms_per_plan"], 3), round(v["kernel_frac_of_fp32_peak"], 3)) for m, v in d.get("cem_plan", {}).items()})
PY
