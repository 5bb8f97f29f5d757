"""cfg2 (BASELINE configs[1]) rollout-kernel time per mode: median of `--reps` blocks of 20 launches each.  HIPETS_LIB selects a variant
build (profiles/build_variant.py), e.g. the timing-only -DHIPETS_TIMING_NO_DRAWS=1 build that bounds what taking the Philox draws off the
step's critical path could gain.  Prints one JSON line."""
import argparse
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mbrl-lib_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import hipets  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=5)
args = ap.parse_args()
dev = torch.device("cuda:0")
eng = hipets.get_engine(dev)
spec = bench.synthetic_spec(dev)
eng.set_model(spec)
pop, P, H = 500, 20, 30
acts = (torch.rand(pop, H, spec.act_dim, generator=torch.Generator().manual_seed(0)) * 2 - 1).to(dev)
s0 = (np.random.default_rng(0).standard_normal(spec.obs_dim) * 0.1).astype(np.float32)
out = {"lib": os.environ.get("HIPETS_LIB", "default"), "workload": "cfg2: pop 500 x 20 particles x H 30, one rollout per launch"}
for mode in ("device", "fast"):
    for i in range(30):
        eng.rollout(acts, s0, P, mode=mode, seed=1, stream_id=i)
    torch.cuda.synchronize()
    meds = []
    for r in range(args.reps):
        eng.timing_enable(True)
        eng.timing_read(reset=True)
        for i in range(20):
            eng.rollout(acts, s0, P, mode=mode, seed=1, stream_id=100 + 20 * r + i)
        torch.cuda.synchronize()
        n, ms = eng.timing_read(reset=True)
        eng.timing_enable(False)
        meds.append(ms / n)
    out[mode] = {"rollout_kernel_ms_per_block": [round(m, 5) for m in meds], "median_ms": statistics.median(meds),
                 "frac_of_fp32_peak": pop * P * H * spec.flops_per_candidate_step() / (statistics.median(meds) * 1e-3) / 157.3e12}
out["device_minus_fast_pct"] = 100.0 * (out["device"]["median_ms"] / out["fast"]["median_ms"] - 1.0)
print(json.dumps(out))
