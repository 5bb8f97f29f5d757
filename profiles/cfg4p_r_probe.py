"""cfg4' (Humanoid-v4, obs 376) DEVICE / FAST rollouts with forced row-tile counts: the WIDE instance with two row tiles (one workgroup
per CU, three turns per step) against ONE row tile.  Written for an EXPERIMENT build (normaliser tables in
global memory so that two one-tile workgroups fit a CU; HIPETS_TURNS_ANY=1 let the turn-based persistent form run two to a CU); with the
library as shipped it measures the one-workgroup-per-CU forms.  Results: profiles/r4_co_residency_experiments.json."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mbrl-lib_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import hipets  # noqa: E402

dev = torch.device("cuda:0")
eng = hipets.get_engine(dev)
spec = bench.synthetic_spec(dev, obs=376, act=17, ensemble=7, elite=[0, 1, 2, 3, 4], termination="humanoid")
eng.set_model(spec)
pop, P, H = 1036, 20, 40
acts = (torch.rand(pop, H, 17) * 2 - 1).to(dev)
s0 = np.zeros(376, np.float32)
s0[0] = 1.4
out = {"turns_any": os.environ.get("HIPETS_TURNS_ANY")}
ref = {}
for mode in ("device", "fast"):
    for R in (0, 1, 2):
        f = lambda i=0: eng.rollout(acts, s0, P, mode=mode, seed=1, stream_id=i, rows_per_group=R)  # noqa: E731
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.4:
            r = f()
            torch.cuda.synchronize()
        if mode == "device":
            ref.setdefault("v", r.clone())
            assert torch.equal(r, ref["v"]), "DEVICE results depend on the geometry?!"
        eng.timing_enable(True)
        eng.timing_read(reset=True)
        n = 6
        for i in range(n):
            f(i)
        torch.cuda.synchronize()
        nl, kms = eng.timing_read(reset=True)
        eng.timing_enable(False)
        out[f"{mode}_R{R}"] = {"rollout_kernel_ms": round(kms / n, 4), "launches_per_rollout": nl / n,
                               "frac_of_fp32_peak": round(pop * P * H * spec.flops_per_candidate_step() / (kms / n * 1e-3) / 157.3e12, 4)}
print(json.dumps(out))
