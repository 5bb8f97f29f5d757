"""cfg4' (Humanoid-v4, obs 376) rollout timing, DEVICE and FAST, for hand-over experiments (HIPETS_LIB selects the build)."""
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
out = {"lib": hipets.LIB_PATH}
for mode in ("device", "fast"):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.4:
        eng.rollout(acts, s0, P, mode=mode, seed=1)
        torch.cuda.synchronize()
    eng.timing_enable(True)
    eng.timing_read(reset=True)
    n = 8
    for i in range(n):
        eng.rollout(acts, s0, P, mode=mode, seed=1, stream_id=i)
    torch.cuda.synchronize()
    nl, kms = eng.timing_read(reset=True)
    eng.timing_enable(False)
    out[mode] = {"rollout_kernel_ms": kms / n, "launches_per_rollout": nl / n, "frac_of_fp32_peak": pop * P * H * spec.flops_per_candidate_step() / (kms / n * 1e-3) / 157.3e12}
print(json.dumps(out))
