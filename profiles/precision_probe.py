"""cfg2 rollouts in one arithmetic mode (f32 | bf16x3) and one randomness mode, for rocprofv3 --pmc passes and quick timing:
    python profiles/precision_probe.py --precision bf16x3 --mode fast --reps 10
Prints one JSON line (average rollout_kernel launch from the library's hipEvents)."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mbrl-lib_amd"))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import hipets  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--precision", default="bf16x3", choices=["f32", "bf16x3"])
ap.add_argument("--mode", default="fast", choices=["fast", "device"])
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--pop", type=int, default=bench.POP)
ap.add_argument("--rows-per-group", type=int, default=0)
a = ap.parse_args()
dev = torch.device("cuda:0")
eng = hipets.get_engine(dev)
eng.set_model(bench.synthetic_spec(dev, precision=a.precision))
g = torch.Generator().manual_seed(0)
actions = (torch.rand(a.pop, bench.HORIZON, bench.ACT, generator=g) * 2 - 1).to(dev)
s0 = np.zeros(bench.OBS, np.float32)
for i in range(3):
    eng.rollout(actions, s0, bench.PARTICLES, mode=a.mode, seed=1, stream_id=i, rows_per_group=a.rows_per_group)
eng.timing_enable(True)
eng.timing_read(reset=True)
for i in range(a.reps):
    eng.rollout(actions, s0, bench.PARTICLES, mode=a.mode, seed=1, stream_id=10 + i, rows_per_group=a.rows_per_group)
n, ms = eng.timing_read(reset=True)
torch.cuda.synchronize()
print(json.dumps({"precision": a.precision, "mode": a.mode, "pop": a.pop, "lib": os.environ.get("HIPETS_LIB", "default"),
                  "launches_per_rollout": n / a.reps, "rollout_kernel_ms": ms / a.reps}))
