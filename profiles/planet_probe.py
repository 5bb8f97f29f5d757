"""PlaNet latent rollouts at conf sizes (pop 1000 / 4000, H 12): the STATIC instance against the generic one (HIPETS_PLANET_GENERIC=1), and the
10-iteration clipped-normal CEM plan.  Prints one JSON line."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "mbrl-lib_amd")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import hipets  # noqa: E402

dev = torch.device("cuda:0")
eng = hipets.get_engine(dev)
spec = bench.synthetic_planet_spec(dev)
eng.planet_set_model(spec)
out = {}
l0, b0 = torch.zeros(30, device=dev), torch.zeros(200, device=dev)
for pop in (1000, 4000):
    acts = (torch.rand(pop, 12, 6) * 2 - 1).to(dev)
    for name, env in (("static", None), ("generic", "1")):
        if env:
            os.environ["HIPETS_PLANET_GENERIC"] = env
        else:
            os.environ.pop("HIPETS_PLANET_GENERIC", None)
        t_w = time.perf_counter()
        while time.perf_counter() - t_w < 0.3:
            eng.planet_rollout(acts, l0, b0, 1, seed=1)
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            eng.planet_rollout(acts, l0, b0, 1, seed=1)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 30
        out[f"pop{pop}_{name}"] = {"rollout_ms": ms, "us_per_step": 1e3 * ms / 12, "frac_of_fp32_peak": pop * 12 * spec.flops_per_candidate_step() / (ms * 1e-3) / 157.3e12}
os.environ.pop("HIPETS_PLANET_GENERIC", None)
print(json.dumps(out))
