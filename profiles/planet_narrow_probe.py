"""PlaNet rollouts (conf/dynamics_model/planet.yaml sizes, H 12) over population sizes in both kernel forms: the 16-row MFMA tiles of planet.hpp
(HIPETS_PLANET_NARROW=0) and the four-row packed-FMA form of the round-6 experiment (=1; profiles/experiments/r6_planet_narrow.patch).  us per step
and the weight bytes a step streams per workgroup and chip-wide.  Run on a GPU box from the repo root."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "mbrl-lib_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import hipets  # noqa: E402
from oracle import planet_oracle as pl  # noqa: E402  (only to BUILD random weights)

dev = torch.device("cuda:0")
eng = hipets.get_engine(dev)
pm = pl.make_synthetic_planet(30, 6, 200, 200, seed=0)
spec = hipets.PlaNetSpec(**{k: getattr(pm, k) for k in pl.PLANET_TENSORS}, min_std=pm.min_std)
eng.planet_set_model(spec)
H = 12
out = {}
for form in ("1", "0"):
    os.environ["HIPETS_PLANET_NARROW"] = form
    for pop in (64, 125, 250, 500, 1000, 2000, 4000):
        acts = (torch.rand(pop, H, 6) * 2 - 1).to(dev)
        l0, b0 = torch.zeros(30, device=dev), torch.zeros(200, device=dev)
        fn = lambda: eng.planet_rollout(acts, l0, b0, 1, seed=1)
        t_w = time.perf_counter()
        while time.perf_counter() - t_w < 0.3:
            fn()
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
        rows = 4 if form == "1" else 16
        out.setdefault("narrow" if form == "1" else "tiles", {})[str(pop)] = {"rollout_ms": round(1e3 * dt, 4), "us_per_step": round(1e6 * dt / H, 2), "workgroups": -(-pop // rows)}
print(json.dumps(out))
