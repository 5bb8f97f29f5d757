"""Small-batch timings (one-tile workgroups, fewer than CUs): cfg1 cartpole, a rank's shard of an 8-way strong-scaled cfg2 plan,
PlaNet at pop 1000 -- with the 16-wave workgroup variant and with the 4-wave kernels (hipets_set_wide_workgroups)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "mbrl-lib_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import hipets  # noqa: E402
from conftest import to_spec  # noqa: E402
from oracle import pets_oracle as po  # noqa: E402
from oracle import planet_oracle as pl  # noqa: E402

dev = torch.device("cuda:0")
eng = hipets.get_engine(dev)


def timed(fn, n=20):
    t_w = time.perf_counter()
    while time.perf_counter() - t_w < 0.3:
        fn()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


out = {}
for name, obs, act, pop, H, P, kw in [("cfg1_cartpole", 4, 1, 100, 15, 5, dict(reward="cartpole", termination="cartpole")),
                                      ("cfg2_shard_of_8", 17, 6, 63, 30, 20, {}), ("cfg2_shard_of_4", 17, 6, 125, 30, 20, {}),
                                      ("cfg2_shard_of_2", 17, 6, 250, 30, 20, {})]:
    om = po.make_synthetic_model(obs, act, ensemble_size=5, hid=200, seed=0, nontrivial_stats=False, **kw)
    eng.set_model(to_spec(om, obs, act))
    acts = (torch.rand(pop, H, act) * 2 - 1).to(dev)
    s0 = np.zeros(obs, np.float32)
    out[name] = {}
    for mode in ("device", "fast"):
        row = {}
        for wide in (True, False):
            eng.set_wide_workgroups(wide)
            row["wide16" if wide else "narrow4"] = timed(lambda: eng.rollout(acts, s0, P, mode=mode, seed=1))
        eng.set_wide_workgroups(True)
        out[name][mode + "_rollout_ms"] = row
pm = pl.make_synthetic_planet(30, 6, 200, 200, seed=0)
spec = hipets.PlaNetSpec(**{k: getattr(pm, k) for k in pl.PLANET_TENSORS}, min_std=pm.min_std)
eng.planet_set_model(spec)
acts = (torch.rand(1000, 12, 6) * 2 - 1).to(dev)
l0, b0 = torch.zeros(30, device=dev), torch.zeros(200, device=dev)
row = {}
for wide in (True, False):
    eng.set_wide_workgroups(wide)
    row["wide16" if wide else "narrow4"] = timed(lambda: eng.planet_rollout(acts, l0, b0, 1, seed=1))
eng.set_wide_workgroups(True)
out["planet_pop1000_rollout_ms"] = row
print(json.dumps(out))
