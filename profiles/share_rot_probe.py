"""A/B of a share rotation of co-resident workgroups that an EXPERIMENT build had (HIPETS_NO_SHARE_ROT=1
switched it off): rollouts whose workgroups meet two to a CU, and controls whose do not.  With the library as shipped both runs measure
the same thing.  Results: profiles/r4_co_residency_experiments.json."""
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
CASES = {
    # name: (model kwargs, pop, P, H, [(mode, rows_per_group)])
    "cfg2": (dict(), 500, 20, 30, [("fast", 0), ("device", 0), ("fast", 1), ("fast", 2)]),
    "pets_halfcheetah": (dict(obs=18, act=6, ensemble=7, elite=[0, 2, 3, 5, 6], obs_process="halfcheetah", no_delta_list=[0]), 400, 20, 30,
                         [("fast", 0), ("device", 0)]),
    "cfg5": (dict(), 2000, 20, 50, [("fast", 0), ("device", 0)]),
    "cfg2_x8_batched_size": (dict(), 4000, 20, 30, [("fast", 0)]),
    "cfg4p": (dict(obs=376, act=17, ensemble=7, elite=[0, 1, 2, 3, 4], termination="humanoid"), 1036, 20, 40, [("fast", 1), ("fast", 2), ("device", 1), ("device", 2)]),
}
out = {"no_share_rot": os.environ.get("HIPETS_NO_SHARE_ROT"), "turns_any": os.environ.get("HIPETS_TURNS_ANY")}
for name, (mkw, pop, P, H, runs) in CASES.items():
    spec = bench.synthetic_spec(dev, **mkw)
    eng.set_model(spec)
    act = spec.act_dim
    acts = (torch.rand(pop, H, act, generator=torch.Generator().manual_seed(0)) * 2 - 1).to(dev)
    s0 = (np.random.default_rng(0).standard_normal(spec.obs_dim) * 0.1).astype(np.float32)
    if "termination" in mkw:
        s0[0] = 1.4
    for mode, R in runs:
        f = lambda i=0: eng.rollout(acts, s0, P, mode=mode, seed=1, stream_id=i, rows_per_group=R)  # noqa: E731
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.3:
            r = f()
            torch.cuda.synchronize()
        eng.timing_enable(True)
        eng.timing_read(reset=True)
        n = 8
        for i in range(n):
            f(i)
        torch.cuda.synchronize()
        nl, kms = eng.timing_read(reset=True)
        eng.timing_enable(False)
        out[f"{name}_{mode}_R{R}"] = {"ms": round(kms / n, 4), "launches": nl / n, "frac": round(pop * P * H * spec.flops_per_candidate_step() / (kms / n * 1e-3) / 157.3e12, 4),
                                      "checksum": float(r.double().sum())}
print(json.dumps(out))
