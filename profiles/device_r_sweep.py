"""DEVICE-mode rollouts of the large batches with every forced row-tile count: where the batch exceeds what the chip holds, R <= 2 instances
(two workgroups per CU) run one launch per step, R >= 3 instances (one per CU) the turn-based persistent form."""
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
    "cfg5 (pop 2000 x 20, H 50)": (dict(), 2000, 20, 50),
    "cfg2 x 2 (pop 1000 x 20, H 30)": (dict(), 1000, 20, 30),
    "cfg2 x 4 (pop 2000 x 20, H 30)": (dict(), 2000, 20, 30),
    "pets_halfcheetah x 2 (pop 800 x 20, H 30)": (dict(obs=18, act=6, ensemble=7, elite=[0, 2, 3, 5, 6], obs_process="halfcheetah", no_delta_list=[0]), 800, 20, 30),
    "cfg4 first iCEM iteration (obs 45, pop 1036 x 20, H 40)": (dict(obs=45, act=17, ensemble=7, elite=[0, 1, 2, 3, 4], termination="humanoid"), 1036, 20, 40),
}
out = {}
for name, (mkw, pop, P, H) in CASES.items():
    spec = bench.synthetic_spec(dev, **mkw)
    eng.set_model(spec)
    acts = (torch.rand(pop, H, spec.act_dim, generator=torch.Generator().manual_seed(0)) * 2 - 1).to(dev)
    s0 = (np.random.default_rng(0).standard_normal(spec.obs_dim) * 0.1).astype(np.float32)
    if "termination" in mkw:
        s0[0] = 1.4
    res = {"kernel_class": list(eng.kernel_class(pop, P, H, "device"))}
    ref = None
    for mode in ("device", "fast"):
        for R in (0, 1, 2, 3, 4):
            f = lambda i=0: eng.rollout(acts, s0, P, mode=mode, seed=1, stream_id=i, rows_per_group=R)  # noqa: E731
            try:
                t0 = time.perf_counter()
                while time.perf_counter() - t0 < 0.25:
                    r = f()
                    torch.cuda.synchronize()
            except hipets.HipetsError as exc:
                res[f"{mode}_R{R}"] = {"error": str(exc)[:80]}
                continue
            if mode == "device":
                ref = r.clone() if ref is None else ref
                assert torch.equal(r, ref)
            eng.timing_enable(True)
            eng.timing_read(reset=True)
            n = 6
            for i in range(n):
                f(i)
            torch.cuda.synchronize()
            nl, kms = eng.timing_read(reset=True)
            eng.timing_enable(False)
            res[f"{mode}_R{R}"] = {"ms": round(kms / n, 4), "launches": nl / n,
                                   "frac": round(pop * P * H * spec.flops_per_candidate_step() / (kms / n * 1e-3) / 157.3e12, 4)}
    out[name] = res
print(json.dumps(out, indent=1))
