"""cfg4' (Humanoid-v4, obs 376) iCEM plan, iteration by iteration: the five decaying population sizes as single rollouts, both modes,
the row-tile count the rule picks and every forced one (where 0.47 of the first iteration becomes 0.40 at plan level).  Prints one
JSON line; run on a GPU box from the repo root."""
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
obs = int(os.environ.get("PROBE_OBS", "376"))
spec = bench.synthetic_spec(dev, obs=obs, act=17, ensemble=7, elite=[0, 1, 2, 3, 4], termination="humanoid")
eng.set_model(spec)
P, H = 20, 40
s0 = np.zeros(obs, np.float32)
s0[0] = 1.4
out = {"obs": obs, "lib": os.environ.get("HIPETS_LIB", "default")}
for pop in [int(x) for x in os.environ.get("PROBE_POPS", "1036,805,630,497,358").split(",")]:
    acts = (torch.rand(pop, H, 17) * 2 - 1).to(dev)
    rec = {}
    for mode in ("fast", "device"):
        try:
            rec[f"{mode}_class"] = list(eng.kernel_class(pop, P, H, mode=mode))
        except Exception as exc:  # noqa: BLE001
            rec[f"{mode}_class"] = str(exc)
        for R in [int(x) for x in os.environ.get("PROBE_RS", "0,1,2,3,4").split(",")]:
            f = lambda i=0: eng.rollout(acts, s0, P, mode=mode, seed=1, stream_id=i, rows_per_group=R)  # noqa: E731
            try:
                t0 = time.perf_counter()
                while time.perf_counter() - t0 < 0.15:
                    f()
                    torch.cuda.synchronize()
            except hipets.HipetsError as exc:
                rec[f"{mode}_R{R}"] = str(exc)[:80]
                continue
            eng.timing_enable(True)
            eng.timing_read(reset=True)
            n = 5
            for i in range(n):
                f(i)
            torch.cuda.synchronize()
            nl, kms = eng.timing_read(reset=True)
            eng.timing_enable(False)
            rec[f"{mode}_R{R}"] = {"ms": round(kms / n, 4), "launches": nl / n,
                                   "frac": round(pop * P * H * spec.flops_per_candidate_step() / (kms / n * 1e-3) / 157.3e12, 4)}
    out[str(pop)] = rec
print(json.dumps(out))
