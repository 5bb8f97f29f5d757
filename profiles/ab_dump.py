"""A/B aid: DEVICE-mode returns of a few multi-turn / straight cases (persistent and per-step launches) with the library HIPETS_LIB
selects, dumped to an .npz -- two runs with two library builds are compared bit for bit by profiles/ab_dump.py --compare a.npz b.npz."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "mbrl-lib_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402

if sys.argv[1] == "--compare":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    for k in a.files:
        same = np.array_equal(a[k], b[k], equal_nan=True)
        print(k, "same" if same else f"DIFFERENT: {int((a[k] != b[k]).sum())} of {a[k].size}, max |d| {np.nanmax(np.abs(a[k] - b[k])):.3e}, finite {np.isfinite(a[k]).all()} / {np.isfinite(b[k]).all()}")
    sys.exit(0)

import torch  # noqa: E402

import hipets  # noqa: E402
from conftest import to_spec  # noqa: E402
from test_gpu_rollout import _random_case  # noqa: E402

dev = "cuda:0"
eng = hipets.get_engine(dev)
out = {}
CASES = {"cfg2_650": (17, 6, 650, 20, 3, dict(hid=200)), "cfg2_500": (17, 6, 500, 20, 5, dict(ensemble_size=5, hid=200)),
         "cfg4_1036": (45, 17, 1036, 20, 4, dict(ensemble_size=7, hid=200, elite=[0, 1, 2, 3, 4], termination="humanoid")),
         "cfg2_1000": (17, 6, 1000, 20, 5, dict(hid=200))}
for name, (obs, act, pop, P, H, mkw) in CASES.items():
    om, actions, s0, _, _ = _random_case(obs, act, pop, P, H, **mkw)
    eng.set_model(to_spec(om, obs, act))
    for pers in (True, False):
        eng.set_persistent(pers)
        for rep in range(int(os.environ.get("AB_REPS", "2"))):
            r = eng.rollout(actions.to(dev), s0, P, mode="device", seed=77, stream_id=9)
            torch.cuda.synchronize()
            timed_out = eng.check_async_error()
            out[f"{name}_{'persistent' if pers else 'per_step'}_{rep}"] = r.cpu().numpy()
            print(name, "persistent" if pers else "per-step", rep, "timed out" if timed_out else "ok", float(r.sum()), flush=True)
    eng.set_persistent(True)
np.savez(sys.argv[1], **out)
