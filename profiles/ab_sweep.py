"""Debugging aid: persistent vs per-step DEVICE-mode returns over a sweep of population sizes (cfg2's model): how many candidates differ."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "mbrl-lib_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import hipets  # noqa: E402
from conftest import to_spec  # noqa: E402
from test_gpu_rollout import _random_case  # noqa: E402

dev = "cuda:0"
eng = hipets.get_engine(dev)
H = int(os.environ.get("AB_H", "3"))
for pop in [int(x) for x in os.environ.get("AB_POPS", "650,648,652,660,700,1000,1004").split(",")]:
    for R in [int(x) for x in os.environ.get("AB_R", "0").split(",")]:
        om, actions, s0, _, _ = _random_case(17, 6, pop, 20, H, hid=200)
        eng.set_model(to_spec(om, 17, 6))
        eng.set_persistent(False)
        ref = eng.rollout(actions.to(dev), s0, 20, mode="device", seed=77, stream_id=9, rows_per_group=R).cpu()
        eng.set_persistent(True)
        res = []
        for rep in range(6):
            r = eng.rollout(actions.to(dev), s0, 20, mode="device", seed=77, stream_id=9, rows_per_group=R).cpu()
            torch.cuda.synchronize()
            to = eng.check_async_error()
            res.append(("T" if to else "") + str(int((r != ref).sum())))
            if to:
                eng.set_persistent(True)
        rows = pop * 20 // 5
        print(f"pop {pop} R {R}: rows/member {rows} (last workgroup: {rows % (16 * max(R, 1)) if R else '?'} rows), candidates differing per rep: {res}", flush=True)
