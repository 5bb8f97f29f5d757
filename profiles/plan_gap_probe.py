"""Where does a small DEVICE-mode plan spend its wall time?  cfg1 (35 one-tile workgroups) and the bf16x3 cfg2 plan: wall time per
plan with a long warm-up, with and without the launch-duration events, against the rollout kernel's own time."""
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
from hipets.planning import _BoundObjective  # noqa: E402

dev = torch.device("cuda:0")
eng = hipets.get_engine(dev)
out = {"env": {k: v for k, v in os.environ.items() if k.startswith("HIPETS")}}
cases = {"cfg1_device": (dict(obs=4, act=1, ensemble=5, reward="cartpole", termination="cartpole"), "f32", 100, 5, 15, "device"),
         "cfg1_fast": (dict(obs=4, act=1, ensemble=5, reward="cartpole", termination="cartpole"), "f32", 100, 5, 15, "fast"),
         "cfg2_bf16x3_device": (dict(), "bf16x3", 500, 20, 30, "device"),
         "cfg2_f32_device": (dict(), "f32", 500, 20, 30, "device")}
for name, (mkw, prec, pop, P, H, mode) in cases.items():
    spec = bench.synthetic_spec(dev, precision=prec, **mkw)
    act, obs = spec.act_dim, spec.obs_dim
    fn = hipets.make_eval_fn(spec, P, engine=eng, seed=0, mode=mode)
    opt = hipets.CEMOptimizer(5, 0.1, pop, [[-1.0] * act] * H, [[1.0] * act] * H, 0.1, dev, return_mean_elites=True, seed=0)
    obj = _BoundObjective(fn, np.zeros(obs, np.float32))
    x0 = torch.zeros(H, act, device=dev)
    plan = lambda: opt.optimize(obj, x0=x0)  # noqa: E731
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.5:
        plan()
        torch.cuda.synchronize()
    res = {}
    for timing in (0, 1):
        eng.timing_enable(timing)
        eng.timing_read(reset=True)
        torch.cuda.synchronize()
        n = 40
        t0 = time.perf_counter()
        for _ in range(n):
            plan()
        t_enq = time.perf_counter() - t0
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        nl, kms = eng.timing_read(reset=True)
        eng.timing_enable(False)
        res["timing_on" if timing else "timing_off"] = {"ms_per_plan": 1e3 * el / n, "host_enqueue_ms_per_plan": 1e3 * t_enq / n,
                                                        "kernel_ms_per_plan": kms / n if nl else None, "launches": nl}
    # synchronous plans: one at a time
    per = []
    for _ in range(20):
        t0 = time.perf_counter()
        plan()
        torch.cuda.synchronize()
        per.append(1e3 * (time.perf_counter() - t0))
    res["synchronous_ms_per_plan_median"] = float(np.median(per))
    out[name] = res
print(json.dumps(out, indent=1))
