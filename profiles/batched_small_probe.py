"""Small problems leave most of the chip idle (cfg1: 35 one-tile workgroups on 256 CUs, a plan is five 0.25 ms latency chains): what
batched planning (hipets.BatchedCEMAgent: n_env environments per launch, same kernels) buys per environment-plan.  cfg1 cartpole and
the shard a rank holds in an 8-GPU strong-scaled cfg2 plan (63 candidates)."""
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
out = {}
for name, mkw, pop, P, H in [("cfg1_cartpole", dict(obs=4, act=1, ensemble=5, reward="cartpole", termination="cartpole"), 100, 5, 15),
                             ("cfg2_shard_63_candidates", dict(), 65, 20, 30)]:
    spec = bench.synthetic_spec(dev, **mkw)
    act, obs = spec.act_dim, spec.obs_dim
    fn = hipets.make_eval_fn(spec, P, engine=eng, seed=0, mode="fast")
    res = {}
    for n_env in (1, 2, 4, 8, 16, 32):
        agent = hipets.BatchedCEMAgent(fn, n_env, [-1.0] * act, [1.0] * act, H, 5, 0.1, pop, 0.1, seed=0)
        s0 = (np.random.default_rng(0).standard_normal((n_env, obs)) * 0.1).astype(np.float32)
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.3:
            agent.plan(s0)
        torch.cuda.synchronize()
        n = 30
        t0 = time.perf_counter()
        for _ in range(n):
            agent.plan(s0)
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / n
        nwg, r = eng.fast_geometry(pop * n_env, P, H)
        res[f"n_env={n_env}"] = {"ms_per_call": 1e3 * el, "ms_per_environment_plan": 1e3 * el / n_env, "workgroups": nwg, "row_tiles": r,
                                 "candidate_steps_per_s": 5 * pop * n_env * P * H / el}
    out[name] = res
print(json.dumps(out, indent=1))
