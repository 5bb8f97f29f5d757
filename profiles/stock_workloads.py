"""Timings of the workloads the reference SHIPS (conf/overrides/pets_halfcheetah.yaml, pets_cartpole.yaml: the planner a user of
`python -m mbrl.examples.main algorithm=pets overrides=pets_halfcheetah` runs) next to the synthetic cfg2 of BASELINE.json:
one rollout (evaluate_action_sequences) per randomness mode with the kernel's own launch duration (hipEvents on the dispatch
packets, hipets_timing_*), optionally for every forced row-tile count R and for the generic kernel, and the whole CEM plan.
Run on a GPU box from the repo root:

    python profiles/stock_workloads.py [--sweep-r] [--generic] > profiles/r4_stock_workloads.json

(uses oracle.make_synthetic_model only to BUILD random weights; nothing under oracle/ is timed)."""
import argparse
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
from hipets.planning import _BoundObjective  # noqa: E402
from oracle import pets_oracle as po  # noqa: E402

PEAK = 157.3e12
WORKLOADS = {
    # name: obs, act, model kwargs, pop, P, H, elite ratio, alpha
    "cfg2_synthetic": (17, 6, dict(ensemble_size=5), 500, 20, 30, 0.1, 0.1),
    "stock_halfcheetah": (18, 6, dict(ensemble_size=7, elite=[0, 2, 3, 5, 6], obs_process="halfcheetah", no_delta_list=[0]), 400, 20, 30, 0.16, 0.12),
    "stock_cartpole": (4, 1, dict(ensemble_size=7, elite=[1, 2, 4, 5, 6], reward="cartpole", termination="cartpole"), 350, 20, 15, 0.1, 0.1),
    # learned rewards, no termination function: conf/overrides/pets_pusher.yaml, pets_reacher.yaml, pets_mppi_halfcheetah.yaml (its model)
    "stock_pusher": (20, 7, dict(ensemble_size=7, elite=[0, 1, 3, 4, 6], learned_rewards=True, reward=None), 350, 20, 25, 0.1, 0.1),
    "stock_reacher": (17, 7, dict(ensemble_size=7, elite=[0, 1, 3, 4, 6], no_delta_list=[0], learned_rewards=True, reward=None), 350, 20, 15, 0.1, 0.1),
    "stock_mppi_halfcheetah_model": (18, 6, dict(ensemble_size=7, elite=[0, 2, 3, 5, 6], obs_process="halfcheetah", no_delta_list=[0],
                                                 learned_rewards=True, reward=None), 350, 20, 30, 0.1, 0.1),
    # conf/overrides/pets_inv_pendulum.yaml: learned reward + the inverted_pendulum termination function (every state dim)
    "stock_inv_pendulum": (4, 1, dict(ensemble_size=7, elite=[0, 2, 3, 5, 6], learned_rewards=True, reward=None, termination="inverted_pendulum"),
                           480, 20, 45, 0.078, 0.134484),
    # conf/overrides/pets_hopper.yaml: learned reward + the hopper termination function (eleven state dims): fused in FAST mode only
    "stock_hopper": (11, 3, dict(ensemble_size=7, elite=[0, 1, 3, 4, 6], learned_rewards=True, reward=None, termination="hopper"), 350, 20, 30, 0.1, 0.1),
}


def flops(om):
    return 2 * sum(int(w.shape[1]) * int(w.shape[2]) for w in om.weights)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    ap.add_argument("--mode", default=None, choices=["fast", "device"])
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--sweep-r", action="store_true", help="also time every forced row-tile count R = 1..4")
    ap.add_argument("--generic", action="store_true", help="also time the generic kernel instance (hipets_rollout_opts.generic_kernel)")
    ap.add_argument("--no-plans", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    eng = hipets.get_engine(dev)
    modes = [args.mode] if args.mode else ["device", "fast"]
    out = {}

    def kernel_ms(fn, reps):
        t_w = time.perf_counter()
        i = 0
        while i < 3 or time.perf_counter() - t_w < 0.3:
            fn()
            torch.cuda.synchronize()
            i += 1
        eng.timing_enable(True)
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / reps
        n, ms = eng.timing_read()
        eng.timing_enable(False)
        return wall * 1e3, ms / reps, n // reps  # wall ms per call, rollout-kernel ms per call, kernel launches per call

    for name, (obs, act, mkw, pop, P, H, ratio, alpha) in WORKLOADS.items():
        if args.only and args.only != name:
            continue
        om = po.make_synthetic_model(obs, act, hid=200, seed=0, **mkw)
        spec = to_spec(om, obs, act)
        eng.set_model(spec)
        acts = (torch.rand(pop, H, act, generator=torch.Generator().manual_seed(0)) * 2 - 1).to(dev)
        s0 = (np.random.default_rng(0).standard_normal(obs) * 0.1).astype(np.float32)
        if om.termination == "hopper":
            s0[0] = 1.25  # a standing hopper (termination_fns.py:12-26)
        fl = flops(om)
        nwg, r = eng.fast_geometry(pop, P, H, 0)
        res = {"flop_per_candidate_step": fl, "candidate_steps_per_rollout": pop * P * H, "fast_geometry": {"workgroups": nwg, "row_tiles": r},
               "kernel_class": {m: list(eng.kernel_class(pop, P, H, m)) for m in ("device", "fast")}}
        for mode in modes:
            variants = [("default", dict())]
            if args.generic:
                variants.append(("generic_kernel", dict(generic_kernel=True)))
            if args.sweep_r:
                variants += [(f"R{R}", dict(rows_per_group=R)) for R in (1, 2, 3, 4)]
            res[mode] = {}
            for vname, kw in variants:
                try:
                    wall, kms, nl = kernel_ms(lambda: eng.rollout(acts, s0, P, mode=mode, seed=1, **kw), args.reps)
                except hipets.HipetsError as exc:
                    res[mode][vname] = {"error": str(exc)[:120]}
                    continue
                res[mode][vname] = {"rollout_wall_ms": wall, "rollout_kernel_ms": kms, "kernel_launches": nl,
                                    "frac_of_fp32_peak": pop * P * H * fl / (kms * 1e-3) / PEAK}
        if not args.no_plans:
            lb, ub = [[-1.0] * act] * H, [[1.0] * act] * H
            res["cem_plan"] = {}
            for mode in modes:
                fn = hipets.make_eval_fn(spec, P, engine=eng, seed=0, mode=mode)
                cem = hipets.CEMOptimizer(5, ratio, pop, lb, ub, alpha, dev, return_mean_elites=True, seed=0)
                x0 = torch.zeros(H, act, device=dev)
                wall, kms, nl = kernel_ms(lambda: cem.optimize(_BoundObjective(fn, s0), x0=x0), max(4, args.reps // 2))
                cs = 5 * pop * P * H
                res["cem_plan"][mode] = {"ms_per_plan": wall, "candidate_steps_per_s": cs / (wall * 1e-3),
                                         "rollout_kernel_ms_per_launch": kms / max(nl, 1), "kernel_launches": nl,
                                         "kernel_frac_of_fp32_peak": pop * P * H * fl / (kms / max(nl, 1) * 1e-3) / PEAK,
                                         "plan_frac_of_fp32_peak_end_to_end": cs * fl / (wall * 1e-3) / PEAK}
        out[name] = res
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
