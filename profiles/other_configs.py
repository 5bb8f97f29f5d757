"""Timings of the BASELINE.json configs that are parity-test cases rather than the bench line (cfg1 cartpole, cfg4 iCEM
Humanoid shapes, cfg5 MPPI cheetah-run) and of the PlaNet latent planner: same kernels as bench.py, synthetic models, both
in-kernel randomness modes.  Run on a GPU box from the repo root:

    python profiles/other_configs.py > profiles/r2_other_configs.json          # everything, one JSON
    rocprofv3 --kernel-trace --stats ... -- python profiles/other_configs.py --only cfg4_icem_plan --mode device --reps 5

(uses oracle.make_synthetic_* only to BUILD random weights; nothing under oracle/ is timed)."""
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
from oracle import planet_oracle as pl  # noqa: E402

PEAK = 157.3e12
ap = argparse.ArgumentParser()
ap.add_argument("--only", default=None, help="run just this workload (for profiler runs)")
ap.add_argument("--mode", default=None, choices=["fast", "device"], help="restrict to one randomness mode")
ap.add_argument("--reps", type=int, default=10)
args = ap.parse_args()
dev = torch.device("cuda:0")
eng = hipets.get_engine(dev)
out = {}
MODES = [args.mode] if args.mode else ["device", "fast"]


def want(name):
    return args.only is None or args.only == name


def timed(fn, warm=3, n=None):
    # warm up for at least 0.3 s of wall time: after seconds of host-only work (building the synthetic models) the first
    # ~50 ms of GPU work run far below the steady-state rate (observed: 7x on the first 20 PlaNet rollouts)
    n = n or args.reps
    t_w = time.perf_counter()
    i = 0
    while i < warm or time.perf_counter() - t_w < 0.3:
        fn()
        torch.cuda.synchronize()
        i += 1
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def flops(om):
    return 2 * sum(int(w.shape[1]) * int(w.shape[2]) for w in om.weights)


# ---- one rollout (evaluate_action_sequences) per config ---------------------------------------------------------
# (termination as the reference's environments have it: cartpole's for cfg1, humanoid's for the cfg4 pair -- with it the cfg4 models
# match their shape-specialised kernel instances; rounds 2 and 3 timed them without, i.e. on the generic kernel)
for name, obs, act, E, elite, pop, H, P, rew, term in [
        ("cfg1_cartpole", 4, 1, 5, None, 100, 15, 5, "cartpole", "cartpole"), ("cfg2_halfcheetah", 17, 6, 5, None, 500, 30, 20, "halfcheetah", "no_termination"),
        ("cfg4_humanoid_truncated_obs", 45, 17, 7, [0, 1, 2, 3, 4], 1036, 40, 20, "halfcheetah", "humanoid"),
        ("cfg4_humanoid_v4_obs376", 376, 17, 7, [0, 1, 2, 3, 4], 1036, 40, 20, "halfcheetah", "humanoid"),
        ("cfg5_cheetah_run", 17, 6, 5, None, 2000, 50, 20, "halfcheetah", "no_termination")]:
    if not want(name):
        continue
    om = po.make_synthetic_model(obs, act, ensemble_size=E, hid=200, seed=0, nontrivial_stats=False, elite=elite, reward=rew, termination=term)
    eng.set_model(to_spec(om, obs, act))
    acts = (torch.rand(pop, H, act) * 2 - 1).to(dev)
    s0 = np.zeros(obs, np.float32)
    if term == "humanoid":
        s0[0] = 1.4  # a standing humanoid (termination_fns.py: z in (1.0, 2.0))
    nwg, r = eng.fast_geometry(pop, P, H, 0)
    out[name] = {"row_tiles_per_workgroup_fast": r, "workgroups_fast": nwg}
    for mode in MODES:
        dt = timed(lambda: eng.rollout(acts, s0, P, mode=mode, seed=1))
        cs = pop * P * H / dt
        out[name][mode] = {"rollout_ms": 1e3 * dt, "candidate_steps_per_s": cs, "frac_of_fp32_peak": cs * flops(om) / PEAK}

# ---- whole plans ----------------------------------------------------------------------------------------------------
if want("cfg5_mppi_plan"):
    obs, act, H, P = 17, 6, 50, 20  # cfg5: MPPI, pop 2000, 5 refinements
    om = po.make_synthetic_model(obs, act, ensemble_size=5, hid=200, seed=0, nontrivial_stats=False)
    out["cfg5_mppi_plan"] = {}
    for mode in MODES:
        fn = hipets.make_eval_fn(to_spec(om, obs, act), P, engine=eng, seed=0, mode=mode)
        lb, ub = [[-1.0] * act] * H, [[1.0] * act] * H
        mppi = hipets.MPPIOptimizer(5, 2000, 0.9, 1.0, 0.9, lb, ub, dev, seed=0)
        s0 = np.zeros(obs, np.float32)
        dt = timed(lambda: mppi.optimize(_BoundObjective(fn, s0)), warm=2, n=max(2, args.reps // 2))
        cs = 5 * 2000 * P * H / dt
        out["cfg5_mppi_plan"][mode] = {"ms_per_plan": 1e3 * dt, "candidate_steps_per_s": cs, "frac_of_fp32_peak_end_to_end": cs * flops(om) / PEAK}

if want("cfg4_icem_plan"):
    obs, act, H, P = 45, 17, 40, 20  # cfg4: iCEM, 7 members / 5 elites, pop 1000 decaying, keep 0.3
    om = po.make_synthetic_model(obs, act, ensemble_size=7, hid=200, seed=0, nontrivial_stats=False, elite=[0, 1, 2, 3, 4], termination="humanoid")
    out["cfg4_icem_plan"] = {}
    for mode in MODES:
        fn = hipets.make_eval_fn(to_spec(om, obs, act), P, engine=eng, seed=0, mode=mode)
        lb, ub = [[-1.0] * act] * H, [[1.0] * act] * H
        icem = hipets.ICEMOptimizer(5, 0.1, 1000, 1.3, 2.0, lb, ub, 0.3, 0.1, dev, return_mean_elites=True, population_size_module=7, seed=0)
        s0 = np.zeros(obs, np.float32)
        s0[0] = 1.4
        x0 = torch.zeros(H, act, device=dev)
        dt = timed(lambda: icem.optimize(_BoundObjective(fn, s0), x0=x0), warm=2, n=max(2, args.reps // 2))
        ncs = sum(n * P * H for n in (1036, 805, 630, 497, 358))
        out["cfg4_icem_plan"][mode] = {"ms_per_plan": 1e3 * dt, "candidate_steps_per_s": ncs / dt, "frac_of_fp32_peak_end_to_end": ncs / dt * flops(om) / PEAK}

if want("cfg1_cem_plan"):
    obs, act, H, P = 4, 1, 15, 5  # cfg1: cartpole, CEM pop 100
    om = po.make_synthetic_model(obs, act, ensemble_size=5, hid=200, seed=0, nontrivial_stats=False, reward="cartpole", termination="cartpole")
    out["cfg1_cem_plan"] = {}
    for mode in MODES:
        fn = hipets.make_eval_fn(to_spec(om, obs, act), P, engine=eng, seed=0, mode=mode)
        lb, ub = [[-1.0] * act] * H, [[1.0] * act] * H
        cem = hipets.CEMOptimizer(5, 0.1, 100, lb, ub, 0.1, dev, return_mean_elites=True, seed=0)
        s0 = np.zeros(obs, np.float32)
        x0 = torch.zeros(H, act, device=dev)
        dt = timed(lambda: cem.optimize(_BoundObjective(fn, s0), x0=x0), warm=2)
        out["cfg1_cem_plan"][mode] = {"ms_per_plan": 1e3 * dt, "candidate_steps_per_s": 5 * 100 * P * H / dt}

# ---- PlaNet latent planner (conf/dynamics_model/planet.yaml sizes, overrides/planet_cheetah_run.yaml planner) --------
if want("planet"):
    pm = pl.make_synthetic_planet(30, 6, 200, 200, seed=0)
    spec = hipets.PlaNetSpec(**{k: getattr(pm, k) for k in pl.PLANET_TENSORS}, min_std=pm.min_std)
    eng.planet_set_model(spec)
    for pop in (1000, 4000):
        acts = (torch.rand(pop, 12, 6) * 2 - 1).to(dev)
        l0, b0 = torch.zeros(30, device=dev), torch.zeros(200, device=dev)
        dt = timed(lambda: eng.planet_rollout(acts, l0, b0, 1, seed=1), n=20)
        cs = pop * 12 / dt
        out[f"planet_rollout_pop{pop}"] = {"rollout_ms": 1e3 * dt, "candidate_steps_per_s": cs, "frac_of_fp32_peak": cs * spec.flops_per_candidate_step() / PEAK}
    fn = hipets.make_eval_fn(spec, 1, engine=eng, seed=0)
    fn.set_state(torch.zeros(1, 30), torch.zeros(1, 200))
    cfg = dict(_target_="hipets.CEMOptimizer", num_iterations=10, elite_ratio=0.1, population_size=1000, alpha=0.0, device=dev, lower_bound="???",
               upper_bound="???", return_mean_elites=True, clipped_normal=True, seed=1)
    agent = hipets.TrajectoryOptimizerAgent(cfg, [-1.0] * 6, [1.0] * 6, planning_horizon=12, keep_last_solution=False)
    agent.set_trajectory_eval_fn(fn)
    obs_img = np.zeros((3, 64, 64), np.float32)
    dt = timed(lambda: agent.plan(obs_img), warm=3, n=10)
    out["planet_cem_plan"] = {"ms_per_plan": 1e3 * dt, "workload": "clipped-normal CEM, pop 1000, H 12, 10 iterations"}

print(json.dumps(out, indent=1))
