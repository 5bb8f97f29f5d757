#!/usr/bin/env python
"""bench.py -- candidate-steps/sec of the PETS planning hot path on MI355X (BASELINE.json metric).

A "step" is ONE PLAN: the whole CEMOptimizer.optimize loop (5 iterations x pop 500 x 20 particles x
horizon 30 rollouts through the 5-member GaussianMLP ensemble + elite refit) = what
TrajectoryOptimizerAgent.act does per environment step, on BASELINE.json configs[1]
("PETS HalfCheetah (obs=17, act=6), ensemble=5, CEM pop=500, horizon=30, particles=20").
Inputs (weights, s0, bounds) are resident in HBM before the timed region; synthetic random-init
weights (reference initialiser), fp32 arithmetic end to end (fp64 input normaliser like the reference).

    python bench.py [--gpus N --steps K --warmup W]        # N>1: launched by torch.distributed.run

N>1: candidates sharded over N ranks (one process per GPU), replicated sampling, one RCCL all-gather of the candidate
returns per CEM iteration.  Default `--scaling weak` keeps cfg2's pop 500 PER RANK (pop 500 N in total); the literal
configs[2] (the same pop-500 plan sharded N ways, latency-bound by construction) is timed too and
reported as the extra `cfg3_strong` block; `--scaling strong` makes it the headline value instead.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "mbrl-lib_amd"))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

OBS, ACT, ENSEMBLE, HID, LAYERS = 17, 6, 5, 200, 4
POP, HORIZON, PARTICLES, ITERS, ELITE_RATIO, ALPHA = 500, 30, 20, 5, 0.1, 0.1
PEAK_FP32_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32 vector == fp32 MFMA dense peak


def synthetic_spec(device):
    """Random-init GaussianMLP ensemble with the reference initialiser (models/util.py:15-28: truncated
    normal std 1/(2 sqrt(in)), zero bias; logvar bounds -10 / 0.5), built on the product side (no oracle)."""
    import hipets

    g = torch.Generator().manual_seed(0)
    dims = [OBS + ACT] + [HID] * LAYERS + [2 * OBS]
    ws, bs = [], []
    for i in range(len(dims) - 1):
        std = 1.0 / (2.0 * np.sqrt(dims[i]))
        w = torch.empty(ENSEMBLE, dims[i], dims[i + 1])
        torch.nn.init.trunc_normal_(w, mean=0.0, std=std, a=-2 * std, b=2 * std, generator=g)
        ws.append(w.to(device))
        bs.append(torch.zeros(ENSEMBLE, 1, dims[i + 1], device=device))
    return hipets.ModelSpec(
        weights=ws, biases=bs, obs_dim=OBS, act_dim=ACT, min_logvar=-10 * torch.ones(1, OBS), max_logvar=0.5 * torch.ones(1, OBS),
        activation="silu", propagation="random_model", norm_mean=torch.zeros(1, OBS + ACT, dtype=torch.float64),
        norm_std=torch.ones(1, OBS + ACT, dtype=torch.float64), target_is_delta=True, learned_rewards=False,
        reward="halfcheetah", termination="no_termination")


def _usable_cores():
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:  # cgroup v2 CPU quota, if any
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def cpu_baseline(budget_s=20.0):
    """The reference's algorithm on this box's host cores: the oracle (a torch-CPU restatement that is bitwise
    equal to mbrl-lib's ModelEnv + CEMOptimizer, see oracle/) timed on a BOUNDED sample of the cfg2 workload.
    Checker code, used here ONLY as the reported baseline, never by the product path.

    The per-step cost of evaluate_action_sequences does not depend on the step index, so the sample is cfg2's
    full batch (pop 500 x 20 particles) rolled for a shortened horizon sized to fit the time budget; the thread
    count is calibrated first (torch's default of one thread per logical core can be catastrophically slow)."""
    from oracle import pets_oracle as po

    om = po.make_synthetic_model(OBS, ACT, ensemble_size=ENSEMBLE, hid=HID, num_layers=LAYERS, seed=0, nontrivial_stats=False)
    s0 = (np.random.default_rng(0).standard_normal(OBS) * 0.1).astype(np.float32)
    gen = torch.Generator().manual_seed(0)
    torch.manual_seed(0)
    g = torch.Generator().manual_seed(1)

    def run(h):
        acts = torch.rand(POP, h, ACT, generator=g) * 2 - 1
        t0 = time.perf_counter()
        po.rollout(om, acts, s0, PARTICLES, global_rng=True, generator=gen)
        return time.perf_counter() - t0

    t_start = time.perf_counter()
    usable = _usable_cores()
    best_threads, best_t = 1, None
    for nt in sorted({1, min(8, usable), min(16, usable), min(32, usable), min(64, usable)}):
        torch.set_num_threads(nt)
        run(1)  # warm-up for this thread count
        t = min(run(1), run(1))
        if best_t is None or t < best_t:
            best_threads, best_t = nt, t
        if time.perf_counter() - t_start > 0.4 * budget_s:
            break
    torch.set_num_threads(best_threads)
    remaining = max(1.0, budget_s - (time.perf_counter() - t_start))
    h = int(max(1, min(HORIZON, remaining / 2 / best_t)))
    reps = int(max(2, min(6, remaining / (h * best_t))))
    times = [run(h) for _ in range(reps)]
    cs = POP * PARTICLES * h
    v = cs / min(times)
    return {"value": v, "unit": "candidate-steps/s", "cores": best_threads, "kind": "port",
            "sample": f"{reps} x evaluate_action_sequences on cfg2's batch (pop {POP} x {PARTICLES} particles) for {h} of {HORIZON} "
                      f"horizon steps ({cs} candidate-steps), min time; thread count calibrated over <= {usable} usable "
                      f"cores; torch {torch.__version__} CPU",
            "plans_per_s_extrapolated": v / (ITERS * POP * PARTICLES * HORIZON)}


def torch_rocm_port(device, plans=2):
    """Informational second comparator: the SAME ATen op sequence as the reference
    (the oracle is bitwise equal to it on CPU) executed on this MI355X through PyTorch-ROCm, i.e. what
    `device="cuda:0"` buys the unmodified reference: ~55 launches per rollout step, launch-bound."""
    from oracle import pets_oracle as po

    om = po.make_synthetic_model(OBS, ACT, ensemble_size=ENSEMBLE, hid=HID, num_layers=LAYERS, seed=0, nontrivial_stats=False)
    for k in ("weights", "biases"):
        setattr(om, k, [t.to(device) for t in getattr(om, k)])
    for k in ("min_logvar", "max_logvar", "norm_mean", "norm_std"):
        setattr(om, k, getattr(om, k).to(device))
    s0 = (np.random.default_rng(0).standard_normal(OBS) * 0.1).astype(np.float32)
    lb, ub = -torch.ones(HORIZON, ACT, device=device), torch.ones(HORIZON, ACT, device=device)
    gen = torch.Generator(device=device).manual_seed(0)
    obj = lambda pop_: po.rollout(om, pop_, s0, PARTICLES, global_rng=True, generator=gen)  # noqa: E731

    def plan():
        return po.cem_optimize(obj, torch.zeros(HORIZON, ACT, device=device), lb, ub, ITERS, ELITE_RATIO, POP, ALPHA, return_mean_elites=True)

    plan()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(plans):
        plan()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / plans
    return {"value": ITERS * POP * PARTICLES * HORIZON / dt, "unit": "candidate-steps/s", "ms_per_plan": 1e3 * dt,
            "what": "reference op sequence (oracle port) on this GPU via PyTorch-ROCm eager ops, not the product path"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--scaling", choices=["strong", "weak"], default="weak")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-batched", action="store_true", help="skip the extra batched-planning measurement")
    ap.add_argument("--cpu-budget", type=float, default=20.0)
    args = ap.parse_args()

    import hipets
    from hipets import dist as hdist
    from hipets.planning import _BoundObjective

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs `python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py ...`")
    # one rank per GPU; HIPETS_DIST_BACKEND=gloo lets several ranks share one GPU (single-GPU smoke test of the N>1 path)
    backend = os.environ.get("HIPETS_DIST_BACKEND", "nccl")
    dev_index = local_rank % torch.cuda.device_count() if backend != "nccl" else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    engine = hipets.get_engine(device)
    spec = synthetic_spec(device)
    eval_fn = hipets.make_eval_fn(spec, PARTICLES, engine=engine, seed=0)
    lb, ub = [[-1.0] * ACT] * HORIZON, [[1.0] * ACT] * HORIZON
    s0 = (np.random.default_rng(0).standard_normal(OBS) * 0.1).astype(np.float32)
    x0 = torch.zeros(HORIZON, ACT, device=device)
    if world > 1:
        objective = _BoundObjective(hdist.ShardedEvalFn(eval_fn), s0)  # generic path + one all-gather per iteration
    else:
        objective = _BoundObjective(eval_fn, s0)  # fused hipets_plan_cem

    def barrier():
        if world > 1:
            import torch.distributed as dist

            dist.barrier()
        torch.cuda.synchronize()

    def run(pop_total, steps, warmup):
        """warmup untimed plans, then exactly `steps` plans between barrier+synchronize; max over ranks."""
        opt = hipets.CEMOptimizer(ITERS, ELITE_RATIO, pop_total, lb, ub, ALPHA, device, return_mean_elites=True, seed=0)
        for _ in range(warmup):
            opt.optimize(objective, x0=x0)
        engine.timing_enable(True)
        engine.timing_read(reset=True)
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            sol = opt.optimize(objective, x0=x0)
        barrier()
        el = time.perf_counter() - t0
        launches_, kernel_ms_ = engine.timing_read(reset=True)
        engine.timing_enable(False)
        if world > 1:
            import torch.distributed as dist

            t = torch.tensor([el], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        assert torch.isfinite(sol).all()
        return el, launches_, kernel_ms_

    # N = 1: BASELINE.json configs[1] (pop 500).  N > 1: per-GPU work fixed (pop 500 per rank, sharded evaluation, one
    # all-gather of returns per iteration) = "weak"; --scaling strong times configs[2] literally (the SAME pop-500 plan
    # sharded N ways), which is also always reported as the extra `cfg3_strong` block of the N > 1 line.
    pop = POP * world if (world > 1 and args.scaling == "weak") else POP
    elapsed, launches, kernel_ms = run(pop, args.steps, args.warmup)
    extra = None
    if world > 1 and args.scaling == "weak":
        e2, _, _ = run(POP, args.steps, max(1, args.warmup // 2))
        extra = {"workload": f"configs[2]: the pop={POP} plan sharded over {world} ranks (strong scaling)", "value": args.steps * ITERS * POP *
                 PARTICLES * HORIZON / e2, "unit": "candidate-steps/s", "ms_per_step": 1e3 * e2 / args.steps, "plans_per_s": args.steps / e2}

    # extra (N = 1 only, outside the timed region above): batched planning, 8 cfg2 environments per launch -- the same
    # kernels with all 256 CUs busy (a single cfg2 plan has 2.4 row tiles per CU).  Reported, never the headline value.
    batched = None
    if world == 1 and not args.no_batched:
        n_env = 8
        agent = hipets.BatchedCEMAgent(eval_fn, n_env, [-1.0] * ACT, [1.0] * ACT, HORIZON, ITERS, ELITE_RATIO, POP, ALPHA, seed=0)
        s0b = (np.random.default_rng(0).standard_normal((n_env, OBS)) * 0.1).astype(np.float32)
        for _ in range(2):
            agent.plan(s0b)
        engine.timing_enable(True)
        engine.timing_read(reset=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        nb = max(2, args.steps // 8)
        for _ in range(nb):
            agent.plan(s0b)
        torch.cuda.synchronize()
        eb = time.perf_counter() - t0
        lb_, kms_ = engine.timing_read(reset=True)
        engine.timing_enable(False)
        tf = spec.flops_per_candidate_step() * n_env * POP * PARTICLES * HORIZON / (kms_ / max(lb_, 1) * 1e-3) / 1e12
        batched = {"workload": f"{n_env} x configs[1] environments planned in one set of launches (hipets_plan_cem_batched)",
                   "value": nb * n_env * ITERS * POP * PARTICLES * HORIZON / eb, "unit": "candidate-steps/s",
                   "ms_per_env_plan": 1e3 * eb / nb / n_env, "rollout_kernel_tflops": tf, "rollout_kernel_frac_of_fp32_peak": tf / PEAK_FP32_TFLOPS}

    # extra (N = 1): the same plan with the reference's EXACT propagation semantics (one global balanced random
    # permutation of all 10 000 rows per step, per-step launches, state through HBM) instead of FAST mode's
    # block-balanced schedule -- shows what the fast mode's restructuring is worth and that it is not needed for speed-up.
    exact_sem = None
    if world == 1 and not args.no_batched:
        fn_x = hipets.make_eval_fn(spec, PARTICLES, engine=engine, seed=0, mode="exact_device")
        opt_x = hipets.CEMOptimizer(ITERS, ELITE_RATIO, POP, lb, ub, ALPHA, device, return_mean_elites=True, seed=0)
        obj_x = _BoundObjective(fn_x, s0)
        for _ in range(2):
            opt_x.optimize(obj_x, x0=x0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        nx = max(2, args.steps // 5)
        for _ in range(nx):
            opt_x.optimize(obj_x, x0=x0)
        torch.cuda.synchronize()
        ex = time.perf_counter() - t0
        exact_sem = {"workload": "configs[1] with mode='exact_device': reference TS1 semantics (global randperm per step), device RNG",
                     "value": nx * ITERS * POP * PARTICLES * HORIZON / ex, "unit": "candidate-steps/s", "ms_per_plan": 1e3 * ex / nx}

    # extra (N = 1 only): the whole drop-in call, TrajectoryOptimizerAgent.act(obs) -> np.ndarray[A]: observation from host
    # memory, plan on the device, ONE D2H of the plan, warm-start shift (what a control loop pays per environment step)
    agent_act = None
    if world == 1 and not args.no_batched:
        cfg = dict(_target_="hipets.CEMOptimizer", num_iterations=ITERS, elite_ratio=ELITE_RATIO, population_size=POP, alpha=ALPHA,
                   device=device, lower_bound="???", upper_bound="???", return_mean_elites=True, seed=0)
        agent = hipets.TrajectoryOptimizerAgent(cfg, [-1.0] * ACT, [1.0] * ACT, planning_horizon=HORIZON)
        agent.set_trajectory_eval_fn(eval_fn)
        for _ in range(3):
            agent.act(s0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        na = max(5, args.steps // 2)
        for _ in range(na):
            agent.act(s0)
        ea = time.perf_counter() - t0  # act() returns host data: it is synchronous by construction
        agent_act = {"workload": "hipets.TrajectoryOptimizerAgent.act(obs) on configs[1], host observation in, host action out",
                     "ms_per_act": 1e3 * ea / na, "acts_per_s": na / ea}

    cand_steps_per_plan = ITERS * pop * PARTICLES * HORIZON
    value = args.steps * cand_steps_per_plan / elapsed
    flops_cs = spec.flops_per_candidate_step()
    # dominant kernel = rollout_kernel: one launch rolls (pop / world) candidates x P particles x H steps
    local_pop = -(-pop // world) if world > 1 else pop  # largest shard
    alg_flops_per_launch = flops_cs * local_pop * PARTICLES * HORIZON
    avg_launch_s = (kernel_ms / max(launches, 1)) * 1e-3
    achieved = alg_flops_per_launch / avg_launch_s / 1e12 if launches else None
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get("rollout_kernel_bytes_per_launch")
        except Exception:
            traffic = None
    out = {
        "metric": "candidate-steps/sec (pop x particles x horizon / plan) per CEM iter; plans/sec",
        "value": value, "unit": "candidate-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
        "scaling": args.scaling if world > 1 else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BASELINE.json configs[1]: PETS HalfCheetah obs=17 act=6, GaussianMLP ensemble=5 (4x200 SiLU, TS1), "
                               f"CEM pop={pop} horizon={HORIZON} particles={PARTICLES} iters={ITERS}; one step = one plan "
                               "(TrajectoryOptimizerAgent.act)",
                   "candidate_steps_per_plan": cand_steps_per_plan, "plans_per_s": args.steps / elapsed,
                   "parallelism": f"population-sharded x{world}" if world > 1 else "single GPU, fused plan",
                   "mode": "FAST (in-kernel Philox, block-balanced TS1)"},
        "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                     "frac": (achieved / PEAK_FP32_TFLOPS) if achieved else None, "traffic": traffic,
                     "kernel": "hipets::rollout_kernel", "launches": launches,
                     "avg_launch_ms": 1e3 * avg_launch_s if launches else None,
                     "algorithmic_flops_per_launch": alg_flops_per_launch, "flops_per_candidate_step": flops_cs},
    }
    if extra is not None:
        out["cfg3_strong"] = extra
    if batched is not None:
        out["batched_planning"] = batched
    if exact_sem is not None:
        out["exact_semantics"] = exact_sem
    if agent_act is not None:
        out["agent_act"] = agent_act
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_budget)
            out["config"]["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
            try:
                out["torch_rocm_port"] = torch_rocm_port(device)
            except Exception as exc:  # informational leg only
                out["torch_rocm_port"] = {"error": str(exc)[:200]}
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist

        dist.destroy_process_group()


if __name__ == "__main__":
    main()
