#!/usr/bin/env python
"""bench.py -- candidate-steps/sec of the PETS planning hot path on MI355X (BASELINE.json metric).

A "step" is ONE PLAN: the whole CEMOptimizer.optimize loop (5 iterations x pop 500 x 20 particles x
horizon 30 rollouts through the 5-member GaussianMLP ensemble + elite refit) = what
TrajectoryOptimizerAgent.act does per environment step, on BASELINE.json configs[1]
("PETS HalfCheetah (obs=17, act=6), ensemble=5, CEM pop=500, horizon=30, particles=20").
Inputs (weights, s0, bounds) are resident in HBM before the timed region; synthetic random-init
weights (reference initialiser), fp32 arithmetic end to end (fp64 input normaliser like the reference).

    python bench.py [--gpus N --steps K --warmup W] [--mode device|fast]     # N>1: launched by torch.distributed.run

--mode (randomness of the rollouts inside the fused plan):
  device  the reference's propagation semantics: ONE balanced permutation of all 10 000 rows per step, iid eps, drawn
          in-kernel; rows change workgroups every step -- through HBM inside one persistent launch per rollout when all
          workgroups are co-resident (cfg2: 210 <= 256 CUs), else one launch per step.  THE HEADLINE.
  fast    one launch per rollout, block-balanced member schedule (reported as the `fast_mode` block of the same line).

N>1 (BASELINE.json configs[2]): the SAME pop-500 plan with its candidates sharded over N ranks (one process per GPU,
"scaling": "strong"), driven by the library itself (hipets_plan_cem_sharded: replicated sampling, local rollouts, ONE
RCCL all-gather of the pop returns per CEM iteration, identical refit on every rank).  `--scaling weak` keeps 500 candidates
per rank instead (pop 500 N).  If the RCCL communicator cannot be created the ranks fall back to independent single-GPU
plans and say so in the line.
"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "mbrl-lib_amd"))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

OBS, ACT, ENSEMBLE, HID, LAYERS = 17, 6, 5, 200, 4
POP, HORIZON, PARTICLES, ITERS, ELITE_RATIO, ALPHA = 500, 30, 20, 5, 0.1, 0.1
METRIC = "candidate-steps/sec (pop\u00d7particles\u00d7horizon / plan) per CEM iter; plans/sec"  # BASELINE.json's metric, verbatim
PEAK_FP32_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32 vector == fp32 MFMA dense peak


PEAK_BF16_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA peak (the sparsity headline is not used)


def synthetic_spec(device, precision="f32", obs=OBS, act=ACT, ensemble=ENSEMBLE, elite=None, obs_process="none", no_delta_list=(),
                   reward="halfcheetah", termination="no_termination", learned_rewards=False, seed=0):
    """Random-init GaussianMLP ensemble with the reference initialiser (models/util.py:15-28: truncated
    normal std 1/(2 sqrt(in)), zero bias; logvar bounds -10 / 0.5), built on the product side (no oracle)."""
    import hipets

    g = torch.Generator().manual_seed(seed)
    n_in = obs + (1 if obs_process == "cartpole_pets" else 0) + act
    n_out = obs + (1 if learned_rewards else 0)  # one_dim_tr_model.py:287: the reward is the model's last output column
    dims = [n_in] + [HID] * LAYERS + [2 * n_out]
    ws, bs = [], []
    for i in range(len(dims) - 1):
        std = 1.0 / (2.0 * np.sqrt(dims[i]))
        w = torch.empty(ensemble, dims[i], dims[i + 1])
        torch.nn.init.trunc_normal_(w, mean=0.0, std=std, a=-2 * std, b=2 * std, generator=g)
        ws.append(w.to(device))
        bs.append(torch.zeros(ensemble, 1, dims[i + 1], device=device))
    return hipets.ModelSpec(
        weights=ws, biases=bs, obs_dim=obs, act_dim=act, min_logvar=-10 * torch.ones(1, n_out), max_logvar=0.5 * torch.ones(1, n_out),
        elite_models=elite, activation="silu", propagation="random_model", norm_mean=torch.zeros(1, n_in, dtype=torch.float64),
        norm_std=torch.ones(1, n_in, dtype=torch.float64), target_is_delta=True, no_delta_list=list(no_delta_list), learned_rewards=learned_rewards,
        obs_process=obs_process, reward=None if learned_rewards else reward, termination=termination, precision=precision)


def synthetic_planet_spec(device, latent=30, action=6, belief=200, hidden=200, seed=0):
    """Random-init PlaNet heads with the reference's initialisers (mbrl/models/planet.py:20-30: orthogonal GRU W_hh, Xavier-uniform
    everything else) at conf/dynamics_model/planet.yaml's sizes, built on the product side (no oracle): what
    PlaNetModel.sample (planet.py:531-581) reads."""
    import hipets

    g = torch.Generator().manual_seed(seed)

    def xavier(out_f, in_f):
        return ((torch.rand(out_f, in_f, generator=g) * 2 - 1) * float(np.sqrt(6.0 / (in_f + out_f)))).to(device)

    def bias(n):
        return ((torch.rand(n, generator=g) * 2 - 1) * 0.05).to(device)

    q, _ = torch.linalg.qr(torch.randn(3 * belief, belief, generator=g))
    return hipets.PlaNetSpec(
        w_embed=xavier(belief, latent + action), b_embed=bias(belief), w_ih=xavier(3 * belief, belief), b_ih=bias(3 * belief),
        w_hh=q.contiguous().to(device), b_hh=bias(3 * belief), w_prior1=xavier(hidden, belief), b_prior1=bias(hidden),
        w_prior2=xavier(2 * latent, hidden), b_prior2=bias(2 * latent), w_rew1=xavier(hidden, belief + latent), b_rew1=bias(hidden),
        w_rew2=xavier(hidden, hidden), b_rew2=bias(hidden), w_rew3=xavier(1, hidden), b_rew3=bias(1), min_std=0.1)


# Other workloads on the same line (never `value`): the configurations the reference ships as its defaults, and the remaining
# BASELINE.json configs (parity-test cases: tests/test_gpu_plans_full_size.py pins each at this size).  Each: model kwargs of
# synthetic_spec, optimizer kind + its stock parameters, P, H.
STOCK_WORKLOADS = {
    # conf/overrides/pets_halfcheetah.yaml:1-24 + conf/dynamics_model/gaussian_mlp_ensemble.yaml + conf/algorithm/pets.yaml:20 +
    # env/pets_halfcheetah.py:91-113: obs 18 through HalfCheetahEnv.preprocess_fn, no_delta_list [0], 7 members / 5 elites
    "pets_halfcheetah": dict(model=dict(obs=18, act=6, ensemble=7, elite=[0, 2, 3, 5, 6], obs_process="halfcheetah", no_delta_list=[0]),
                             optimizer="cem", pop=400, elite_ratio=0.16, alpha=0.12, P=20, H=30,
                             source="conf/overrides/pets_halfcheetah.yaml (the reference's own default PETS workload)"),
    # conf/overrides/pets_cartpole.yaml:1-21 + util/env.py:71-74
    "pets_cartpole": dict(model=dict(obs=4, act=1, ensemble=7, elite=[1, 2, 4, 5, 6], reward="cartpole", termination="cartpole"),
                          optimizer="cem", pop=350, elite_ratio=0.1, alpha=0.1, P=20, H=15, source="conf/overrides/pets_cartpole.yaml"),
    # conf/overrides/pets_pusher.yaml:1-20: learned reward (the model's last output column), no termination function
    "pets_pusher": dict(model=dict(obs=20, act=7, ensemble=7, elite=[0, 1, 3, 4, 6], learned_rewards=True),
                        optimizer="cem", pop=350, elite_ratio=0.1, alpha=0.1, P=20, H=25, source="conf/overrides/pets_pusher.yaml"),
}
OTHER_CONFIGS = {
    "configs[0] cfg1 cartpole": dict(model=dict(obs=4, act=1, ensemble=5, reward="cartpole", termination="cartpole"), optimizer="cem", pop=100,
                                     elite_ratio=0.1, alpha=0.1, P=5, H=15),
    "configs[3] cfg4 iCEM humanoid_truncated_obs (obs 45)": dict(model=dict(obs=45, act=17, ensemble=7, elite=[0, 1, 2, 3, 4], termination="humanoid"),
                                                                  optimizer="icem", pop=1000, P=20, H=40, s0_first=1.4),
    "configs[3] cfg4' iCEM Humanoid-v4 (obs 376)": dict(model=dict(obs=376, act=17, ensemble=7, elite=[0, 1, 2, 3, 4], termination="humanoid"),
                                                         optimizer="icem", pop=1000, P=20, H=40, s0_first=1.4),
    "configs[4] cfg5 MPPI cheetah-run": dict(model=dict(obs=17, act=6, ensemble=5), optimizer="mppi", pop=2000, P=20, H=50),
}


def _usable_cores():
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:  # cgroup v2 CPU quota, if any
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def reference_calibration():
    """port / reference speed ratio from the newest committed calibration (profiles/r*_cpu_baseline_reference_vs_port.json: `python
    bench.py --cpu-baseline-only` where /root/reference is mounted times the UNMODIFIED reference classes and the bitwise port on the
    same plans, same threads).  The GPU box has no reference tree: its `cpu_baseline` is the port, and this ratio converts the port's
    rate into what the reference's own classes would be expected to reach on those cores (the two run the same ATen ops; the port's
    injected-randomness plumbing costs it ~12 %)."""
    import glob

    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r*_cpu_baseline_reference_vs_port.json")),
                   key=lambda f: int(os.path.basename(f)[1:].split("_")[0]))
    if not files:
        return None
    try:
        c = json.load(open(files[-1]))["cpu_baseline"]
        return {"port_over_reference": float(c["port_over_reference"]), "file": "profiles/" + os.path.basename(files[-1]),
                "measured_on": f"{c['host']['cpu_model']}, {c['cores']} threads, {c['plans_timed']} plans each"}
    except Exception:
        return None


def cpu_baseline(budget_s=30.0):
    """kind "reference" where /root/reference is mounted (the reference's own classes), else kind "port":
    the reference's algorithm on this box's host cores: the oracle -- a torch-CPU restatement
    that is BITWISE equal to mbrl-lib's TrajectoryOptimizerAgent + CEMOptimizer + ModelEnv at this exact size
    (tests/test_oracle_full_size.py pins it against a golden recorded from the unmodified reference) -- timed on FULL
    cfg2 plans: one warm-up plan, then >= 5 timed plans (CEM loop included), min and median reported.  Checker code, used
    here ONLY as the reported baseline, never by the product path.  The thread count is calibrated first (torch's
    default of one thread per logical core can be catastrophically slow on a many-core host)."""
    from oracle import pets_oracle as po

    om = po.make_synthetic_model(OBS, ACT, ensemble_size=ENSEMBLE, hid=HID, num_layers=LAYERS, seed=0, nontrivial_stats=False)
    s0 = (np.random.default_rng(0).standard_normal(OBS) * 0.1).astype(np.float32)
    gen = torch.Generator().manual_seed(0)
    torch.manual_seed(0)
    g = torch.Generator().manual_seed(1)
    lb, ub = -torch.ones(HORIZON, ACT), torch.ones(HORIZON, ACT)

    def one_step_batch():
        acts = torch.rand(POP, 1, ACT, generator=g) * 2 - 1
        t0 = time.perf_counter()
        po.rollout(om, acts, s0, PARTICLES, global_rng=True, generator=gen)
        return time.perf_counter() - t0

    def plan():
        obj = lambda pop_: po.rollout(om, pop_, s0, PARTICLES, global_rng=True, generator=gen)  # noqa: E731
        t0 = time.perf_counter()
        po.cem_optimize(obj, torch.zeros(HORIZON, ACT), lb, ub, ITERS, ELITE_RATIO, POP, ALPHA, return_mean_elites=True)
        return time.perf_counter() - t0

    t_start = time.perf_counter()
    usable = _usable_cores()
    best_threads, best_t = 1, None
    for nt in sorted({1, min(8, usable), min(16, usable), min(32, usable), min(64, usable)}):
        torch.set_num_threads(nt)
        one_step_batch()  # warm-up for this thread count
        t = min(one_step_batch(), one_step_batch(), one_step_batch())
        if best_t is None or t < best_t:
            best_threads, best_t = nt, t
        if time.perf_counter() - t_start > 0.2 * budget_s:
            break
    torch.set_num_threads(best_threads)
    plan()  # warm-up plan
    times = []
    while len(times) < 5 or (time.perf_counter() - t_start < budget_s and len(times) < 12):
        times.append(plan())
        if len(times) >= 5 and time.perf_counter() - t_start > budget_s:
            break
    cs = ITERS * POP * PARTICLES * HORIZON
    port = {"value": cs / min(times), "unit": "candidate-steps/s", "ms_per_plan_min": 1e3 * min(times), "plans_timed": len(times)}
    # Where the reference itself is mounted (the build container; never the GPU box), its OWN classes are timed -- the unmodified
    # mbrl.planning.CEMOptimizer + mbrl.models.ModelEnv.evaluate_action_sequences (trajectory_opt.py:142-188, model_env.py:145-191)
    # imported through oracle/ref_bridge.py -- and reported as kind "reference" with the port beside it.
    try:
        from oracle import ref_bridge

        have_ref = ref_bridge.reference_available()
    except Exception:
        have_ref = False
    if have_ref:
        mbrl = ref_bridge.import_reference()
        me, _, _ = ref_bridge.build_reference_model_env(om, OBS, ACT, generator=torch.Generator().manual_seed(0))
        ref_opt = mbrl.planning.CEMOptimizer(ITERS, ELITE_RATIO, POP, lb.tolist(), ub.tolist(), ALPHA, "cpu", return_mean_elites=True)

        def ref_plan():
            t0 = time.perf_counter()
            ref_opt.optimize(lambda a: me.evaluate_action_sequences(a, initial_state=s0, num_particles=PARTICLES), x0=torch.zeros(HORIZON, ACT))
            return time.perf_counter() - t0

        ref_plan()
        rtimes = [ref_plan() for _ in range(max(5, len(times)))]
        return {"value": cs / min(rtimes), "unit": "candidate-steps/s", "cores": best_threads, "kind": "reference",
                "value_median": cs / statistics.median(rtimes), "ms_per_plan_min": 1e3 * min(rtimes), "ms_per_plan_median": 1e3 * statistics.median(rtimes),
                "plans_timed": len(rtimes), "port_beside_it": port, "port_over_reference": port["value"] / (cs / min(rtimes)),
                "sample": f"{len(rtimes)} full cfg2 plans of the UNMODIFIED reference classes (mbrl.planning.CEMOptimizer + ModelEnv.evaluate_action_sequences, "
                          f"{cs} candidate-steps each) after 1 warm-up plan, on {best_threads} threads (calibrated on the port); the bitwise port timed the same way beside it",
                "host": {"nproc": os.cpu_count(), "usable_cores": usable, "cpu_model": _cpu_model(), "torch": torch.__version__},
                "plans_per_s": 1.0 / min(rtimes)}
    cal = reference_calibration()
    return {"value": cs / min(times), "unit": "candidate-steps/s", "cores": best_threads, "kind": "port",
            **({"port_over_reference": cal["port_over_reference"], "reference_equivalent_value": cs / min(times) / cal["port_over_reference"],
                "calibration": cal} if cal else {}),
            "value_median": cs / statistics.median(times), "ms_per_plan_min": 1e3 * min(times),
            "ms_per_plan_median": 1e3 * statistics.median(times), "plans_timed": len(times),
            "sample": f"{len(times)} full cfg2 plans (CEM x {ITERS} iterations x pop {POP} x {PARTICLES} particles x H {HORIZON} = {cs} "
                      f"candidate-steps each) after 1 warm-up plan; min time -> value, median -> value_median; threads calibrated over "
                      f"<= {usable} usable cores",
            "host": {"nproc": os.cpu_count(), "usable_cores": usable, "cpu_model": _cpu_model(), "torch": torch.__version__,
                     "parallel_info": " | ".join(ln.strip() for ln in torch.__config__.parallel_info().splitlines()
                                                 if any(k in ln for k in ("threads", "OpenMP", "MKL")))[:400]},
            "plans_per_s": 1.0 / min(times)}


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
    ap.add_argument("--mode", choices=["device", "fast"], default=None,
                    help="randomness mode of the HEADLINE plan; default: none given -- the objective is built WITHOUT a mode argument, i.e. the "
                         "headline measures the library's default (hipets.make_eval_fn(model, P): 'device', the reference's TS1 semantics)")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", "--no-batched", dest="no_extras", action="store_true", help="skip the extra blocks (other mode, batched planning, agent.act)")
    ap.add_argument("--cpu-budget", type=float, default=30.0)
    ap.add_argument("--cpu-baseline-only", action="store_true", help="print the cpu_baseline object alone (needs no GPU) and exit")
    args = ap.parse_args()
    if args.cpu_baseline_only:
        print(json.dumps({"cpu_baseline": cpu_baseline(args.cpu_budget)}))
        return

    import hipets
    from hipets import dist as hdist
    from hipets.planning import _BoundObjective

    headline_is_default = args.mode is None
    if headline_is_default:  # what `hipets.make_eval_fn(model, P)` runs when the caller names no mode
        args.mode = hipets.HipTrajectoryEvalFn.__init__.__defaults__[1]
        assert args.mode == "device", "the library's default objective mode is expected to be the reference's semantics"

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
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    engine = hipets.get_engine(device)
    spec = synthetic_spec(device)
    lb, ub = [[-1.0] * ACT] * HORIZON, [[1.0] * ACT] * HORIZON
    s0 = (np.random.default_rng(0).standard_normal(OBS) * 0.1).astype(np.float32)
    x0 = torch.zeros(HORIZON, ACT, device=device)
    flops_cs = spec.flops_per_candidate_step()

    # ---- N > 1: how the sharded plan is driven -------------------------------------------------------------------------
    comm_info = {}
    sharded = "single"
    if world > 1:
        if backend == "nccl":
            try:
                hdist.init_engine_comm(engine)  # the library's own RCCL communicator; torch.distributed only carried the id
                sharded = "library"
                seen_rank, seen_world = engine.comm_info()  # ncclCommUserRank / ncclCommCount of the communicator the plans run on
                comm_info["world_seen"] = seen_world
                assert (seen_rank, seen_world) == (rank, world), f"communicator reports rank {seen_rank} of {seen_world}, launched as {rank} of {world}"
            except Exception as exc:  # SURVEY.md section 5 "failure detection": RCCL error -> single-GPU plans + a report
                sharded = "fallback"
                comm_info["fallback_reason"] = f"hipets_comm_init failed: {str(exc)[:200]}"
                print(f"[bench rank {rank}] RCCL communicator unavailable, falling back to independent single-GPU plans: {exc}", file=sys.stderr)
            flags = torch.tensor([1 if sharded == "library" else 0], device=device)
            dist.all_reduce(flags, op=dist.ReduceOp.MIN)  # all ranks take the same path
            if int(flags.item()) == 0 and sharded == "library":
                sharded = "fallback"
                comm_info["fallback_reason"] = "another rank failed to join the communicator"
            try:
                comm_info["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception:
                pass
        else:
            sharded = "torch.distributed"  # gloo: several ranks on one GPU, per-iteration all-gather through the host
            # ranks SHARE a GPU here: their kernels compete for the CUs, so the one-launch form of DEVICE mode (which needs all of
            # its workgroups resident at once) is off; with one process per GPU (the nccl path) it stays on
            engine.set_persistent(False)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def run(mode, pop_total, steps, warmup, spec_=None):
        """`warmup` untimed plans, then exactly `steps` plans between barrier + synchronize; MAX over ranks.
        Returns (seconds, rollout-kernel launches, summed kernel ms, steps per launch, event sampling stride)."""
        mode_kw = {} if (headline_is_default and mode == args.mode) else {"mode": mode}  # the headline: no mode argument at all
        eval_fn = hipets.make_eval_fn(spec_ if spec_ is not None else spec, PARTICLES, engine=engine, seed=0, **mode_kw)
        assert eval_fn.mode == mode
        opt = hipets.CEMOptimizer(ITERS, ELITE_RATIO, pop_total, lb, ub, ALPHA, device, return_mean_elites=True, seed=0)
        if sharded == "library":
            # through the drop-in seam: CEMOptimizer.optimize sees the engine's communicator and runs hipets_plan_cem_sharded under
            # hipets.dist.run_sharded (stream sync + the ranks' agreement on the outcome included: what agent.act() pays per plan)
            objective = _BoundObjective(eval_fn, s0)
            plan = lambda: opt.optimize(objective, x0=x0)  # noqa: E731
        elif sharded == "torch.distributed":
            objective = _BoundObjective(hdist.ShardedEvalFn(eval_fn), s0)  # generic path + one all-gather per iteration
            plan = lambda: opt.optimize(objective, x0=x0)  # noqa: E731
        else:
            objective = _BoundObjective(eval_fn, s0)  # fused hipets_plan_cem
            plan = lambda: opt.optimize(objective, x0=x0)  # noqa: E731
        t_w, i_w = time.perf_counter(), 0
        while i_w < max(0, warmup - 1) or (spec_ is not None and time.perf_counter() - t_w < 0.25):  # (a new model: see measure_workload)
            plan()
            i_w += 1
            if spec_ is not None:
                torch.cuda.synchronize()
        # the last warm-up plan counts the rollout-kernel launches of a plan: FAST mode and the persistent form of DEVICE mode
        # launch once per rollout (the whole horizon), per-step DEVICE mode once per step
        engine.timing_enable(1)
        engine.timing_read(reset=True)
        plan()
        torch.cuda.synchronize()
        per_plan, _ = engine.timing_read(reset=True)
        # hipEvents ride on the dispatch packets of the rollout kernel INSIDE the timed region.  Back-to-back short launches
        # (150 per plan in per-step DEVICE mode) pay ~4.6 us per packet that carries a completion signal (measured: 7.83 vs
        # 7.14 ms per plan), so there every 8th launch is sampled; otherwise every launch is timed.
        stride = 8 if per_plan > 4 * ITERS else 1
        engine.timing_enable(stride)
        engine.timing_read(reset=True)
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            sol = plan()
        barrier()
        el = time.perf_counter() - t0
        launches_, kernel_ms_ = engine.timing_read(reset=True)
        engine.timing_enable(False)
        if world > 1:
            t = torch.tensor([el], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        assert torch.isfinite(sol).all()
        return el, launches_, kernel_ms_, max(1, round(ITERS * HORIZON / max(per_plan, 1))), stride

    def roofline_block(mode, pop_total, launches, kernel_ms, steps_per_launch, stride):
        """Dominant kernel = hipets::rollout_kernel.  One launch covers `steps_per_launch` steps of the local rows: the whole
        horizon in FAST mode and in DEVICE mode's persistent form, ONE step when DEVICE mode launches per step (batches too
        large to be co-resident).  achieved = algorithmic FLOP of a launch / its average duration (hipEvents riding on the
        dispatch packets of the timed region, on the launch stream)."""
        local_pop = -(-pop_total // world) if (world > 1 and sharded != "fallback") else pop_total  # largest shard
        alg = flops_cs * local_pop * PARTICLES * steps_per_launch
        avg_s = (kernel_ms / max(launches, 1)) * 1e-3
        ach = alg / avg_s / 1e12 if launches else None
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(f"rollout_kernel_bytes_per_launch_{mode}")
            except Exception:
                traffic = None
        return {"bound": "mfma", "achieved": ach, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": (ach / PEAK_FP32_TFLOPS) if ach else None,
                "traffic": traffic, "traffic_source": "profiles/hbm_traffic.json: FETCH_SIZE / WRITE_SIZE PMC passes of the same command (profiles/collect.sh), "
                                                      "gfx950 correction of MI355X_MICROARCH.md; read from the committed file, NOT measured by this run",
                "kernel": "hipets::rollout_kernel", "launches": launches, "avg_launch_ms": 1e3 * avg_s if launches else None,
                "algorithmic_flops_per_launch": alg, "flops_per_candidate_step": flops_cs,
                "launch_covers": f"{local_pop} candidates x {PARTICLES} particles x {steps_per_launch} step(s)",
                "launches_timed": "every launch of the timed region" if stride == 1 else f"every {stride}-th launch of the timed region"}

    # N = 1: BASELINE.json configs[1].  N > 1: configs[2] = the SAME pop-500 plan sharded ("strong"); --scaling weak keeps pop 500 / rank
    pop = POP * world if (world > 1 and args.scaling == "weak") else POP
    elapsed, launches, kernel_ms, spl, stride = run(args.mode, pop, args.steps, args.warmup)
    roof = roofline_block(args.mode, pop, launches, kernel_ms, spl, stride)
    if world > 1:
        # every rank's own rollout kernel: its shard (the first pop % world ranks hold one more candidate), its average launch
        # duration from its own hipEvents, priced against the same per-candidate-step FLOP count.  `roofline` above is rank 0's
        # view with the LARGEST shard; this makes a multi-GPU line self-explaining (which rank is slow, how small the shards are)
        lo_, hi_ = hdist.shard_bounds(pop, world, rank) if sharded != "fallback" else (0, pop)
        mine = {"rank": rank, "candidates": hi_ - lo_, "launches": launches, "avg_launch_ms": kernel_ms / max(launches, 1),
                "achieved_tflops": (flops_cs * (hi_ - lo_) * PARTICLES * spl / (kernel_ms / max(launches, 1) * 1e-3) / 1e12) if launches else None}
        mine["frac_of_fp32_peak"] = mine["achieved_tflops"] / PEAK_FP32_TFLOPS if mine["achieved_tflops"] else None
        box = [None] * world
        dist.all_gather_object(box, mine)
        roof["per_rank"] = box
        if args.scaling == "strong":
            comm_info["north_star_target_8gpu_over_1gpu"] = 6.0
            comm_info["design_prediction_8gpu_over_1gpu_strong"] = ("1.9-2.0x: a 63-candidate shard is 80 one-tile workgroups whose step is a ~17-20 us "
                                                                    "latency chain whatever the batch (DESIGN.md section 7); >= 6x holds for the weak-scaled plan only")
    extras = {}
    other = "fast" if args.mode == "device" else "device"
    if not args.no_extras:
        n_other = max(3, args.steps // 3)
        e2, l2, k2, spl2, stride2 = run(other, pop, n_other, 2)
        r2 = roofline_block(other, pop, l2, k2, spl2, stride2)
        extras[f"{other}_mode"] = {"workload": f"the same plan with mode='{other}' rollouts", "value": n_other * ITERS * pop * PARTICLES * HORIZON / e2,
                                   "unit": "candidate-steps/s", "ms_per_plan": 1e3 * e2 / n_other, "roofline": r2}
    if world == 1 and not args.no_extras:
        # SEPARATELY reported arithmetic mode (the graded mode is fp32 MFMA): precision='bf16x3' -- fp32 operands as
        # three bf16 pieces, six exact partial products per product on the bf16 matrix pipe, fp32 accumulate; its own parity
        # evidence is tests/test_gpu_bf16x3.py (same oracle, same tolerances)
        spec3 = synthetic_spec(device, precision="bf16x3")
        blk = {"arithmetic": "fp32 operands split into 3 bf16 pieces, 6 v_mfma_f32_16x16x32_bf16 partial products per 16x16x32 block, fp32 accumulate",
               "parity": "tests/test_gpu_bf16x3.py: T1 / T2 against the oracle at the fp32 mode's tolerances; <= 2e-5 relative to the fp32-MFMA kernel"}
        for m in ("device", "fast"):
            n3 = max(3, args.steps // 3)
            e3_, l3, k3, spl3, stride3 = run(m, pop, n3, 2, spec_=spec3)
            avg_s = (k3 / max(l3, 1)) * 1e-3
            alg = flops_cs * pop * PARTICLES * spl3
            blk[m] = {"value": n3 * ITERS * pop * PARTICLES * HORIZON / e3_, "unit": "candidate-steps/s", "ms_per_plan": 1e3 * e3_ / n3,
                      "roofline": {"bound": "mfma", "achieved": 6 * alg / avg_s / 1e12, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s (bf16 MFMA issued: 6 x algorithmic)",
                                   "frac": 6 * alg / avg_s / 1e12 / PEAK_BF16_TFLOPS, "avg_launch_ms": 1e3 * avg_s, "launches": l3,
                                   "fp32_equivalent_tflops": alg / avg_s / 1e12, "fp32_equivalent_over_fp32_mfma_peak": alg / avg_s / 1e12 / PEAK_FP32_TFLOPS}}
        extras["bf16x3_precision"] = blk
        engine.set_model(spec)  # back to the graded arithmetic for the blocks below
    if world > 1 and not args.no_extras:
        alt_pop = POP if args.scaling == "weak" else POP * world
        n_alt = max(3, args.steps // 3)
        e3 = run(args.mode, alt_pop, n_alt, 2)[0]
        extras["weak_scaling" if args.scaling == "strong" else "cfg3_strong"] = {
            "workload": f"pop {alt_pop} in total ({alt_pop // world} per rank) sharded over {world} ranks", "value": n_alt * ITERS * alt_pop * PARTICLES * HORIZON / e3,
            "unit": "candidate-steps/s", "ms_per_plan": 1e3 * e3 / n_alt}
        if sharded in ("library", "torch.distributed"):  # what one all-gather of the returns costs on this fabric
            buf_in = torch.zeros(-(-POP // world), device=device if backend == "nccl" else "cpu")
            buf_out = torch.zeros(world * buf_in.numel(), device=buf_in.device)
            for _ in range(5):
                dist.all_gather_into_tensor(buf_out, buf_in) if backend == "nccl" else dist.all_gather(list(buf_out.view(world, -1).unbind(0)), buf_in)
            barrier()
            t0 = time.perf_counter()
            for _ in range(50):
                dist.all_gather_into_tensor(buf_out, buf_in) if backend == "nccl" else dist.all_gather(list(buf_out.view(world, -1).unbind(0)), buf_in)
            barrier()
            comm_info["allgather_us_per_iteration"] = 1e6 * (time.perf_counter() - t0) / 50

    if world == 1 and not args.no_extras:
        eval_fast = hipets.make_eval_fn(spec, PARTICLES, engine=engine, seed=0, mode="fast")
        # batched planning: 8 cfg2 environments per launch -- the same kernels with all 256 CUs busy.  Never the headline.
        n_env = 8
        agent = hipets.BatchedCEMAgent(eval_fast, n_env, [-1.0] * ACT, [1.0] * ACT, HORIZON, ITERS, ELITE_RATIO, POP, ALPHA, seed=0)
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
        tf = flops_cs * n_env * POP * PARTICLES * HORIZON / (kms_ / max(lb_, 1) * 1e-3) / 1e12
        extras["batched_planning"] = {"workload": f"{n_env} x configs[1] environments planned in one set of launches (hipets_plan_cem_batched, FAST mode)",
                                      "value": nb * n_env * ITERS * POP * PARTICLES * HORIZON / eb, "unit": "candidate-steps/s",
                                      "ms_per_env_plan": 1e3 * eb / nb / n_env, "rollout_kernel_tflops": tf,
                                      "rollout_kernel_frac_of_fp32_peak": tf / PEAK_FP32_TFLOPS}
        # the whole drop-in call, TrajectoryOptimizerAgent.act(obs) -> np.ndarray[A]: host observation in, plan on the device,
        # ONE D2H of the plan, warm-start shift (what a control loop pays per environment step); same mode as the headline
        cfg = dict(_target_="hipets.CEMOptimizer", num_iterations=ITERS, elite_ratio=ELITE_RATIO, population_size=POP, alpha=ALPHA,
                   device=device, lower_bound="???", upper_bound="???", return_mean_elites=True, seed=0)
        ag = hipets.TrajectoryOptimizerAgent(cfg, [-1.0] * ACT, [1.0] * ACT, planning_horizon=HORIZON)
        ag.set_trajectory_eval_fn(hipets.make_eval_fn(spec, PARTICLES, engine=engine, seed=0, mode=args.mode))
        for _ in range(3):
            ag.act(s0)
        torch.cuda.synchronize()
        # act() returns host data, so it is synchronous by construction: per-call wall times.  A generation-2 pass of Python's
        # garbage collector landing inside the loop costs one act ~40 ms (and the idle GPU a few acts of clock ramp after it):
        # collect before the loop, and report the median beside the mean so that such a host pause is visible, not averaged in
        import gc
        gc.collect()
        na = max(5, args.steps // 2)
        per_act = []
        for _ in range(na):
            t0 = time.perf_counter()
            ag.act(s0)
            per_act.append(time.perf_counter() - t0)
        ea = sum(per_act)
        extras["agent_act"] = {"workload": f"hipets.TrajectoryOptimizerAgent.act(obs) on configs[1] (mode='{args.mode}'), host observation in, host action out",
                               "ms_per_act": 1e3 * ea / na, "acts_per_s": na / ea, "ms_per_act_median": 1e3 * float(np.median(per_act)),
                               "ms_per_act_max": 1e3 * max(per_act), "acts": na}

    def measure_workload(w, mode, n_plans, warm=2):
        """One of STOCK_WORKLOADS / OTHER_CONFIGS through its optimizer class (fused plan: hipets_plan_{cem,mppi,icem}): plan time,
        candidate-steps/s, and the rollout kernel's own launch durations (hipEvents on the dispatch packets of the timed plans)
        priced with SURVEY.md 8(d)'s FLOP formula for THIS model."""
        m = dict(w["model"])
        spec_w = synthetic_spec(device, **m)
        obs_w, act_w, H_w, P_w, kind = m["obs"], m["act"], w["H"], w["P"], w["optimizer"]
        fn = hipets.make_eval_fn(spec_w, P_w, engine=engine, seed=0, mode=mode)
        lb_w, ub_w = [[-1.0] * act_w] * H_w, [[1.0] * act_w] * H_w
        s0_w = (np.random.default_rng(0).standard_normal(obs_w) * 0.1).astype(np.float32)
        if "s0_first" in w:
            s0_w[0] = w["s0_first"]  # a standing humanoid (termination_fns.py:88-95): rollouts that do not end at step 0
        x0_w = torch.zeros(H_w, act_w, device=device)
        if kind == "cem":
            opt = hipets.CEMOptimizer(ITERS, w["elite_ratio"], w["pop"], lb_w, ub_w, w["alpha"], device, return_mean_elites=True, seed=0)
            rows = [w["pop"]] * ITERS
        elif kind == "mppi":  # conf/overrides/pets_mppi_halfcheetah.yaml:19-24
            opt = hipets.MPPIOptimizer(ITERS, w["pop"], 0.9, 1.0, 0.9, lb_w, ub_w, device, seed=0)
            rows = [w["pop"]] * ITERS
        else:  # conf/overrides/pets_icem_cartpole.yaml:16-23; population sizes rounded up to multiples of the ensemble size
            opt = hipets.ICEMOptimizer(ITERS, 0.1, w["pop"], 1.3, 2.0, lb_w, ub_w, 0.3, 0.1, device, return_mean_elites=True,
                                       population_size_module=m["ensemble"], seed=0)
            # every plan but the very first evaluates the kept elites too (and the extra `mu` row in its last iteration)
            rows = [opt._iteration_size(i) + (1 if i == ITERS - 1 else int(opt.keep_elite_size)) for i in range(ITERS)]
        objective = _BoundObjective(fn, s0_w)
        plan = lambda: opt.optimize(objective, x0=x0_w)  # noqa: E731
        # warm up for >= 0.25 s of wall time: building a model is host-only work, and the first tens of ms of GPU work after it run
        # far below the steady-state rate (measured: a cfg1 plan 4.2 ms in a 60 ms burst, 1.34 ms steady; profiles/plan_gap_probe.py)
        t_w, i_w = time.perf_counter(), 0
        while i_w < warm or time.perf_counter() - t_w < 0.25:
            plan()
            torch.cuda.synchronize()
            i_w += 1
        engine.timing_enable(1)
        engine.timing_read(reset=True)
        plan()
        torch.cuda.synchronize()
        per_plan, _ = engine.timing_read(reset=True)
        stride_w = 8 if per_plan > 4 * ITERS else 1
        engine.timing_enable(stride_w)
        engine.timing_read(reset=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n_plans):
            sol = plan()
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / n_plans
        n_l, k_ms = engine.timing_read(reset=True)
        engine.timing_enable(False)
        assert torch.isfinite(sol).all()
        cs = sum(rows) * P_w * H_w
        fl = spec_w.flops_per_candidate_step()
        avg_ms = k_ms / max(n_l, 1)
        kernel_ms_per_plan = avg_ms * per_plan
        ach = cs * fl / (kernel_ms_per_plan * 1e-3) / 1e12
        # which rollout-kernel instance each iteration's population runs, and with how many row tiles per workgroup (iCEM's decaying
        # populations land on different geometries: hipets_kernel_class)
        inst = [list(engine.kernel_class(r_, P_w, H_w, mode)) for r_ in sorted(set(rows), reverse=True)]
        return {"ms_per_plan": 1e3 * el, "value": cs / el, "unit": "candidate-steps/s", "candidate_steps_per_plan": cs,
                "candidates_per_iteration": rows, "kernel_instance_per_population_size": dict(zip(map(str, sorted(set(rows), reverse=True)), inst)),
                "plans_timed": n_plans,
                "roofline": {"bound": "mfma", "achieved": ach, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_FP32_TFLOPS,
                             "kernel": "hipets::rollout_kernel", "launches_per_plan": per_plan, "avg_launch_ms": avg_ms,
                             "rollout_kernel_ms_per_plan": kernel_ms_per_plan, "flops_per_candidate_step": fl,
                             "algorithmic_flops_per_plan": cs * fl,
                             "launches_timed": "every launch of the timed plans" if stride_w == 1 else f"every {stride_w}-th launch of the timed plans"},
                "plan_frac_of_fp32_peak_end_to_end": cs * fl / el / 1e12 / PEAK_FP32_TFLOPS}

    if world == 1 and not args.no_extras:
        # the configurations the reference ships as its defaults (what `python -m mbrl.examples.main algorithm=pets overrides=...`
        # plans with): obs preprocessing, no_delta_list, 7 members / 5 elites, their own population sizes -- both randomness modes
        stock = {}
        for name, w in STOCK_WORKLOADS.items():
            stock[name] = {"workload": f"{w['source']}: obs {w['model']['obs']} act {w['model']['act']}, {w['model']['ensemble']} members / "
                                       f"{len(w['model']['elite'])} elites, CEM pop {w['pop']} x {w['P']} particles x H {w['H']}, elite ratio "
                                       f"{w['elite_ratio']}, alpha {w['alpha']}, {ITERS} iterations"}
            for m_ in (args.mode, other):
                stock[name][m_] = measure_workload(w, m_, max(4, args.steps // 4))
        extras["stock_defaults"] = stock
        # the other BASELINE.json configs, headline mode (parity: tests/test_gpu_plans_full_size.py at exactly these sizes)
        others = {}
        for name, w in OTHER_CONFIGS.items():
            others[name] = {"optimizer": w["optimizer"], args.mode: measure_workload(w, args.mode, 3 if w["pop"] >= 1000 else max(4, args.steps // 4))}
        others["configs[3] cfg4' iCEM Humanoid-v4 (obs 376)"][other] = measure_workload(OTHER_CONFIGS["configs[3] cfg4' iCEM Humanoid-v4 (obs 376)"], other, 3)
        extras["other_configs"] = others
        engine.set_model(spec)

    if world == 1 and not args.no_extras:
        # SURVEY.md 8(f) rows with parity and, until round 5, no driver-run number.
        # f2 -- ModelEnv.step at MBPO's shape (mbrl/algorithms/mbpo.py:30-63 rollout_model_and_populate_sac_buffer: ONE model
        # transition per call for a batch of rows; conf/overrides/mbpo_halfcheetah.yaml:13-15: 400 x 250 = 100 000 rows; obs 17 /
        # act 6, conf/dynamics_model/gaussian_mlp_ensemble.yaml: 7 members / 5 elites): hipets_step, both randomness modes.
        spec_s = synthetic_spec(device, ensemble=7, elite=[0, 1, 2, 3, 4])
        engine.set_model(spec_s)
        B_s = 100_000
        g_s = torch.Generator().manual_seed(3)
        obs_b = (torch.randn(B_s, OBS, generator=g_s) * 0.1).to(device)
        act_b = (torch.rand(B_s, ACT, generator=g_s) * 2 - 1).to(device)
        blk_s = {"workload": f"hipets_step (ModelEnv.step, mbrl/models/model_env.py:87-140) on {B_s} rows per call, obs {OBS} act {ACT}, 7 members / 5 elites "
                             "(MBPO's model rollouts: mbrl/algorithms/mbpo.py:30-63, conf/overrides/mbpo_halfcheetah.yaml:13-15), next_obs / rewards / dones "
                             "written to HBM every call"}
        fl_s = spec_s.flops_per_candidate_step()
        for m_ in ("device", "fast"):
            t_w, i_w = time.perf_counter(), 0
            while i_w < 3 or time.perf_counter() - t_w < 0.25:
                engine.step(obs_b, act_b, mode=m_, seed=1, stream_id=i_w)
                torch.cuda.synchronize()
                i_w += 1
            n_s = max(10, args.steps)
            engine.timing_enable(1)
            engine.timing_read(reset=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(n_s):
                engine.step(obs_b, act_b, mode=m_, seed=1, stream_id=i)
            torch.cuda.synchronize()
            el_s = (time.perf_counter() - t0) / n_s
            n_l, k_ms = engine.timing_read(reset=True)
            engine.timing_enable(False)
            avg_ms = k_ms / max(n_l, 1)
            ach = B_s * fl_s / (avg_ms * 1e-3) / 1e12 if n_l else None
            blk_s[m_] = {"value": B_s / el_s, "unit": "rows (model transitions)/s", "ms_per_call": 1e3 * el_s, "calls_timed": n_s,
                         "roofline": {"bound": "mfma", "achieved": ach, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_FP32_TFLOPS if ach else None,
                                      "kernel": "hipets::rollout_kernel (one step per launch)", "launches_per_call": n_l / n_s, "avg_launch_ms": avg_ms,
                                      "flops_per_candidate_step": fl_s, "algorithmic_flops_per_launch": B_s * fl_s,
                                      "algorithmic_hbm_bytes_per_call": 4 * B_s * (OBS + ACT + OBS + 1) + B_s}}
        extras["model_env_step"] = blk_s
        # f4 -- the PlaNet latent planner (mbrl/models/planet.py:531-581 through ModelEnv.evaluate_action_sequences and the CEM loop,
        # conf/overrides/planet_cheetah_run.yaml:29-35: clipped-normal CEM, pop 1000, H 12, 10 iterations, alpha 0; conf/dynamics_model/
        # planet.yaml sizes: latent 30, belief 200, hidden 200): ONE library call per plan (hipets_plan_planet_cem)
        spec_p = synthetic_planet_spec(device)
        engine.planet_set_model(spec_p)
        fn_p = hipets.make_eval_fn(spec_p, 1, engine=engine, seed=0)
        fn_p.set_state(torch.zeros(1, 30), torch.zeros(1, 200))
        P_ITERS, P_POP, P_H = 10, 1000, 12
        opt_p = hipets.CEMOptimizer(P_ITERS, 0.1, P_POP, [[-1.0] * ACT] * P_H, [[1.0] * ACT] * P_H, 0.0, device, return_mean_elites=True,
                                    clipped_normal=True, seed=1)
        obj_p = _BoundObjective(fn_p, np.zeros((3, 64, 64), np.float32))
        x0_p = torch.zeros(P_H, ACT, device=device)
        plan_p = lambda: opt_p.optimize(obj_p, x0=x0_p)  # noqa: E731
        t_w, i_w = time.perf_counter(), 0
        while i_w < 3 or time.perf_counter() - t_w < 0.25:
            plan_p()
            torch.cuda.synchronize()
            i_w += 1
        n_p = max(5, args.steps // 4)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n_p):
            sol_p = plan_p()
        torch.cuda.synchronize()
        el_p = (time.perf_counter() - t0) / n_p
        assert torch.isfinite(sol_p).all()
        # the rollout kernel alone: one launch per CEM iteration (pop x H candidate-steps), torch events around a burst of launches on
        # the current stream (the library launches on it)
        acts_p = (torch.rand(P_POP, P_H, ACT, generator=g_s) * 2 - 1).to(device)
        l0_p, b0_p = torch.zeros(30, device=device), torch.zeros(200, device=device)
        for _ in range(5):
            engine.planet_rollout(acts_p, l0_p, b0_p, 1, seed=1)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n_r = 20
        ev0.record()
        for _ in range(n_r):
            engine.planet_rollout(acts_p, l0_p, b0_p, 1, seed=1)
        ev1.record()
        torch.cuda.synchronize()
        roll_ms = ev0.elapsed_time(ev1) / n_r
        fl_p = spec_p.flops_per_candidate_step()
        cs_p = P_ITERS * P_POP * P_H
        ach_p = P_POP * P_H * fl_p / (roll_ms * 1e-3) / 1e12
        extras["planet"] = {"workload": "PlaNet latent planner, conf/overrides/planet_cheetah_run.yaml:29-35 (clipped-normal CEM, pop 1000, H 12, 10 iterations, alpha 0) on "
                                        "conf/dynamics_model/planet.yaml sizes (latent 30, belief 200, hidden 200, action 6), one particle; one hipets_plan_planet_cem call per plan",
                            "ms_per_plan": 1e3 * el_p, "value": cs_p / el_p, "unit": "candidate-steps/s", "candidate_steps_per_plan": cs_p, "plans_timed": n_p,
                            "roofline": {"bound": "mfma", "achieved": ach_p, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": ach_p / PEAK_FP32_TFLOPS,
                                         "kernel": "hipets::planet_rollout_kernel", "avg_launch_ms": roll_ms, "flops_per_candidate_step": fl_p,
                                         "algorithmic_flops_per_launch": P_POP * P_H * fl_p,
                                         "launches_timed": f"{n_r} back-to-back hipets_planet_rollout launches (pop {P_POP} x H {P_H}) between two events on the launch stream, "
                                                           "including the ~2 us launch gap each"},
                            "plan_frac_of_fp32_peak_end_to_end": cs_p * fl_p / el_p / 1e12 / PEAK_FP32_TFLOPS}
        engine.set_model(spec)

    cand_steps_per_plan = ITERS * pop * PARTICLES * HORIZON
    plans_done = args.steps * (world if sharded == "fallback" else 1)  # fallback: every rank planned on its own
    value = plans_done * cand_steps_per_plan / elapsed
    mode_text = {"device": "DEVICE (reference TS1 semantics: one balanced permutation of all rows per step + iid eps, drawn in-kernel; "
                           + ("one persistent launch per rollout, rows handed over between workgroups through HBM)" if spl > 1 else "one launch per step)"),
                 "fast": "FAST (in-kernel Philox, block-balanced TS1, one launch per rollout)"}[args.mode]
    if world == 1:
        workload = ("BASELINE.json configs[1]: PETS HalfCheetah obs=17 act=6, GaussianMLP ensemble=5 (4x200 SiLU, TS1), "
                    f"CEM pop={pop} horizon={HORIZON} particles={PARTICLES} iters={ITERS}; one step = one plan (TrajectoryOptimizerAgent.act)")
    elif args.scaling == "strong":
        workload = (f"BASELINE.json configs[2]: the configs[1] plan (CEM pop={pop} horizon={HORIZON} particles={PARTICLES} iters={ITERS}, ensemble=5) "
                    f"with its population sharded across {world} GPUs, one RCCL all-gather of the returns per CEM iteration")
    else:
        workload = (f"configs[1] scaled weakly: pop {pop} = {POP} per rank over {world} GPUs (configs[2] is the `cfg3_strong` block)")
    out = {
        "metric": METRIC,
        "value": value, "unit": "candidate-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
        "scaling": args.scaling if world > 1 else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": workload, "candidate_steps_per_plan": cand_steps_per_plan, "plans_per_s": plans_done / elapsed,
                   "parallelism": {"single": "single GPU, fused plan (hipets_plan_cem)",
                                   "library": f"population-sharded x{world}, in-library RCCL (hipets_plan_cem_sharded)",
                                   "torch.distributed": f"population-sharded x{world}, torch.distributed {backend} all-gather per iteration",
                                   "fallback": f"FALLBACK: {world} independent single-GPU plans (no communicator)"}[sharded],
                   "mode": mode_text,
                   "mode_is_the_library_default": bool(headline_is_default)},
        "roofline": roof,
    }
    if comm_info:
        out["comm"] = comm_info
    out.update(extras)
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_budget)
            out["config"]["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
            # against the reference's OWN classes: measured directly where they are mounted (kind "reference"), else the port's rate
            # scaled by the committed calibration "port / reference" (round-5 verdict: the bitwise port runs slower than the classes it
            # restates, so GPU / port overstates GPU / reference)
            por = out["cpu_baseline"].get("port_over_reference") if out["cpu_baseline"]["kind"] == "port" else 1.0
            if por:
                out["config"]["gpu_over_cpu_reference_equivalent"] = out["config"]["gpu_over_cpu"] * por
            try:
                out["torch_rocm_port"] = torch_rocm_port(device)
            except Exception as exc:  # informational leg only
                out["torch_rocm_port"] = {"error": str(exc)[:200]}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
