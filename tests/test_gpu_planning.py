"""Optimizer kernels and the host-side drop-in classes on the GPU, against the reference's golden vectors
(tests/golden/cem_*.npz) and the oracle.  Tolerances: T3 elite sets equal; T4 refit mu/var and
returned plan atol 1e-4 (measured ~1e-7)."""
import glob
import json
import os

import numpy as np
import pytest
import torch

import hipets
from conftest import GOLDEN, to_spec
from oracle import pets_oracle as po
from oracle import device_draws
from oracle.golden_io import load_case

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
CEM_FILES = sorted(glob.glob(os.path.join(GOLDEN, "cem_*.npz")))


def load_cem(path):
    z = np.load(path)
    meta = json.loads(bytes(z["meta_json"]).decode())
    return meta, {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("x_")}


@pytest.mark.parametrize("path", CEM_FILES, ids=lambda p: os.path.basename(p)[4:-4])
def test_cem_kernels_match_reference_iteration_by_iteration(engine, path):
    meta, a = load_cem(path)
    pop, H, A, iters = meta["pop"], meta["H"], meta["A"], meta["iters"]
    K = po.elite_count(pop, meta["elite_ratio"])
    p = engine.cem_params(pop, H, A, iters, K, meta["alpha"], meta["return_mean"], meta["clipped"])
    lower, upper = a["lower"].to(DEV), a["upper"].to(DEV)
    mu = a["x0"].clone().to(DEV)
    disp = (torch.ones(H, A) if meta["clipped"] else ((a["upper"] - a["lower"]) ** 2) / 16).to(DEV)
    best_v = torch.full((1,), -float("inf"), device=DEV)
    best = torch.zeros(H, A, device=DEV)
    population = torch.empty(pop, H, A, device=DEV)
    eidx = torch.empty(K, dtype=torch.int32, device=DEV)
    for i in range(iters):
        engine.cem_sample(p, mu, disp, lower, upper, population, z=a["z"][i].to(DEV))
        assert torch.allclose(population.cpu(), a["populations"][i], rtol=0, atol=1e-6)
        raw = a["values"][i].clone()
        raw[meta["nan_index"]] = float("nan")
        values = raw.to(DEV)
        engine.cem_refit(p, values, population, mu, disp, best_v, best, eidx)
        assert values[meta["nan_index"]].item() == pytest.approx(-1e-10)  # Appendix B1, in place like the reference
        assert set(eidx.cpu().tolist()) == set(a["elite_idx"][i].tolist())  # T3
        assert eidx[0].item() == a["elite_idx"][i][0].item()
        assert torch.allclose(mu.cpu(), a["mus"][i], rtol=0, atol=1e-5)  # T4
        assert torch.allclose(disp.cpu(), a["disps"][i], rtol=1e-5, atol=1e-6)
    result = mu if meta["return_mean"] else best
    assert torch.allclose(result.cpu(), a["result"], rtol=0, atol=1e-5)


@pytest.mark.parametrize("path", CEM_FILES, ids=lambda p: os.path.basename(p)[4:-4])
def test_cem_optimizer_class_generic_path(path):
    """hipets.CEMOptimizer with an arbitrary Python objective (the diagnostics/control_env.py use of the seam)."""
    meta, a = load_cem(path)
    target = a["target"].to(DEV)

    def obj(x):
        v = -((x - target) ** 2).sum(dim=(1, 2)).clone()
        v[meta["nan_index"]] = float("nan")
        return v

    seen = []
    opt = hipets.CEMOptimizer(meta["iters"], meta["elite_ratio"], meta["pop"], a["lower"].tolist(), a["upper"].tolist(),
                              meta["alpha"], DEV, return_mean_elites=meta["return_mean"], clipped_normal=meta["clipped"])
    out = opt.optimize(obj, x0=a["x0"], callback=lambda pop_, v, i: seen.append((pop_.shape, i)), noise=list(a["z"]))
    assert out.device.type == "cuda" and tuple(out.shape) == (meta["H"], meta["A"])
    assert torch.allclose(out.cpu(), a["result"], rtol=0, atol=1e-5)
    assert [s[1] for s in seen] == list(range(meta["iters"]))


def test_cem_rosenbrock_known_answer():
    """notebooks/cem_rosenbrock_ex.ipynb cell 2: CEMOptimizer(5, 0.01, 1000, [-2,-2], [2,2], 0.1) -> ~[1, 1]."""
    def rosenbrock(x, a=1.0, b=100.0):
        return -((a - x[:, 0]) ** 2 + b * (x[:, 1] - x[:, 0] ** 2) ** 2)

    opt = hipets.CEMOptimizer(5, 0.01, 1000, [-2.0, -2.0], [2.0, 2.0], 0.1, torch.device(DEV), seed=0)
    best = opt.optimize(rosenbrock, torch.zeros(2))
    assert tuple(best.shape) == (2,)
    assert torch.allclose(best.cpu(), torch.ones(2), atol=0.05)
    assert rosenbrock(best.view(1, 2)).item() > -1e-2


def test_philox_truncated_normal_sampler(engine):
    """tests/core/test_common_utils.py:419-423 (support) + the law: N(0,1) truncated to [-2,2] has var 0.7737."""
    pop, H, A = 4096, 10, 5
    p = engine.cem_params(pop, H, A, 1, 10, 0.1)
    z = torch.empty(pop, H, A, device=DEV)
    one, zero = torch.ones(H, A, device=DEV), torch.zeros(H, A, device=DEV)
    # lower/upper far away, dispersion 1 => population == z
    engine.cem_sample(p, zero, one, -1e3 * one, 1e3 * one, z, seed=3, stream_id=1)
    v = z.double().cpu().flatten()
    assert (v > -2).all() and (v < 2).all()
    n = v.numel()
    assert abs(v.mean()) < 5 * np.sqrt(0.7737 / n)
    assert abs(v.var() - 0.77374) < 0.01
    z2 = torch.empty_like(z)
    engine.cem_sample(p, zero, one, -1e3 * one, 1e3 * one, z2, seed=3, stream_id=1)
    assert torch.equal(z, z2)
    engine.cem_sample(p, zero, one, -1e3 * one, 1e3 * one, z2, seed=3, stream_id=2)
    assert not torch.equal(z, z2)
    # bound-aware variance (trajectory_opt.py:122-127): near a bound samples stay inside it
    mu = 0.9 * one
    engine.cem_sample(p, mu, one, -one, one, z, seed=4, stream_id=0)
    assert (z <= 1.0).all() and (z >= -1.0).all()


def _model_env_like(om, obs, act):
    class Space:
        def __init__(self, n):
            self.shape = (n,)
            self.low, self.high = -np.ones(n, np.float32), np.ones(n, np.float32)

    class ME:
        observation_space, action_space = Space(obs), Space(act)

    return ME()


def test_agent_act_fused_path_is_deterministic_and_improves_return(engine):
    """TrajectoryOptimizerAgent.act (trajectory_opt.py:655-694) end to end on the fused CEM plan."""
    obs, act, H, P, pop = 17, 6, 12, 10, 200
    om = po.make_synthetic_model(obs, act, ensemble_size=5, hid=64, seed=3)
    om.max_logvar = torch.full_like(om.max_logvar, -8.0)  # low-noise model: returns are dominated by the actions
    spec = to_spec(om, obs, act)
    cfg = dict(_target_="hipets.CEMOptimizer", num_iterations=4, elite_ratio=0.1, population_size=pop, alpha=0.1,
               device=DEV, lower_bound="???", upper_bound="???", return_mean_elites=True, seed=11)

    def make_agent():
        agent = hipets.TrajectoryOptimizerAgent(cfg, [-1.0] * act, [1.0] * act, planning_horizon=H)
        with pytest.raises(RuntimeError, match="set_trajectory_eval_fn"):  # trajectory_opt.py:673-676
            agent.act(np.zeros(obs))
        agent.set_trajectory_eval_fn(hipets.make_eval_fn(spec, P, engine=engine, seed=5))
        return agent

    s0 = (np.random.default_rng(0).standard_normal(obs) * 0.1).astype(np.float32)
    a1, a2 = make_agent(), make_agent()
    act1, act2 = a1.act(s0), a2.act(s0)
    assert act1.shape == (act,) and act1.dtype == np.float32
    assert np.array_equal(act1, act2)  # same seeds -> identical action selection
    plan = a1.plan(s0)
    assert plan.shape == (H, act) and (np.abs(plan) <= 1).all()
    # the optimised plan beats random action sequences under the model (scored by the oracle with independent
    # randomness; the control cost makes random plans score about -2.5 here and a converged plan about 0)
    g = torch.Generator().manual_seed(0)
    cands = torch.cat([torch.from_numpy(plan)[None], torch.rand(9, H, act, generator=g) * 2 - 1])
    B = 10 * 50
    perms = torch.stack([torch.randperm(B, generator=g) for _ in range(H)])
    eps = torch.randn(H, B, obs, generator=g)
    r = po.rollout(om, cands, s0, 50, perms=perms, eps=eps)
    assert r[0] > r[1:].max() + 1.0
    # warm start: previous solution is the plan shifted by replan_freq with (lb+ub)/2 refill (:563-567)
    prev = a1.optimizer.previous_solution.cpu().numpy()
    assert np.allclose(prev[:-1], plan[1:]) and np.allclose(prev[-1], 0.0)
    a1.reset()
    assert torch.equal(a1.optimizer.previous_solution, a1.optimizer.initial_solution)


def test_generic_and_fused_paths_agree_statistically(engine):
    """The fused hipets_plan_cem and the per-iteration generic path are the same algorithm: both produce plans of
    the same quality (scored by the oracle), far better than random action sequences."""
    obs, act, H, P, pop = 17, 6, 8, 10, 300
    om = po.make_synthetic_model(obs, act, ensemble_size=5, hid=32, seed=4)
    om.max_logvar = torch.full_like(om.max_logvar, -8.0)
    fn = hipets.make_eval_fn(to_spec(om, obs, act), P, engine=engine, seed=1)
    from hipets.planning import _BoundObjective

    lb, ub = [[-1.0] * act] * H, [[1.0] * act] * H
    opt = hipets.CEMOptimizer(5, 0.1, pop, lb, ub, 0.1, DEV, return_mean_elites=True, seed=2)
    s0 = np.zeros(obs, np.float32)
    fused = opt.optimize(_BoundObjective(fn, s0), x0=torch.zeros(H, act))
    generic = opt.optimize(_BoundObjective(fn, s0), x0=torch.zeros(H, act), force_generic=True)
    assert torch.isfinite(fused).all() and torch.isfinite(generic).all()
    g = torch.Generator().manual_seed(0)
    cands = torch.cat([fused.cpu()[None], generic.cpu()[None], torch.rand(8, H, act, generator=g) * 2 - 1])
    B = 10 * 50
    perms = torch.stack([torch.randperm(B, generator=g) for _ in range(H)])
    eps = torch.randn(H, B, obs, generator=g)
    r = po.rollout(om, cands, s0, 50, perms=perms, eps=eps)
    assert r[0] > r[2:].max() + 0.5 and r[1] > r[2:].max() + 0.5
    assert abs(r[0] - r[1]) < 0.3


def test_eval_fn_exact_mode_replays_reference_rng_order(engine):
    """mode='exact' consumes torch's RNGs in the reference's order, so with the same
    seeds it reproduces ModelEnv.evaluate_action_sequences (here: the oracle drawing in that same order)."""
    obs, act, pop, P, H = 17, 6, 20, 5, 6
    om = po.make_synthetic_model(obs, act, ensemble_size=5, hid=48, seed=6)
    g = torch.Generator().manual_seed(1)
    actions = torch.rand(pop, H, act, generator=g) * 2 - 1
    s0 = np.zeros(obs, np.float32)
    fn = hipets.make_eval_fn(to_spec(om, obs, act), P, engine=engine, mode="exact", rng=torch.Generator().manual_seed(3))
    torch.manual_seed(8)
    out = fn(s0, actions.to(DEV))
    torch.manual_seed(8)
    ref = po.rollout(om, actions, s0, P, global_rng=True, generator=torch.Generator().manual_seed(3))
    assert torch.allclose(out.cpu(), ref, rtol=0, atol=1e-4)
    fn3 = hipets.make_eval_fn(to_spec(om, obs, act), 3, engine=engine, mode="fast")
    with pytest.raises(ValueError, match="multiple of the number of models"):  # 4 x 3 rows, 5 members
        fn3(s0, actions[:4].to(DEV))


def test_create_agent_for_model_env_and_resnapshot(engine):
    """create_trajectory_optim_agent_for_model (trajectory_opt.py:719-749) on a duck-typed ModelEnv; the
    objective re-packs weights when the live parameters change."""
    from test_host_logic import _FakeModelEnv

    me = _FakeModelEnv()
    me.device = DEV
    for layer in me.dynamics_model.model.hidden_layers:
        torch.nn.init.normal_(layer[0].weight, std=0.3)
    cfg = dict(_target_="hipets.TrajectoryOptimizerAgent", action_lb="???", action_ub="???", planning_horizon=4,
               optimizer_cfg=dict(_target_="mbrl.planning.CEMOptimizer", num_iterations=2, elite_ratio=0.2, population_size=30,
                                  alpha=0.1, device=DEV, lower_bound="???", upper_bound="???", return_mean_elites=True))
    agent = hipets.create_trajectory_optim_agent_for_model(me, cfg, num_particles=3, engine=engine)
    assert isinstance(agent.optimizer.optimizer, hipets.CEMOptimizer)  # stock target redirected to the fused class
    a = agent.act(np.zeros(5, np.float32))
    assert a.shape == (2,)
    fn = agent.trajectory_eval_fn
    v0 = fn._version
    with torch.no_grad():
        me.dynamics_model.model.mean_and_logvar.weight.add_(0.1)
    agent.act(np.zeros(5, np.float32))
    assert fn._version != v0


def _load_npz(name):
    z = np.load(os.path.join(GOLDEN, name))
    meta = json.loads(bytes(z["meta_json"]).decode())
    return meta, {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("x_")}


def _quad_dev(target, nan_at):
    t = target.to(DEV)

    def f(x):
        v = -((x - t) ** 2).sum(dim=(1, 2)).clone()
        v[nan_at] = float("nan")
        return v

    return f


def test_mppi_optimizer_matches_reference_two_calls():
    """hipets.MPPIOptimizer against the reference's golden run: same injected truncated normals, two consecutive
    plans (pins the persistent mean, the shifted past_action alias and the dead sigma, Appendix B4-B6)."""
    meta, a = _load_npz("mppi_two_calls.npz")
    obj = _quad_dev(a["target"], meta["nan_index"])
    seen = []
    opt = hipets.MPPIOptimizer(meta["iters"], meta["pop"], meta["gamma"], meta["sigma"], meta["beta"], a["lower"].tolist(),
                               a["upper"].tolist(), DEV)
    for c in range(meta["calls"]):
        out = opt.optimize(obj, noise=list(a[f"noise{c}"]),
                           callback=lambda p_, v, k, c=c: seen.append((c, k, p_.detach().cpu().clone(), v.detach().cpu().clone())))
        assert torch.allclose(out.cpu(), a[f"result{c}"], rtol=0, atol=1e-5)
    for c, k, pop_, v in seen:
        assert torch.allclose(pop_, a[f"populations{c}"][k], rtol=0, atol=1e-5)
        assert v[meta["nan_index"]].item() == pytest.approx(-1e-10)
    other = hipets.MPPIOptimizer(meta["iters"], meta["pop"], meta["gamma"], 5.0, meta["beta"], a["lower"].tolist(),
                                 a["upper"].tolist(), DEV)  # Appendix B5: sigma is dead
    assert torch.allclose(other.optimize(obj, noise=list(a["noise0"])).cpu(), a["result0"], rtol=0, atol=1e-5)


def test_icem_kernels_and_optimizer_match_reference_two_calls(engine):
    """Coloured-noise kernel (device inverse real DFT) from the recorded unit normals, then the whole
    hipets.ICEMOptimizer with the reference's recorded draws injected: population sizes, kept / shifted elites,
    appended mean, biased variance, persistent elite set across two plans."""
    meta, a = _load_npz("icem_two_calls.npz")
    H, A = meta["H"], meta["A"]
    lower, upper = a["lower"].to(DEV), a["upper"].to(DEV)
    # kernel level: iteration 0 of call 0 has no appended rows
    n0 = a["normals_0_0"].shape[1]
    pop0 = torch.empty(n0, H, A, device=DEV)
    var0 = (((a["upper"] - a["lower"]) ** 2) / 16).to(DEV)
    engine.icem_sample(n0, H, A, meta["exponent"], a["x0_0"].to(DEV), var0, lower, upper, pop0, normals=a["normals_0_0"].to(DEV))
    assert torch.allclose(pop0.cpu(), a["population_0_0"], rtol=0, atol=2e-5)
    # law of the Philox-driven coloured noise == law of the reference's sampler (oracle, torch RNG): pooled variance
    # (1 from the sigma normalisation of util/math.py:361-363 plus the un-normalised DC term) and lag-1 correlation
    big = torch.empty(8000, 30, 6, device=DEV)
    one = torch.ones(30, 6, device=DEV)
    engine.icem_sample(8000, 30, 6, 2.0, 0 * one, one, -1e3 * one, 1e3 * one, big, seed=3, stream_id=9)
    torch.manual_seed(0)
    ref = po.powerlaw_psd_gaussian(2.0, size=(8000, 6, 30)).transpose(1, 2).double()
    bigd = big.double().cpu()
    assert abs(bigd.var().item() - ref.var().item()) < 0.05 and abs(bigd.mean().item()) < 0.03
    lag = lambda x: (x[:, 1:] * x[:, :-1]).mean().item()  # noqa: E731
    assert abs(lag(bigd) - lag(ref)) < 0.05 and lag(bigd) > 0.5  # beta = 2: strongly correlated in time
    # optimizer level
    obj = _quad_dev(a["target"], meta["nan_index"])
    opt = hipets.ICEMOptimizer(meta["iters"], meta["elite_ratio"], meta["pop"], meta["decay"], meta["exponent"], a["lower"].tolist(),
                               a["upper"].tolist(), meta["keep_frac"], meta["alpha"], DEV, return_mean_elites=True,
                               population_size_module=meta["module"])
    for c in range(meta["calls"]):
        inject, sizes = [], []
        for i in range(meta["iters"]):
            d = {"normals": a[f"normals_{c}_{i}"]}
            for k in ("keep_perm", "end_noise"):
                if f"{k}_{c}_{i}" in a:
                    d[k] = a[f"{k}_{c}_{i}"]
            inject.append(d)
        pops = []
        out = opt.optimize(obj, x0=a[f"x0_{c}"], inject=inject, callback=lambda p_, v, i: pops.append(p_.detach().cpu().clone()))
        assert [int(p_.shape[0]) for p_ in pops] == meta["evaluated_sizes"][c]
        for i, p_ in enumerate(pops):
            assert torch.allclose(p_, a[f"population_{c}_{i}"], rtol=0, atol=5e-5), (c, i)
        assert torch.allclose(out.cpu(), a[f"result{c}"], rtol=0, atol=1e-4)
    assert tuple(opt.elite.shape) == (int(opt.elite_num), H, A)


# every NFMAX instance of the sampler (horizons up to 16 / 32 / 48 / 64 / 96 / 128), even and odd horizons (the Nyquist term exists for even ones
# only), and the three ways a series' steps are dealt to threads (n x A >= 65 536: one thread per series; >= 32 768: two; below: four)
ICEM_SHAPES = [(3, 50, 2), (8, 700, 3), (15, 64, 17), (16, 33, 5), (17, 2000, 17), (31, 129, 6), (33, 40, 7), (40, 1036, 17), (47, 300, 2), (48, 17, 17),
               (64, 70000, 1), (65, 33, 3), (96, 40, 6), (97, 8200, 4), (127, 20, 2), (128, 300, 1)]


@pytest.mark.parametrize("H,n,A", ICEM_SHAPES, ids=[f"H{h}_n{n}_A{a}" for h, n, a in ICEM_SHAPES])
def test_icem_sampler_instances_match_reference_irfft(engine, H, n, A):
    """powerlaw_psd_gaussian (util/math.py:318-396) from injected spectrum normals against the oracle's torch.fft.irfft form, for every
    instance of the device sampler (coefficients in registers, frequency loop unrolled over NFMAX, padded frequencies dropped by a
    select) and every split of a series' time steps over threads; mean / std / bounds applied as trajectory_opt.py:433-441 does."""
    g = torch.Generator().manual_seed(H * 1000 + A)
    normals = torch.randn(2, n, A, H // 2 + 1, generator=g)
    ref = po.powerlaw_psd_gaussian(1.5, size=(n, A, H), normals=(normals[0], normals[1])).transpose(1, 2)
    mu = torch.randn(H, A, generator=g) * 0.3
    var = torch.rand(H, A, generator=g) + 0.1
    lo, hi = -torch.ones(H, A), torch.ones(H, A) * 0.8
    want = torch.maximum(torch.minimum(ref * var.sqrt() + mu, hi), lo)
    out = torch.full((n, H, A), float("nan"), device=DEV)
    engine.icem_sample(n, H, A, 1.5, mu.to(DEV), var.to(DEV), lo.to(DEV), hi.to(DEV), out, normals=normals.to(DEV).contiguous())
    assert torch.allclose(out.cpu(), want, rtol=1e-5, atol=2e-5)
    # in-kernel draws: every element written, deterministic in (seed, stream), different across streams
    a = torch.full((n, H, A), float("nan"), device=DEV)
    b = torch.empty_like(a)
    engine.icem_sample(n, H, A, 1.5, mu.to(DEV), var.to(DEV), lo.to(DEV), hi.to(DEV), a, seed=5, stream_id=2)
    engine.icem_sample(n, H, A, 1.5, mu.to(DEV), var.to(DEV), lo.to(DEV), hi.to(DEV), b, seed=5, stream_id=2)
    assert torch.isfinite(a).all() and torch.equal(a, b)
    engine.icem_sample(n, H, A, 1.5, mu.to(DEV), var.to(DEV), lo.to(DEV), hi.to(DEV), b, seed=5, stream_id=3)
    assert not torch.equal(a, b)


def test_icem_and_mppi_with_engine_objective(engine):
    """iCEM (cfg4-like: 7 members / 5 elites, Appendix B7) and MPPI driving the fused rollout objective."""
    from hipets.planning import _BoundObjective

    obs, act, H, P = 17, 6, 10, 20
    om = po.make_synthetic_model(obs, act, ensemble_size=7, hid=64, seed=5, elite=[0, 1, 2, 3, 4])
    om.max_logvar = torch.full_like(om.max_logvar, -8.0)
    fn = hipets.make_eval_fn(to_spec(om, obs, act), P, engine=engine, seed=3)
    lb, ub = [[-1.0] * act] * H, [[1.0] * act] * H
    s0 = np.zeros(obs, np.float32)
    icem = hipets.ICEMOptimizer(4, 0.1, 200, 1.3, 2.0, lb, ub, 0.3, 0.1, DEV, return_mean_elites=True, population_size_module=7, seed=1)
    mppi = hipets.MPPIOptimizer(4, 210, 0.9, 1.0, 0.9, lb, ub, DEV, seed=1)
    plans = [icem.optimize(_BoundObjective(fn, s0), x0=torch.zeros(H, act)), mppi.optimize(_BoundObjective(fn, s0))]
    plans.append(icem.optimize(_BoundObjective(fn, s0), x0=plans[0]))  # second plan uses the kept elites
    g = torch.Generator().manual_seed(0)
    cands = torch.cat([torch.stack([p_.cpu() for p_ in plans]), torch.rand(7, H, act, generator=g) * 2 - 1])
    B = 10 * 50
    perms = torch.stack([torch.randperm(B, generator=g) for _ in range(H)])
    eps = torch.randn(H, B, obs, generator=g)
    r = po.rollout(om, cands, s0, 50, perms=perms, eps=eps)
    assert (r[:3] > r[3:].max() + 0.5).all(), r


@pytest.mark.parametrize("persistent", [True, False])
@pytest.mark.parametrize("mode", ["device", "fast"])
def test_batched_rollout_replayed_through_oracle_per_environment(engine, mode, persistent):
    """n_env environments in one launch: candidates of environment g start from s0[g]; replayed per environment.  DEVICE mode (round 6):
    one balanced permutation per step over the rows of ALL environments, in the persistent form (rows start from their environment's
    s0 inside the kernel) and in per-step launches (init_state_kernel tiles the environments' s0)."""
    from test_gpu_batched_plans import batched_members

    if mode == "fast" and not persistent:
        pytest.skip("FAST mode has one launch form")
    obs, act, H, P, pop_env, n_env = 17, 6, 5, 5, 30, 4
    om = po.make_synthetic_model(obs, act, ensemble_size=5, hid=48, seed=8)
    engine.set_model(to_spec(om, obs, act))
    g = torch.Generator().manual_seed(2)
    actions = torch.rand(n_env * pop_env, H, act, generator=g) * 2 - 1
    s0 = (torch.randn(n_env, obs, generator=g) * 0.3).numpy().astype(np.float32)
    seed, sid = 5, 9
    engine.set_persistent(persistent)
    try:
        out = engine.rollout(actions.to(DEV), s0, P, mode=mode, seed=seed, stream_id=sid, n_env=n_env).cpu()
    finally:
        engine.set_persistent(True)
    pop = n_env * pop_env
    eps = engine.fast_normals(H, pop * P, seed, sid).cpu()
    members = batched_members(engine, om, pop, P, H, seed, sid, mode)
    for e_ in range(n_env):
        sl = slice(e_ * pop_env * P, (e_ + 1) * pop_env * P)
        ref = po.rollout(om, actions[e_ * pop_env:(e_ + 1) * pop_env], s0[e_], P, members=members[:, sl], eps=eps[:, sl])
        assert torch.allclose(out[e_ * pop_env:(e_ + 1) * pop_env], ref, rtol=0, atol=1e-4)


@pytest.mark.parametrize("mode", ["device", "fast"])
def test_batched_cem_planning(engine, mode):
    """hipets_plan_cem_batched: n_env = 1 is bit-identical to the single-environment plan; a batch gives every environment
    a good plan for ITS observation (scored by the oracle), and the warm start shifts per environment."""
    from hipets.planning import _BoundObjective

    obs, act, H, P, pop, n_env = 17, 6, 8, 10, 200, 3
    om = po.make_synthetic_model(obs, act, ensemble_size=5, hid=48, seed=6)
    om.max_logvar = torch.full_like(om.max_logvar, -8.0)
    spec = to_spec(om, obs, act)
    fn = hipets.make_eval_fn(spec, P, engine=engine, seed=3, mode=mode)
    lb, ub = [-1.0] * act, [1.0] * act
    single = hipets.CEMOptimizer(4, 0.1, pop, [lb] * H, [ub] * H, 0.1, DEV, return_mean_elites=True, seed=7)
    s0 = (np.random.default_rng(1).standard_normal((n_env, obs)) * 0.3).astype(np.float32)
    one = single.optimize(_BoundObjective(fn, s0[0]), x0=torch.zeros(H, act))
    agent1 = hipets.BatchedCEMAgent(fn, 1, lb, ub, H, 4, 0.1, pop, 0.1, seed=7)
    assert torch.equal(torch.from_numpy(agent1.plan(s0[:1]))[0], one.cpu())
    agent = hipets.BatchedCEMAgent(fn, n_env, lb, ub, H, 4, 0.1, pop, 0.1, seed=7)
    plans = agent.plan(s0)
    assert plans.shape == (n_env, H, act) and np.isfinite(plans).all()
    prev = agent.previous_solution.cpu().numpy()
    assert np.allclose(prev[:, :-1], plans[:, 1:]) and np.allclose(prev[:, -1], 0.0)
    acts = agent.act(s0)
    assert acts.shape == (n_env, act)
    g = torch.Generator().manual_seed(0)
    for e_ in range(n_env):
        cands = torch.cat([torch.from_numpy(plans[e_])[None], torch.rand(7, H, act, generator=g) * 2 - 1])
        B = 8 * 50
        perms = torch.stack([torch.randperm(B, generator=g) for _ in range(H)])
        eps = torch.randn(H, B, obs, generator=g)
        r = po.rollout(om, cands, s0[e_], 50, perms=perms, eps=eps)
        assert r[0] > r[1:].max() + 0.5, (e_, r)


def test_exact_device_mode_has_reference_semantics(engine):
    """mode='exact_device': a global balanced permutation per step drawn on the device.  Statistically identical to the
    oracle's reference-order sampling (same estimator), deterministic under a fixed seed."""
    obs, act, pop, P, H = 17, 6, 40, 20, 8
    om = po.make_synthetic_model(obs, act, ensemble_size=5, hid=48, seed=2)
    g = torch.Generator().manual_seed(1)
    actions = torch.rand(pop, H, act, generator=g) * 2 - 1
    s0 = np.zeros(obs, np.float32)
    spec = to_spec(om, obs, act)
    outs = []
    for rep in range(2):
        fn = hipets.make_eval_fn(spec, P, engine=engine, mode="exact_device", seed=4)
        outs.append(torch.stack([fn(s0, actions.to(DEV)) for _ in range(12)]).cpu())
    assert torch.equal(outs[0], outs[1])
    dev_runs = outs[0]
    torch.manual_seed(0)
    ref_runs = torch.stack([po.rollout(om, actions, s0, P, global_rng=True, generator=g) for _ in range(12)])
    se = torch.sqrt(dev_runs.var(0) / 12 + ref_runs.var(0) / 12)
    z = (dev_runs.mean(0) - ref_runs.mean(0)) / se
    assert z.abs().max() < 4.5 and abs(z.mean()) < 1.0


def _deterministic_objective(engine, obs, act, H, P, members=5, elite=None):
    """A model whose rollouts do not consume randomness (expectation propagation, deterministic head), so fused and
    per-iteration optimizer paths can be compared bit for bit."""
    from hipets.planning import _BoundObjective

    om = po.make_synthetic_model(obs, act, ensemble_size=members, hid=48, seed=9, deterministic=True, elite=elite)
    om.propagation = "expectation"
    fn = hipets.make_eval_fn(to_spec(om, obs, act), P, engine=engine, seed=0)
    return _BoundObjective(fn, (np.random.default_rng(3).standard_normal(obs) * 0.1).astype(np.float32))


MPPI_SAMPLE_SHAPES = [(2000, 50, 6), (350, 30, 6), (5, 3, 1), (1030, 7, 3), (4, 128, 100)]  # the last: H x A > 12 288, one thread per series


@pytest.mark.parametrize("pop,H,A", MPPI_SAMPLE_SHAPES, ids=[f"pop{c[0]}_H{c[1]}_A{c[2]}" for c in MPPI_SAMPLE_SHAPES])
def test_mppi_sample_shapes(engine, pop, H, A):
    """trajectory_opt.py:262-295 on its own with injected noise: the beta recurrence on unclipped values, clipping applied to what is stored --
    bitwise against the same f32 operations in torch (whole candidates staged in LDS, optim.hpp round 6); and the Philox draws of a candidate
    do not depend on how many candidates share a workgroup."""
    g = torch.Generator().manual_seed(pop + H)
    z = torch.randn(pop, H, A, generator=g).clamp(-2, 2)
    mean, past = torch.randn(H, A, generator=g) * 0.3, torch.randn(A, generator=g) * 0.3
    lower, upper = -torch.ones(H, A) * 0.8, torch.ones(H, A) * 0.9
    beta = 0.7
    out = torch.empty(pop, H, A, device=DEV)
    engine.mppi_sample(pop, H, A, beta, mean.to(DEV), past.to(DEV), lower.to(DEV), upper.to(DEV), out, z=z.to(DEV).contiguous())
    b, omb = torch.tensor(beta, dtype=torch.float32), torch.tensor(1.0, dtype=torch.float32) - torch.tensor(beta, dtype=torch.float32)
    prev, want = past.expand(pop, A).clone(), torch.empty(pop, H, A)
    for h in range(H):
        x = b * (mean[h] + z[:, h]) + omb * prev
        prev = x
        want[:, h] = torch.maximum(torch.minimum(x, upper[h]), lower[h])
    assert torch.equal(out.cpu(), want)
    if pop >= 1030:  # G = pop // 512 candidates per workgroup here, one per workgroup in the smaller call: the same series
        few = torch.empty(300, H, A, device=DEV)
        engine.mppi_sample(pop, H, A, beta, mean.to(DEV), past.to(DEV), lower.to(DEV), upper.to(DEV), out, seed=11, stream_id=5)
        engine.mppi_sample(300, H, A, beta, mean.to(DEV), past.to(DEV), lower.to(DEV), upper.to(DEV), few, seed=11, stream_id=5)
        assert torch.equal(out[:300].cpu(), few.cpu()) and out.abs().max().item() <= 0.9


# (pop, H, A): populations on both sides of a 256-candidate tile and of a 16-row load round, dimension counts below / at / above one wave of 64 and
# above 32 workgroups' worth (a workgroup then walks several 64-dimension slices), the largest population the ABI accepts
MPPI_UPDATE_SHAPES = [(2000, 50, 6), (350, 30, 6), (7, 3, 1), (17, 5, 13), (256, 8, 8), (257, 1, 64), (4097, 9, 7), (12000, 4, 3), (300, 64, 40)]


@pytest.mark.parametrize("pop,H,A", MPPI_UPDATE_SHAPES, ids=[f"pop{c[0]}_D{c[1] * c[2]}" for c in MPPI_UPDATE_SHAPES])
def test_mppi_update_shapes(engine, pop, H, A):
    """trajectory_opt.py:296-309 on its own: NaN -> -1e-10 in place, w = exp(gamma (v - max v)), mean = sum(w x) / (sum w + 1e-10).  The device
    sum of a dimension is one f32 chain over the candidates, its population staged through LDS tiles (optim.hpp, round 6): against f64."""
    g = torch.Generator().manual_seed(pop * 31 + H)
    values = torch.randn(pop, generator=g) * 3
    values[pop // 2] = float("nan")
    population = torch.randn(pop, H, A, generator=g)
    mean = torch.full((H, A), 7.0, device=DEV)
    vals_dev = values.to(DEV)
    engine.mppi_update(pop, H, A, 0.9, vals_dev, population.to(DEV).contiguous(), mean)
    v = values.clone()
    v[v.isnan()] = -1e-10
    assert torch.equal(vals_dev.cpu(), v)
    w = torch.exp(0.9 * (v.double() - v.double().max()))
    want = (w[:, None, None] * population.double()).sum(0) / (w.sum() + 1e-10)
    assert torch.allclose(mean.cpu().double(), want, rtol=2e-5, atol=2e-6)


def test_fused_mppi_plan_equals_per_iteration_path(engine):
    """hipets_plan_mppi is the MPPIOptimizer.optimize loop (trajectory_opt.py:238-311) enqueued by the library: same
    kernels, same Philox streams => bitwise the same means over consecutive calls (persistent, shifted mean)."""
    obs, act, H, P, pop = 17, 6, 9, 5, 120
    obj = _deterministic_objective(engine, obs, act, H, P)
    lb, ub = [[-1.0] * act] * H, [[1.0] * act] * H
    a = hipets.MPPIOptimizer(3, pop, 0.9, 1.0, 0.9, lb, ub, DEV, seed=21)
    b = hipets.MPPIOptimizer(3, pop, 0.9, 1.0, 0.9, lb, ub, DEV, seed=21)
    for _ in range(3):
        fused = a.optimize(obj)
        generic = b.optimize(obj, force_generic=True)
        assert torch.equal(fused, generic)
        assert torch.equal(a.mean, b.mean)
    assert fused.abs().max() > 0 and torch.isfinite(fused).all()


@pytest.mark.parametrize("return_mean", [True, False])
def test_fused_icem_plan_equals_per_iteration_path(engine, return_mean):
    """hipets_plan_icem vs the per-iteration ICEMOptimizer path with the kept-elite draws injected into both: bitwise
    equal plans and persistent elites over two calls (first call has no elites, second shifts them, :450-462)."""
    obs, act, H, P = 17, 6, 8, 5
    obj = _deterministic_objective(engine, obs, act, H, P, members=7, elite=[0, 1, 2, 3, 4])
    lb, ub = [[-1.0] * act] * H, [[1.0] * act] * H
    mk = lambda: hipets.ICEMOptimizer(4, 0.1, 150, 1.3, 2.0, lb, ub, 0.3, 0.1, DEV, return_mean_elites=return_mean,  # noqa: E731
                                      population_size_module=5, seed=17)
    a, b = mk(), mk()
    K, keep = int(a.elite_num), int(a.keep_elite_size)
    g = torch.Generator().manual_seed(0)
    x0 = torch.zeros(H, act)
    for call in range(2):
        perms = [torch.randperm(K, generator=g) for _ in range(4)]
        keep_idx = torch.stack([p_[:keep] for p_ in perms]).to(torch.int32).to(DEV).contiguous()
        fused = a.optimize(obj, x0=x0, keep_idx=keep_idx)
        generic = b.optimize(obj, x0=x0, inject=[{"keep_perm": p_} for p_ in perms])
        assert torch.equal(fused, generic), call
        assert torch.equal(a.elite, b.elite), call
        x0 = fused.cpu()
    assert torch.isfinite(fused).all() and (fused.abs() <= 1).all()


def test_fused_icem_draws_its_own_kept_elites(engine):
    """Without injected keep indices the library draws randperm(elite_num)[:keep] itself: distinct, in range."""
    from hipets._lib import IcemParams  # noqa: F401  (struct is part of the binding)

    obs, act, H, P = 17, 6, 8, 5
    obj = _deterministic_objective(engine, obs, act, H, P)
    lb, ub = [[-1.0] * act] * H, [[1.0] * act] * H
    opt = hipets.ICEMOptimizer(3, 0.1, 100, 1.3, 2.0, lb, ub, 0.5, 0.1, DEV, return_mean_elites=True, population_size_module=5, seed=4)
    p1 = opt.optimize(obj, x0=torch.zeros(H, act))
    p2 = opt.optimize(obj, x0=p1)
    assert torch.isfinite(p2).all() and tuple(opt.elite.shape) == (int(opt.elite_num), H, act)
    g = torch.Generator().manual_seed(1)
    rand = (torch.rand(8, H, act, generator=g) * 2 - 1).to(DEV)
    vals = obj(torch.cat([p2[None], rand]).contiguous())
    assert vals[0] > vals[1:].median()  # the refined plan beats typical random plans under the same deterministic model


def test_full_plan_reference_order_same_elites_and_action(engine):
    """North-star parity at plan level: a whole CEM plan with the model rollout as objective, every random draw injected
    (truncated normals of the sampler, per-step randperms and eps of the rollouts: what the reference consumes under
    fixed seeds).  The engine must select the SAME elites at every iteration (T3, up to value ties within T2) and return
    the same plan / first action as the oracle (T4)."""
    obs, act, H, P, pop, iters = 17, 6, 8, 5, 60, 4
    om = po.make_synthetic_model(obs, act, ensemble_size=5, hid=48, seed=12, no_delta_list=[0])
    engine.set_model(to_spec(om, obs, act))
    g = torch.Generator().manual_seed(4)
    s0 = (np.random.default_rng(2).standard_normal(obs) * 0.1).astype(np.float32)
    B = pop * P
    z = [po.truncated_normal_(torch.zeros(pop, H, act), generator=g) for _ in range(iters)]
    perms = [torch.stack([torch.randperm(B, generator=g) for _ in range(H)]) for _ in range(iters)]
    eps = [torch.randn(H, B, obs, generator=g) for _ in range(iters)]
    lower, upper = -torch.ones(H, act), torch.ones(H, act)

    calls = {"n": 0}

    def oracle_obj(population):
        i = calls["n"]; calls["n"] += 1
        return po.rollout(om, population, s0, P, perms=perms[i], eps=eps[i])

    rec = []
    ref = po.cem_optimize(oracle_obj, torch.zeros(H, act), lower, upper, iters, 0.1, pop, 0.1, return_mean_elites=True, noise=z, record=rec)

    dcalls = {"n": 0}

    def engine_obj(population):
        i = dcalls["n"]; dcalls["n"] += 1
        return engine.rollout(population.contiguous(), s0, P, mode="exact", perms=perms[i].to(DEV), eps=eps[i].to(DEV))

    seen = []
    opt = hipets.CEMOptimizer(iters, 0.1, pop, lower.tolist(), upper.tolist(), 0.1, DEV, return_mean_elites=True, seed=0)
    out = opt.optimize(engine_obj, x0=torch.zeros(H, act), noise=z, callback=lambda p_, v_, i_: seen.append((p_.clone(), v_.clone())))
    K = int(opt.elite_num)
    for i in range(iters):
        population, values = seen[i]
        assert torch.allclose(population.cpu(), rec[i]["population"], rtol=0, atol=1e-5)
        v_ref = rec[i]["values"]
        assert ((values.cpu() - v_ref).abs() <= 1e-4 * torch.clamp(v_ref.abs(), min=1.0)).all()  # T2
        mine = set(values.cpu().topk(K).indices.tolist())
        theirs = set(rec[i]["elite_idx"].tolist())
        kth = v_ref.topk(K + 1).values
        if (kth[K - 1] - kth[K]).abs() > 2e-4 * max(1.0, float(kth[K - 1].abs())):  # no tie at the elite boundary
            assert mine == theirs, i  # T3
    assert torch.allclose(out.cpu(), ref, rtol=0, atol=1e-4)  # T4: same plan, hence the same first action


@pytest.mark.parametrize("case", ["cem_two_steps", "mppi_two_steps", "icem_two_steps"])
def test_agent_act_reproduces_the_reference_agent_under_fixed_seeds(case):
    """North star, end to end: `TrajectoryOptimizerAgent.act` with the same torch seeds as the unmodified reference agent
    (golden recorded from mbrl.planning.TrajectoryOptimizerAgent + mbrl.models.ModelEnv on CPU, oracle/make_golden.py)
    selects the same actions.  sampler='torch' + mode='exact' consume torch's generators in the reference's order:
    truncated-normal population noise and per-step randperms from the global generator, eps from ModelEnv's."""
    om, meta, a = load_case(os.path.join(GOLDEN, f"agent_{case}.npz"))
    obs, act, H, P = meta["obs_dim"], meta["act_dim"], meta["H"], meta["P"]
    fn = hipets.make_eval_fn(to_spec(om, obs, act), P, mode="exact", rng=torch.Generator().manual_seed(meta["generator_seed"]))
    if meta.get("optimizer", "cem") == "icem":
        cfg = dict(_target_="hipets.ICEMOptimizer", num_iterations=meta["iters"], elite_ratio=0.1, population_size=meta["pop"],
                   population_decay_factor=1.3, colored_noise_exponent=2.0, keep_elite_frac=0.3, alpha=0.1, device=DEV,
                   lower_bound="???", upper_bound="???", return_mean_elites=True, population_size_module=5, sampler="torch")
    elif meta.get("optimizer", "cem") == "mppi":
        cfg = dict(_target_="hipets.MPPIOptimizer", num_iterations=meta["iters"], population_size=meta["pop"], gamma=0.9, sigma=1.0,
                   beta=0.9, device=DEV, lower_bound="???", upper_bound="???", sampler="torch")
    else:
        cfg = dict(_target_="hipets.CEMOptimizer", num_iterations=meta["iters"], elite_ratio=0.1, population_size=meta["pop"], alpha=0.1,
                   device=DEV, lower_bound="???", upper_bound="???", return_mean_elites=True, clipped_normal=False, sampler="torch")
    agent = hipets.TrajectoryOptimizerAgent(cfg, [-1.0] * act, [1.0] * act, planning_horizon=H, replan_freq=1)
    agent.set_trajectory_eval_fn(fn)
    torch.manual_seed(meta["torch_seed"])
    for t in range(a["observations"].shape[0]):
        action = agent.act(a["observations"][t].numpy())
        assert np.allclose(action, a["actions"][t].numpy(), rtol=0, atol=1e-4), t
        assert np.allclose(agent.optimizer.previous_solution.cpu().numpy(), a["shifted_plans"][t].numpy(), rtol=0, atol=1e-4)


# (H, A): the plan's H x A dimensions are dealt to up to 8 workgroups in slices of <= 128 (cem.hpp, round 6), eight threads per dimension, a
# thread's share of the K elites held in registers (2 / 4 / 8 / 16 per thread) or walked 16 at a time (K > 128): every tier, one and several
# workgroups, populations on both sides of the rank-counting / radix-select switch
REFIT_SHAPES = [(4000, 400, False, 5, 3), (2000, 200, True, 5, 3), (8192, 820, False, 5, 3), (1500, 1, True, 5, 3), (3000, 1500, False, 5, 3),
                (500, 50, False, 30, 6), (500, 12, True, 30, 6), (1036, 103, False, 40, 17), (1036, 30, True, 40, 17), (600, 300, False, 40, 17),
                (350, 35, False, 25, 7), (100, 10, True, 15, 1), (640, 64, True, 50, 24), (641, 7, False, 50, 24),
                # rank counting four keys to an LDS read with wave-uniform bounds (round 6): populations around the 64-element wave blocks, ties on
                (130, 13, True, 5, 3), (65, 6, True, 5, 3), (257, 25, True, 5, 3), (63, 6, True, 5, 3), (3, 2, False, 2, 2), (2, 1, False, 2, 2)]


@pytest.mark.parametrize("pop,K,ties,H,A", REFIT_SHAPES, ids=[f"pop{c[0]}_K{c[1]}_{'ties_' if c[2] else ''}D{c[3] * c[4]}" for c in REFIT_SHAPES])
def test_refit_large_populations_select_path(engine, pop, K, ties, H, A):
    """Populations of sharded multi-GPU plans (pop 500 x 8 ranks) and beyond: the top-k is a radix select + a sort of the
    elites (a full bitonic sort when K > 1024).  Elite indices in order (ties -> lower index first, NaN -> -1e-10 first),
    refit statistics and best-so-far against torch / f64."""
    g = torch.Generator().manual_seed(pop)
    values = torch.randn(pop, generator=g)
    if ties:
        values = (values * 4).round() / 4  # many exact duplicates, also across the elite threshold
        values[5] = float("nan"); values[min(77, pop - 1)] = float("nan")
    population = torch.randn(pop, H, A, generator=g)
    p = hipets.Engine.cem_params(pop, H, A, 1, K, 0.1, True, False, True)
    mu, disp = torch.zeros(H, A, device=DEV), torch.ones(H, A, device=DEV)
    best_v, best_s = torch.full((1,), -float("inf"), device=DEV), torch.zeros(H, A, device=DEV)
    eidx = torch.empty(K, dtype=torch.int32, device=DEV)
    vals_dev = values.to(DEV)
    engine.cem_refit(p, vals_dev, population.to(DEV).contiguous(), mu, disp, best_v, best_s, eidx)
    v = values.clone()
    v[v.isnan()] = -1e-10
    order = sorted(range(pop), key=lambda i: (-float(v[i]), i))[:K]  # value descending, index ascending
    assert eidx.cpu().tolist() == order
    elite = population[order].double()
    new_mu = elite.mean(0)
    assert torch.allclose(mu.cpu().double(), 0.9 * new_mu, atol=1e-6)
    if K > 1:
        assert torch.allclose(disp.cpu().double(), 0.1 + 0.9 * elite.var(0, unbiased=True), atol=1e-5)
    assert float(best_v) == float(v[order[0]]) and torch.equal(best_s.cpu(), population[order[0]])
    assert torch.equal(vals_dev.cpu(), v)  # NaNs replaced in place like the reference
