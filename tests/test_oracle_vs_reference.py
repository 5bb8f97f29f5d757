"""Pins the oracle against the UNMODIFIED reference (imported from /root/reference through
oracle/ref_stubs).  Runs only in the build container; skipped wherever the reference is absent
(e.g. the GPU box), where tests/golden/*.npz take over."""
import numpy as np
import pytest
import torch

from oracle import pets_oracle as po
from oracle.ref_bridge import build_reference_model_env, import_reference, reference_available

pytestmark = pytest.mark.skipif(not reference_available(), reason="reference not mounted")


@pytest.fixture(scope="module")
def mbrl():
    return import_reference()


CASES = [
    dict(obs=17, act=6, mkw=dict(ensemble_size=5, hid=40, seed=0, no_delta_list=[0, 3]), pop=20, P=5, H=4),
    dict(obs=4, act=1, mkw=dict(ensemble_size=5, hid=32, seed=1, reward="cartpole", termination="cartpole"), pop=15, P=5, H=10),
    dict(obs=8, act=2, mkw=dict(ensemble_size=7, hid=16, seed=2, elite=[1, 2, 4, 5, 6], propagation="fixed_model",
                                termination="walker2d"), pop=10, P=5, H=5),
    dict(obs=6, act=2, mkw=dict(ensemble_size=3, hid=16, seed=3, propagation="expectation", normalizer="f32",
                                termination="ant", activation="sigmoid"), pop=6, P=2, H=5),
    dict(obs=5, act=2, mkw=dict(ensemble_size=2, hid=16, seed=4, termination="inverted_pendulum", reward="inverted_pendulum",
                                normalizer="none"), pop=8, P=3, H=6),
    dict(obs=20, act=7, mkw=dict(ensemble_size=2, hid=16, seed=5, reward="pusher", activation="leaky_relu"), pop=4, P=2, H=3),
    # BasicEnsemble of single-member GaussianMLPs: randint member maps from the generator, any batch size
    dict(obs=9, act=3, mkw=dict(ensemble_size=5, hid=32, seed=6, ensemble_kind="basic_ensemble", propagation="fixed_model"),
         pop=13, P=3, H=10),
    dict(obs=17, act=6, mkw=dict(ensemble_size=3, hid=24, seed=7, ensemble_kind="basic_ensemble"), pop=11, P=4, H=5),
    dict(obs=8, act=2, mkw=dict(ensemble_size=3, hid=16, seed=8, ensemble_kind="basic_ensemble", propagation="expectation"),
         pop=7, P=2, H=4),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: f"obs{c['obs']}_{c['mkw'].get('ensemble_kind', 'gmlp')}_{c['mkw'].get('propagation', 'random_model')}")
def test_rollout_bitwise_vs_reference(mbrl, case):
    om = po.make_synthetic_model(case["obs"], case["act"], **case["mkw"])
    g = torch.Generator().manual_seed(9)
    actions = torch.rand(case["pop"], case["H"], case["act"], generator=g) * 2 - 1
    s0 = (np.random.default_rng(3).standard_normal(case["obs"]) * 0.1).astype(np.float32)
    if om.termination == "walker2d":
        s0[0] = 1.0
    if om.termination == "ant":
        s0[0] = 0.6
    me, _, _ = build_reference_model_env(om, case["obs"], case["act"], generator=torch.Generator().manual_seed(1))
    torch.manual_seed(42)
    ref = me.evaluate_action_sequences(actions, s0, case["P"])
    torch.manual_seed(42)
    mine = po.rollout(om, actions, s0, case["P"], global_rng=True, generator=torch.Generator().manual_seed(1))
    assert torch.equal(ref, mine)


@pytest.mark.parametrize("propagation", ["random_model", "fixed_model", "expectation"])
def test_basic_ensemble_per_member_logvar_bounds_bitwise(mbrl, propagation):
    """Every BasicEnsemble member is its own GaussianMLP with its own (learned) logvar bounds."""
    obs, act, E, pop, P, H = 7, 2, 4, 9, 3, 5
    om = po.make_synthetic_model(obs, act, ensemble_size=E, hid=16, seed=31, ensemble_kind="basic_ensemble", propagation=propagation)
    g = torch.Generator().manual_seed(2)
    om.min_logvar = -10 + torch.rand(E, obs, generator=g) * 6
    om.max_logvar = -3 + torch.rand(E, obs, generator=g) * 3
    actions = torch.rand(pop, H, act, generator=g) * 2 - 1
    s0 = (np.random.default_rng(4).standard_normal(obs) * 0.1).astype(np.float32)
    me, _, _ = build_reference_model_env(om, obs, act, generator=torch.Generator().manual_seed(1))
    ref = me.evaluate_action_sequences(actions, s0, P)
    mine = po.rollout(om, actions, s0, P, generator=torch.Generator().manual_seed(1))
    assert torch.equal(ref, mine)


def test_member_map_equals_perm(mbrl):
    """The oracle's explicit row->member map (used to check FAST mode) == the reference's perm form."""
    om = po.make_synthetic_model(17, 6, ensemble_size=5, hid=24, seed=0)
    B, H, P = 50, 3, 5
    g = torch.Generator().manual_seed(0)
    actions = torch.rand(B // P, H, 6, generator=g)
    s0 = np.zeros(17, np.float32)
    perms = torch.stack([torch.randperm(B, generator=g) for _ in range(H)])
    eps = torch.randn(H, B, 17, generator=g)
    members = torch.empty(H, B, dtype=torch.long)
    for t in range(H):
        members[t][perms[t]] = torch.arange(B) // (B // 5)
    a = po.rollout(om, actions, s0, P, perms=perms, eps=eps)
    b = po.rollout(om, actions, s0, P, members=members, eps=eps)
    assert torch.allclose(a, b, rtol=0, atol=1e-6)


def _quadratic(target, nan_at=None):
    def f(x):
        v = -((x - target) ** 2).sum(dim=(1, 2)).clone()
        if nan_at is not None:
            v[nan_at] = float("nan")
        return v

    return f


@pytest.mark.parametrize("clipped,return_mean", [(False, True), (False, False), (True, True)])
def test_cem_bitwise_vs_reference(mbrl, clipped, return_mean):
    H, A, pop = 5, 2, 30
    lb, ub = [[-1.0, -2.0]] * H, [[1.0, 0.5]] * H
    obj = _quadratic(torch.linspace(-0.3, 0.3, H * A).view(H, A), nan_at=2)
    opt = mbrl.planning.CEMOptimizer(3, 0.2, pop, lb, ub, 0.1, "cpu", return_mean_elites=return_mean, clipped_normal=clipped)
    torch.manual_seed(0)
    ref = opt.optimize(obj, x0=torch.zeros(H, A))
    torch.manual_seed(0)
    mine = po.cem_optimize(obj, torch.zeros(H, A), torch.tensor(lb), torch.tensor(ub), 3, 0.2, pop, 0.1,
                           return_mean_elites=return_mean, clipped_normal=clipped)
    assert torch.equal(ref, mine)


def test_mppi_bitwise_vs_reference_two_calls(mbrl):
    """Two consecutive plans: pins the persistent mean and the past_action aliasing quirk (Appendix B4/B6)."""
    H, A, pop = 6, 2, 40
    lb, ub = [[-1.0, -1.0]] * H, [[1.0, 1.0]] * H
    obj = _quadratic(torch.full((H, A), 0.2))
    opt = mbrl.planning.MPPIOptimizer(3, pop, 0.9, 1.0, 0.9, lb, ub, "cpu")
    st = po.MPPIState(H, A)
    torch.manual_seed(1)
    r1, r2 = opt.optimize(obj), opt.optimize(obj)
    torch.manual_seed(1)
    m1 = po.mppi_optimize(obj, st, torch.tensor(lb), torch.tensor(ub), 3, pop, 0.9, 1.0, 0.9)
    m2 = po.mppi_optimize(obj, st, torch.tensor(lb), torch.tensor(ub), 3, pop, 0.9, 1.0, 0.9)
    assert torch.equal(r1, m1) and torch.equal(r2, m2)


def test_mppi_sigma_is_dead(mbrl):
    """Appendix B5: sigma does not influence the sampled population."""
    H, A, pop = 4, 2, 16
    lb, ub = torch.full((H, A), -1.0), torch.full((H, A), 1.0)
    obj = _quadratic(torch.zeros(H, A))
    outs = []
    for sigma in (0.01, 5.0):
        torch.manual_seed(3)
        outs.append(po.mppi_optimize(obj, po.MPPIState(H, A), lb, ub, 2, pop, 0.9, sigma, 0.9))
    assert torch.equal(outs[0], outs[1])


def test_icem_bitwise_vs_reference_two_calls(mbrl):
    """Two plans: pins coloured noise, population decay, kept-elite shift / append, biased variance."""
    H, A, pop = 8, 3, 60
    lb, ub = [[-1.0] * A] * H, [[1.0] * A] * H
    obj = _quadratic(torch.full((H, A), -0.1), nan_at=1)
    kw = dict(num_iterations=4, elite_ratio=0.1, population_size=pop, population_decay_factor=1.3,
              colored_noise_exponent=2.0, keep_elite_frac=0.3, alpha=0.1)
    opt = mbrl.planning.ICEMOptimizer(lower_bound=lb, upper_bound=ub, device="cpu", return_mean_elites=True,
                                      population_size_module=5, **kw)
    st = po.ICEMState()
    torch.manual_seed(2)
    r1 = opt.optimize(obj, x0=torch.zeros(H, A))
    r2 = opt.optimize(obj, x0=r1.clone())
    torch.manual_seed(2)
    m1 = po.icem_optimize(obj, st, torch.zeros(H, A), torch.tensor(lb), torch.tensor(ub), return_mean_elites=True,
                          population_size_module=5, **kw)
    m2 = po.icem_optimize(obj, st, m1.clone(), torch.tensor(lb), torch.tensor(ub), return_mean_elites=True,
                          population_size_module=5, **kw)
    assert torch.equal(r1, m1) and torch.equal(r2, m2)


@pytest.mark.parametrize("n", [15, 30, 40])
def test_powerlaw_noise_bitwise(mbrl, n):
    import mbrl.util.math as rm

    torch.manual_seed(0)
    ref = rm.powerlaw_psd_gaussian(2.0, size=(7, 3, n), device="cpu")
    torch.manual_seed(0)
    mine = po.powerlaw_psd_gaussian(2.0, size=(7, 3, n))
    assert torch.equal(ref, mine)


def test_truncated_normal_bitwise(mbrl):
    import mbrl.util.math as rm

    torch.manual_seed(0)
    ref = rm.truncated_normal_(torch.empty(50, 7))
    torch.manual_seed(0)
    mine = po.truncated_normal_(torch.empty(50, 7))
    assert torch.equal(ref, mine)


def test_trajectory_optimizer_shift(mbrl):
    """trajectory_opt.py:563-568: warm start rolled by replan_freq and refilled with (lb+ub)/2."""
    import omegaconf

    H, A = 5, 2
    lbv, ubv = np.array([-1.0, 0.0]), np.array([1.0, 2.0])
    cfg = omegaconf.OmegaConf.create(dict(_target_="mbrl.planning.CEMOptimizer", num_iterations=2, elite_ratio=0.2,
                                          population_size=20, alpha=0.1, device="cpu", lower_bound="???", upper_bound="???",
                                          return_mean_elites=True))
    ref = mbrl.planning.TrajectoryOptimizer(cfg, lbv, ubv, H, replan_freq=2)
    st = po.TrajectoryOptimizerState(lbv, ubv, H, replan_freq=2)
    obj = _quadratic(torch.full((H, A), 0.3))
    torch.manual_seed(0)
    r = [ref.optimize(obj), ref.optimize(obj)]
    torch.manual_seed(0)
    opt = lambda x0: po.cem_optimize(obj, x0, st.lower, st.upper, 2, 0.2, 20, 0.1, return_mean_elites=True)  # noqa: E731
    m = [st.step(opt), st.step(opt)]
    assert np.array_equal(r[0], m[0]) and np.array_equal(r[1], m[1])
    assert torch.equal(ref.previous_solution, st.previous_solution)


def test_spec_extraction_from_live_reference_objects(mbrl):
    """hipets.spec_from_model_env reads the real mbrl objects (seam 3) and reproduces the OracleModel."""
    import hipets

    om = po.make_synthetic_model(18, 6, ensemble_size=7, hid=16, seed=0, elite=[0, 1, 3, 4, 6], obs_process="halfcheetah",
                                 no_delta_list=[2], termination="hopper")
    me, dm, model = build_reference_model_env(om, 18, 6, generator=torch.Generator())
    spec = hipets.spec_from_model_env(me)
    assert spec.members == [0, 1, 3, 4, 6] and spec.activation == "silu" and spec.propagation == "random_model"
    assert spec.obs_process == "halfcheetah" and spec.reward == "halfcheetah" and spec.termination == "hopper"
    assert spec.no_delta_list == [2] and spec.norm_mean.dtype == torch.float64 and spec.in_dim == 24 and spec.out_dim == 18
    for a, b in zip(spec.weights, om.weights):
        assert torch.equal(a, b)
    v0 = hipets.model_version(me)
    with torch.no_grad():
        model.hidden_layers[0][0].weight.add_(1.0)
    assert hipets.model_version(me) != v0
    model.set_elite([0, 1, 2, 3, 4])
    assert hipets.spec_from_model_env(me).members == [0, 1, 2, 3, 4]


def test_spec_extraction_from_live_basic_ensemble(mbrl):
    """The second model config of the reference's tests (conf/dynamics_model/basic_ensemble.yaml,
    tests/algorithms/test_algorithms.py:197-198): BasicEnsemble of single-member GaussianMLPs -> one stacked spec."""
    import hipets

    om = po.make_synthetic_model(8, 2, ensemble_size=4, hid=16, seed=3, ensemble_kind="basic_ensemble", propagation="fixed_model")
    g = torch.Generator().manual_seed(0)
    om.min_logvar, om.max_logvar = -10 + torch.rand(4, 8, generator=g), torch.rand(4, 8, generator=g)
    me, dm, model = build_reference_model_env(om, 8, 2, generator=torch.Generator())
    assert type(model).__name__ == "BasicEnsemble"
    spec = hipets.spec_from_model_env(me)
    assert spec.ensemble_kind == "basic_ensemble" and spec.members == [0, 1, 2, 3] and spec.propagation == "fixed_model"
    for a, b in zip(spec.weights, om.weights):
        assert torch.equal(a, b)
    assert torch.equal(spec.min_logvar, om.min_logvar) and torch.equal(spec.max_logvar, om.max_logvar)
    v0 = hipets.model_version(me)
    with torch.no_grad():
        model.members[2].mean_and_logvar.weight.add_(1.0)
    assert hipets.model_version(me) != v0


def test_spec_from_checkpoint_written_by_the_reference(mbrl, tmp_path):
    """A checkpoint saved by the real OneDTransitionRewardModel.save loads into the same ModelSpec as the live objects."""
    import hipets

    om = po.make_synthetic_model(17, 6, ensemble_size=7, hid=16, seed=0, elite=[0, 1, 3, 4, 6])
    me, dm, model = build_reference_model_env(om, 17, 6, generator=torch.Generator())
    dm.save(str(tmp_path))
    live = hipets.spec_from_model_env(me)
    disk = hipets.spec_from_checkpoint(tmp_path, 17, 6)
    assert disk.members == live.members and disk.in_dim == live.in_dim
    for a, b in zip(disk.weights + disk.biases, live.weights + live.biases):
        assert torch.equal(a, b)
    assert torch.equal(disk.norm_mean, live.norm_mean) and disk.norm_mean.dtype == torch.float64
    assert torch.equal(disk.min_logvar, live.min_logvar)


@pytest.mark.parametrize("sample", [False, True])
def test_oracle_step_bitwise_vs_reference_model_env_step(mbrl, sample):
    """ModelEnv.reset/step (model_env.py:62-140), the MBPO-style one-transition path."""
    om = po.make_synthetic_model(11, 3, ensemble_size=5, hid=24, seed=7, termination="hopper", no_delta_list=[1])
    me, _, _ = build_reference_model_env(om, 11, 3, generator=torch.Generator().manual_seed(2))
    g = torch.Generator().manual_seed(0)
    obs0 = (torch.randn(40, 11, generator=g) * 0.3).numpy()
    obs0[:, 0] += 1.0
    act = torch.rand(40, 3, generator=g) * 2 - 1
    torch.manual_seed(5)
    state = me.reset(obs0, return_as_np=False)
    n1, r1, d1, state = me.step(act, state, sample=sample)
    n2, r2, d2, _ = me.step(act, state, sample=sample)
    torch.manual_seed(5)
    gen = torch.Generator().manual_seed(2)
    x = torch.from_numpy(obs0.astype(np.float32))
    outs = []
    for _ in range(2):
        perm = torch.randperm(40)
        eps = torch.empty(40, 11).normal_(0, 1, generator=gen) if sample else None
        x, r, d = po.step(om, x, act, perm=perm, eps=eps, sample=sample)
        outs.append((x, r, d))
    assert torch.equal(n1, outs[0][0]) and torch.equal(r1, outs[0][1]) and torch.equal(d1, outs[0][2])
    assert torch.equal(n2, outs[1][0]) and torch.equal(r2, outs[1][1]) and torch.equal(d2, outs[1][2])


# ---- PlaNet latent planning path -------------------------------------------------------------
@pytest.mark.parametrize("P", [1, 3])
def test_planet_rollout_bitwise_vs_reference(mbrl, P):
    from oracle import planet_oracle as pl
    from oracle.ref_bridge import build_reference_planet_env

    pm = pl.make_synthetic_planet(latent=10, action=3, belief=24, hidden=20, seed=2)
    g = torch.Generator().manual_seed(0)
    latent0, belief0 = torch.randn(1, 10, generator=g) * 0.3, torch.randn(1, 24, generator=g) * 0.3
    pop, H = 9, 6
    actions = torch.rand(pop, H, 3, generator=g) * 2 - 1
    me, model = build_reference_planet_env(pm, latent0, belief0, generator=torch.Generator().manual_seed(5))
    obs = np.zeros((3, 16, 16), np.float32)  # only its batch dimension is used (planet.py:669-672)
    ref = me.evaluate_action_sequences(actions, obs, P)
    mine = pl.planet_rollout(pm, actions, latent0, belief0, P, generator=torch.Generator().manual_seed(5))
    assert torch.equal(ref, mine)
    # one step, against PlaNetModel.sample directly (deterministic and sampled)
    B = 7
    lat, bel, act = torch.randn(B, 10, generator=g), torch.randn(B, 24, generator=g), torch.rand(B, 3, generator=g)
    for det in (True, False):
        r_lat, r_rew, _, r_state = model.sample(act, {"latent": lat, "belief": bel}, deterministic=det,
                                                rng=torch.Generator().manual_seed(8))
        o_lat, o_rew, o_bel = pl.planet_step(pm, lat, bel, act, generator=torch.Generator().manual_seed(8), deterministic=det)
        assert torch.equal(r_lat, o_lat) and torch.equal(r_rew, o_rew) and torch.equal(r_state["belief"], o_bel)


def test_planet_spec_extraction_from_live_reference_model(mbrl):
    """hipets.spec_from_planet_model reads the live PlaNetModel's planning heads; the freshness token follows updates."""
    import hipets
    from hipets.model import planet_version
    from oracle import planet_oracle as pl
    from oracle.ref_bridge import build_reference_planet_env

    pm = pl.make_synthetic_planet(latent=10, action=3, belief=24, hidden=20, seed=6)
    me, model = build_reference_planet_env(pm, torch.zeros(1, 10), torch.zeros(1, 24))
    for source in (model, me):  # the model itself or the ModelEnv wrapping it
        spec = hipets.spec_from_planet_model(getattr(source, "dynamics_model", source))
        assert (spec.latent_size, spec.action_size, spec.belief_size, spec.hidden_size) == (10, 3, 24, 20)
        for n in pl.PLANET_TENSORS:
            assert torch.equal(getattr(spec, n), getattr(pm, n)), n
        assert spec.min_std == pytest.approx(pm.min_std)
    assert spec.flops_per_candidate_step() == pm.flops_per_candidate_step()
    v0 = planet_version(model)
    with torch.no_grad():
        model.reward_model[2].weight.mul_(0.5)
    assert planet_version(model) != v0


def test_port_plan_time_matches_the_reference_classes_on_cfg2(mbrl):
    """bench.py's cpu_baseline times the oracle ("kind": "port") because the reference does not travel to the GPU box.
    The stand-in is justified: on full cfg2 plans (BASELINE.json configs[1]) the port and the UNMODIFIED reference classes
    (CEMOptimizer + ModelEnv.evaluate_action_sequences) take the same time within noise, and return bitwise the same plan."""
    import time

    obs, act, P, H, pop, iters = 17, 6, 20, 30, 500, 5
    om = po.make_synthetic_model(obs, act, ensemble_size=5, hid=200, seed=0, nontrivial_stats=False)
    s0 = (np.random.default_rng(0).standard_normal(obs) * 0.1).astype(np.float32)
    lb, ub = [[-1.0] * act] * H, [[1.0] * act] * H
    me, _, _ = build_reference_model_env(om, obs, act, generator=torch.Generator().manual_seed(0))
    ref_opt = mbrl.planning.CEMOptimizer(iters, 0.1, pop, lb, ub, 0.1, "cpu", return_mean_elites=True)
    gen = torch.Generator().manual_seed(0)

    def ref_plan():
        return ref_opt.optimize(lambda a: me.evaluate_action_sequences(a, initial_state=s0, num_particles=P), x0=torch.zeros(H, act))

    def port_plan():
        obj = lambda a: po.rollout(om, a, s0, P, global_rng=True, generator=gen)  # noqa: E731
        return po.cem_optimize(obj, torch.zeros(H, act), torch.tensor(lb), torch.tensor(ub), iters, 0.1, pop, 0.1, return_mean_elites=True)

    torch.manual_seed(5)
    r = ref_plan()
    torch.manual_seed(5)
    p = port_plan()
    assert torch.equal(r, p)
    t_ref, t_port = [], []
    for _ in range(3):  # interleaved so that machine noise hits both alike
        t0 = time.perf_counter(); ref_plan(); t_ref.append(time.perf_counter() - t0)  # noqa: E702
        t0 = time.perf_counter(); port_plan(); t_port.append(time.perf_counter() - t0)  # noqa: E702
    ratio = min(t_port) / min(t_ref)
    assert 0.6 < ratio < 1.5, f"port {min(t_port):.3f}s vs reference {min(t_ref):.3f}s per cfg2 plan"


def test_propagate_helpers_equal_the_reference(mbrl):
    """hipets.propagate* vs mbrl.util.math.propagate* (util/math.py:179-303), bitwise, same global-generator consumption."""
    import hipets

    g = torch.Generator().manual_seed(0)
    preds = (torch.randn(5, 12, 7, generator=g), torch.randn(5, 12, 7, generator=g))
    idx = torch.randint(5, (12,), generator=g)
    ref = mbrl.util.math
    assert torch.equal(hipets.propagate_from_indices(preds[0], idx), ref.propagate_from_indices(preds[0], idx))
    for method in ("expectation", "fixed_model", "random_model"):
        torch.manual_seed(3)
        mine = hipets.propagate(preds, method, idx)
        torch.manual_seed(3)
        theirs = ref.propagate(preds, method, idx)
        assert all(torch.equal(a, b) for a, b in zip(mine, theirs)), method
    with pytest.raises(ValueError, match="Invalid propagation method"):
        hipets.propagate(preds, "nope")
