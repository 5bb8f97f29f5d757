"""Parity tests proper: the HIP rollout (through the C-ABI) against the reference's golden vectors and
against the oracle on seeded inputs.  Tolerances (SURVEY.md section 8c tiers), written here:
  T1  one rollout step, injected perm/eps: next_obs, reward    rtol 1e-5, atol 2e-6
  T2  H-step returns [pop]                                     |err| <= 1e-4 * max(1, |v|)
f32 MFMA accumulation is an fmaf chain in a different k order than ATen's bmm, hence not bitwise."""
import glob
import os

import numpy as np
import pytest
import torch

import hipets
import oracle_cache as oc
from conftest import GOLDEN, to_spec
from oracle import pets_oracle as po
from oracle import device_draws
from oracle.golden_io import load_case

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def assert_returns_close(out, ref):
    out, ref = out.detach().cpu(), ref.detach().cpu()
    assert torch.isfinite(out).all(), "non-finite returns"  # NaN compares false with everything: never let it pass the bound below
    tol = 1e-4 * torch.clamp(ref.abs(), min=1.0)
    bad = (out - ref).abs() > tol
    assert not bad.any(), f"max err {(out - ref).abs().max():.3e} at {int(bad.nonzero()[0])}"


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "rollout_*.npz"))),
                         ids=lambda p: os.path.basename(p)[8:-4])
@pytest.mark.parametrize("R", [0, 1, 3])
def test_golden_reference_vectors_exact_mode(engine, path, R):
    om, meta, a = load_case(path)
    engine.set_model(to_spec(om, meta["obs_dim"], meta["act_dim"]))
    H, B = meta["H"], meta["pop"] * meta["P"]
    tno = torch.zeros(H, B, meta["obs_dim"], device=DEV)
    trw = torch.zeros(H, B, device=DEV)
    perms = a["perms"].to(DEV) if "perms" in a else None
    eps = a["eps"].to(DEV) if "eps" in a else None
    members = a.get("members")  # BasicEnsemble cases: the reference's randint member maps (CPU or device, both accepted)
    if members is not None and R == 1:
        members = members.to(DEV)
    out = engine.rollout(a["actions"].to(DEV), a["s0"].numpy(), meta["P"], mode="exact", perms=perms, eps=eps, members=members,
                         trace_next_obs=tno, trace_rewards=trw, rows_per_group=R)
    assert torch.allclose(tno[0].cpu(), a["next_obs_step0"], rtol=1e-5, atol=2e-6)  # T1
    assert torch.allclose(trw[0].cpu(), a["rewards_step0"].flatten(), rtol=1e-5, atol=2e-6)
    assert_returns_close(out, a["returns"])  # T2


def _random_case(obs, act, pop, P, H, seed=0, **mkw):
    om = po.make_synthetic_model(obs, act, seed=seed, **mkw)
    g = torch.Generator().manual_seed(seed + 1)
    actions = torch.rand(pop, H, act, generator=g) * 2 - 1
    s0 = (np.random.default_rng(seed).standard_normal(obs) * 0.1).astype(np.float32)
    if om.termination == "humanoid":
        s0[0] = 1.4  # inside the healthy z range (termination_fns.py:88-95): rows survive several steps instead of all ending at step 0
    if om.termination == "hopper":
        s0[0] = 1.25  # above the height threshold (:12-26)
    B = pop * P
    if om.propagation == "random_model":
        perms = torch.stack([torch.randperm(B, generator=g) for _ in range(H)])
    elif om.propagation == "fixed_model":
        perms = torch.randperm(B, generator=g)
    else:
        perms = None
    eps = None if om.deterministic else torch.randn(H, B, om.out_size, generator=g)
    return om, actions, s0, perms, eps


SIZES = [
    # obs, act, pop, P, H, model kwargs  (ragged tiles, tiny batches, wide/narrow nets, all propagation modes)
    (17, 6, 500, 20, 30, dict(ensemble_size=5, hid=200)),  # cfg2 at BASELINE.json's full size
    (4, 1, 100, 5, 15, dict(ensemble_size=5, hid=200, reward="cartpole", termination="cartpole")),  # cfg1
    (45, 17, 35, 20, 6, dict(ensemble_size=7, hid=200, elite=[0, 1, 2, 3, 4], termination="humanoid")),  # cfg4 shape
    (45, 17, 1036, 20, 4, dict(ensemble_size=7, hid=200, elite=[0, 1, 2, 3, 4], termination="humanoid")),  # cfg4 batch (iter 0)
    (17, 6, 2000, 20, 5, dict(ensemble_size=5, hid=200)),  # cfg5 batch: 40 000 rows
    (17, 6, 1, 5, 3, dict(ensemble_size=5, hid=200)),  # single candidate
    (17, 6, 7, 5, 4, dict(ensemble_size=5, hid=33)),  # odd hidden width
    (17, 6, 48, 10, 3, dict(ensemble_size=2, hid=256, num_layers=2)),
    (17, 6, 30, 4, 5, dict(ensemble_size=3, hid=512, num_layers=3, propagation="fixed_model")),
    (6, 2, 10, 3, 5, dict(ensemble_size=5, hid=16, propagation="expectation", normalizer="f32")),
    (376, 17, 10, 5, 2, dict(ensemble_size=5, hid=200, termination="humanoid")),  # cfg4' input width 393
    # cfg4' (BASELINE configs[3] literally: Humanoid-v4, obs 376 -> 752 output columns, 7 members / 5 elites) at a multi-turn size:
    # 4 200 rows = 840 per member = 53 one-tile workgroups per member, 265 logical workgroups on 256 CUs
    (376, 17, 210, 20, 3, dict(ensemble_size=7, hid=200, elite=[0, 1, 2, 3, 4], termination="humanoid")),
    # The workloads the reference SHIPS, at full size (oracle/make_golden.py FULL_CASES stock_*):
    # pets_halfcheetah (conf/overrides/pets_halfcheetah.yaml:1-24, env/pets_halfcheetah.py:91-113): obs 18 through preprocess_fn,
    # no_delta_list [0], 7 members / 5 elites, pop 400 x 20 particles, H 30
    (18, 6, 400, 20, 30, dict(ensemble_size=7, hid=200, elite=[0, 2, 3, 5, 6], obs_process="halfcheetah", no_delta_list=[0])),
    # pets_cartpole (conf/overrides/pets_cartpole.yaml:1-21, util/env.py:71-74): pop 350 x 20 particles, H 15, 7 members / 5 elites
    (4, 1, 350, 20, 15, dict(ensemble_size=7, hid=200, elite=[1, 2, 4, 5, 6], reward="cartpole", termination="cartpole")),
    # the other PETS obs preprocessor (env/pets_cartpole.py:78-101: [sin s1, cos s1, s0, s2:], one input column more than obs dims) with
    # its reward function, in-kernel randomness
    (4, 1, 80, 5, 6, dict(ensemble_size=5, hid=200, obs_process="cartpole_pets", reward="cartpole_pets")),
    # shipped workloads with LEARNED rewards and no termination function, at full size: pets_pusher (conf/overrides/pets_pusher.yaml:
    # obs 20 / act 7, pop 350 x 20 particles, H 25) and pets_mppi_halfcheetah (pets_mppi_halfcheetah.yaml: obs 18 through
    # preprocess_fn, no_delta_list [0], 350 x 20, H 30): the reward is the sampled last output column (one_dim_tr_model.py:287)
    (20, 7, 350, 20, 25, dict(ensemble_size=7, hid=200, elite=[0, 1, 3, 4, 6], learned_rewards=True, reward=None)),
    (18, 6, 350, 20, 30, dict(ensemble_size=7, hid=200, elite=[0, 2, 3, 5, 6], obs_process="halfcheetah", no_delta_list=[0],
                              learned_rewards=True, reward=None)),
    # pets_inv_pendulum (conf/overrides/pets_inv_pendulum.yaml: obs 4 / act 1, learned reward + the inverted_pendulum termination
    # function -- isfinite over every state dim, termination_fns.py:47-55 --, pop 480 x 20 particles, H 45)
    (4, 1, 480, 20, 45, dict(ensemble_size=7, hid=200, elite=[0, 2, 3, 5, 6], learned_rewards=True, reward=None, termination="inverted_pendulum")),
    # pets_hopper (conf/overrides/pets_hopper.yaml: obs 11 / act 3, learned reward + the hopper termination function over all eleven
    # dims, termination_fns.py:12-26; pop 350 x 20 particles, H 30) and pets_reacher (pets_reacher.yaml: obs 17 / act 7, learned reward,
    # no_delta_list [0], pop 350 x 20, H 15) at full size (round 5: until then compared with the generic kernel only at this size)
    (11, 3, 350, 20, 30, dict(ensemble_size=7, hid=200, elite=[0, 1, 3, 4, 6], learned_rewards=True, reward=None, termination="hopper")),
    (17, 7, 350, 20, 15, dict(ensemble_size=7, hid=200, elite=[0, 1, 3, 4, 6], no_delta_list=[0], learned_rewards=True, reward=None)),
]
# in-kernel randomness replays (FAST / DEVICE): everything but the expectation-propagation f32-normaliser case in FAST
FAST_SIZES = SIZES[:9] + SIZES[10:]
DEVICE_SIZES = SIZES


@pytest.mark.parametrize("case", SIZES, ids=lambda c: f"obs{c[0]}_pop{c[2]}x{c[3]}_H{c[4]}_hid{c[5]['hid']}")
def test_exact_mode_matches_oracle(engine, case):
    obs, act, pop, P, H, mkw = case
    om, actions, s0, perms, eps = _random_case(obs, act, pop, P, H, **mkw)
    engine.set_model(to_spec(om, obs, act))
    # (every input derives from seeds on the CPU: the oracle's answer is memoised under their digest, tests/oracle_cache.py)
    ref = oc.cached("rollout_sizes", ["exact", *oc.model_parts(om), actions, s0, P, perms, eps], lambda: po.rollout(om, actions, s0, P, perms=perms, eps=eps),
                    verify=case is SIZES[0])  # cfg2 at full size: the oracle always runs and the stored entry must equal its answer
    out = engine.rollout(actions.to(DEV), s0, P, mode="exact", perms=None if perms is None else perms.to(DEV),
                         eps=None if eps is None else eps.to(DEV))
    assert_returns_close(out, ref)


@pytest.mark.parametrize("case", FAST_SIZES, ids=lambda c: f"obs{c[0]}_pop{c[2]}x{c[3]}_H{c[4]}_hid{c[5]['hid']}")
def test_fast_mode_replayed_through_oracle(engine, case):
    """FAST mode end to end (balanced member schedule + Philox eps drawn in-kernel): export the kernel's own
    randomness through the ABI, replay it through the oracle's explicit row->member form, compare returns."""
    obs, act, pop, P, H, mkw = case
    om, actions, s0, _, _ = _random_case(obs, act, pop, P, H, **mkw)
    engine.set_model(to_spec(om, obs, act))
    seed, sid = 1234, 77
    nwg, r = engine.fast_geometry(pop, P, H)
    out = engine.rollout(actions.to(DEV), s0, P, mode="fast", seed=seed, stream_id=sid)
    sched = engine.fast_schedule(H, nwg, seed, sid).cpu()
    rows = torch.arange(pop * P)
    wg = device_draws.fast_row_workgroup(rows, P, r)
    members = torch.stack([sched[0 if om.propagation == "fixed_model" else t][wg].long() for t in range(H)])
    # eps = the library's Philox normals of (seed, stream): a function of the counters, exported only when the oracle has to run
    ref = oc.cached("rollout_sizes", ["fast", *oc.model_parts(om), actions, s0, P, members, ("philox", seed, sid)],
                    lambda: po.rollout(om, actions, s0, P, members=members, eps=engine.fast_normals(H, pop * P, seed, sid).cpu()),
                    verify=case is SIZES[0])
    assert_returns_close(out, ref)
    # the schedule is balanced: every member slot gets floor/ceil(nwg / M) workgroups at every step
    M = len(om.active_members)
    for t in range(H):
        counts = torch.bincount(sched[t].long(), minlength=M)
        assert counts.max() - counts.min() <= 1 and counts.sum() == nwg
    if om.propagation == "fixed_model":
        assert (sched == sched[0]).all()  # TS-infinity: one member per particle for the whole horizon


def test_known_answer_dummy_model(engine):
    """tests/core/test_models.py:365-385 on the GPU: returns == H (H+1) / 2 * a, P, H in 1..9."""
    from test_oracle_golden import dummy_model_as_mlp

    act_dim = 2
    om = dummy_model_as_mlp(act_dim)
    engine.set_model(to_spec(om, 1, act_dim))
    for P in range(1, 10):
        for H in (1, 2, 5, 9):
            acts = torch.stack([torch.ones(H, act_dim), 2 * torch.ones(H, act_dim)])
            expected = H * (H + 1) * acts[..., 0, 0] / 2
            for mode in ("exact", "fast"):
                out = engine.rollout(acts.to(DEV), np.zeros(1), P, mode=mode)
                assert torch.allclose(expected, out.cpu()), (P, H, mode)


def test_fast_normals_are_standard_normal(engine):
    om = po.make_synthetic_model(17, 6, ensemble_size=5, hid=16)
    engine.set_model(to_spec(om, 17, 6))
    z = engine.fast_normals(30, 10000, seed=5, stream_id=3).double().flatten().cpu()
    n = z.numel()
    assert abs(z.mean()) < 5 / np.sqrt(n)
    assert abs(z.var() - 1) < 5 * np.sqrt(2 / n)
    assert abs((z**4).mean() - 3) < 0.05
    z2 = engine.fast_normals(30, 10000, seed=5, stream_id=4).double().flatten().cpu()
    assert abs(torch.corrcoef(torch.stack([z, z2]))[0, 1]) < 5 / np.sqrt(n)  # streams are independent
    assert torch.equal(engine.fast_normals(3, 50, seed=5, stream_id=3).cpu(), engine.fast_normals(3, 50, seed=5, stream_id=3).cpu())


def test_fast_mode_statistics_match_exact_mode(engine):
    """T5: same action sequences evaluated in FAST mode (Philox, block-balanced TS1) and in EXACT mode (torch
    perms / eps): per-candidate return means agree within 4 sigma of the seed-to-seed spread."""
    obs, act, pop, P, H = 17, 6, 64, 20, 10
    om = po.make_synthetic_model(obs, act, ensemble_size=5, hid=64, seed=2)
    engine.set_model(to_spec(om, obs, act))
    g = torch.Generator().manual_seed(0)
    actions = (torch.rand(pop, H, act, generator=g) * 2 - 1).to(DEV)
    s0 = np.zeros(obs, np.float32)
    n = 24
    fast = torch.stack([engine.rollout(actions, s0, P, mode="fast", seed=9, stream_id=i) for i in range(n)]).cpu()
    exact = []
    for i in range(n):
        perms = torch.stack([torch.randperm(pop * P, generator=g) for _ in range(H)]).to(DEV)
        eps = torch.randn(H, pop * P, obs, generator=g).to(DEV)
        exact.append(engine.rollout(actions, s0, P, mode="exact", perms=perms, eps=eps))
    exact = torch.stack(exact).cpu()
    se = torch.sqrt(fast.var(0) / n + exact.var(0) / n)
    zscore = (fast.mean(0) - exact.mean(0)) / se
    assert zscore.abs().max() < 4.5, zscore.abs().max()
    assert abs(zscore.mean()) < 4.5 / np.sqrt(pop)
    ratio = fast.var(0).mean() / exact.var(0).mean()  # spread of the particle-mean estimator is comparable
    assert 0.6 < ratio < 1.6, ratio


def test_fast_mode_is_deterministic_and_seed_sensitive(engine):
    om, actions, s0, _, _ = _random_case(17, 6, 50, 10, 8, ensemble_size=5, hid=64)
    engine.set_model(to_spec(om, 17, 6))
    a = engine.rollout(actions.to(DEV), s0, 10, mode="fast", seed=1, stream_id=2)
    b = engine.rollout(actions.to(DEV), s0, 10, mode="fast", seed=1, stream_id=2)
    c = engine.rollout(actions.to(DEV), s0, 10, mode="fast", seed=1, stream_id=3)
    assert torch.equal(a, b) and not torch.equal(a, c)


def test_linearity_in_particles_full_size(engine):
    """Size-independent property at cfg2 scale: with a deterministic model every particle of a candidate is
    identical, so returns do not depend on P, and candidates are independent of their batch neighbours."""
    obs, act, pop, H = 17, 6, 500, 30
    om = po.make_synthetic_model(obs, act, ensemble_size=1, hid=200, deterministic=True, propagation="expectation")
    engine.set_model(to_spec(om, obs, act))
    g = torch.Generator().manual_seed(0)
    actions = (torch.rand(pop, H, act, generator=g) * 2 - 1).to(DEV)
    s0 = np.zeros(obs, np.float32)
    r1 = engine.rollout(actions, s0, 1, mode="fast")
    r20 = engine.rollout(actions, s0, 20, mode="fast")
    assert torch.allclose(r1, r20, rtol=0, atol=1e-5)
    sub = engine.rollout(actions[100:137].contiguous(), s0, 20, mode="fast")
    assert torch.allclose(sub, r20[100:137], rtol=0, atol=1e-5)
    rex = engine.rollout(actions, s0, 20, mode="exact")
    assert torch.allclose(rex, r20, rtol=0, atol=1e-5)


def test_error_behaviour(engine):
    import hipets

    om = po.make_synthetic_model(17, 6, ensemble_size=5, hid=16)
    engine.set_model(to_spec(om, 17, 6))
    acts = torch.zeros(3, 4, 6, device=DEV)
    with pytest.raises(hipets.HipetsError, match="multiple of the number of models"):  # gaussian_mlp.py:195-200
        engine.rollout(acts, np.zeros(17), 4, mode="exact", perms=torch.zeros(4, 12, dtype=torch.int64, device=DEV))
    with pytest.raises(hipets.HipetsError, match="needs opts.perms"):
        engine.rollout(acts, np.zeros(17), 5, mode="exact")
    with pytest.raises(ValueError):
        engine.rollout(torch.zeros(3, 4, 5, device=DEV), np.zeros(17), 5)
    with pytest.raises(ValueError):
        engine.rollout(acts, np.zeros(16), 5)
    with pytest.raises(ValueError):
        engine.rollout(acts.cpu(), np.zeros(17), 5)


def test_weights_resnapshot_after_training_step(engine):
    """Engine reads live parameters again after they change (ModelTrainer hook, model_trainer.py:288-296)."""
    om, actions, s0, perms, eps = _random_case(17, 6, 20, 5, 4, ensemble_size=5, hid=32)
    engine.set_model(to_spec(om, 17, 6))
    a = engine.rollout(actions.to(DEV), s0, 5, mode="exact", perms=perms.to(DEV), eps=eps.to(DEV))
    om.weights[1] = om.weights[1] * 1.5
    om.elite_models = [0, 2, 3, 4, 1]
    engine.set_model(to_spec(om, 17, 6))
    b = engine.rollout(actions.to(DEV), s0, 5, mode="exact", perms=perms.to(DEV), eps=eps.to(DEV))
    assert not torch.allclose(a, b)
    assert_returns_close(b, po.rollout(om, actions, s0, 5, perms=perms, eps=eps))


@pytest.mark.parametrize("sample", [False, True])
@pytest.mark.parametrize("prop", ["random_model", "fixed_model", "expectation"])
def test_step_exact_matches_oracle(engine, prop, sample):
    """hipets_step (ModelEnv.step, model_env.py:87-140): per-row initial states, one transition."""
    obs, act, B = 11, 3, 120
    om = po.make_synthetic_model(obs, act, ensemble_size=5, hid=40, seed=9, termination="hopper", no_delta_list=[1], propagation=prop)
    engine.set_model(to_spec(om, obs, act))
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, obs, generator=g) * 0.3
    x[:, 0] += 0.75
    a = torch.rand(B, act, generator=g) * 2 - 1
    perm = None if prop == "expectation" else torch.randperm(B, generator=g)
    eps = torch.randn(B, obs, generator=g) if sample else None
    rn, rr, rd = po.step(om, x, a, perm=perm, eps=eps, sample=sample)
    nobs, rew, done = engine.step(x.to(DEV), a.to(DEV), mode="exact", sample=sample, perm=None if perm is None else perm.to(DEV),
                                  eps=None if eps is None else eps.to(DEV))
    assert torch.allclose(nobs.cpu(), rn, rtol=1e-5, atol=2e-6)
    assert torch.allclose(rew.cpu(), rr, rtol=1e-5, atol=2e-6)
    assert torch.equal(done.cpu(), rd) and 0 < int(rd.sum()) < B  # a mix of terminated / alive rows


@pytest.mark.parametrize("prop", ["random_model", "fixed_model"])
def test_model_env_class_default_mode_has_reference_semantics(engine, prop):
    """hipets.ModelEnv() WITHOUT a mode argument (round 6: 'device'): reset / step draw the reference's per-row balanced member shuffle
    (gaussian_mlp.py:201-211; TS-infinity: ONE shuffle at reset, model.py:404-407, kept for every step while the eps change) and iid eps
    in-kernel.  Two consecutive steps replayed through the oracle with the exported permutation(s) and normals."""
    import hipets

    obs, act, B = 17, 6, 120
    om = po.make_synthetic_model(obs, act, ensemble_size=5, hid=32, seed=3, propagation=prop)
    env = hipets.ModelEnv(to_spec(om, obs, act), engine=engine, seed=11)
    assert env.mode == "device"
    g = torch.Generator().manual_seed(0)
    obs0 = (torch.randn(B, obs, generator=g) * 0.2).numpy()
    a1, a2 = torch.rand(B, act, generator=g) * 2 - 1, torch.rand(B, act, generator=g) * 2 - 1
    state = env.reset(obs0, return_as_np=False)
    n1, r1, d1, state = env.step(a1.to(DEV), state, sample=True)
    n2, r2, d2, state = env.step(a2.to(DEV), state, sample=True)
    if prop == "fixed_model":  # the reset's stream (1) keys the permutation of both steps; propagation_indices is that permutation
        p1 = p2 = engine.device_perms(1, B, 11, 1).cpu()
        assert torch.equal(state["propagation_indices"].cpu(), p1)
    else:
        p1, p2 = engine.device_perms(1, B, 11, 1).cpu()[0], engine.device_perms(1, B, 11, 2).cpu()[0]
        assert not torch.equal(p1, p2)
    e1, e2 = engine.fast_normals(1, B, 11, 1).cpu()[0], engine.fast_normals(1, B, 11, 2).cpu()[0]
    x0 = torch.from_numpy(obs0.astype(np.float32))
    rn1, rr1, _ = po.step(om, x0, a1, perm=p1, eps=e1, sample=True)
    assert torch.allclose(n1.cpu(), rn1, rtol=1e-5, atol=2e-6) and torch.allclose(r1.cpu(), rr1, rtol=1e-5, atol=2e-6)
    rn2, rr2, _ = po.step(om, n1.cpu(), a2, perm=p2, eps=e2, sample=True)
    assert torch.allclose(n2.cpu(), rn2, rtol=1e-5, atol=2e-6) and torch.allclose(r2.cpu(), rr2, rtol=1e-5, atol=2e-6)
    assert not torch.equal(e1, e2)
    # the objective built without a mode argument is the same mode (and evaluate_action_sequences runs it)
    fn = hipets.make_eval_fn(to_spec(om, obs, act), 5, engine=engine)
    assert fn.mode == "device" and fn.kernel_mode == "device"
    vals = env.evaluate_action_sequences(torch.zeros(10, 4, act, device=DEV), obs0[0], 5)
    assert vals.shape == (10,) and torch.isfinite(vals).all()


@pytest.mark.parametrize("M,fixed,iid", [(5, False, False), (5, True, False), (7, False, False), (4, False, True)])
def test_fast_schedule_export_is_the_cpu_restatement(engine, M, fixed, iid):
    """hipets_fast_schedule (what every FAST workgroup draws for itself in its prologue, common.hpp fast_member) == oracle/device_draws.
    member_schedule, integer for integer, from one workgroup to thousands; balanced to within one workgroup per slot; and with fewer
    workgroups than members every member is still drawn (the per-step rotation)."""
    kind = "basic_ensemble" if iid else "gaussian_mlp"
    om = po.make_synthetic_model(6, 2, ensemble_size=M, hid=16, seed=1, propagation="fixed_model" if fixed else "random_model", ensemble_kind=kind)
    engine.set_model(to_spec(om, 6, 2))
    for nwg, H in [(1, 40), (3, 40), (32, 15), (209, 30), (504, 7), (6250, 2)]:
        got = engine.fast_schedule(H, nwg, 1234, 77).cpu().numpy()
        want = device_draws.member_schedule(H, nwg, M, 1234, 77, fixed=fixed, iid=iid)
        assert np.array_equal(got, want), (nwg, H)
        if not iid:
            for t in range(H):
                c = np.bincount(got[t], minlength=M)
                assert c.max() - c.min() <= 1
        if fixed:
            assert (got == got[0]).all()
    if not fixed:
        few = engine.fast_schedule(400, 3, 5, 6).cpu().numpy()
        assert set(np.unique(few).tolist()) == set(range(M))


def test_member_schedule_of_another_geometry_is_refused(engine):
    """ABI v6: a caller-provided member schedule states its length; one sized for another geometry fails the call instead of being read
    with the wrong stride (round-5 advice: the FAST geometry changed under ABI v5)."""
    import ctypes as C

    from hipets import _lib

    obs, act, pop, P, H = 17, 6, 64, 5, 4
    om = po.make_synthetic_model(obs, act, ensemble_size=5, hid=32, seed=3)
    engine.set_model(to_spec(om, obs, act))
    actions = torch.zeros(pop, H, act, device=DEV)
    nwg, r = engine.fast_geometry(pop, P, H)
    good = engine.fast_schedule(H, nwg, 1, 2)
    ref = engine.rollout(actions, np.zeros(obs, np.float32), P, mode="fast", seed=1, stream_id=2)
    same = engine.rollout(actions, np.zeros(obs, np.float32), P, mode="fast", seed=1, stream_id=2, member_schedule=good)
    assert torch.equal(ref, same)  # the injected schedule IS the in-kernel draw
    o = _lib.RolloutOpts()
    o.mode, o.seed, o.stream_id = _lib.MODES["fast"], 1, 2
    o.member_schedule, o.member_schedule_len = good.data_ptr(), H * nwg + 1
    out = torch.empty(pop, device=DEV)
    s0 = np.zeros(obs, np.float32)
    rc = engine._lib.hipets_rollout(engine._h, C.c_void_p(actions.data_ptr()), s0.ctypes.data_as(C.c_void_p), pop, H, P, C.byref(o),
                                    C.c_void_p(out.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc != 0 and b"member_schedule holds" in engine._lib.hipets_last_error()


def test_model_env_class_reset_step_fast_mode_replayed(engine):
    """hipets.ModelEnv.reset/step in FAST mode, replayed through the oracle with the exported schedule and normals."""
    import hipets

    obs, act, B = 17, 6, 100
    om = po.make_synthetic_model(obs, act, ensemble_size=5, hid=32, seed=3)
    env = hipets.ModelEnv(to_spec(om, obs, act), engine=engine, mode="fast", seed=11)
    g = torch.Generator().manual_seed(0)
    obs0 = (torch.randn(B, obs, generator=g) * 0.2).numpy()
    a = torch.rand(B, act, generator=g) * 2 - 1
    state = env.reset(obs0, return_as_np=False)
    n1, r1, d1, state = env.step(a.to(DEV), state, sample=True)
    nwg, r = engine.fast_geometry(B, 1, 1)
    sched = engine.fast_schedule(1, nwg, 11, 1).cpu()[0]
    eps = engine.fast_normals(1, B, 11, 1).cpu()[0]
    members = sched[torch.arange(B) // (16 * r)].long()
    rn, rr, rd = po.step(om, torch.from_numpy(obs0.astype(np.float32)), a, member_of_row=members, eps=eps, sample=True)
    assert torch.allclose(n1.cpu(), rn, rtol=1e-5, atol=2e-6) and torch.allclose(r1.cpu(), rr, rtol=1e-5, atol=2e-6)
    n2, _, _, _ = env.step(a.to(DEV), state, sample=False)  # deterministic mean
    assert torch.isfinite(n2).all() and not torch.equal(n2, n1)
    nnp = env.reset(obs0)  # return_as_np=True default like the reference
    out = env.step(a.numpy(), nnp)
    assert isinstance(out[0], np.ndarray) and out[1].shape == (B, 1) and out[2].dtype == bool
    with pytest.raises(ValueError, match="multiple of the number of models"):
        env.reset(obs0[:7])
    vals = env.evaluate_action_sequences(torch.zeros(10, 4, act, device=DEV), obs0[0], 5)
    assert vals.shape == (10,)


def test_basic_ensemble_per_member_logvar_bounds_and_fast_mode(engine):
    """BasicEnsemble of single-member GaussianMLPs: every member clamps with ITS OWN learned logvar bounds
    (each member is its own GaussianMLP, gaussian_mlp.py:117-122); members are drawn iid.  EXACT vs the oracle with an
    injected unbalanced map; FAST replayed through the oracle with the exported (iid, unbalanced) schedule."""
    obs, act, pop, P, H, E = 9, 3, 37, 3, 6, 4
    om = po.make_synthetic_model(obs, act, ensemble_size=E, hid=32, seed=21, ensemble_kind="basic_ensemble")
    g = torch.Generator().manual_seed(5)
    lo = -10 + torch.rand(E, obs, generator=g) * 6      # per-member bounds, far enough apart to matter
    hi = -3 + torch.rand(E, obs, generator=g) * 3
    om.min_logvar, om.max_logvar = lo, hi
    engine.set_model(to_spec(om, obs, act))
    B = pop * P
    actions = torch.rand(pop, H, act, generator=g) * 2 - 1
    s0 = (np.random.default_rng(0).standard_normal(obs) * 0.1).astype(np.float32)
    members = torch.randint(E, (H, B), generator=g)
    members[:, : B // 2] = 0  # strongly unbalanced on purpose
    eps = torch.randn(H, B, obs, generator=g)

    def oracle(member_map, eps_):
        return po.rollout(om, actions, s0, P, members=member_map, eps=eps_)

    ref = oracle(members, eps)
    out = engine.rollout(actions.to(DEV), s0, P, mode="exact", members=members, eps=eps.to(DEV))
    assert_returns_close(out, ref)
    # FAST: iid schedule (not balanced), replay
    seed, sid = 3, 8
    nwg, r = engine.fast_geometry(pop, P, H)
    out_f = engine.rollout(actions.to(DEV), s0, P, mode="fast", seed=seed, stream_id=sid)
    sched = engine.fast_schedule(H, nwg, seed, sid).cpu()
    assert sched.min() >= 0 and sched.max() < E
    rows = torch.arange(B)
    wg = device_draws.fast_row_workgroup(rows, P, r)
    fast_members = torch.stack([sched[t][wg].long() for t in range(H)])
    assert_returns_close(out_f, oracle(fast_members, engine.fast_normals(H, B, seed, sid).cpu()))
    # batch sizes that are not multiples of the ensemble size are accepted (no GaussianMLP batch rule)
    assert B % E != 0


@pytest.mark.parametrize("propagation", ["random_model", "fixed_model"])
def test_model_env_exact_mode_basic_ensemble_replays_generator_order(engine, propagation):
    """hipets.ModelEnv in EXACT mode on a BasicEnsemble model draws what the reference draws, from the same generator in
    the same order (reset: fixed_model randint; step: random_model randint, then the normal): two steps replayed through
    the oracle with the draws regenerated from an identically seeded generator."""
    import hipets

    obs, act, B, E = 7, 2, 23, 4  # 23 rows: not a multiple of the ensemble size, accepted for BasicEnsemble
    om = po.make_synthetic_model(obs, act, ensemble_size=E, hid=24, seed=4, ensemble_kind="basic_ensemble", propagation=propagation)
    env = hipets.ModelEnv(to_spec(om, obs, act), engine=engine, mode="exact", seed=0)
    env._eval._rng = torch.Generator().manual_seed(77)
    g = torch.Generator().manual_seed(1)
    obs0 = (torch.randn(B, obs, generator=g) * 0.2).numpy().astype(np.float32)
    acts = [torch.rand(B, act, generator=g) * 2 - 1 for _ in range(2)]
    state = env.reset(obs0, return_as_np=False)
    replay = torch.Generator().manual_seed(77)
    fixed = torch.randint(E, (B,), generator=replay) if propagation == "fixed_model" else None
    x = torch.from_numpy(obs0)
    for a in acts:
        nobs, rew, done, state = env.step(a.to(DEV), state, sample=True)
        members = fixed if fixed is not None else torch.randint(E, (B,), generator=replay)
        eps = torch.empty(B, obs).normal_(0.0, 1.0, generator=replay)
        rn, rr, rd = po.step(om, x, a, member_of_row=members, eps=eps, sample=True)
        assert torch.allclose(nobs.cpu(), rn, rtol=1e-5, atol=2e-6) and torch.allclose(rew.cpu(), rr, rtol=1e-5, atol=2e-6)
        assert torch.equal(done.cpu(), rd)
        x = rn


@pytest.mark.parametrize("prop", ["random_model", "fixed_model"])
def test_gaussian_mlp_with_explicit_member_maps(engine, prop):
    """per-row member assignment in the sense of mbrl.util.math.propagate_from_indices
    (util/math.py:180-196) for a GaussianMLP model -- every row evaluated by the member an arbitrary (unbalanced) index
    tensor names, batch size NOT a multiple of the member count -- through EXACT mode's members= input, vs the oracle."""
    obs, act, pop, P, H, E = 11, 3, 23, 3, 5, 5  # B = 69, not a multiple of 5
    om = po.make_synthetic_model(obs, act, ensemble_size=E, hid=40, seed=31, propagation=prop)
    engine.set_model(to_spec(om, obs, act))
    g = torch.Generator().manual_seed(9)
    B = pop * P
    actions = torch.rand(pop, H, act, generator=g) * 2 - 1
    s0 = (np.random.default_rng(1).standard_normal(obs) * 0.1).astype(np.float32)
    members = torch.randint(E, (B,) if prop == "fixed_model" else (H, B), generator=g)
    members[..., : B // 3] = 2  # unbalanced on purpose
    eps = torch.randn(H, B, obs, generator=g)
    ref = po.rollout(om, actions, s0, P, members=members, eps=eps)
    out = engine.rollout(actions.to(DEV), s0, P, mode="exact", members=members, eps=eps.to(DEV))
    assert_returns_close(out, ref)
    # one transition, the same way
    x = torch.randn(B, obs, generator=g) * 0.1
    a = torch.rand(B, act, generator=g) * 2 - 1
    m1 = members if members.ndim == 1 else members[0]
    nobs, rew, done = engine.step(x.to(DEV), a.to(DEV), mode="exact", sample=True, members=m1, eps=eps[0].to(DEV))
    r_nobs, r_rew, _ = po.step(om, x, a, member_of_row=m1, eps=eps[0], sample=True)
    assert torch.allclose(nobs.cpu(), r_nobs, rtol=1e-5, atol=2e-6) and torch.allclose(rew.cpu(), r_rew, rtol=1e-5, atol=2e-6)


def assert_same_arithmetic(a, b, kernel_class):
    """Two instances of the rollout kernel on the same call.  Bit for bit -- except where one of them is a K-SPLIT instance: the
    FUSED (not WIDE) instances with ONE row tile per workgroup deal the k range of the 13th hidden column tile to the four waves
    (rollout.hpp KSpec::KSPLIT, round 5), i.e. sum hidden columns 192..207 in another order; they agree with every other instance to
    rounding, which is held to a tenth of the tolerance the same returns get against the oracle (T2).  `kernel_class` =
    Engine.kernel_class(...) of the shape-specialised side: (class name, row tiles).  Everything else -- WIDE, hidden-static and
    generic instances at any row-tile count, fused ones at R >= 2 -- must be bit for bit (round-5 advice: a blanket tolerance for
    every one-tile call would let a summation-order or hazard regression in those through)."""
    assert torch.isfinite(a).all() and torch.isfinite(b).all()
    cls, row_tiles = kernel_class
    if not (cls == "fused" and row_tiles == 1):
        assert torch.equal(a, b)
        return
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    bad = (a - b).abs() > 1e-5 * torch.clamp(b.abs(), min=1.0)
    assert not bad.any(), f"max |a - b| {(a - b).abs().max():.3e} at candidate {int(bad.nonzero()[0])}"


@pytest.mark.parametrize("mode", ["fast", "device"])
@pytest.mark.parametrize("case", [SIZES[0], SIZES[1], SIZES[3], SIZES[4]], ids=lambda c: f"obs{c[0]}_pop{c[2]}x{c[3]}_H{c[4]}")
def test_shape_specialised_kernels_equal_the_generic_kernel_bitwise(engine, case, mode):
    """The BASELINE shapes (cfg2 / cfg1 / cfg4 / cfg5) run shape-specialised ("lean") instances of the rollout kernel -- layer
    shapes, normaliser kind, reward / termination functions as template arguments, unused features compiled out.  Same
    arithmetic: their returns equal the generic instance's bit for bit (opts.generic_kernel forces the latter)."""
    obs, act, pop, P, H, mkw = case
    om, actions, s0, _, _ = _random_case(obs, act, pop, P, H, **mkw)
    engine.set_model(to_spec(om, obs, act))
    a = engine.rollout(actions.to(DEV), s0, P, mode=mode, seed=11, stream_id=3)
    b = engine.rollout(actions.to(DEV), s0, P, mode=mode, seed=11, stream_id=3, generic_kernel=True)
    c = engine.rollout(actions.to(DEV), s0, P, mode=mode, seed=11, stream_id=3, generic_kernel=2)  # the hidden-static instance
    assert torch.equal(b, c)
    assert_same_arithmetic(a, b, engine.kernel_class(pop, P, H, mode))


# every model of the reference's default hidden width (200) that has no shape-specialised instance: the workloads the reference
# ships (obs preprocessing, other reward / termination functions), propagation methods, normalisers, learned rewards
HID200_CASES = [SIZES[12], SIZES[13], SIZES[14],
                (11, 3, 60, 5, 6, dict(ensemble_size=5, hid=200, termination="hopper", normalizer="f32")),
                (17, 6, 40, 6, 5, dict(ensemble_size=3, hid=200, propagation="fixed_model", learned_rewards=True, reward=None)),
                (23, 7, 25, 4, 4, dict(ensemble_size=4, hid=208, reward="pusher", propagation="expectation")),
                (6, 2, 33, 5, 5, dict(ensemble_size=5, hid=193, deterministic=True, normalizer="none")),
                # other depths: the hidden-static path serves "the first op", "every op whose K and N are the hidden width", "the last op"
                (17, 6, 30, 5, 4, dict(ensemble_size=5, hid=200, num_layers=1)),
                (17, 6, 30, 5, 4, dict(ensemble_size=5, hid=200, num_layers=6, termination="walker2d")),
                # learned rewards in the fused tail (pets_pusher, pets_mppi_halfcheetah at full size; pets_reacher's shape): the lane
                # that holds output column obs_dim keeps the row's total -- first / second dim of its pair, first / last lane group,
                # the first column of a tile
                SIZES[15], SIZES[16], SIZES[17],
                (17, 7, 64, 5, 6, dict(ensemble_size=5, hid=200, no_delta_list=[0], learned_rewards=True, reward=None)),
                (16, 3, 40, 5, 5, dict(ensemble_size=5, hid=200, learned_rewards=True, reward=None)),
                (23, 4, 40, 5, 5, dict(ensemble_size=5, hid=200, learned_rewards=True, reward=None)),
                # ... next to a termination function that tests every state dim (pets_inv_pendulum: obs 4): the lane with dims 0, 1
                # keeps the total and fetches the reward from the column's lane -- second / third lane group, second / first dim
                (4, 1, 96, 5, 8, dict(ensemble_size=5, hid=200, learned_rewards=True, reward=None, termination="inverted_pendulum")),
                (3, 1, 96, 5, 8, dict(ensemble_size=5, hid=200, learned_rewards=True, reward=None, termination="inverted_pendulum")),
                # pets_hopper's shape (FAST instances for one / two row tiles: flags of the rows' dims through LDS), at full size and small
                (11, 3, 350, 20, 30, dict(ensemble_size=7, hid=200, elite=[0, 1, 3, 4, 6], learned_rewards=True, reward=None, termination="hopper")),
                (11, 3, 64, 5, 8, dict(ensemble_size=5, hid=200, learned_rewards=True, reward=None, termination="hopper")),
                (10, 3, 64, 5, 8, dict(ensemble_size=5, hid=200, learned_rewards=True, reward=None, termination="hopper"))]


@pytest.mark.parametrize("mode", ["fast", "device"])
@pytest.mark.parametrize("rows_per_group", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("case", HID200_CASES, ids=lambda c: f"obs{c[0]}_pop{c[2]}x{c[3]}_H{c[4]}_hid{c[5]['hid']}")
def test_hidden_static_instances_equal_the_generic_kernel_bitwise(engine, case, mode, rows_per_group):
    """Models whose hidden layers have 13 column tiles (hid 193..208: the reference's default 200) run an instance of the rollout
    kernel with the hidden layers' shape as a compile-time fact and everything else -- obs preprocessing, reward / termination
    functions, normaliser, output width, propagation -- decided at run time (KSpec::HID_STATIC).  Same arithmetic as the fully
    generic instance: same bits, for every row-tile count."""
    obs, act, pop, P, H, mkw = case
    if rows_per_group and pop * P * H > 100000 and rows_per_group != 2:
        pytest.skip("full-size cases: the default geometry and one forced one")
    om, actions, s0, _, _ = _random_case(obs, act, pop, P, H, **mkw)
    engine.set_model(to_spec(om, obs, act))
    a = engine.rollout(actions.to(DEV), s0, P, mode=mode, seed=11, stream_id=3, rows_per_group=rows_per_group)
    b = engine.rollout(actions.to(DEV), s0, P, mode=mode, seed=11, stream_id=3, rows_per_group=rows_per_group, generic_kernel=True)
    c = engine.rollout(actions.to(DEV), s0, P, mode=mode, seed=11, stream_id=3, rows_per_group=rows_per_group, generic_kernel=2)
    assert torch.equal(b, c)  # hidden-static == generic, for every row-tile count
    assert_same_arithmetic(a, b, engine.kernel_class(pop, P, H, mode, rows_per_group=rows_per_group))  # ... and the call's own (maybe fused) instance


# other hidden widths, incl. the 8 and 16 column tiles (hid 113..128, 241..256) that had hidden-static instances in rounds 4-5
HIDW_CASES = [(17, 6, 500, 20, 4, dict(ensemble_size=5, hid=256)),
              (17, 6, 40, 6, 5, dict(ensemble_size=3, hid=128)),
              (11, 3, 60, 5, 6, dict(ensemble_size=5, hid=120, termination="hopper", normalizer="f32")),
              (18, 6, 48, 5, 5, dict(ensemble_size=7, hid=250, elite=[0, 2, 3, 5, 6], obs_process="halfcheetah", no_delta_list=[0])),
              (23, 7, 25, 4, 4, dict(ensemble_size=4, hid=128, reward="pusher", propagation="expectation")),
              (6, 2, 33, 5, 5, dict(ensemble_size=5, hid=241, deterministic=True, normalizer="none")),
              (17, 6, 30, 5, 4, dict(ensemble_size=5, hid=256, num_layers=2, propagation="fixed_model", learned_rewards=True, reward=None)),
              (4, 1, 64, 5, 6, dict(ensemble_size=5, hid=113, num_layers=5, reward="cartpole", termination="cartpole"))]


@pytest.mark.parametrize("mode", ["fast", "device"])
@pytest.mark.parametrize("rows_per_group", [0, 1, 4])
@pytest.mark.parametrize("case", HIDW_CASES, ids=lambda c: f"obs{c[0]}_pop{c[2]}x{c[3]}_H{c[4]}_hid{c[5]['hid']}")
def test_models_of_other_hidden_widths_run_the_generic_instance(engine, case, mode, rows_per_group):
    """SiLU models whose hidden layers are not 13 column tiles wide (the reference ships 200 everywhere, but hid_size is the first thing
    people change) run the fully generic instance since round 6 (the hidden-static instances for 8 and 16 tiles were trimmed from the
    build): `generic_kernel=2` -- "hidden-static allowed" -- and the default call return what the forced generic instance returns, for
    every row-tile count that fits the LDS."""
    obs, act, pop, P, H, mkw = case
    om, actions, s0, _, _ = _random_case(obs, act, pop, P, H, **mkw)
    engine.set_model(to_spec(om, obs, act))
    assert engine.kernel_class(pop, P, H, mode)[0] == "generic"
    try:
        a = engine.rollout(actions.to(DEV), s0, P, mode=mode, seed=11, stream_id=3, rows_per_group=rows_per_group)
    except hipets.HipetsError as exc:
        if "does not fit LDS" in str(exc):
            pytest.skip(f"R = {rows_per_group} does not fit the LDS at this width")
        raise
    b = engine.rollout(actions.to(DEV), s0, P, mode=mode, seed=11, stream_id=3, rows_per_group=rows_per_group, generic_kernel=True)
    assert torch.equal(a, b) and torch.isfinite(a).all()


# conf/overrides/pets_*.yaml as synthetic models of the same shape (conf/dynamics_model/gaussian_mlp_ensemble.yaml: 7 members / 5 elites,
# 4 x 200 SiLU): (name, obs, act, pop, H, model kwargs, instance class of a default rollout).  INTEGRATION.md section 3a is this table.
SHIPPED = [
    ("pets_halfcheetah", 18, 6, 400, 30, dict(obs_process="halfcheetah", no_delta_list=[0]), "fused"),
    ("pets_cartpole", 4, 1, 350, 15, dict(reward="cartpole", termination="cartpole"), "fused"),
    ("pets_cartpole_paper_version", 4, 1, 500, 30, dict(obs_process="cartpole_pets", reward="cartpole_pets"), "fused"),
    ("pets_mppi_halfcheetah", 18, 6, 350, 30, dict(obs_process="halfcheetah", no_delta_list=[0], learned_rewards=True, reward=None), "fused"),
    ("pets_pusher", 20, 7, 350, 25, dict(learned_rewards=True, reward=None), "fused"),
    ("pets_reacher", 17, 7, 350, 15, dict(no_delta_list=[0], learned_rewards=True, reward=None), "fused"),
    # a termination function that tests every state dim (termination_fns.py:47-55: isfinite(next_obs).all()) is fused where the four dims
    # the reward / termination lane sees are all there are ...
    ("pets_inv_pendulum", 4, 1, 480, 45, dict(learned_rewards=True, reward=None, termination="inverted_pendulum"), "fused"),
    # hopper (:12-26) tests eleven dims that sit in two column tiles, i.e. two waves: every lane judges its own, the flag goes through LDS
    # and is folded in one step later; in the persistent DEVICE form the row's NEXT owner judges the dims it receives (round 5)
    ("pets_hopper", 11, 3, 350, 30, dict(learned_rewards=True, reward=None, termination="hopper"), "fused"),
]


@pytest.mark.parametrize("mode", ["fast", "device"])
@pytest.mark.parametrize("wl", SHIPPED, ids=lambda w: w[0])
def test_shipped_workloads_run_the_instance_class_the_docs_say(engine, wl, mode):
    """hipets_kernel_class: which instance of the rollout kernel a default call runs.  Every workload the reference ships gets at
    least the hidden-static instance, and the ones without an all-dims termination function the fused one; forcing the generic
    kernel returns the same bits (the arithmetic is the same in all classes)."""
    name, obs, act, pop, H, mkw, want = wl
    om, actions, s0, _, _ = _random_case(obs, act, pop, 20, 3, ensemble_size=7, hid=200, elite=[0, 2, 3, 5, 6], **mkw)
    engine.set_model(to_spec(om, obs, act))
    cls, r = engine.kernel_class(pop, 20, H, mode)
    assert cls == (want[mode] if isinstance(want, dict) else want) and 1 <= r <= 4
    a = engine.rollout(actions.to(DEV), s0, 20, mode=mode, seed=5, stream_id=9, rows_per_group=r)
    b = engine.rollout(actions.to(DEV), s0, 20, mode=mode, seed=5, stream_id=9, rows_per_group=r, generic_kernel=True)
    assert_same_arithmetic(a, b, (cls, r))


def test_kernel_class_of_other_models(engine):
    for hid, want in ((64, "generic"), (128, "generic"), (256, "generic"), (512, "generic")):
        om, *_ = _random_case(17, 6, 8, 5, 2, ensemble_size=5, hid=hid)
        engine.set_model(to_spec(om, 17, 6))
        assert engine.kernel_class(500, 20, 30, "device")[0] == want
    om, *_ = _random_case(376, 17, 8, 5, 2, ensemble_size=7, hid=200, elite=[0, 1, 2, 3, 4], termination="humanoid")
    engine.set_model(to_spec(om, 376, 17))
    assert engine.kernel_class(1036, 20, 40, "device") == ("wide", 2)
    om, *_ = _random_case(17, 6, 8, 5, 2, ensemble_size=5, hid=200)
    engine.set_model(to_spec(om, 17, 6))
    assert engine.kernel_class(500, 20, 30, "device") == ("fused", 3)  # BASELINE.json configs[1]
    # the inverted_pendulum termination function tests EVERY state dim: fused only where the tail's lane holds them all (obs_dim <= 4)
    for obs, want in ((4, "fused"), (5, "hidden_static")):
        om, *_ = _random_case(obs, 1, 8, 5, 2, ensemble_size=5, hid=200, learned_rewards=True, reward=None, termination="inverted_pendulum")
        engine.set_model(to_spec(om, obs, 1))
        assert engine.kernel_class(480, 20, 45, "device") == (want, 3)
    with pytest.raises(Exception, match="mode must be"):
        engine.kernel_class(500, 20, 30, "exact")


@pytest.mark.parametrize("mode", ["fast", "device"])
@pytest.mark.parametrize("rows_per_group", [2, 4])
def test_cfg4_instances_for_the_decaying_icem_populations_equal_the_generic_kernel_bitwise(engine, rows_per_group, mode):
    """iCEM's population decays (cfg4: 1036, 805, 630, 497, 358 candidates) and the cost model gives the later iterations two or
    four row tiles per workgroup instead of three: the cfg4 shape has shape-specialised instances for those tile counts too."""
    obs, act, pop, P, H, mkw = SIZES[3]
    om, actions, s0, _, _ = _random_case(obs, act, 497, P, H, **mkw)
    engine.set_model(to_spec(om, obs, act))
    a = engine.rollout(actions.to(DEV), s0, P, mode=mode, seed=11, stream_id=3, rows_per_group=rows_per_group)
    b = engine.rollout(actions.to(DEV), s0, P, mode=mode, seed=11, stream_id=3, rows_per_group=rows_per_group, generic_kernel=True)
    assert torch.equal(a, b) and torch.isfinite(a).all()


WIDE_CASES = [SIZES[10], SIZES[11],
              # cfg4' at its first iCEM iteration's batch: 20 020 rows = 4 004 per member = 126 two-tile workgroups per member, 630
              # logical workgroups on 256 CUs: the turn-based persistent form
              (376, 17, 1001, 20, 2, dict(ensemble_size=7, hid=200, elite=[0, 1, 2, 3, 4], termination="humanoid"))]


@pytest.mark.parametrize("case", WIDE_CASES, ids=lambda c: f"obs{c[0]}_pop{c[2]}x{c[3]}_H{c[4]}")
def test_wide_output_instance_equals_the_generic_kernel_bitwise(engine, case):
    """cfg4' (Humanoid-v4: 752 output columns) runs a KSpec::WIDE instance: the output layer's accumulators go straight into the
    step's tail in passes of 3 column tiles, no LDS image of the outputs, hidden-width activation buffers, the 393-wide model input
    with its own row stride -- which is what lets TWO row tiles per workgroup fit the LDS (the general layout holds one).  DEVICE
    mode: the permutation decides which member a row visits, not the workgroup geometry, so the instance (R = 2) and the generic
    kernel (R = 1) must return the same bits; FAST mode's member schedule is per workgroup, so there both run R = 1."""
    obs, act, pop, P, H, mkw = case
    om, actions, s0, _, _ = _random_case(obs, act, pop, P, H, **mkw)
    engine.set_model(to_spec(om, obs, act))
    _, r = engine.fast_geometry(pop, P, H)
    _, r_general = engine.fast_geometry(pop, P, H, -1)
    assert r_general == 1 and r in (1, 2)  # (the cost model may still prefer one tile: FAST at pop 1001 runs 5 rounds of R = 1 against 3 of R = 2)
    a = engine.rollout(actions.to(DEV), s0, P, mode="device", seed=11, stream_id=3)
    b = engine.rollout(actions.to(DEV), s0, P, mode="device", seed=11, stream_id=3, generic_kernel=True)
    assert torch.equal(a, b) and torch.isfinite(a).all()
    a = engine.rollout(actions.to(DEV), s0, P, mode="fast", seed=11, stream_id=3, rows_per_group=1)
    b = engine.rollout(actions.to(DEV), s0, P, mode="fast", seed=11, stream_id=3, rows_per_group=1, generic_kernel=True)
    assert torch.equal(a, b) and torch.isfinite(a).all()
