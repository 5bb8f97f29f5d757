"""DEVICE mode: the reference's TS1 / TS-infinity propagation semantics (ONE balanced permutation of all pop x particles rows
per step, gaussian_mlp.py:203-205; iid eps, model.py:471-473) with the draws made in-kernel.  The kernel's own
permutations and normals are exported through the ABI (hipets_device_perms, hipets_fast_normals) and the rollout is
replayed through the oracle with the reference's `perms=` / `eps=` inputs -- exact arithmetic parity, tolerance T2."""
import numpy as np
import pytest
import torch

import hipets
import oracle_cache as oc
from conftest import to_spec
from oracle import feistel_perm as fp
from oracle import pets_oracle as po
from test_gpu_rollout import DEVICE_SIZES, SIZES, _random_case, assert_returns_close

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("B", [5, 500, 2500, 10000, 20720, 40000])
def test_exported_permutations_equal_the_cpu_restatement(engine, B):
    om = po.make_synthetic_model(6, 2, ensemble_size=5, hid=16, seed=0)
    engine.set_model(to_spec(om, 6, 2))
    H = 3
    perms = engine.device_perms(H, B, seed=99, stream_id=12).cpu().numpy()
    assert perms.shape == (H, B)
    for t in range(H):
        assert np.array_equal(perms[t], fp.permutation(B, 99, 12, t))
        assert np.unique(perms[t]).size == B


@pytest.mark.parametrize("case", DEVICE_SIZES, ids=lambda c: f"obs{c[0]}_pop{c[2]}x{c[3]}_H{c[4]}_hid{c[5]['hid']}")
def test_device_mode_replayed_through_oracle(engine, case):
    obs, act, pop, P, H, mkw = case
    om, actions, s0, _, _ = _random_case(obs, act, pop, P, H, **mkw)
    engine.set_model(to_spec(om, obs, act))
    seed, sid = 4321, 5
    out = engine.rollout(actions.to(DEV), s0, P, mode="device", seed=seed, stream_id=sid)
    perms = None
    if om.propagation != "expectation":
        perms = engine.device_perms(H, pop * P, seed, sid).cpu()
        assert tuple(perms.shape) == ((pop * P,) if om.propagation == "fixed_model" else (H, pop * P))
    # (eps = the library's Philox normals of (seed, stream), exported only when the oracle has to run: tests/oracle_cache.py)
    ref = oc.cached("rollout_sizes", ["device", *oc.model_parts(om), actions, s0, P, perms, ("philox", seed, sid)],
                    lambda: po.rollout(om, actions, s0, P, perms=perms, eps=None if om.deterministic else engine.fast_normals(H, pop * P, seed, sid).cpu()),
                    verify=case is DEVICE_SIZES[0])  # cfg2 at full size: never served from the memo alone
    assert_returns_close(out, ref)
    # determinism and seed sensitivity
    again = engine.rollout(actions.to(DEV), s0, P, mode="device", seed=seed, stream_id=sid)
    assert torch.equal(out, again)
    other = engine.rollout(actions.to(DEV), s0, P, mode="device", seed=seed, stream_id=sid + 1)
    if not om.deterministic:
        assert not torch.equal(out, other)


def test_device_mode_member_balance_like_the_reference(engine):
    """tests/core/test_models.py:116-152 on the exported maps: every member gets exactly B / M rows at every step, the
    assignment changes between steps (TS1) and stays fixed for fixed_model (TS-infinity)."""
    om = po.make_synthetic_model(17, 6, ensemble_size=5, hid=32, seed=1)
    spec = to_spec(om, 17, 6)
    engine.set_model(spec)
    B, H, M = 2000, 12, 5
    perms = engine.device_perms(H, B, 3, 1).cpu()
    member = torch.empty(H, B, dtype=torch.long)
    for t in range(H):
        member[t, perms[t]] = torch.arange(B) // (B // M)
        assert torch.bincount(member[t], minlength=M).tolist() == [B // M] * M
    assert (member[1:] != member[:-1]).float().mean() > 0.7  # ~ 1 - 1/M of the rows change member between steps
    om.propagation = "fixed_model"
    engine.set_model(to_spec(om, 17, 6))
    assert tuple(engine.device_perms(H, B, 3, 1).shape) == (B,)


@pytest.mark.parametrize("prop", ["random_model", "fixed_model", "expectation"])
@pytest.mark.parametrize("sample", [True, False])
def test_step_device_mode_matches_oracle(engine, prop, sample):
    obs, act, B = 17, 6, 240
    om = po.make_synthetic_model(obs, act, ensemble_size=5, hid=64, seed=5, propagation=prop)
    engine.set_model(to_spec(om, obs, act))
    g = torch.Generator().manual_seed(2)
    x = torch.randn(B, obs, generator=g) * 0.1
    a = torch.rand(B, act, generator=g) * 2 - 1
    nobs, rew, done = engine.step(x.to(DEV), a.to(DEV), mode="device", sample=sample, seed=8, stream_id=3)
    perm = None
    if prop != "expectation":
        perm = engine.device_perms(1, B, 8, 3).cpu()
        perm = perm if perm.ndim == 1 else perm[0]
    eps = engine.fast_normals(1, B, 8, 3).cpu()[0]
    r_nobs, r_rew, r_done = po.step(om, x, a, perm=perm, eps=eps, sample=sample)
    assert torch.allclose(nobs.cpu(), r_nobs, rtol=1e-5, atol=2e-6)  # T1
    assert torch.allclose(rew.cpu(), r_rew, rtol=1e-5, atol=2e-6)
    assert torch.equal(done.cpu(), r_done)


def test_eval_fn_device_mode_statistics_match_reference_order_sampling(engine):
    """mode='device' is the same estimator as the reference (global balanced permutation per step + iid eps): over
    repeated evaluations its returns have the reference's mean, deterministic under a fixed seed."""
    obs, act, pop, P, H = 17, 6, 40, 20, 8
    om = po.make_synthetic_model(obs, act, ensemble_size=5, hid=48, seed=2)
    g = torch.Generator().manual_seed(1)
    actions = torch.rand(pop, H, act, generator=g) * 2 - 1
    s0 = np.zeros(obs, np.float32)
    spec = to_spec(om, obs, act)
    outs = []
    for rep in range(2):
        fn = hipets.make_eval_fn(spec, P, engine=engine, mode="device", seed=4)
        outs.append(torch.stack([fn(s0, actions.to(DEV)) for _ in range(16)]).cpu())
    assert torch.equal(outs[0], outs[1])
    dev_runs = outs[0]
    torch.manual_seed(0)
    ref_runs = torch.stack([po.rollout(om, actions, s0, P, global_rng=True, generator=g) for _ in range(16)])
    se = torch.sqrt(dev_runs.var(0) / 16 + ref_runs.var(0) / 16)
    z = (dev_runs.mean(0) - ref_runs.mean(0)) / se
    assert z.abs().max() < 4.5 and abs(z.mean()) < 1.0


def test_device_mode_rejects_basic_ensemble_member_maps(engine):
    om = po.make_synthetic_model(17, 6, ensemble_size=3, hid=32, seed=1, ensemble_kind="basic_ensemble")
    engine.set_model(to_spec(om, 17, 6))
    a = torch.zeros(4, 3, 6, device=DEV)
    with pytest.raises(hipets.HipetsError, match="BasicEnsemble"):
        engine.rollout(a, np.zeros(17, np.float32), 2, mode="device")


# beyond the chip: more logical workgroups than CUs, served in turns by the launched ones (420 at R = 3; 1 250 one-tile workgroups
# of a small model; a ragged last turn)
BIG = [(17, 6, 1000, 20, 5, dict(hid=200)), (5, 2, 4000, 5, 4, dict(hid=32)), (17, 6, 650, 20, 3, dict(hid=200))]
# pets_hopper's fused DEVICE-mode instances (round 5): persistent form -- the row's next owner judges the dims it receives --, the same in
# turns (2 600 rows per member = 163 one-tile / 82 two-tile workgroups per member on 256 CUs), and one launch per step (the launch's own
# flag folded in before the write-back).  s0 sits next to the thresholds (_random_case): rows terminate at every step.
HOPPER = [SIZES[18], (11, 3, 650, 20, 6, dict(ensemble_size=5, hid=200, learned_rewards=True, reward=None, termination="hopper"))]


@pytest.mark.parametrize("case", [SIZES[0], SIZES[1], SIZES[5], SIZES[6], SIZES[11]] + BIG + HOPPER, ids=lambda c: f"obs{c[0]}_pop{c[2]}x{c[3]}_H{c[4]}_hid{c[5]['hid']}")
def test_persistent_and_per_step_launches_agree_bitwise(engine, case):
    """DEVICE-mode rollouts run as ONE launch with the rows handed over between workgroups through tagged granules (batches
    with more workgroups than CUs: every launched workgroup serves several logical ones per step); forbidding that (one launch
    per step, state through HBM between kernels) must not change a bit."""
    obs, act, pop, P, H, mkw = case
    om, actions, s0, _, _ = _random_case(obs, act, pop, P, H, **mkw)
    engine.set_model(to_spec(om, obs, act))
    a = engine.rollout(actions.to(DEV), s0, P, mode="device", seed=77, stream_id=9)
    engine.set_persistent(False)
    try:
        b = engine.rollout(actions.to(DEV), s0, P, mode="device", seed=77, stream_id=9)
    finally:
        engine.set_persistent(True)
    assert torch.equal(a, b)


# Every reference-semantics (DEVICE-mode) shape-specialised instance the library ships -- launch.hpp's per-R tables: (model kwargs
# that make a synthetic model of the shape, the row-tile counts it is instantiated for)
SHIPPED_DEVICE_INSTANCES = [
    ("cartpole", 4, 1, dict(reward="cartpole", termination="cartpole"), (1, 2)),
    ("cfg2_halfcheetah", 17, 6, dict(), (1, 2, 3)),
    ("pets_halfcheetah", 18, 6, dict(obs_process="halfcheetah", no_delta_list=[0]), (1, 2, 3)),
    ("learned_reward", 20, 7, dict(learned_rewards=True, reward=None), (1, 2)),
    ("learned_reward_obs_hc", 18, 6, dict(obs_process="halfcheetah", no_delta_list=[0], learned_rewards=True, reward=None), (1, 2)),
    ("hopper", 11, 3, dict(learned_rewards=True, reward=None, termination="hopper"), (1, 2)),
    ("cfg4_humanoid45", 45, 17, dict(termination="humanoid"), (2, 3, 4)),
    ("cartpole_pets", 4, 1, dict(obs_process="cartpole_pets", reward="cartpole_pets"), (3,)),
    ("inv_pendulum", 4, 1, dict(learned_rewards=True, reward=None, termination="inverted_pendulum"), (3,)),
    ("humanoid_v4_wide", 376, 17, dict(termination="humanoid"), (1, 2)),
]
_INSTANCE_CASES = [(name, obs, act, mkw, R, load) for name, obs, act, mkw, Rs in SHIPPED_DEVICE_INSTANCES for R in Rs for load in ("one_per_cu", "beyond_256")]


@pytest.mark.parametrize("name,obs,act,mkw,R,load", _INSTANCE_CASES, ids=[f"{c[0]}_R{c[4]}_{c[5]}" for c in _INSTANCE_CASES])
def test_every_shipped_device_instance_persistent_equals_per_step(engine, name, obs, act, mkw, R, load):
    """Round-5 verdict (weak 10): the hand-over's inline-asm store once had its data register rewritten one wait state later, and the
    bitwise persistent-vs-per-step comparison caught it only on the instance and occupancy where it happened to run.  Here: EVERY
    shipped DEVICE-mode instance (shape x row-tile count), with ~200 logical workgroups (one per CU) and with ~450 -- for the narrow
    R <= 2 instances that is TWO co-resident workgroups per CU (the condition under which that bug published scratch values), for
    R >= 3 and the WIDE instances the turn-based form.  One persistent launch must equal H per-step launches bit for bit."""
    P, H, M = 20, 4, 5
    pop = (160 if load == "one_per_cu" else 360) * R
    om, actions, s0, _, _ = _random_case(obs, act, pop, P, H, ensemble_size=M, hid=200, **mkw)
    engine.set_model(to_spec(om, obs, act))
    cls, r = engine.kernel_class(pop, P, H, "device", rows_per_group=R)
    assert r == R and cls == ("wide" if obs == 376 else "fused"), (cls, r)
    groups = -(-(pop * P // M) // (16 * R))
    assert (M * groups <= 256) == (load == "one_per_cu")
    a = engine.rollout(actions.to(DEV), s0, P, mode="device", seed=77, stream_id=9, rows_per_group=R)
    engine.set_persistent(False)
    try:
        b = engine.rollout(actions.to(DEV), s0, P, mode="device", seed=77, stream_id=9, rows_per_group=R)
    finally:
        engine.set_persistent(True)
    assert torch.isfinite(a).all() and torch.equal(a, b)


def ragged_last_turn(pop, P, M, grid=256):
    """rollout.hpp 'Ragged last turn', restated: (turns, row tiles behind the full two-tile turns, dealt one per workgroup?)"""
    tpd = -(-(pop * P // M) // 16)
    groups = -(-tpd // 2)
    n2 = M * groups
    turns = -(-n2 // grid)
    if turns < 2:
        return turns, 0, False
    full = (turns - 1) * grid
    d0, p0 = divmod(full, groups)
    rem = (tpd - 2 * p0) + (M - 1 - d0) * tpd
    return turns, rem, rem <= grid


# (population, (turns, row tiles behind the full turns, ragged?)) on 256 CUs with 5 elite members x 20 particles
RAGGED = [(497, (2, 117, True)),    # cfg4' iCEM, fourth iteration: the one-tile range starts inside the last member's domain
          (505, (2, 127, True)),    # ... exactly on a member boundary
          (520, (2, 138, True)),    # ... inside member 3 and runs on through member 4
          (609, (2, 256, True)),    # one tile for every launched workgroup: the limit
          (613, (2, 258, False)),   # two tiles too many: two-tile turns throughout
          (1001, (3, 235, True)),   # two full turns first
          (1017, (3, 255, True))]


@pytest.mark.parametrize("pop,expect", RAGGED, ids=[f"pop{p}" for p, _ in RAGGED])
def test_ragged_last_turn_of_the_wide_instance_equals_per_step_launches(engine, pop, expect):
    """Round 6: the WIDE two-tile instance deals the LAST turn of a step in one-tile logical workgroups when the row tiles left for it
    fit one per launched workgroup, and runs the R = 1 bodies of the MLP ops for that turn (rollout.hpp 'Ragged last turn').  Which
    workgroup holds a row never enters the arithmetic: one persistent launch must equal H per-step launches (whose geometry knows no
    turns) bit for bit -- at sizes on both sides of the limit, with the one-tile range starting inside a member domain and on a
    member boundary."""
    obs, act, P, H, M = 376, 17, 20, 3, 5
    if torch.cuda.get_device_properties(0).multi_processor_count == 256:
        assert ragged_last_turn(pop, P, M) == expect
    om, actions, s0, _, _ = _random_case(obs, act, pop, P, H, ensemble_size=7, hid=200, elite=list(range(M)), termination="humanoid")
    engine.set_model(to_spec(om, obs, act))
    assert engine.kernel_class(pop, P, H, "device") == ("wide", 2)
    a = engine.rollout(actions.to(DEV), s0, P, mode="device", seed=5, stream_id=3)
    engine.set_persistent(False)
    try:
        b = engine.rollout(actions.to(DEV), s0, P, mode="device", seed=5, stream_id=3)
    finally:
        engine.set_persistent(True)
    assert torch.isfinite(a).all() and torch.equal(a, b)


def test_error_behaviour_of_the_round2_entry_points(engine):
    """Plan mode / trace / batched-plan misuse fails with a message, never silently: non-zero C return -> HipetsError."""
    from hipets.planning import _BoundObjective

    obs, act, H, P, pop = 17, 6, 5, 5, 40
    om = po.make_synthetic_model(obs, act, ensemble_size=5, hid=32, seed=1)
    spec = to_spec(om, obs, act)
    engine.set_model(spec)
    with pytest.raises(ValueError, match="plan mode"):
        engine.set_plan_mode("exact")
    with pytest.raises(ValueError, match="perms / eps / members must be None"):
        engine.rollout(torch.zeros(pop, H, act, device=DEV), np.zeros(obs, np.float32), P, mode="device",
                       eps=torch.zeros(H, pop * P, obs, device=DEV))
    with pytest.raises(hipets.HipetsError, match="multiple of the number of models"):  # gaussian_mlp.py:195-200 in DEVICE mode too
        engine.rollout(torch.zeros(3, H, act, device=DEV), np.zeros(obs, np.float32), 4, mode="device")
    # a trace too small for the plan is reported by the plan call
    fn = hipets.make_eval_fn(spec, P, engine=engine, seed=1, mode="device")
    lb, ub = [[-1.0] * act] * H, [[1.0] * act] * H
    opt = hipets.CEMOptimizer(2, 0.1, pop, lb, ub, 0.1, DEV, return_mean_elites=True, seed=1)
    engine.set_plan_trace(2, pop - 1, H, act, int(opt.elite_num))
    try:
        with pytest.raises(hipets.HipetsError, match="plan trace"):
            opt.optimize(_BoundObjective(fn, np.zeros(obs, np.float32)), x0=torch.zeros(H, act))
    finally:
        engine.set_plan_trace(0)
    # batched plans run either in-kernel randomness mode since round 6; an objective that draws on the host ('exact') is refused
    with pytest.raises(ValueError, match="in-kernel randomness"):
        hipets.BatchedMPPIAgent(hipets.make_eval_fn(to_spec(om, obs, act), P, engine=engine, mode="exact"), 2, [-1.0] * act, [1.0] * act, H, 2, pop, 0.9, 1.0, 0.9)
    hipets.BatchedMPPIAgent(fn, 2, [-1.0] * act, [1.0] * act, H, 2, pop, 0.9, 1.0, 0.9)  # a DEVICE-mode objective: accepted
    # the persistent switch is reversible and DEVICE rollouts work either way (covered bit for bit elsewhere)
    engine.set_persistent(False)
    engine.set_persistent(True)


def test_fast_mode_statistics_match_reference_semantics_at_cfg2_size(engine):
    """SURVEY 8c T5 at BASELINE size: FAST mode's block-balanced member schedule against DEVICE mode (the reference's global
    balanced permutation per step) on the cfg2 batch (pop 500 x 20 particles x H 30), 32 seeds each.  Per candidate the two
    estimators of the expected return agree within 3 sigma-equivalents (max |z| over 500 candidates < 4.5, mean z ~ 0), and
    so do the across-seed variances (ratio within the F-distribution's range for 31 / 31 degrees of freedom)."""
    obs, act, pop, P, H, seeds = 17, 6, 500, 20, 30, 32
    om = po.make_synthetic_model(obs, act, ensemble_size=5, hid=200, seed=0)
    engine.set_model(to_spec(om, obs, act))
    g = torch.Generator().manual_seed(3)
    actions = (torch.rand(pop, H, act, generator=g) * 2 - 1).to(DEV)
    s0 = (np.random.default_rng(0).standard_normal(obs) * 0.1).astype(np.float32)
    runs = {}
    for mode in ("fast", "device"):
        runs[mode] = torch.stack([engine.rollout(actions, s0, P, mode=mode, seed=100 + i, stream_id=i) for i in range(seeds)]).double().cpu()
    mf, md_ = runs["fast"].mean(0), runs["device"].mean(0)
    vf, vd = runs["fast"].var(0), runs["device"].var(0)
    z = (mf - md_) / torch.sqrt(vf / seeds + vd / seeds)
    assert z.abs().max() < 4.5, float(z.abs().max())
    assert abs(float(z.mean())) < 0.25
    ratio = (vf / vd)
    assert 0.2 < float(ratio.median()) < 5.0 and float(ratio.log().mean().abs()) < 0.35  # no systematic variance inflation / deflation
    # balance of the FAST schedule per step, like tests/core/test_models.py:116-131
    nwg, _ = engine.fast_geometry(pop, P, H)
    sched = engine.fast_schedule(H, nwg, 100, 0).cpu()
    for t in range(H):
        c = torch.bincount(sched[t].long(), minlength=5)
        assert c.max() - c.min() <= 1



def test_timed_out_persistent_rollout_is_reported_and_the_plan_is_rerun(engine):
    """SURVEY.md 8(b) "never silently approximate": a persistent DEVICE-mode rollout whose hand-over polls give up (here forced:
    a zero bound, so the first poll that does not find its rows at once quits -- in production: CUs taken by another process)
    leaves invalid returns.  (1) hipets_check_async_error says so once the results are on the host and the engine falls back to
    per-step launches; (2) the agent never hands such a plan out: TrajectoryOptimizer.optimize asks after its device-to-host
    copy, puts the optimizer's counters / persistent state back and runs the SAME plan again -- the action equals, bit for bit,
    the one an engine that never used the persistent form returns."""
    obs, act, pop, P, H = 17, 6, 500, 20, 12
    om = po.make_synthetic_model(obs, act, ensemble_size=5, hid=200, seed=0)
    spec = to_spec(om, obs, act)
    engine.set_model(spec)
    g = torch.Generator().manual_seed(0)
    actions = (torch.rand(pop, H, act, generator=g) * 2 - 1).to(DEV)
    s0 = (np.random.default_rng(0).standard_normal(obs) * 0.1).astype(np.float32)

    def agent():
        cfg = dict(_target_="hipets.MPPIOptimizer", num_iterations=3, population_size=pop, gamma=0.9, sigma=1.0, beta=0.9, device=DEV,
                   lower_bound="???", upper_bound="???", seed=5)
        ag = hipets.TrajectoryOptimizerAgent(cfg, [-1.0] * act, [1.0] * act, planning_horizon=H)
        ag.set_trajectory_eval_fn(hipets.make_eval_fn(spec, P, engine=engine, seed=3, mode="device"))
        return ag

    try:
        engine.set_persistent(False)
        ref_rollout = engine.rollout(actions, s0, P, mode="device", seed=7, stream_id=1).clone()
        ref_agent = agent()
        ref_actions = [ref_agent.act(s0).copy() for _ in range(2)]  # two plans: MPPI's mean persists from one to the next
        assert not engine.check_async_error()
        # (1) the raw entry point
        engine.set_persistent(True)
        engine.set_handover_timeout(0.0)
        bad = engine.rollout(actions, s0, P, mode="device", seed=7, stream_id=1)
        torch.cuda.synchronize()
        assert engine.check_async_error(), "a zero poll bound did not trip at cfg2 size: the test does not exercise the time-out path"
        assert not engine.check_async_error()  # reported once; persistent launches are off now
        again = engine.rollout(actions, s0, P, mode="device", seed=7, stream_id=1)
        assert torch.equal(again, ref_rollout)
        del bad
        # (2) the agent
        engine.set_persistent(True)
        ag = agent()
        got = [ag.act(s0).copy() for _ in range(2)]
        assert np.array_equal(got[0], ref_actions[0]) and np.array_equal(got[1], ref_actions[1])
        assert not engine.check_async_error()
    finally:
        engine.set_handover_timeout(0.2)
        engine.set_persistent(True)
    ok = engine.rollout(actions, s0, P, mode="device", seed=7, stream_id=1)  # the persistent form works again afterwards
    assert torch.equal(ok, ref_rollout) and not engine.check_async_error()


def test_timed_out_lds_dma_collect_of_the_wide_instance_is_reported(engine):
    """The WIDE instances (Humanoid-v4's 376-dim state) collect a turn's rows by LDS-DMA (rollout.hpp dma_collect, round 6): a chunk with a
    stale lane is fetched again until the poll bound.  With a zero bound the first miss gives up: the flag must trip (a step's first turn
    always finds some row unpublished at this size), nothing may hang or fault (no DMA may still be landing in the activation buffers when
    the flow goes on), and the per-step launches the engine falls back to return what an engine that never used the persistent form does."""
    obs, act, pop, P, H = 376, 17, 660, 20, 3
    om, actions, s0, _, _ = _random_case(obs, act, pop, P, H, ensemble_size=5, hid=200, termination="humanoid")
    engine.set_model(to_spec(om, obs, act))
    assert engine.kernel_class(pop, P, H, "device") == ("wide", 2)
    a = actions.to(DEV)
    try:
        engine.set_persistent(False)
        ref = engine.rollout(a, s0, P, mode="device", seed=7, stream_id=1).clone()
        engine.set_persistent(True)
        ok = engine.rollout(a, s0, P, mode="device", seed=7, stream_id=1)
        assert torch.equal(ok, ref) and not engine.check_async_error()
        engine.set_handover_timeout(0.0)
        bad = engine.rollout(a, s0, P, mode="device", seed=7, stream_id=1)
        torch.cuda.synchronize()
        assert engine.check_async_error(), "a zero poll bound did not trip: the test does not exercise dma_collect's give-up path"
        del bad
        again = engine.rollout(a, s0, P, mode="device", seed=7, stream_id=1)  # per-step launches now
        assert torch.equal(again, ref)
    finally:
        engine.set_handover_timeout(0.2)
        engine.set_persistent(True)
    assert torch.equal(engine.rollout(a, s0, P, mode="device", seed=7, stream_id=1), ref) and not engine.check_async_error()
