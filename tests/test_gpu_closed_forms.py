"""Every closed-form device function the library ships, exercised ON THE GPU against the oracle (which is bitwise the reference,
tests/test_oracle_vs_reference.py CASES): termination_fns.walker2d (mbrl/env/termination_fns.py:66-74), ant (:77-85),
inverted_pendulum (:47-55), reward_fns.inverted_pendulum (mbrl/env/reward_fns.py:27-30), pusher (:41-53) and the
torch.nn.Sigmoid activation GaussianMLP accepts (mbrl/models/gaussian_mlp.py:94-95) -- plus the ones the other test files
already cover, so this file is the complete table.  Three randomness modes each:
   EXACT   the reference's draws injected (perms, eps)
   DEVICE  in-kernel keyed permutation + Philox normals, exported through the ABI and replayed through the oracle
   FAST    block-balanced member schedule + Philox normals, exported and replayed
Two entry points: whole rollouts (hipets_rollout) whose start state sits next to the termination thresholds, so that a part
of the rows terminates at every step, with NaN actions for two candidates from some step on (their rows go non-finite: the
`isfinite` branches of ant / inverted_pendulum / hopper); and single transitions (hipets_step) on batches holding NaN and
infinite rows.  Tolerances: T1 for one step, T2 for returns; a candidate with a row closer than 1e-4 to a
threshold is excluded (the comparison is discontinuous there), at most two per case."""
import numpy as np
import pytest
import torch

from conftest import to_spec
from oracle import pets_oracle as po
from oracle import device_draws

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

# name, obs, act, model kwargs, start-state overrides {dim: value} that put the rows next to the thresholds
COMBOS = [
    ("walker2d", 8, 2, dict(termination="walker2d"), {0: 0.86, 1: 0.0}),
    ("ant", 6, 2, dict(termination="ant"), {0: 0.27}),
    ("inverted_pendulum", 5, 2, dict(termination="inverted_pendulum", reward="inverted_pendulum"), {1: 0.16}),
    ("pusher", 20, 7, dict(reward="pusher"), {}),
    ("sigmoid_ant", 6, 2, dict(activation="sigmoid", termination="ant"), {0: 0.27}),
    ("sigmoid_f32norm", 17, 6, dict(activation="sigmoid", normalizer="f32", no_delta_list=[2]), {}),
    ("hopper_relu", 11, 3, dict(termination="hopper", activation="relu"), {0: 0.76, 1: 0.0}),
    ("cartpole_pets_leaky", 4, 1, dict(reward="cartpole_pets", termination="cartpole", obs_process="cartpole_pets", activation="leaky_relu"), {}),
    ("humanoid_tanh", 9, 3, dict(termination="humanoid", activation="tanh"), {0: 1.05}),
    ("cartpole", 4, 1, dict(reward="cartpole", termination="cartpole"), {0: 2.3}),
    # the all-dims termination functions INSIDE the fused tail (hid 200, SiLU, learned reward: the pets_hopper and pets_inv_pendulum
    # instances -- hopper's per-lane judgement with the flag folded in one step later (persistent DEVICE form: by the row's next owner); inverted_pendulum's four dims on one lane)
    ("hopper_learned_fused", 11, 3, dict(termination="hopper", learned_rewards=True, reward=None, hid=200), {0: 0.76, 1: 0.0}),
    ("inv_pendulum_learned_fused", 4, 1, dict(termination="inverted_pendulum", learned_rewards=True, reward=None, hid=200), {1: 0.16}),
]
FUSED_AT = {"hopper_learned_fused": (0, ("fast", "device")), "inv_pendulum_learned_fused": (3, ("fast", "device"))}  # forced row tiles, modes that run the fused instance
IDS = [c[0] for c in COMBOS]


def threshold_margin(name, nobs):
    """distance of every row of nobs [.., obs] from the nearest discontinuity of the termination function (inf: none)"""
    inf = torch.full(nobs.shape[:-1], float("inf"))
    if name == "walker2d":
        h, a = nobs[..., 0], nobs[..., 1]
        return torch.stack([(h - 0.8).abs(), (h - 2.0).abs(), (a + 1.0).abs(), (a - 1.0).abs()]).min(0).values
    if name == "ant":
        return torch.stack([(nobs[..., 0] - 0.2).abs(), (nobs[..., 0] - 1.0).abs()]).min(0).values
    if name == "inverted_pendulum":
        return (nobs[..., 1].abs() - 0.2).abs()
    if name == "hopper":
        return torch.stack([(nobs[..., 0] - 0.7).abs(), (nobs[..., 1].abs() - 0.2).abs(), (nobs[..., 1:].abs() - 100.0).abs().min(-1).values]).min(0).values
    if name == "cartpole":
        thr = 12 * 2 * np.pi / 360
        return torch.stack([(nobs[..., 0].abs() - 2.4).abs(), (nobs[..., 2].abs() - thr).abs()]).min(0).values
    if name == "humanoid":
        return torch.stack([(nobs[..., 0] - 1.0).abs(), (nobs[..., 0] - 2.0).abs()]).min(0).values
    return inf


def make(combo, seed=0):
    name, obs, act, mkw, s0_fix = combo
    mkw = dict(mkw)
    om = po.make_synthetic_model(obs, act, ensemble_size=5, hid=mkw.pop("hid", 40), seed=seed + 3, **mkw)
    s0 = (np.random.default_rng(seed).standard_normal(obs) * 0.05).astype(np.float32)
    for d, v in s0_fix.items():
        s0[d] = v
    return om, s0


def fast_members(engine, pop, P, H, seed, sid, fixed=False, rows_per_group=0):
    nwg, r = engine.fast_geometry(pop, P, H, rows_per_group)
    sched = engine.fast_schedule(H, nwg, seed, sid).cpu()
    rows = torch.arange(pop * P)
    wg = device_draws.fast_row_workgroup(rows, P, r)
    return torch.stack([sched[0 if fixed else t][wg].long() for t in range(H)])


def assert_returns_close_nan_aware(out, ref, skip):
    out, ref = out.detach().cpu(), ref.detach().cpu()
    keep = ~skip
    assert torch.equal(torch.isnan(out)[keep], torch.isnan(ref)[keep]), "NaN returns in different places"
    ok = keep & ~torch.isnan(ref)
    assert torch.isfinite(out[ok]).all()
    tol = 1e-4 * torch.clamp(ref[ok].abs(), min=1.0)  # T2
    err = (out[ok] - ref[ok]).abs()
    assert (err <= tol).all(), f"max err {err.max():.3e}"


@pytest.mark.parametrize("mode", ["exact", "device", "fast"])
@pytest.mark.parametrize("combo", COMBOS, ids=IDS)
def test_rollouts_with_terminating_and_non_finite_rows(engine, combo, mode):
    name, obs, act = combo[0], combo[1], combo[2]
    om, s0 = make(combo)
    engine.set_model(to_spec(om, obs, act))
    pop, P, H = 40, 5, 8
    B = pop * P
    g = torch.Generator().manual_seed(11)
    actions = torch.rand(pop, H, act, generator=g) * 2 - 1
    actions[3, 2:] = float("nan")  # candidate 3 from step 2 on, candidate 17 from step 5 on: their rows go non-finite
    actions[17, 5:] = float("nan")
    seed, sid = 321, 4
    rpg, fused_modes = FUSED_AT.get(name, (0, ()))
    if mode in fused_modes:  # the case exists to exercise the fused instance: make sure it is the one that runs
        cls, r = engine.kernel_class(pop, P, H, mode)
        assert cls == "fused" or rpg, (cls, r)
    if mode == "exact":
        perms = torch.stack([torch.randperm(B, generator=g) for _ in range(H)])
        eps = torch.randn(H, B, om.out_size, generator=g)
        out = engine.rollout(actions.to(DEV), s0, P, mode="exact", perms=perms.to(DEV), eps=eps.to(DEV))
        kw = dict(perms=perms, eps=eps)
    elif mode == "device":
        out = engine.rollout(actions.to(DEV), s0, P, mode="device", seed=seed, stream_id=sid, rows_per_group=rpg)
        kw = dict(perms=engine.device_perms(H, B, seed, sid).cpu(), eps=engine.fast_normals(H, B, seed, sid).cpu())
    else:
        out = engine.rollout(actions.to(DEV), s0, P, mode="fast", seed=seed, stream_id=sid, rows_per_group=rpg)
        kw = dict(members=fast_members(engine, pop, P, H, seed, sid, rows_per_group=rpg), eps=engine.fast_normals(H, B, seed, sid).cpu())
    trace = {}
    ref = po.rollout(om, actions, s0, P, trace=trace, **kw)
    nobs = torch.stack(trace["next_obs"])      # [H, B, obs]
    dones = torch.stack(trace["dones"])[..., 0]  # [H, B]
    term_name = om.termination
    if term_name != "no_termination":  # the case must exercise both branches: some rows end, some go on, at the first steps
        assert 0 < int(dones[0].sum()) < B or 0 < int(dones[1].sum()) < B, "degenerate case: no mix of terminated / alive rows"
    nonfinite_rows = ~torch.isfinite(nobs).all(-1)
    assert nonfinite_rows[2, 3 * P:(3 + 1) * P].all() and not nonfinite_rows[1].any()  # the NaN actions did their job
    if term_name in ("ant", "inverted_pendulum", "hopper"):
        assert dones[2, 3 * P:(3 + 1) * P].all()  # the isfinite branch (termination_fns.py:49, :79, :17)
    margin = threshold_margin(term_name, torch.nan_to_num(nobs, nan=1e9, posinf=1e9, neginf=-1e9))
    skip = (margin < 1e-4).any(0).view(pop, P).any(1)
    assert int(skip.sum()) <= 2, "too many candidates on a threshold: pick another seed"
    assert_returns_close_nan_aware(out, ref, skip)


@pytest.mark.parametrize("mode", ["exact", "device", "fast"])
@pytest.mark.parametrize("combo", COMBOS, ids=IDS)
def test_single_transitions_with_nan_and_infinite_rows(engine, combo, mode):
    """hipets_step (ModelEnv.step, model_env.py:87-140) on per-row states: next_obs / reward T1, done flags equal -- including
    rows that enter non-finite (NaN in one dim, +inf in another row) and rows on either side of every threshold."""
    name, obs, act = combo[0], combo[1], combo[2]
    om, s0 = make(combo, seed=1)
    engine.set_model(to_spec(om, obs, act))
    B = 120
    g = torch.Generator().manual_seed(5)
    x = torch.from_numpy(s0).repeat(B, 1) + torch.randn(B, obs, generator=g) * 0.15
    x[7, obs - 1] = float("nan")
    x[33, 1] = float("nan")
    x[50, obs - 2] = float("inf")
    a = torch.rand(B, act, generator=g) * 2 - 1
    seed, sid = 77, 9
    if mode == "exact":
        perm = torch.randperm(B, generator=g)
        eps = torch.randn(B, om.out_size, generator=g)
        got = engine.step(x.to(DEV), a.to(DEV), mode="exact", sample=True, perm=perm.to(DEV), eps=eps.to(DEV))
        ref = po.step(om, x, a, perm=perm, eps=eps, sample=True)
    elif mode == "device":
        got = engine.step(x.to(DEV), a.to(DEV), mode="device", sample=True, seed=seed, stream_id=sid)
        perm = engine.device_perms(1, B, seed, sid).cpu()[0]
        ref = po.step(om, x, a, perm=perm, eps=engine.fast_normals(1, B, seed, sid).cpu()[0], sample=True)
    else:
        got = engine.step(x.to(DEV), a.to(DEV), mode="fast", sample=True, seed=seed, stream_id=sid)
        nwg, r = engine.fast_geometry(B, 1, 1)
        sched = engine.fast_schedule(1, nwg, seed, sid).cpu()[0]
        members = sched[torch.arange(B) // (16 * r)].long()
        ref = po.step(om, x, a, member_of_row=members, eps=engine.fast_normals(1, B, seed, sid).cpu()[0], sample=True)
    nobs, rew, done = (t.cpu() for t in got)
    r_nobs, r_rew, r_done = ref
    bad_rows = ~torch.isfinite(r_nobs).all(-1)
    assert bad_rows[7] and bad_rows[33] and bad_rows[50] and int(bad_rows.sum()) == 3
    assert torch.equal(~torch.isfinite(nobs).all(-1), bad_rows)
    good = ~bad_rows
    assert torch.allclose(nobs[good], r_nobs[good], rtol=1e-5, atol=2e-6)  # T1
    near = threshold_margin(om.termination, torch.nan_to_num(r_nobs, nan=1e9, posinf=1e9, neginf=-1e9)) < 1e-4
    assert int(near.sum()) <= 2
    cmp = ~near
    assert torch.equal(done[cmp], r_done[cmp])
    if om.termination != "no_termination":
        assert 0 < int(r_done.sum()) < B  # a mix of terminated / alive rows
    if om.termination in ("ant", "inverted_pendulum", "hopper"):
        assert r_done[7] and r_done[33] and r_done[50]
    # rewards: equal where finite in the reference (T1); non-finite in the same rows
    fin = torch.isfinite(r_rew[:, 0]) & cmp
    assert torch.equal(torch.isfinite(rew[:, 0])[cmp], torch.isfinite(r_rew[:, 0])[cmp])
    assert torch.allclose(rew[fin], r_rew[fin], rtol=1e-5, atol=2e-6)
