"""precision='bf16x3' (a SEPARATELY reported arithmetic mode, include/hipets.h HIPETS_PREC_BF16X3): the ensemble MLP's linear layers
with fp32 operands carried as three bf16 pieces and six exact partial products per product on v_mfma_f32_16x16x32_bf16, fp32
accumulate.  Its own parity evidence, against the SAME oracle and the SAME tolerances as the fp32-MFMA mode (T2: |err| <= 1e-4
max(1, |v|); one step: rtol 1e-5, atol 2e-6), plus the measured error against fp32 MFMA results."""
import numpy as np
import pytest
import torch

import hipets
from conftest import to_spec
from hipets.planning import _BoundObjective
from oracle import pets_oracle as po
from oracle import device_draws
from test_gpu_rollout import SIZES, _random_case, assert_returns_close

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
# (cfg1's shape -- one-tile workgroups only -- lost its bf16x3 instances in round 6's trim of the build: the mode runs R = 3 instances)
B3_CASES = [SIZES[0], SIZES[3], SIZES[4], (17, 6, 7, 5, 4, dict(ensemble_size=5, hid=200))]


@pytest.mark.parametrize("mode", ["fast", "device"])
@pytest.mark.parametrize("case", B3_CASES, ids=lambda c: f"obs{c[0]}_pop{c[2]}x{c[3]}_H{c[4]}")
def test_bf16x3_rollouts_replayed_through_the_oracle(engine, case, mode):
    obs, act, pop, P, H, mkw = case
    om, actions, s0, _, _ = _random_case(obs, act, pop, P, H, **mkw)
    engine.set_model(to_spec(om, obs, act, precision="bf16x3"))
    seed, sid = 31, 4
    out = engine.rollout(actions.to(DEV), s0, P, mode=mode, seed=seed, stream_id=sid)
    eps = engine.fast_normals(H, pop * P, seed, sid).cpu()
    nwg, r = engine.fast_geometry(pop, P, H)  # (of the bf16x3 model: the row-tile counts that have a bf16x3 instance)
    tr = {}
    if mode == "device":
        ref = po.rollout(om, actions, s0, P, perms=engine.device_perms(H, pop * P, seed, sid).cpu(), eps=eps, trace=tr)
    else:
        sched = engine.fast_schedule(H, nwg, seed, sid).cpu()
        rows = torch.arange(pop * P)
        wg = device_draws.fast_row_workgroup(rows, P, r)
        ref = po.rollout(om, actions, s0, P, members=torch.stack([sched[t][wg].long() for t in range(H)]), eps=eps, trace=tr)
    keep = torch.ones(pop, dtype=torch.bool)
    if om.termination == "humanoid":
        # a termination threshold is a discontinuity: a row whose height lands within an ulp of 1.0 / 2.0 (termination_fns.py:88-95) ends
        # on one side in the oracle and may end on the other on the device.  Round 5's FAST row dealing put such a row into this case
        # (candidate 567, step 1: z = 2.0000002, re-derived on the CPU from oracle/device_draws.py); candidates with a row within 2e-5 of a
        # threshold are not compared -- a handful of 1036 (20 720 rows x 4 steps over a range of ~4)
        z = torch.stack([n[:, 0] for n in tr["next_obs"]])
        keep = ~(torch.minimum((z - 1.0).abs(), (z - 2.0).abs()) < 2e-5).any(0).view(pop, P).any(1)
        assert int((~keep).sum()) <= 8
    assert_returns_close(out.cpu()[keep], ref[keep])  # T2, the fp32 mode's own tolerance
    # against the fp32-MFMA kernel on the same draws (FAST: and the same geometry -- the member schedule is per workgroup, and the fp32
    # model is free to pick another row-tile count): the two arithmetic modes agree far inside T2
    engine.set_model(to_spec(om, obs, act))
    f32 = engine.rollout(actions.to(DEV), s0, P, mode=mode, seed=seed, stream_id=sid, rows_per_group=r if mode == "fast" else 0)
    err = ((out - f32).abs() / torch.clamp(f32.abs(), min=1.0)).cpu()[keep].max().item()
    assert err < 2e-5, err


def test_bf16x3_one_step_accuracy(engine):
    """One transition (H = 1: the return is the step's reward s'[0] - 0.1 |a|^2, i.e. one MLP evaluation + sampling) at the one-step
    tolerance T1 (rtol 1e-5, atol 2e-6)."""
    obs, act, pop, P = 17, 6, 500, 20
    om, actions, s0, _, _ = _random_case(obs, act, pop, P, 1, ensemble_size=5, hid=200)
    engine.set_model(to_spec(om, obs, act, precision="bf16x3"))
    out = engine.rollout(actions.to(DEV), s0, P, mode="device", seed=3, stream_id=1).cpu()
    ref = po.rollout(om, actions, s0, P, perms=engine.device_perms(1, pop * P, 3, 1).cpu(), eps=engine.fast_normals(1, pop * P, 3, 1).cpu())
    assert torch.allclose(out, ref, rtol=1e-5, atol=2e-6), float((out - ref).abs().max())


def test_bf16x3_fused_plan_equals_per_iteration_path_and_needs_a_specialised_shape(engine):
    obs, act, H, P, pop = 17, 6, 10, 5, 120
    om = po.make_synthetic_model(obs, act, ensemble_size=5, hid=200, seed=3)
    fn = hipets.make_eval_fn(to_spec(om, obs, act, precision="bf16x3"), P, engine=engine, seed=13, mode="device")
    obj = _BoundObjective(fn, (np.random.default_rng(1).standard_normal(obs) * 0.1).astype(np.float32))
    lb, ub = [[-1.0] * act] * H, [[1.0] * act] * H
    a = hipets.CEMOptimizer(4, 0.1, pop, lb, ub, 0.1, DEV, return_mean_elites=True, seed=21)
    b = hipets.CEMOptimizer(4, 0.1, pop, lb, ub, 0.1, DEV, return_mean_elites=True, seed=21)
    x0 = torch.zeros(H, act)
    assert torch.equal(a.optimize(obj, x0=x0), b.optimize(obj, x0=x0, callback=lambda *_: None))
    # no instance for other shapes / calls: fails loudly, never falls back to another arithmetic
    om2 = po.make_synthetic_model(obs, act, ensemble_size=5, hid=64, seed=3)
    engine.set_model(to_spec(om2, obs, act, precision="bf16x3"))
    with pytest.raises(hipets.HipetsError, match="bf16x3"):
        engine.rollout(torch.zeros(40, 3, act, device=DEV), np.zeros(obs, np.float32), 5, mode="device")
    engine.set_model(to_spec(om, obs, act, precision="bf16x3"))
    with pytest.raises(hipets.HipetsError, match="bf16x3"):  # injected eps need the generic kernel
        engine.rollout(torch.zeros(40, 3, act, device=DEV), np.zeros(obs, np.float32), 5, mode="exact",
                       perms=torch.stack([torch.randperm(200) for _ in range(3)]).to(DEV), eps=torch.zeros(3, 200, obs, device=DEV))
