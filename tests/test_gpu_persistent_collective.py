"""The combination a real multi-GPU node runs and no earlier test did (round-4 verdict, missing #1): the PERSISTENT DEVICE-mode
rollout (one launch per rollout, rows handed over between workgroups through HBM) followed by the collective of the sharded plan
on the same stream, iteration after iteration.  Every other sharded test sets HIPETS_NO_PERSISTENT=1 because its ranks share
one GPU without leaving each other room.

  * one-rank REAL RCCL communicator (all a 1-GPU box can hold), persistent form ON: hipets_plan_{cem,mppi,icem}_sharded over 25
    rollouts + ncclAllGather each, bit for bit the unsharded fused plan, and the launch count proves the persistent form ran;
  * world 2 on tests/fake_rccl with every rank's persistent grid capped (HIPETS_MAX_WORKGROUPS) so that both ranks' grids are
    co-resident on the one GPU: the 63-tile-per-member shard is served in turns, the gathered values are the ranks' own shard
    rollouts (recomputed with per-step launches: the two launch forms return the same bits), nobody timed out.
Reference semantics: mbrl/planning/trajectory_opt.py:142-188 (CEM), :238-311 (MPPI), :391-487 (iCEM) over
mbrl/models/model_env.py:145-191."""
import os
import sys
import time

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = 0x9E3779B97F4A7C15
OBS, ACT, H, P = 17, 6, 10, 20


def _setup_paths():
    for p in (ROOT, os.path.join(ROOT, "mbrl-lib_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)


def _model():
    from conftest import to_spec
    from oracle import pets_oracle as po

    return to_spec(po.make_synthetic_model(OBS, ACT, ensemble_size=5, hid=200, seed=4), OBS, ACT)  # cfg2's shape: the fused instances


def _launches(engine, fn):
    engine.timing_enable(1)
    engine.timing_read(reset=True)
    out = fn()
    torch.cuda.synchronize()
    n, _ = engine.timing_read(reset=True)
    engine.timing_enable(False)
    return out, n


@pytest.mark.parametrize("kind", ["cem", "mppi", "icem"])
def test_persistent_device_rollouts_with_a_real_one_rank_communicator_equal_the_unsharded_plan(engine, kind):
    import hipets
    from hipets._lib import IcemParams

    engine.set_model(_model())
    engine.set_plan_mode("device")
    engine.set_persistent(True)
    lower, upper = -torch.ones(H, ACT, device=DEV), torch.ones(H, ACT, device=DEV)
    x0 = torch.zeros(H, ACT, device=DEV)
    s0 = (np.random.default_rng(0).standard_normal(OBS) * 0.1).astype(np.float32)
    iters, pop = 25, 200
    engine.comm_init(engine.comm_unique_id(), 0, 1)
    try:
        if kind == "cem":
            p = hipets.Engine.cem_params(pop, H, ACT, iters, 20, 0.1, True, False, True)
            a, n = _launches(engine, lambda: engine.plan_cem_sharded(p, x0, lower, upper, s0, P, seed=3, plan_id=1).clone())
            b = engine.plan_cem(p, x0, lower, upper, s0, P, seed=3, plan_id=1)
        elif kind == "mppi":
            ma, mb = torch.zeros(H, ACT, device=DEV), torch.zeros(H, ACT, device=DEV)
            a, n = _launches(engine, lambda: engine.plan_mppi_sharded(pop, H, ACT, iters, 0.9, 0.9, ma, lower, upper, s0, P, seed=3, plan_id=1).clone())
            b = engine.plan_mppi(pop, H, ACT, iters, 0.9, 0.9, mb, lower, upper, s0, P, seed=3, plan_id=1)
        else:
            K, keep = 20, 6
            p = IcemParams(population_size=pop, horizon=H, act_dim=ACT, num_iterations=iters, elite_num=K, keep_elite_size=keep, population_size_module=5,
                           return_mean_elites=1, alpha=0.1, population_decay_factor=1.05, colored_noise_exponent=2.0)
            ea, eb = torch.empty(K, H, ACT, device=DEV), torch.empty(K, H, ACT, device=DEV)
            a, n = _launches(engine, lambda: engine.plan_icem_sharded(p, x0, lower, upper, ea, False, s0, P, seed=3, plan_id=1).clone())
            b = engine.plan_icem(p, x0, lower, upper, eb, False, s0, P, seed=3, plan_id=1)
            assert torch.equal(ea, eb)
        torch.cuda.synchronize()
        assert not engine.check_async_error()
        assert n == iters, f"{n} rollout-kernel launches for {iters} iterations: the persistent form (one launch per rollout) did not run"
        assert torch.isfinite(a).all() and torch.equal(a, b)
    finally:
        engine.comm_destroy()
        engine.set_plan_mode("fast")  # the session engine's default


def _worker(rank, world, cap, tmpdir):
    _setup_paths()
    os.environ["HIPETS_MAX_WORKGROUPS"] = str(cap)  # both ranks' persistent grids fit the one GPU at once
    os.environ.pop("HIPETS_NO_PERSISTENT", None)
    import hipets
    from hipets import dist as hdist

    eng = hipets.get_engine(DEV)
    eng.set_model(_model())
    eng.set_plan_mode("device")
    uid_path = os.path.join(tmpdir, "uid.bin")
    if rank == 0:
        with open(uid_path + ".tmp", "wb") as f:
            f.write(eng.comm_unique_id())
        os.rename(uid_path + ".tmp", uid_path)
    t0 = time.time()
    while not os.path.exists(uid_path):
        time.sleep(0.01)
        assert time.time() - t0 < 120
    eng.comm_init(open(uid_path, "rb").read(), rank, world)
    pop, iters, K = 500, 5, 50
    p = hipets.Engine.cem_params(pop, H, ACT, iters, K, 0.1, True, False, True)
    lower, upper = -torch.ones(H, ACT, device=DEV), torch.ones(H, ACT, device=DEV)
    x0 = torch.zeros(H, ACT, device=DEV)
    s0 = (np.random.default_rng(0).standard_normal(OBS) * 0.1).astype(np.float32)
    tr = eng.set_plan_trace(iters, pop, H, ACT, K)
    res = {}
    try:
        eng.timing_enable(1)
        eng.timing_read(reset=True)
        plan = eng.plan_cem_sharded(p, x0, lower, upper, s0, P, seed=5, plan_id=2)
        torch.cuda.synchronize()
        res["launches"] = eng.timing_read(reset=True)[0]
        eng.timing_enable(False)
        res["timed_out"] = bool(eng.check_async_error())
        res.update(plan=plan.cpu(), populations=tr["populations"].cpu(), values=tr["values"].cpu(), mus=tr["mus"].cpu())
        lo, hi = hdist.shard_bounds(pop, world, rank)
        eng.set_persistent(False)  # the shard again with one launch per step: the two launch forms return the same bits
        res["mine"] = torch.stack([eng.rollout(tr["populations"][i][lo:hi].contiguous(), s0, P, mode="device", seed=(5 + rank * GOLD) & (2**64 - 1),
                                               stream_id=2 * iters + i).cpu() for i in range(iters)])
        res["bounds"] = (lo, hi)
    except hipets.HipetsError as exc:
        res["error"] = str(exc)
    finally:
        eng.set_plan_trace(0)
    torch.save(res, os.path.join(tmpdir, f"r{rank}.pt"))


def test_two_ranks_with_capped_persistent_grids_on_one_gpu(tmp_path):
    _setup_paths()
    import __graft_entry__ as ge

    world, cap = 2, 100  # 105 logical workgroups per rank (R = 3: 21 per member) on 100 launched: turns; 2 x 100 <= 256 CUs
    os.environ["HIPETS_RCCL_LIB"] = ge.build_fake_rccl()
    try:
        mp.spawn(_worker, args=(world, cap, str(tmp_path)), nprocs=world, join=True)
    finally:
        del os.environ["HIPETS_RCCL_LIB"]
    res = [torch.load(tmp_path / f"r{r}.pt") for r in range(world)]
    for r, a in enumerate(res):
        assert "error" not in a, a.get("error")
        assert not a["timed_out"], f"rank {r}: a persistent rollout timed out"
        assert a["launches"] == 5, f"rank {r}: {a['launches']} rollout launches for 5 iterations (persistent form: one each)"
        for k in ("plan", "populations", "values", "mus"):
            assert torch.equal(a[k], res[0][k]), (r, k)
    for i in range(5):  # gathered values = the ranks' own shard rollouts, in candidate order
        assert torch.equal(res[0]["values"][i][:500], torch.cat([x["mine"][i] for x in res])), i
    assert torch.isfinite(res[0]["plan"]).all()
