"""The N>1 path end to end on ONE GPU: two ranks (gloo, both on cuda:0) shard the population, run the real rollout
kernel on their shard, all-gather the returns and refit.  (The driver's 8-GPU run uses the same code with nccl.)"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

import hipets  # noqa: E402

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, tmpdir):
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "mbrl-lib_amd"), os.path.join(root, "tests")):
        sys.path.insert(0, p)
    import torch.distributed as dist

    import hipets
    from conftest import to_spec
    from hipets.planning import _BoundObjective
    from oracle import pets_oracle as po

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        obs, act, H, P, pop = 17, 6, 8, 10, 101  # uneven shards: 51 + 50
        om = po.make_synthetic_model(obs, act, ensemble_size=5, hid=64, seed=4)
        om.max_logvar = torch.full_like(om.max_logvar, -8.0)
        fn = hipets.dist.ShardedEvalFn(hipets.make_eval_fn(to_spec(om, obs, act), P, seed=1))
        opt = hipets.CEMOptimizer(4, 0.1, pop, [[-1.0] * act] * H, [[1.0] * act] * H, 0.1, "cuda:0", return_mean_elites=True, seed=7)
        s0 = np.zeros(obs, np.float32)
        seen = []
        plan = opt.optimize(_BoundObjective(fn, s0), x0=torch.zeros(H, act), callback=lambda p_, v, i: seen.append((p_.cpu().clone(), v.cpu().clone())))
        torch.cuda.synchronize()
        torch.save(dict(plan=plan.cpu(), pops=[s_[0] for s_ in seen], vals=[s_[1] for s_ in seen]), os.path.join(tmpdir, f"r{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_two_ranks_one_gpu_sharded_cem(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    a, b = torch.load(tmp_path / "r0.pt"), torch.load(tmp_path / "r1.pt")
    assert torch.equal(a["plan"], b["plan"]) and torch.isfinite(a["plan"]).all()  # replicated refit, no broadcast
    for pa, pb, va, vb in zip(a["pops"], b["pops"], a["vals"], b["vals"]):
        assert torch.equal(pa, pb)  # replicated sampling
        assert torch.equal(va, vb) and va.shape[0] == 101  # every rank holds all gathered returns


DEV = "cuda:0"


def test_in_library_rccl_communicator_world_one_equals_fused_plan(engine):
    """hipets_comm_unique_id / hipets_comm_init / hipets_plan_cem_sharded with a one-rank communicator (all a 1-GPU box can
    hold): librccl is loaded lazily, the communicator comes up, and the sharded plan is the fused plan bit for bit."""
    import numpy as np

    from conftest import to_spec
    from oracle import pets_oracle as po

    obs, act, H, P, pop = 17, 6, 8, 5, 60
    om = po.make_synthetic_model(obs, act, ensemble_size=5, hid=48, seed=2)
    engine.set_model(to_spec(om, obs, act))
    p = hipets.Engine.cem_params(pop, H, act, 3, 6, 0.1, True, False, True)
    lower, upper = -torch.ones(H, act, device=DEV), torch.ones(H, act, device=DEV)
    x0 = torch.zeros(H, act, device=DEV)
    s0 = (np.random.default_rng(0).standard_normal(obs) * 0.1).astype(np.float32)
    with pytest.raises(hipets.HipetsError, match="no communicator"):
        engine.plan_cem_sharded(p, x0, lower, upper, s0, P, seed=3, plan_id=1)
    uid = engine.comm_unique_id()
    assert len(uid) == 128
    engine.comm_init(uid, 0, 1)
    try:
        a = engine.plan_cem_sharded(p, x0, lower, upper, s0, P, seed=3, plan_id=1)
        b = engine.plan_cem(p, x0, lower, upper, s0, P, seed=3, plan_id=1)
        assert torch.equal(a, b) and torch.isfinite(a).all()
    finally:
        engine.comm_destroy()
