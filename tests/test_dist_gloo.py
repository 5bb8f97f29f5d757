"""The N>1 path on CPU: world_size 2, gloo backend.  Exercises the real sharding + all-gather code
(hipets.dist) with a CPU objective; the rollout kernel itself is covered by the -m gpu tests."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _objective(initial_state, action_sequences):
    # any deterministic per-candidate function: candidates are independent, like rollouts
    return -(action_sequences ** 2).sum(dim=(1, 2)) + float(initial_state[0])


def _worker(rank, world, port, pop, tmpdir):
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "mbrl-lib_amd"))
    from hipets import dist as hdist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(0)  # replicated sampling: same seed on every rank
        population = torch.rand(pop, 4, 3, generator=g)
        s0 = np.array([0.5], np.float32)
        calls = []

        def counting(initial_state, seqs):
            calls.append(seqs.shape[0])
            return _objective(initial_state, seqs)

        fn = hdist.ShardedEvalFn(counting)
        values = fn(s0, population)
        lo, hi = hdist.shard_bounds(pop, world, rank)
        assert calls == [hi - lo]
        assert torch.equal(values, _objective(s0, population))  # identical on every rank
        # elite statistics computed from the gathered values are replicated: compare across ranks
        idx = values.topk(3).indices
        mu = population[idx].mean(0)
        gathered = [torch.empty_like(mu) for _ in range(world)]
        dist.all_gather(gathered, mu)
        assert all(torch.equal(gathered[0], t) for t in gathered)
        assert hdist.is_distributed()
        torch.save(values, os.path.join(tmpdir, f"v{rank}.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("pop", [10, 7, 501])
def test_sharded_eval_world2_gloo(tmp_path, pop):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), pop, str(tmp_path)), nprocs=world, join=True)
    v0 = torch.load(tmp_path / "v0.pt")
    v1 = torch.load(tmp_path / "v1.pt")
    assert torch.equal(v0, v1) and v0.shape[0] == pop


def test_sharded_plan_falls_back_to_a_single_gpu_plan_when_rccl_fails():
    """SURVEY.md section 5 "failure detection": an RCCL error inside the sharded plan -> warn, drop the communicator, plan on
    this GPU alone; errors that are not communication errors still propagate."""
    import hipets
    from hipets import dist as hdist

    class FakeEngine:
        comm_world, comm_rank = 2, 1
        destroyed = False

        def __init__(self, message):
            self.message = message

        def plan_cem_sharded(self, *a, **k):
            raise hipets.HipetsError(self.message)

        def plan_cem(self, params, x0, lower, upper, s0, P, seed=0, plan_id=0):
            return ("single-gpu plan", seed, plan_id)

        def comm_destroy(self):
            self.destroyed = True
            self.comm_world, self.comm_rank = 1, 0

        def synchronize(self):
            pass

        def check_async_error(self):
            return False

    eng = FakeEngine("RCCL error 5 (unhandled system error) at hipets.hip:1200")
    with pytest.warns(UserWarning, match="falling back to a single-GPU plan"):
        plan, fell_back = hdist.plan_cem_sharded(eng, None, None, None, None, None, 20, seed=3, plan_id=9)
    assert plan == ("single-gpu plan", 3, 9) and fell_back and eng.destroyed and eng.comm_world == 1
    plan, fell_back = hdist.plan_cem_sharded(eng, None, None, None, None, None, 20, seed=3, plan_id=10)  # no communicator any more
    assert plan[0] == "single-gpu plan" and fell_back
    with pytest.raises(hipets.HipetsError, match="act_dim"):
        hdist.plan_cem_sharded(FakeEngine("act_dim 5 != model act_dim 6"), None, None, None, None, None, 20)

    class TimedOutEngine(FakeEngine):  # the plan was enqueued and "ran", but a persistent rollout inside it gave up
        def plan_cem_sharded(self, *a, **k):
            return "sharded plan built on invalid returns"

        def check_async_error(self):
            return True

    eng = TimedOutEngine("")
    with pytest.warns(UserWarning, match="timed out"):
        plan, fell_back = hdist.plan_cem_sharded(eng, None, None, None, None, None, 20, seed=1, plan_id=2)
    assert plan == ("single-gpu plan", 1, 2) and fell_back and eng.destroyed
