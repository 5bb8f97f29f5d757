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

        def __init__(self, message, kind=hipets.ERR_RUNTIME):
            self.message, self.kind = message, kind

        def plan_cem_sharded(self, *a, **k):
            raise hipets.HipetsError(self.message, self.kind)

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
        hdist.plan_cem_sharded(FakeEngine("act_dim 5 != model act_dim 6", hipets.ERR_INVALID_ARGUMENT), None, None, None, None, None, 20)
    # the CLASS of the error decides (hipets_last_error_kind), not its wording: the time-out report an earlier unasked launch leaves
    # contains none of "RCCL / failed: / hipMalloc" and is a runtime failure all the same (round-3 advice)
    eng = FakeEngine("a persistent DEVICE-mode rollout timed out waiting for rows of another workgroup ... nobody asked", hipets.ERR_TIMEOUT)
    with pytest.warns(UserWarning, match="falling back to a single-GPU plan"):
        plan, fell_back = hdist.plan_cem_sharded(eng, None, None, None, None, None, 20, seed=4, plan_id=1)
    assert plan == ("single-gpu plan", 4, 1) and fell_back and eng.destroyed

    class TimedOutEngine(FakeEngine):  # the plan was enqueued and "ran", but a persistent rollout inside it gave up
        def plan_cem_sharded(self, *a, **k):
            return "sharded plan built on invalid returns"

        def check_async_error(self):
            return True

    eng = TimedOutEngine("")
    with pytest.warns(UserWarning, match="timed out"):
        plan, fell_back = hdist.plan_cem_sharded(eng, None, None, None, None, None, 20, seed=1, plan_id=2)
    assert plan == ("single-gpu plan", 1, 2) and fell_back and eng.destroyed

    # round-4 advice: an exception that is NOT the library's (a torch RuntimeError out of the stream sync: a HIP fault on this rank)
    # must still reach the ranks' agreement -- as a runtime failure, so the peers fall back instead of blocking in the all-reduce --
    # and then leave this rank as what it is; an error raised on the Python side without a kind is a runtime failure, not a rejection
    agreed = []

    class FaultingEngine(FakeEngine):
        def plan_cem_sharded(self, *a, **k):
            return "enqueued"

        def synchronize(self):
            raise RuntimeError("HIP error: an illegal memory access was encountered")

    orig = hdist._worst_status
    hdist._worst_status = lambda code, group=None: (agreed.append(code), orig(code, group))[1]
    try:
        with pytest.raises(RuntimeError, match="illegal memory access"):
            hdist.plan_cem_sharded(FaultingEngine(""), None, None, None, None, None, 20)
    finally:
        hdist._worst_status = orig
    assert agreed == [hdist._RUNTIME_FAILURE]
    assert hipets.HipetsError("Engine.set_model() has not been called").kind == hipets.ERR_RUNTIME


def test_failed_sharded_mppi_and_icem_attempts_are_undone_before_the_single_gpu_plan():
    """MPPI's persistent mean and iCEM's persistent elites are modified IN PLACE by a plan (trajectory_opt.py:238-311 self.mean,
    :476 self.elite): when the sharded attempt fails half way, the single-GPU fallback must start from what they were."""
    import hipets
    from hipets import dist as hdist

    class Eng:
        comm_world, comm_rank, comm_group = 2, 0, None
        seen = None

        def synchronize(self):
            pass

        def check_async_error(self):
            return False

        def comm_destroy(self):
            self.comm_world = 1

        def plan_mppi_sharded(self, pop, H, A, it, gamma, beta, mean, *a, **k):
            mean += 7.0  # half-way state of a plan that then dies in a collective
            raise hipets.HipetsError("RCCL error 2", hipets.ERR_RUNTIME)

        def plan_mppi(self, pop, H, A, it, gamma, beta, mean, *a, **k):
            self.seen = mean.clone()
            mean += 1.0
            return mean

        def plan_icem_sharded(self, p, x0, lo, up, elite, has_elite, *a, **k):
            elite.fill_(-3.0)
            raise hipets.HipetsError("hipMalloc(123) failed", hipets.ERR_RUNTIME)

        def plan_icem(self, p, x0, lo, up, elite, has_elite, *a, **k):
            self.seen = elite.clone()
            return "icem plan"

    eng, mean = Eng(), torch.full((3, 2), 0.5)
    with pytest.warns(UserWarning, match="falling back"):
        out, fell_back = hdist.plan_mppi_sharded(eng, 10, 3, 2, 2, 0.9, 0.9, mean, None, None, None, 5)
    assert fell_back and torch.equal(eng.seen, torch.full((3, 2), 0.5)) and torch.equal(mean, torch.full((3, 2), 1.5)) and out is mean
    eng, elite = Eng(), torch.arange(12.0).reshape(2, 3, 2)
    with pytest.warns(UserWarning, match="falling back"):
        out, fell_back = hdist.plan_icem_sharded(eng, None, None, None, None, elite, True, None, 5)
    assert fell_back and out == "icem plan" and torch.equal(eng.seen, torch.arange(12.0).reshape(2, 3, 2))


def _timeout_worker(rank, world, port, tmpdir):
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "mbrl-lib_amd"))
    from hipets import dist as hdist
    from hipets.engine import Engine
    from hipets.planning import _BoundObjective, _OptimizerSnapshot

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        class Eng(Engine):  # an engine whose time-out flag is scripted (no GPU here)
            def __init__(self, hits):
                self.hits = list(hits)

            def check_async_error(self):
                return self.hits.pop(0) if self.hits else False

        class Inner:
            calls = 0

            def __init__(self, eng):
                self.engine = eng

            def __call__(self, s0, a):
                return a.sum(dim=(1, 2))

        class Opt:
            calls = 3

        # plan 1: only rank 1's engine reports a timed-out rollout; plan 2: nobody does
        eng = Eng([rank == 1, False])
        obj = _BoundObjective(hdist.ShardedEvalFn(Inner(eng)), np.zeros(2, np.float32))
        snap = _OptimizerSnapshot(Opt(), obj)
        first = snap.engines_report_timeout()   # must be True on BOTH ranks: re-run everywhere or nowhere
        second = snap.engines_report_timeout()
        # an objective that is not sharded never starts a collective (independent agents per rank must not be coupled)
        lone = _OptimizerSnapshot(Opt(), _BoundObjective(Inner(Eng([rank == 0])), np.zeros(2, np.float32))).engines_report_timeout()
        torch.save((first, second, lone), os.path.join(tmpdir, f"t{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_timed_out_plan_is_rerun_on_every_rank_or_none_world2_gloo(tmp_path):
    """Round-3 advice: TrajectoryOptimizer.optimize re-runs a plan whose rollouts timed out; with a ShardedEvalFn objective a
    rank re-running alone would issue all-gathers its peers never match.  The flag is all-reduced over the objective's group."""
    world = 2
    mp.spawn(_timeout_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0, r1 = torch.load(tmp_path / "t0.pt"), torch.load(tmp_path / "t1.pt")
    assert r0[:2] == (True, False) and r1[:2] == (True, False)
    assert r0[2] is True and r1[2] is False  # purely local
