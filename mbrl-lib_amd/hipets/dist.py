"""Population-sharded data parallelism: one process per GPU, one small all-gather per optimizer
iteration.  The reference has no distributed path at all; this is the one
collective the engine adds.

Sampling is replicated (same counter-based seed on every rank => identical population, cheaper to
recompute than to broadcast 360 KB); rank g rolls out candidates ``shard_bounds(pop, world, g)`` with all
their particles; the per-candidate returns (``pop`` floats: 2 KB at pop=500) are all-gathered; every rank
then runs the identical top-k / refit on identical data, so ``mu`` / ``var`` stay bit-identical across
ranks with no broadcast.  Backend "nccl" is RCCL over xGMI on ROCm; "gloo" is used by the CPU tests.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import numpy as np
import torch
import torch.distributed as dist


def shard_bounds(pop: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous candidate range of ``rank``; the first ``pop % world`` ranks get one extra."""
    base, extra = divmod(pop, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_values(local: torch.Tensor, pop: int, group=None) -> torch.Tensor:
    """All-gather the per-candidate returns of every shard into a [pop] tensor (identical on all ranks).
    Uneven shards are padded to ceil(pop / world) so a single fixed-size all-gather suffices."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    width = -(-pop // world)
    lo, hi = shard_bounds(pop, world, rank)
    if local.shape[0] != hi - lo:
        raise ValueError(f"rank {rank} holds {local.shape[0]} values, expected {hi - lo}")
    backend = dist.get_backend(group)
    if backend == "nccl" and pop % world == 0 and local.is_contiguous():
        out = torch.empty(pop, dtype=local.dtype, device=local.device)  # even shards: the collective writes the result in place
        dist.all_gather_into_tensor(out, local, group=group)
        return out
    # gloo (CPU tests, or several ranks sharing one GPU in the single-GPU smoke test) gathers through host memory
    comm_dev = local.device if backend == "nccl" else torch.device("cpu")
    send = torch.zeros(width, dtype=local.dtype, device=comm_dev)
    send[: hi - lo] = local.to(comm_dev)
    recv = torch.empty(world * width, dtype=local.dtype, device=comm_dev)
    if backend == "nccl":
        dist.all_gather_into_tensor(recv, send, group=group)  # one RCCL collective over xGMI
    else:
        dist.all_gather(list(recv.view(world, width).unbind(0)), send, group=group)
    recv = recv.view(world, width).to(local.device)
    parts = []
    for r in range(world):
        a, b = shard_bounds(pop, world, r)
        parts.append(recv[r, : b - a])
    return torch.cat(parts)


class ShardedEvalFn:
    """Wraps any ``trajectory_eval_fn`` so that each rank evaluates only its candidate shard and the full
    value vector comes back through one all-gather.  Requires identical ``action_sequences`` on every
    rank (replicated sampling)."""

    def __init__(self, eval_fn: Callable, group=None):
        self.eval_fn = eval_fn
        self.group = group
        self.mode = getattr(eval_fn, "mode", None)

    def __call__(self, initial_state: np.ndarray, action_sequences: torch.Tensor) -> torch.Tensor:
        pop = action_sequences.shape[0]
        world = dist.get_world_size(self.group)
        rank = dist.get_rank(self.group)
        lo, hi = shard_bounds(pop, world, rank)
        local = self.eval_fn(initial_state, action_sequences[lo:hi].contiguous())
        return gather_values(local, pop, self.group)


def is_distributed() -> bool:
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def init_engine_comm(engine, group=None):
    """Give ``engine`` its own RCCL communicator over the ranks of ``group`` (hipets_comm_init): rank 0 creates the id,
    torch.distributed only carries its 128 bytes.  Afterwards the optimizer classes of ``hipets.planning`` run their fused plans
    SHARDED over these ranks (``TrajectoryOptimizerAgent.act`` needs no other change): one device-side loop per rank, one
    ncclAllGather of the candidates' returns per iteration, no per-iteration host work."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    box = [engine.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    engine.comm_init(box[0], rank, world)
    engine.comm_group = group  # the process group the ranks agree over (plan_*_sharded below)
    return engine


def _worst_status(code: int, group=None) -> int:
    """Maximum of ``code`` over the ranks (torch.distributed, host side): the ranks agree on the outcome of a sharded plan
    before any of them acts on it, so that a fallback is taken by ALL of them or none (a lone rank planning on its own while
    its peers wait in a collective would desynchronise the job)."""
    if not is_distributed():
        return code
    t = torch.tensor([code], dtype=torch.int32)
    if dist.get_backend(group) == "nccl":
        t = t.cuda()
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return int(t.item())


_OK, _RUNTIME_FAILURE, _REJECTED = 0, 1, 2


def run_sharded(engine, sharded_call: Callable, single_call: Callable, group=None, restore: Optional[Callable] = None):
    """The failure policy of SURVEY.md section 5 around one sharded plan (CEM, MPPI or iCEM).  ``sharded_call()`` runs when the
    engine has a communicator; afterwards its stream is synchronised, every rank asks its engine whether a rollout was cut
    short (``check_async_error``) and the ranks AGREE (one host-side all-reduce over ``group`` when torch.distributed is up)
    on whether the plan stands.  If any rank had a runtime failure -- RCCL reported an error while the plan was enqueued, a
    persistent rollout timed out, a HIP call or an allocation failed: ``HipetsError.kind`` (hipets_last_error_kind), not the
    wording of the message, says so -- every rank drops its communicator, warns, puts back what the attempt changed in place
    (``restore()``: MPPI's mean, iCEM's elites) and runs ``single_call()``: the WHOLE population on its own GPU with the same
    sampler streams, so all ranks still return the same valid plan for the observation instead of failing the control loop.
    Arguments the library rejects (a population the shards cannot hold, a wrong action width: identical on every rank) still
    raise.  Returns ``(result, used_fallback)``."""
    import warnings

    from ._lib import ERR_INVALID_ARGUMENT, HipetsError

    if engine.comm_world > 1:
        code, reason, result, error, foreign = _OK, None, None, None, None
        try:
            result = sharded_call()
            engine.synchronize()
            if engine.check_async_error():
                code, reason = _RUNTIME_FAILURE, "a persistent DEVICE-mode rollout timed out"
        except HipetsError as exc:
            error, reason = exc, str(exc)
            code = _REJECTED if exc.kind == ERR_INVALID_ARGUMENT else _RUNTIME_FAILURE
        except Exception as exc:  # not the library's: a torch RuntimeError out of engine.synchronize() (a HIP fault on THIS rank), ...
            # the peers are (or will be) waiting in the agreement below: take part in it as a runtime failure so that they fall
            # back instead of blocking in the all-reduce, THEN let the exception out -- this rank's device state is unknown
            foreign, reason, code = exc, repr(exc), _RUNTIME_FAILURE
        worst = _worst_status(code, group)
        if foreign is not None:
            raise foreign
        if worst == _OK:
            return result, False
        if worst == _REJECTED:
            raise error if code == _REJECTED else HipetsError("a peer rank rejected the arguments of the sharded plan", ERR_INVALID_ARGUMENT)
        warnings.warn(f"hipets: sharded plan failed on rank {engine.comm_rank} or a peer ({reason or 'peer failure'}); "
                      f"falling back to a single-GPU plan on every rank")
        try:
            engine.comm_destroy()
        except HipetsError:
            engine.comm_world, engine.comm_rank = 1, 0
        if restore is not None:
            restore()
    return single_call(), True


def plan_cem_sharded(engine, params, x0, lower, upper, s0, num_particles: int, seed: int = 0, plan_id: int = 0, group=None):
    """``Engine.plan_cem_sharded`` under :func:`run_sharded`'s failure policy (fallback: ``Engine.plan_cem`` of the whole
    population).  Returns ``(plan, used_fallback)``."""
    return run_sharded(engine,
                       lambda: engine.plan_cem_sharded(params, x0, lower, upper, s0, num_particles, seed=seed, plan_id=plan_id),
                       lambda: engine.plan_cem(params, x0, lower, upper, s0, num_particles, seed=seed, plan_id=plan_id), group)


def plan_mppi_sharded(engine, pop: int, H: int, A: int, num_iterations: int, gamma: float, beta: float, mean: torch.Tensor, lower, upper, s0,
                      num_particles: int, seed: int = 0, plan_id: int = 0, group=None):
    """``Engine.plan_mppi_sharded`` (hipets_plan_mppi_sharded) under :func:`run_sharded`'s failure policy.  ``mean`` -- the
    optimizer's persistent mean -- is shifted and refined IN PLACE; a failed sharded attempt is undone before the single-GPU plan
    runs.  Returns ``(mean, used_fallback)``."""
    before = mean.clone()
    return run_sharded(engine,
                       lambda: engine.plan_mppi_sharded(pop, H, A, num_iterations, gamma, beta, mean, lower, upper, s0, num_particles, seed=seed, plan_id=plan_id),
                       lambda: engine.plan_mppi(pop, H, A, num_iterations, gamma, beta, mean, lower, upper, s0, num_particles, seed=seed, plan_id=plan_id),
                       group, restore=lambda: mean.copy_(before))


def plan_icem_sharded(engine, params, x0, lower, upper, elite: torch.Tensor, has_elite: bool, s0, num_particles: int, seed: int = 0,
                      plan_id: int = 0, keep_idx=None, group=None):
    """``Engine.plan_icem_sharded`` (hipets_plan_icem_sharded) under :func:`run_sharded`'s failure policy.  ``elite`` -- the
    optimizer's persistent elite set -- is overwritten by every iteration; a failed sharded attempt is undone before the single-GPU
    plan runs.  Returns ``(plan, used_fallback)``."""
    before = elite.clone()
    return run_sharded(engine,
                       lambda: engine.plan_icem_sharded(params, x0, lower, upper, elite, has_elite, s0, num_particles, seed=seed, plan_id=plan_id, keep_idx=keep_idx),
                       lambda: engine.plan_icem(params, x0, lower, upper, elite, has_elite, s0, num_particles, seed=seed, plan_id=plan_id, keep_idx=keep_idx),
                       group, restore=lambda: elite.copy_(before))
