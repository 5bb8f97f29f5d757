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

from typing import Callable, Tuple

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
    torch.distributed only carries its 128 bytes.  Afterwards ``Engine.plan_cem_sharded`` runs the whole sharded plan as one
    device-side loop per rank (no per-iteration host work)."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    box = [engine.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    engine.comm_init(box[0], rank, world)
    return engine


def plan_cem_sharded(engine, params, x0, lower, upper, s0, num_particles: int, seed: int = 0, plan_id: int = 0):
    """``Engine.plan_cem_sharded`` with the failure policy of SURVEY.md section 5: if the engine has no communicator, or RCCL
    reports an error while the sharded plan is enqueued, warn and plan the WHOLE population on this GPU alone
    (``hipets_plan_cem``: same sampler streams, so every rank that falls back still returns a valid plan for its
    observation) instead of failing the control loop.  Returns ``(plan, used_fallback)``."""
    import warnings

    from ._lib import HipetsError

    if engine.comm_world > 1:
        try:
            return engine.plan_cem_sharded(params, x0, lower, upper, s0, num_particles, seed=seed, plan_id=plan_id), False
        except HipetsError as exc:
            if "RCCL" not in str(exc) and "communicator" not in str(exc):
                raise
            warnings.warn(f"hipets: sharded plan failed ({exc}); falling back to a single-GPU plan on rank {engine.comm_rank}")
            try:
                engine.comm_destroy()
            except HipetsError:
                engine.comm_world, engine.comm_rank = 1, 0
    return engine.plan_cem(params, x0, lower, upper, s0, num_particles, seed=seed, plan_id=plan_id), engine.comm_world <= 1
