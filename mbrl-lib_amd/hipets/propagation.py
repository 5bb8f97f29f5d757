"""Uncertainty propagation over ALREADY computed ensemble predictions -- the public names of mbrl.util.math
(mbrl/util/math.py:180-303: ``propagate``, ``propagate_from_indices``, ``propagate_random_model``, ``propagate_fixed_model``,
``propagate_expectation``), written here from their contract rather than from their code:

    a prediction is a stack ``[E, B, D]`` (member, row, output dim); propagating it yields ``[B, D]``:
      from indices   row i takes member ``indices[i]``                                 (:180-196)
      random_model   a fresh uniform member per row, drawn anew FOR EVERY TENSOR of the tuple -- mean and logvar of one row
                     may come from different members; that is the reference's behaviour (:199-220) and it is kept
      fixed_model    the caller's member per row, shared by all tensors of the tuple   (:244-264)
      expectation    the mean over members                                              (:223-241)

Everything is one ``gather`` along the member axis (or one ``mean``): index plumbing on whatever device the tensors live on,
not kernel work.  Inside the engine these selections are never materialised: the fused rollout evaluates for every row only
the member that row is assigned to -- ``Engine.rollout(..., mode="exact", members=indices)`` is ``propagate_from_indices``
for a whole rollout (any batch size, unbalanced maps), ``mode="device"`` / ``"fast"`` are TS1 / TS-infinity with balanced maps,
``propagation="expectation"`` the member mean.  Random draws use ``torch.randint`` on the tensor's device from torch's global
generator, one draw of ``B`` members per tensor in tuple order, so a seeded caller sees the reference's stream."""
from __future__ import annotations

from typing import Callable, Dict, Optional, Sequence, Tuple

import torch

Stack = torch.Tensor  # [E, B, D]


def _member_axis_gather(stack: Stack, member_of_row: torch.Tensor) -> torch.Tensor:
    """``out[i, :] = stack[member_of_row[i], i, :]`` as a single gather over dim 0."""
    if stack.ndim != 3:
        raise AssertionError(f"predictions must be [E, B, D] stacks, got a tensor of rank {stack.ndim}")
    _, rows, width = stack.shape
    pick = member_of_row.to(device=stack.device, dtype=torch.long).reshape(1, rows, 1).expand(1, rows, width)
    return torch.gather(stack, 0, pick).squeeze(0)


def _per_tensor(stacks: Sequence[Stack], member_source: Callable[[Stack], torch.Tensor]) -> Tuple[torch.Tensor, ...]:
    return tuple(_member_axis_gather(s, member_source(s)) for s in stacks)


def propagate_from_indices(predicted_tensor: Stack, indices: torch.Tensor) -> torch.Tensor:
    return _member_axis_gather(predicted_tensor, indices)


def propagate_random_model(predictions: Tuple[Stack, ...]) -> Tuple[torch.Tensor, ...]:
    def fresh_members(stack: Stack) -> torch.Tensor:
        if stack.ndim != 3:
            raise AssertionError(f"predictions must be [E, B, D] stacks, got a tensor of rank {stack.ndim}")
        return torch.randint(stack.shape[0], size=(stack.shape[1],), device=stack.device)

    return _per_tensor(predictions, fresh_members)


def propagate_fixed_model(predictions: Tuple[Stack, ...], propagation_indices: torch.Tensor) -> Tuple[torch.Tensor, ...]:
    return _per_tensor(predictions, lambda _stack: propagation_indices)


def propagate_expectation(predictions: Tuple[Stack, ...]) -> Tuple[torch.Tensor, ...]:
    for s in predictions:
        if s.ndim != 3:
            raise AssertionError(f"predictions must be [E, B, D] stacks, got a tensor of rank {s.ndim}")
    return tuple(torch.mean(s, dim=0) for s in predictions)


_BY_NAME: Dict[str, Callable[[Tuple[Stack, ...], Optional[torch.Tensor]], Tuple[torch.Tensor, ...]]] = {
    "random_model": lambda preds, _idx: propagate_random_model(preds),        # TS1 of the PETS paper
    "fixed_model": lambda preds, idx: propagate_fixed_model(preds, idx),      # TS-infinity
    "expectation": lambda preds, _idx: propagate_expectation(preds),
}


def propagate(predictions: Tuple[Stack, ...], propagation_method: str = "expectation",
              propagation_indices: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, ...]:
    """Dispatch by name (mbrl/util/math.py:267-303); ``propagation_indices`` is read by ``"fixed_model"`` only."""
    try:
        rule = _BY_NAME[propagation_method]
    except KeyError:
        raise ValueError(f"Invalid propagation method {propagation_method}.") from None
    return rule(predictions, propagation_indices)
