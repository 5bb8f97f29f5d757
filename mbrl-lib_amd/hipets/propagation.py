"""Uncertainty-propagation helpers of mbrl.util.math (mbrl/util/math.py:179-303), same names and semantics, on
whatever device the tensors live on.  They select among ALREADY computed ensemble predictions ``[E, B, Od]``: pure index
plumbing (gathers / means), so they are torch ops, not kernels.

Inside the engine the same selections are never materialised: the fused rollout evaluates, for every row, only the member
that row is assigned to.  ``propagate_from_indices`` semantics for a whole rollout are reached with
``Engine.rollout(..., mode="exact", members=indices)`` (explicit per-row member maps, any batch size), TS1 / TS-infinity
with balanced maps through ``mode="device"`` / ``"fast"``, the expectation through ``propagation="expectation"``."""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch


def propagate_from_indices(predicted_tensor: torch.Tensor, indices: torch.Tensor) -> torch.Tensor:
    """``output[i, :] = predicted_tensor[indices[i], i, :]`` (util/math.py:180-196)."""
    return predicted_tensor[indices, torch.arange(predicted_tensor.shape[1], device=predicted_tensor.device), :]


def propagate_random_model(predictions: Tuple[torch.Tensor, ...]) -> Tuple[torch.Tensor, ...]:
    """A random member per row, drawn independently for every tensor of the tuple (util/math.py:199-220)."""
    output: List[torch.Tensor] = []
    for predicted_tensor in predictions:
        assert predicted_tensor.ndim == 3
        num_models, batch_size, _ = predicted_tensor.shape
        model_indices = torch.randint(num_models, size=(batch_size,), device=predicted_tensor.device)
        output.append(propagate_from_indices(predicted_tensor, model_indices))
    return tuple(output)


def propagate_expectation(predictions: Tuple[torch.Tensor, ...]) -> Tuple[torch.Tensor, ...]:
    """Mean over the members (util/math.py:223-241)."""
    output: List[torch.Tensor] = []
    for predicted_tensor in predictions:
        assert predicted_tensor.ndim == 3
        output.append(predicted_tensor.mean(dim=0))
    return tuple(output)


def propagate_fixed_model(predictions: Tuple[torch.Tensor, ...], propagation_indices: torch.Tensor) -> Tuple[torch.Tensor, ...]:
    """The same given member per row for every tensor (util/math.py:244-264)."""
    output: List[torch.Tensor] = []
    for predicted_tensor in predictions:
        assert predicted_tensor.ndim == 3
        output.append(propagate_from_indices(predicted_tensor, propagation_indices))
    return tuple(output)


def propagate(predictions: Tuple[torch.Tensor, ...], propagation_method: str = "expectation",
              propagation_indices: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, ...]:
    """util/math.py:267-303: "random_model" (TS1), "fixed_model" (TS-infinity), "expectation"."""
    if propagation_method == "random_model":
        return propagate_random_model(predictions)
    if propagation_method == "fixed_model":
        return propagate_fixed_model(predictions, propagation_indices)
    if propagation_method == "expectation":
        return propagate_expectation(predictions)
    raise ValueError(f"Invalid propagation method {propagation_method}.")
