"""TEST INFRASTRUCTURE ONLY: (de)serialise an OracleModel plus case arrays to a small .npz fixture."""
from __future__ import annotations

import json

import numpy as np
import torch

from .pets_oracle import OracleModel

_SCALARS = ("elite_models", "activation", "propagation", "deterministic", "target_is_delta", "no_delta_list",
            "learned_rewards", "obs_process", "reward", "termination")


def save_case(path: str, om: OracleModel, meta: dict, arrays: dict):
    d = {}
    for i, (w, b) in enumerate(zip(om.weights, om.biases)):
        d[f"w{i}"] = w.numpy()
        d[f"b{i}"] = b.numpy()
    if om.min_logvar is not None:
        d["min_logvar"] = om.min_logvar.numpy()
        d["max_logvar"] = om.max_logvar.numpy()
    if om.norm_mean is not None:
        d["norm_mean"] = om.norm_mean.numpy()
        d["norm_std"] = om.norm_std.numpy()
    m = {k: getattr(om, k) for k in _SCALARS}
    m["ensemble_kind"] = om.ensemble_kind
    m["elite_models"] = None if m["elite_models"] is None else [int(x) for x in m["elite_models"]]
    m["no_delta_list"] = [int(x) for x in m["no_delta_list"]]
    m["n_layers"] = len(om.weights)
    m.update(meta)
    d["meta_json"] = np.frombuffer(json.dumps(m).encode(), dtype=np.uint8)
    for k, v in arrays.items():
        d["x_" + k] = v.numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
    np.savez(path, **d)


def load_case(path: str):
    z = np.load(path)
    meta = json.loads(bytes(z["meta_json"]).decode())
    n = meta["n_layers"]
    om = OracleModel(
        weights=[torch.from_numpy(z[f"w{i}"]) for i in range(n)],
        biases=[torch.from_numpy(z[f"b{i}"]) for i in range(n)],
        min_logvar=torch.from_numpy(z["min_logvar"]) if "min_logvar" in z else None,
        max_logvar=torch.from_numpy(z["max_logvar"]) if "max_logvar" in z else None,
        norm_mean=torch.from_numpy(z["norm_mean"]) if "norm_mean" in z else None,
        norm_std=torch.from_numpy(z["norm_std"]) if "norm_std" in z else None,
        ensemble_kind=meta.get("ensemble_kind", "gaussian_mlp"),
        **{k: meta[k] for k in _SCALARS},
    )
    arrays = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("x_")}
    return om, meta, arrays
