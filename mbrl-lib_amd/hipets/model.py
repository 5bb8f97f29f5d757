"""ModelSpec: the snapshot of a PETS dynamics model the engine consumes, and its extraction from
live mbrl-lib objects (duck-typed -- mbrl itself is never imported here).

Mirrors what ModelEnv.evaluate_action_sequences reads:
``model_env.dynamics_model`` (OneDTransitionRewardModel, mbrl/models/one_dim_tr_model.py:29-116) ->
``.model`` (GaussianMLP, mbrl/models/gaussian_mlp.py:69-127), normaliser (mbrl/util/math.py:95-143),
``model_env.reward_fn`` / ``.termination_fn`` (mbrl/env/reward_fns.py, termination_fns.py).
Anything the fused kernel cannot express raises UnsupportedModelError so that callers fall back to
the reference path instead of silently approximating.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np
import torch


class UnsupportedModelError(ValueError):
    pass


_ACT_BY_CLASS = {"SiLU": "silu", "ReLU": "relu", "LeakyReLU": "leaky_relu", "Tanh": "tanh", "Sigmoid": "sigmoid"}
_KNOWN_REWARDS = ("cartpole", "cartpole_pets", "inverted_pendulum", "halfcheetah", "pusher", "none")
_KNOWN_TERMS = ("no_termination", "cartpole", "inverted_pendulum", "hopper", "walker2d", "ant", "humanoid")


@dataclass
class ModelSpec:
    weights: List[torch.Tensor]  # per linear layer [E, in_l, out_l] f32
    biases: List[torch.Tensor]  # per linear layer [E, 1, out_l] f32
    obs_dim: int
    act_dim: int
    min_logvar: Optional[torch.Tensor] = None  # [1, out]
    max_logvar: Optional[torch.Tensor] = None
    elite_models: Optional[Sequence[int]] = None
    activation: str = "silu"
    leaky_slope: float = 0.01
    propagation: str = "random_model"
    deterministic: bool = False
    norm_mean: Optional[torch.Tensor] = None  # [1, in] f32 / f64
    norm_std: Optional[torch.Tensor] = None
    target_is_delta: bool = True
    no_delta_list: Sequence[int] = field(default_factory=list)
    learned_rewards: bool = False
    obs_process: str = "none"
    reward: Optional[str] = "halfcheetah"  # None => learned reward (last model output); "none" => caller's callable
    termination: str = "no_termination"
    custom_reward_fn: Optional[object] = None  # arbitrary torch callables (act, next_obs) -> [B,1]; UNFUSED path only
    custom_termination_fn: Optional[object] = None
    # "gaussian_mlp": one GaussianMLP with E members (balanced shuffles, batch % members rule, elites).
    # "basic_ensemble": mbrl.models.BasicEnsemble of E single-member GaussianMLPs: every row draws its member
    #   independently from the generator (basic_ensemble.py:122-129, 255-260), any batch size, no elites,
    #   min/max_logvar are [E, out] (every member owns its bounds).
    ensemble_kind: str = "gaussian_mlp"
    # arithmetic of the linear layers: "f32" (fp32 MFMA, the default and the graded mode) or "bf16x3" (fp32 operands as three
    # bf16 pieces on the bf16 matrix pipe: fp32-accurate to a few product ulps, ~1.5x faster; available for the shapes that have
    # a shape-specialised kernel instance -- anything else fails loudly at the rollout call)
    precision: str = "f32"

    # ---- derived ---------------------------------------------------------------------------
    @property
    def ensemble_size(self) -> int:
        return int(self.weights[0].shape[0])

    @property
    def in_dim(self) -> int:
        return int(self.weights[0].shape[1])

    @property
    def hid(self) -> int:
        return int(self.weights[0].shape[2])

    @property
    def out_dim(self) -> int:
        n = int(self.weights[-1].shape[2])
        return n if self.deterministic else n // 2

    @property
    def members(self) -> List[int]:
        if self.elite_models is not None and self.ensemble_kind != "basic_ensemble":  # basic_ensemble.py:262-266
            return [int(i) for i in self.elite_models]
        return list(range(self.ensemble_size))

    def flops_per_candidate_step(self) -> int:
        """2 * (in*hid + (L-1)*hid^2 + hid*out_total)."""
        return 2 * sum(int(w.shape[1]) * int(w.shape[2]) for w in self.weights)

    def validate(self):
        if self.precision not in ("f32", "bf16x3"):
            raise ValueError(f"precision must be 'f32' or 'bf16x3', got {self.precision!r}")
        if self.ensemble_kind not in ("gaussian_mlp", "basic_ensemble"):
            raise UnsupportedModelError(f"ensemble kind {self.ensemble_kind!r} has no fused implementation")
        if self.activation not in _ACT_BY_CLASS.values():
            raise UnsupportedModelError(f"activation {self.activation!r} has no fused implementation")
        if self.propagation not in ("random_model", "fixed_model", "expectation"):
            raise ValueError(f"Invalid propagation method {self.propagation}.")  # gaussian_mlp.py:216
        if self.reward is not None and self.reward not in _KNOWN_REWARDS:
            raise UnsupportedModelError(f"reward_fn {self.reward!r} has no fused implementation")
        if self.reward is None and not self.learned_rewards:
            raise UnsupportedModelError("reward_fn is None but the model does not learn rewards")
        if self.termination not in _KNOWN_TERMS:
            raise UnsupportedModelError(f"termination_fn {self.termination!r} has no fused implementation")
        if self.obs_process not in ("none", "halfcheetah", "cartpole_pets"):
            raise UnsupportedModelError(f"obs_process_fn {self.obs_process!r} has no fused implementation")
        if len(self.weights) < 2 or len(self.weights) > 8:
            raise UnsupportedModelError("need 2..8 linear layers")
        for li in range(1, len(self.weights) - 1):
            if tuple(self.weights[li].shape[1:]) != (self.hid, self.hid):
                raise UnsupportedModelError("hidden layers must share one width")
        exp_in = self.obs_dim + (1 if self.obs_process == "cartpole_pets" else 0) + self.act_dim
        if self.in_dim != exp_in:
            raise UnsupportedModelError(f"model in_size {self.in_dim} != obs'+act = {exp_in}")
        if self.out_dim != self.obs_dim + (1 if self.learned_rewards else 0):
            raise UnsupportedModelError("model out_size inconsistent with obs_dim / learned_rewards")


_CLOSED_FORM_MODULES = {"reward": "mbrl.env.reward_fns", "termination": "mbrl.env.termination_fns"}
_OBS_PROCESS_FNS = {("mbrl.env.pets_halfcheetah", "HalfCheetahEnv.preprocess_fn"): "halfcheetah",
                    ("mbrl.env.pets_cartpole", "CartPoleEnv.preprocess_fn"): "cartpole_pets"}


def _closed_form(fn, kind: str, known) -> Optional[str]:
    """Name of the mbrl.env closed form ``fn`` IS, or None.  A function is recognised by where it is defined
    (``mbrl.env.reward_fns.halfcheetah`` ...), never by its bare name: a user function that happens to be called
    ``halfcheetah`` must not be replaced by the built-in formula.  Anything else may opt in explicitly with an attribute
    ``hipets_closed_form = "<name>"`` (tests and duck-typed models do)."""
    if fn is None:
        return None
    tag = getattr(fn, "hipets_closed_form", None)
    if tag is not None:
        return tag if tag in known else None
    name = getattr(fn, "__name__", None)
    if getattr(fn, "__module__", None) == _CLOSED_FORM_MODULES[kind] and name in known:
        return name
    return None


def _obs_process_name(fn) -> Optional[str]:
    tag = getattr(fn, "hipets_closed_form", None)
    if tag is not None:
        return tag if tag in ("halfcheetah", "cartpole_pets") else None
    return _OBS_PROCESS_FNS.get((getattr(fn, "__module__", None), getattr(fn, "__qualname__", None)))


def _read_gaussian_mlp(mlp):
    """(weights, biases, activation, leaky_slope, deterministic, min_logvar, max_logvar) of a live GaussianMLP
    (mbrl/models/gaussian_mlp.py:69-127); tensors are references to the live parameters."""
    ws, bs = [], []
    act_name = None
    slope = 0.01
    for layer in mlp.hidden_layers:
        lin, act = layer[0], layer[1]
        if not getattr(lin, "use_bias", True):
            raise UnsupportedModelError("EnsembleLinearLayer without bias")
        ws.append(lin.weight.detach())
        bs.append(lin.bias.detach())
        name = _ACT_BY_CLASS.get(type(act).__name__)
        if name is None:
            raise UnsupportedModelError(f"activation module {type(act).__name__} has no fused implementation")
        if act_name is not None and name != act_name:
            raise UnsupportedModelError("mixed activation modules")
        act_name = name
        slope = float(getattr(act, "negative_slope", 0.01))
    ws.append(mlp.mean_and_logvar.weight.detach())
    bs.append(mlp.mean_and_logvar.bias.detach())
    if ws[0].ndim != 3:
        raise UnsupportedModelError("expected ensemble weights [E, in, out]")
    deterministic = bool(getattr(mlp, "deterministic", False))
    lo = None if deterministic else mlp.min_logvar.detach()
    hi = None if deterministic else mlp.max_logvar.detach()
    return ws, bs, act_name, slope, deterministic, lo, hi


def spec_from_model_env(model_env, obs_dim: Optional[int] = None, act_dim: Optional[int] = None,
                        allow_custom_fns: bool = False) -> ModelSpec:
    """Read a live ``mbrl.models.ModelEnv`` (or anything shaped like it).  No copy of the big tensors:
    the spec holds references to the live parameters; ``Engine.set_model`` packs them on device.

    ``allow_custom_fns``: a ``reward_fn`` / ``termination_fn`` that is not one of mbrl.env's closed forms is kept as a
    Python callable (``spec.custom_reward_fn`` / ``spec.custom_termination_fn``) for the UNFUSED path (the model
    transition stays fused, the callables run as torch ops between steps) instead of raising."""
    dm = model_env.dynamics_model
    mlp = getattr(dm, "model", None)
    ensemble_kind = "gaussian_mlp"
    if mlp is not None and hasattr(mlp, "members") and not hasattr(mlp, "hidden_layers"):
        # mbrl.models.BasicEnsemble (basic_ensemble.py:59-81): E members built from member_cfg; the fused engine takes the
        # conf/dynamics_model/basic_ensemble.yaml shape, i.e. single-member GaussianMLPs, stacked into one [E, in, out] set
        parts = [_read_gaussian_mlp(m) for m in mlp.members]
        if any(p[0][0].shape[0] != 1 for p in parts):
            raise UnsupportedModelError("BasicEnsemble members must be single-member GaussianMLPs")
        if any(p[2:5] != parts[0][2:5] for p in parts[1:]):
            raise UnsupportedModelError("BasicEnsemble members differ in activation / determinism")
        ws = [torch.cat([p[0][i] for p in parts], dim=0) for i in range(len(parts[0][0]))]
        bs = [torch.cat([p[1][i] for p in parts], dim=0) for i in range(len(parts[0][1]))]
        act_name, slope, deterministic = parts[0][2:5]
        lv_lo = None if deterministic else torch.cat([p[5] for p in parts], dim=0)  # [E, out]: per-member bounds
        lv_hi = None if deterministic else torch.cat([p[6] for p in parts], dim=0)
        ensemble_kind = "basic_ensemble"
    elif mlp is None or not hasattr(mlp, "hidden_layers") or not hasattr(mlp, "mean_and_logvar"):
        raise UnsupportedModelError("dynamics_model.model is not a GaussianMLP-shaped ensemble")
    else:
        ws, bs, act_name, slope, deterministic, lv_lo, lv_hi = _read_gaussian_mlp(mlp)
    norm = getattr(dm, "input_normalizer", None)
    obs_fn = getattr(dm, "obs_process_fn", None)
    obs_process = "none"
    if obs_fn is not None:
        obs_process = _obs_process_name(obs_fn)
        if obs_process is None:
            raise UnsupportedModelError(f"obs_process_fn {getattr(obs_fn, '__qualname__', obs_fn)!r} has no fused implementation")
    od = obs_dim if obs_dim is not None else int(model_env.observation_space.shape[0])
    ad = act_dim if act_dim is not None else int(model_env.action_space.shape[0])
    rew, term = model_env.reward_fn, model_env.termination_fn
    # model_env.py:124-128: a reward_fn that is not None ALWAYS wins over the learned reward, so an unrecognised callable
    # is never mapped to "learned" -- it is either kept for the unfused path or rejected
    rew_name = None if rew is None else _closed_form(rew, "reward", _KNOWN_REWARDS)
    term_name = _closed_form(term, "termination", _KNOWN_TERMS)
    custom_rew = custom_term = None
    if rew is not None and rew_name is None:
        if not allow_custom_fns:
            raise UnsupportedModelError(f"reward_fn {getattr(rew, '__qualname__', rew)!r} is not one of mbrl.env.reward_fns' closed forms")
        custom_rew, rew_name = rew, "none"
    if term_name is None:
        if not allow_custom_fns:
            raise UnsupportedModelError(f"termination_fn {getattr(term, '__qualname__', term)!r} is not one of mbrl.env.termination_fns' closed forms")
        custom_term, term_name = term, "no_termination"
    spec = ModelSpec(
        weights=ws, biases=bs, obs_dim=od, act_dim=ad,
        min_logvar=lv_lo, max_logvar=lv_hi, ensemble_kind=ensemble_kind,
        elite_models=list(mlp.elite_models) if getattr(mlp, "elite_models", None) is not None else None,
        activation=act_name or "relu", leaky_slope=slope,
        propagation=mlp.propagation_method, deterministic=deterministic,
        norm_mean=norm.mean.detach() if norm is not None else None,
        norm_std=norm.std.detach() if norm is not None else None,
        target_is_delta=bool(dm.target_is_delta), no_delta_list=list(dm.no_delta_list or []),
        learned_rewards=bool(dm.learned_rewards), obs_process=obs_process,
        reward=rew_name, termination=term_name,
    )
    spec.custom_reward_fn, spec.custom_termination_fn = custom_rew, custom_term
    spec.validate()
    return spec


def model_version(model_env) -> tuple:
    """Cheap freshness token: changes when ModelTrainer.train rewrote weights / normaliser / elites
    (mbrl/models/model_trainer.py:288-296)."""
    dm = model_env.dynamics_model
    mlp = dm.model
    vers = tuple(int(p._version) for p in mlp.parameters())
    norm = getattr(dm, "input_normalizer", None)
    # the stats are re-assigned tensors (util/math.py:120-127): (storage address, in-place version) of the LIVE tensors, which
    # the spec keeps referenced, so an address cannot be recycled while it is the recorded one
    nid = tuple((int(t.data_ptr()), int(t._version)) for t in (norm.mean, norm.std)) if norm is not None else ()
    el = tuple(mlp.elite_models) if getattr(mlp, "elite_models", None) is not None else None
    return (vers, nid, el, getattr(mlp, "propagation_method", None))


def spec_from_checkpoint(model_dir, obs_dim: int, act_dim: int, **spec_kwargs) -> ModelSpec:
    """Build a ModelSpec straight from a saved PETS run, without constructing any mbrl object:

    * ``model.pth``        = ``{"state_dict", "elite_models"}`` written by ``GaussianMLP.save``
      (mbrl/models/gaussian_mlp.py:381-387; keys ``hidden_layers.<i>.0.weight|bias``,
      ``mean_and_logvar.weight|bias``, ``min_logvar``, ``max_logvar``)
    * ``env_stats.pickle`` = ``{"mean", "std"}`` numpy arrays written by ``Normalizer.save``
      (mbrl/util/math.py:168-174); absent => no input normaliser.

    Everything a checkpoint does not record (activation, propagation, reward / termination fns, delta targets ...) comes
    from ``spec_kwargs`` with the ModelSpec defaults (SiLU, TS1, delta targets, halfcheetah reward)."""
    import os
    import pickle

    blob = torch.load(os.path.join(str(model_dir), "model.pth"), map_location="cpu", weights_only=False)
    sd = blob["state_dict"]
    ws, bs = [], []
    i = 0
    while f"hidden_layers.{i}.0.weight" in sd:
        ws.append(sd[f"hidden_layers.{i}.0.weight"])
        bs.append(sd[f"hidden_layers.{i}.0.bias"])
        i += 1
    if not ws or "mean_and_logvar.weight" not in sd:
        raise UnsupportedModelError("model.pth does not hold a GaussianMLP state dict")
    ws.append(sd["mean_and_logvar.weight"])
    bs.append(sd["mean_and_logvar.bias"])
    deterministic = "min_logvar" not in sd
    kw = dict(spec_kwargs)
    stats_path = os.path.join(str(model_dir), "env_stats.pickle")
    if os.path.exists(stats_path):
        with open(stats_path, "rb") as f:
            stats = pickle.load(f)
        kw.setdefault("norm_mean", torch.from_numpy(np.asarray(stats["mean"])))
        kw.setdefault("norm_std", torch.from_numpy(np.asarray(stats["std"])))
    elite = blob.get("elite_models")
    spec = ModelSpec(weights=ws, biases=bs, obs_dim=obs_dim, act_dim=act_dim,
                     min_logvar=None if deterministic else sd["min_logvar"], max_logvar=None if deterministic else sd["max_logvar"],
                     elite_models=list(elite) if elite is not None else None, deterministic=deterministic, **kw)
    spec.validate()
    return spec


# ---------------------------------------------------------------------------------------------
# PlaNet latent planner
# ---------------------------------------------------------------------------------------------
PLANET_TENSORS = ("w_embed", "b_embed", "w_ih", "b_ih", "w_hh", "b_hh", "w_prior1", "b_prior1", "w_prior2", "b_prior2",
                  "w_rew1", "b_rew1", "w_rew2", "b_rew2", "w_rew3", "b_rew3")


@dataclass
class PlaNetSpec:
    """The tensors ``PlaNetModel.sample`` reads (mbrl/models/planet.py:531-581), nn.Linear layout ([out, in] / [out])."""

    w_embed: torch.Tensor  # belief_model.embedding_layer[0]
    b_embed: torch.Tensor
    w_ih: torch.Tensor  # belief_model.rnn (GRUCell), gates r | z | n
    b_ih: torch.Tensor
    w_hh: torch.Tensor
    b_hh: torch.Tensor
    w_prior1: torch.Tensor  # prior_transition_model[0], [2]
    b_prior1: torch.Tensor
    w_prior2: torch.Tensor
    b_prior2: torch.Tensor
    w_rew1: torch.Tensor  # reward_model[0], [2], [4]
    b_rew1: torch.Tensor
    w_rew2: torch.Tensor
    b_rew2: torch.Tensor
    w_rew3: torch.Tensor
    b_rew3: torch.Tensor
    min_std: float = 0.1

    @property
    def latent_size(self) -> int:
        return int(self.w_prior2.shape[0]) // 2

    @property
    def belief_size(self) -> int:
        return int(self.w_hh.shape[1])

    @property
    def action_size(self) -> int:
        return int(self.w_embed.shape[1]) - self.latent_size

    @property
    def hidden_size(self) -> int:
        return int(self.w_prior1.shape[0])

    def flops_per_candidate_step(self) -> int:
        return 2 * sum(int(getattr(self, n).shape[0]) * int(getattr(self, n).shape[1]) for n in PLANET_TENSORS if n[0] == "w")

    def validate(self):
        L, A, Hb, F = self.latent_size, self.action_size, self.belief_size, self.hidden_size
        want = {"w_embed": (Hb, L + A), "b_embed": (Hb,), "w_ih": (3 * Hb, Hb), "b_ih": (3 * Hb,), "w_hh": (3 * Hb, Hb), "b_hh": (3 * Hb,),
                "w_prior1": (F, Hb), "b_prior1": (F,), "w_prior2": (2 * L, F), "b_prior2": (2 * L,), "w_rew1": (F, Hb + L), "b_rew1": (F,),
                "w_rew2": (F, F), "b_rew2": (F,), "w_rew3": (1, F), "b_rew3": (1,)}
        if A < 1 or L < 1:
            raise UnsupportedModelError("PlaNet heads have inconsistent sizes")
        for n, shp in want.items():
            if tuple(getattr(self, n).shape) != shp:
                raise UnsupportedModelError(f"PlaNet tensor {n} has shape {tuple(getattr(self, n).shape)}, expected {shp}")


def is_planet_model(model) -> bool:
    return all(hasattr(model, a) for a in ("belief_model", "prior_transition_model", "reward_model", "latent_state_size"))


def spec_from_planet_model(model) -> PlaNetSpec:
    """Read a live ``mbrl.models.PlaNetModel`` (planet.py:196-272); tensors are references to the live parameters."""
    if not is_planet_model(model):
        raise UnsupportedModelError("not a PlaNetModel")
    bm, pr, rw = model.belief_model, model.prior_transition_model, model.reward_model
    if type(bm.embedding_layer[1]).__name__ != "ReLU" or type(pr[1]).__name__ != "ReLU" or type(rw[1]).__name__ != "ReLU":
        raise UnsupportedModelError("PlaNet heads with a non-ReLU activation have no fused implementation")
    d = lambda t: t.detach()  # noqa: E731
    spec = PlaNetSpec(
        w_embed=d(bm.embedding_layer[0].weight), b_embed=d(bm.embedding_layer[0].bias),
        w_ih=d(bm.rnn.weight_ih), b_ih=d(bm.rnn.bias_ih), w_hh=d(bm.rnn.weight_hh), b_hh=d(bm.rnn.bias_hh),
        w_prior1=d(pr[0].weight), b_prior1=d(pr[0].bias), w_prior2=d(pr[2].weight), b_prior2=d(pr[2].bias),
        w_rew1=d(rw[0].weight), b_rew1=d(rw[0].bias), w_rew2=d(rw[2].weight), b_rew2=d(rw[2].bias),
        w_rew3=d(rw[4].weight), b_rew3=d(rw[4].bias), min_std=float(model.min_std),
    )
    spec.validate()
    return spec


def planet_version(model) -> tuple:
    """Freshness token of the planning heads (PlaNetModel.update rewrites them in place, planet.py:485-519)."""
    mods = (model.belief_model, model.prior_transition_model, model.reward_model)
    return tuple(int(p._version) for m in mods for p in m.parameters())
