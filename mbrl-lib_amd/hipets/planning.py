"""Drop-in counterparts of mbrl.planning's trajectory-optimisation classes, backed by libhipets.

Same names, constructor arguments and error behaviour as the reference so that the stock Hydra
configs only swap ``_target_``:

* ``CEMOptimizer``                 <- mbrl/planning/trajectory_opt.py:43-188
* ``TrajectoryOptimizer``          <- mbrl/planning/trajectory_opt.py:490-572
* ``TrajectoryOptimizerAgent``     <- mbrl/planning/trajectory_opt.py:575-716
* ``create_trajectory_optim_agent_for_model`` <- :719-749
* ``make_eval_fn`` / ``HipTrajectoryEvalFn``  <- the closure at :743-748 around
  ``ModelEnv.evaluate_action_sequences`` (mbrl/models/model_env.py:145-191)

There is no CPU fallback anywhere in this module: every optimizer needs a gfx950 device.
"""
from __future__ import annotations

import importlib
import time
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np
import torch

from . import dist as hdist
from ._lib import HipetsError, IcemParams
from .engine import Engine
from .model import (ModelSpec, PlaNetSpec, UnsupportedModelError, is_planet_model, model_version, planet_version,
                    spec_from_model_env, spec_from_planet_model)

_ENGINES: Dict[int, Engine] = {}


def get_engine(device) -> Engine:
    """One Engine per GPU (the reference is single-device, single-threaded)."""
    device = torch.device(device)
    if device.type != "cuda":
        raise HipetsError(f"hipets needs a GPU device, got {device} (there is no CPU fallback)")
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if idx not in _ENGINES:
        _ENGINES[idx] = Engine(torch.device("cuda", idx))
    return _ENGINES[idx]


# ---------------------------------------------------------------------------------------------
# objective: ModelEnv.evaluate_action_sequences on the fused kernel
# ---------------------------------------------------------------------------------------------
class HipTrajectoryEvalFn:
    """``trajectory_eval_fn(initial_state, action_sequences) -> Tensor[B]`` (mbrl/types.py:15).

    Built from a live ``mbrl.models.ModelEnv`` (weights are re-snapshotted whenever
    ``ModelTrainer.train`` changed them) or from a ``ModelSpec``.  Randomness modes:

    * ``'device'`` (THE DEFAULT since round 6): the reference's propagation semantics -- ONE balanced random permutation of all
      ``pop * particles`` rows per step (gaussian_mlp.py:203-205), iid eps per row and dim -- with both drawn
      in-kernel from ``(seed, call counter)`` (a keyed bijection + Philox).  ONE persistent launch for the horizon (rows
      change workgroups through an in-kernel hand-over table); one launch per step where that form does not apply
      (``Engine.set_persistent(False)``, batches beyond two workgroups per CU) -- same bits either way.
    * ``'fast'`` (opt-in; ~8 % faster at cfg2): one launch for the whole horizon; each workgroup (particle p of 16-48 consecutive
      candidates) draws one member per step from a balanced schedule: same marginals, block-wise common random numbers --
      NOT the reference's per-row shuffle (held to the statistical tests only; with fewer than 16-48 candidates several particles
      of one candidate share a member at every step, include/hipets.h hipets_fast_schedule).
    * ``'exact'``: replays the reference's own draws from torch's RNGs in the reference's order (one
      ``randperm(B)`` per step from the global generator, one ``normal_`` per step from ``rng``): seed-identical
      to ``ModelEnv.evaluate_action_sequences`` (a parity aid: it synchronises with the host).
    * ``'exact_device'``: alias of ``'device'`` (kept for round-1 callers; BasicEnsemble models draw their iid
      member maps with torch's device generator).
    """

    def __init__(self, model, num_particles: int, engine: Optional[Engine] = None, mode: str = "device",
                 seed: int = 0, device=None, rng: Optional[torch.Generator] = None):
        if mode not in ("fast", "device", "exact", "exact_device"):
            raise ValueError("mode must be 'fast', 'device', 'exact' or 'exact_device'")
        self.num_particles = int(num_particles)
        self.mode = mode
        self.seed = int(seed)
        self.calls = 0
        self._model_env = None
        self._version = None
        if isinstance(model, ModelSpec):
            spec = model
            dev = device if device is not None else "cuda:0"
        else:
            self._model_env = model
            spec = spec_from_model_env(model)
            dev = device if device is not None else getattr(model, "device", "cuda:0")
            self._version = model_version(model)
            if rng is None:
                rng = getattr(model, "_rng", None)
        self.engine = engine if engine is not None else get_engine(dev)
        self.device = self.engine.device
        self.engine.set_model(spec)
        self.spec = spec
        self._rng = rng
        # for multi-GPU: evaluate only candidates [lo, hi) (set by dist.ShardedEvalFn)

    def refresh(self, force: bool = False):
        """Re-pack weights if the live model changed (mbrl/models/model_trainer.py:288-296)."""
        if self._model_env is None:
            return
        v = model_version(self._model_env)
        if force or v != self._version:
            self.spec = spec_from_model_env(self._model_env)
            self.engine.set_model(self.spec)
            self._version = v

    def _prep(self, action_sequences: torch.Tensor) -> torch.Tensor:
        self.refresh()
        if self.engine.spec is not self.spec:  # engine shared with another eval fn
            self.engine.set_model(self.spec)
        a = action_sequences
        if a.device != self.device or a.dtype != torch.float32 or not a.is_contiguous():
            a = a.to(device=self.device, dtype=torch.float32).contiguous()
        self.check_batch(a.shape[0])
        return a

    @property
    def kernel_mode(self) -> Optional[str]:
        """'fast' / 'device' when the objective draws its randomness in-kernel from (seed, stream_id) -- the modes the
        fused plans can run --, else None."""
        if self.mode == "fast":
            return "fast"
        if self.mode in ("device", "exact_device") and (self.spec.ensemble_kind != "basic_ensemble" or self.spec.propagation == "expectation"):
            return "device"
        return None

    def evaluate_seeded(self, initial_state: np.ndarray, action_sequences: torch.Tensor, seed: int, stream_id: int) -> torch.Tensor:
        """One objective evaluation with explicit counter-based randomness: what iteration ``stream_id`` of a fused plan
        runs, callable from the per-iteration optimizer paths so that both produce the same numbers bit for bit."""
        a = self._prep(action_sequences)
        return self.engine.rollout(a, initial_state, self.num_particles, mode=self.kernel_mode, seed=seed, stream_id=stream_id)

    def __call__(self, initial_state: np.ndarray, action_sequences: torch.Tensor) -> torch.Tensor:
        a = self._prep(action_sequences)
        self.calls += 1
        if self.kernel_mode is not None:
            return self.engine.rollout(a, initial_state, self.num_particles, mode=self.kernel_mode, seed=self.seed, stream_id=self.calls)
        if self.mode in ("device", "exact_device"):
            # BasicEnsemble models only (GaussianMLP models take the in-kernel 'device' mode above): the reference's iid
            # randint member maps (basic_ensemble.py:122-129, 255-260) and eps drawn by torch's device generator
            pop, H, _ = a.shape
            B = pop * self.num_particles
            g = self._device_rng()
            eps = None
            members = None
            M = len(self.spec.members)
            if self.spec.propagation == "random_model":
                members = torch.randint(M, (H, B), device=self.device, generator=g)
            elif self.spec.propagation == "fixed_model":
                members = torch.randint(M, (B,), device=self.device, generator=g)
            if not self.spec.deterministic:
                eps = torch.randn(H, B, self.spec.out_dim, device=self.device, generator=g)
            return self.engine.rollout(a, initial_state, self.num_particles, mode="exact", members=members, eps=eps)
        pop, H, _ = a.shape
        B = pop * self.num_particles
        perms = eps = None
        if self.spec.ensemble_kind == "basic_ensemble":
            # BasicEnsemble draws its member maps with randint FROM THE GENERATOR (basic_ensemble.py:122-129, 255-260):
            # reset -> [fixed_model map], then per step [random_model map], normal
            rng, M = self._cpu_rng(), len(self.spec.members)
            members = torch.randint(M, (B,), generator=rng) if self.spec.propagation == "fixed_model" else None
            m_list, e_list = [], []
            for _ in range(H):
                if self.spec.propagation == "random_model":
                    m_list.append(torch.randint(M, (B,), generator=rng))
                if not self.spec.deterministic:
                    e_list.append(torch.empty(B, self.spec.out_dim).normal_(0.0, 1.0, generator=rng))
            if m_list:
                members = torch.stack(m_list)
            eps = torch.stack(e_list).to(self.device) if e_list else None
            return self.engine.rollout(a, initial_state, self.num_particles, mode="exact", members=members, eps=eps)
        if self.spec.propagation == "fixed_model":
            perms = torch.randperm(B).to(self.device)  # gaussian_mlp.py:375 at reset
        if self.spec.propagation == "random_model" or not self.spec.deterministic:
            p_list, e_list = [], []
            for _ in range(H):  # reference consumption order, SURVEY.md Appendix A.4
                if self.spec.propagation == "random_model":
                    p_list.append(torch.randperm(B))
                if not self.spec.deterministic:
                    e_list.append(torch.empty(B, self.spec.out_dim).normal_(0.0, 1.0, generator=self._cpu_rng()))
            if p_list:
                perms = torch.stack(p_list).to(self.device)
            if e_list:
                eps = torch.stack(e_list).to(self.device)
        return self.engine.rollout(a, initial_state, self.num_particles, mode="exact", perms=perms, eps=eps)

    def check_batch(self, pop: int):
        """The reference's ValueError (gaussian_mlp.py:195-200), raised for every propagation method and kept in
        FAST mode too so that switching engines never changes which configurations are accepted."""
        B, M = pop * self.num_particles, len(self.spec.members)
        if self.spec.ensemble_kind == "basic_ensemble":  # BasicEnsemble.forward has no such rule (basic_ensemble.py:142-196)
            return
        if B % M != 0:
            raise ValueError(
                f"GaussianMLP ensemble requires batch size to be a multiple of the "
                f"number of models. Current batch size is {B} for "
                f"{M} models."
            )

    def _device_rng(self):
        if not hasattr(self, "_dev_rng"):
            self._dev_rng = torch.Generator(device=self.device).manual_seed(self.seed)
        return self._dev_rng

    def _cpu_rng(self):
        if self._rng is not None and self._rng.device.type == "cpu":
            return self._rng
        if not hasattr(self, "_own_rng"):
            self._own_rng = torch.Generator().manual_seed(self.seed)
        return self._own_rng


class ModelEnv:
    """The model-as-environment interface of mbrl/models/model_env.py:15-191 on the fused kernels:
    ``reset`` / ``step`` (one transition for a batch of independent rows: what MBPO-style model rollouts and the
    visualisers call) and ``evaluate_action_sequences``.  Built from a ``ModelSpec`` or a live mbrl ``ModelEnv``."""

    def __init__(self, model, engine: Optional[Engine] = None, mode: str = "device", seed: int = 0, device=None,
                 generator: Optional[torch.Generator] = None):
        """``mode`` as for :class:`HipTrajectoryEvalFn`: 'device' (default; the reference's per-row balanced member shuffle and iid
        eps, drawn in-kernel), 'fast' (one member per workgroup of 16-48 consecutive rows), 'exact' (the reference's own torch draws)."""
        self._eval = HipTrajectoryEvalFn(model, 1, engine=engine, mode=mode, seed=seed, device=device, rng=generator)
        self.engine, self.device, self.mode, self.seed = self._eval.engine, self._eval.device, mode, int(seed)
        self._return_as_np = True
        self._steps = 0
        self._fixed_perm = None
        self._fixed_members = None
        self._fixed_schedule = None
        self._reset_stream = 0

    def _step_mode(self) -> str:
        """Kernel mode of ``step`` for the in-kernel randomness modes: 'device' where the library has it (GaussianMLP ensembles; any
        model under expectation propagation), else 'fast' (BasicEnsemble: iid member draws per workgroup)."""
        if self.mode in ("device", "exact_device") and (self.spec.ensemble_kind != "basic_ensemble" or self.spec.propagation == "expectation"):
            return "device"
        return "fast"

    @property
    def spec(self) -> ModelSpec:
        return self._eval.spec

    def reset(self, initial_obs_batch: np.ndarray, return_as_np: bool = True) -> Dict[str, torch.Tensor]:
        """model_env.py:62-85: returns the model state {"obs", "propagation_indices"}."""
        assert len(initial_obs_batch.shape) == 2  # batch, obs_dim
        self._eval.refresh()
        obs = torch.as_tensor(np.asarray(initial_obs_batch, dtype=np.float32)).to(self.device).contiguous()
        self._return_as_np = return_as_np
        B = obs.shape[0]
        self._eval.num_particles = 1
        self._eval.check_batch(B)
        self._fixed_perm = self._fixed_schedule = None
        self._fixed_members = None
        if self.spec.propagation == "fixed_model":  # model.py:404-407 -> gaussian_mlp.py:363-375
            if self.mode == "exact" and self.spec.ensemble_kind == "basic_ensemble":  # basic_ensemble.py:255-260
                self._fixed_members = torch.randint(len(self.spec.members), (B,), generator=self._eval._cpu_rng())
                return {"obs": obs, "propagation_indices": self._fixed_members}
            if self.mode == "exact":
                self._fixed_perm = torch.randperm(B).to(self.device)
            elif self._step_mode() == "device":
                # the TS-infinity permutation of this rollout: keyed by the stream of the reset, evaluated in-kernel at every step
                # (hipets_rollout_opts.perm_stream_id); exported here as the reference's ``propagation_indices``
                self._reset_stream = self._steps + 1
                self._fixed_perm = self.engine.device_perms(1, B, self.seed, self._reset_stream)
            else:
                nwg, _ = self.engine.fast_geometry(B, 1, 1, -1)  # hipets_step runs the general kernel layout
                self._fixed_schedule = self.engine.fast_schedule(1, nwg, self.seed, self._steps + 1).contiguous()
        return {"obs": obs, "propagation_indices": self._fixed_perm}

    def step(self, actions, model_state: Dict[str, torch.Tensor], sample: bool = False):
        """model_env.py:87-140: (next_observs, rewards, dones, next_model_state)."""
        assert len(actions.shape) == 2  # batch, action_dim
        self._eval.refresh()
        if self.engine.spec is not self.spec:
            self.engine.set_model(self.spec)
        if isinstance(actions, np.ndarray):
            actions = torch.from_numpy(actions)
        actions = actions.to(device=self.device, dtype=torch.float32).contiguous()
        obs = model_state["obs"].to(device=self.device, dtype=torch.float32).contiguous()
        B = obs.shape[0]
        self._steps += 1
        if self.mode == "exact":
            perm = eps = members = None
            basic = self.spec.ensemble_kind == "basic_ensemble"
            if self.spec.propagation == "random_model":
                if basic:  # basic_ensemble.py:122-129 (the generator, before this step's normal)
                    members = torch.randint(len(self.spec.members), (B,), generator=self._eval._cpu_rng())
                else:
                    perm = torch.randperm(B).to(self.device)  # gaussian_mlp.py:205 (global RNG)
            elif self.spec.propagation == "fixed_model":
                perm, members = self._fixed_perm, self._fixed_members
            if sample and not self.spec.deterministic:
                eps = torch.empty(B, self.spec.out_dim).normal_(0.0, 1.0, generator=self._eval._cpu_rng()).to(self.device)
            nobs, rew, done = self.engine.step(obs, actions, mode="exact", sample=sample, perm=perm, eps=eps, members=members)
        elif self._step_mode() == "device":
            fixed = self.spec.propagation == "fixed_model"
            nobs, rew, done = self.engine.step(obs, actions, mode="device", sample=sample, seed=self.seed, stream_id=self._steps,
                                               perm_stream_id=self._reset_stream if fixed else 0)
        else:
            nobs, rew, done = self.engine.step(obs, actions, mode="fast", sample=sample, seed=self.seed, stream_id=self._steps,
                                               member_schedule=self._fixed_schedule)
        next_state = {"obs": nobs, "propagation_indices": model_state.get("propagation_indices")}
        if self._return_as_np:
            return nobs.cpu().numpy(), rew.cpu().numpy(), done.cpu().numpy(), next_state
        return nobs, rew, done, next_state

    def evaluate_action_sequences(self, action_sequences: torch.Tensor, initial_state: np.ndarray, num_particles: int) -> torch.Tensor:
        """model_env.py:145-191."""
        assert len(action_sequences.shape) == 3
        self._eval.num_particles = int(num_particles)
        return self._eval(initial_state, action_sequences)


class UnfusedTrajectoryEvalFn:
    """``trajectory_eval_fn`` for models whose ``reward_fn`` / ``termination_fn`` are arbitrary Python callables
    (SURVEY.md section 2.1 row 6 "documented unfused fallback"): the horizon loop of
    ``ModelEnv.evaluate_action_sequences`` (model_env.py:178-191) runs on the host, every model transition is ONE fused
    ``hipets_step`` launch (input build, ensemble MLP, sampling, delta), and the user's callables run as torch ops on
    the returned device tensors.  ``step_mode='device'`` (default): every step draws the reference's balanced per-row member
    shuffle and iid eps in-kernel; ``'fast'``: one member per workgroup of 16-48 consecutive rows (also what BasicEnsemble
    models run: the library's DEVICE mode has no iid-member variant)."""

    mode = "unfused"

    def __init__(self, model, num_particles: int, reward_fn=None, termination_fn=None, engine: Optional[Engine] = None,
                 seed: int = 0, device=None, step_mode: str = "device"):
        if step_mode not in ("device", "fast"):
            raise ValueError("step_mode must be 'device' or 'fast'")
        self.step_mode = step_mode
        self.num_particles, self.seed, self.calls = int(num_particles), int(seed), 0
        self._model_env, self._version = None, None
        if isinstance(model, ModelSpec):
            spec = model
            dev = device if device is not None else "cuda:0"
        else:
            self._model_env = model
            spec = spec_from_model_env(model, allow_custom_fns=True)
            dev = device if device is not None else getattr(model, "device", "cuda:0")
            self._version = model_version(model)
        self.reward_fn = reward_fn if reward_fn is not None else spec.custom_reward_fn
        self.termination_fn = termination_fn if termination_fn is not None else spec.custom_termination_fn
        self.engine = engine if engine is not None else get_engine(dev)
        self.device = self.engine.device
        self.spec = spec
        self.engine.set_model(spec)

    def refresh(self):
        if self._model_env is not None and model_version(self._model_env) != self._version:
            self.spec = spec_from_model_env(self._model_env, allow_custom_fns=True)
            self.engine.set_model(self.spec)
            self._version = model_version(self._model_env)

    def check_batch(self, pop: int):
        B, M = pop * self.num_particles, len(self.spec.members)
        if self.spec.ensemble_kind == "basic_ensemble":  # BasicEnsemble.forward has no such rule (basic_ensemble.py:142-196)
            return
        if B % M != 0:
            raise ValueError(
                f"GaussianMLP ensemble requires batch size to be a multiple of the "
                f"number of models. Current batch size is {B} for "
                f"{M} models."
            )

    def __call__(self, initial_state: np.ndarray, action_sequences: torch.Tensor) -> torch.Tensor:
        self.refresh()
        if self.engine.spec is not self.spec:
            self.engine.set_model(self.spec)
        a_seq = action_sequences.to(device=self.device, dtype=torch.float32)
        pop, H, _ = a_seq.shape
        P = self.num_particles
        self.check_batch(pop)
        self.calls += 1
        obs = torch.as_tensor(np.asarray(initial_state, np.float32), device=self.device).repeat(pop * P, 1).contiguous()
        total = torch.zeros(pop * P, 1, device=self.device)
        terminated = torch.zeros(pop * P, 1, dtype=torch.bool, device=self.device)
        schedule = None
        device_mode = self.step_mode == "device" and (self.spec.ensemble_kind != "basic_ensemble" or self.spec.propagation == "expectation")
        fixed = self.spec.propagation == "fixed_model"  # TS-infinity: one member map for the whole horizon (model.py:404-407)
        if fixed and not device_mode:
            nwg, _ = self.engine.fast_geometry(pop * P, 1, 1, -1)  # hipets_step runs the general kernel layout
            schedule = self.engine.fast_schedule(1, nwg, self.seed, self.calls * 4096).contiguous()
        for t in range(H):
            act = torch.repeat_interleave(a_seq[:, t, :], P, dim=0).contiguous()  # model_env.py:179-182
            if device_mode:  # (stream ids of a call start at calls * 4096 + 1: 0 means "none" for perm_stream_id)
                nobs, rew, done = self.engine.step(obs, act, mode="device", sample=True, seed=self.seed, stream_id=self.calls * 4096 + 1 + t,
                                                   perm_stream_id=self.calls * 4096 + 1 if fixed else 0)
            else:
                nobs, rew, done = self.engine.step(obs, act, mode="fast", sample=True, seed=self.seed, stream_id=self.calls * 4096 + t,
                                                   member_schedule=schedule)
            if self.reward_fn is not None:
                rew = self.reward_fn(act, nobs)
            if self.termination_fn is not None:
                done = self.termination_fn(act, nobs)
            rew = rew.clone()
            rew[terminated] = 0  # :186
            terminated |= done  # :187
            total += rew  # :188
            obs = nobs
        return total.reshape(-1, P).mean(dim=1)


class PlaNetTrajectoryEvalFn:
    """``trajectory_eval_fn`` for a PlaNet latent model: ``ModelEnv.evaluate_action_sequences`` with
    ``PlaNetModel.sample`` as the transition (mbrl/models/planet.py:531-581, mbrl/algorithms/planet.py), the whole horizon in
    one kernel launch.  Like the reference, the observation argument only fixes the batch size: rollouts start from the
    model's saved posterior sample and belief (``update_posterior``, planet.py:600-640), read from the live model at every
    call, or set with :meth:`set_state` when built from a ``PlaNetSpec``.

    ``mode='device'`` (default; ``'fast'`` is the same thing here): iid standard normals per (row, step, latent dim) drawn
    in-kernel from Philox counters -- a PlaNet model has no ensemble, so there is no member shuffle to approximate and the two
    in-kernel modes of the PETS objective coincide with the reference's semantics; ``mode='exact'``: the reference's draws (one
    ``randn([B, latent])`` per step from the generator) made on the host and injected."""

    def __init__(self, model, num_particles: int = 1, engine: Optional[Engine] = None, mode: str = "device", seed: int = 0,
                 device=None, rng: Optional[torch.Generator] = None):
        if mode not in ("device", "fast", "exact"):
            raise ValueError("mode must be 'device' (= 'fast': in-kernel draws) or 'exact'")
        self.num_particles, self.mode, self.seed, self.calls = int(num_particles), mode, int(seed), 0
        self._planet, self._version, self._state = None, None, None
        if isinstance(model, PlaNetSpec):
            spec, dev = model, (device if device is not None else "cuda:0")
        else:
            planet = getattr(model, "dynamics_model", model)  # a ModelEnv or the PlaNetModel itself
            self._planet = planet
            spec = spec_from_planet_model(planet)
            self._version = planet_version(planet)
            dev = device if device is not None else getattr(planet, "device", "cuda:0")
            if rng is None:
                rng = getattr(model, "_rng", None)
        self.engine = engine if engine is not None else get_engine(dev)
        self.device = self.engine.device
        self.spec = spec
        self._rng = rng
        self.engine.planet_set_model(spec)

    def set_state(self, latent: torch.Tensor, belief: torch.Tensor):
        """The posterior sample s_t and belief h_t rollouts start from ([1, latent] / [1, belief])."""
        self._state = (latent.detach().to(self.device, torch.float32).reshape(-1).contiguous(),
                       belief.detach().to(self.device, torch.float32).reshape(-1).contiguous())

    def refresh(self):
        if self._planet is not None and planet_version(self._planet) != self._version:
            self.spec = spec_from_planet_model(self._planet)
            self.engine.planet_set_model(self.spec)
            self._version = planet_version(self._planet)

    def prepare(self):
        """What a call does before its rollout: re-pack changed weights, make them the engine's PlaNet model, fetch the live
        model's saved posterior sample / belief.  Returns (latent0, belief0)."""
        self.refresh()
        if self.engine.planet_spec is not self.spec:
            self.engine.planet_set_model(self.spec)
        if self._planet is not None:  # planet.py:669-672
            if self._planet._current_posterior_sample is None or self._planet._current_belief is None:
                raise RuntimeError("PlaNetModel has no saved posterior: call update_posterior() before planning")
            self.set_state(self._planet._current_posterior_sample, self._planet._current_belief)
        if self._state is None:
            raise RuntimeError("no latent state: call set_state(latent, belief) first")
        return self._state

    def evaluate_seeded(self, initial_state, action_sequences: torch.Tensor, seed: int, stream_id: int) -> torch.Tensor:
        """One evaluation with explicit counter-based randomness (what iteration ``stream_id`` of the fused plan runs)."""
        latent0, belief0 = self.prepare()
        a = action_sequences
        if a.device != self.device or a.dtype != torch.float32 or not a.is_contiguous():
            a = a.to(device=self.device, dtype=torch.float32).contiguous()
        return self.engine.planet_rollout(a, latent0, belief0, self.num_particles, seed=seed, stream_id=stream_id)

    def __call__(self, initial_state, action_sequences: torch.Tensor) -> torch.Tensor:
        self.refresh()
        if self.engine.planet_spec is not self.spec:
            self.engine.planet_set_model(self.spec)
        if self._planet is not None:  # planet.py:669-672
            if self._planet._current_posterior_sample is None or self._planet._current_belief is None:
                raise RuntimeError("PlaNetModel has no saved posterior: call update_posterior() before planning")
            self.set_state(self._planet._current_posterior_sample, self._planet._current_belief)
        if self._state is None:
            raise RuntimeError("no latent state: call set_state(latent, belief) first")
        a = action_sequences
        if a.device != self.device or a.dtype != torch.float32 or not a.is_contiguous():
            a = a.to(device=self.device, dtype=torch.float32).contiguous()
        self.calls += 1
        latent0, belief0 = self._state
        if self.mode in ("fast", "device"):
            return self.engine.planet_rollout(a, latent0, belief0, self.num_particles, seed=self.seed, stream_id=self.calls)
        pop, H, _ = a.shape
        B = pop * self.num_particles
        if self._rng is None:
            self._rng = torch.Generator().manual_seed(self.seed)
        if self._rng.device.type == "cpu":
            eps = torch.stack([torch.randn(B, self.spec.latent_size, generator=self._rng) for _ in range(H)]).to(self.device)
        else:
            eps = torch.stack([torch.randn(B, self.spec.latent_size, generator=self._rng, device=self._rng.device) for _ in range(H)])
            eps = eps.to(self.device)
        return self.engine.planet_rollout(a, latent0, belief0, self.num_particles, eps=eps.contiguous())


def make_eval_fn(model, num_particles: int, **kw):
    """``agent.set_trajectory_eval_fn(hipets.make_eval_fn(model_env, num_particles))`` on a stock or a
    hipets agent (seam 3 of SURVEY.md section 8b).  Returns the fully fused objective when reward / termination are
    mbrl.env closed forms, the unfused one (fused model step + Python callables) when they are arbitrary callables.
    Without a ``mode=`` argument the objective runs ``mode='device'``: the reference's TS1 semantics (one balanced permutation
    of all rows per step, gaussian_mlp.py:201-211), every draw made in-kernel; ``mode='fast'`` is the opt-in block-balanced variant."""
    if isinstance(model, PlaNetSpec) or is_planet_model(getattr(model, "dynamics_model", model)):
        return PlaNetTrajectoryEvalFn(model, num_particles, **kw)
    try:
        return HipTrajectoryEvalFn(model, num_particles, **kw)
    except UnsupportedModelError:
        if isinstance(model, ModelSpec):
            raise
        spec = spec_from_model_env(model, allow_custom_fns=True)  # raises again if something else is unsupported
        if spec.custom_reward_fn is None and spec.custom_termination_fn is None:
            raise
        kw2 = {k: v for k, v in kw.items() if k in ("engine", "seed", "device")}
        if kw.get("mode") in ("fast", "device"):
            kw2["step_mode"] = kw["mode"]
        return UnfusedTrajectoryEvalFn(model, num_particles, **kw2)


class _BoundObjective:
    """``obj_fun(action_sequences)`` with the observation bound (trajectory_opt.py:680-681); carries the
    engine handle so optimizers can take the fused path."""

    def __init__(self, eval_fn, obs):
        self.eval_fn = eval_fn
        self.obs = obs

    def __call__(self, action_sequences):
        return self.eval_fn(self.obs, action_sequences)


def _fused_target(obj_fun, planet_ok: bool = False):
    """The hipets objective behind ``obj_fun`` when it draws its randomness in-kernel (so that a whole optimisation can run
    inside the library), else None.  ``planet_ok``: PlaNet latent objectives count too (CEM has a fused PlaNet plan)."""
    if isinstance(obj_fun, _BoundObjective):
        fn = obj_fun.eval_fn
        if isinstance(fn, HipTrajectoryEvalFn) and fn.kernel_mode is not None:
            return fn
        if planet_ok and isinstance(fn, PlaNetTrajectoryEvalFn) and fn.mode in ("fast", "device"):
            return fn
    return None


def _prepare_fused(fused: HipTrajectoryEvalFn, population_sizes: Sequence[int]):
    """What ``fused.__call__`` would do before a rollout, for plans that run as one library call: re-pack the
    weights if the live model changed, make them the engine's current model, validate every batch size."""
    fused.refresh()
    for n in population_sizes:
        fused.check_batch(int(n))
    if fused.engine.spec is not fused.spec:
        fused.engine.set_model(fused.spec)
    if fused.engine.plan_mode != fused.kernel_mode:
        fused.engine.set_plan_mode(fused.kernel_mode)


# ---------------------------------------------------------------------------------------------
# optimizers
# ---------------------------------------------------------------------------------------------
class Optimizer:  # trajectory_opt.py:21-40
    def __init__(self):
        pass

    def optimize(self, obj_fun, x0=None, callback=None, **kwargs) -> torch.Tensor:
        raise NotImplementedError


_SEED_COUNTER = [0]


def _default_seed(seed: Optional[int]) -> int:
    """Seed of an optimizer's counter-based streams.  ``None`` derives one from ``torch.initial_seed()`` (what
    ``torch.manual_seed`` set) and a per-process construction counter: reproducible under ``torch.manual_seed`` +
    the same construction order, different for every optimizer built -- WITHOUT consuming torch's global generator (the
    reference's constructors draw nothing: an extra draw here would shift every later reference-order draw, e.g. the
    ``sampler='torch'`` / ``mode='exact'`` replays and model initialisation, by one)."""
    if seed is None:
        _SEED_COUNTER[0] += 1
        z = (int(torch.initial_seed()) + 0x9E3779B97F4A7C15 * _SEED_COUNTER[0]) & (2**64 - 1)  # splitmix64 finaliser
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & (2**64 - 1)
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & (2**64 - 1)
        seed = z ^ (z >> 31)
    return int(seed) & (2**63 - 1)


def _reference_elites(values: torch.Tensor, elite_num: int, device) -> torch.Tensor:
    """The elite indices the reference's optimizers pick on a CPU device (trajectory_opt.py:178-179, 467-470): NaN -> -1e-10, then
    ``torch.topk`` -- whose order among EQUAL values is an artefact of its partial sort.  The 0 / 1 rewards of the cartpole family
    (env/reward_fns.py:10-13, 27-30) tie dozens of candidates at the elite boundary, so a seed-identical replay
    (``sampler='torch'``: it synchronises with the host anyway) has to ask the same routine.  int32 indices on ``device``."""
    v = values.detach().to("cpu", torch.float32).clone()
    v[v.isnan()] = -1e-10
    return torch.topk(v, int(elite_num)).indices.to(torch.int32).to(device).contiguous()


def _reference_noise(shape, clipped_normal: bool) -> torch.Tensor:
    """The standard-normal draws of CEMOptimizer._sample_population on a CPU device, from torch's global generator:
    ``randn`` for the clipped-normal branch (trajectory_opt.py:116-117), otherwise mbrl.util.math.truncated_normal_
    (util/math.py:69-92): N(0, 1), entries outside [-2, 2] redrawn until none is left."""
    if clipped_normal:
        return torch.randn(shape)
    t = torch.zeros(shape)
    torch.nn.init.normal_(t, mean=0.0, std=1.0)
    while True:
        cond = torch.logical_or(t < -2.0, t > 2.0)
        n = int(torch.sum(cond).item())
        if n == 0:
            return t
        t[cond] = torch.normal(0.0, 1.0, size=(n,))


class CEMOptimizer(Optimizer):
    """Cross-Entropy Method with device-side sampling and elite refit (trajectory_opt.py:43-188).

    Works with ANY ``obj_fun`` (generic path: one sample kernel + ``obj_fun`` + one refit kernel per
    iteration, no host synchronisation of its own); when ``obj_fun`` is a hipets objective that draws in-kernel (device or fast mode)
    and no callback is given, the whole optimisation is one ``hipets_plan_cem`` call."""

    def __init__(self, num_iterations: int, elite_ratio: float, population_size: int,
                 lower_bound: Sequence[Sequence[float]], upper_bound: Sequence[Sequence[float]], alpha: float,
                 device: torch.device, return_mean_elites: bool = False, clipped_normal: bool = False,
                 seed: Optional[int] = None, sampler: str = "philox"):
        super().__init__()
        if sampler not in ("philox", "torch"):
            raise ValueError("sampler must be 'philox' (device-side, default) or 'torch' (the reference's draws)")
        # sampler='torch': the population noise is drawn exactly like the reference does on a CPU device (torch's GLOBAL
        # generator, redraw-until-inside loop of mbrl.util.math.truncated_normal_, util/math.py:69-92), so that with the
        # same torch.manual_seed an agent reproduces the reference's action selection (a parity aid: it synchronises)
        self.sampler = sampler
        self.num_iterations = num_iterations
        self.elite_ratio = elite_ratio
        self.population_size = population_size
        self.elite_num = np.ceil(self.population_size * self.elite_ratio).astype(np.int32)  # :89-91
        self.device = torch.device(device)
        self.engine = get_engine(self.device)
        self.device = self.engine.device
        self.lower_bound = torch.tensor(lower_bound, device=self.device, dtype=torch.float32).contiguous()
        self.upper_bound = torch.tensor(upper_bound, device=self.device, dtype=torch.float32).contiguous()
        self.alpha = alpha
        self.return_mean_elites = return_mean_elites
        self._clipped_normal = clipped_normal
        self.seed = _default_seed(seed)
        self.calls = 0
        # the reference's CEM is shape-generic (notebooks/cem_rosenbrock_ex.ipynb optimises a [2] vector):
        # kernels only see the flattened variable; [H, A] bounds keep their meaning for the fused plan path
        if self.lower_bound.ndim == 2:
            H, A = self.lower_bound.shape
        else:
            H, A = int(self.lower_bound.numel()), 1
        self._params = Engine.cem_params(population_size, H, A, num_iterations, int(self.elite_num), alpha,
                                         return_mean_elites, clipped_normal, unbiased_var=True)

    def _init_population_params(self, x0: torch.Tensor):  # :100-108
        mean = x0.clone()
        if self._clipped_normal:
            dispersion = torch.ones_like(mean)
        else:
            dispersion = ((self.upper_bound - self.lower_bound) ** 2) / 16
        return mean, dispersion

    def optimize(self, obj_fun: Callable[[torch.Tensor], torch.Tensor], x0: Optional[torch.Tensor] = None,
                 callback: Optional[Callable[[torch.Tensor, torch.Tensor, int], None]] = None, **kwargs) -> torch.Tensor:
        x0 = x0.to(device=self.device, dtype=torch.float32).contiguous()
        self.calls += 1
        # a hipets objective that draws its randomness in-kernel: iteration i of this call samples AND rolls out with the
        # counter-based streams (seed ^ objective seed, calls * iterations + i), whether the loop runs inside the library
        # (one hipets_plan_cem call) or here (callback / injected noise / force_generic): both give the same numbers
        fused = _fused_target(obj_fun, planet_ok=True) if (x0.ndim == 2 and self.sampler == "philox") else None
        if fused is not None and fused.engine is not self.engine:
            fused = None
        seed = (self.seed ^ fused.seed) if fused is not None else self.seed
        noise = kwargs.get("noise")  # optional injected z per iteration (parity tests)
        if fused is not None and callback is None and noise is None and not kwargs.get("force_generic", False):
            if isinstance(fused, PlaNetTrajectoryEvalFn):  # the PlaNet latent planner: hipets_plan_planet_cem
                latent0, belief0 = fused.prepare()
                return self.engine.plan_planet_cem(self._params, x0, self.lower_bound, self.upper_bound, latent0, belief0,
                                                   fused.num_particles, seed=seed, plan_id=self.calls)
            _prepare_fused(fused, [self.population_size])
            if self.engine.comm_world > 1:  # the engine has a communicator (hipets.dist.init_engine_comm): the ranks share the population
                return hdist.plan_cem_sharded(self.engine, self._params, x0, self.lower_bound, self.upper_bound, obj_fun.obs,
                                              fused.num_particles, seed=seed, plan_id=self.calls, group=self.engine.comm_group)[0]
            return self.engine.plan_cem(self._params, x0, self.lower_bound, self.upper_bound, obj_fun.obs,
                                        fused.num_particles, seed=seed, plan_id=self.calls)
        p = self._params
        mu, dispersion = self._init_population_params(x0)
        mu, dispersion = mu.contiguous(), dispersion.contiguous()
        best_solution = torch.zeros_like(mu)
        best_value = torch.full((1,), -float("inf"), device=self.device, dtype=torch.float32)
        population = torch.empty((self.population_size,) + tuple(x0.shape), device=self.device, dtype=torch.float32)
        for i in range(self.num_iterations):
            stream = self.calls * self.num_iterations + i
            z = None if noise is None else noise[i].to(self.device, torch.float32).contiguous()
            if z is None and self.sampler == "torch":
                z = _reference_noise(tuple(population.shape), self._clipped_normal).to(self.device).contiguous()
            self.engine.cem_sample(p, mu, dispersion, self.lower_bound, self.upper_bound, population, z=z, seed=seed, stream_id=stream)
            values = fused.evaluate_seeded(obj_fun.obs, population, seed, stream) if fused is not None else obj_fun(population)
            if callback is not None:
                callback(population, values, i)
            if values.device != self.device or values.dtype != torch.float32 or not values.is_contiguous():
                values = values.to(device=self.device, dtype=torch.float32).contiguous()
            elites = _reference_elites(values, self.elite_num, self.device) if self.sampler == "torch" else None
            self.engine.cem_refit(p, values, population, mu, dispersion, best_value, best_solution, elites=elites)
        return mu if self.return_mean_elites else best_solution


class MPPIOptimizer(Optimizer):
    """Model Predictive Path Integral optimizer (trajectory_opt.py:191-311) with device-side sampling, smoothing
    recurrence and importance-weighted update.  Reproduces the reference's behaviour including its quirks
    (SURVEY.md Appendix B4-B6): ``self.mean`` persists across calls and is NOT cleared by ``agent.reset()``;
    ``past_action`` aliases the already-shifted ``mean[0]``; ``sigma`` never reaches the population."""

    def __init__(self, num_iterations: int, population_size: int, gamma: float, sigma: float, beta: float,
                 lower_bound: Sequence[Sequence[float]], upper_bound: Sequence[Sequence[float]], device: torch.device,
                 seed: Optional[int] = None, sampler: str = "philox"):
        super().__init__()
        if sampler not in ("philox", "torch"):
            raise ValueError("sampler must be 'philox' (device-side, default) or 'torch' (the reference's draws)")
        self.sampler = sampler  # 'torch': noise like the reference (global generator, truncated_normal_, :262-271)
        self.planning_horizon = len(lower_bound)
        self.population_size = population_size
        self.action_dimension = len(lower_bound[0])
        self.engine = get_engine(device)
        self.device = self.engine.device
        self.mean = torch.zeros((self.planning_horizon, self.action_dimension), device=self.device, dtype=torch.float32)
        self.lower_bound = torch.tensor(lower_bound, device=self.device, dtype=torch.float32).contiguous()
        self.upper_bound = torch.tensor(upper_bound, device=self.device, dtype=torch.float32).contiguous()
        self.var = sigma**2 * torch.ones_like(self.lower_bound)  # kept for API parity; dead in the reference too
        self.beta = beta
        self.gamma = gamma
        self.refinements = num_iterations
        self.seed = _default_seed(seed)
        self.calls = 0

    def optimize(self, obj_fun: Callable[[torch.Tensor], torch.Tensor], x0: Optional[torch.Tensor] = None,
                 callback: Optional[Callable[[torch.Tensor, torch.Tensor, int], None]] = None, **kwargs) -> torch.Tensor:
        H, A, pop = self.planning_horizon, self.action_dimension, self.population_size
        self.calls += 1
        fused = _fused_target(obj_fun) if self.sampler == "philox" else None  # see CEMOptimizer.optimize
        if fused is not None and fused.engine is not self.engine:
            fused = None
        seed = (self.seed ^ fused.seed) if fused is not None else self.seed
        noise = kwargs.get("noise")
        if fused is not None and callback is None and noise is None and not kwargs.get("force_generic", False):
            _prepare_fused(fused, [pop])
            self.mean = self.mean.contiguous()
            if self.engine.comm_world > 1:  # sharded over the engine's communicator; the persistent mean stays replicated bit for bit
                hdist.plan_mppi_sharded(self.engine, pop, H, A, self.refinements, self.gamma, self.beta, self.mean, self.lower_bound,
                                        self.upper_bound, obj_fun.obs, fused.num_particles, seed=seed, plan_id=self.calls,
                                        group=self.engine.comm_group)
                return self.mean.clone()
            self.engine.plan_mppi(pop, H, A, self.refinements, self.gamma, self.beta, self.mean, self.lower_bound, self.upper_bound,
                                  obj_fun.obs, fused.num_particles, seed=seed, plan_id=self.calls)
            return self.mean.clone()
        shifted = self.mean.clone()
        shifted[:-1] = self.mean[1:]  # :258
        self.mean = shifted.contiguous()
        past_action = self.mean[0].clone()  # :257 (a view of the shifted tensor; constant across refinements)
        population = torch.empty((pop, H, A), device=self.device, dtype=torch.float32)
        for k in range(self.refinements):
            stream = self.calls * self.refinements + k
            z = None if noise is None else noise[k].to(self.device, torch.float32).contiguous()
            if z is None and self.sampler == "torch":
                z = _reference_noise((pop, H, A), False).to(self.device).contiguous()
            self.engine.mppi_sample(pop, H, A, self.beta, self.mean, past_action, self.lower_bound, self.upper_bound, population,
                                    z=z, seed=seed, stream_id=stream)
            values = fused.evaluate_seeded(obj_fun.obs, population, seed, stream) if fused is not None else obj_fun(population)
            if values.device != self.device or values.dtype != torch.float32 or not values.is_contiguous():
                values = values.to(device=self.device, dtype=torch.float32).contiguous()
            if callback is not None:  # the reference calls back after the NaN filter here (:297-300)
                values[values.isnan()] = -1e-10
                callback(population, values, k)
            new_mean = torch.empty_like(self.mean)
            self.engine.mppi_update(pop, H, A, self.gamma, values, population, new_mean)
            self.mean = new_mean
        return self.mean.clone()


class ICEMOptimizer(Optimizer):
    """Improved CEM (trajectory_opt.py:314-487): decaying population, coloured-noise sampling (device-side inverse
    real DFT), kept / shifted elites, biased variance refit.  ``self.elite`` persists across calls (Appendix B6)."""

    def __init__(self, num_iterations: int, elite_ratio: float, population_size: int, population_decay_factor: float,
                 colored_noise_exponent: float, lower_bound: Sequence[Sequence[float]], upper_bound: Sequence[Sequence[float]],
                 keep_elite_frac: float, alpha: float, device: torch.device, return_mean_elites: bool = False,
                 population_size_module: Optional[int] = None, seed: Optional[int] = None, sampler: str = "philox"):
        super().__init__()
        if sampler not in ("philox", "torch"):
            raise ValueError("sampler must be 'philox' (device-side, default) or 'torch' (the reference's draws)")
        # 'torch': every draw of an iteration comes from torch's global CPU generator in the reference's order -- the two
        # spectrum normals of powerlaw_psd_gaussian (util/math.py:372-377), randperm(elite_num) for the kept elites
        # (trajectory_opt.py:446-448), the tail-action normal of the shifted elites (:451-457)
        self.sampler = sampler
        self.num_iterations = num_iterations
        self.elite_ratio = elite_ratio
        self.population_size = population_size
        self.population_decay_factor = population_decay_factor
        self.elite_num = np.ceil(self.population_size * self.elite_ratio).astype(np.int32)
        self.colored_noise_exponent = colored_noise_exponent
        self.engine = get_engine(device)
        self.device = self.engine.device
        self.lower_bound = torch.tensor(lower_bound, device=self.device, dtype=torch.float32).contiguous()
        self.upper_bound = torch.tensor(upper_bound, device=self.device, dtype=torch.float32).contiguous()
        self.initial_var = ((self.upper_bound - self.lower_bound) ** 2) / 16
        self.keep_elite_frac = keep_elite_frac
        self.keep_elite_size = np.ceil(keep_elite_frac * self.elite_num).astype(np.int32)
        self.elite = None
        self.alpha = alpha
        self.return_mean_elites = return_mean_elites
        self.population_size_module = population_size_module
        if self.population_size_module:
            self.keep_elite_size = self._round_up_to_module(self.keep_elite_size, self.population_size_module)
        self.seed = _default_seed(seed)
        self.calls = 0

    @staticmethod
    def _round_up_to_module(value: int, module: int) -> int:  # :385-389
        if value % module == 0:
            return value
        return value + (module - value % module)

    def _iteration_size(self, i: int) -> int:  # :419-431
        n = np.ceil(np.max((self.population_size * self.population_decay_factor**-i, 2 * self.elite_num))).astype(np.int32)
        if self.population_size_module:
            n = self._round_up_to_module(n, self.population_size_module)
        return int(n)

    def optimize(self, obj_fun: Callable[[torch.Tensor], torch.Tensor], x0: Optional[torch.Tensor] = None,
                 callback: Optional[Callable[[torch.Tensor, torch.Tensor, int], None]] = None, **kwargs) -> torch.Tensor:
        eng = self.engine
        x0 = x0.to(device=self.device, dtype=torch.float32).contiguous()
        H, A = x0.shape
        K, keep = int(self.elite_num), int(self.keep_elite_size)
        self.calls += 1
        fused = _fused_target(obj_fun) if self.sampler == "philox" else None  # see CEMOptimizer.optimize
        if fused is not None and fused.engine is not self.engine:
            fused = None
        seed = (self.seed ^ fused.seed) if fused is not None else self.seed
        if fused is not None and callback is None and kwargs.get("inject") is None and not kwargs.get("force_generic", False):
            sizes = []
            for i in range(self.num_iterations):
                extra = 0
                if self.elite is not None or i > 0:
                    extra = 1 if (i == self.num_iterations - 1 and i != 0) else keep
                sizes.append(self._iteration_size(i) + extra)
            _prepare_fused(fused, sizes)
            p = IcemParams(population_size=int(self.population_size), horizon=H, act_dim=A, num_iterations=int(self.num_iterations),
                           elite_num=K, keep_elite_size=keep, population_size_module=int(self.population_size_module or 0),
                           return_mean_elites=int(bool(self.return_mean_elites)), alpha=float(self.alpha),
                           population_decay_factor=float(self.population_decay_factor),
                           colored_noise_exponent=float(self.colored_noise_exponent))
            has_elite = self.elite is not None
            elite = self.elite.contiguous() if has_elite else torch.empty((K, H, A), device=self.device, dtype=torch.float32)
            if eng.comm_world > 1:  # sharded over the engine's communicator; the persistent elites stay replicated bit for bit
                out = hdist.plan_icem_sharded(eng, p, x0, self.lower_bound, self.upper_bound, elite, has_elite, obj_fun.obs, fused.num_particles,
                                              seed=seed, plan_id=self.calls, keep_idx=kwargs.get("keep_idx"), group=eng.comm_group)[0]
            else:
                out = eng.plan_icem(p, x0, self.lower_bound, self.upper_bound, elite, has_elite, obj_fun.obs, fused.num_particles,
                                    seed=seed, plan_id=self.calls, keep_idx=kwargs.get("keep_idx"))
            if self.num_iterations > 0:
                self.elite = elite
            return out
        mu = x0.clone()
        var = self.initial_var.clone().contiguous()
        best_solution = torch.zeros_like(mu)
        best_value = torch.full((1,), -float("inf"), device=self.device, dtype=torch.float32)
        elite_idx = torch.empty(K, dtype=torch.int32, device=self.device)
        inject = kwargs.get("inject")
        for i in range(self.num_iterations):
            n = self._iteration_size(i)
            inj = inject[i] if inject is not None else {}
            if inject is None and self.sampler == "torch":
                F = H // 2 + 1
                inj = {"normals": torch.stack([torch.empty(n, A, F).normal_(0.0, 1.0), torch.empty(n, A, F).normal_(0.0, 1.0)])}
                if self.elite is not None:
                    inj["keep_perm"] = torch.randperm(K)
                    if i == 0:
                        inj["end_noise"] = torch.empty(keep, A).normal_(0.0, 1.0)
            sid = (self.calls * self.num_iterations + i) * 4
            extra = 0
            if self.elite is not None:
                extra = 1 if (i == self.num_iterations - 1 and i != 0) else keep
            population = torch.empty((n + extra, H, A), device=self.device, dtype=torch.float32)
            normals = inj.get("normals")
            if normals is not None:
                normals = normals.to(self.device, torch.float32).contiguous()
            eng.icem_sample(n, H, A, self.colored_noise_exponent, mu, var, self.lower_bound, self.upper_bound, population,
                            normals=normals, seed=seed, stream_id=sid)
            if self.elite is not None:
                if "keep_perm" in inj:
                    perm = inj["keep_perm"].to(self.device)
                else:  # torch.randperm(elite_num)[:keep] (:446-448): index plumbing, stays a torch op
                    perm = torch.randperm(K, device=self.device)
                kept = torch.index_select(self.elite, dim=0, index=perm[:keep]).contiguous()
                if i == 0:  # :450-462
                    en = inj.get("end_noise")
                    if en is not None:
                        en = en.to(self.device, torch.float32).contiguous()
                    eng.icem_shift(kept.shape[0], H, A, kept, mu, var, population[n:], end_noise=en, seed=seed,
                                   stream_id=sid + 1)
                elif i == self.num_iterations - 1:  # :463-464
                    population[n:] = mu.unsqueeze(0)
                else:  # :465-466
                    population[n:] = kept
            values = fused.evaluate_seeded(obj_fun.obs, population, seed, sid + 3) if fused is not None else obj_fun(population)
            if callback is not None:
                callback(population, values, i)
            if values.device != self.device or values.dtype != torch.float32 or not values.is_contiguous():
                values = values.to(device=self.device, dtype=torch.float32).contiguous()
            p = Engine.cem_params(population.shape[0], H, A, self.num_iterations, K, self.alpha, self.return_mean_elites,
                                  clipped_normal=False, unbiased_var=False)  # biased variance (:479)
            elites = _reference_elites(values, K, self.device) if (self.sampler == "torch" and inject is None) else None
            eng.cem_refit(p, values, population, mu, var, best_value, best_solution, elite_idx, elites=elites)
            new_elite = torch.empty((K, H, A), device=self.device, dtype=torch.float32)
            eng.gather_rows(population, elite_idx, new_elite)  # self.elite = population[elite_idx] (:476)
            self.elite = new_elite
        return mu if self.return_mean_elites else best_solution


# ---------------------------------------------------------------------------------------------
# TrajectoryOptimizer / Agent
# ---------------------------------------------------------------------------------------------
_TARGET_ALIASES = {
    # stock targets are redirected to the fused implementations when an agent of this module builds them
    "mbrl.planning.CEMOptimizer": "hipets.planning.CEMOptimizer",
    "mbrl.planning.trajectory_opt.CEMOptimizer": "hipets.planning.CEMOptimizer",
    "mbrl.planning.ICEMOptimizer": "hipets.planning.ICEMOptimizer",
    "mbrl.planning.trajectory_opt.ICEMOptimizer": "hipets.planning.ICEMOptimizer",
    "mbrl.planning.MPPIOptimizer": "hipets.planning.MPPIOptimizer",
    "mbrl.planning.trajectory_opt.MPPIOptimizer": "hipets.planning.MPPIOptimizer",
    # conf/algorithm/pets.yaml:5 handed to hipets.create_trajectory_optim_agent_for_model unchanged
    "mbrl.planning.TrajectoryOptimizerAgent": "hipets.planning.TrajectoryOptimizerAgent",
    "mbrl.planning.trajectory_opt.TrajectoryOptimizerAgent": "hipets.planning.TrajectoryOptimizerAgent",
}


def _cfg_to_dict(cfg) -> dict:
    """Top-level keys of a plain dict or an OmegaConf ``DictConfig`` as a dict.  OmegaConf raises ``MissingMandatoryValue``
    (not a KeyError) when a key that holds ``???`` is read -- and the stock configs ship ``lower_bound: ???``,
    ``upper_bound: ???``, ``action_lb: ???``, ``action_ub: ???`` (conf/action_optimizer/*.yaml, conf/algorithm/pets.yaml)
    -- so missing values are returned as the string "???" and filtered by the callers, like hydra's instantiate is fed by
    the reference only after it has written the bounds into the config (trajectory_opt.py:525-527, core.py:101-106)."""
    try:
        from omegaconf import OmegaConf  # real OmegaConf: resolves interpolations too

        if OmegaConf.is_config(cfg):
            return dict(OmegaConf.to_container(cfg, resolve=True, throw_on_missing=False))
    except ImportError:
        pass
    out = {}
    for k in list(cfg.keys()):
        try:
            out[k] = cfg[k]
        except Exception as exc:  # omegaconf.errors.MissingMandatoryValue of a DictConfig-like object
            if type(exc).__name__ != "MissingMandatoryValue":
                raise
            out[k] = "???"
    return out


def _is_missing(v) -> bool:
    return isinstance(v, str) and v == "???"


def _instantiate(cfg, **overrides):
    """A minimal ``_target_`` resolver (object construction only; the reference does exactly this through
    hydra.utils.instantiate at trajectory_opt.py:527,741).  Works on plain dicts and OmegaConf nodes; placeholders
    ("???") that no override filled are dropped so the target's own defaults / errors apply."""
    kwargs = _cfg_to_dict(cfg)
    kwargs.update(overrides)
    target = kwargs.pop("_target_")
    target = _TARGET_ALIASES.get(target, target)
    kwargs = {k: v for k, v in kwargs.items() if not _is_missing(v)}
    mod, _, name = target.rpartition(".")
    return getattr(importlib.import_module(mod), name)(**kwargs)


class _OptimizerSnapshot:
    """What one ``optimizer.optimize`` call changes besides returning a plan -- the counter-based stream positions
    (``calls`` of the optimizer and of a hipets objective) and the state that persists across plans (MPPI ``mean``, iCEM
    ``elite``: SURVEY.md Appendix B6) -- so that a plan whose rollouts were cut short can be re-run as if it never ran."""

    def __init__(self, optimizer, obj_fun):
        self.optimizer = optimizer
        self.eval_fn = getattr(obj_fun, "eval_fn", obj_fun)
        inner = getattr(self.eval_fn, "eval_fn", None)  # dist.ShardedEvalFn wraps the hipets objective
        self.counters = [o for o in (optimizer, self.eval_fn, inner) if isinstance(getattr(o, "calls", None), int)]
        self.calls = [o.calls for o in self.counters]
        self.state = {k: (getattr(optimizer, k).clone() if torch.is_tensor(getattr(optimizer, k)) else getattr(optimizer, k))
                      for k in ("mean", "elite") if hasattr(optimizer, k)}
        self.engines = []
        for o in (optimizer, self.eval_fn, inner):
            eng = getattr(o, "engine", None)
            if isinstance(eng, Engine) and eng not in self.engines:
                self.engines.append(eng)

    def engines_report_timeout(self) -> bool:
        """Did a persistent DEVICE-mode rollout of the plan give up on THIS rank -- or, when the ranks plan in lockstep, on ANY rank?
        With a ``dist.ShardedEvalFn`` objective every iteration is a host-side collective all ranks must take part in: a rank that
        re-ran its plan alone would issue a second series of all-gathers its peers never match.  The flag is therefore all-reduced
        over the objective's group first, and the plan is re-run on every rank or on none.  (The fused sharded plans agree inside
        ``hipets.dist.run_sharded`` and have consumed the flag by the time this is asked.)"""
        hit = False
        for eng in self.engines:
            hit = eng.check_async_error() or hit
        if isinstance(self.eval_fn, hdist.ShardedEvalFn) and hdist.is_distributed():
            hit = bool(hdist._worst_status(int(hit), self.eval_fn.group))
        return hit

    def restore(self):
        for o, c in zip(self.counters, self.calls):
            o.calls = c
        for k, v in self.state.items():
            setattr(self.optimizer, k, v.clone() if torch.is_tensor(v) else v)


class TrajectoryOptimizer:
    """trajectory_opt.py:490-572: tiles the action bounds over the horizon, instantiates the optimizer,
    warm-starts each call from the previous solution shifted by ``replan_freq``."""

    def __init__(self, optimizer_cfg, action_lb: np.ndarray, action_ub: np.ndarray, planning_horizon: int,
                 replan_freq: int = 1, keep_last_solution: bool = True):
        lower = np.tile(action_lb, (planning_horizon, 1)).tolist()  # :525
        upper = np.tile(action_ub, (planning_horizon, 1)).tolist()  # :526
        self.optimizer: Optimizer = _instantiate(optimizer_cfg, lower_bound=lower, upper_bound=upper)  # :527
        device = self.optimizer.device
        self.initial_solution = ((torch.tensor(action_lb) + torch.tensor(action_ub)) / 2).float().to(device)
        self.initial_solution = self.initial_solution.repeat((planning_horizon, 1))
        self.previous_solution = self.initial_solution.clone()
        self.replan_freq = replan_freq
        self.keep_last_solution = keep_last_solution
        self.horizon = planning_horizon

    def optimize(self, trajectory_eval_fn: Callable[[torch.Tensor], torch.Tensor],
                 callback: Optional[Callable] = None) -> np.ndarray:
        """(A plan that is re-run after a timed-out rollout -- see below -- invokes ``callback`` again for every iteration of the
        second run: a callback that accumulates sees the iterations of the voided attempt followed by those of the valid one.)"""
        snapshot = _OptimizerSnapshot(self.optimizer, trajectory_eval_fn)
        best_solution = self.optimizer.optimize(trajectory_eval_fn, x0=self.previous_solution, callback=callback)
        plan = best_solution.cpu().numpy()  # the one device->host sync of a plan (:568)
        # Everything the plan enqueued has executed now.  If a persistent DEVICE-mode rollout inside it gave up waiting for
        # another workgroup's rows (CUs taken by another process: hipets.h, hipets_check_async_error) the plan was built on
        # invalid returns: never hand it out.  The engine has switched to per-step launches, which return the same bits the
        # persistent form would have: put the optimizer back where it was and run the SAME plan again.
        for _ in range(2):
            if not snapshot.engines_report_timeout():
                break
            snapshot.restore()
            best_solution = self.optimizer.optimize(trajectory_eval_fn, x0=self.previous_solution, callback=callback)
            plan = best_solution.cpu().numpy()
        else:
            if snapshot.engines_report_timeout():
                raise HipetsError("DEVICE-mode rollouts keep timing out although persistent launches are off")
        if self.keep_last_solution:  # :563-567
            self.previous_solution = best_solution.roll(-self.replan_freq, dims=0)
            self.previous_solution[-self.replan_freq:] = self.initial_solution[0]
        return plan

    def reset(self):
        self.previous_solution = self.initial_solution.clone()


class Agent:  # mbrl/planning/core.py:18-49
    def act(self, obs: np.ndarray, **_kwargs) -> np.ndarray:
        raise NotImplementedError

    def plan(self, obs: np.ndarray, **_kwargs) -> np.ndarray:
        return self.act(obs, **_kwargs)

    def reset(self):
        pass


class TrajectoryOptimizerAgent(Agent):
    """trajectory_opt.py:575-716 with the same public methods (``set_trajectory_eval_fn``, ``reset``,
    ``act``, ``plan``) and the same RuntimeError when no objective was set (:673-676)."""

    def __init__(self, optimizer_cfg, action_lb: Sequence[float], action_ub: Sequence[float], planning_horizon: int = 1,
                 replan_freq: int = 1, verbose: bool = False, keep_last_solution: bool = True):
        self.optimizer = TrajectoryOptimizer(optimizer_cfg, np.array(action_lb), np.array(action_ub),
                                             planning_horizon=planning_horizon, replan_freq=replan_freq,
                                             keep_last_solution=keep_last_solution)
        self.optimizer_args = {"optimizer_cfg": optimizer_cfg, "action_lb": np.array(action_lb),
                               "action_ub": np.array(action_ub)}
        self.trajectory_eval_fn = None
        self.actions_to_use: List[np.ndarray] = []
        self.replan_freq = replan_freq
        self.verbose = verbose

    def set_trajectory_eval_fn(self, trajectory_eval_fn):
        self.trajectory_eval_fn = trajectory_eval_fn

    def reset(self, planning_horizon: Optional[int] = None):
        if planning_horizon:  # :644-651
            old = self.optimizer.optimizer
            self.optimizer = TrajectoryOptimizer(self.optimizer_args["optimizer_cfg"], self.optimizer_args["action_lb"],
                                                 self.optimizer_args["action_ub"], planning_horizon=planning_horizon,
                                                 replan_freq=self.replan_freq)
            # the rebuilt optimizer continues the old one's counter-based streams (same seed, call counter carried over)
            # instead of replaying them from plan 1: the reference's global generator keeps advancing across resets too
            new = self.optimizer.optimizer
            if hasattr(old, "calls") and hasattr(new, "calls"):
                new.calls = old.calls
                if hasattr(old, "seed") and _cfg_to_dict(self.optimizer_args["optimizer_cfg"]).get("seed") is None:
                    new.seed = old.seed
        self.optimizer.reset()

    def _require_eval_fn(self):
        if self.trajectory_eval_fn is None:
            raise RuntimeError("Please call `set_trajectory_eval_fn()` before using TrajectoryOptimizerAgent")

    def act(self, obs: np.ndarray, optimizer_callback: Optional[Callable] = None, **_kwargs) -> np.ndarray:
        self._require_eval_fn()
        plan_time = 0.0
        if not self.actions_to_use:  # re-plan is necessary (:678)
            start_time = time.time()
            plan = self.optimizer.optimize(_BoundObjective(self.trajectory_eval_fn, obs), callback=optimizer_callback)
            plan_time = time.time() - start_time
            self.actions_to_use.extend([a for a in plan[: self.replan_freq]])
        action = self.actions_to_use.pop(0)
        if self.verbose:
            print(f"Planning time: {plan_time:.3f}")
        return action

    def plan(self, obs: np.ndarray, **_kwargs) -> np.ndarray:
        self._require_eval_fn()
        return self.optimizer.optimize(_BoundObjective(self.trajectory_eval_fn, obs))


class BatchedCEMAgent(Agent):
    """Batched planning: one CEM plan per environment for ``n_env`` environments (vectorised envs,
    MPC for many agents) in ONE set of launches.  Same algorithm per environment as ``TrajectoryOptimizerAgent`` +
    ``CEMOptimizer`` (warm start shifted by ``replan_freq`` per environment, trajectory_opt.py:563-567); a single cfg2
    plan leaves 36 of 256 CUs idle, a batch fills the chip.  The rollouts run the objective's randomness mode: 'device' (default:
    one balanced permutation per step over the rows of ALL environments -- every row meets every member with probability 1 / M and
    the members stay exactly balanced, as in a single reference plan) or 'fast'."""

    def __init__(self, eval_fn: HipTrajectoryEvalFn, n_env: int, action_lb: Sequence[float], action_ub: Sequence[float],
                 planning_horizon: int, num_iterations: int, elite_ratio: float, population_size: int, alpha: float,
                 return_mean_elites: bool = True, clipped_normal: bool = False, replan_freq: int = 1, seed: int = 0):
        if eval_fn.kernel_mode is None:
            raise ValueError("batched planning needs an objective with in-kernel randomness (mode='device' or 'fast')")
        self.eval_fn, self.engine, self.device = eval_fn, eval_fn.engine, eval_fn.device
        self.n_env, self.horizon, self.replan_freq = int(n_env), int(planning_horizon), int(replan_freq)
        lb, ub = np.asarray(action_lb, np.float32), np.asarray(action_ub, np.float32)
        A = lb.shape[0]
        self.lower = torch.tensor(np.tile(lb, (planning_horizon, 1)), device=self.device).contiguous()
        self.upper = torch.tensor(np.tile(ub, (planning_horizon, 1)), device=self.device).contiguous()
        self.initial_solution = torch.tensor((lb + ub) / 2, device=self.device).repeat(self.n_env, planning_horizon, 1).contiguous()
        self.previous_solution = self.initial_solution.clone()
        self.elite_num = int(np.ceil(population_size * elite_ratio))
        self._params = Engine.cem_params(population_size, planning_horizon, A, num_iterations, self.elite_num, alpha,
                                         return_mean_elites, clipped_normal, unbiased_var=True)
        self.seed, self.calls = int(seed), 0

    def reset(self):
        self.previous_solution = self.initial_solution.clone()

    def plan(self, obs_batch: np.ndarray, **_kwargs) -> np.ndarray:
        obs_batch = np.asarray(obs_batch, dtype=np.float32)
        assert obs_batch.shape[0] == self.n_env
        self.eval_fn.refresh()
        if self.engine.spec is not self.eval_fn.spec:
            self.engine.set_model(self.eval_fn.spec)
        self.eval_fn.check_batch(self._params.population_size)
        if self.engine.plan_mode != self.eval_fn.kernel_mode:
            self.engine.set_plan_mode(self.eval_fn.kernel_mode)
        self.calls += 1
        best = self.engine.plan_cem(self._params, self.previous_solution, self.lower, self.upper, obs_batch,
                                    self.eval_fn.num_particles, seed=self.seed ^ self.eval_fn.seed, plan_id=self.calls,
                                    n_env=self.n_env)
        self.previous_solution = best.roll(-self.replan_freq, dims=1)
        self.previous_solution[:, -self.replan_freq:] = self.initial_solution[:, :1]
        self.previous_solution = self.previous_solution.contiguous()
        return best.cpu().numpy()

    def act(self, obs_batch: np.ndarray, **_kwargs) -> np.ndarray:
        """One action per environment, [n_env, A]."""
        return self.plan(obs_batch)[:, 0]


class BatchedMPPIAgent(Agent):
    """Batched planning with MPPI: ``MPPIOptimizer.optimize`` (trajectory_opt.py:238-311) for ``n_env``
    environments in one set of launches (hipets_plan_mppi_batched).  Every environment keeps its own persistent mean,
    shifted one step per plan like the reference's (Appendix B4-B6)."""

    def __init__(self, eval_fn: HipTrajectoryEvalFn, n_env: int, action_lb: Sequence[float], action_ub: Sequence[float],
                 planning_horizon: int, num_iterations: int, population_size: int, gamma: float, sigma: float, beta: float,
                 seed: int = 0):
        if eval_fn.kernel_mode is None:
            raise ValueError("batched planning needs an objective with in-kernel randomness (mode='device' or 'fast')")
        self.eval_fn, self.engine, self.device = eval_fn, eval_fn.engine, eval_fn.device
        self.n_env, self.horizon = int(n_env), int(planning_horizon)
        lb, ub = np.asarray(action_lb, np.float32), np.asarray(action_ub, np.float32)
        self.act_dim = int(lb.shape[0])
        self.lower = torch.tensor(np.tile(lb, (planning_horizon, 1)), device=self.device).contiguous()
        self.upper = torch.tensor(np.tile(ub, (planning_horizon, 1)), device=self.device).contiguous()
        self.mean = torch.zeros(self.n_env, self.horizon, self.act_dim, device=self.device)
        self.refinements, self.population_size, self.gamma, self.sigma, self.beta = int(num_iterations), int(population_size), gamma, sigma, beta
        self.seed, self.calls = int(seed), 0

    def plan(self, obs_batch: np.ndarray, **_kwargs) -> np.ndarray:
        obs_batch = np.asarray(obs_batch, dtype=np.float32)
        assert obs_batch.shape[0] == self.n_env
        _prepare_fused(self.eval_fn, [self.population_size])
        self.calls += 1
        self.engine.plan_mppi(self.population_size, self.horizon, self.act_dim, self.refinements, self.gamma, self.beta, self.mean,
                              self.lower, self.upper, obs_batch, self.eval_fn.num_particles, seed=self.seed ^ self.eval_fn.seed,
                              plan_id=self.calls, n_env=self.n_env)
        return self.mean.cpu().numpy()

    def act(self, obs_batch: np.ndarray, **_kwargs) -> np.ndarray:
        return self.plan(obs_batch)[:, 0]


class BatchedICEMAgent(Agent):
    """Batched planning with iCEM: ``ICEMOptimizer.optimize`` (trajectory_opt.py:391-487) for ``n_env``
    environments in one set of launches (hipets_plan_icem_batched): per-environment mean / variance / persistent elites,
    warm start shifted by ``replan_freq`` per environment (trajectory_opt.py:563-567)."""

    def __init__(self, eval_fn: HipTrajectoryEvalFn, n_env: int, action_lb: Sequence[float], action_ub: Sequence[float],
                 planning_horizon: int, num_iterations: int, elite_ratio: float, population_size: int, population_decay_factor: float,
                 colored_noise_exponent: float, keep_elite_frac: float, alpha: float, return_mean_elites: bool = True,
                 population_size_module: Optional[int] = None, replan_freq: int = 1, seed: int = 0):
        if eval_fn.kernel_mode is None:
            raise ValueError("batched planning needs an objective with in-kernel randomness (mode='device' or 'fast')")
        self.eval_fn, self.engine, self.device = eval_fn, eval_fn.engine, eval_fn.device
        self.n_env, self.horizon, self.replan_freq = int(n_env), int(planning_horizon), int(replan_freq)
        lb, ub = np.asarray(action_lb, np.float32), np.asarray(action_ub, np.float32)
        A = int(lb.shape[0])
        self.lower = torch.tensor(np.tile(lb, (planning_horizon, 1)), device=self.device).contiguous()
        self.upper = torch.tensor(np.tile(ub, (planning_horizon, 1)), device=self.device).contiguous()
        self.initial_solution = torch.tensor((lb + ub) / 2, device=self.device).repeat(self.n_env, planning_horizon, 1).contiguous()
        self.previous_solution = self.initial_solution.clone()
        # sizes exactly as ICEMOptimizer computes them (:363-389)
        self._opt = ICEMOptimizer(num_iterations, elite_ratio, population_size, population_decay_factor, colored_noise_exponent,
                                  self.lower.tolist(), self.upper.tolist(), keep_elite_frac, alpha, self.device,
                                  return_mean_elites=return_mean_elites, population_size_module=population_size_module, seed=seed)
        K, keep = int(self._opt.elite_num), int(self._opt.keep_elite_size)
        self._params = IcemParams(population_size=int(population_size), horizon=self.horizon, act_dim=A, num_iterations=int(num_iterations),
                                  elite_num=K, keep_elite_size=keep, population_size_module=int(population_size_module or 0),
                                  return_mean_elites=int(bool(return_mean_elites)), alpha=float(alpha),
                                  population_decay_factor=float(population_decay_factor), colored_noise_exponent=float(colored_noise_exponent))
        self.elite = torch.empty(self.n_env, K, self.horizon, A, device=self.device)
        self.has_elite = False
        self.seed, self.calls = int(seed), 0

    def reset(self):
        self.previous_solution = self.initial_solution.clone()  # the elites persist, like ICEMOptimizer.elite (Appendix B6)

    def plan(self, obs_batch: np.ndarray, keep_idx: Optional[torch.Tensor] = None, **_kwargs) -> np.ndarray:
        obs_batch = np.asarray(obs_batch, dtype=np.float32)
        assert obs_batch.shape[0] == self.n_env
        o, iters = self._opt, self._params.num_iterations
        sizes = []
        for i in range(iters):
            extra = 0
            if self.has_elite or i > 0:
                extra = 1 if (i == iters - 1 and i != 0) else int(o.keep_elite_size)
            sizes.append(o._iteration_size(i) + extra)
        _prepare_fused(self.eval_fn, sizes)
        self.calls += 1
        best = self.engine.plan_icem(self._params, self.previous_solution, self.lower, self.upper, self.elite, self.has_elite, obs_batch,
                                     self.eval_fn.num_particles, seed=self.seed ^ self.eval_fn.seed, plan_id=self.calls, keep_idx=keep_idx,
                                     n_env=self.n_env)
        if iters > 0:
            self.has_elite = True
        self.previous_solution = best.roll(-self.replan_freq, dims=1)
        self.previous_solution[:, -self.replan_freq:] = self.initial_solution[:, :1]
        self.previous_solution = self.previous_solution.contiguous()
        return best.cpu().numpy()

    def act(self, obs_batch: np.ndarray, **_kwargs) -> np.ndarray:
        return self.plan(obs_batch)[:, 0]


def complete_agent_cfg(env, agent_cfg):
    """The subset of mbrl/planning/core.py:71-123 a trajectory-optimizer agent config needs: fill
    ``action_lb`` / ``action_ub`` placeholders ("???") from the action space.  Works on plain dicts and on
    OmegaConf DictConfigs (whose "???" values raise MissingMandatoryValue when read)."""
    have = _cfg_to_dict(agent_cfg)
    if "action_lb" in have and _is_missing(have["action_lb"]):
        agent_cfg["action_lb"] = env.action_space.low.tolist()
    if "action_ub" in have and _is_missing(have["action_ub"]):
        agent_cfg["action_ub"] = env.action_space.high.tolist()
    return agent_cfg


def create_trajectory_optim_agent_for_model(model_env, agent_cfg, num_particles: int = 1, **eval_kw):
    """trajectory_opt.py:719-749, with the objective bound to the fused kernel."""
    complete_agent_cfg(model_env, agent_cfg)
    agent = _instantiate(agent_cfg)
    agent.set_trajectory_eval_fn(make_eval_fn(model_env, num_particles, **eval_kw))
    return agent
