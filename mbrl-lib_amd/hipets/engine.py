"""Engine: thin Python owner of one ``hipets_engine`` (one per GPU).  PyTorch is used only for
device memory and streams; every computation is a libhipets call."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np
import torch

from . import _lib
from ._lib import CemParams, HipetsError, ModelDesc, PlanTrace, RolloutOpts
from .model import ModelSpec


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream(device) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _check_dev(t: torch.Tensor, dtype, device, name: str, shape=None, numel=None):
    if not isinstance(t, torch.Tensor) or t.device != device:
        raise ValueError(f"{name} must be a tensor on {device}")
    if t.dtype != dtype:
        raise ValueError(f"{name} must have dtype {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise ValueError(f"{name} must have shape {tuple(shape)}, got {tuple(t.shape)}")
    if numel is not None and t.numel() != numel:
        raise ValueError(f"{name} must have {numel} elements, got {t.numel()}")


def member_slots(members: torch.Tensor, n_members: int, device) -> tuple:
    """Row -> member map(s) ``members`` int64 [T, B] -> (slots int64 [T, M * rpm] on ``device``, rpm): slot m * rpm + j holds
    the j-th row (ascending) of member m, -1 pads the tail (hipets_rollout_opts.rows_per_member).  Pure index plumbing
    with torch ops on whichever device ``members`` lives: a CPU map gets the tight rpm = largest member count; a device
    map uses rpm = B so that no count has to travel to the host (all-padding workgroups exit immediately)."""
    T, B = members.shape
    members = members.to(torch.int64)
    counts = torch.zeros(T, n_members, dtype=torch.int64, device=members.device).scatter_add_(1, members, torch.ones_like(members))
    rpm = int(counts.max()) if members.device.type == "cpu" else B
    order = members.argsort(dim=1, stable=True)
    sorted_m = members.gather(1, order)
    starts = counts.cumsum(1) - counts
    pos = torch.arange(B, device=members.device).unsqueeze(0) - starts.gather(1, sorted_m)
    slots = torch.full((T, n_members * rpm), -1, dtype=torch.int64, device=members.device)
    slots.scatter_(1, sorted_m * rpm + pos, order)
    return slots.to(device).contiguous(), max(rpm, 1)


class Engine:
    """One fused planning engine bound to ``device`` (a gfx950 GPU).  Not thread-safe."""

    def __init__(self, device="cuda:0"):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise HipetsError("hipets.Engine needs a GPU device (libhipets has no CPU fallback)")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self._lib = _lib.load()
        h = C.c_void_p()
        _lib.check(self._lib.hipets_create(self.device.index, C.byref(h)))
        self._h = h
        self.spec: Optional[ModelSpec] = None
        self.planet_spec = None
        self.comm_world, self.comm_rank = 1, 0
        self.comm_group = None  # torch.distributed group of the communicator's ranks (hipets.dist.init_engine_comm)
        self.plan_mode = "fast"  # (the library's initial value; every fused plan sets its objective's mode: planning._prepare_fused)
        self._trace = None
        self._keep = []  # device tensors that must outlive async set_model work

    def close(self):
        if getattr(self, "_h", None):
            self._lib.hipets_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- model ------------------------------------------------------------------------------
    def set_model(self, spec: ModelSpec):
        spec.validate()
        dev = self.device
        ws = [w.detach().to(device=dev, dtype=torch.float32).contiguous() for w in spec.weights]
        bs = [b.detach().to(device=dev, dtype=torch.float32).contiguous() for b in spec.biases]
        n = len(ws)
        members = spec.members
        d = ModelDesc()
        d.obs_dim, d.act_dim, d.in_dim, d.out_dim = spec.obs_dim, spec.act_dim, spec.in_dim, spec.out_dim
        d.hid, d.n_layers, d.ensemble_size, d.n_members = spec.hid, n, spec.ensemble_size, len(members)
        mem_arr = (C.c_int32 * len(members))(*members)
        d.members = mem_arr
        d.activation = _lib.ACT[spec.activation]
        d.leaky_slope = float(spec.leaky_slope)
        d.propagation = _lib.PROP[spec.propagation]
        d.deterministic = int(spec.deterministic)
        d.obs_process = _lib.OBS[spec.obs_process]
        d.reward_fn = _lib.REW[spec.reward]
        d.termination_fn = _lib.TERM[spec.termination]
        d.target_is_delta = int(spec.target_is_delta)
        d.learned_rewards = int(spec.learned_rewards)
        nd = [int(i) for i in spec.no_delta_list]
        nd_arr = (C.c_int32 * max(1, len(nd)))(*nd)
        d.n_no_delta, d.no_delta = len(nd), nd_arr
        if spec.norm_mean is not None:
            d.normalizer = _lib.NORM["f64" if spec.norm_mean.dtype == torch.float64 else "f32"]
            nm = np.ascontiguousarray(spec.norm_mean.detach().cpu().double().numpy().reshape(-1))
            ns = np.ascontiguousarray(spec.norm_std.detach().cpu().double().numpy().reshape(-1))
            d.norm_mean = nm.ctypes.data_as(C.POINTER(C.c_double))
            d.norm_std = ns.ctypes.data_as(C.POINTER(C.c_double))
        else:
            d.normalizer = _lib.NORM["none"]
        basic = spec.ensemble_kind == "basic_ensemble"
        d.ensemble_kind = _lib.ENSEMBLE[spec.ensemble_kind]
        d.precision = _lib.PREC[spec.precision]
        if not spec.deterministic:
            lo_t, hi_t = spec.min_logvar.detach().cpu().float(), spec.max_logvar.detach().cpu().float()
            if basic:  # every member owns its bounds: [M, out] (a shared [1, out] is broadcast)
                lo_t = lo_t.reshape(-1, spec.out_dim).expand(len(members), spec.out_dim)
                hi_t = hi_t.reshape(-1, spec.out_dim).expand(len(members), spec.out_dim)
            lo = np.ascontiguousarray(lo_t.numpy().reshape(-1))
            hi = np.ascontiguousarray(hi_t.numpy().reshape(-1))
            d.min_logvar = lo.ctypes.data_as(C.POINTER(C.c_float))
            d.max_logvar = hi.ctypes.data_as(C.POINTER(C.c_float))
        w_arr = (C.c_void_p * n)(*[w.data_ptr() for w in ws])
        b_arr = (C.c_void_p * n)(*[b.data_ptr() for b in bs])
        d.weights, d.biases = w_arr, b_arr
        with torch.cuda.device(dev):
            _lib.check(self._lib.hipets_set_model(self._h, C.byref(d), _stream(dev)))
        self.spec = spec

    # ---- ModelEnv.evaluate_action_sequences ------------------------------------------------------
    def rollout(self, actions: torch.Tensor, s0: np.ndarray, num_particles: int, *, mode: str = "fast",
                perms: Optional[torch.Tensor] = None, eps: Optional[torch.Tensor] = None, seed: int = 0,
                stream_id: int = 0, member_schedule: Optional[torch.Tensor] = None,
                trace_next_obs: Optional[torch.Tensor] = None, trace_rewards: Optional[torch.Tensor] = None,
                rows_per_group: int = 0, out: Optional[torch.Tensor] = None,
                phase_cycles: Optional[torch.Tensor] = None, n_env: int = 1,
                members: Optional[torch.Tensor] = None, generic_kernel=False) -> torch.Tensor:
        """``members`` (EXACT mode): int64 [H, B] (random_model) or [B] (fixed_model) active-member slot of every row: the
        reference's ``torch.randint`` draws for BasicEnsemble models (basic_ensemble.py:122-129, 255-260), or any
        ``propagate_from_indices``-style assignment (util/math.py:180-196) for GaussianMLP models (no batch % members rule)."""
        if self.spec is None:
            raise HipetsError("Engine.set_model() has not been called")
        dev = self.device
        if actions.ndim != 3:
            raise ValueError("action_sequences must be [B, H, A]")
        _check_dev(actions, torch.float32, dev, "action_sequences")
        pop, H, A = actions.shape
        if A != self.spec.act_dim:
            raise ValueError(f"action dim {A} != model act_dim {self.spec.act_dim}")
        s0 = np.ascontiguousarray(np.asarray(s0, dtype=np.float32).reshape(-1))
        if s0.shape[0] != self.spec.obs_dim * max(1, n_env):
            raise ValueError(f"initial_state has {s0.shape[0]} values, expected n_env x obs_dim = {max(1, n_env)} x {self.spec.obs_dim}")
        if n_env > 1 and (mode not in ("fast", "device") or pop % n_env):
            raise ValueError("batched rollouts (n_env > 1) need mode='fast' or 'device' and a population divisible by n_env")
        B = pop * num_particles
        o = RolloutOpts()
        o.n_env = int(n_env)
        o.generic_kernel = int(generic_kernel)  # False / True (1: fully generic instance only) / 2 (hidden-static allowed, shape-specialised not)
        if mode not in _lib.MODES:
            raise ValueError("mode must be 'exact', 'fast' or 'device'")
        o.mode = _lib.MODES[mode]
        if mode == "device" and (perms is not None or eps is not None or members is not None):
            raise ValueError("mode='device' draws its permutations and eps in-kernel: perms / eps / members must be None")
        if members is not None:
            if mode != "exact" or perms is not None:
                raise ValueError("members= (explicit per-row member maps) is an EXACT-mode input, exclusive with perms=")
            want = (B,) if self.spec.propagation == "fixed_model" else (H, B)
            if tuple(members.shape) != want:
                raise ValueError(f"members must have shape {want}")
            perms, o.rows_per_member = member_slots(members.reshape(-1, B), len(self.spec.members), dev)
        elif perms is not None:
            _check_dev(perms, torch.int64, dev, "perms")
            want = (B,) if self.spec.propagation == "fixed_model" else (H, B)
            if tuple(perms.shape) != want:
                raise ValueError(f"perms must have shape {want}")
        if eps is not None:
            _check_dev(eps, torch.float32, dev, "eps", (H, B, self.spec.out_dim))
        if mode == "exact":
            o.perms, o.eps = _ptr(perms), _ptr(eps)
        elif mode == "fast":
            o.fast_eps = _ptr(eps)
            if member_schedule is not None:
                nwg, _ = self.fast_geometry(pop, num_particles, H, rows_per_group)
                _check_dev(member_schedule, torch.int32, dev, "member_schedule", (H, nwg))
                o.member_schedule, o.member_schedule_len = _ptr(member_schedule), int(member_schedule.numel())
        o.seed, o.stream_id = int(seed) & (2**64 - 1), int(stream_id) & (2**64 - 1)
        if trace_next_obs is not None:
            _check_dev(trace_next_obs, torch.float32, dev, "trace_next_obs", (H, B, self.spec.obs_dim))
            o.trace_next_obs = _ptr(trace_next_obs)
        if trace_rewards is not None:
            _check_dev(trace_rewards, torch.float32, dev, "trace_rewards", (H, B))
            o.trace_rewards = _ptr(trace_rewards)
        o.rows_per_group = int(rows_per_group)
        if phase_cycles is not None:
            _check_dev(phase_cycles, torch.int64, dev, "phase_cycles", (8, 16))
            o.phase_cycles = _ptr(phase_cycles)
        if out is None:
            out = torch.empty(pop, dtype=torch.float32, device=dev)
        else:
            _check_dev(out, torch.float32, dev, "out", (pop,))
        with torch.cuda.device(dev):
            _lib.check(self._lib.hipets_rollout(self._h, _ptr(actions), s0.ctypes.data_as(C.c_void_p), pop, H,
                                                num_particles, C.byref(o), _ptr(out), _stream(dev)))
        return out

    def step(self, obs: torch.Tensor, actions: torch.Tensor, *, mode: str = "fast", sample: bool = True,
             perm: Optional[torch.Tensor] = None, eps: Optional[torch.Tensor] = None, seed: int = 0, stream_id: int = 0,
             member_schedule: Optional[torch.Tensor] = None, rows_per_group: int = 0,
             members: Optional[torch.Tensor] = None, perm_stream_id: int = 0):
        """One model transition for B independent rows (ModelEnv.step, mbrl/models/model_env.py:87-140).
        Returns (next_obs [B,obs], rewards [B,1], dones [B,1] bool) on the device.  ``members`` int64 [B]: EXACT-mode
        member of every row for BasicEnsemble models (GaussianMLP models take ``perm``).  ``perm_stream_id`` (DEVICE mode,
        fixed_model propagation): the stream whose TS-infinity permutation the step uses -- the rollout's reset -- while
        ``stream_id`` keys this step's eps; 0 = ``stream_id``."""
        if self.spec is None:
            raise HipetsError("Engine.set_model() has not been called")
        dev = self.device
        B = int(obs.shape[0])
        _check_dev(obs, torch.float32, dev, "obs", (B, self.spec.obs_dim))
        _check_dev(actions, torch.float32, dev, "actions", (B, self.spec.act_dim))
        o = RolloutOpts()
        if mode not in _lib.MODES:
            raise ValueError("mode must be 'exact', 'fast' or 'device'")
        o.mode = _lib.MODES[mode]
        o.seed, o.stream_id = int(seed) & (2**64 - 1), int(stream_id) & (2**64 - 1)
        o.rows_per_group = int(rows_per_group)
        o.no_sample = int(not sample)
        o.perm_stream_id = int(perm_stream_id) & (2**64 - 1)
        if mode == "device":
            if perm is not None or eps is not None or members is not None:
                raise ValueError("mode='device' draws its permutation and eps in-kernel")
        elif mode == "exact":
            if members is not None:
                if perm is not None or tuple(members.shape) != (B,):
                    raise ValueError("members= must be int64 [B], exclusive with perm=")
                perm, o.rows_per_member = member_slots(members.reshape(1, B), len(self.spec.members), dev)
                o.perms = _ptr(perm)
            elif perm is not None:
                _check_dev(perm, torch.int64, dev, "perm", (B,))
                o.perms = _ptr(perm)
            if sample and not self.spec.deterministic:
                if eps is None:
                    raise ValueError("EXACT step with sample=True needs eps [B, out_dim]")
                _check_dev(eps, torch.float32, dev, "eps", numel=B * self.spec.out_dim)
                o.eps = _ptr(eps)
        else:
            if eps is not None:
                _check_dev(eps, torch.float32, dev, "eps", numel=B * self.spec.out_dim)
                o.fast_eps = _ptr(eps)
            if member_schedule is not None:
                _check_dev(member_schedule, torch.int32, dev, "member_schedule")
                o.member_schedule, o.member_schedule_len = _ptr(member_schedule), int(member_schedule.numel())
        next_obs = torch.empty_like(obs)
        rewards = torch.empty(B, dtype=torch.float32, device=dev)
        dones = torch.empty(B, dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            _lib.check(self._lib.hipets_step(self._h, _ptr(obs), _ptr(actions), B, C.byref(o), _ptr(next_obs), _ptr(rewards),
                                             _ptr(dones), _stream(dev)))
        return next_obs, rewards.view(B, 1), dones.view(B, 1).bool()

    def fast_geometry(self, pop: int, num_particles: int, horizon: int, rows_per_group: int = 0):
        nwg, r = C.c_int32(), C.c_int32()
        _lib.check(self._lib.hipets_fast_geometry(self._h, pop, num_particles, horizon, rows_per_group, C.byref(nwg),
                                                  C.byref(r)))
        return nwg.value, r.value

    def kernel_class(self, pop: int, num_particles: int, horizon: int, mode: str = "device", rows_per_group: int = 0):
        """(class name, row tiles per workgroup) of the rollout-kernel instance a default rollout / fused plan of this size runs on
        the engine's model: "generic", "hidden_static", "fused" or "wide" (include/hipets.h, hipets_kernel_class); ``rows_per_group``
        > 0: the answer for a call that forces that row-tile count.  Diagnostic."""
        cls, r = C.c_int32(), C.c_int32()
        _lib.check(self._lib.hipets_kernel_class(self._h, pop, num_particles, horizon, _lib.MODES[mode], int(rows_per_group), C.byref(cls),
                                                 C.byref(r)))
        return _lib.KERNEL_CLASSES[cls.value], r.value

    def fast_schedule(self, horizon: int, n_workgroups: int, seed: int = 0, stream_id: int = 0) -> torch.Tensor:
        """The member schedule a FAST rollout with (seed, stream_id) uses: int32 [H, n_workgroups]."""
        out = torch.empty(horizon, n_workgroups, dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.hipets_fast_schedule(self._h, horizon, n_workgroups, int(seed) & (2**64 - 1),
                                                      int(stream_id) & (2**64 - 1), _ptr(out), _stream(self.device)))
        return out

    def fast_normals(self, horizon: int, batch: int, seed: int = 0, stream_id: int = 0) -> torch.Tensor:
        """The eps a FAST rollout with (seed, stream_id) draws: f32 [H, B, out_dim]."""
        out = torch.empty(horizon, batch, self.spec.out_dim, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.hipets_fast_normals(self._h, horizon, batch, int(seed) & (2**64 - 1),
                                                     int(stream_id) & (2**64 - 1), _ptr(out), _stream(self.device)))
        return out

    def device_perms(self, horizon: int, batch: int, seed: int = 0, stream_id: int = 0) -> torch.Tensor:
        """The permutations a DEVICE-mode rollout with (seed, stream_id) uses, in the reference's convention
        (``model_shuffle_indices``): int64 [H, B] for random_model, [B] for fixed_model."""
        fixed = self.spec.propagation == "fixed_model"
        out = torch.empty(1 if fixed else horizon, batch, dtype=torch.int64, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.hipets_device_perms(self._h, horizon, batch, int(seed) & (2**64 - 1), int(stream_id) & (2**64 - 1),
                                                     _ptr(out), _stream(self.device)))
        return out[0] if fixed else out

    def set_plan_mode(self, mode: str):
        """Randomness mode of the rollouts inside the fused plans: 'fast' or 'device'."""
        if mode not in ("fast", "device"):
            raise ValueError("plan mode must be 'fast' or 'device'")
        _lib.check(self._lib.hipets_set_plan_mode(self._h, _lib.MODES[mode]))
        self.plan_mode = mode

    def set_persistent(self, on: bool = True):
        """Allow (default) or forbid the persistent one-launch form of DEVICE-mode rollouts (hipets_set_persistent)."""
        _lib.check(self._lib.hipets_set_persistent(self._h, int(bool(on))))

    def synchronize(self):
        """Wait for everything enqueued on this engine's device (torch.cuda.synchronize)."""
        torch.cuda.synchronize(self.device)

    def set_handover_timeout(self, seconds: float = 0.2):
        """Bound of every hand-over poll of the persistent DEVICE-mode form (hipets_set_handover_timeout)."""
        _lib.check(self._lib.hipets_set_handover_timeout(self._h, float(seconds)))

    def check_async_error(self) -> bool:
        """True if a persistent DEVICE-mode rollout enqueued on this engine gave up waiting for another workgroup's rows
        (hipets_check_async_error): call after the results reached the host; on True they are invalid, the engine has
        switched to per-step launches and the call must be re-run (``TrajectoryOptimizer.optimize`` does)."""
        flag = C.c_int32(0)
        _lib.check(self._lib.hipets_check_async_error(self._h, C.byref(flag)))
        return bool(flag.value)

    def set_plan_trace(self, iters: int = 0, max_rows: int = 0, horizon: int = 0, act_dim: int = 0, elite_num: int = 0, n_env: int = 1):
        """Record the following fused plans iteration by iteration (hipets_set_plan_trace); returns the dict of device
        tensors the library writes into.  ``iters=0`` switches recording off.  ``max_rows`` counts the candidates of ALL
        environments of a batched plan."""
        if iters <= 0:
            _lib.check(self._lib.hipets_set_plan_trace(self._h, None))
            self._trace = None
            return None
        dev = self.device
        tr = {"populations": torch.zeros(iters, max_rows, horizon, act_dim, device=dev),
              "values": torch.zeros(iters, max_rows, device=dev),
              "mus": torch.zeros((iters, horizon, act_dim) if n_env == 1 else (iters, n_env, horizon, act_dim), device=dev),
              "dispersions": torch.zeros((iters, horizon, act_dim) if n_env == 1 else (iters, n_env, horizon, act_dim), device=dev),
              "elite_idx": torch.zeros((iters, max(1, elite_num)) if n_env == 1 else (iters, n_env, max(1, elite_num)), dtype=torch.int32,
                                       device=dev)}
        t = PlanTrace()
        t.populations, t.values, t.mus = tr["populations"].data_ptr(), tr["values"].data_ptr(), tr["mus"].data_ptr()
        t.dispersions, t.elite_idx, t.max_rows = tr["dispersions"].data_ptr(), tr["elite_idx"].data_ptr(), int(max_rows)
        _lib.check(self._lib.hipets_set_plan_trace(self._h, C.byref(t)))
        self._trace = tr  # keeps the buffers alive while the library holds their addresses
        return tr

    # ---- optimizer pieces ---------------------------------------------------------------------------
    @staticmethod
    def cem_params(population_size, horizon, act_dim, num_iterations, elite_num, alpha, return_mean_elites=False,
                   clipped_normal=False, unbiased_var=True) -> CemParams:
        p = CemParams()
        p.population_size, p.horizon, p.act_dim = int(population_size), int(horizon), int(act_dim)
        p.num_iterations, p.elite_num, p.alpha = int(num_iterations), int(elite_num), float(alpha)
        p.return_mean_elites, p.clipped_normal, p.unbiased_var = int(return_mean_elites), int(clipped_normal), int(unbiased_var)
        return p

    def cem_sample(self, p: CemParams, mu, dispersion, lower, upper, population, z=None, seed=0, stream_id=0):
        dev = self.device
        D = p.horizon * p.act_dim
        for n_, t in (("mu", mu), ("dispersion", dispersion), ("lower", lower), ("upper", upper)):
            _check_dev(t, torch.float32, dev, n_, numel=D)
        _check_dev(population, torch.float32, dev, "population", numel=p.population_size * D)
        if z is not None:
            _check_dev(z, torch.float32, dev, "z", numel=p.population_size * D)
        with torch.cuda.device(dev):
            _lib.check(self._lib.hipets_cem_sample(self._h, C.byref(p), _ptr(mu), _ptr(dispersion), _ptr(lower), _ptr(upper),
                                                   _ptr(z), int(seed) & (2**64 - 1), int(stream_id) & (2**64 - 1),
                                                   _ptr(population), _stream(dev)))
        return population

    def cem_refit(self, p: CemParams, values, population, mu, dispersion, best_value, best_solution, elite_idx=None, elites=None):
        """``elites`` int32 [elite_num] (device): refit on THESE candidates, best first, instead of the kernel's own top-k
        (hipets_cem_refit_elites: the reference-order parity mode passes torch.topk's indices, ties and all)."""
        dev = self.device
        D = p.horizon * p.act_dim
        _check_dev(values, torch.float32, dev, "values", (p.population_size,))
        _check_dev(population, torch.float32, dev, "population", numel=p.population_size * D)
        for n_, t in (("mu", mu), ("dispersion", dispersion), ("best_solution", best_solution)):
            _check_dev(t, torch.float32, dev, n_, numel=D)
        _check_dev(best_value, torch.float32, dev, "best_value", (1,))
        if elite_idx is not None:
            _check_dev(elite_idx, torch.int32, dev, "elite_idx", (p.elite_num,))
        if elites is not None:
            _check_dev(elites, torch.int32, dev, "elites", (p.elite_num,))
            # (this path follows a host-side torch.topk, i.e. it has synchronised with the host already: the range check is free)
            lo, hi = int(elites.min()), int(elites.max())
            if lo < 0 or hi >= p.population_size:
                raise ValueError(f"elites must index the population [0, {p.population_size}): got {lo} .. {hi}")
            with torch.cuda.device(dev):
                _lib.check(self._lib.hipets_cem_refit_elites(self._h, C.byref(p), _ptr(values), _ptr(population), _ptr(elites), _ptr(mu),
                                                             _ptr(dispersion), _ptr(best_value), _ptr(best_solution), _stream(dev)))
            if elite_idx is not None:
                elite_idx.copy_(elites)
            return
        with torch.cuda.device(dev):
            _lib.check(self._lib.hipets_cem_refit(self._h, C.byref(p), _ptr(values), _ptr(population), _ptr(mu), _ptr(dispersion),
                                                  _ptr(best_value), _ptr(best_solution), _ptr(elite_idx), _stream(dev)))

    def gather_rows(self, src: torch.Tensor, index: torch.Tensor, out: torch.Tensor):
        dev = self.device
        rows, dim = int(index.numel()), int(out.numel() // max(1, index.numel()))
        _check_dev(src, torch.float32, dev, "src")
        _check_dev(index, torch.int32, dev, "index")
        _check_dev(out, torch.float32, dev, "out", numel=rows * dim)
        with torch.cuda.device(dev):
            _lib.check(self._lib.hipets_gather_rows(self._h, rows, dim, _ptr(src), _ptr(index), _ptr(out), _stream(dev)))
        return out

    def mppi_sample(self, pop, H, A, beta, mean, past_action, lower, upper, population, z=None, seed=0, stream_id=0):
        dev = self.device
        for n_, t in (("mean", mean), ("lower", lower), ("upper", upper)):
            _check_dev(t, torch.float32, dev, n_, numel=H * A)
        _check_dev(past_action, torch.float32, dev, "past_action", numel=A)
        _check_dev(population, torch.float32, dev, "population", numel=pop * H * A)
        if z is not None:
            _check_dev(z, torch.float32, dev, "z", numel=pop * H * A)
        with torch.cuda.device(dev):
            _lib.check(self._lib.hipets_mppi_sample(self._h, pop, H, A, float(beta), _ptr(mean), _ptr(past_action), _ptr(lower),
                                                    _ptr(upper), _ptr(z), int(seed) & (2**64 - 1), int(stream_id) & (2**64 - 1),
                                                    _ptr(population), _stream(dev)))
        return population

    def mppi_update(self, pop, H, A, gamma, values, population, mean):
        dev = self.device
        _check_dev(values, torch.float32, dev, "values", (pop,))
        _check_dev(population, torch.float32, dev, "population", numel=pop * H * A)
        _check_dev(mean, torch.float32, dev, "mean", numel=H * A)
        with torch.cuda.device(dev):
            _lib.check(self._lib.hipets_mppi_update(self._h, pop, H, A, float(gamma), _ptr(values), _ptr(population), _ptr(mean),
                                                    _stream(dev)))
        return mean

    def icem_sample(self, n, H, A, exponent, mu, var, lower, upper, population, normals=None, seed=0, stream_id=0):
        dev = self.device
        for n_, t in (("mu", mu), ("var", var), ("lower", lower), ("upper", upper)):
            _check_dev(t, torch.float32, dev, n_, numel=H * A)
        _check_dev(population, torch.float32, dev, "population")
        if population.numel() < n * H * A:
            raise ValueError("population buffer too small")
        if normals is not None:
            _check_dev(normals, torch.float32, dev, "normals", (2, n, A, H // 2 + 1))
        with torch.cuda.device(dev):
            _lib.check(self._lib.hipets_icem_sample(self._h, n, H, A, float(exponent), _ptr(mu), _ptr(var), _ptr(lower), _ptr(upper),
                                                    _ptr(normals), int(seed) & (2**64 - 1), int(stream_id) & (2**64 - 1),
                                                    _ptr(population), _stream(dev)))
        return population

    def icem_shift(self, keep, H, A, kept, mu, var, out, end_noise=None, seed=0, stream_id=0):
        dev = self.device
        _check_dev(kept, torch.float32, dev, "kept", numel=keep * H * A)
        _check_dev(out, torch.float32, dev, "out", numel=keep * H * A)
        for n_, t in (("mu", mu), ("var", var)):
            _check_dev(t, torch.float32, dev, n_, numel=H * A)
        if end_noise is not None:
            _check_dev(end_noise, torch.float32, dev, "end_noise", numel=keep * A)
        with torch.cuda.device(dev):
            _lib.check(self._lib.hipets_icem_shift(self._h, keep, H, A, _ptr(kept), _ptr(mu), _ptr(var), _ptr(end_noise),
                                                   int(seed) & (2**64 - 1), int(stream_id) & (2**64 - 1), _ptr(out), _stream(dev)))
        return out

    def plan_cem(self, p: CemParams, x0, lower, upper, s0: np.ndarray, num_particles: int, seed: int = 0,
                 plan_id: int = 0, out: Optional[torch.Tensor] = None, n_env: int = 1) -> torch.Tensor:
        """Whole CEM plan on the device.  n_env > 1: x0 / out are [n_env, H, A], s0 is [n_env, obs_dim]; p.population_size is
        per environment (hipets_plan_cem_batched)."""
        if self.spec is None:
            raise HipetsError("Engine.set_model() has not been called")
        dev = self.device
        shp = (p.horizon, p.act_dim)
        _check_dev(x0, torch.float32, dev, "x0", numel=n_env * p.horizon * p.act_dim)
        xshp = tuple(x0.shape)  # [H, A] or [n_env, H, A]
        for n_, t in (("lower", lower), ("upper", upper)):
            _check_dev(t, torch.float32, dev, n_, shp)
        s0 = np.ascontiguousarray(np.asarray(s0, dtype=np.float32).reshape(-1))
        if s0.shape[0] != self.spec.obs_dim * n_env:
            raise ValueError(f"initial_state has {s0.shape[0]} values, expected {n_env} x {self.spec.obs_dim}")
        if out is None:
            out = torch.empty(xshp, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(self._lib.hipets_plan_cem_batched(self._h, C.byref(p), int(n_env), _ptr(x0), _ptr(lower), _ptr(upper),
                                                         s0.ctypes.data_as(C.c_void_p), num_particles, int(seed) & (2**64 - 1),
                                                         int(plan_id) & (2**64 - 1), _ptr(out), _stream(dev)))
        return out

    def plan_mppi(self, pop: int, H: int, A: int, num_iterations: int, gamma: float, beta: float, mean: torch.Tensor, lower,
                  upper, s0: np.ndarray, num_particles: int, seed: int = 0, plan_id: int = 0, n_env: int = 1) -> torch.Tensor:
        """Whole MPPI plan on the device (hipets_plan_mppi[_batched]).  ``mean`` [H, A] ([n_env, H, A] for a batch of
        environments, ``s0`` then [n_env, obs_dim]) is the optimizer's persistent mean: shifted and refined IN PLACE."""
        if self.spec is None:
            raise HipetsError("Engine.set_model() has not been called")
        dev = self.device
        _check_dev(mean, torch.float32, dev, "mean", numel=n_env * H * A)
        for n_, t in (("lower", lower), ("upper", upper)):
            _check_dev(t, torch.float32, dev, n_, (H, A))
        s0 = np.ascontiguousarray(np.asarray(s0, dtype=np.float32).reshape(-1))
        if s0.shape[0] != self.spec.obs_dim * n_env:
            raise ValueError(f"initial_state has {s0.shape[0]} values, expected {n_env} x {self.spec.obs_dim}")
        with torch.cuda.device(dev):
            _lib.check(self._lib.hipets_plan_mppi_batched(self._h, pop, H, A, num_iterations, float(gamma), float(beta), int(n_env), _ptr(mean),
                                                          _ptr(lower), _ptr(upper), s0.ctypes.data_as(C.c_void_p), num_particles,
                                                          int(seed) & (2**64 - 1), int(plan_id) & (2**64 - 1), _stream(dev)))
        return mean

    def plan_icem(self, p: "_lib.IcemParams", x0, lower, upper, elite: torch.Tensor, has_elite: bool, s0: np.ndarray,
                  num_particles: int, seed: int = 0, plan_id: int = 0, keep_idx: Optional[torch.Tensor] = None,
                  out: Optional[torch.Tensor] = None, n_env: int = 1) -> torch.Tensor:
        """Whole iCEM plan on the device (hipets_plan_icem[_batched]).  ``elite`` [elite_num, H, A] is the optimizer's
        persistent elite set (read when ``has_elite``, always overwritten); ``keep_idx`` int32 [num_iterations, keep]
        optionally injects the kept-elite draws.  n_env > 1: x0 / out [n_env, H, A], elite [n_env, elite_num, H, A],
        keep_idx [num_iterations, n_env, keep], s0 [n_env, obs_dim]."""
        if self.spec is None:
            raise HipetsError("Engine.set_model() has not been called")
        dev = self.device
        shp = (p.horizon, p.act_dim)
        _check_dev(x0, torch.float32, dev, "x0", numel=n_env * p.horizon * p.act_dim)
        for n_, t in (("lower", lower), ("upper", upper)):
            _check_dev(t, torch.float32, dev, n_, shp)
        _check_dev(elite, torch.float32, dev, "elite", numel=n_env * p.elite_num * p.horizon * p.act_dim)
        if keep_idx is not None:
            _check_dev(keep_idx, torch.int32, dev, "keep_idx", numel=p.num_iterations * n_env * p.keep_elite_size)
        s0 = np.ascontiguousarray(np.asarray(s0, dtype=np.float32).reshape(-1))
        if s0.shape[0] != self.spec.obs_dim * n_env:
            raise ValueError(f"initial_state has {s0.shape[0]} values, expected {n_env} x {self.spec.obs_dim}")
        if out is None:
            out = torch.empty(tuple(x0.shape), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(self._lib.hipets_plan_icem_batched(self._h, C.byref(p), int(n_env), _ptr(x0), _ptr(lower), _ptr(upper), _ptr(elite),
                                                          int(bool(has_elite)), _ptr(keep_idx) if keep_idx is not None else None,
                                                          s0.ctypes.data_as(C.c_void_p), num_particles, int(seed) & (2**64 - 1),
                                                          int(plan_id) & (2**64 - 1), _ptr(out), _stream(dev)))
        return out

    # ---- in-library RCCL communicator (population-sharded fused plans) -----------------------------
    def comm_unique_id(self) -> bytes:
        """A fresh communicator id (rank 0 makes it, the host broadcasts the bytes to the other ranks)."""
        buf = C.create_string_buffer(_lib.COMM_ID_BYTES)
        _lib.check(self._lib.hipets_comm_unique_id(buf))
        return buf.raw

    def comm_init(self, unique_id: bytes, rank: int, world_size: int):
        if len(unique_id) != _lib.COMM_ID_BYTES:
            raise ValueError(f"unique_id must be {_lib.COMM_ID_BYTES} bytes")
        with torch.cuda.device(self.device):
            _lib.check(self._lib.hipets_comm_init(self._h, C.c_char_p(unique_id), int(rank), int(world_size)))
        self.comm_world, self.comm_rank = int(world_size), int(rank)

    def comm_destroy(self):
        _lib.check(self._lib.hipets_comm_destroy(self._h))
        self.comm_world, self.comm_rank, self.comm_group = 1, 0, None

    def comm_info(self) -> tuple:
        """(rank, world size) as the communicator itself reports them (hipets_comm_info)."""
        r, w = C.c_int32(0), C.c_int32(1)
        _lib.check(self._lib.hipets_comm_info(self._h, C.byref(r), C.byref(w)))
        return r.value, w.value

    def plan_cem_sharded(self, p: CemParams, x0, lower, upper, s0: np.ndarray, num_particles: int, seed: int = 0, plan_id: int = 0,
                         out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """hipets_plan_cem over all ranks of the communicator (identical arguments on every rank)."""
        if self.spec is None:
            raise HipetsError("Engine.set_model() has not been called")
        dev = self.device
        shp = (p.horizon, p.act_dim)
        for n_, t in (("x0", x0), ("lower", lower), ("upper", upper)):
            _check_dev(t, torch.float32, dev, n_, shp)
        s0 = np.ascontiguousarray(np.asarray(s0, dtype=np.float32).reshape(-1))
        if s0.shape[0] != self.spec.obs_dim:
            raise ValueError(f"initial_state has {s0.shape[0]} values, expected {self.spec.obs_dim}")
        if out is None:
            out = torch.empty(shp, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(self._lib.hipets_plan_cem_sharded(self._h, C.byref(p), _ptr(x0), _ptr(lower), _ptr(upper),
                                                         s0.ctypes.data_as(C.c_void_p), num_particles, int(seed) & (2**64 - 1),
                                                         int(plan_id) & (2**64 - 1), _ptr(out), _stream(dev)))
        return out

    def plan_mppi_sharded(self, pop: int, H: int, A: int, num_iterations: int, gamma: float, beta: float, mean: torch.Tensor, lower,
                          upper, s0: np.ndarray, num_particles: int, seed: int = 0, plan_id: int = 0) -> torch.Tensor:
        """hipets_plan_mppi over all ranks of the communicator (identical arguments on every rank; ``mean`` in place)."""
        if self.spec is None:
            raise HipetsError("Engine.set_model() has not been called")
        dev = self.device
        for n_, t in (("mean", mean), ("lower", lower), ("upper", upper)):
            _check_dev(t, torch.float32, dev, n_, (H, A))
        s0 = np.ascontiguousarray(np.asarray(s0, dtype=np.float32).reshape(-1))
        if s0.shape[0] != self.spec.obs_dim:
            raise ValueError(f"initial_state has {s0.shape[0]} values, expected {self.spec.obs_dim}")
        with torch.cuda.device(dev):
            _lib.check(self._lib.hipets_plan_mppi_sharded(self._h, pop, H, A, num_iterations, float(gamma), float(beta), _ptr(mean), _ptr(lower),
                                                          _ptr(upper), s0.ctypes.data_as(C.c_void_p), num_particles, int(seed) & (2**64 - 1),
                                                          int(plan_id) & (2**64 - 1), _stream(dev)))
        return mean

    def plan_icem_sharded(self, p: "_lib.IcemParams", x0, lower, upper, elite: torch.Tensor, has_elite: bool, s0: np.ndarray,
                          num_particles: int, seed: int = 0, plan_id: int = 0, keep_idx: Optional[torch.Tensor] = None,
                          out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """hipets_plan_icem over all ranks of the communicator (identical arguments on every rank; ``elite`` in place)."""
        if self.spec is None:
            raise HipetsError("Engine.set_model() has not been called")
        dev = self.device
        shp = (p.horizon, p.act_dim)
        for n_, t in (("x0", x0), ("lower", lower), ("upper", upper)):
            _check_dev(t, torch.float32, dev, n_, shp)
        _check_dev(elite, torch.float32, dev, "elite", numel=p.elite_num * p.horizon * p.act_dim)
        if keep_idx is not None:
            _check_dev(keep_idx, torch.int32, dev, "keep_idx", numel=p.num_iterations * p.keep_elite_size)
        s0 = np.ascontiguousarray(np.asarray(s0, dtype=np.float32).reshape(-1))
        if s0.shape[0] != self.spec.obs_dim:
            raise ValueError(f"initial_state has {s0.shape[0]} values, expected {self.spec.obs_dim}")
        if out is None:
            out = torch.empty(shp, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(self._lib.hipets_plan_icem_sharded(self._h, C.byref(p), _ptr(x0), _ptr(lower), _ptr(upper), _ptr(elite), int(bool(has_elite)),
                                                          _ptr(keep_idx) if keep_idx is not None else None, s0.ctypes.data_as(C.c_void_p),
                                                          num_particles, int(seed) & (2**64 - 1), int(plan_id) & (2**64 - 1), _ptr(out), _stream(dev)))
        return out

    # ---- PlaNet latent planner -----------------------------------------------
    def planet_set_model(self, spec):
        """Pack the planning heads of a PlaNet model (``hipets.PlaNetSpec``)."""
        spec.validate()
        dev = self.device
        d = _lib.PlanetDesc()
        d.latent_size, d.action_size, d.belief_size, d.hidden_size = spec.latent_size, spec.action_size, spec.belief_size, spec.hidden_size
        d.min_std = float(spec.min_std)
        keep = []
        for n in _lib._PLANET_TENSORS:
            t = getattr(spec, n).detach().to(device=dev, dtype=torch.float32).contiguous()
            keep.append(t)
            setattr(d, n, t.data_ptr())
        with torch.cuda.device(dev):
            _lib.check(self._lib.hipets_planet_set_model(self._h, C.byref(d), _stream(dev)))
        self.planet_spec = spec

    def planet_rollout(self, actions: torch.Tensor, latent0: torch.Tensor, belief0: torch.Tensor, num_particles: int, *,
                       eps: Optional[torch.Tensor] = None, sample: bool = True, seed: int = 0, stream_id: int = 0,
                       trace_latent: Optional[torch.Tensor] = None, trace_belief: Optional[torch.Tensor] = None,
                       trace_rewards: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
                       phase_cycles: Optional[torch.Tensor] = None) -> torch.Tensor:
        """ModelEnv.evaluate_action_sequences on the PlaNet model (model_env.py:145-191): actions [pop, H, A]; latent0 /
        belief0 = the saved posterior sample / belief ([latent] / [belief], any leading 1s) on the device."""
        spec = getattr(self, "planet_spec", None)
        if spec is None:
            raise HipetsError("Engine.planet_set_model() has not been called")
        dev = self.device
        if actions.ndim != 3:
            raise ValueError("action_sequences must be [B, H, A]")
        _check_dev(actions, torch.float32, dev, "action_sequences")
        pop, H, A = actions.shape
        if A != spec.action_size:
            raise ValueError(f"action dim {A} != model action_size {spec.action_size}")
        _check_dev(latent0, torch.float32, dev, "latent0", numel=spec.latent_size)
        _check_dev(belief0, torch.float32, dev, "belief0", numel=spec.belief_size)
        B = pop * num_particles
        o = _lib.PlanetOpts()
        o.seed, o.stream_id = int(seed) & (2**64 - 1), int(stream_id) & (2**64 - 1)
        o.no_sample = int(not sample)
        if eps is not None:
            _check_dev(eps, torch.float32, dev, "eps", (H, B, spec.latent_size))
            o.eps = _ptr(eps)
        for name, t, width in (("trace_latent", trace_latent, spec.latent_size), ("trace_belief", trace_belief, spec.belief_size),
                               ("trace_rewards", trace_rewards, None)):
            if t is not None:
                _check_dev(t, torch.float32, dev, name, (H, B) if width is None else (H, B, width))
                setattr(o, name, _ptr(t))
        if phase_cycles is not None:  # (filled by -DHIPETS_LEAN_PROF=1 builds only: profiles/one_tile_phase_profile.py)
            _check_dev(phase_cycles, torch.int64, dev, "phase_cycles", (8, 16))
            o.phase_cycles = _ptr(phase_cycles)
        if out is None:
            out = torch.empty(pop, dtype=torch.float32, device=dev)
        else:
            _check_dev(out, torch.float32, dev, "out", (pop,))
        with torch.cuda.device(dev):
            _lib.check(self._lib.hipets_planet_rollout(self._h, _ptr(actions), _ptr(latent0), _ptr(belief0), pop, H, num_particles,
                                                       C.byref(o), _ptr(out), _stream(dev)))
        return out

    def plan_planet_cem(self, p: CemParams, x0, lower, upper, latent0: torch.Tensor, belief0: torch.Tensor, num_particles: int,
                        seed: int = 0, plan_id: int = 0, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Whole CEM plan over the PlaNet latent model on the device (hipets_plan_planet_cem)."""
        spec = getattr(self, "planet_spec", None)
        if spec is None:
            raise HipetsError("Engine.planet_set_model() has not been called")
        dev = self.device
        shp = (p.horizon, p.act_dim)
        for n_, t in (("x0", x0), ("lower", lower), ("upper", upper)):
            _check_dev(t, torch.float32, dev, n_, shp)
        _check_dev(latent0, torch.float32, dev, "latent0", numel=spec.latent_size)
        _check_dev(belief0, torch.float32, dev, "belief0", numel=spec.belief_size)
        if out is None:
            out = torch.empty(shp, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(self._lib.hipets_plan_planet_cem(self._h, C.byref(p), _ptr(x0), _ptr(lower), _ptr(upper), _ptr(latent0), _ptr(belief0),
                                                        num_particles, int(seed) & (2**64 - 1), int(plan_id) & (2**64 - 1), _ptr(out),
                                                        _stream(dev)))
        return out

    # ---- instrumentation ---------------------------------------------------------------------------
    def timing_enable(self, on=True):
        """True / 1: time every rollout-kernel launch; k > 1: every k-th launch; False / 0: off."""
        _lib.check(self._lib.hipets_timing_enable(self._h, int(on)))

    def timing_read(self, reset: bool = True):
        n, ms = C.c_int64(), C.c_double()
        _lib.check(self._lib.hipets_timing_read(self._h, C.byref(n), C.byref(ms), int(reset)))
        return n.value, ms.value
