"""CPU oracle for the PETS planning / rollout hot path.  *** TEST INFRASTRUCTURE ONLY ***

This file restates, on the CPU with plain torch ops, the algorithm of the reference
(facebookresearch/mbrl-lib v0.2.0) for the path named in BASELINE.json:

    TrajectoryOptimizerAgent.act -> CEM / iCEM / MPPI -> ModelEnv.evaluate_action_sequences
        -> GaussianMLP ensemble (TS1 / TSinf / expectation) -> reward / termination -> refit

It is the *checker* for the HIP engine in ``mbrl-lib_amd/``.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it; the
product package never does (and fails loudly when its HIP library is missing).

Parity status: PINNED.  ``tests/test_oracle_vs_reference.py`` (runs where /root/reference is
mounted) asserts this restatement is bitwise equal to the reference classes, and
``tests/golden/*.npz`` (made by ``oracle/make_golden.py`` from the reference itself) pin it on
machines where the reference is absent.

Every function cites the reference file:line it follows (paths relative to the reference
root).  Nothing here is copied: the reference spreads this logic over seven classes; this is
a flat functional restatement with injected randomness.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Callable, List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

# ----------------------------------------------------------------------------------------
# Model description (plain container; the product has its own, this one is duck-typed)
# ----------------------------------------------------------------------------------------


@dataclass
class OracleModel:
    """Everything ``evaluate_action_sequences`` reads from the live reference objects
    (SURVEY.md section 8b "What the engine must read from live objects")."""

    weights: List[torch.Tensor]  # per layer [E, in_l, out_l]   (models/util.py:31-50)
    biases: List[torch.Tensor]  # per layer [E, 1, out_l]
    min_logvar: Optional[torch.Tensor] = None  # [1, out]       (gaussian_mlp.py:117-122)
    max_logvar: Optional[torch.Tensor] = None
    elite_models: Optional[Sequence[int]] = None  # gaussian_mlp.py:377-379
    activation: str = "silu"  # "silu" | "relu" | "leaky_relu" | "tanh" | "sigmoid"
    propagation: str = "random_model"  # | "fixed_model" | "expectation"
    deterministic: bool = False
    norm_mean: Optional[torch.Tensor] = None  # [1, in]  f32 or f64 (util/math.py:108-111)
    norm_std: Optional[torch.Tensor] = None
    target_is_delta: bool = True
    no_delta_list: Sequence[int] = field(default_factory=list)
    learned_rewards: bool = False
    obs_process: str = "none"  # "none" | "halfcheetah" | "cartpole_pets"
    reward: Optional[str] = "halfcheetah"  # name in REWARD_FNS, or None => learned reward
    termination: str = "no_termination"
    # "gaussian_mlp": one GaussianMLP with E members, balanced shuffles (gaussian_mlp.py:179-216).
    # "basic_ensemble": BasicEnsemble of E single-member GaussianMLPs (conf/dynamics_model/basic_ensemble.yaml):
    #   every row draws its member independently with randint FROM THE GENERATOR (basic_ensemble.py:122-129,
    #   255-260), no batch-size check, elites ignored (:262-266).
    ensemble_kind: str = "gaussian_mlp"

    @property
    def out_size(self) -> int:
        n = self.weights[-1].shape[-1]
        return n if self.deterministic else n // 2

    @property
    def active_members(self) -> List[int]:
        e = self.weights[0].shape[0]
        if self.ensemble_kind == "basic_ensemble":
            return list(range(e))
        return list(self.elite_models) if self.elite_models is not None else list(range(e))


# ----------------------------------------------------------------------------------------
# Closed-form reward / termination / obs-preprocess functions
# ----------------------------------------------------------------------------------------


def term_no_termination(act, nobs):  # env/termination_fns.py:58-63
    return torch.zeros(len(nobs), 1, dtype=torch.bool, device=nobs.device)


def term_cartpole(act, nobs):  # env/termination_fns.py:29-44
    x, theta = nobs[:, 0], nobs[:, 2]
    thr = 12 * 2 * math.pi / 360
    not_done = (x > -2.4) * (x < 2.4) * (theta > -thr) * (theta < thr)
    return (~not_done)[:, None]


def term_inverted_pendulum(act, nobs):  # env/termination_fns.py:47-55
    not_done = torch.isfinite(nobs).all(-1) * (nobs[:, 1].abs() <= 0.2)
    return (~not_done)[:, None]


def term_hopper(act, nobs):  # env/termination_fns.py:12-26
    height, angle = nobs[:, 0], nobs[:, 1]
    not_done = (
        torch.isfinite(nobs).all(-1)
        * (nobs[:, 1:].abs() < 100).all(-1)
        * (height > 0.7)
        * (angle.abs() < 0.2)
    )
    return (~not_done)[:, None]


def term_walker2d(act, nobs):  # env/termination_fns.py:66-74
    height, angle = nobs[:, 0], nobs[:, 1]
    not_done = (height > 0.8) * (height < 2.0) * (angle > -1.0) * (angle < 1.0)
    return (~not_done)[:, None]


def term_ant(act, nobs):  # env/termination_fns.py:77-85
    x = nobs[:, 0]
    not_done = torch.isfinite(nobs).all(-1) * (x >= 0.2) * (x <= 1.0)
    return (~not_done)[:, None]


def term_humanoid(act, nobs):  # env/termination_fns.py:88-95
    z = nobs[:, 0]
    return ((z < 1.0) + (z > 2.0))[:, None]


TERMINATION_FNS = {
    "no_termination": term_no_termination,
    "cartpole": term_cartpole,
    "inverted_pendulum": term_inverted_pendulum,
    "hopper": term_hopper,
    "walker2d": term_walker2d,
    "ant": term_ant,
    "humanoid": term_humanoid,
}


def rew_cartpole(act, nobs):  # env/reward_fns.py:10-13
    return (~term_cartpole(act, nobs)).float().view(-1, 1)


def rew_inverted_pendulum(act, nobs):  # env/reward_fns.py:27-30
    return (~term_inverted_pendulum(act, nobs)).float().view(-1, 1)


def rew_cartpole_pets(act, nobs):  # env/reward_fns.py:16-24
    goal = torch.tensor([0.0, 0.6], device=nobs.device)
    x0 = nobs[:, :1]
    theta = nobs[:, 1:2]
    ee = torch.cat([x0 - 0.6 * theta.sin(), -0.6 * theta.cos()], dim=1)
    obs_cost = torch.exp(-torch.sum((ee - goal) ** 2, dim=1) / (0.6**2))
    act_cost = -0.01 * torch.sum(act**2, dim=1)
    return (obs_cost + act_cost).view(-1, 1)


def rew_halfcheetah(act, nobs):  # env/reward_fns.py:33-38
    ctrl = -0.1 * act.square().sum(dim=1)
    run = nobs[:, 0] - 0.0 * nobs[:, 2].square()
    return (run + ctrl).view(-1, 1)


def rew_pusher(act, nobs):  # env/reward_fns.py:41-53
    goal = torch.tensor([0.45, -0.05, -0.323], device=nobs.device)
    tip, obj = nobs[:, 14:17], nobs[:, 17:20]
    tip_obj = (tip - obj).abs().sum(axis=1)
    obj_goal = (goal - obj).abs().sum(axis=1)
    obs_cost = 0.5 * tip_obj + 1.25 * obj_goal
    act_cost = 0.1 * (act**2).sum(axis=1)
    return -(obs_cost + act_cost).view(-1, 1)


REWARD_FNS = {
    "cartpole": rew_cartpole,
    "inverted_pendulum": rew_inverted_pendulum,
    "cartpole_pets": rew_cartpole_pets,
    "halfcheetah": rew_halfcheetah,
    "pusher": rew_pusher,
}


def obs_halfcheetah(s):  # env/pets_halfcheetah.py:91-113
    return torch.cat([s[..., 1:2], torch.sin(s[..., 2:3]), torch.cos(s[..., 2:3]), s[..., 3:]], dim=-1)


def obs_cartpole_pets(s):  # env/pets_cartpole.py:78-101
    return torch.cat([torch.sin(s[..., 1:2]), torch.cos(s[..., 1:2]), s[..., :1], s[..., 2:]], dim=-1)


OBS_PROCESS_FNS = {"none": None, "halfcheetah": obs_halfcheetah, "cartpole_pets": obs_cartpole_pets}

_ACT = {
    "silu": F.silu,
    "relu": F.relu,
    "leaky_relu": F.leaky_relu,
    "tanh": torch.tanh,
    "sigmoid": torch.sigmoid,
}


# ----------------------------------------------------------------------------------------
# GaussianMLP forward for the active members (gaussian_mlp.py:129-175, models/util.py:53-65)
# ----------------------------------------------------------------------------------------


def _members_forward(m: OracleModel, x: torch.Tensor):
    """x: [M, n, in] -> (mean [M,n,out], logvar [M,n,out] | None); only active members,
    like ``_default_forward(only_elite=True)`` (gaussian_mlp.py:140-154)."""
    el = m.active_members
    act = _ACT[m.activation]
    h = x
    nl = len(m.weights)
    for li in range(nl):
        w = m.weights[li][el, ...]
        b = m.biases[li][el, ...]
        h = h.matmul(w) + b  # models/util.py:54-59
        if li < nl - 1:
            h = act(h)
    if m.deterministic:
        return h, None
    out = m.out_size
    mean = h[..., :out]
    logvar = h[..., out:]
    logvar = m.max_logvar - F.softplus(m.max_logvar - logvar)  # gaussian_mlp.py:152
    logvar = m.min_logvar + F.softplus(logvar - m.min_logvar)  # gaussian_mlp.py:153
    return mean, logvar


def _single_member(m: OracleModel, slot: int) -> OracleModel:
    """Active member ``slot`` as a one-member GaussianMLP.  A BasicEnsemble member is its own GaussianMLP with its own
    learned logvar bounds (gaussian_mlp.py:117-122): ``min_logvar`` / ``max_logvar`` may then be [E, out]."""
    member = m.active_members[slot]
    kw = {"elite_models": [member], "ensemble_kind": "gaussian_mlp"}
    if m.min_logvar is not None and m.min_logvar.shape[0] > 1:
        kw["min_logvar"], kw["max_logvar"] = m.min_logvar[member:member + 1], m.max_logvar[member:member + 1]
    return OracleModel(**{**m.__dict__, **kw})


def ensemble_forward(m: OracleModel, x: torch.Tensor, perm: Optional[torch.Tensor] = None,
                     member_of_row: Optional[torch.Tensor] = None):
    """Propagation-aware forward (gaussian_mlp.py:179-216, 156-177).

    ``perm`` is the reference's ``model_shuffle_indices``: shuffled row j (= x[perm[j]]) goes to
    active member ``j // (B/M)``.  ``member_of_row`` is the oracle's generalisation used to check
    the engine's fast mode: an explicit row -> active-member-slot map (not necessarily balanced);
    for a balanced map it is equivalent to ``perm = argsort(member_of_row, stable)``
    (checked in tests/test_oracle_vs_reference.py).
    """
    B = x.shape[0]
    M = len(m.active_members)
    if member_of_row is None and m.ensemble_kind == "gaussian_mlp" and B % M != 0:  # gaussian_mlp.py:195-200 (EVERY method)
        raise ValueError(
            f"GaussianMLP ensemble requires batch size to be a multiple of the "
            f"number of models. Current batch size is {B} for "
            f"{M} models."
        )
    if m.propagation in ("random_model", "fixed_model"):
        if member_of_row is not None:
            mean = torch.empty(B, m.weights[-1].shape[-1] // (1 if m.deterministic else 2))
            logvar = None if m.deterministic else torch.empty_like(mean)
            for s in range(M):
                rows = (member_of_row == s).nonzero().flatten()
                if rows.numel() == 0:
                    continue
                mu_s, lv_s = _members_forward(_single_member(m, s), x[rows].unsqueeze(0))
                mean[rows] = mu_s[0]
                if lv_s is not None:
                    logvar[rows] = lv_s[0]
            return mean, logvar
        shuffled = x.unsqueeze(0)[:, perm, ...].view(M, B // M, -1)  # :164-166
        mean, logvar = _members_forward(m, shuffled)
        mean = mean.reshape(B, -1)
        mean[perm] = mean.clone()  # :170
        if logvar is not None:
            logvar = logvar.reshape(B, -1)
            logvar[perm] = logvar.clone()  # :174
        return mean, logvar
    if m.propagation == "expectation" and m.ensemble_kind == "basic_ensemble":
        # basic_ensemble.py:100-108, 131-137: one forward per member, stacked, then averaged (logvars too)
        outs = [_members_forward(_single_member(m, s), x.unsqueeze(0)) for s in range(len(m.active_members))]
        mean = torch.stack([o[0][0] for o in outs], dim=0).mean(dim=0)
        logvar = None if outs[0][1] is None else torch.stack([o[1][0] for o in outs], dim=0).mean(dim=0)
        return mean, logvar
    if m.propagation == "expectation":  # :213-215 (averages log-variances, Appendix B12)
        mean, logvar = _members_forward(m, x.unsqueeze(0))
        return mean.mean(dim=0), (logvar.mean(dim=0) if logvar is not None else None)
    raise ValueError(f"Invalid propagation method {m.propagation}.")


def model_input(m: OracleModel, obs: torch.Tensor, act: torch.Tensor) -> torch.Tensor:
    """one_dim_tr_model.py:103-116 + util/math.py:129-143 (normalise in the stats' dtype)."""
    f = OBS_PROCESS_FNS[m.obs_process]
    o = f(obs) if f is not None else obs
    inp = torch.cat([o, act], dim=1)
    if m.norm_mean is not None:
        inp = ((inp - m.norm_mean) / m.norm_std).float()
    return inp


# ----------------------------------------------------------------------------------------
# ModelEnv.step  (models/model_env.py:87-140) -> OneDTransitionRewardModel.sample (one_dim_tr_model.py:245-289)
# ----------------------------------------------------------------------------------------


def step(m: OracleModel, x: torch.Tensor, a: torch.Tensor, perm: Optional[torch.Tensor] = None,
         eps: Optional[torch.Tensor] = None, member_of_row: Optional[torch.Tensor] = None, sample: bool = True):
    """One transition for a batch: returns (next_obs [B,obs], rewards [B,1], dones [B,1] bool).
    ``sample=False`` is ModelEnv.step's default (deterministic mean, model.py:458-466)."""
    inp = model_input(m, x, a)
    mean, logvar = ensemble_forward(m, inp, perm, member_of_row)
    if m.deterministic or logvar is None or not sample:
        pred = mean
    else:
        pred = mean + torch.sqrt(logvar.exp()) * eps  # model.py:471-473
    nobs = pred[:, :-1] if m.learned_rewards else pred
    if m.target_is_delta:
        tmp = nobs + x
        for d in m.no_delta_list:
            tmp[:, d] = nobs[:, d]
        nobs = tmp
    rew_fn = REWARD_FNS[m.reward] if m.reward is not None else None
    r = pred[:, -1:].clone() if rew_fn is None else rew_fn(a, nobs)
    d = TERMINATION_FNS[m.termination](a, nobs)
    return nobs, r, d


# ----------------------------------------------------------------------------------------
# ModelEnv.evaluate_action_sequences  (models/model_env.py:145-191)
# ----------------------------------------------------------------------------------------


def rollout(
    m: OracleModel,
    actions: torch.Tensor,  # [pop, H, A] f32
    s0: np.ndarray,  # [obs]
    num_particles: int,
    perms: Optional[torch.Tensor] = None,  # [H, B] (random_model) or [B] (fixed_model) int64
    eps: Optional[torch.Tensor] = None,  # [H, B, out] f32 standard normals
    members: Optional[torch.Tensor] = None,  # [H, B] explicit row->member-slot map (fast-mode check)
    global_rng: bool = False,  # draw perms from torch's global RNG like the reference
    generator: Optional[torch.Generator] = None,  # draw eps like ModelEnv._rng
    trace: Optional[dict] = None,
) -> torch.Tensor:
    """Returns the particle-averaged return of every action sequence, [pop] f32."""
    pop, H, A = actions.shape
    P = num_particles
    B = pop * P
    dev = actions.device  # CPU for every parity use; a GPU only in bench.py's informational PyTorch-ROCm leg
    x = torch.from_numpy(np.tile(np.asarray(s0), (B, 1)).astype(np.float32)).to(dev)  # model_env.py:170-174
    tot = torch.zeros(B, 1, device=dev)
    term = torch.zeros(B, 1, dtype=torch.bool, device=dev)
    out = m.weights[-1].shape[-1] // (1 if m.deterministic else 2)
    fixed_perm = None
    if m.ensemble_kind == "basic_ensemble" and members is None and m.propagation == "fixed_model":
        # model.py:404-407 -> basic_ensemble.py:255-260: randint from ModelEnv's generator at reset
        members = torch.randint(len(m.active_members), (B,), generator=generator, device=dev)
    draw_members = m.ensemble_kind == "basic_ensemble" and members is None and m.propagation == "random_model"
    if m.propagation == "fixed_model" and members is None:
        # model.py:404-407 -> gaussian_mlp.py:363-375: randperm from the GLOBAL rng (the generator is
        # deliberately ignored there, see the comment at gaussian_mlp.py:374)
        if perms is not None:
            fixed_perm = perms if perms.ndim == 1 else perms[0]
        else:
            assert global_rng
            fixed_perm = torch.randperm(B, device=dev)
    rew_fn = REWARD_FNS[m.reward] if m.reward is not None else None
    term_fn = TERMINATION_FNS[m.termination]
    for t in range(H):
        a = torch.repeat_interleave(actions[:, t, :], P, dim=0)  # model_env.py:179-182
        inp = model_input(m, x, a)
        perm = None
        mem = None
        if members is not None:
            mem = members[t] if members.ndim == 2 else members
        elif draw_members:  # basic_ensemble.py:122-129: one randint per step, BEFORE that step's normal draw
            mem = torch.randint(len(m.active_members), (B,), generator=generator, device=dev)
        elif m.propagation == "random_model":
            if perms is not None:
                perm = perms[t]
            else:
                assert global_rng
                perm = torch.randperm(B, device=dev)  # gaussian_mlp.py:203-205 (GLOBAL rng)
        elif m.propagation == "fixed_model":
            perm = fixed_perm
        mean, logvar = ensemble_forward(m, inp, perm, mem)
        if m.deterministic or logvar is None:  # model.py:458-466
            pred = mean
        else:
            std = torch.sqrt(logvar.exp())  # model.py:471-472
            if eps is not None:
                e = eps[t]
            else:
                e = torch.empty(B, out, device=dev).normal_(0.0, 1.0, generator=generator)
            pred = mean + std * e  # == torch.normal(mean, std, generator) on CPU
        nobs = pred[:, :-1] if m.learned_rewards else pred  # one_dim_tr_model.py:280
        if m.target_is_delta:  # :281-286
            tmp = nobs + x
            for d in m.no_delta_list:
                tmp[:, d] = nobs[:, d]
            nobs = tmp
        r = pred[:, -1:].clone() if rew_fn is None else rew_fn(a, nobs)  # model_env.py:124-128
        d = term_fn(a, nobs)  # :129
        if trace is not None:
            trace.setdefault("next_obs", []).append(nobs.clone())
            trace.setdefault("rewards", []).append(r.clone())
            trace.setdefault("dones", []).append(d.clone())
        r[term] = 0  # :186
        term |= d  # :187
        tot += r  # :188
        x = nobs
    return tot.reshape(-1, P).mean(dim=1)  # :190-191


# ----------------------------------------------------------------------------------------
# Sampling helpers (util/math.py)
# ----------------------------------------------------------------------------------------


def truncated_normal_(t: torch.Tensor, mean: float = 0.0, std: float = 1.0,
                      generator: Optional[torch.Generator] = None) -> torch.Tensor:
    """util/math.py:69-92: N(mean,std) redrawn elementwise until inside mean +- 2 std."""
    t.normal_(mean, std, generator=generator)
    while True:
        cond = torch.logical_or(t < mean - 2 * std, t > mean + 2 * std)
        n = int(torch.sum(cond).item())
        if n == 0:
            break
        t[cond] = torch.normal(mean, std, size=(n,), generator=generator, device=t.device)
    return t


def rfftfreq(n: int) -> torch.Tensor:  # util/math.py:306-310 (torch >= 1.8 branch)
    return torch.fft.rfftfreq(n)


def powerlaw_psd_gaussian(exponent: float, size: Sequence[int], fmin: float = 0,
                          normals: Optional[Sequence[torch.Tensor]] = None, record_normals: Optional[list] = None) -> torch.Tensor:
    """util/math.py:318-396: coloured noise with PSD ~ 1/f^exponent along the last axis, unit
    variance.  ``normals`` optionally injects (sr, si) of shape size[:-1]+[len(f)]."""
    size = list(size)
    samples = size[-1]
    f = rfftfreq(samples)
    s_scale = f.clone()
    fmin = max(fmin, 1.0 / samples)
    ix = int(torch.sum(s_scale < fmin).item())
    if ix and ix < len(s_scale):
        s_scale[:ix] = s_scale[ix]
    s_scale = s_scale ** (-exponent / 2.0)
    w = s_scale[1:].clone()
    w[-1] *= (1 + (samples % 2)) / 2.0
    sigma = 2 * torch.sqrt(torch.sum(w**2)) / samples
    size[-1] = len(f)
    dims_to_add = len(size) - 1
    s_scale = s_scale[(None,) * dims_to_add + (Ellipsis,)]
    if normals is None:
        # torch.distributions.Normal(0, scale).sample(shape) == N(0,1).mul_(scale).add_(0) on CPU (same RNG stream):
        # drawn here as explicit unit normals so they can be recorded and injected into the device kernel
        shape = tuple(size[:-1]) + (len(f),)
        normals = (torch.empty(shape).normal_(0.0, 1.0), torch.empty(shape).normal_(0.0, 1.0))
    if record_normals is not None:
        record_normals.append((normals[0].clone(), normals[1].clone()))
    sr = normals[0] * s_scale + 0.0
    si = normals[1] * s_scale + 0.0
    if not (samples % 2):
        si[..., -1] = 0
    si[..., 0] = 0
    s = sr + 1j * si
    y = torch.fft.irfft(s, n=samples, dim=-1) / sigma
    return y


# ----------------------------------------------------------------------------------------
# Optimizers (planning/trajectory_opt.py)
# ----------------------------------------------------------------------------------------


def elite_count(pop: int, ratio: float) -> int:  # trajectory_opt.py:89-91
    return int(np.ceil(pop * ratio).astype(np.int32))


def cem_optimize(
    obj_fun: Callable[[torch.Tensor], torch.Tensor],
    x0: torch.Tensor,  # [H, A]
    lower: torch.Tensor,
    upper: torch.Tensor,
    num_iterations: int,
    elite_ratio: float,
    population_size: int,
    alpha: float,
    return_mean_elites: bool = False,
    clipped_normal: bool = False,
    noise: Optional[Sequence[torch.Tensor]] = None,  # per-iteration z [pop,H,A] (already truncated / randn)
    record: Optional[list] = None,
    teacher: Optional[Sequence[tuple]] = None,  # teacher[i] = (mu, dispersion) to START iteration i + 1 from (replay aid)
) -> torch.Tensor:
    """CEMOptimizer.optimize (trajectory_opt.py:142-188) with :100-140 inlined.

    ``teacher`` (never used by the pinned-to-reference paths) re-bases every iteration on another implementation's
    recorded state, so a replay checks each iteration on its own instead of compounding a legitimate elite tie."""
    K = elite_count(population_size, elite_ratio)
    mu = x0.clone()
    disp = torch.ones_like(mu) if clipped_normal else ((upper - lower) ** 2) / 16  # :103-108
    best = torch.empty_like(mu)
    best_val = -np.inf
    for i in range(num_iterations):
        if teacher is not None and i > 0:
            mu, disp = teacher[i - 1][0].clone(), teacher[i - 1][1].clone()
        if noise is not None:
            z = noise[i]
        elif clipped_normal:
            z = torch.randn((population_size,) + tuple(x0.shape), device=x0.device)
        else:
            z = truncated_normal_(torch.zeros((population_size,) + tuple(x0.shape), device=x0.device))
        if clipped_normal:  # :116-120
            p = mu + disp * z
            p = torch.where(p > lower, p, lower)
            population = torch.where(p < upper, p, upper)
        else:  # :122-128
            lb_dist = mu - lower
            ub_dist = upper - mu
            mv = torch.min(torch.square(lb_dist / 2), torch.square(ub_dist / 2))
            cv = torch.min(mv, disp)
            population = z * torch.sqrt(cv) + mu
        values = obj_fun(population)
        values = values.clone()
        values[values.isnan()] = -1e-10  # :178
        best_values, elite_idx = values.topk(K)  # :179
        elite = population[elite_idx]
        new_mu = torch.mean(elite, dim=0)  # :133
        new_disp = torch.std(elite, dim=0) if clipped_normal else torch.var(elite, dim=0)  # :134-137
        mu = alpha * mu + (1 - alpha) * new_mu  # :138
        disp = alpha * disp + (1 - alpha) * new_disp  # :139
        if best_values[0] > best_val:  # :184-186
            best_val = best_values[0]
            best = population[elite_idx[0]].clone()
        if record is not None:
            record.append(dict(population=population.clone(), values=values.clone(),
                               elite_idx=elite_idx.clone(), mu=mu.clone(), disp=disp.clone(),
                               best=best.clone(), best_val=float(best_val)))
    return mu if return_mean_elites else best


class MPPIState:
    """Persistent MPPIOptimizer.mean (trajectory_opt.py:224-228); survives agent.reset (App. B6)."""

    def __init__(self, H: int, A: int):
        self.mean = torch.zeros(H, A)


def mppi_optimize(
    obj_fun, state: MPPIState, lower: torch.Tensor, upper: torch.Tensor, num_iterations: int,
    population_size: int, gamma: float, sigma: float, beta: float,
    noise: Optional[Sequence[torch.Tensor]] = None, record: Optional[list] = None,
    teacher: Optional[Sequence[torch.Tensor]] = None,  # teacher[k] = mean to START refinement k + 1 from (replay aid)
) -> torch.Tensor:
    """MPPIOptimizer.optimize (trajectory_opt.py:238-311), quirks of Appendix B4/B5 included:
    ``past_action`` aliases ``mean[0]`` and is overwritten by the shift (:257-258); ``sigma`` only
    feeds ``constrained_var`` whose product is fully overwritten by the beta-recurrence."""
    H, A = state.mean.shape
    var = sigma**2 * torch.ones_like(lower)
    past_action = state.mean[0]  # view (:257)
    state.mean[:-1] = state.mean[1:].clone()  # :258  (past_action now == old mean[1])
    for k in range(num_iterations):
        if teacher is not None and k > 0:
            state.mean = teacher[k - 1].clone()
        if noise is not None:
            z = noise[k]
        else:
            z = truncated_normal_(torch.empty(population_size, H, A))
        lb_dist = state.mean - lower
        ub_dist = upper - state.mean
        mv = torch.minimum(torch.square(lb_dist / 2), torch.square(ub_dist / 2))
        cv = torch.minimum(mv, var)
        population = z.clone() * torch.sqrt(cv)  # :276 (dead, kept for fidelity)
        population[:, 0, :] = beta * (state.mean[0, :] + z[:, 0, :]) + (1 - beta) * past_action  # :279-282
        for i in range(max(H - 1, 0)):  # :283-287
            population[:, i + 1, :] = beta * (state.mean[i + 1] + z[:, i + 1, :]) + (1 - beta) * population[:, i, :]
        population = torch.where(population > upper, upper, population)  # :290-295
        population = torch.where(population < lower, lower, population)
        values = obj_fun(population).clone()
        values[values.isnan()] = -1e-10  # :297
        weights = torch.reshape(torch.exp(gamma * (values - values.max())), (population_size, 1, 1))  # :303-306
        norm = torch.sum(weights) + 1e-10
        weighted = population * weights
        state.mean = torch.sum(weighted, dim=0) / norm  # :309
        if record is not None:
            record.append(dict(noise=z.clone(), population=population.clone(), values=values.clone(), mean=state.mean.clone()))
    return state.mean.clone()


class ICEMState:
    """Persistent ICEMOptimizer.elite (trajectory_opt.py:374,476)."""

    def __init__(self):
        self.elite: Optional[torch.Tensor] = None


def _round_up(v: int, mod: int) -> int:  # trajectory_opt.py:385-389
    return v if v % mod == 0 else v + (mod - v % mod)


def icem_sizes(num_iterations, elite_ratio, population_size, decay, keep_frac, module):
    """Per-iteration sampled population and kept-elite count (trajectory_opt.py:363-383,419-431)."""
    K = elite_count(population_size, elite_ratio)
    keep = int(np.ceil(keep_frac * K).astype(np.int32))
    if module:
        keep = _round_up(keep, module)
    sizes = []
    for i in range(num_iterations):
        n = int(np.ceil(np.max((population_size * decay**-i, 2 * K))).astype(np.int32))
        if module:
            n = _round_up(n, module)
        sizes.append(n)
    return K, keep, sizes


def icem_optimize(
    obj_fun, state: ICEMState, x0: torch.Tensor, lower: torch.Tensor, upper: torch.Tensor,
    num_iterations: int, elite_ratio: float, population_size: int, population_decay_factor: float,
    colored_noise_exponent: float, keep_elite_frac: float, alpha: float,
    return_mean_elites: bool = False, population_size_module: Optional[int] = None,
    inject: Optional[Sequence[dict]] = None, record: Optional[list] = None,
    teacher: Optional[Sequence[tuple]] = None,  # teacher[i] = (mu, var, elite) to START iteration i + 1 from (replay aid)
) -> torch.Tensor:
    """ICEMOptimizer.optimize (trajectory_opt.py:391-487).  ``inject[i]`` may carry
    ``noise`` [n_i,H,A] (coloured, unit variance, already transposed), ``keep_perm`` [K] and
    ``end_noise`` [keep, A] standard normals for the shifted tail."""
    K, keep, sizes = icem_sizes(num_iterations, elite_ratio, population_size,
                                population_decay_factor, keep_elite_frac, population_size_module)
    H, A = x0.shape
    mu = x0.clone()
    var = (((upper - lower) ** 2) / 16).clone()
    best = torch.empty_like(mu)
    best_val = -np.inf
    for i in range(num_iterations):
        if teacher is not None and i > 0:
            mu, var = teacher[i - 1][0].clone(), teacher[i - 1][1].clone()
            state.elite = teacher[i - 1][2].clone()
        n = sizes[i]
        inj = inject[i] if inject is not None else {}
        rec_i = {}
        if "noise" in inj:
            cn = inj["noise"]
        else:
            rn = []
            cn = powerlaw_psd_gaussian(colored_noise_exponent, size=(n, A, H), normals=inj.get("normals"),
                                       record_normals=rn).transpose(1, 2)  # :433-437
            rec_i["normals"] = torch.stack(rn[0])  # [2, n, A, H/2+1] unit normals
        rec_i["noise"] = cn.clone()
        population = torch.minimum(cn * torch.sqrt(var) + mu, upper)  # :438-440
        population = torch.maximum(population, lower)  # :441
        if state.elite is not None:
            kp = inj["keep_perm"] if "keep_perm" in inj else torch.randperm(K)
            rec_i["keep_perm"] = kp.clone()
            kept = torch.index_select(state.elite, dim=0, index=kp[:keep])  # :443-449
            if i == 0:  # :450-462
                m_ = mu[-1, :].repeat(kept.shape[0], 1)
                s_ = torch.sqrt(var[-1, :]).repeat(kept.shape[0], 1)
                # torch.normal(mean_tensor, std_tensor) == N(0,1) * std + mean on CPU (same RNG stream)
                en = inj["end_noise"] if "end_noise" in inj else torch.empty_like(m_).normal_(0.0, 1.0)
                rec_i["end_noise"] = en.clone()
                end_action = (en * s_ + m_).unsqueeze(1)
                shifted = torch.cat((kept[:, 1:, :], end_action), dim=1)
                population = torch.cat((population, shifted), dim=0)
            elif i == num_iterations - 1:  # :463-464
                population = torch.cat((population, mu.unsqueeze(dim=0)), dim=0)
            else:  # :465-466
                population = torch.cat((population, kept), dim=0)
        values = obj_fun(population).clone()
        values[values.isnan()] = -1e-10  # :474
        best_values, elite_idx = values.topk(K)
        state.elite = population[elite_idx]  # :476
        new_mu = torch.mean(state.elite, dim=0)
        new_var = torch.var(state.elite, unbiased=False, dim=0)  # :479 (biased, Appendix B8)
        mu = alpha * mu + (1 - alpha) * new_mu
        var = alpha * var + (1 - alpha) * new_var
        if best_values[0] > best_val:
            best_val = best_values[0]
            best = population[elite_idx[0]].clone()
        if record is not None:
            rec_i.update(population=population.clone(), values=values.clone(), elite_idx=elite_idx.clone(), mu=mu.clone(),
                         var=var.clone())
            record.append(rec_i)
    return mu if return_mean_elites else best


class TrajectoryOptimizerState:
    """TrajectoryOptimizer warm start (trajectory_opt.py:525-572)."""

    def __init__(self, action_lb: np.ndarray, action_ub: np.ndarray, horizon: int, replan_freq: int = 1,
                 keep_last_solution: bool = True):
        self.lower = torch.tensor(np.tile(action_lb, (horizon, 1)).tolist(), dtype=torch.float32)
        self.upper = torch.tensor(np.tile(action_ub, (horizon, 1)).tolist(), dtype=torch.float32)
        init = ((torch.tensor(action_lb) + torch.tensor(action_ub)) / 2).float()
        self.initial_solution = init.repeat((horizon, 1))
        self.previous_solution = self.initial_solution.clone()
        self.replan_freq = replan_freq
        self.keep_last_solution = keep_last_solution

    def step(self, optimize: Callable[[torch.Tensor], torch.Tensor]) -> np.ndarray:
        best = optimize(self.previous_solution)
        if self.keep_last_solution:  # :563-567
            self.previous_solution = best.roll(-self.replan_freq, dims=0)
            self.previous_solution[-self.replan_freq:] = self.initial_solution[0]
        return best.cpu().numpy()

    def reset(self):
        self.previous_solution = self.initial_solution.clone()


# ----------------------------------------------------------------------------------------
# Synthetic model factory (SURVEY.md section 8d "Synthetic inputs")
# ----------------------------------------------------------------------------------------


def make_synthetic_model(
    obs_dim: int, act_dim: int, ensemble_size: int = 5, hid: int = 200, num_layers: int = 4,
    seed: int = 0, normalizer: str = "f64", nontrivial_stats: bool = True, elite: Optional[Sequence[int]] = None,
    obs_process: str = "none", learned_rewards: bool = False, deterministic: bool = False, **kw,
) -> OracleModel:
    """Random-init ensemble following models/util.py:15-28 (truncated normal, std 1/(2 sqrt(in)),
    zero bias) and gaussian_mlp.py:117-122 (logvar bounds -10 / 0.5).  Uses its own generator so
    it is reproducible on any machine with the same torch build."""
    g = torch.Generator().manual_seed(seed)
    in_obs = {"none": obs_dim, "halfcheetah": obs_dim, "cartpole_pets": obs_dim + 1}[obs_process]
    in_size = in_obs + act_dim
    out_size = obs_dim + (1 if learned_rewards else 0)
    dims = [in_size] + [hid] * num_layers + [out_size * (1 if deterministic else 2)]
    ws, bs = [], []
    for li in range(len(dims) - 1):
        w = torch.empty(ensemble_size, dims[li], dims[li + 1])
        std = 1 / (2 * np.sqrt(dims[li]))
        for e in range(ensemble_size):
            truncated_normal_(w[e], std=float(std), generator=g)
        ws.append(w)
        bs.append(torch.zeros(ensemble_size, 1, dims[li + 1]))
    rng = np.random.default_rng(seed + 1)
    if normalizer == "none":
        nm = ns = None
    else:
        dt = torch.float64 if normalizer == "f64" else torch.float32
        if nontrivial_stats:
            nm = torch.from_numpy(rng.normal(0, 0.1, size=(1, in_size))).to(dt)
            ns = torch.from_numpy(rng.uniform(0.5, 2.0, size=(1, in_size))).to(dt)
        else:
            nm = torch.zeros(1, in_size, dtype=dt)
            ns = torch.ones(1, in_size, dtype=dt)
    return OracleModel(
        weights=ws, biases=bs,
        min_logvar=None if deterministic else -10 * torch.ones(1, out_size),
        max_logvar=None if deterministic else 0.5 * torch.ones(1, out_size),
        elite_models=list(elite) if elite is not None else None,
        norm_mean=nm, norm_std=ns, obs_process=obs_process, learned_rewards=learned_rewards,
        deterministic=deterministic, **kw,
    )
