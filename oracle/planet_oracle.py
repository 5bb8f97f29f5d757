"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the PlaNet latent planning path.

    ModelEnv.evaluate_action_sequences            mbrl/models/model_env.py:145-191
     └ PlaNetModel.reset                          mbrl/models/planet.py:656-672   (saved posterior / belief, tiled)
     └ PlaNetModel.sample                         mbrl/models/planet.py:531-581
        └ BeliefModel.forward                     :83-101   Linear + ReLU -> GRUCell
        └ prior_transition_model                  :229-234  Linear, ReLU, Linear, MeanStdCat (:104-115)
        └ _sample_state_from_params               :288-306  mean + std * randn(generator)
        └ reward_model                            :260-266  Linear, ReLU, Linear, ReLU, Linear

Plain torch-CPU ops on plain tensors, randomness injected (``eps``) or drawn from a generator in the reference's
order.  Pinned bitwise against the unmodified reference classes in tests/test_oracle_vs_reference.py and through the
golden vectors tests/golden/planet_*.npz (oracle/make_golden.py).  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module; the product never does.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch
import torch.nn.functional as F


@dataclass
class PlaNetOracleModel:
    """The tensors PlaNetModel.sample reads, in nn.Linear layout ([out, in] weights, [out] biases)."""

    w_embed: torch.Tensor  # belief_model.embedding_layer[0]: [belief, latent + action]
    b_embed: torch.Tensor
    w_ih: torch.Tensor  # belief_model.rnn (GRUCell): [3 * belief, belief], gate order r | z | n
    b_ih: torch.Tensor
    w_hh: torch.Tensor
    b_hh: torch.Tensor
    w_prior1: torch.Tensor  # prior_transition_model[0]: [hidden, belief]
    b_prior1: torch.Tensor
    w_prior2: torch.Tensor  # prior_transition_model[2]: [2 * latent, hidden]
    b_prior2: torch.Tensor
    w_rew1: torch.Tensor  # reward_model[0]: [hidden, belief + latent]
    b_rew1: torch.Tensor
    w_rew2: torch.Tensor  # reward_model[2]: [hidden, hidden]
    b_rew2: torch.Tensor
    w_rew3: torch.Tensor  # reward_model[4]: [1, hidden]
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
        ws = (self.w_embed, self.w_ih, self.w_hh, self.w_prior1, self.w_prior2, self.w_rew1, self.w_rew2, self.w_rew3)
        return 2 * sum(int(w.shape[0]) * int(w.shape[1]) for w in ws)


def planet_step(m: PlaNetOracleModel, latent: torch.Tensor, belief: torch.Tensor, action: torch.Tensor,
                eps: Optional[torch.Tensor] = None, generator: Optional[torch.Generator] = None, deterministic: bool = False):
    """PlaNetModel.sample (planet.py:531-581): returns (next_latent [B, latent], reward [B, 1], next_belief [B, belief])."""
    emb = F.relu(F.linear(torch.cat([latent, action], dim=1), m.w_embed, m.b_embed))  # :94-100
    next_belief = torch.gru_cell(emb, belief, m.w_ih, m.w_hh, m.b_ih, m.b_hh)  # nn.GRUCell.forward
    params = F.linear(F.relu(F.linear(next_belief, m.w_prior1, m.b_prior1)), m.w_prior2, m.b_prior2)
    L = m.latent_size
    mean = params[:, :L]
    std = F.softplus(params[:, L:]) + m.min_std  # MeanStdCat, :111-115
    if deterministic:
        next_latent = mean
    else:
        if eps is None:  # :299-305
            eps = torch.randn(mean.size(), dtype=mean.dtype, device=mean.device, generator=generator)
        next_latent = mean + std * eps
    h = F.relu(F.linear(torch.cat([next_belief, next_latent], dim=1), m.w_rew1, m.b_rew1))
    h = F.relu(F.linear(h, m.w_rew2, m.b_rew2))
    reward = F.linear(h, m.w_rew3, m.b_rew3)
    return next_latent, reward, next_belief


def planet_rollout(m: PlaNetOracleModel, actions: torch.Tensor, latent0: torch.Tensor, belief0: torch.Tensor, num_particles: int,
                   eps: Optional[torch.Tensor] = None, generator: Optional[torch.Generator] = None,
                   trace: Optional[dict] = None) -> torch.Tensor:
    """ModelEnv.evaluate_action_sequences on a PlaNetModel (model_env.py:145-191): ``latent0`` [1, latent] /
    ``belief0`` [1, belief] are the model's saved posterior sample and belief (planet.py:669-672); termination is
    no_termination (mbrl/algorithms/planet.py), rewards come from the reward head.  ``eps`` [H, B, latent] injects the
    draws of planet.py:299-305.  Returns the particle-averaged returns [pop]."""
    pop, H, _ = actions.shape
    P = num_particles
    B = pop * P
    latent = latent0.reshape(1, -1).repeat(B, 1)
    belief = belief0.reshape(1, -1).repeat(B, 1)
    total = torch.zeros(B, 1)
    terminated = torch.zeros(B, 1, dtype=torch.bool)
    for t in range(H):
        a = torch.repeat_interleave(actions[:, t, :], P, dim=0)  # model_env.py:179-182
        latent, rew, belief = planet_step(m, latent, belief, a, None if eps is None else eps[t], generator)
        if trace is not None:
            trace.setdefault("latent", []).append(latent.clone())
            trace.setdefault("belief", []).append(belief.clone())
            trace.setdefault("rewards", []).append(rew.clone())
        rew = rew.clone()
        rew[terminated] = 0  # :186 (never set: no_termination)
        total += rew  # :188
    return total.reshape(-1, P).mean(dim=1)  # :190-191


def make_synthetic_planet(latent: int = 30, action: int = 6, belief: int = 200, hidden: int = 200, seed: int = 0,
                          min_std: float = 0.1, scale: float = 1.0) -> PlaNetOracleModel:
    """Random PlaNet heads with the reference's initialisers (planet.py:20-30: orthogonal W_hh, Xavier-uniform everything
    else, zero biases -- biases get small random values here so that parity tests exercise them)."""
    g = torch.Generator().manual_seed(seed)

    def xavier(out_f, in_f):
        bound = float(np.sqrt(6.0 / (in_f + out_f))) * scale
        return (torch.rand(out_f, in_f, generator=g) * 2 - 1) * bound

    def bias(n):
        return (torch.rand(n, generator=g) * 2 - 1) * 0.05

    q, _ = torch.linalg.qr(torch.randn(3 * belief, belief, generator=g))
    return PlaNetOracleModel(
        w_embed=xavier(belief, latent + action), b_embed=bias(belief),
        w_ih=xavier(3 * belief, belief), b_ih=bias(3 * belief), w_hh=q.contiguous(), b_hh=bias(3 * belief),
        w_prior1=xavier(hidden, belief), b_prior1=bias(hidden), w_prior2=xavier(2 * latent, hidden), b_prior2=bias(2 * latent),
        w_rew1=xavier(hidden, belief + latent), b_rew1=bias(hidden), w_rew2=xavier(hidden, hidden), b_rew2=bias(hidden),
        w_rew3=xavier(1, hidden), b_rew3=bias(1), min_std=min_std,
    )


PLANET_TENSORS = ("w_embed", "b_embed", "w_ih", "b_ih", "w_hh", "b_hh", "w_prior1", "b_prior1", "w_prior2", "b_prior2",
                  "w_rew1", "b_rew1", "w_rew2", "b_rew2", "w_rew3", "b_rew3")
