"""TEST INFRASTRUCTURE ONLY: build the *unmodified reference classes* from an OracleModel.

Works only where /root/reference is mounted (this build container).  It puts the stand-in
modules of ``oracle/ref_stubs`` (hydra / omegaconf / gymnasium / termcolor -- absent here, no
network) and the reference root on ``sys.path`` and imports ``mbrl`` untouched.  Used by
``oracle/make_golden.py`` and ``tests/test_oracle_vs_reference.py``; never by the product, the
GPU tests, smoke() or bench.py (the reference does not exist on the GPU box).
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

REFERENCE_ROOT = os.environ.get("HIPETS_REFERENCE_ROOT", "/root/reference")
_STUBS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_stubs")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "mbrl"))


def import_reference():
    if not reference_available():
        raise RuntimeError(f"reference not mounted at {REFERENCE_ROOT}")
    for p in (REFERENCE_ROOT, _STUBS):
        if p in sys.path:
            sys.path.remove(p)
    sys.path.insert(0, REFERENCE_ROOT)
    sys.path.insert(0, _STUBS)
    import mbrl.models  # noqa: F401
    import mbrl.planning  # noqa: F401
    import mbrl.util.math  # noqa: F401
    import mbrl  # noqa: F401

    return mbrl


_ACT_TARGET = {
    "silu": "torch.nn.SiLU",
    "relu": "torch.nn.ReLU",
    "leaky_relu": "torch.nn.LeakyReLU",
    "tanh": "torch.nn.Tanh",
    "sigmoid": "torch.nn.Sigmoid",
}


class _FakeEnv:
    def __init__(self, obs_dim, act_dim, lo=-1.0, hi=1.0):
        import gymnasium as gym

        self.observation_space = gym.spaces.Box(-np.inf, np.inf, shape=(obs_dim,))
        self.action_space = gym.spaces.Box(lo, hi, shape=(act_dim,))


def build_reference_model_env(om, obs_dim: int, act_dim: int, generator=None):
    """OracleModel -> (mbrl.models.ModelEnv, OneDTransitionRewardModel, GaussianMLP)."""
    mbrl = import_reference()
    from mbrl.env import reward_fns, termination_fns
    import mbrl.env.pets_halfcheetah  # noqa: F401  (needs the gymnasium stand-in only)

    E = om.weights[0].shape[0]
    in_size = om.weights[0].shape[1]
    hid = om.weights[0].shape[2]
    out_total = om.weights[-1].shape[2]
    out_size = out_total if om.deterministic else out_total // 2
    def fill(gmlp, members):
        with torch.no_grad():
            for li in range(len(om.weights) - 1):
                gmlp.hidden_layers[li][0].weight.copy_(om.weights[li][members])
                gmlp.hidden_layers[li][0].bias.copy_(om.biases[li][members])
            gmlp.mean_and_logvar.weight.copy_(om.weights[-1][members])
            gmlp.mean_and_logvar.bias.copy_(om.biases[-1][members])
            if not om.deterministic:
                per_member = om.min_logvar.shape[0] > 1  # BasicEnsemble: each member owns its bounds
                gmlp.min_logvar.copy_(om.min_logvar[members] if per_member else om.min_logvar)
                gmlp.max_logvar.copy_(om.max_logvar[members] if per_member else om.max_logvar)

    if getattr(om, "ensemble_kind", "gaussian_mlp") == "basic_ensemble":
        # conf/dynamics_model/basic_ensemble.yaml: E single-member GaussianMLPs built by hydra.utils.instantiate
        member_cfg = {
            "_target_": "mbrl.models.GaussianMLP", "device": "cpu", "num_layers": len(om.weights) - 1, "in_size": in_size,
            "out_size": out_size, "hid_size": hid, "deterministic": om.deterministic,
            "activation_fn_cfg": {"_target_": _ACT_TARGET[om.activation]},
        }
        model = mbrl.models.BasicEnsemble(E, "cpu", member_cfg, propagation_method=om.propagation)
        for i, member in enumerate(model.members):
            fill(member, [i])
    else:
        model = mbrl.models.GaussianMLP(
            in_size, out_size, "cpu", num_layers=len(om.weights) - 1, ensemble_size=E, hid_size=hid,
            deterministic=om.deterministic, propagation_method=om.propagation,
            activation_fn_cfg={"_target_": _ACT_TARGET[om.activation]},
        )
        fill(model, list(range(E)))
        if om.elite_models is not None:
            model.set_elite(list(om.elite_models))
    obs_fn = None
    if om.obs_process == "halfcheetah":
        from mbrl.env.pets_halfcheetah import HalfCheetahEnv

        obs_fn = HalfCheetahEnv.preprocess_fn
    elif om.obs_process == "cartpole_pets":
        from mbrl.env.pets_cartpole import CartPoleEnv

        obs_fn = CartPoleEnv.preprocess_fn
    dm = mbrl.models.OneDTransitionRewardModel(
        model, target_is_delta=om.target_is_delta, normalize=om.norm_mean is not None,
        normalize_double_precision=(om.norm_mean is not None and om.norm_mean.dtype == torch.float64),
        learned_rewards=om.learned_rewards, obs_process_fn=obs_fn,
        no_delta_list=list(om.no_delta_list) if om.no_delta_list else None,
    )
    if om.norm_mean is not None:
        dm.input_normalizer.mean = om.norm_mean.clone()
        dm.input_normalizer.std = om.norm_std.clone()
    reward_fn = getattr(reward_fns, om.reward) if om.reward is not None else None
    term_fn = getattr(termination_fns, om.termination)
    env = _FakeEnv(obs_dim, act_dim)
    me = mbrl.models.ModelEnv(env, dm, term_fn, reward_fn, generator=generator)
    return me, dm, model


def build_reference_planet_env(pm, latent0, belief0, generator=None, obs_hw: int = 16):
    """PlaNetOracleModel -> (mbrl.models.ModelEnv, PlaNetModel) of the unmodified reference, with the planning heads
    overwritten by ``pm``'s tensors and the saved posterior / belief set as update_posterior would leave them
    (planet.py:600-640).  The conv encoder / decoder are irrelevant to planning and kept tiny."""
    mbrl = import_reference()
    from mbrl.env import termination_fns

    model = mbrl.models.PlaNetModel(
        obs_shape=(3, obs_hw, obs_hw), obs_encoding_size=32, encoder_config=((3, 4, 4, 2),),
        decoder_config=((8, 1, 1), ((8, 3, obs_hw, 1),)), latent_state_size=pm.latent_size, action_size=pm.action_size,
        belief_size=pm.belief_size, hidden_size_fcs=pm.hidden_size, device="cpu", min_std=pm.min_std,
    )
    with torch.no_grad():
        model.belief_model.embedding_layer[0].weight.copy_(pm.w_embed)
        model.belief_model.embedding_layer[0].bias.copy_(pm.b_embed)
        model.belief_model.rnn.weight_ih.copy_(pm.w_ih)
        model.belief_model.rnn.bias_ih.copy_(pm.b_ih)
        model.belief_model.rnn.weight_hh.copy_(pm.w_hh)
        model.belief_model.rnn.bias_hh.copy_(pm.b_hh)
        for seq, names in ((model.prior_transition_model, ("prior1", None, "prior2")),
                           (model.reward_model, ("rew1", None, "rew2", None, "rew3"))):
            for i, n in enumerate(names):
                if n is not None:
                    seq[i].weight.copy_(getattr(pm, "w_" + n))
                    seq[i].bias.copy_(getattr(pm, "b_" + n))
    model._current_posterior_sample = latent0.reshape(1, -1).clone()
    model._current_belief = belief0.reshape(1, -1).clone()

    class _Env:
        import gymnasium as gym

        observation_space = gym.spaces.Box(0, 255, shape=(3, obs_hw, obs_hw))
        action_space = gym.spaces.Box(-1.0, 1.0, shape=(pm.action_size,))

    me = mbrl.models.ModelEnv(_Env(), model, termination_fns.no_termination, generator=generator)  # algorithms/planet.py
    return me, model
