"""The Hydra seam (SURVEY.md 8b seams 1-2) with the STOCK config values: conf/action_optimizer/{cem,icem,mppi}.yaml resolved
against the overrides the reference ships for them, presented the way `hydra.utils.instantiate` receives them at
trajectory_opt.py:527 / :741 (an OmegaConf-like node whose `???` fields raise when read), then built through
hipets.create_trajectory_optim_agent_for_model -- once with the stock `_target_: mbrl.planning.*` (redirected) and once with the
documented opt-in override `_target_=hipets.*`.

hydra-core / omegaconf / the reference tree are not on the GPU box, so the YAML texts are embedded here verbatim; the CPU test
below compares them with the files under /root/reference whenever that tree is mounted (it is, where this suite runs without
a GPU), so the embedded copies cannot drift from what the reference ships."""
import math
import os
import re

import numpy as np
import pytest
import torch
import yaml

REF_CONF = "/root/reference/mbrl/examples/conf"

# conf/action_optimizer/*.yaml, verbatim
ACTION_OPTIMIZER_YAML = {
    "cem": """# @package _group_

_target_: mbrl.planning.CEMOptimizer
num_iterations: ${overrides.cem_num_iters}
elite_ratio: ${overrides.cem_elite_ratio}
population_size: ${overrides.cem_population_size}
alpha: ${overrides.cem_alpha}
lower_bound: ???
upper_bound: ???
return_mean_elites: true
device: ${device}
clipped_normal: ${overrides.cem_clipped_normal}
""",
    "icem": """# @package _group_

_target_: mbrl.planning.ICEMOptimizer
num_iterations: ${overrides.cem_num_iters}
elite_ratio: ${overrides.cem_elite_ratio}
population_size: ${overrides.cem_population_size}
population_decay_factor: ${overrides.cem_population_decay_factor}
colored_noise_exponent: ${overrides.cem_colored_noise_exponent}
keep_elite_frac: ${overrides.cem_keep_elite_frac}
alpha: ${overrides.cem_alpha}
lower_bound: ???
upper_bound: ???
return_mean_elites: true
population_size_module: ${dynamics_model.ensemble_size}
device: ${device}
""",
    "mppi": """# @package _group_

_target_: mbrl.planning.MPPIOptimizer
num_iterations: ${overrides.mppi_num_iters}
gamma: ${overrides.mppi_gamma}
population_size: ${overrides.mppi_population_size}
sigma: ${overrides.mppi_sigma}
beta: ${overrides.mppi_beta}
lower_bound: ???
upper_bound: ???
device: ${device}
""",
}
# the planner keys of the override files the reference pairs with each optimizer (file: key lines)
OVERRIDES = {
    "cem": ("overrides/pets_halfcheetah.yaml", dict(planning_horizon=30, cem_num_iters=5, cem_elite_ratio=0.16, cem_population_size=400,
                                                    cem_alpha=0.12, cem_clipped_normal=False)),
    "icem": ("overrides/pets_icem_cartpole.yaml", dict(planning_horizon=10, cem_num_iters=5, cem_elite_ratio=0.1, cem_population_size=200,
                                                       cem_population_decay_factor=1.3, cem_colored_noise_exponent=2, cem_keep_elite_frac=0.3,
                                                       cem_alpha=0.1)),
    "mppi": ("overrides/pets_mppi_halfcheetah.yaml", dict(planning_horizon=30, mppi_num_iters=5, mppi_population_size=350, mppi_gamma=0.9,
                                                          mppi_sigma=1.0, mppi_beta=0.9)),
}
# conf/algorithm/pets.yaml: the agent node and num_particles
PETS_AGENT = dict(_target_="mbrl.planning.TrajectoryOptimizerAgent", action_lb="???", action_ub="???", replan_freq=1, verbose=False)
NUM_PARTICLES = 20   # conf/algorithm/pets.yaml:20
ENSEMBLE_SIZE = 7    # conf/dynamics_model/gaussian_mlp_ensemble.yaml: ensemble_size (what ${dynamics_model.ensemble_size} resolves to)


def resolved(name, device):
    """The action_optimizer node after OmegaConf interpolation: ${overrides.x}, ${device}, ${dynamics_model.ensemble_size}."""
    node = yaml.safe_load(ACTION_OPTIMIZER_YAML[name])
    scope = {"overrides": OVERRIDES[name][1], "device": device, "dynamics_model": {"ensemble_size": ENSEMBLE_SIZE}}
    out = {}
    for k, v in node.items():
        m = re.fullmatch(r"\$\{([a-z_.]+)\}", v) if isinstance(v, str) else None
        if m:
            cur = scope
            for part in m.group(1).split("."):
                cur = cur[part] if isinstance(cur, dict) else getattr(cur, part)
            v = cur
        out[k] = v
    return out


@pytest.mark.skipif(not os.path.isdir(REF_CONF), reason="the reference tree is not mounted (GPU box)")
def test_embedded_yaml_is_what_the_reference_ships():
    for name, text in ACTION_OPTIMIZER_YAML.items():
        assert open(os.path.join(REF_CONF, "action_optimizer", f"{name}.yaml")).read().strip() == text.strip(), name
    for name, (path, values) in OVERRIDES.items():
        shipped = yaml.safe_load(open(os.path.join(REF_CONF, path)))
        for k, v in values.items():
            assert shipped[k] == v, (path, k)
    pets = yaml.safe_load(open(os.path.join(REF_CONF, "algorithm", "pets.yaml")))
    assert pets["num_particles"] == NUM_PARTICLES
    assert {k: pets["agent"][k] for k in ("_target_", "action_lb", "action_ub", "replan_freq")} == {k: PETS_AGENT[k] for k in ("_target_", "action_lb", "action_ub", "replan_freq")}
    gmlp = yaml.safe_load(open(os.path.join(REF_CONF, "dynamics_model", "gaussian_mlp_ensemble.yaml")))
    assert gmlp["ensemble_size"] == ENSEMBLE_SIZE


def test_resolver_produces_plain_values():
    cem = resolved("cem", "cpu")
    assert cem["population_size"] == 400 and cem["elite_ratio"] == 0.16 and cem["lower_bound"] == "???" and cem["device"] == "cpu"
    icem = resolved("icem", "cpu")
    assert icem["population_size_module"] == ENSEMBLE_SIZE and icem["colored_noise_exponent"] == 2
    assert resolved("mppi", "cpu")["sigma"] == 1.0


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["cem", "icem", "mppi"])
@pytest.mark.parametrize("target", ["stock", "hipets"])
def test_stock_optimizer_configs_build_and_plan_through_the_hydra_seam(name, target):
    import hipets
    from test_host_logic import _DictConfigLike, _FakeMLP, _FakeModelEnv

    dev = "cuda:0"
    me = _FakeModelEnv()
    me.device = dev
    # five members like the elite set the stock overrides train with (num_elites: 5): 20 particles x any population is a
    # multiple of 5 (gaussian_mlp.py:195-200; SURVEY Appendix B7 -- with all 7 members iCEM's last iteration could not run)
    E, obs, act, hid = 5, 5, 2, 8
    g = torch.Generator().manual_seed(0)
    five = hipets.ModelSpec(weights=[torch.randn(E, obs + act, hid, generator=g) * 0.3, torch.randn(E, hid, hid, generator=g) * 0.3,
                                     torch.randn(E, hid, 2 * obs, generator=g) * 0.1],
                            biases=[torch.zeros(E, 1, hid), torch.zeros(E, 1, hid), torch.zeros(E, 1, 2 * obs)], obs_dim=obs, act_dim=act,
                            min_logvar=-10 * torch.ones(1, obs), max_logvar=0.5 * torch.ones(1, obs))
    me.dynamics_model.model = _FakeMLP(five, torch.nn.SiLU())
    opt_cfg = resolved(name, dev)
    if target == "hipets":  # the documented opt-in: action_optimizer._target_=hipets.<Class>, algorithm.agent._target_=hipets.TrajectoryOptimizerAgent
        opt_cfg["_target_"] = opt_cfg["_target_"].replace("mbrl.planning", "hipets")
    horizon = OVERRIDES[name][1]["planning_horizon"]
    agent_cfg = dict(PETS_AGENT, planning_horizon=horizon, optimizer_cfg=_DictConfigLike(**opt_cfg))
    if target == "hipets":
        agent_cfg["_target_"] = "hipets.TrajectoryOptimizerAgent"
    agent_cfg = _DictConfigLike(**agent_cfg)
    agent = hipets.create_trajectory_optim_agent_for_model(me, agent_cfg, num_particles=NUM_PARTICLES)
    opt = agent.optimizer.optimizer
    v = OVERRIDES[name][1]
    assert type(opt) is {"cem": hipets.CEMOptimizer, "icem": hipets.ICEMOptimizer, "mppi": hipets.MPPIOptimizer}[name]
    assert tuple(opt.lower_bound.shape) == (horizon, 2) and float(opt.lower_bound.min()) == -1.0 and float(opt.upper_bound.max()) == 1.0
    if name == "cem":
        assert (opt.num_iterations, opt.population_size, opt.alpha, opt.return_mean_elites) == (5, 400, 0.12, True)
        assert int(opt.elite_num) == math.ceil(400 * 0.16) == 64  # np.ceil, trajectory_opt.py:89-91
    elif name == "icem":
        assert (opt.num_iterations, opt.population_size, opt.alpha) == (5, 200, 0.1)
        assert int(opt.elite_num) == 20 and opt.population_size_module == ENSEMBLE_SIZE
        assert int(opt.keep_elite_size) == 7  # ceil(0.3 * 20) = 6, rounded up to the module 7 (trajectory_opt.py:371-383)
    else:
        assert (opt.refinements, opt.population_size, opt.gamma, opt.beta) == (5, 350, 0.9, 0.9)
    obs = np.zeros(5, np.float32)
    a0 = agent.act(obs)
    a1 = agent.act(obs)  # a second plan: warm start, persistent optimizer state
    assert a0.shape == (2,) and np.isfinite(a0).all() and np.isfinite(a1).all() and (np.abs(a0) <= 1).all()
    plan = agent.plan(obs)
    assert plan.shape == (horizon, 2)
    agent.reset()
