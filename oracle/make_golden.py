"""TEST INFRASTRUCTURE ONLY: generate tests/golden/*.npz by running the UNMODIFIED reference
(/root/reference, through oracle/ref_stubs) with seeded RNG and recording the random draws it
consumed (in its consumption order) next to its outputs.

    python -m oracle.make_golden            # from the repo root, in the build container

The fixtures pin the oracle (and the HIP engine) on machines where the reference is absent.
While generating, the script also asserts oracle == reference bitwise for every case.
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import pets_oracle as po  # noqa: E402
from oracle.golden_io import save_case  # noqa: E402
from oracle.ref_bridge import build_reference_model_env, import_reference  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")

ROLLOUT_CASES = {
    # name: (obs, act, model kwargs, pop, P, H)
    "ts1_halfcheetah": (17, 6, dict(ensemble_size=5, hid=64, seed=10, no_delta_list=[0]), 30, 5, 5),
    "ts1_cartpole_term": (4, 1, dict(ensemble_size=3, hid=200, seed=11, reward="cartpole", termination="cartpole"), 21, 5, 15),
    "tsinf_elite_humanoid": (45, 17, dict(ensemble_size=7, hid=32, seed=12, elite=[0, 2, 3, 5, 6], propagation="fixed_model",
                                            termination="humanoid"), 10, 5, 4),
    "expectation_learned_rew": (18, 6, dict(ensemble_size=4, hid=48, seed=13, propagation="expectation", learned_rewards=True,
                                            reward=None, obs_process="halfcheetah", normalizer="f32", activation="relu"), 12, 3, 6),
    "deterministic_cartpole_pets": (4, 1, dict(ensemble_size=2, hid=40, seed=14, deterministic=True, obs_process="cartpole_pets",
                                               reward="cartpole_pets", normalizer="none", activation="leaky_relu",
                                               target_is_delta=False), 16, 4, 7),
    "ts1_hopper_tanh": (11, 3, dict(ensemble_size=5, hid=24, num_layers=2, seed=15, activation="tanh", termination="hopper",
                                    reward="halfcheetah"), 25, 4, 8),
    # BasicEnsemble of single-member GaussianMLPs (conf/dynamics_model/basic_ensemble.yaml): iid randint member maps,
    # batch sizes that are NOT multiples of the ensemble size
    "basic_tsinf_hopper": (11, 3, dict(ensemble_size=5, hid=32, seed=16, ensemble_kind="basic_ensemble", propagation="fixed_model",
                                       termination="hopper"), 13, 3, 10),
    "basic_ts1_halfcheetah": (17, 6, dict(ensemble_size=3, hid=40, seed=17, ensemble_kind="basic_ensemble", no_delta_list=[0]),
                              11, 4, 5),
    "basic_expectation": (8, 2, dict(ensemble_size=3, hid=24, seed=18, ensemble_kind="basic_ensemble", propagation="expectation",
                                     normalizer="f32"), 7, 2, 4),
}


def gen_rollout(name, obs, act, mkw, pop, P, H):
    om = po.make_synthetic_model(obs, act, **mkw)
    g = torch.Generator().manual_seed(100 + len(name))
    # termination cases: start near the boundary so that some particles terminate mid-horizon
    scale = 0.1
    s0 = (np.random.default_rng(5).standard_normal(obs) * scale).astype(np.float32)
    if om.termination == "humanoid":
        s0[0] = 1.05
    if om.termination == "hopper":
        s0[0] = 0.75
    actions = torch.rand(pop, H, act, generator=g) * 2 - 1
    B = pop * P
    gen = torch.Generator().manual_seed(7)
    me, dm, model = build_reference_model_env(om, obs, act, generator=gen)
    torch.manual_seed(1234)
    ref = me.evaluate_action_sequences(actions, s0, P)
    # replay the reference's RNG consumption to record the draws (Appendix A.4)
    torch.manual_seed(1234)
    gen2 = torch.Generator().manual_seed(7)
    arrays = dict(actions=actions, s0=s0, returns=ref)
    out = om.out_size
    perms = members = None
    basic = om.ensemble_kind == "basic_ensemble"
    M = len(om.active_members)
    if om.propagation == "fixed_model":
        if basic:  # reset: randint from ModelEnv's generator (basic_ensemble.py:255-260)
            members = torch.randint(M, (B,), generator=gen2)
        else:
            perms = torch.randperm(B)  # reset: sample_propagation_indices ignores the generator (gaussian_mlp.py:375)
    plist, mlist, elist = [], [], []
    for _ in range(H):
        if om.propagation == "random_model":
            if basic:  # basic_ensemble.py:122-129: generator draw, before this step's normal
                mlist.append(torch.randint(M, (B,), generator=gen2))
            else:
                plist.append(torch.randperm(B))
        if not om.deterministic:
            elist.append(torch.empty(B, out).normal_(0, 1, generator=gen2))
    if plist:
        perms = torch.stack(plist)
    if mlist:
        members = torch.stack(mlist)
    eps = torch.stack(elist) if elist else None
    trace = {}
    mine = po.rollout(om, actions, s0, P, perms=perms, eps=eps, members=members, trace=trace)
    assert torch.equal(ref, mine), f"{name}: oracle != reference (max diff {(ref - mine).abs().max()})"
    if members is not None:
        arrays["members"] = members
    if perms is not None:
        arrays["perms"] = perms
    if eps is not None:
        arrays["eps"] = eps
    arrays["next_obs_step0"] = trace["next_obs"][0]
    arrays["rewards_step0"] = trace["rewards"][0]
    n_term = int(torch.stack(trace["dones"]).any(0).sum())
    save_case(os.path.join(OUT, f"rollout_{name}.npz"), om, dict(kind="rollout", obs_dim=obs, act_dim=act, pop=pop, P=P, H=H,
                                                                  rows_terminated=n_term), arrays)
    print(f"rollout_{name}: B={B} returns[{ref.min():.3f},{ref.max():.3f}] terminated rows={n_term}  oracle==reference bitwise")


def gen_cem(name, clipped, return_mean, pop=40, H=6, A=3, iters=4, ratio=0.15, alpha=0.1):
    mbrl = import_reference()
    torch.manual_seed(77)
    lb = [[-1.0, -0.5, -2.0][:A]] * H
    ub = [[1.0, 1.5, 0.5][:A]] * H
    target = torch.linspace(-0.4, 0.4, H * A).view(H, A)

    def obj(x):  # smooth objective with a few NaNs to pin the NaN -> -1e-10 rule
        v = -((x - target) ** 2).sum(dim=(1, 2))
        v = v.clone()
        v[3] = float("nan")
        return v

    opt = mbrl.planning.CEMOptimizer(iters, ratio, pop, lb, ub, alpha, "cpu", return_mean_elites=return_mean,
                                     clipped_normal=clipped)
    x0 = torch.zeros(H, A)
    pops, vals = [], []

    def cb(population, values, i):
        pops.append(population.clone())
        vals.append(values.clone())

    torch.manual_seed(5)
    ref = opt.optimize(obj, x0=x0, callback=cb)
    # recover z of every iteration by replaying the same RNG stream
    torch.manual_seed(5)
    zs = []
    for i in range(iters):
        if clipped:
            zs.append(torch.randn(pop, H, A))
        else:
            zs.append(po.truncated_normal_(torch.zeros(pop, H, A)))
    rec = []
    mine = po.cem_optimize(obj, x0, torch.tensor(lb), torch.tensor(ub), iters, ratio, pop, alpha, return_mean_elites=return_mean,
                           clipped_normal=clipped, noise=zs, record=rec)
    assert torch.equal(ref, mine), f"{name}: oracle CEM != reference"
    for i in range(iters):
        assert torch.equal(rec[i]["population"], pops[i])
    arrays = dict(z=torch.stack(zs), lower=torch.tensor(lb), upper=torch.tensor(ub), x0=x0, target=target, result=ref,
                  populations=torch.stack(pops), values=torch.stack([r["values"] for r in rec]),
                  mus=torch.stack([r["mu"] for r in rec]), disps=torch.stack([r["disp"] for r in rec]),
                  elite_idx=torch.stack([r["elite_idx"] for r in rec]))
    meta = dict(kind="cem", pop=pop, H=H, A=A, iters=iters, elite_ratio=ratio, alpha=alpha, clipped=clipped,
                return_mean=return_mean, nan_index=3, n_layers=0)
    d = {"x_" + k: v.numpy() for k, v in arrays.items()}
    import json

    d["meta_json"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez(os.path.join(OUT, f"cem_{name}.npz"), **d)
    print(f"cem_{name}: oracle==reference bitwise; result[0]={ref[0].tolist()}")


def _save_npz(name, meta, arrays):
    import json

    d = {"x_" + k: (v.numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in arrays.items()}
    d["meta_json"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez(os.path.join(OUT, name), **d)


def _quad(target, nan_at=None):
    def f(x):
        v = -((x - target) ** 2).sum(dim=(1, 2)).clone()
        if nan_at is not None:
            v[nan_at] = float("nan")
        return v

    return f


def gen_mppi(pop=48, H=7, A=3, iters=3, gamma=0.9, sigma=1.0, beta=0.9, calls=2):
    """Two consecutive MPPIOptimizer.optimize calls of the reference (persistent mean, Appendix B4-B6)."""
    mbrl = import_reference()
    lb, ub = [[-1.0, -0.5, -2.0][:A]] * H, [[1.0, 1.5, 0.5][:A]] * H
    target = torch.linspace(-0.3, 0.4, H * A).view(H, A)
    obj = _quad(target, nan_at=5)
    opt = mbrl.planning.MPPIOptimizer(iters, pop, gamma, sigma, beta, lb, ub, "cpu")
    torch.manual_seed(21)
    refs = [opt.optimize(obj) for _ in range(calls)]
    torch.manual_seed(21)
    st = po.MPPIState(H, A)
    arrays = dict(lower=torch.tensor(lb), upper=torch.tensor(ub), target=target)
    for c in range(calls):
        rec = []
        mine = po.mppi_optimize(obj, st, torch.tensor(lb), torch.tensor(ub), iters, pop, gamma, sigma, beta, record=rec)
        assert torch.equal(mine, refs[c]), "oracle MPPI != reference"
        arrays[f"noise{c}"] = torch.stack([r["noise"] for r in rec])
        arrays[f"populations{c}"] = torch.stack([r["population"] for r in rec])
        arrays[f"means{c}"] = torch.stack([r["mean"] for r in rec])
        arrays[f"result{c}"] = refs[c]
    _save_npz("mppi_two_calls.npz", dict(kind="mppi", pop=pop, H=H, A=A, iters=iters, gamma=gamma, sigma=sigma, beta=beta,
                                         calls=calls, nan_index=5), arrays)
    print(f"mppi_two_calls: oracle==reference bitwise; result[0]={refs[-1][0].tolist()}")


def gen_icem(pop=60, H=8, A=3, iters=4, ratio=0.1, decay=1.3, exponent=2.0, keep=0.3, alpha=0.1, module=5, calls=2):
    """Two consecutive ICEMOptimizer.optimize calls of the reference (persistent elite set, kept / shifted elites)."""
    mbrl = import_reference()
    lb, ub = [[-1.0] * A] * H, [[1.0, 1.5, 0.5][:A]] * H
    target = torch.full((H, A), -0.1)
    obj = _quad(target, nan_at=1)
    kw = dict(num_iterations=iters, elite_ratio=ratio, population_size=pop, population_decay_factor=decay,
              colored_noise_exponent=exponent, keep_elite_frac=keep, alpha=alpha)
    opt = mbrl.planning.ICEMOptimizer(lower_bound=lb, upper_bound=ub, device="cpu", return_mean_elites=True,
                                      population_size_module=module, **kw)
    torch.manual_seed(31)
    x0s, refs = [torch.zeros(H, A)], []
    for c in range(calls):
        refs.append(opt.optimize(obj, x0=x0s[-1]))
        x0s.append(refs[-1].clone())
    torch.manual_seed(31)
    st = po.ICEMState()
    arrays = dict(lower=torch.tensor(lb), upper=torch.tensor(ub), target=target)
    sizes = []
    for c in range(calls):
        rec = []
        mine = po.icem_optimize(obj, st, x0s[c], torch.tensor(lb), torch.tensor(ub), return_mean_elites=True,
                                population_size_module=module, record=rec, **kw)
        assert torch.equal(mine, refs[c]), "oracle iCEM != reference"
        arrays[f"x0_{c}"] = x0s[c]
        arrays[f"result{c}"] = refs[c]
        for i, r in enumerate(rec):
            for k in ("normals", "noise", "population", "mu", "var", "keep_perm", "end_noise", "elite_idx"):
                if k in r:
                    arrays[f"{k}_{c}_{i}"] = r[k]
        sizes.append([int(r["population"].shape[0]) for r in rec])
    _save_npz("icem_two_calls.npz", dict(kind="icem", pop=pop, H=H, A=A, iters=iters, elite_ratio=ratio, decay=decay,
                                         exponent=exponent, keep_frac=keep, alpha=alpha, module=module, calls=calls,
                                         nan_index=1, evaluated_sizes=sizes), arrays)
    print(f"icem_two_calls: oracle==reference bitwise; evaluated population sizes {sizes}")


def gen_planet(name, latent, action, belief, hidden, pop, P, H, seed):
    """PlaNet latent rollouts through the unmodified reference (ModelEnv + PlaNetModel), draws recorded by replaying the
    generator (one randn [B, latent] per step, planet.py:299-305)."""
    from oracle import planet_oracle as pl
    from oracle.ref_bridge import build_reference_planet_env

    pm = pl.make_synthetic_planet(latent, action, belief, hidden, seed=seed)
    g = torch.Generator().manual_seed(seed + 100)
    latent0, belief0 = torch.randn(1, latent, generator=g) * 0.3, torch.randn(1, belief, generator=g) * 0.3
    actions = torch.rand(pop, H, action, generator=g) * 2 - 1
    me, _ = build_reference_planet_env(pm, latent0, belief0, generator=torch.Generator().manual_seed(11))
    ref = me.evaluate_action_sequences(actions, np.zeros((3, 16, 16), np.float32), P)
    gen2 = torch.Generator().manual_seed(11)
    eps = torch.stack([torch.randn(pop * P, latent, generator=gen2) for _ in range(H)])
    trace = {}
    mine = pl.planet_rollout(pm, actions, latent0, belief0, P, eps=eps, trace=trace)
    assert torch.equal(ref, mine), f"{name}: oracle != reference"
    d = {k: getattr(pm, k).numpy() for k in pl.PLANET_TENSORS}
    d.update(x_actions=actions.numpy(), x_latent0=latent0.numpy(), x_belief0=belief0.numpy(), x_eps=eps.numpy(), x_returns=ref.numpy(),
             x_latent_step0=trace["latent"][0].numpy(), x_belief_step0=trace["belief"][0].numpy(),
             x_rewards_step0=trace["rewards"][0].numpy())
    meta = dict(kind="planet", pop=pop, P=P, H=H, min_std=pm.min_std)
    d["meta_json"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    np.savez(os.path.join(OUT, f"planet_{name}.npz"), **d)
    print(f"planet_{name}: B={pop * P} returns[{ref.min():.3f},{ref.max():.3f}]  oracle==reference bitwise")


def gen_agent_act(name, obs, act, mkw, pop, P, H, iters, n_steps=2, optimizer="cem"):
    """The whole seam: the UNMODIFIED reference TrajectoryOptimizerAgent (CEM, truncated normal) planning through the
    reference ModelEnv on CPU under fixed seeds (torch.manual_seed for the sampler + randperms, a generator for eps);
    records the actions / plans of consecutive act() calls (the second uses the shifted warm start)."""
    mbrl = import_reference()
    import omegaconf

    om = po.make_synthetic_model(obs, act, **mkw)
    gen = torch.Generator().manual_seed(21)
    me, _, _ = build_reference_model_env(om, obs, act, generator=gen)
    if optimizer == "icem":  # conf/action_optimizer/icem.yaml with overrides/pets_icem_cartpole.yaml:16-23
        cfg = omegaconf.OmegaConf.create(dict(_target_="mbrl.planning.ICEMOptimizer", num_iterations=iters, elite_ratio=0.1,
                                              population_size=pop, population_decay_factor=1.3, colored_noise_exponent=2.0,
                                              keep_elite_frac=0.3, alpha=0.1, device="cpu", lower_bound="???", upper_bound="???",
                                              return_mean_elites=True, population_size_module=5))
    elif optimizer == "mppi":  # conf/action_optimizer/mppi.yaml with overrides/pets_mppi_halfcheetah.yaml:19-24
        cfg = omegaconf.OmegaConf.create(dict(_target_="mbrl.planning.MPPIOptimizer", num_iterations=iters, population_size=pop, gamma=0.9,
                                              sigma=1.0, beta=0.9, device="cpu", lower_bound="???", upper_bound="???"))
    else:
        cfg = omegaconf.OmegaConf.create(dict(_target_="mbrl.planning.CEMOptimizer", num_iterations=iters, elite_ratio=0.1,
                                              population_size=pop, alpha=0.1, device="cpu", lower_bound="???", upper_bound="???",
                                              return_mean_elites=True, clipped_normal=False))
    agent = mbrl.planning.TrajectoryOptimizerAgent(cfg, [-1.0] * act, [1.0] * act, planning_horizon=H, replan_freq=1)
    agent.set_trajectory_eval_fn(lambda s, a: me.evaluate_action_sequences(a, initial_state=s, num_particles=P))
    rng = np.random.default_rng(8)
    observations = (rng.standard_normal((n_steps, obs)) * 0.1).astype(np.float32)
    torch.manual_seed(4321)
    actions, plans = [], []
    for t in range(n_steps):
        a = agent.act(observations[t])
        actions.append(np.asarray(a, np.float32))
        plans.append(agent.optimizer.previous_solution.numpy().copy())  # shifted plan kept for the next call
    arrays = dict(observations=observations, actions=np.stack(actions), shifted_plans=np.stack(plans))
    save_case(os.path.join(OUT, f"agent_{name}.npz"), om, dict(kind="agent", obs_dim=obs, act_dim=act, pop=pop, P=P, H=H, iters=iters,
                                                              torch_seed=4321, generator_seed=21, optimizer=optimizer), arrays)
    print(f"agent_{name}: actions {np.stack(actions).round(4).tolist()}")


FULL_CASES = {
    # BASELINE.json configs at their FULL sizes (SURVEY.md section 8 shorthand).  Only seeds and outputs are stored: the
    # model is rebuilt from its seed by po.make_synthetic_model (a checksum of the weights guards that), the draws are the
    # reference's own under torch.manual_seed / the ModelEnv generator.
    "cfg2_cem": dict(obs=17, act=6, mkw=dict(ensemble_size=5, hid=200, seed=30), pop=500, P=20, H=30, iters=5, optimizer="cem"),
    "cfg4_icem": dict(obs=45, act=17, mkw=dict(ensemble_size=7, hid=200, seed=31, elite=[0, 1, 2, 3, 4], termination="humanoid"),
                      pop=1000, P=20, H=40, iters=5, optimizer="icem", module=7),
    "cfg5_mppi": dict(obs=17, act=6, mkw=dict(ensemble_size=5, hid=200, seed=32), pop=2000, P=20, H=50, iters=5, optimizer="mppi"),
    # BASELINE.json configs[3] taken literally: Gymnasium Humanoid-v4 (obs 376, 752 output columns) instead of the truncated-obs variant
    "cfg4p_icem": dict(obs=376, act=17, mkw=dict(ensemble_size=7, hid=200, seed=33, elite=[0, 1, 2, 3, 4], termination="humanoid"),
                       pop=1000, P=20, H=40, iters=5, optimizer="icem", module=7),
    # The workloads the reference SHIPS (what `python -m mbrl.examples.main algorithm=pets overrides=...` plans with):
    # conf/overrides/pets_halfcheetah.yaml:1-24 + conf/dynamics_model/gaussian_mlp_ensemble.yaml:4-13 + conf/algorithm/pets.yaml:20
    # + util/env.py:79-82 + env/pets_halfcheetah.py:91-113: obs 18 through HalfCheetahEnv.preprocess_fn (18 model inputs + 6 actions),
    # no_delta_list [0], 7 members of which 5 elites (a non-trivial subset), pop 400, elite ratio 0.16, alpha 0.12, P 20, H 30
    "stock_halfcheetah": dict(obs=18, act=6, mkw=dict(ensemble_size=7, hid=200, seed=34, elite=[0, 2, 3, 5, 6], obs_process="halfcheetah",
                                                      no_delta_list=[0]),
                              pop=400, P=20, H=30, iters=5, optimizer="cem", elite_ratio=0.16, alpha=0.12),
    # conf/overrides/pets_cartpole.yaml:1-21 + util/env.py:71-74: cartpole_continuous (obs 4, act 1) with its reward / termination
    # functions, 7 members / 5 elites, pop 350, elite ratio 0.1, alpha 0.1, P 20, H 15
    "stock_cartpole": dict(obs=4, act=1, mkw=dict(ensemble_size=7, hid=200, seed=35, elite=[1, 2, 4, 5, 6], reward="cartpole",
                                                  termination="cartpole"),
                           pop=350, P=20, H=15, iters=5, optimizer="cem", elite_ratio=0.1, alpha=0.1),
    # conf/overrides/pets_pusher.yaml:1-20: obs 20 / act 7, LEARNED reward (the model's last output column, one_dim_tr_model.py:287),
    # no termination function, pop 350, elite ratio 0.1, alpha 0.1, P 20, H 25
    "stock_pusher": dict(obs=20, act=7, mkw=dict(ensemble_size=7, hid=200, seed=36, elite=[0, 1, 3, 4, 6], learned_rewards=True, reward=None),
                         pop=350, P=20, H=25, iters=5, optimizer="cem", elite_ratio=0.1, alpha=0.1),
    # conf/overrides/pets_mppi_halfcheetah.yaml:1-24: the pets_halfcheetah model (obs 18 through preprocess_fn, no_delta_list [0]) with
    # a learned reward, planned by MPPI: pop 350, gamma 0.9, sigma 1.0, beta 0.9, P 20, H 30
    "stock_mppi_halfcheetah": dict(obs=18, act=6, mkw=dict(ensemble_size=7, hid=200, seed=37, elite=[0, 2, 3, 5, 6], obs_process="halfcheetah",
                                                           no_delta_list=[0], learned_rewards=True, reward=None),
                                   pop=350, P=20, H=30, iters=5, optimizer="mppi"),
}


def weights_checksum(om) -> list:
    return [float(w.double().abs().sum()) for w in om.weights]


def full_case_agent_cfg(c, target_prefix, device, **extra):
    """Optimizer config of a FULL_CASES entry (conf/action_optimizer/{cem,icem,mppi}.yaml with the PETS overrides)."""
    if c["optimizer"] == "icem":
        cfg = dict(_target_=f"{target_prefix}.ICEMOptimizer", num_iterations=c["iters"], elite_ratio=0.1, population_size=c["pop"],
                   population_decay_factor=1.3, colored_noise_exponent=2.0, keep_elite_frac=0.3, alpha=0.1, device=device,
                   lower_bound="???", upper_bound="???", return_mean_elites=True, population_size_module=c.get("module"))
    elif c["optimizer"] == "mppi":
        cfg = dict(_target_=f"{target_prefix}.MPPIOptimizer", num_iterations=c["iters"], population_size=c["pop"], gamma=0.9,
                   sigma=1.0, beta=0.9, device=device, lower_bound="???", upper_bound="???")
    else:
        cfg = dict(_target_=f"{target_prefix}.CEMOptimizer", num_iterations=c["iters"], elite_ratio=c.get("elite_ratio", 0.1),
                   population_size=c["pop"], alpha=c.get("alpha", 0.1), device=device, lower_bound="???", upper_bound="???", return_mean_elites=True, clipped_normal=False)
    cfg.update(extra)
    return cfg


def gen_agent_full(name, n_steps=2):
    """A FULL_CASES config through the UNMODIFIED reference agent + ModelEnv on CPU under fixed seeds: consecutive act() calls
    (the second call uses the shifted warm start and, for iCEM / MPPI, the persistent elites / mean)."""
    mbrl = import_reference()
    import omegaconf

    c = FULL_CASES[name]
    obs, act, P, H = c["obs"], c["act"], c["P"], c["H"]
    om = po.make_synthetic_model(obs, act, **c["mkw"])
    gen = torch.Generator().manual_seed(21)
    me, _, _ = build_reference_model_env(om, obs, act, generator=gen)
    cfg = omegaconf.OmegaConf.create(full_case_agent_cfg(c, "mbrl.planning", "cpu"))
    agent = mbrl.planning.TrajectoryOptimizerAgent(cfg, [-1.0] * act, [1.0] * act, planning_horizon=H, replan_freq=1)
    agent.set_trajectory_eval_fn(lambda s, a: me.evaluate_action_sequences(a, initial_state=s, num_particles=P))
    observations = (np.random.default_rng(8).standard_normal((n_steps, obs)) * 0.1).astype(np.float32)
    if om.termination == "humanoid":
        observations[:, 0] = 1.4  # inside the healthy z range (termination_fns.py:88-95) so that plans matter
    torch.manual_seed(4321)
    actions, plans = [], []
    for t in range(n_steps):
        actions.append(np.asarray(agent.act(observations[t]), np.float32))
        plans.append(agent.optimizer.previous_solution.numpy().copy())
    meta = dict(kind="agent_full", name=name, torch_seed=4321, generator_seed=21, weights_checksum=weights_checksum(om))
    _save_npz(f"agentfull_{name}.npz", meta, dict(observations=observations, actions=np.stack(actions), shifted_plans=np.stack(plans)))
    print(f"agentfull_{name}: actions[0][:3]={np.stack(actions)[0][:3].round(4).tolist()}")


def main():
    os.makedirs(OUT, exist_ok=True)
    if "--full-only" in sys.argv:
        torch.set_num_threads(8)
        for name in FULL_CASES:
            gen_agent_full(name)
        return
    if "--full" in sys.argv:  # one FULL_CASES entry by name (the others' fixtures stay as committed)
        torch.set_num_threads(8)
        gen_agent_full(sys.argv[sys.argv.index("--full") + 1])
        return
    torch.set_num_threads(4)
    for name, (obs, act, mkw, pop, P, H) in ROLLOUT_CASES.items():
        gen_rollout(name, obs, act, mkw, pop, P, H)
    gen_cem("truncated_mean", clipped=False, return_mean=True)
    gen_cem("truncated_best", clipped=False, return_mean=False)
    gen_cem("clipped_mean", clipped=True, return_mean=True)
    gen_mppi()
    gen_icem()
    gen_agent_act("cem_two_steps", 17, 6, dict(ensemble_size=5, hid=48, seed=20, no_delta_list=[0]), pop=60, P=5, H=8, iters=4)
    gen_agent_act("mppi_two_steps", 17, 6, dict(ensemble_size=5, hid=48, seed=22), pop=50, P=5, H=7, iters=3, optimizer="mppi")
    gen_agent_act("icem_two_steps", 17, 6, dict(ensemble_size=5, hid=48, seed=23), pop=60, P=5, H=8, iters=3, optimizer="icem")
    gen_planet("cheetah_shape", 30, 6, 200, 200, pop=40, P=1, H=12, seed=1)   # conf/dynamics_model/planet.yaml sizes
    gen_planet("small_particles", 10, 3, 40, 24, pop=11, P=3, H=5, seed=2)
    for name in FULL_CASES:
        gen_agent_full(name)


if __name__ == "__main__":
    main()
