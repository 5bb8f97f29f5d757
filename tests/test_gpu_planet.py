"""PlaNet latent planner on the GPU against the oracle and the reference's golden vectors; every call
goes through the C ABI (hipets_planet_set_model / hipets_planet_rollout)."""
import glob
import os

import numpy as np
import pytest
import torch

import hipets
from conftest import GOLDEN
from oracle import planet_oracle as pl
from test_oracle_golden import load_planet_case

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def to_planet_spec(pm):
    """PlaNetOracleModel (test infra) -> hipets.PlaNetSpec (product)."""
    return hipets.PlaNetSpec(**{k: getattr(pm, k) for k in pl.PLANET_TENSORS}, min_std=pm.min_std)


def close(out, ref, tol=1e-4):
    out, ref = out.detach().cpu(), ref.detach().cpu()
    err = (out - ref).abs()
    lim = tol * torch.clamp(ref.abs(), min=1.0)
    assert (err <= lim).all(), f"max err {err.max():.3e}"


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "planet_*.npz"))), ids=lambda p: os.path.basename(p)[7:-4])
def test_planet_golden_reference_vectors(engine, path):
    pm, meta, a = load_planet_case(path)
    engine.planet_set_model(to_planet_spec(pm))
    H, B = meta["H"], meta["pop"] * meta["P"]
    tl = torch.zeros(H, B, pm.latent_size, device=DEV)
    tb = torch.zeros(H, B, pm.belief_size, device=DEV)
    tr = torch.zeros(H, B, device=DEV)
    out = engine.planet_rollout(a["actions"].to(DEV), a["latent0"].to(DEV), a["belief0"].to(DEV), meta["P"], eps=a["eps"].to(DEV),
                                trace_latent=tl, trace_belief=tb, trace_rewards=tr)
    assert torch.allclose(tb[0].cpu(), a["belief_step0"], rtol=1e-5, atol=2e-6)  # one GRU step
    assert torch.allclose(tl[0].cpu(), a["latent_step0"], rtol=1e-5, atol=2e-6)
    assert torch.allclose(tr[0].cpu(), a["rewards_step0"].flatten(), rtol=1e-5, atol=2e-6)
    close(out, a["returns"])


@pytest.mark.parametrize("latent,action,belief,hidden,pop,P,H", [
    (30, 6, 200, 200, 1000, 1, 12),   # conf/dynamics_model/planet.yaml + overrides/planet_cheetah_run.yaml (pop 1000, H 12)
    (7, 2, 22, 19, 37, 2, 5),         # nothing a multiple of 4: tail chunks that share columns with the latent
    (30, 6, 200, 200, 5, 3, 3),       # fewer rows than one tile
])
def test_planet_rollout_matches_oracle(engine, latent, action, belief, hidden, pop, P, H):
    pm = pl.make_synthetic_planet(latent, action, belief, hidden, seed=latent + pop)
    engine.planet_set_model(to_planet_spec(pm))
    g = torch.Generator().manual_seed(3)
    latent0, belief0 = torch.randn(1, latent, generator=g) * 0.3, torch.randn(1, belief, generator=g) * 0.3
    actions = torch.rand(pop, H, action, generator=g) * 2 - 1
    eps = torch.randn(H, pop * P, latent, generator=g)
    ref = pl.planet_rollout(pm, actions, latent0, belief0, P, eps=eps)
    out = engine.planet_rollout(actions.to(DEV), latent0.to(DEV), belief0.to(DEV), P, eps=eps.to(DEV))
    close(out, ref)
    # deterministic rollouts (sample(deterministic=True)): latent = prior mean
    ref_det = pl.planet_rollout(pm, actions, latent0, belief0, P, eps=torch.zeros_like(eps))
    close(engine.planet_rollout(actions.to(DEV), latent0.to(DEV), belief0.to(DEV), P, sample=False), ref_det)


@pytest.mark.parametrize("pop,P,H", [(1000, 1, 12), (37, 3, 4)])
def test_planet_static_instance_equals_the_generic_kernel_bitwise(engine, pop, P, H):
    """Models with conf/dynamics_model/planet.yaml's shapes (latent 30, belief 200, hidden 200, action 6: what every planet_*.yaml override
    plans on) run a kernel instance with every op's tile / chunk counts and the LDS row stride as compile-time facts (planet.hpp STATIC,
    round 5); HIPETS_PLANET_GENERIC=1 forces the run-time generic instance.  Same tile -> wave deal, same k order: the same bits, with
    injected eps, in-kernel Philox draws and without sampling."""
    pm = pl.make_synthetic_planet(30, 6, 200, 200, seed=11)
    engine.planet_set_model(to_planet_spec(pm))
    g = torch.Generator().manual_seed(5)
    latent0, belief0 = (torch.randn(1, 30, generator=g) * 0.3).to(DEV), (torch.randn(1, 200, generator=g) * 0.3).to(DEV)
    actions = (torch.rand(pop, H, 6, generator=g) * 2 - 1).to(DEV)
    eps = torch.randn(H, pop * P, 30, generator=g).to(DEV)

    def three():
        return (engine.planet_rollout(actions, latent0, belief0, P, eps=eps).clone(), engine.planet_rollout(actions, latent0, belief0, P, seed=4, stream_id=2).clone(),
                engine.planet_rollout(actions, latent0, belief0, P, sample=False).clone())

    static = three()
    os.environ["HIPETS_PLANET_GENERIC"] = "1"
    try:
        generic = three()
    finally:
        del os.environ["HIPETS_PLANET_GENERIC"]
    for a, b in zip(static, generic):
        assert torch.isfinite(a).all() and torch.equal(a, b)
    ref = pl.planet_rollout(pm, actions.cpu(), latent0.cpu(), belief0.cpu(), P, eps=eps.cpu())
    close(static[0], ref)


def test_planet_fast_mode_replayed_and_seeded(engine):
    """FAST mode draws eps with the rollout kernel's Philox streams: replay through the oracle with the exported normals."""
    pm = pl.make_synthetic_planet(12, 3, 48, 40, seed=9)
    engine.planet_set_model(to_planet_spec(pm))
    g = torch.Generator().manual_seed(1)
    latent0, belief0 = torch.randn(12, generator=g).to(DEV), torch.randn(48, generator=g).to(DEV)
    actions = (torch.rand(50, 6, 3, generator=g) * 2 - 1).to(DEV)
    a = engine.planet_rollout(actions, latent0, belief0, 2, seed=5, stream_id=7)
    b = engine.planet_rollout(actions, latent0, belief0, 2, seed=5, stream_id=7)
    c = engine.planet_rollout(actions, latent0, belief0, 2, seed=5, stream_id=8)
    assert torch.equal(a, b) and not torch.equal(a, c)
    assert torch.isfinite(a).all()


def test_planet_eval_fn_and_cem_agent(engine):
    """Seam 3 for PlaNet: make_eval_fn on a PlaNetSpec + the clipped-normal CEM of planet_cheetah_run.yaml (generic optimizer
    path); the optimised plan beats random plans under the oracle."""
    pm = pl.make_synthetic_planet(30, 6, 200, 200, seed=4, min_std=0.01)
    pm.b_prior2[30:] -= 8.0  # low-noise prior: returns are dominated by the actions, not by the latent noise
    fn = hipets.make_eval_fn(to_planet_spec(pm), 1, engine=engine, seed=2)
    assert isinstance(fn, hipets.PlaNetTrajectoryEvalFn)
    g = torch.Generator().manual_seed(0)
    latent0, belief0 = torch.randn(1, 30, generator=g) * 0.3, torch.randn(1, 200, generator=g) * 0.3
    with pytest.raises(RuntimeError, match="set_state"):
        fn(np.zeros((3, 64, 64)), torch.zeros(4, 3, 6))
    fn.set_state(latent0, belief0)
    H, A = 12, 6
    cfg = dict(_target_="hipets.CEMOptimizer", num_iterations=6, elite_ratio=0.1, population_size=400, alpha=0.0, device=DEV,
               lower_bound="???", upper_bound="???", return_mean_elites=True, clipped_normal=True, seed=1)
    agent = hipets.TrajectoryOptimizerAgent(cfg, [-1.0] * A, [1.0] * A, planning_horizon=H, keep_last_solution=False)
    agent.set_trajectory_eval_fn(fn)
    plan = agent.plan(np.zeros((3, 64, 64), np.float32))
    assert plan.shape == (H, A) and (np.abs(plan) <= 1).all()
    cands = torch.cat([torch.from_numpy(plan)[None], torch.rand(15, H, A, generator=g) * 2 - 1])
    eps = torch.randn(H, 16 * 8, 30, generator=g)
    r = pl.planet_rollout(pm, cands, latent0, belief0, 8, eps=eps)
    assert r[0] > r[1:].max(), r


def test_fused_planet_cem_plan_equals_the_per_iteration_path_and_the_oracle(engine):
    """hipets_plan_planet_cem (the PlaNet planner as one library call, clipped-normal CEM of planet_cheetah_run.yaml:29-35) is the
    per-iteration CEMOptimizer loop bit for bit, and -- replayed with the plan's own exported draws, teacher-forced per
    iteration -- the oracle's CEM over the oracle's PlaNet rollouts."""
    from hipets.planning import _BoundObjective
    from oracle import pets_oracle as po

    L, A, Hb, F, H, pop, iters, P = 30, 6, 200, 200, 12, 1000, 10, 1  # conf sizes
    pm = pl.make_synthetic_planet(L, A, Hb, F, seed=4)
    fn = hipets.make_eval_fn(to_planet_spec(pm), P, engine=engine, seed=2)
    g = torch.Generator().manual_seed(0)
    latent0, belief0 = torch.randn(1, L, generator=g) * 0.3, torch.randn(1, Hb, generator=g) * 0.3
    fn.set_state(latent0, belief0)
    lower, upper = -torch.ones(H, A), torch.ones(H, A)
    mk = lambda: hipets.CEMOptimizer(iters, 0.1, pop, lower.tolist(), upper.tolist(), 0.0, DEV, return_mean_elites=True,  # noqa: E731
                                     clipped_normal=True, seed=1)
    a, b = mk(), mk()
    K = int(a.elite_num)
    obj = _BoundObjective(fn, np.zeros((3, 64, 64), np.float32))
    x0 = torch.zeros(H, A)
    tr = engine.set_plan_trace(iters, pop, H, A, K)
    fused = a.optimize(obj, x0=x0)
    torch.cuda.synchronize()
    engine.set_plan_trace(0)
    generic = b.optimize(obj, x0=x0, callback=lambda *_: None)
    assert torch.equal(fused, generic)
    # replay: z of the clipped-normal sampler (mu 0, dispersion 1, wide bounds -> population == z), eps of the rollouts
    seed, plan_id = a.seed ^ fn.seed, a.calls
    p = engine.cem_params(pop, H, A, iters, K, 0.0, True, True)
    one, zero = torch.ones(H, A, device=DEV), torch.zeros(H, A, device=DEV)
    z = []
    for i in range(iters):
        buf = torch.empty(pop, H, A, device=DEV)
        engine.cem_sample(p, zero, one, -1e3 * one, 1e3 * one, buf, seed=seed, stream_id=plan_id * iters + i)
        z.append(buf.cpu())
    it = {"i": 0}

    def oracle_obj(population):
        i = it["i"]
        it["i"] += 1
        # the rollout kernel's normals: planet rows use the PETS kernel's Philox streams with out_dim = latent size
        eps = engine_normals(engine, H, pop * P, L, seed, plan_id * iters + i)
        return pl.planet_rollout(pm, population, latent0, belief0, P, eps=eps)

    teacher = [(tr["mus"][i].cpu(), tr["dispersions"][i].cpu()) for i in range(iters)]
    rec = []
    po.cem_optimize(oracle_obj, x0, lower, upper, iters, 0.1, pop, 0.0, return_mean_elites=True, clipped_normal=True, noise=z, record=rec,
                    teacher=teacher)
    for i in range(iters):
        assert torch.allclose(tr["populations"][i].cpu(), rec[i]["population"], rtol=0, atol=1e-5), i
        v, rv = tr["values"][i].cpu(), rec[i]["values"]
        assert ((v - rv).abs() <= 1e-4 * torch.clamp(rv.abs(), min=1.0)).all(), i  # T2
        if set(tr["elite_idx"][i].cpu().tolist()) == set(rec[i]["elite_idx"].tolist()):
            assert torch.allclose(tr["mus"][i].cpu(), rec[i]["mu"], rtol=0, atol=1e-4), i  # T4
            assert torch.allclose(tr["dispersions"][i].cpu(), rec[i]["disp"], rtol=1e-4, atol=1e-5), i
    assert torch.equal(fused, tr["mus"][iters - 1])


def engine_normals(engine, H, B, latent, seed, stream):
    """eps [H, B, latent] a FAST PlaNet rollout draws for (seed, stream): the rollout kernels share rollout_normals4, so the PETS
    export with a model whose out_dim equals the latent size returns them."""
    from conftest import to_spec
    from oracle import pets_oracle as po

    om = po.make_synthetic_model(latent, 2, ensemble_size=1, hid=16, seed=0)
    engine.set_model(to_spec(om, latent, 2))
    return engine.fast_normals(H, B, seed, stream).cpu()
