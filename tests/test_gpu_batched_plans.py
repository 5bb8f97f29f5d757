"""Batched planning for MPPI and iCEM (hipets_plan_mppi_batched / hipets_plan_icem_batched): n_env
environments per set of launches.  n_env = 1 is bit-identical to the single-environment fused plan; a batch is replayed
PER ENVIRONMENT through the oracle with the engine's exported draws (teacher-forced per iteration for iCEM), in both in-kernel
randomness modes: 'device' (the default of the Python layer: one balanced permutation per step over the rows of ALL environments) and
'fast'."""
import numpy as np
import pytest
import torch

import hipets
from conftest import to_spec
from hipets.planning import _BoundObjective
from oracle import pets_oracle as po
from oracle import device_draws
from test_gpu_plans_full_size import check_values, elites_agree

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def batched_members(engine, om, pop, P, H, seed, stream, mode):
    """[H, pop * P] member slot of every row of a batched launch, from the engine's exported draws of (seed, stream)."""
    B = pop * P
    if mode == "device":  # slot j holds row perms[t][j] and runs member j // (B / M) (gaussian_mlp.py:164-166, 203-205), over ALL environments' rows
        perms = engine.device_perms(H, B, seed, stream).cpu()
        M = len(om.active_members)
        members = torch.empty(H, B, dtype=torch.long)
        for t in range(H):
            members[t][perms[t]] = torch.arange(B) // (B // M)
        return members
    nwg, r = engine.fast_geometry(pop, P, H)
    sched = engine.fast_schedule(H, nwg, seed, stream).cpu()
    wg = device_draws.fast_row_workgroup(torch.arange(B), P, r)
    return torch.stack([sched[t][wg].long() for t in range(H)])


def batched_replay(engine, om, s0, P, H, seed, mode):
    """values of ALL environments' candidates [n_env * rows] for (population_all, stream): every environment's slice goes through
    the oracle with that slice of the launch's row -> member map and eps."""
    def f(population_all, stream):
        n_env = s0.shape[0]
        pop = population_all.shape[0]
        rows_env = pop // n_env
        eps = engine.fast_normals(H, pop * P, seed, stream).cpu()
        members = batched_members(engine, om, pop, P, H, seed, stream, mode)
        out = []
        for e_ in range(n_env):
            sl = slice(e_ * rows_env * P, (e_ + 1) * rows_env * P)
            out.append(po.rollout(om, population_all[e_ * rows_env:(e_ + 1) * rows_env], s0[e_], P, members=members[:, sl], eps=eps[:, sl]))
        return torch.cat(out)

    return f


@pytest.mark.parametrize("mode", ["device", "fast"])
def test_batched_mppi_plans(engine, mode):
    obs, act, H, P, pop, n_env, iters = 17, 6, 9, 5, 120, 3, 3
    om = po.make_synthetic_model(obs, act, ensemble_size=5, hid=48, seed=6)
    fn = hipets.make_eval_fn(to_spec(om, obs, act), P, engine=engine, seed=3, mode=mode)
    lb, ub = [-1.0] * act, [1.0] * act
    s0 = (np.random.default_rng(1).standard_normal((n_env, obs)) * 0.3).astype(np.float32)
    # n_env = 1 == the single-environment fused plan, bit for bit, over two consecutive (shifted) plans
    single = hipets.MPPIOptimizer(iters, pop, 0.9, 1.0, 0.9, [lb] * H, [ub] * H, DEV, seed=7)
    agent1 = hipets.BatchedMPPIAgent(fn, 1, lb, ub, H, iters, pop, 0.9, 1.0, 0.9, seed=7)
    for _ in range(2):
        one = single.optimize(_BoundObjective(fn, s0[0]))
        assert torch.equal(torch.from_numpy(agent1.plan(s0[:1]))[0], one.cpu())
    # a batch, replayed per environment (free-running: MPPI has no discrete selection)
    agent = hipets.BatchedMPPIAgent(fn, n_env, lb, ub, H, iters, pop, 0.9, 1.0, 0.9, seed=7)
    lower, upper = -torch.ones(H, act), torch.ones(H, act)
    states = [po.MPPIState(H, act) for _ in range(n_env)]
    one_t, zero_t = torch.ones(H, act, device=DEV), torch.zeros(H, act, device=DEV)
    for call in range(2):
        tr = engine.set_plan_trace(iters, n_env * pop, H, act, 1, n_env=n_env)
        plans = agent.plan(s0)
        torch.cuda.synchronize()
        engine.set_plan_trace(0)
        seed, plan_id = agent.seed ^ fn.seed, agent.calls
        roll = batched_replay(engine, om, s0, P, H, seed, mode)
        z = []
        for k in range(iters):
            buf = torch.empty(n_env * pop, H, act, device=DEV)
            engine.mppi_sample(n_env * pop, H, act, 1.0, zero_t, torch.zeros(act, device=DEV), -1e3 * one_t, 1e3 * one_t, buf, seed=seed,
                               stream_id=plan_id * iters + k)
            z.append(buf.cpu())
        # the batch's values per iteration, from the engine's own populations (what each environment's objective returns)
        vals = [roll(tr["populations"][k].cpu(), plan_id * iters + k) for k in range(iters)]
        for k in range(iters):
            check_values(tr["values"][k].cpu(), vals[k])
        for e_ in range(n_env):
            it = {"i": 0}

            def obj(population, e_=e_):
                k = it["i"]
                it["i"] += 1
                assert torch.allclose(population, tr["populations"][k].cpu()[e_ * pop:(e_ + 1) * pop], rtol=0, atol=2e-5)
                return vals[k][e_ * pop:(e_ + 1) * pop]

            ref = po.mppi_optimize(obj, states[e_], lower, upper, iters, pop, 0.9, 1.0, 0.9, noise=[zk[e_ * pop:(e_ + 1) * pop] for zk in z])
            assert np.allclose(plans[e_], ref.numpy(), rtol=0, atol=1e-4), (call, e_)
    assert plans.shape == (n_env, H, act) and agent.act(s0).shape == (n_env, act)


@pytest.mark.parametrize("mode", ["device", "fast"])
def test_batched_icem_plans(engine, mode):
    obs, act, H, P, pop, n_env, iters, module = 17, 6, 8, 5, 150, 3, 4, 5
    om = po.make_synthetic_model(obs, act, ensemble_size=7, hid=48, seed=9, elite=[0, 1, 2, 3, 4])
    fn = hipets.make_eval_fn(to_spec(om, obs, act), P, engine=engine, seed=3, mode=mode)
    lb, ub = [-1.0] * act, [1.0] * act
    s0 = (np.random.default_rng(2).standard_normal((n_env, obs)) * 0.3).astype(np.float32)
    kw = dict(num_iterations=iters, elite_ratio=0.1, population_size=pop, population_decay_factor=1.3, colored_noise_exponent=2.0,
              keep_elite_frac=0.3, alpha=0.1, return_mean_elites=True, population_size_module=module)
    K, keep, sizes = po.icem_sizes(iters, 0.1, pop, 1.3, 0.3, module)
    g = torch.Generator().manual_seed(0)
    # n_env = 1 == the single-environment fused plan with the same injected kept-elite draws, over two plans
    single = hipets.ICEMOptimizer(lower_bound=[lb] * H, upper_bound=[ub] * H, device=DEV, seed=7, **kw)
    agent1 = hipets.BatchedICEMAgent(fn, 1, lb, ub, H, seed=7, **kw)
    x0 = torch.zeros(H, act)
    for _ in range(2):
        kidx = torch.stack([torch.randperm(K, generator=g)[:keep] for _ in range(iters)]).to(torch.int32).to(DEV).contiguous()
        one = single.optimize(_BoundObjective(fn, s0[0]), x0=x0, keep_idx=kidx)
        got = torch.from_numpy(agent1.plan(s0[:1], keep_idx=kidx.view(iters, 1, keep).contiguous()))[0]
        assert torch.equal(got, one.cpu())
        x0 = one.cpu().roll(-1, dims=0)
        x0[-1] = 0.0
    # a batch, replayed per environment with teacher forcing
    agent = hipets.BatchedICEMAgent(fn, n_env, lb, ub, H, seed=7, **kw)
    lower, upper = -torch.ones(H, act), torch.ones(H, act)
    states = [po.ICEMState() for _ in range(n_env)]
    one_t, zero_t = torch.ones(H, act, device=DEV), torch.zeros(H, act, device=DEV)
    for call in range(2):
        perms = [[torch.randperm(K, generator=g) for _ in range(n_env)] for _ in range(iters)]
        kidx = torch.stack([torch.stack([pe[:keep] for pe in perms[i]]) for i in range(iters)]).to(torch.int32).to(DEV).contiguous()
        had_elite = agent.has_elite
        x0_all = agent.previous_solution.cpu().clone()
        max_rows = n_env * (sizes[0] + keep)
        tr = engine.set_plan_trace(iters, max_rows, H, act, K, n_env=n_env)
        plans = agent.plan(s0, keep_idx=kidx)
        torch.cuda.synchronize()
        engine.set_plan_trace(0)
        seed, plan_id = agent.seed ^ fn.seed, agent.calls
        roll = batched_replay(engine, om, s0, P, H, seed, mode)
        rows_i, noise_i, tail_i = [], [], None
        for i in range(iters):
            sid = (plan_id * iters + i) * 4
            extra = 0
            if had_elite or i > 0:
                extra = 1 if (i == iters - 1 and i != 0) else keep
            rows_i.append(sizes[i] + extra)
            buf = torch.empty(n_env * sizes[i], H, act, device=DEV)
            engine.icem_sample(n_env * sizes[i], H, act, 2.0, zero_t, one_t, -1e3 * one_t, 1e3 * one_t, buf, seed=seed, stream_id=sid)
            noise_i.append(buf.cpu())
            if i == 0 and had_elite:
                sh = torch.empty(n_env * keep, H, act, device=DEV)
                engine.icem_shift(n_env * keep, H, act, torch.zeros(n_env * keep, H, act, device=DEV), zero_t, one_t, sh, seed=seed, stream_id=sid + 1)
                tail_i = sh[:, H - 1, :].cpu()
        vals = [roll(tr["populations"][i][: n_env * rows_i[i]].cpu(), (plan_id * iters + i) * 4 + 3) for i in range(iters)]
        for i in range(iters):
            check_values(tr["values"][i][: n_env * rows_i[i]].cpu(), vals[i])
        for e_ in range(n_env):
            inject, teacher = [], []
            for i in range(iters):
                n, r = sizes[i], rows_i[i]
                inj = {"noise": noise_i[i][e_ * n:(e_ + 1) * n], "keep_perm": perms[i][e_]}
                if i == 0 and had_elite:
                    inj["end_noise"] = tail_i[e_ * keep:(e_ + 1) * keep]
                inject.append(inj)
                pop_e = tr["populations"][i].cpu()[e_ * r:(e_ + 1) * r]
                teacher.append((tr["mus"][i][e_].cpu(), tr["dispersions"][i][e_].cpu(), pop_e[tr["elite_idx"][i][e_].cpu().long()]))
            it = {"i": 0}

            def obj(population, e_=e_):
                i = it["i"]
                it["i"] += 1
                r = rows_i[i]
                assert torch.allclose(population, tr["populations"][i].cpu()[e_ * r:(e_ + 1) * r], rtol=0, atol=1e-5), (call, e_, i)
                return vals[i][e_ * r:(e_ + 1) * r]

            rec = []
            po.icem_optimize(obj, states[e_], x0_all[e_], lower, upper, iters, 0.1, pop, 1.3, 2.0, 0.3, 0.1, return_mean_elites=True,
                             population_size_module=module, inject=inject, record=rec, teacher=teacher)
            states[e_].elite = agent.elite[e_].cpu()
            for i in range(iters):
                if elites_agree(tr["elite_idx"][i][e_].cpu(), rec[i]["values"], K):
                    assert torch.allclose(tr["mus"][i][e_].cpu(), rec[i]["mu"], rtol=0, atol=1e-4), (call, e_, i)
                    assert torch.allclose(tr["dispersions"][i][e_].cpu(), rec[i]["var"], rtol=1e-4, atol=1e-5), (call, e_, i)
            assert np.array_equal(plans[e_], tr["mus"][iters - 1][e_].cpu().numpy())
    assert agent.act(s0).shape == (n_env, act)
