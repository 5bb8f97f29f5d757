"""Closed loop, the reference's own end-to-end criterion (tests/algorithms/test_algorithms.py:29,44-68,131): a point mass
starts at 1.0 and must return to 0.0; PETS (ensemble model + CEM planner) must reach an episode return > -20 * 0.001
within a few trials.  The model is trained here with a minimal torch loop (training is out of scope for the engine and is
test infrastructure only); planning runs through hipets (agent -> CEMOptimizer -> rollouts on the GPU)."""
import numpy as np
import pytest
import torch

import hipets

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TRIAL_LEN, REW_C = 30, 0.001
TARGET = -20 * REW_C


class MockLineEnv:  # tests/algorithms/test_algorithms.py:44-68
    def reset(self):
        self.pos, self.vel, self.left = 1.0, 0.0, TRIAL_LEN
        return np.array([self.pos, self.vel])

    def step(self, action):
        self.vel += float(np.asarray(action).item())
        self.pos += self.vel
        self.left -= 1
        return np.array([self.pos, self.vel]), -REW_C * self.pos**2, self.left == 0


def mock_reward_fn(action, obs):  # :71-72, an arbitrary Python callable (not one of mbrl.env's closed forms)
    return -REW_C * (obs[:, 0] ** 2).unsqueeze(1)


def train_ensemble(obs, act, nxt, E=5, hid=64, steps=1500, seed=0):
    """Gaussian NLL training of an E-member MLP ensemble on (obs, act) -> delta obs; returns ModelSpec ingredients."""
    g = torch.Generator().manual_seed(seed)
    x = torch.cat([obs, act], 1)
    y = nxt - obs
    mean, std = x.mean(0, keepdim=True), x.std(0, keepdim=True).clamp_min(1e-5)
    dims = [x.shape[1], hid, hid, 2 * y.shape[1]]
    ws = [torch.nn.Parameter(torch.randn(E, dims[i], dims[i + 1], generator=g) / (2 * np.sqrt(dims[i]))) for i in range(3)]
    bs = [torch.nn.Parameter(torch.zeros(E, 1, dims[i + 1])) for i in range(3)]
    min_lv, max_lv = -10 * torch.ones(1, y.shape[1]), 0.5 * torch.ones(1, y.shape[1])
    opt = torch.optim.Adam(ws + bs, lr=3e-3)
    xn = ((x - mean) / std).unsqueeze(0).expand(E, -1, -1)
    for it in range(steps):
        idx = torch.randint(0, x.shape[0], (E, 128), generator=g)
        xb = torch.gather(xn, 1, idx.unsqueeze(-1).expand(-1, -1, x.shape[1]))
        yb = torch.gather(y.unsqueeze(0).expand(E, -1, -1), 1, idx.unsqueeze(-1).expand(-1, -1, y.shape[1]))
        h = xb
        for li in range(3):
            h = h.matmul(ws[li]) + bs[li]
            if li < 2:
                h = torch.nn.functional.silu(h)
        mu, lv = h[..., : y.shape[1]], h[..., y.shape[1]:]
        lv = max_lv - torch.nn.functional.softplus(max_lv - lv)
        lv = min_lv + torch.nn.functional.softplus(lv - min_lv)
        loss = (((mu - yb) ** 2) * (-lv).exp() + lv).mean()
        opt.zero_grad()
        loss.backward()
        opt.step()
    return [w.detach() for w in ws], [b.detach() for b in bs], min_lv, max_lv, mean.double(), std.double()


def collect_random(env, n, rng):
    o, a, nx = [], [], []
    obs = env.reset()
    for _ in range(n):
        act = rng.uniform(-1, 1, size=(1,))
        nobs, _, done = env.step(act)
        o.append(obs); a.append(act); nx.append(nobs)
        obs = env.reset() if done else nobs
    f = lambda v: torch.tensor(np.array(v), dtype=torch.float32)  # noqa: E731
    return f(o), f(a), f(nx)


@pytest.mark.parametrize("objective", ["unfused_python_reward", "unfused_basic_ensemble_tsinf", "fused_enum_reward"])
def test_pets_solves_the_point_mass_task(engine, objective):
    torch.manual_seed(12345)
    rng = np.random.default_rng(12345)
    env = MockLineEnv()
    o, a, nx = collect_random(env, 600, rng)
    ws, bs, mn, mx, nmean, nstd = train_ensemble(o, a, nx)
    kw = dict(weights=ws, biases=bs, obs_dim=2, act_dim=1, min_logvar=mn, max_logvar=mx, norm_mean=nmean, norm_std=nstd,
              activation="silu", propagation="random_model")
    if objective == "unfused_basic_ensemble_tsinf":
        # the reference's second model config (tests/algorithms/test_algorithms.py:197-198, conf/dynamics_model/
        # basic_ensemble.yaml): BasicEnsemble semantics (iid member draws, any batch size) with TS-infinity propagation
        kw.update(ensemble_kind="basic_ensemble", propagation="fixed_model")
        spec = hipets.ModelSpec(reward="none", termination="no_termination", **kw)
        fn = hipets.UnfusedTrajectoryEvalFn(spec, 7, reward_fn=mock_reward_fn, engine=engine, seed=1)  # 500 * 7 % 5 == 0 not needed
    elif objective == "unfused_python_reward":
        spec = hipets.ModelSpec(reward="none", termination="no_termination", **kw)
        fn = hipets.UnfusedTrajectoryEvalFn(spec, 20, reward_fn=mock_reward_fn, engine=engine, seed=1)
    else:
        # The task's reward is not one of mbrl.env's closed forms, so the fully fused kernel cannot score it.  Check the
        # fused path on the SAME trained model instead: with a halfcheetah-form reward (obs[0] - 0.1 |a|^2) the fused
        # objective must rank random plans like the unfused objective given the same formula as a Python callable.
        spec_u = hipets.ModelSpec(reward="none", termination="no_termination", **kw)
        fn_u = hipets.UnfusedTrajectoryEvalFn(spec_u, 20, reward_fn=lambda act, ob: ob[:, :1] - 0.1 * act.square().sum(1, keepdim=True),
                                              engine=engine, seed=1)
        padded = dict(kw)
        # halfcheetah reward reads obs[0] and obs[2]: give the model a third, constant observation dimension
        padded["weights"] = [torch.cat([ws[0][:, :2], torch.zeros(5, 1, ws[0].shape[2]), ws[0][:, 2:]], 1), ws[1],
                             torch.cat([ws[2][:, :, :2], torch.zeros(5, ws[2].shape[1], 1), ws[2][:, :, 2:4],
                                        torch.zeros(5, ws[2].shape[1], 1)], 2)]
        padded["biases"] = [bs[0], bs[1], torch.cat([bs[2][:, :, :2], torch.zeros(5, 1, 1), bs[2][:, :, 2:4], torch.full((5, 1, 1), -20.0)], 2)]
        padded.update(obs_dim=3, min_logvar=torch.cat([mn, -10 * torch.ones(1, 1)], 1), max_logvar=torch.cat([mx, -9 * torch.ones(1, 1)], 1),
                      norm_mean=torch.cat([nmean[:, :2], torch.zeros(1, 1, dtype=torch.float64), nmean[:, 2:]], 1),
                      norm_std=torch.cat([nstd[:, :2], torch.ones(1, 1, dtype=torch.float64), nstd[:, 2:]], 1))
        spec_f = hipets.ModelSpec(reward="halfcheetah", termination="no_termination", **padded)
        fn_f = hipets.make_eval_fn(spec_f, 20, engine=engine, seed=1)
        g = torch.Generator().manual_seed(0)
        plans = (torch.rand(64, 10, 1, generator=g) * 2 - 1).to(DEV)
        vu = fn_u(np.array([1.0, 0.0], np.float32), plans).cpu()
        vf = fn_f(np.array([1.0, 0.0, 0.0], np.float32), plans).cpu()
        assert torch.corrcoef(torch.stack([vu, vf]))[0, 1] > 0.97  # same model, same ranking of plans
        return
    cfg = dict(_target_="hipets.CEMOptimizer", num_iterations=5, elite_ratio=0.1, population_size=500, alpha=0.1, device=DEV,
               lower_bound="???", upper_bound="???", return_mean_elites=True, seed=3)
    agent = hipets.TrajectoryOptimizerAgent(cfg, [-1.0], [1.0], planning_horizon=15)
    agent.set_trajectory_eval_fn(fn)
    best = -np.inf
    for trial in range(3):
        obs = env.reset()
        agent.reset()
        total, done = 0.0, False
        while not done:
            obs, r, done = env.step(agent.act(obs))
            total += r
        best = max(best, total)
        if best > TARGET:
            break
    assert best > TARGET, best
