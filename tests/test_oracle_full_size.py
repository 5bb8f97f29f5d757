"""The oracle at BASELINE.json's FULL sizes (cfg2 CEM pop 500 x 20 x H 30; cfg4 iCEM pop 1000, 7 members / 5 elites, H 40 incl. the
"+1 mu row" iteration; cfg5 MPPI pop 2000 x H 50) against goldens recorded from the UNMODIFIED reference
agent + ModelEnv (oracle/make_golden.py gen_agent_full; only seeds and outputs are stored).  Bitwise: the oracle consumes
torch's generators in the reference's order, so two consecutive act() calls return exactly the reference's actions."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import pets_oracle as po
from oracle.make_golden import FULL_CASES, weights_checksum


def load_full(name):
    z = np.load(os.path.join(GOLDEN, f"agentfull_{name}.npz"))
    meta = json.loads(bytes(z["meta_json"]).decode())
    return meta, {k[2:]: z[k] for k in z.files if k.startswith("x_")}


def oracle_agent(name, meta):
    """The reference's agent loop restated with oracle pieces: returns act(obs) -> (action, shifted plan)."""
    c = FULL_CASES[name]
    obs, act, P, H = c["obs"], c["act"], c["P"], c["H"]
    om = po.make_synthetic_model(obs, act, **c["mkw"])
    # f64 sums over 200 k weights: the last digit depends on ATen's thread-count dependent summation order
    assert np.allclose(weights_checksum(om), meta["weights_checksum"], rtol=1e-12, atol=0), "synthetic model differs from the recorded one"
    gen = torch.Generator().manual_seed(meta["generator_seed"])
    st = po.TrajectoryOptimizerState(-np.ones(act), np.ones(act), H)
    mppi, icem = po.MPPIState(H, act), po.ICEMState()

    def act_fn(observation):
        obj = lambda pop_: po.rollout(om, pop_, observation, P, global_rng=True, generator=gen)  # noqa: E731
        if c["optimizer"] == "cem":
            opt = lambda x0: po.cem_optimize(obj, x0, st.lower, st.upper, c["iters"], c.get("elite_ratio", 0.1), c["pop"], c.get("alpha", 0.1),  # noqa: E731
                                             return_mean_elites=True)  # noqa: E731
        elif c["optimizer"] == "mppi":
            opt = lambda x0: po.mppi_optimize(obj, mppi, st.lower, st.upper, c["iters"], c["pop"], 0.9, 1.0, 0.9)  # noqa: E731
        else:
            opt = lambda x0: po.icem_optimize(obj, icem, x0, st.lower, st.upper, c["iters"], 0.1, c["pop"], 1.3, 2.0, 0.3, 0.1,  # noqa: E731
                                              return_mean_elites=True, population_size_module=c.get("module"))
        best = st.step(opt)
        return best[0], st.previous_solution.numpy()

    return act_fn


@pytest.mark.parametrize("name", sorted(FULL_CASES))
def test_oracle_reproduces_the_reference_agent_at_full_size(name):
    meta, a = load_full(name)
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    act_fn = oracle_agent(name, meta)
    torch.manual_seed(meta["torch_seed"])
    for t in range(a["observations"].shape[0]):
        action, shifted = act_fn(a["observations"][t])
        assert np.array_equal(np.asarray(action, np.float32), a["actions"][t]), (name, t)
        assert np.array_equal(shifted, a["shifted_plans"][t]), (name, t)
