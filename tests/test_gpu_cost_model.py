"""tests/cost_model_replica.py restates the library's row-tile rule: hold it to the library (hipets_kernel_class) on the workloads of
the committed sweeps, so that the CPU test of the calibration (tests/test_cost_model.py) speaks about the code that ships."""
import pytest

import cost_model_replica as cm
from conftest import to_spec
from oracle import pets_oracle as po

pytestmark = pytest.mark.gpu

MODELS = {
    "cfg2": (17, 6, dict(ensemble_size=5)),
    "halfcheetah": (18, 6, dict(ensemble_size=7, elite=[0, 2, 3, 5, 6], obs_process="halfcheetah", no_delta_list=[0])),
    "cfg4": (45, 17, dict(ensemble_size=7, elite=[0, 1, 2, 3, 4], termination="humanoid")),
    "cartpole": (4, 1, dict(ensemble_size=7, elite=[1, 2, 4, 5, 6], reward="cartpole", termination="cartpole")),
    "pusher": (20, 7, dict(ensemble_size=7, elite=[0, 1, 3, 4, 6], learned_rewards=True, reward=None)),
    "inv_pendulum": (4, 1, dict(ensemble_size=7, elite=[0, 2, 3, 5, 6], learned_rewards=True, reward=None, termination="inverted_pendulum")),
}
CASES = [("cfg2", 500, 30, {1, 2, 3}), ("cfg2", 1000, 30, {1, 2, 3}), ("cfg2", 2000, 50, {1, 2, 3}), ("halfcheetah", 400, 30, {1, 2, 3}),
         ("halfcheetah", 800, 30, {1, 2, 3}), ("cfg4", 1036, 40, {2, 3, 4}), ("cartpole", 350, 15, {1, 2}), ("pusher", 350, 25, {1, 2}),
         ("inv_pendulum", 480, 45, {3})]


@pytest.mark.parametrize("mode", ["fast", "device"])
@pytest.mark.parametrize("case", CASES, ids=lambda c: f"{c[0]}_pop{c[1]}")
def test_replica_equals_the_library(engine, case, mode):
    name, pop, H, lean = case
    obs, act, mkw = MODELS[name]
    om = po.make_synthetic_model(obs, act, hid=200, seed=0, **mkw)
    engine.set_model(to_spec(om, obs, act))
    cls, r = engine.kernel_class(pop, 20, H, mode)
    assert r == cm.choose_r(pop, 20, 5, mode, lean)
    assert cls == ("fused" if r in lean else "hidden_static")


@pytest.mark.parametrize("mode", ["fast", "device"])
@pytest.mark.parametrize("pop", [1036, 805, 630, 497, 358])  # the cfg4' iCEM plan's five population sizes (profiles/r5_cfg4p_iterations.json)
def test_wide_instances_follow_the_replica_too(engine, pop, mode):
    om = po.make_synthetic_model(376, 17, hid=200, seed=0, ensemble_size=7, elite=[0, 1, 2, 3, 4], termination="humanoid")
    engine.set_model(to_spec(om, 376, 17))
    cls, r = engine.kernel_class(pop, 20, 40, mode)
    assert cls == "wide" and r == cm.choose_r(pop, 20, 5, mode, {1, 2}, rs=(1, 2), wide=True) == 2
