"""Host-side logic that needs no GPU: model snapshot validation, config plumbing, sharding maths."""
import numpy as np
import pytest
import torch

import hipets
from hipets import dist as hdist
from hipets.model import ModelSpec, UnsupportedModelError
from hipets.planning import _instantiate, complete_agent_cfg


def small_spec(**kw):
    E, obs, act, hid = 3, 5, 2, 8
    d = dict(
        weights=[torch.zeros(E, obs + act, hid), torch.zeros(E, hid, hid), torch.zeros(E, hid, 2 * obs)],
        biases=[torch.zeros(E, 1, hid), torch.zeros(E, 1, hid), torch.zeros(E, 1, 2 * obs)],
        obs_dim=obs, act_dim=act, min_logvar=-10 * torch.ones(1, obs), max_logvar=0.5 * torch.ones(1, obs),
    )
    d.update(kw)
    return ModelSpec(**d)


def test_spec_derived_fields_and_flops():
    s = small_spec(elite_models=[0, 2])
    s.validate()
    assert (s.in_dim, s.hid, s.out_dim, s.ensemble_size, s.members) == (7, 8, 5, 3, [0, 2])
    assert s.flops_per_candidate_step() == 2 * (7 * 8 + 8 * 8 + 8 * 10)


def test_cfg2_flops_match_survey():
    """262 800 FLOP per candidate-step for obs 17 / act 6 / hid 200 / 4 hidden layers."""
    E = 1
    ws = [torch.zeros(E, 23, 200)] + [torch.zeros(E, 200, 200)] * 3 + [torch.zeros(E, 200, 34)]
    s = ModelSpec(weights=ws, biases=[torch.zeros(E, 1, w.shape[2]) for w in ws], obs_dim=17, act_dim=6,
                  min_logvar=torch.zeros(1, 17), max_logvar=torch.zeros(1, 17))
    assert s.flops_per_candidate_step() == 262800


@pytest.mark.parametrize("kw,exc", [
    (dict(activation="gelu"), UnsupportedModelError),
    (dict(propagation="nope"), ValueError),
    (dict(reward="my_reward"), UnsupportedModelError),
    (dict(reward=None), UnsupportedModelError),
    (dict(termination="my_term"), UnsupportedModelError),
    (dict(obs_process="custom"), UnsupportedModelError),
    (dict(obs_dim=6), UnsupportedModelError),
])
def test_unsupported_models_are_rejected_not_approximated(kw, exc):
    with pytest.raises(exc):
        small_spec(**kw).validate()


class _Lin:
    def __init__(self, w, b):
        self.weight, self.bias, self.use_bias = torch.nn.Parameter(w), torch.nn.Parameter(b), True


class _FakeMLP:
    def __init__(self, spec, act):
        self.hidden_layers = [[_Lin(w, b), act] for w, b in zip(spec.weights[:-1], spec.biases[:-1])]
        self.mean_and_logvar = _Lin(spec.weights[-1], spec.biases[-1])
        self.min_logvar, self.max_logvar = spec.min_logvar, spec.max_logvar
        self.elite_models, self.propagation_method, self.deterministic = None, "random_model", False

    def parameters(self):
        for layer in self.hidden_layers:
            yield layer[0].weight
            yield layer[0].bias
        yield self.mean_and_logvar.weight


def halfcheetah(act, next_obs):
    return next_obs[:, :1]


def no_termination(act, next_obs):
    return torch.zeros(len(next_obs), 1, dtype=torch.bool)


# stand-ins for mbrl.env.reward_fns.halfcheetah / termination_fns.no_termination: functions defined elsewhere must opt in
# explicitly, a bare name is not enough (see test_user_functions_with_builtin_names_are_not_silently_replaced)
halfcheetah.hipets_closed_form = "halfcheetah"
no_termination.hipets_closed_form = "no_termination"


class _Space:
    def __init__(self, n):
        self.shape = (n,)
        self.low, self.high = -np.ones(n), np.ones(n)


class _FakeModelEnv:
    def __init__(self, act_module=torch.nn.SiLU()):
        s = small_spec()

        class DM:
            pass

        self.dynamics_model = DM()
        self.dynamics_model.model = _FakeMLP(s, act_module)
        self.dynamics_model.input_normalizer = None
        self.dynamics_model.obs_process_fn = None
        self.dynamics_model.target_is_delta = True
        self.dynamics_model.no_delta_list = []
        self.dynamics_model.learned_rewards = False
        self.reward_fn, self.termination_fn = halfcheetah, no_termination
        self.observation_space, self.action_space = _Space(5), _Space(2)


def test_spec_from_duck_typed_model_env():
    me = _FakeModelEnv()
    spec = hipets.spec_from_model_env(me)
    assert spec.activation == "silu" and spec.reward == "halfcheetah" and spec.termination == "no_termination"
    assert spec.obs_dim == 5 and spec.act_dim == 2 and len(spec.weights) == 3


def test_spec_rejects_unknown_activation_and_python_reward():
    with pytest.raises(UnsupportedModelError):
        hipets.spec_from_model_env(_FakeModelEnv(torch.nn.GELU()))
    me = _FakeModelEnv()
    me.reward_fn = lambda a, o: o[:, :1]
    with pytest.raises(UnsupportedModelError):
        hipets.spec_from_model_env(me)


def test_user_functions_with_builtin_names_are_not_silently_replaced():
    """A user reward called `halfcheetah` (or an env class called HalfCheetahEnv) is NOT the closed form of mbrl.env:
    recognition goes by defining module, so such models are rejected (or routed to the unfused path), never approximated."""
    def halfcheetah(act, next_obs):  # noqa: F811  same NAME as the built-in, different function
        return 2.0 * next_obs[:, :1]

    me = _FakeModelEnv()
    me.reward_fn = halfcheetah
    with pytest.raises(UnsupportedModelError, match="reward_fn"):
        hipets.spec_from_model_env(me)
    spec = hipets.spec_from_model_env(me, allow_custom_fns=True)
    assert spec.reward == "none" and spec.custom_reward_fn is halfcheetah

    class HalfCheetahEnv:  # same class name as mbrl.env.pets_halfcheetah.HalfCheetahEnv, different preprocessing
        @staticmethod
        def preprocess_fn(x):
            return x

    me = _FakeModelEnv()
    me.dynamics_model.obs_process_fn = HalfCheetahEnv.preprocess_fn
    with pytest.raises(UnsupportedModelError, match="obs_process_fn"):
        hipets.spec_from_model_env(me)
    # a callable without a name (functools.partial) with learned rewards: still not "learned reward" (model_env.py:124-128)
    import functools

    me = _FakeModelEnv()
    me.dynamics_model.learned_rewards = True
    me.reward_fn = functools.partial(lambda scale, a, o: scale * o[:, :1], 3.0)
    with pytest.raises(UnsupportedModelError):
        hipets.spec_from_model_env(me)
    # functions that really live in mbrl.env.* are recognised by module + name
    fn = lambda a, o: o[:, :1]  # noqa: E731
    fn.__module__, fn.__name__ = "mbrl.env.reward_fns", "halfcheetah"
    me = _FakeModelEnv()
    me.reward_fn = fn
    assert hipets.spec_from_model_env(me).reward == "halfcheetah"


class MissingMandatoryValue(Exception):
    """Same name as omegaconf.errors.MissingMandatoryValue (omegaconf itself is not installed here)."""


class _DictConfigLike:
    """Behaves like an OmegaConf DictConfig where it matters: reading a key that holds "???" RAISES (it is not a KeyError),
    assignment works, keys() lists missing keys too."""

    def __init__(self, **kw):
        self._d = dict(kw)

    def keys(self):
        return self._d.keys()

    def __getitem__(self, k):
        v = self._d[k]
        if isinstance(v, str) and v == "???":
            raise MissingMandatoryValue(f"Missing mandatory value: {k}")
        return v

    def __setitem__(self, k, v):
        self._d[k] = v

    def __contains__(self, k):
        return k in self._d


def test_stock_configs_with_missing_values_as_omegaconf_presents_them():
    """conf/action_optimizer/cem.yaml ships `lower_bound: ???`, conf/algorithm/pets.yaml `action_lb: ???`: under OmegaConf
    reading them raises MissingMandatoryValue.  complete_agent_cfg fills the action bounds, _instantiate drops what is
    still missing once the overrides are applied (the reference writes the bounds into the cfg first, trajectory_opt.py:525-527)."""
    cfg = _DictConfigLike(_target_="hipets.TrajectoryOptimizerAgent", action_lb="???", action_ub="???", planning_horizon=3)
    complete_agent_cfg(_FakeModelEnv(), cfg)
    assert cfg["action_lb"] == [-1.0, -1.0] and cfg["action_ub"] == [1.0, 1.0]
    obj = _instantiate(_DictConfigLike(_target_="fractions.Fraction", numerator="???", denominator="???"), numerator=3)
    assert obj == 3


def test_complete_agent_cfg_fills_placeholders():
    cfg = dict(_target_="hipets.TrajectoryOptimizerAgent", action_lb="???", action_ub="???", planning_horizon=3)
    complete_agent_cfg(_FakeModelEnv(), cfg)
    assert cfg["action_lb"] == [-1.0, -1.0] and cfg["action_ub"] == [1.0, 1.0]


def test_instantiate_resolves_target_and_drops_placeholders():
    obj = _instantiate(dict(_target_="fractions.Fraction", numerator=3, denominator="???"))
    assert obj == 3


def test_shard_bounds_cover_population():
    for pop, world in [(500, 8), (500, 1), (7, 8), (2000, 4), (1001, 3)]:
        spans = [hdist.shard_bounds(pop, world, r) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == pop
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1
    assert [b - a for a, b in (hdist.shard_bounds(500, 8, r) for r in range(8))] == [63, 63, 63, 63, 62, 62, 62, 62]


def test_spec_from_checkpoint_files(tmp_path):
    """model.pth + env_stats.pickle in the reference's on-disk format (gaussian_mlp.py:381-387, util/math.py:168-174)."""
    import pickle

    s = small_spec()
    sd = {}
    for i, (w, b) in enumerate(zip(s.weights[:-1], s.biases[:-1])):
        sd[f"hidden_layers.{i}.0.weight"], sd[f"hidden_layers.{i}.0.bias"] = w + i, b
    sd["mean_and_logvar.weight"], sd["mean_and_logvar.bias"] = s.weights[-1], s.biases[-1]
    sd["min_logvar"], sd["max_logvar"] = s.min_logvar, s.max_logvar
    torch.save({"state_dict": sd, "elite_models": [2, 0]}, tmp_path / "model.pth")
    with open(tmp_path / "env_stats.pickle", "wb") as f:
        pickle.dump({"mean": np.full((1, 7), 0.5), "std": np.full((1, 7), 2.0)}, f)
    spec = hipets.spec_from_checkpoint(tmp_path, obs_dim=5, act_dim=2, termination="hopper")
    assert spec.members == [2, 0] and spec.termination == "hopper" and not spec.deterministic
    assert spec.norm_mean.dtype == torch.float64 and float(spec.norm_std[0, 0]) == 2.0
    assert len(spec.weights) == 3 and torch.equal(spec.weights[1], s.weights[1] + 1)


def test_member_slots_pads_unbalanced_member_maps():
    """Row -> member maps of BasicEnsemble models (randint, unbalanced) become padded per-member slot tables."""
    from hipets.engine import member_slots

    g = torch.Generator().manual_seed(0)
    T, B, M = 3, 29, 4
    members = torch.randint(M, (T, B), generator=g)
    members[1] = 2  # one step where a single member owns every row
    slots, rpm = member_slots(members, M, "cpu")
    assert rpm == B and tuple(slots.shape) == (T, M * rpm)
    for t in range(T):
        seen = []
        for m in range(M):
            blk = slots[t, m * rpm:(m + 1) * rpm]
            rows = blk[blk >= 0]
            assert torch.equal(rows, (members[t] == m).nonzero().flatten())  # ascending rows of member m ...
            assert (blk[len(rows):] == -1).all()  # ... then padding only
            seen.append(rows)
        assert torch.equal(torch.cat(seen).sort().values, torch.arange(B))
    tight, rpm2 = member_slots(members[:1], M, "cpu")
    assert rpm2 == int(torch.bincount(members[0], minlength=M).max()) and tight.shape[1] == M * rpm2


def test_reference_noise_consumes_the_global_generator_like_the_reference():
    """sampler='torch': the product's draw routine == mbrl.util.math.truncated_normal_ (util/math.py:69-92, restated in the
    oracle and pinned bitwise against the reference) on the same torch.manual_seed; the clipped-normal branch is randn."""
    from hipets.planning import _reference_noise
    from oracle import pets_oracle as po

    torch.manual_seed(123)
    a = _reference_noise((40, 6, 3), clipped_normal=False)
    after_a = torch.rand(1)
    torch.manual_seed(123)
    b = po.truncated_normal_(torch.zeros(40, 6, 3))
    after_b = torch.rand(1)
    assert torch.equal(a, b) and torch.equal(after_a, after_b)  # same values AND same generator state afterwards
    assert a.abs().max() <= 2.0
    torch.manual_seed(5)
    c = _reference_noise((7, 2), clipped_normal=True)
    torch.manual_seed(5)
    assert torch.equal(c, torch.randn(7, 2))


def test_oracle_memo_keys_on_the_exact_input_bytes(tmp_path, monkeypatch):
    """tests/oracle_cache.py: a hit needs byte-identical inputs (one ulp in a population is a miss), misses are written where
    HIPETS_ORACLE_CACHE_OUT says, HIPETS_ORACLE_CACHE=0 ignores stored entries."""
    import importlib

    import numpy as np
    import torch

    import oracle_cache as oc

    oc = importlib.reload(oc)
    monkeypatch.setattr(oc, "_DIR", str(tmp_path / "store"))
    monkeypatch.setenv("HIPETS_ORACLE_CACHE_OUT", str(tmp_path / "store"))
    calls = []

    def compute(v):
        def f():
            calls.append(v)
            return torch.full((3,), float(v))
        return f

    x = torch.linspace(0, 1, 8)
    a = oc.cached("unit", ["fast", x, np.float32(0.5), ("philox", 1, 2)], compute(1))
    b = oc.cached("unit", ["fast", x.clone(), np.float32(0.5), ("philox", 1, 2)], compute(2))  # same bytes: a hit, compute not called
    assert torch.equal(a, b) and calls == [1]
    y = x.clone()
    y[3] = torch.nextafter(y[3], torch.tensor(2.0))  # one ulp
    c = oc.cached("unit", ["fast", y, np.float32(0.5), ("philox", 1, 2)], compute(3))
    d = oc.cached("unit", ["fast", x, np.float32(0.5), ("philox", 1, 3)], compute(4))  # another stream counter
    assert calls == [1, 3, 4] and float(c[0]) == 3.0 and float(d[0]) == 4.0
    assert (tmp_path / "store" / "unit.npz").exists()
    oc2 = importlib.reload(oc)  # a fresh process: entries come back from the file
    monkeypatch.setattr(oc2, "_DIR", str(tmp_path / "store"))
    e = oc2.cached("unit", ["fast", x, np.float32(0.5), ("philox", 1, 2)], compute(5))
    assert float(e[0]) == 1.0 and calls == [1, 3, 4] and oc2.stats["hits"] == 1
    monkeypatch.setenv("HIPETS_ORACLE_CACHE", "0")
    oc3 = importlib.reload(oc2)
    monkeypatch.setattr(oc3, "_DIR", str(tmp_path / "store"))
    f = oc3.cached("unit", ["fast", x, np.float32(0.5), ("philox", 1, 2)], compute(6))
    assert float(f[0]) == 6.0


def test_committed_oracle_memo_is_loadable_and_small():
    import os

    import numpy as np

    from conftest import GOLDEN

    d = os.path.join(GOLDEN, "oracle_cache")
    files = [f for f in os.listdir(d) if f.endswith(".npz")]
    assert files, "tests/golden/oracle_cache is empty: run the GPU suite with HIPETS_ORACLE_CACHE_OUT and commit its output"
    total = 0
    for f in files:
        total += os.path.getsize(os.path.join(d, f))
        with np.load(os.path.join(d, f)) as z:
            # blake2b-128 hex digests, plus the pin records of the plans memo (oracle_cache._pin: "<digest>:meta" / "<digest>:head")
            assert len(z.files) > 0 and all(len(k.split(":")[0]) == 32 and k.split(":")[1:] in ([], ["meta"], ["head"]) for k in z.files)
    assert total < 4 << 20
