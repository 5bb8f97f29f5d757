"""The oracle against the committed golden vectors (made from the real reference by
oracle/make_golden.py) and against the reference's own known-answer tests.  CPU only."""
import glob
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import pets_oracle as po
from oracle.golden_io import load_case

ROLLOUT_FILES = sorted(glob.glob(os.path.join(GOLDEN, "rollout_*.npz")))
CEM_FILES = sorted(glob.glob(os.path.join(GOLDEN, "cem_*.npz")))


def test_golden_present():
    assert len(ROLLOUT_FILES) >= 9 and len(CEM_FILES) >= 3


@pytest.mark.parametrize("path", ROLLOUT_FILES, ids=lambda p: os.path.basename(p)[8:-4])
def test_rollout_matches_reference_golden(path):
    om, meta, a = load_case(path)
    trace = {}
    out = po.rollout(om, a["actions"], a["s0"].numpy(), meta["P"], perms=a.get("perms"), eps=a.get("eps"),
                     members=a.get("members"), trace=trace)
    assert torch.equal(out, a["returns"])  # T0: bitwise
    assert torch.equal(trace["next_obs"][0], a["next_obs_step0"])
    assert torch.equal(trace["rewards"][0], a["rewards_step0"])


@pytest.mark.parametrize("path", CEM_FILES, ids=lambda p: os.path.basename(p)[4:-4])
def test_cem_matches_reference_golden(path):
    z = np.load(path)
    meta = json.loads(bytes(z["meta_json"]).decode())
    a = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("x_")}
    target = a["target"]

    def obj(x):
        v = -((x - target) ** 2).sum(dim=(1, 2))
        v = v.clone()
        v[meta["nan_index"]] = float("nan")
        return v

    rec = []
    out = po.cem_optimize(obj, a["x0"], a["lower"], a["upper"], meta["iters"], meta["elite_ratio"], meta["pop"], meta["alpha"],
                          return_mean_elites=meta["return_mean"], clipped_normal=meta["clipped"], noise=list(a["z"]), record=rec)
    assert torch.equal(out, a["result"])
    for i, r in enumerate(rec):
        assert torch.equal(r["population"], a["populations"][i])
        assert torch.equal(r["mu"], a["mus"][i])
        assert torch.equal(r["disp"], a["disps"][i])
        assert r["values"][meta["nan_index"]] == pytest.approx(-1e-10)  # Appendix B1


def dummy_model_as_mlp(act_dim):
    """The reference's DummyModel (tests/core/test_models.py:337-362: next_obs = obs + mean(act), reward =
    next_obs) written as a deterministic 1-member ReLU MLP: h0 = relu(obs + mean(act)), out = [h0, h0]."""
    hid = 4
    w0 = torch.zeros(1, 1 + act_dim, hid)
    w0[0, 0, 0] = 1.0
    w0[0, 1:, 0] = 1.0 / act_dim
    w1 = torch.zeros(1, hid, 2)
    w1[0, 0, 0] = 1.0
    w1[0, 0, 1] = 1.0
    return po.OracleModel(weights=[w0, w1], biases=[torch.zeros(1, 1, hid), torch.zeros(1, 1, 2)], activation="relu",
                          propagation="expectation", deterministic=True, target_is_delta=False, learned_rewards=True,
                          reward=None, termination="no_termination")


def test_known_answer_evaluate_action_sequences():
    """tests/core/test_models.py:365-385: returns == H (H+1) / 2 * a for P, H in 1..9."""
    act_dim = 2
    om = dummy_model_as_mlp(act_dim)
    for P in range(1, 10):
        for H in range(1, 10):
            acts = torch.stack([torch.ones(H, act_dim), 2 * torch.ones(H, act_dim)])
            expected = H * (H + 1) * acts[..., 0, 0] / 2
            out = po.rollout(om, acts, np.zeros(1), P)
            assert torch.allclose(expected, out)


def test_truncated_normal_support():
    """tests/core/test_common_utils.py:419-423."""
    t = torch.empty((100, 2))
    t2 = po.truncated_normal_(t)
    assert t is t2
    assert (t > -2).all().item() and (t < 2).all().item()


def _tagging_model(E, in_size, out):
    """An ensemble whose member e outputs the constant e in every mean dim (the reference's mocked
    _default_forward trick, tests/core/test_models.py:95-113) -- realised with zero weights + bias e."""
    ws = [torch.zeros(E, in_size, 8), torch.zeros(E, 8, 2 * out)]
    bs = [torch.zeros(E, 1, 8), torch.zeros(E, 1, 2 * out)]
    for e in range(E):
        bs[1][e, 0, :out] = float(e)
    return po.OracleModel(weights=ws, biases=bs, min_logvar=-10 * torch.ones(1, out), max_logvar=0.5 * torch.ones(1, out),
                          activation="relu")


def test_ts1_balance_and_alignment():
    """tests/core/test_models.py:116-152: every member gets B/E rows; outputs stay aligned to inputs."""
    E, B = 5, 100
    om = _tagging_model(E, 3, 2)
    x = torch.randn(B, 3)
    perm = torch.randperm(B)
    mean, _ = po.ensemble_forward(om, x, perm)
    counts = torch.bincount(mean[:, 0].long(), minlength=E)
    assert (counts == B // E).all()
    expect = torch.empty(B)
    expect[perm] = (torch.arange(B) // (B // E)).float()
    assert torch.equal(mean[:, 0], expect)


def test_expectation_averages_members():
    """tests/core/test_models.py:180-198."""
    om = _tagging_model(4, 3, 2)
    om.propagation = "expectation"
    mean, _ = po.ensemble_forward(om, torch.randn(8, 3))
    assert torch.allclose(mean, torch.full_like(mean, 1.5))


def test_batch_not_multiple_of_members_raises():
    om = _tagging_model(5, 3, 2)
    with pytest.raises(ValueError, match="multiple of the number of models"):
        po.ensemble_forward(om, torch.randn(7, 3), torch.randperm(7))


def test_icem_sizes_match_survey():
    """cfg4 sampled populations 1001, 770, 595, 462, 357, kept 35."""
    K, keep, sizes = po.icem_sizes(5, 0.1, 1000, 1.3, 0.3, 7)
    assert (K, keep, sizes) == (100, 35, [1001, 770, 595, 462, 357])


def test_elite_count_ceil():
    assert po.elite_count(500, 0.1) == 50 and po.elite_count(400, 0.16) == 64  # Appendix B9


def _load_npz(name):
    z = np.load(os.path.join(GOLDEN, name))
    meta = json.loads(bytes(z["meta_json"]).decode())
    return meta, {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("x_")}


def _quad(target, nan_at):
    def f(x):
        v = -((x - target) ** 2).sum(dim=(1, 2)).clone()
        v[nan_at] = float("nan")
        return v

    return f


def test_mppi_matches_reference_golden():
    meta, a = _load_npz("mppi_two_calls.npz")
    st = po.MPPIState(meta["H"], meta["A"])
    obj = _quad(a["target"], meta["nan_index"])
    for c in range(meta["calls"]):
        out = po.mppi_optimize(obj, st, a["lower"], a["upper"], meta["iters"], meta["pop"], meta["gamma"], meta["sigma"],
                               meta["beta"], noise=list(a[f"noise{c}"]))
        assert torch.equal(out, a[f"result{c}"])


def test_icem_matches_reference_golden():
    meta, a = _load_npz("icem_two_calls.npz")
    st = po.ICEMState()
    obj = _quad(a["target"], meta["nan_index"])
    for c in range(meta["calls"]):
        inject = []
        for i in range(meta["iters"]):
            d = {"noise": a[f"noise_{c}_{i}"]}
            for k in ("keep_perm", "end_noise"):
                if f"{k}_{c}_{i}" in a:
                    d[k] = a[f"{k}_{c}_{i}"]
            inject.append(d)
        rec = []
        out = po.icem_optimize(obj, st, a[f"x0_{c}"], a["lower"], a["upper"], meta["iters"], meta["elite_ratio"], meta["pop"],
                               meta["decay"], meta["exponent"], meta["keep_frac"], meta["alpha"], return_mean_elites=True,
                               population_size_module=meta["module"], inject=inject, record=rec)
        assert torch.equal(out, a[f"result{c}"])
        assert [int(r["population"].shape[0]) for r in rec] == meta["evaluated_sizes"][c]


def test_colored_noise_from_injected_normals_matches_recorded_noise():
    """The unit normals recorded next to the reference's coloured noise reproduce it (what the device kernel gets)."""
    meta, a = _load_npz("icem_two_calls.npz")
    n = a["normals_0_0"]
    cn = po.powerlaw_psd_gaussian(meta["exponent"], size=(n.shape[1], meta["A"], meta["H"]), normals=(n[0], n[1])).transpose(1, 2)
    assert torch.equal(cn, a["noise_0_0"])


def load_planet_case(path):
    from oracle import planet_oracle as pl

    z = np.load(path)
    meta = json.loads(bytes(z["meta_json"]).decode())
    pm = pl.PlaNetOracleModel(**{k: torch.from_numpy(z[k]) for k in pl.PLANET_TENSORS}, min_std=meta["min_std"])
    arrays = {k[2:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("x_")}
    return pm, meta, arrays


PLANET_FILES = sorted(glob.glob(os.path.join(GOLDEN, "planet_*.npz")))


@pytest.mark.parametrize("path", PLANET_FILES, ids=lambda p: os.path.basename(p)[7:-4])
def test_planet_rollout_matches_reference_golden(path):
    from oracle import planet_oracle as pl

    pm, meta, a = load_planet_case(path)
    trace = {}
    out = pl.planet_rollout(pm, a["actions"], a["latent0"], a["belief0"], meta["P"], eps=a["eps"], trace=trace)
    assert torch.equal(out, a["returns"])
    assert torch.equal(trace["latent"][0], a["latent_step0"]) and torch.equal(trace["belief"][0], a["belief_step0"])
    assert torch.equal(trace["rewards"][0], a["rewards_step0"])
    assert len(PLANET_FILES) >= 2
