"""The oracle memos (tests/golden/oracle_cache/rollout_sizes.npz, plans_full_size.npz) re-derived WITHOUT a GPU.

The full-size GPU parity tests (tests/test_gpu_rollout.py, test_gpu_device_mode.py) compare the HIP rollout with the oracle's
answer for the same seeded inputs, and since round 4 they read that answer from a memo keyed by the exact input bytes
(tests/oracle_cache.py).  This file closes the link "oracle pinned to the reference -> GPU compared with the oracle": every entry of
the rollout memo is recomputed here, on the CPU, with the CURRENT oracle, and must (1) exist under the key the GPU test will look
up and (2) hold what the oracle returns now.

  EXACT   every input is a pure function of seeds (`_random_case`): the key and the value are reproduced exactly.
  DEVICE  the per-step permutations come from oracle/feistel_perm.py (integer-exact restatement of common.hpp perm_apply: the key
          includes their bytes, so a hit also proves the restatement equals what the device exported when the entry was recorded),
          the eps from oracle/device_draws.py (numpy Philox4x32-10 + Box-Muller).
  FAST    the member schedule from oracle/device_draws.member_schedule (integer-exact; in the key likewise), the same eps.

The memoised FAST / DEVICE values were computed from eps the DEVICE exported (hardware log / sqrt / sin / cos, ~1 ulp each); the
numpy normals agree with those to ~1e-6 relative, so the recomputed returns agree to ~1e-6 x H, not bit for bit: the bound below
is the tests' own T2 (|err| <= 1e-4 max(1, |v|)).  Workloads whose reward / termination is a threshold function of the state
(cartpole: 0 / 1 per step) can flip one row's step on such an eps difference -- a return then moves by 1 / P; at most 0.5 % of a
population's candidates may exceed T2 there, and never by more than 2 / P per step flipped (asserted).

Plans memo (round 6): the replays of the fused plans (tests/test_gpu_plans_full_size.py) feed the oracle the populations the
ENGINE recorded -- megabytes per entry, not functions of a seed.  Every entry therefore carries a pin record (oracle_cache._pin):
the (seed, stream) counters of the engine's draws, the randomness mode, the FAST geometry's row-tile count, and the population's first
candidates.  A candidate's return depends on its own rows only (their members, their eps), so the head is replayed here on its own:
members from the restated permutations (DEVICE) / member schedule + row dealing (FAST), eps from the restated Philox normals, and the
result must equal the head of the stored values to T2.  A change of oracle/device_draws.fast_row_workgroup, of the schedule or of
the permutation restatement makes THIS test fail, on a machine without a GPU.
"""
import json

import numpy as np
import pytest
import torch

import oracle_cache as oc
from oracle import device_draws, feistel_perm
from oracle import pets_oracle as po
from test_gpu_rollout import DEVICE_SIZES, FAST_SIZES, SIZES, _random_case

_IDS = lambda c: f"obs{c[0]}_pop{c[2]}x{c[3]}_H{c[4]}_hid{c[5]['hid']}"  # noqa: E731


def _group():
    g = oc._groups.get("rollout_sizes") or oc._Group("rollout_sizes")
    assert g.data, "tests/golden/oracle_cache/rollout_sizes.npz is missing or empty"
    return g


def _close(fresh, cached, om, P):
    fresh, cached = fresh.double(), torch.from_numpy(cached).double()
    err = (fresh - cached).abs()
    tol = 1e-4 * torch.clamp(cached.abs(), min=1.0)
    bad = err > tol
    if om.reward in ("cartpole", "inverted_pendulum") or om.termination not in (None, "none"):
        # threshold functions: an eps difference of 1e-7 may flip a row's step (see the module docstring)
        assert bad.double().mean() <= 0.005, f"{int(bad.sum())} of {bad.numel()} candidates beyond T2"
        return
    assert not bad.any(), f"max |cached - fresh| {err.max():.3e} (T2 bound {tol[err.argmax()]:.3e})"


@pytest.mark.parametrize("case", SIZES, ids=_IDS)
def test_exact_entries_are_what_the_oracle_returns_now(case):
    obs, act, pop, P, H, mkw = case
    om, actions, s0, perms, eps = _random_case(obs, act, pop, P, H, **mkw)
    key = oc._digest(["exact", *oc.model_parts(om), actions, s0, P, perms, eps])
    g = _group()
    assert key in g.data, "no memo entry under the key test_exact_mode_matches_oracle looks up"
    fresh = po.rollout(om, actions, s0, P, perms=perms, eps=eps)
    cached = torch.from_numpy(g.data[key])
    # same inputs, same oracle, same torch build: equal up to the host's BLAS summation order
    assert (fresh - cached).abs().max() <= 1e-4 * max(1.0, float(cached.abs().max())) * 0.5
    assert torch.allclose(fresh, cached, rtol=2e-5, atol=2e-5)


def _cpu_eps(om, H, B, seed, sid):
    return None if om.deterministic else torch.from_numpy(device_draws.fast_normals(H, B, om.out_size, seed, sid))


@pytest.mark.parametrize("case", DEVICE_SIZES, ids=_IDS)
def test_device_entries_replayed_with_cpu_restatements_of_the_draws(case):
    obs, act, pop, P, H, mkw = case
    om, actions, s0, _, _ = _random_case(obs, act, pop, P, H, **mkw)
    seed, sid = 4321, 5  # test_gpu_device_mode.py::test_device_mode_replayed_through_oracle
    B = pop * P
    perms = None
    if om.propagation == "fixed_model":
        perms = torch.from_numpy(feistel_perm.permutation(B, seed, sid, 0xFFFFFFFF))
    elif om.propagation != "expectation":
        perms = torch.from_numpy(np.stack([feistel_perm.permutation(B, seed, sid, t) for t in range(H)]))
    key = oc._digest(["device", *oc.model_parts(om), actions, s0, P, perms, ("philox", seed, sid)])
    g = _group()
    assert key in g.data, "no memo entry under this key: the device's exported permutations differ from oracle/feistel_perm.py, or the entry was never recorded"
    fresh = po.rollout(om, actions, s0, P, perms=perms, eps=_cpu_eps(om, H, B, seed, sid))
    _close(fresh, g.data[key], om, P)


@pytest.mark.parametrize("case", FAST_SIZES, ids=_IDS)
def test_fast_entries_replayed_with_cpu_restatements_of_the_draws(case):
    obs, act, pop, P, H, mkw = case
    om, actions, s0, _, _ = _random_case(obs, act, pop, P, H, **mkw)
    seed, sid = 1234, 77  # test_gpu_rollout.py::test_fast_mode_replayed_through_oracle
    B, M = pop * P, len(om.active_members)
    fixed = om.propagation == "fixed_model"
    g = _group()
    rows = torch.arange(B)
    hit = None
    for r in (1, 2, 3, 4):  # the row-tile count is the engine's choice (hipets_fast_geometry): the recorded entry names it through its key
        nwg = device_draws.fast_workgroups(B, r)
        sched = torch.from_numpy(device_draws.member_schedule(1 if fixed else H, nwg, M, seed, sid, fixed=fixed, iid=om.ensemble_kind == "basic_ensemble"))
        wg = device_draws.fast_row_workgroup(rows, P, r)
        members = torch.stack([sched[0 if fixed else t][wg].long() for t in range(H)])
        key = oc._digest(["fast", *oc.model_parts(om), actions, s0, P, members, ("philox", seed, sid)])
        if key in g.data:
            hit = (key, members)
            break
    assert hit, "no memo entry for any row-tile count: the device's exported schedule differs from oracle/device_draws.py, or the entry was never recorded"
    key, members = hit
    fresh = po.rollout(om, actions, s0, P, members=members, eps=_cpu_eps(om, H, B, seed, sid))
    _close(fresh, g.data[key], om, P)


def test_fast_row_dealing_restatement_is_the_headers_statement():
    """include/hipets.h (hipets_fast_schedule): the B = pop * P rows form one particle-major run, run index g = p * pop + c for batch row
    c * P + p, and workgroup w owns run indices [w * 16 * row_tiles, (w + 1) * 16 * row_tiles) -- spelled out with loops here and compared
    with oracle/device_draws.fast_row_workgroup / fast_workgroups, which every FAST replay of the GPU suite uses."""
    for pop, P, r in [(805, 20, 2), (500, 20, 3), (100, 5, 1), (7, 3, 1), (16, 1, 1), (17, 1, 4), (1036, 20, 2)]:
        B = pop * P
        owner = np.full(B, -1)
        for g in range(B):
            p, c = divmod(g, pop)
            owner[c * P + p] = g // (16 * r)
        assert (owner >= 0).all()
        assert np.array_equal(device_draws.fast_row_workgroup(np.arange(B), P, r), owner)
        assert np.array_equal(device_draws.fast_row_workgroup(torch.arange(B), P, r).numpy(), owner)
        assert device_draws.fast_workgroups(B, r) == owner.max() + 1 == -(-(-(-B // 16)) // r)
    assert device_draws.fast_workgroups(805 * 20, 2) == 504  # (per particle it was 26 x 20 = 520: a third round on 256 CUs)


# ---- plans memo ------------------------------------------------------------------------------------------------------------

def _plans_group():
    g = oc._groups.get("plans_full_size") or oc._Group("plans_full_size")
    assert g.data, "tests/golden/oracle_cache/plans_full_size.npz is missing or empty"
    return g


def _plan_pins():
    g = _plans_group()
    return sorted(k[: -len(oc.PIN_META)] for k in g.data if k.endswith(oc.PIN_META))


def test_every_plans_memo_entry_carries_a_pin_record():
    g = _plans_group()
    values = [k for k in g.data if not k.endswith(oc.PIN_META) and not k.endswith(oc.PIN_HEAD)]
    pins = set(_plan_pins())
    assert values and set(values) == pins, f"{len(set(values) - pins)} entries without a pin record, {len(pins - set(values))} pins without an entry"


def _head_members(meta, om, head_rows, B):
    """[H, len(head_rows)] member slot of the head's rows at every step, from the CPU restatements of the engine's draws."""
    H, seed, stream, P = meta["H"], meta["seed"], meta["stream"], meta["P"]
    M = len(om.active_members)
    fixed = om.propagation == "fixed_model"
    if meta["mode"] == "device":  # slot j holds row perm[j] and runs member j // (B / M) (gaussian_mlp.py:164-166, 203-205)
        out = np.empty((H, len(head_rows)), dtype=np.int64)
        for t in range(H):
            perm = feistel_perm.permutation(B, seed, stream, 0xFFFFFFFF if fixed else t)
            slot_of_row = np.empty(B, dtype=np.int64)
            slot_of_row[perm] = np.arange(B)
            out[t] = slot_of_row[head_rows] // (B // M)
        return torch.from_numpy(out)
    r = meta["row_tiles"]
    nwg = device_draws.fast_workgroups(B, r)
    sched = device_draws.member_schedule(1 if fixed else H, nwg, M, seed, stream, fixed=fixed, iid=om.ensemble_kind == "basic_ensemble")
    wg = device_draws.fast_row_workgroup(np.arange(B), P, r)[head_rows]
    return torch.from_numpy(np.stack([sched[0 if fixed else t][wg] for t in range(H)]).astype(np.int64))


@pytest.mark.parametrize("key", _plan_pins() or [None])
def test_plans_entries_replayed_from_their_pin_records(key):
    if key is None:
        pytest.fail("tests/golden/oracle_cache/plans_full_size.npz holds no pin records: re-record it (profiles/session_r6_memo.sh)")
    from test_gpu_plans_full_size import make_case

    g = _plans_group()
    meta = json.loads(bytes(g.data[key + oc.PIN_META]).decode())
    head = torch.from_numpy(g.data[key + oc.PIN_HEAD])
    stored = torch.from_numpy(g.data[key])
    c, om, s0 = make_case(meta["case"])
    P, H, pop = meta["P"], meta["H"], meta["pop"]
    assert stored.shape == (pop,) and head.shape[1:] == (H, c["act"]) and 1 <= head.shape[0] <= pop
    n = head.shape[0]
    B = pop * P
    head_rows = np.arange(n * P)  # batch row c * P + p: the head's candidates own the first n * P rows
    members = _head_members(meta, om, head_rows, B)
    eps = torch.from_numpy(device_draws.fast_normals(H, B, om.out_size, meta["seed"], meta["stream"], rows=head_rows))
    fresh = po.rollout(om, head, s0, P, members=members, eps=eps).double()
    want = stored[:n].double()
    err = (fresh - want).abs()
    tol = 1e-4 * torch.clamp(want.abs(), min=1.0)
    bad = err > tol
    if om.reward in ("cartpole", "inverted_pendulum") or om.termination not in (None, "none", "no_termination"):
        # threshold functions: an eps difference of 1e-7 may flip one row's step: that candidate's return moves by k / P
        assert int(bad.sum()) <= 1 and float(err.max()) <= 3.0 / P + 1e-4, f"{int(bad.sum())} of {n} head candidates beyond T2, max {err.max():.3e}"
        return
    assert not bad.any(), f"max |stored - fresh| {err.max():.3e} (T2 bound {tol[err.argmax()]:.3e}): {meta}"
