"""The counter-based random permutation behind DEVICE-mode rollouts (csrc/common.hpp perm_apply), restated in
oracle/feistel_perm.py: it must be a bijection and behave like the reference's torch.randperm(B) for the purposes the
reference uses it (gaussian_mlp.py:164-166, 203-205: a balanced uniform assignment of rows to members, fresh every step).
Statistical checks mirror the reference's own propagation tests (tests/core/test_models.py:116-152)."""
import numpy as np
import pytest

from oracle import feistel_perm as fp


@pytest.mark.parametrize("n", [1, 2, 5, 100, 500, 2500, 10000, 20720, 40000])
def test_is_a_bijection(n):
    for step in (0, 1, 29, 0xFFFFFFFF):
        p = fp.permutation(n, seed=7, stream=3, step=step)
        assert p.min() == 0 and p.max() == n - 1 and np.unique(p).size == n


def test_depends_on_seed_stream_and_step():
    base = fp.permutation(1000, 1, 2, 3)
    assert np.array_equal(base, fp.permutation(1000, 1, 2, 3))
    for other in (fp.permutation(1000, 2, 2, 3), fp.permutation(1000, 1, 3, 3), fp.permutation(1000, 1, 2, 4)):
        assert (other != base).mean() > 0.95


def test_member_balance_is_exact_and_assignment_is_uniform():
    """Every member gets exactly B / M rows each step (balance, like test_models.py:116-131); over many steps every row
    visits every member with probability 1 / M (chi-square over the row x member table)."""
    B, M, steps = 1000, 5, 400
    counts = np.zeros((B, M))
    for t in range(steps):
        p = fp.permutation(B, seed=11, stream=5, step=t)
        member_of_slot = np.arange(B) // (B // M)
        m = np.empty(B, dtype=np.int64)
        m[p] = member_of_slot  # row p[j] runs on member j // (B / M)
        assert np.bincount(m, minlength=M).tolist() == [B // M] * M
        counts[np.arange(B), m] += 1
    chi2 = ((counts - steps / M) ** 2 / (steps / M)).sum()
    dof = B * (M - 1)
    assert abs(chi2 - dof) < 5 * np.sqrt(2 * dof)  # a true uniform assignment gives chi2 ~ dof +- sqrt(2 dof)


def test_positions_are_uniform_and_successive_steps_independent():
    n, steps = 256, 3000
    pos = np.zeros((n, n))
    same = 0
    prev = None
    for t in range(steps):
        p = fp.permutation(n, seed=3, stream=9, step=t)
        pos[np.arange(n), p] += 1
        if prev is not None:
            same += int((p == prev).sum())  # fixed points between consecutive permutations: expectation 1 per pair
        prev = p
    chi2 = ((pos - steps / n) ** 2 / (steps / n)).sum()
    dof = n * n - 2 * n + 1  # doubly stochastic table
    assert abs(chi2 - dof) < 6 * np.sqrt(2 * dof)
    assert abs(same / (steps - 1) - 1.0) < 0.15


def test_pairs_of_rows_share_a_member_as_often_as_under_a_uniform_permutation():
    """Two given rows land on the same member with probability (B/M - 1) / (B - 1) under a uniform balanced shuffle; rows
    that are neighbours in the batch (particles of one candidate) must not be correlated."""
    B, M, steps = 200, 5, 4000
    rpm = B // M
    hits_adjacent = hits_far = 0
    for t in range(steps):
        p = fp.permutation(B, seed=21, stream=1, step=t)
        m = np.empty(B, dtype=np.int64)
        m[p] = np.arange(B) // rpm
        hits_adjacent += int((m[0::2] == m[1::2]).sum())
        hits_far += int((m[: B // 2] == m[B // 2:]).sum())
    expect = (rpm - 1) / (B - 1)
    n_pairs = steps * (B // 2)
    se = np.sqrt(expect * (1 - expect) / n_pairs)
    assert abs(hits_adjacent / n_pairs - expect) < 5 * se
    assert abs(hits_far / n_pairs - expect) < 5 * se
