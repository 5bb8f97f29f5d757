"""Test infrastructure: a numpy restatement of the device-side random permutation of DEVICE-mode rollouts
(mbrl-lib_amd/csrc/common.hpp: perm_radices / perm_hash / perm_mix64 / perm_key / perm_round_keys / perm_scale / perm_apply).

The engine replaces the reference's per-step ``torch.randperm(B)`` (mbrl/models/gaussian_mlp.py:203-205) by a keyed
bijection of [0, B) evaluated in-kernel.  This file restates that bijection so that (a) its statistical quality can be
checked on the CPU (tests/test_perm_feistel.py: bijectivity, uniform positions, pairwise independence, member balance like
the reference's tests/core/test_models.py:116-152) and (b) the GPU export (hipets_device_perms) can be compared with it
element by element.  Never imported by the product.
"""
import numpy as np

ROUNDS = 6
_M32 = np.uint64(0xFFFFFFFF)


def radices(n: int):
    r = 1
    while r * r < n:
        r += 1
    return r, max(1, (n + r - 1) // r)


def mix64(z):
    z = (np.asarray(z, dtype=np.uint64) + np.uint64(0x9E3779B97F4A7C15))
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def perm_key(seed: int, stream: int, step: int) -> np.uint64:
    with np.errstate(over="ignore"):
        inner = mix64(np.uint64(stream) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(0x5045524D))
        return mix64(np.uint64(seed) ^ inner ^ (np.uint64(step) << np.uint64(32)))


def _hash32(v, k):
    h = (v.astype(np.uint64) * np.uint64(0x9E3779B1) + np.uint64(k)) & _M32
    h ^= h >> np.uint64(16)
    h = (h * np.uint64(0x85EBCA6B)) & _M32
    h ^= h >> np.uint64(13)
    h = (h * np.uint64(0xC2B2AE35)) & _M32
    h ^= h >> np.uint64(16)
    return h


def permutation(n: int, seed: int, stream: int, step: int) -> np.ndarray:
    """perm[j] for j in [0, n): the row slot j holds at `step` (step = 0xFFFFFFFF: the TS-infinity permutation)."""
    with np.errstate(over="ignore"):
        return permutation_from_key(n, perm_key(seed, stream, step))


def permutation_from_key(n: int, key) -> np.ndarray:
    """perm_apply(j, n, radices(n), perm_round_keys(key)) for j in [0, n) (also behind FAST mode's member schedule: device_draws)."""
    a, b = radices(n)
    with np.errstate(over="ignore"):
        ks = [int(mix64(np.uint64(key) + np.uint64(r)) >> np.uint64(16)) & 0xFFFFFFFF for r in range(ROUNDS)]
    x = np.arange(n, dtype=np.uint64)
    out = np.empty(n, dtype=np.int64)
    todo = np.arange(n)
    while todo.size:
        L, R = x // np.uint64(b), x % np.uint64(b)
        for r in range(ROUNDS):
            if r & 1:
                R = R + ((_hash32(L, ks[r]) * np.uint64(b)) >> np.uint64(32))  # perm_scale: floor(h * b / 2^32)
                R = np.where(R >= b, R - np.uint64(b), R)
            else:
                L = L + ((_hash32(R, ks[r]) * np.uint64(a)) >> np.uint64(32))
                L = np.where(L >= a, L - np.uint64(a), L)
        x = L * np.uint64(b) + R
        done = x < n
        out[todo[done]] = x[done].astype(np.int64)
        todo, x = todo[~done], x[~done]
    return out
