"""Memo of the ORACLE's outputs for the full-size GPU parity tests (test infrastructure).

The replays of tests/test_gpu_plans_full_size.py and the BASELINE-size cases of tests/test_gpu_rollout.py feed the engine's own
exported randomness (permutations / member schedules, eps, sampler noise: all counter-based, i.e. functions of seeds) and its
recorded populations to `oracle.pets_oracle.rollout` -- 10^5 .. 10^7 candidate-steps of torch-CPU work per call, 139 CPU-minutes
per run of the suite in round 3 (GPUTEST_r03: 633 s of the driver's 1 200 s).  The oracle is a pure function of its inputs, so its
output is stored under a digest of the EXACT input bytes (model seed and kwargs, s0, the population, every injected draw):

    tests/golden/oracle_cache/<group>.npz      key = blake2b of the inputs, value = the oracle's returns

A hit returns what the oracle returned for precisely these inputs on the machine that recorded the entry (the GPU box's host CPU,
in the session that wrote the file: the same torch build as everywhere else; between CPU models MKL may pick kernels with another
summation order, i.e. differences at fp32 rounding level, three orders of magnitude inside the T2 bound the tests apply); a miss -- a
kernel change that moves a population by one ulp, a new case -- computes the oracle as before and, when HIPETS_ORACLE_CACHE_OUT
names a directory, writes the merged file there for committing (profiles/session_full.sh sets it; copy gpurun_out/oracle_cache/*.npz
to tests/golden/oracle_cache/).  HIPETS_ORACLE_CACHE=0 ignores the stored entries: the suite then runs exactly as in round 3.
Nothing here touches the product, and the comparison the tests make is unchanged.  A stale entry (say, after a change of the Philox
counter layout that the seed-keyed entries cannot see) can only make a test FAIL: the device side is always computed afresh."""
import hashlib
import os

import numpy as np
import torch

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_cache")
_groups = {}
stats = {"hits": 0, "misses": 0}


def _digest(parts) -> str:
    h = hashlib.blake2b(digest_size=16)
    for p in parts:
        if p is None:
            h.update(b"<none>")
        elif isinstance(p, torch.Tensor):
            a = p.detach().cpu().contiguous().numpy()
            h.update(str((a.dtype, a.shape)).encode())
            h.update(memoryview(a).cast("B"))
        elif isinstance(p, np.ndarray):
            a = np.ascontiguousarray(p)
            h.update(str((a.dtype, a.shape)).encode())
            h.update(memoryview(a).cast("B"))
        else:
            h.update(repr(p).encode())
    return h.hexdigest()


class _Group:
    def __init__(self, name):
        self.name, self.path = name, os.path.join(_DIR, name + ".npz")
        self.data, self.dirty, self.used = {}, False, set()
        if os.environ.get("HIPETS_ORACLE_CACHE", "1") != "0" and os.path.exists(self.path):
            with np.load(self.path) as z:
                self.data = {k: z[k] for k in z.files}

    def flush(self):
        out = os.environ.get("HIPETS_ORACLE_CACHE_OUT")
        if self.dirty and out:
            os.makedirs(out, exist_ok=True)
            np.savez_compressed(os.path.join(out, self.name + ".npz"), **self.data)
            self.dirty = False


PIN_META, PIN_HEAD = ":meta", ":head"  # suffixes of an entry's pin records (see `cached`)


def _pin(g, key, pin):
    """Store, next to entry `key`, what a machine WITHOUT a GPU needs to re-derive a part of it (tests/test_oracle_memo_pinned.py):
    `meta` -- a JSON-able dict naming the case, the randomness mode and the (seed, stream) counters of the engine's draws -- and
    `head` -- the first few candidates of the population the engine recorded (the whole population would be megabytes per entry;
    a candidate's return depends on its own rows only, so the head can be replayed on its own)."""
    import json

    g.used.update((key + PIN_META, key + PIN_HEAD))
    if key + PIN_META in g.data and key + PIN_HEAD in g.data:
        return
    g.data[key + PIN_META] = np.frombuffer(json.dumps(pin["meta"], sort_keys=True).encode(), dtype=np.uint8).copy()
    g.data[key + PIN_HEAD] = pin["head"].detach().cpu().contiguous().numpy().copy()
    g.dirty = True


def cached(group: str, inputs, compute, verify: bool = False, pin=None):
    """`compute()` (an oracle call) memoised under the digest of `inputs` (tensors, arrays, plain values) in file `group`.
    verify=True (one full-size case per randomness mode, round-4 advice): the oracle runs even on a hit -- so the exports its
    `compute` pulls from the engine (hipets_fast_normals, ...) stay exercised at full size -- and the stored entry must equal what
    it returns NOW (bound: a tenth of the tests' T2); the fresh value is what the test then compares the device with.
    pin={"meta": dict, "head": tensor}: see `_pin`."""
    g = _groups.get(group)
    if g is None:
        g = _groups[group] = _Group(group)
    key = _digest(inputs)
    g.used.add(key)
    if pin is not None:
        _pin(g, key, pin)
    if key in g.data and verify:
        stats["verified"] = stats.get("verified", 0) + 1
        fresh = compute()
        stored = torch.from_numpy(g.data[key].copy())
        err = (fresh.detach().cpu().double() - stored.double()).abs()
        assert bool((err <= 1e-5 * torch.clamp(stored.double().abs(), min=1.0)).all()), \
            f"oracle memo entry {group}/{key} is stale: max |stored - fresh| = {float(err.max()):.3e}"
        return fresh
    if key in g.data:
        stats["hits"] += 1
        return torch.from_numpy(g.data[key].copy())
    stats["misses"] += 1
    out = compute()
    g.data[key] = out.detach().cpu().numpy()
    g.dirty = True
    g.flush()
    return out


def model_parts(om):
    """Everything of an OracleModel the oracle's rollout reads (weights included: ~10 ms of hashing for the widest model)."""
    return (list(om.weights) + list(om.biases) + [om.min_logvar, om.max_logvar, om.norm_mean, om.norm_std] +
            [("fields", om.elite_models, om.activation, om.propagation, om.deterministic, om.target_is_delta, list(om.no_delta_list),
              om.learned_rewards, om.obs_process, om.reward, om.termination, om.ensemble_kind)])


def flush_all():
    # HIPETS_ORACLE_CACHE_PRUNE=1 (whole-suite runs only): entries no test asked for are dropped from the files written -- what a change of
    # the device's randomness or geometry leaves behind (round 5: FAST rows dealt as one run changed every FAST entry's key)
    prune = os.environ.get("HIPETS_ORACLE_CACHE_PRUNE") == "1"
    for g in _groups.values():
        if prune and set(g.data) - g.used:
            g.data = {k: v for k, v in g.data.items() if k in g.used}
            g.dirty = True
        g.flush()
