"""Test infrastructure: numpy restatements of the draws the rollout kernels make IN-KERNEL, so that what the engine exports on a
GPU (hipets_fast_normals, hipets_fast_schedule; hipets_device_perms has oracle/feistel_perm.py) can be re-derived where there is no
GPU -- tests/test_oracle_memo_pinned.py replays the memoised FAST / DEVICE entries of tests/golden/oracle_cache/ with them.

Restated from mbrl-lib_amd/csrc/common.hpp (philox4x32_10, u01, box_muller, fast_member / fast_member_key), csrc/rollout.hpp
(rollout_normals4: counter = (row, step, block of four output dims, stream), key = seed) and csrc/rollout_helpers.hpp
(member_schedule_kernel, export_normals_kernel).  The integer parts (Philox, the schedule) are exact; Box-Muller runs on the CPU's float32 log2 / sqrt /
sin / cos where the device uses v_log_f32 / v_sqrt_f32 / v_sin_f32 / v_cos_f32 (about 1 ulp each): the normals agree to ~1e-6
relative, not bit for bit.  They replace torch.randn draws of the reference (mbrl/models/model.py:471-473) and its per-step
member assignment (mbrl/models/gaussian_mlp.py:203-205, 267-275).  Never imported by the product.
"""
import numpy as np

_M32 = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Philox4x32-10 on arrays of 32-bit counters (held in uint64 for the 32 x 32 -> 64 products); returns four uint64 arrays < 2^32."""
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint64) & _M32 for c in (c0, c1, c2, c3))
    k0, k1 = np.uint64(k0 & 0xFFFFFFFF), np.uint64(k1 & 0xFFFFFFFF)
    m0, m1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
    for _ in range(10):
        p0, p1 = m0 * c0, m1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & _M32
        hi1, lo1 = p1 >> np.uint64(32), p1 & _M32
        c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
        k0 = (k0 + np.uint64(0x9E3779B9)) & _M32
        k1 = (k1 + np.uint64(0xBB67AE85)) & _M32
    return c0, c1, c2, c3


def _u01(x):
    return ((x >> np.uint64(8)).astype(np.float32) + np.float32(0.5)) * np.float32(1.0 / 16777216.0)


def _box_muller(a, b):
    r = np.sqrt(np.float32(-1.38629436111989061883) * np.log2(_u01(a)), dtype=np.float32)
    ang = np.float32(2.0 * np.pi) * _u01(b)  # v_sin / v_cos take revolutions
    return (r * np.cos(ang, dtype=np.float32)).astype(np.float32), (r * np.sin(ang, dtype=np.float32)).astype(np.float32)


def fast_normals(H: int, B: int, out_dim: int, seed: int, stream_id: int, rows=None) -> np.ndarray:
    """[H, B, out_dim] float32: what hipets_fast_normals exports (eps of row `rid` at step `t`, output dim `d`).  `rows`: only these
    batch rows ([H, len(rows), out_dim]; the counters are per row, so a subset costs a subset)."""
    nblk = (out_dim + 3) // 4
    rids = np.arange(B, dtype=np.uint64) if rows is None else np.asarray(rows, dtype=np.uint64)
    t, rid, blk = np.meshgrid(np.arange(H, dtype=np.uint64), rids, np.arange(nblk, dtype=np.uint64), indexing="ij")
    k0 = seed & 0xFFFFFFFF
    k1 = ((seed >> 32) ^ (stream_id >> 32)) & 0xFFFFFFFF
    x, y, z, w = philox4x32_10(rid, t, blk, np.uint64(stream_id & 0xFFFFFFFF), k0, k1)
    n0, n1 = _box_muller(x, y)
    n2, n3 = _box_muller(z, w)
    full = np.stack([n0, n1, n2, n3], axis=-1).reshape(H, len(rids), nblk * 4)
    return np.ascontiguousarray(full[:, :, :out_dim])


def member_schedule(H: int, nwg: int, M: int, seed: int, stream_id: int, fixed: bool = False, iid: bool = False) -> np.ndarray:
    """[H, nwg] int32: what hipets_fast_schedule exports (member slot of workgroup `wg` at step `t`; csrc/common.hpp fast_member).
    Step t's keyed bijection p_t of the workgroup indices -- the Feistel network of oracle/feistel_perm.py under the key
    fast_member_key(seed, stream, t) -- cut into M equal runs, rotated by the step's offset r: slot = ((p_t(wg) * M + r) // nwg) % M
    (every slot gets floor / ceil(nwg / M) workgroups); fixed (TS-infinity): one draw, step 0xFFFFFFFF; iid (BasicEnsemble): independent uniform draws."""
    from . import feistel_perm as fp

    out = np.empty((H, nwg), dtype=np.int32)
    idx = np.arange(nwg, dtype=np.uint64)
    for t in range(H):
        step = 0xFFFFFFFF if fixed else t
        with np.errstate(over="ignore"):
            inner = fp.mix64(np.uint64(stream_id) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(0x46415354))  # "FAST"
            key = fp.mix64(np.uint64(seed) ^ inner ^ (np.uint64(step) << np.uint64(32)))
            if iid:
                out[t] = (((fp.mix64(key + idx) >> np.uint64(32)) * np.uint64(M)) >> np.uint64(32)).astype(np.int32)
                continue
        pos = fp.permutation_from_key(nwg, key)
        with np.errstate(over="ignore"):
            r = int(((fp.mix64(key ^ np.uint64(0x4F46465345545F52)) >> np.uint64(32)) * np.uint64(nwg * M)) >> np.uint64(32))  # the step's rotation in [0, nwg * M)
        out[t] = (((pos * M + r) // nwg) % M).astype(np.int32)
    return out


def fast_workgroups(B: int, row_tiles: int) -> int:
    """Workgroups of a FAST rollout of B = pop * P rows with `row_tiles` 16-row tiles each (hipets_fast_geometry)."""
    return -(-(-(-B // 16)) // row_tiles)


def fast_row_workgroup(rows, P: int, row_tiles: int):
    """Workgroup that owns each batch row of a FAST rollout; `rows` = arange(B), row = candidate * P + particle.  The rows form one
    particle-major run -- run index particle * pop + candidate -- dealt 16 * row_tiles at a time (rollout.hpp, prologue;
    include/hipets.h hipets_fast_schedule)."""
    pop = len(rows) // P
    return ((rows % P) * pop + rows // P) // (16 * row_tiles)
