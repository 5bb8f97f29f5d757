"""Test infrastructure: numpy restatements of the draws the rollout kernels make IN-KERNEL, so that what the engine exports on a
GPU (hipets_fast_normals, hipets_fast_schedule; hipets_device_perms has oracle/feistel_perm.py) can be re-derived where there is no
GPU -- tests/test_oracle_memo_pinned.py replays the memoised FAST / DEVICE entries of tests/golden/oracle_cache/ with them.

Restated from mbrl-lib_amd/csrc/common.hpp (philox4x32_10, u01, box_muller, mix64), csrc/rollout.hpp (rollout_normals4: counter =
(row, step, block of four output dims, stream), key = seed) and csrc/rollout_helpers.hpp (member_schedule_kernel,
export_normals_kernel).  The integer parts (Philox, the schedule) are exact; Box-Muller runs on the CPU's float32 log2 / sqrt /
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


def fast_normals(H: int, B: int, out_dim: int, seed: int, stream_id: int) -> np.ndarray:
    """[H, B, out_dim] float32: what hipets_fast_normals exports (eps of row `rid` at step `t`, output dim `d`)."""
    nblk = (out_dim + 3) // 4
    t, rid, blk = np.meshgrid(np.arange(H, dtype=np.uint64), np.arange(B, dtype=np.uint64), np.arange(nblk, dtype=np.uint64), indexing="ij")
    k0 = seed & 0xFFFFFFFF
    k1 = ((seed >> 32) ^ (stream_id >> 32)) & 0xFFFFFFFF
    x, y, z, w = philox4x32_10(rid, t, blk, np.uint64(stream_id & 0xFFFFFFFF), k0, k1)
    n0, n1 = _box_muller(x, y)
    n2, n3 = _box_muller(z, w)
    full = np.stack([n0, n1, n2, n3], axis=-1).reshape(H, B, nblk * 4)
    return np.ascontiguousarray(full[:, :, :out_dim])


def _mix64(z):
    with np.errstate(over="ignore"):
        z = np.asarray(z, dtype=np.uint64) + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def member_schedule(H: int, nwg: int, M: int, seed: int, stream_id: int, fixed: bool = False, iid: bool = False) -> np.ndarray:
    """[H, nwg] int32: what hipets_fast_schedule exports (member slot of workgroup `wg` at step `t`)."""
    out = np.empty((H, nwg), dtype=np.int32)
    idx = np.arange(nwg, dtype=np.uint64)
    for t in range(H):
        tk = np.uint64(0xFFFFFFFF if fixed else t)
        with np.errstate(over="ignore"):
            base = _mix64(np.uint64(seed) ^ _mix64(np.uint64(stream_id) * np.uint64(0x9E3779B97F4A7C15) + tk))
            keys = _mix64(base + idx)
        if iid:
            out[t] = (((keys >> np.uint64(32)) * np.uint64(M)) >> np.uint64(32)).astype(np.int32)
        else:
            rank = np.empty(nwg, dtype=np.int64)
            rank[np.lexsort((np.arange(nwg), keys))] = np.arange(nwg)  # ties broken by index, like the kernel's rank count
            out[t] = ((rank * M) // nwg).astype(np.int32)
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
