// common.hpp -- device helpers shared by the rollout and optimizer kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/hipets.h"

namespace hipets {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int kWave = 64;          // CDNA wavefront
#ifndef HIPETS_WAVES
#define HIPETS_WAVES 4
#endif
constexpr int kWaves = HIPETS_WAVES;  // waves per workgroup: 4 = one per SIMD (measured best on cfg2: 1.27 ms/rollout);
                                      // 8 = two per SIMD builds and passes parity but measured 1.31 ms (VALU phases
                                      // are shared by the SIMD partners and the kernel is capped at 256 VGPRs)
constexpr int kThreads = kWaves * kWave;
constexpr int kMaxExtras = 4;         // max leftover (column tile, row tile) units per wave
constexpr int kTile = 16;          // rows / cols of one v_mfma_f32_16x16x4_f32 tile
constexpr int kKChunk = 16;        // k extent of one packed B fragment (4 MFMA k-steps of 4)

// ---------------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al. 2011): counter-based, so every (row, step, dim) owns its stream and
// results do not depend on launch geometry.
// ---------------------------------------------------------------------------------------------
struct Philox4 { uint32_t x, y, z, w; };

__device__ __forceinline__ Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                 uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // one 32 x 32 -> 64 product each (v_mad_u64_u32: high and low half from ONE quarter-rate instruction; written as __umulhi and
        // `*` the compiler issued v_mul_hi_u32 AND v_mul_lo_u32 -- 40 quarter-rate multiplies per call instead of 20, and the 10 rounds
        // are half of a fused tail unit's VALU time)
#ifndef HIPETS_PHILOX_MAD64
#define HIPETS_PHILOX_MAD64 1
#endif
#if HIPETS_PHILOX_MAD64
        const uint64_t p0 = (uint64_t)0xD2511F53u * (uint64_t)c0, p1 = (uint64_t)0xCD9E8D57u * (uint64_t)c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
#else
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
#endif
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return Philox4{c0, c1, c2, c3};
}

__device__ __forceinline__ float u01(uint32_t x) {  // (0,1), 24 random bits
    return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f);
}

// two uniforms -> two standard normals (Box-Muller)
__device__ __forceinline__ void box_muller(uint32_t a, uint32_t b, float& n0, float& n1) {
    // r = sqrt(-2 ln u1) = sqrt(-2 ln2 * log2 u1); v_sin/v_cos take their argument in revolutions
    const float r = __builtin_amdgcn_sqrtf(-1.38629436111989061883f * __builtin_amdgcn_logf(u01(a)));
    const float u = u01(b);
    n0 = r * __builtin_amdgcn_cosf(u);
    n1 = r * __builtin_amdgcn_sinf(u);
}

// ---------------------------------------------------------------------------------------------
// Counter-based random permutation of [0, n): the device-side stand-in for the reference's per-step
// torch.randperm(B) (mbrl/models/gaussian_mlp.py:203-205).  A mixed-radix alternating Feistel network on
// [0, a) x [0, b) (a * b >= n, a ~ b ~ sqrt(n)) with cycle walking: every round adds a keyed hash of one half
// to the other half modulo its radix, which is invertible whatever the hash, so x -> perm(x) is a bijection
// of [0, a*b); re-applying it until the image falls below n restricts it to a bijection of [0, n).  O(1) per
// element, no sort, no memory: every workgroup computes the rows it owns.  Balance (each member gets exactly
// n / M rows) is exact by construction; uniformity is checked statistically (tests/test_perm_feistel.py).
// ---------------------------------------------------------------------------------------------
constexpr int kPermRounds = 6;

__host__ __device__ inline void perm_radices(uint32_t n, uint32_t* a, uint32_t* b) {
    uint32_t r = 1;
    while ((uint64_t)r * r < n) ++r;  // ceil(sqrt(n)); n < 2^31 so r < 46342
    *a = r;
    *b = (n + r - 1) / r;
    if (*b < 1) *b = 1;
}

__host__ __device__ inline uint32_t perm_hash(uint32_t v, uint32_t k) {  // murmur3 finaliser of a keyed multiply
    uint32_t h = v * 0x9E3779B1u + k;
    h ^= h >> 16; h *= 0x85EBCA6Bu;
    h ^= h >> 13; h *= 0xC2B2AE35u;
    h ^= h >> 16;
    return h;
}

// 64-bit mix (splitmix64 finaliser), host + device
__host__ __device__ inline uint64_t perm_mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// key of the permutation of (seed, stream, step); step = 0xFFFFFFFF for the one permutation of a TS-infinity rollout
__host__ __device__ inline uint64_t perm_key(uint64_t seed, uint64_t stream, uint32_t step) {
    return perm_mix64(seed ^ perm_mix64(stream * 0x9E3779B97F4A7C15ull + 0x5045524Dull /* "PERM" */) ^ ((uint64_t)step << 32));
}

// the kPermRounds round keys of a permutation (wave-uniform: computed on the host per launch, or once per thread)
struct PermKeys { uint32_t k[kPermRounds]; };
__host__ __device__ inline PermKeys perm_round_keys(uint64_t key) {
    PermKeys pk;
#pragma unroll
    for (int r = 0; r < kPermRounds; ++r) pk.k[r] = (uint32_t)(perm_mix64(key + (uint64_t)r) >> 16);
    return pk;
}

// uniform map of a 32-bit hash onto [0, m): floor(h * m / 2^32) (one multiply-high instead of a division)
__host__ __device__ inline uint32_t perm_scale(uint32_t h, uint32_t m) { return (uint32_t)(((uint64_t)h * m) >> 32); }

__host__ __device__ inline uint32_t perm_apply(uint32_t x, uint32_t n, uint32_t a, uint32_t b, const PermKeys& pk) {
    do {
        uint32_t L = x / b, R = x - L * b;  // x = L * b + R, L in [0, a), R in [0, b)
#pragma unroll
        for (int r = 0; r < kPermRounds; ++r) {
            if (r & 1) { R += perm_scale(perm_hash(L, pk.k[r]), b); if (R >= b) R -= b; }  // R, increment < b < 2^16: no overflow
            else { L += perm_scale(perm_hash(R, pk.k[r]), a); if (L >= a) L -= a; }
        }
        x = L * b + R;
    } while (x >= n);
    return x;
}

// ---------------------------------------------------------------------------------------------
// FAST mode: member slot of workgroup `wg` (of `nwg`) at `step` -- the block-balanced stand-in for the reference's per-row
// balanced shuffle (mbrl/models/gaussian_mlp.py:203-205, 267-275) at workgroup granularity.  The step's keyed bijection of
// [0, nwg) (the same Feistel network DEVICE mode applies to rows) is cut into M equal runs (position p -> slot, with a per-step
// rotation): every slot gets floor / ceil(nwg / M) workgroups, exactly, and every workgroup meets every slot with probability 1 / M.  O(1) per workgroup: every workgroup evaluates its own entry
// in its prologue (until round 6 a separate kernel ranked nwg sort keys, O(nwg^2): 80 us in front of a ModelEnv.step of
// 100 000 rows).  (a, b) = perm_radices(nwg).  step = 0xFFFFFFFF: the one draw of a TS-infinity (fixed_model) rollout.
// BasicEnsemble (iid != 0): every workgroup draws independently and uniformly (randint, basic_ensemble.py:122-129).
// hipets_fast_schedule exports exactly these integers; oracle/device_draws.member_schedule restates them.
// ---------------------------------------------------------------------------------------------
__host__ __device__ inline uint64_t fast_member_key(uint64_t seed, uint64_t stream, uint32_t step) {
    return perm_mix64(seed ^ perm_mix64(stream * 0x9E3779B97F4A7C15ull + 0x46415354ull /* "FAST" */) ^ ((uint64_t)step << 32));
}
__host__ __device__ inline int fast_member(uint32_t wg, uint32_t nwg, uint32_t a, uint32_t b, int M, int iid, uint64_t seed, uint64_t stream,
                                           uint32_t step) {
    const uint64_t key = fast_member_key(seed, stream, step);
    if (iid) return (int)(((perm_mix64(key + (uint64_t)wg) >> 32) * (uint64_t)M) >> 32);
    const uint32_t p = perm_apply(wg, nwg, a, b, perm_round_keys(key));
    // positions p * M + r on a circle of nwg * M points cut into M arcs of nwg: floor / ceil(nwg / M) workgroups per slot whatever the
    // step's rotation r in [0, nwg * M) -- which makes every slot equally likely for every workgroup (it decides WHICH slots get the
    // extra workgroup and, for nwg < M, which slots are used at all: without it three workgroups of a five-member model would only
    // ever run members 0, 1 and 3)
    const uint32_t r = perm_scale((uint32_t)(perm_mix64(key ^ 0x4F46465345545F52ull /* "OFFSET_R" */) >> 32), nwg * (uint32_t)M);
    return (int)((((uint64_t)p * (uint64_t)M + r) / nwg) % (uint64_t)M);
}

}  // namespace hipets
