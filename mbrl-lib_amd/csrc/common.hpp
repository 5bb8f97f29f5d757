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
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
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

// 64-bit mix (splitmix64 finaliser) for sort keys of the balanced member schedule
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

}  // namespace hipets
