// planet_types.hpp -- host-visible descriptors of the PlaNet latent rollout kernel (planet.hpp).
#pragma once
#include "rollout.hpp"

namespace hipets {

constexpr int kPlanetOps = 8;
enum { PL_EMBED = 0, PL_GI = 1, PL_GH = 2, PL_PRIOR1 = 3, PL_PRIOR2 = 4, PL_REW1 = 5, PL_REW2 = 6, PL_REW3 = 7 };

// One linear op of the step: packed weights / biases (LayerMeta), where it reads and writes inside a row, whether a
// ReLU follows, and what happens after it (barrier / elementwise phase).
enum { PL_POST_NONE = 0, PL_POST_SYNC = 1, PL_POST_GRU = 2, PL_POST_SAMPLE = 3, PL_POST_REWARD = 4 };
struct PlanetOp {
    LayerMeta lm;
    int in_off, out_off;
    int relu;
    int post;
};

struct PlanetDev {
    int latent, action, belief, hidden;
    float min_std;
    int ld;                          // LDS row stride in floats (== 8 mod 64)
    int segA, segB, segC, segD, segE;  // segment offsets inside a row (multiples of 16)
    int widA, widE;                  // padded widths of the GEMM-input segments written elementwise
    const float* w;                  // packed weight fragments of the 8 ops
    const float* b;                  // padded biases
    const PlanetOp* ops;             // DEVICE [kPlanetOps], in execution order (a table in memory, staged in LDS: one
                                     // shared copy of the GEMM code instead of eight inlined ones)
};

struct PlanetArgs {
    int pop, P, H, B;
    const float* actions;  // [pop,H,A]
    const float* latent0;  // DEVICE [latent]
    const float* belief0;  // DEVICE [belief]
    float* totals;         // [B]
    const float* eps;      // [H,B,latent] or null
    int use_philox;
    unsigned long long seed, stream_id;
    float* trace_latent;   // optional [H,B,latent]
    float* trace_belief;   // optional [H,B,belief]
    float* trace_rewards;  // optional [H,B]
};

__host__ __device__ inline size_t planet_smem_bytes(int ld) {
    return (size_t)kTile * ld * 4 + 2 * kTile * 4 + sizeof(PlanetOp) * kPlanetOps;
}

}  // namespace hipets
