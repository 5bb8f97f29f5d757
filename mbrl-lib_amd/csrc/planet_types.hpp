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
    long long* phase_cycles;  // optional [kWaves][16] phase accumulators of workgroup 0 (rollout.hpp Prof; -DHIPETS_LEAN_PROF=1 builds only)
};

// The shapes of conf/dynamics_model/planet.yaml (latent 30, belief 200, hidden 200, action 6 -- every planet_*.yaml override the
// reference ships plans on them): per op (column tiles, k chunks) in execution order and the LDS row stride, as compile-time facts of
// the STATIC instance (round 5).  Its ops run the shape-specialised one-tile path of the rollout kernel: per-wave share through one
// branch, A-fragment reads with immediate offsets, fragment loads interleaved with the MFMAs and fetched two chunks ahead
// (wave_gemm kTriple) -- the generic instance dispatches every op at run time through one shared copy of the GEMM code.
// Same arithmetic per column (same k order, same tile -> wave deal): the two instances return the same bits.
struct PlanetConfShape {
    static constexpr int LD = 1736;
    static constexpr int kC[kPlanetOps] = {13, 38, 38, 13, 4, 13, 13, 1};    // embed, hidden gates, input gates, prior 1 / 2, reward 1 / 2 / 3
    static constexpr int kKC[kPlanetOps] = {3, 13, 13, 13, 13, 15, 13, 13};
};
inline bool planet_static_shape(const PlanetDev& pd, const PlanetOp* host_ops) {
    if (pd.ld != PlanetConfShape::LD) return false;
    for (int i = 0; i < kPlanetOps; ++i)
        if (host_ops[i].lm.Np / kTile != PlanetConfShape::kC[i] || host_ops[i].lm.Kp / kKChunk != PlanetConfShape::kKC[i]) return false;
    return true;
}

__host__ __device__ inline size_t planet_smem_bytes(int ld) {
    const size_t n = (size_t)kTile * ld * 4 + 2 * kTile * 4 + sizeof(PlanetOp) * kPlanetOps;
    return (n + 7) / 8 * 8 + kWaves * 16 * sizeof(long long);  // ... + the phase profiler's accumulators (profiling builds; 512 bytes)
}

}  // namespace hipets
