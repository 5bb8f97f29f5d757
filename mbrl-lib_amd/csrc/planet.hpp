// planet.hpp -- latent-space rollouts of a PlaNet model as ONE kernel launch.
//
//   ModelEnv.evaluate_action_sequences            mbrl/models/model_env.py:145-191
//    └ PlaNetModel.reset                          mbrl/models/planet.py:656-672  (saved posterior sample / belief, tiled)
//    └ PlaNetModel.sample                         mbrl/models/planet.py:531-581
//       └ BeliefModel.forward                     :83-101   Linear + ReLU, GRUCell
//       └ prior_transition_model + MeanStdCat     :229-234, :104-115
//       └ _sample_state_from_params               :288-306
//       └ reward_model                            :260-266
//
// A workgroup owns 16 rows (candidate x particle) for the whole horizon.  All activations of a row live in one LDS
// row of `ld` floats, cut into five segments that the eight linear ops ping-pong through:
//
//   A [latent | action]            B emb / prior hidden / reward hidden 1      C GRU input gates / reward hidden 2
//   D GRU hidden gates / prior params / reward        E [belief | latent]  (the recurrent state)
//
// The linear ops are the rollout kernel's wave_gemm (fp32 MFMA 16x16x4, weights pre-packed as fragments); segments that
// feed a GEMM are kept in its chunk-transposed column order (lds_col), segments read elementwise in natural order.
#pragma once
#include "planet_types.hpp"
#include "rollout.hpp"

namespace hipets {

__device__ __forceinline__ float sigmoid_hw(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.44269504088896340736f)); }
// tanh(x) = 2 sigmoid(2x) - 1 (hardware exp2 / rcp, ~1e-7 absolute)
__device__ __forceinline__ float tanh_hw(float x) { return 2.0f * sigmoid_hw(2.0f * x) - 1.0f; }

template <bool STATIC>
__global__ __launch_bounds__(kThreads) void planet_rollout_kernel(const PlanetDev pd, const PlanetArgs ra) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* rows = reinterpret_cast<float*>(smem_raw);           // [16][ld]
    float* tot = rows + (size_t)kTile * pd.ld;                  // [16]
    int* rowid = reinterpret_cast<int*>(tot + kTile);           // [16]
    PlanetOp* ops = reinterpret_cast<PlanetOp*>(rowid + kTile);  // [kPlanetOps]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ld = pd.ld;
    // phase profiler (rollout.hpp Prof; the marks inside the linear ops are wave_gemm's own): profiling builds only, workgroup 0.
    // (Its accumulators sit at the end of the DYNAMIC LDS: the kernel opts in to the full 160 KB of dynamic LDS, and a static array on
    // top of that made the launch fail with "invalid argument".)
    long long* const prof_slots = reinterpret_cast<long long*>(smem_raw + planet_smem_bytes(pd.ld) - kWaves * 16 * sizeof(long long));
    Prof prof;
    prof.on = HIPETS_LEAN_PROF && ra.phase_cycles != nullptr && blockIdx.x == 0 && lane == 0;
    prof.slot = prof_slots + wave * 16;
    if (prof.on) {
#pragma unroll
        for (int i = 0; i < 16; ++i) prof.slot[i] = 0;
    }
    prof.t = prof.on ? clock64() : 0;

    if (tid < kPlanetOps) ops[tid] = pd.ops[tid];
    if (tid < kTile) {
        const int rid = blockIdx.x * kTile + tid;
        rowid[tid] = rid < ra.B ? rid : -1;
        tot[tid] = 0.f;
    }
    // recurrent state E = [belief | latent] and the latent half of A, in GEMM column order; padding columns are zeroed
    // once (their weights are zero, but 0 * garbage could be NaN) and never written again
    for (int i = tid; i < kTile * pd.widE; i += kThreads) {
        const int s = i / pd.widE, c = i % pd.widE;
        float v = 0.f;
        if (c < pd.belief) v = ra.belief0[c];
        else if (c < pd.belief + pd.latent) v = ra.latent0[c - pd.belief];
        rows[s * ld + pd.segE + lds_col(c)] = v;
    }
    __syncthreads();
    auto load_input = [&](const int t) __attribute__((always_inline)) {  // A = [latent | action(t)] (model_env.py:179-182)
        for (int i = tid; i < kTile * pd.widA; i += kThreads) {
            const int s = i / pd.widA, c = i % pd.widA;
            const int rid = rowid[s];
            float v = 0.f;
            if (c < pd.latent) v = rows[s * ld + pd.segE + lds_col(pd.belief + c)];
            else if (c < pd.latent + pd.action && rid >= 0)
                v = ra.actions[((size_t)(rid / ra.P) * ra.H + t) * pd.action + (c - pd.latent)];
            rows[s * ld + pd.segA + lds_col(c)] = v;
        }
    };
    load_input(0);
    __syncthreads();
    prof.mark(0);

    float* Cs = rows + pd.segC;
    float* Ds = rows + pd.segD;
    float* E = rows + pd.segE;
    const int nblk = (pd.latent + 3) / 4;

    // GRUCell: r = s(gi_r + gh_r), z = s(gi_z + gh_z), n = tanh(gi_n + r * gh_n), h' = (1 - z) n + z h     planet.py:98-100
    auto gru = [&](const int t) __attribute__((always_inline)) {
        for (int i = tid; i < kTile * pd.belief; i += kThreads) {
            const int s = i / pd.belief, j = i % pd.belief;
            const float* gi = Cs + s * ld;
            const float* gh = Ds + s * ld;
            const float r = sigmoid_hw(gi[j] + gh[j]);
            const float z = sigmoid_hw(gi[pd.belief + j] + gh[pd.belief + j]);
            const float n = tanh_hw(gi[2 * pd.belief + j] + r * gh[2 * pd.belief + j]);
            float* hp = E + s * ld + lds_col(j);
            const float hn = (1.0f - z) * n + z * (*hp);
            *hp = hn;
            const int rid = rowid[s];
            if (ra.trace_belief && rid >= 0) ra.trace_belief[((size_t)t * ra.B + rid) * pd.belief + j] = hn;
        }
    };
    // s_t = mean + (softplus(raw) + min_std) * eps                                             :111-115, :288-306
    auto sample = [&](const int t) __attribute__((always_inline)) {
        for (int item = tid; item < kTile * nblk; item += kThreads) {
            const int s = item / nblk, blk = item % nblk;
            const int rid = rowid[s];
            float nrm[4] = {0.f, 0.f, 0.f, 0.f};
            if (rid >= 0) {
                if (ra.eps) {
                    for (int q = 0; q < 4; ++q) {
                        const int d = min(blk * 4 + q, pd.latent - 1);
                        nrm[q] = ra.eps[((size_t)t * ra.B + rid) * pd.latent + d];
                    }
                } else if (ra.use_philox) {
                    rollout_normals4(rid, t, blk, ra.seed, ra.stream_id, nrm);
                }
            }
            for (int q = 0; q < 4; ++q) {
                const int d = blk * 4 + q;
                if (d >= pd.latent) break;
                const float mean = Ds[s * ld + d];
                const float std_ = softplus_fast(Ds[s * ld + pd.latent + d]) + pd.min_std;
                const float v = mean + std_ * nrm[q];
                E[s * ld + lds_col(pd.belief + d)] = v;
                if (ra.trace_latent && rid >= 0) ra.trace_latent[((size_t)t * ra.B + rid) * pd.latent + d] = v;
            }
        }
    };

    for (int t = 0; t < ra.H; ++t) {
        // embed (A -> B, ReLU) | hidden gates (E -> D) | input gates (B -> C) | GRU | prior (E -> B -> D) | sample |
        // reward head (E -> B -> C -> D) | accumulate, next input
        for (int i = 0; i < kPlanetOps; ++i) {
            const PlanetOp op = ops[i];
            prof.mark(12);
            if constexpr (STATIC) {
                // one shape-specialised instantiation per DISTINCT (column tiles, k chunks) pair: six for the eight ops
                using PS = PlanetConfShape;
                auto run = [&](auto cs, auto kcs) __attribute__((always_inline)) {
                    linear_op<1, HIPETS_ACT_RELU, decltype(cs)::value, false, NoTail, PS::LD, false, decltype(kcs)::value, 4>(
                        pd.w + op.lm.woff, pd.b + op.lm.boff, op.lm, PS::LD, op.relu != 0, HIPETS_ACT_RELU, 0.f, rows + op.in_off, rows + op.out_off, wave, lane, prof);
                };
                using std::integral_constant;
                switch (i) {  // (wave-uniform)
                    case 0: run(integral_constant<int, PS::kC[0]>{}, integral_constant<int, PS::kKC[0]>{}); break;
                    case 1: case 2: run(integral_constant<int, PS::kC[1]>{}, integral_constant<int, PS::kKC[1]>{}); break;
                    case 3: case 6: run(integral_constant<int, PS::kC[3]>{}, integral_constant<int, PS::kKC[3]>{}); break;
                    case 4: run(integral_constant<int, PS::kC[4]>{}, integral_constant<int, PS::kKC[4]>{}); break;
                    case 5: run(integral_constant<int, PS::kC[5]>{}, integral_constant<int, PS::kKC[5]>{}); break;
                    default: run(integral_constant<int, PS::kC[7]>{}, integral_constant<int, PS::kKC[7]>{}); break;
                }
            } else {
                linear_op<1, HIPETS_ACT_RELU>(pd.w + op.lm.woff, pd.b + op.lm.boff, op.lm, ld, op.relu != 0, HIPETS_ACT_RELU, 0.f, rows + op.in_off,
                             rows + op.out_off, wave, lane, prof);
            }
            if (op.post == PL_POST_NONE) continue;  // the next op touches other segments: same barrier interval
            __syncthreads();
            prof.mark(8);
            if (op.post == PL_POST_GRU) {
                gru(t);
                __syncthreads();
                prof.mark(9);
            } else if (op.post == PL_POST_SAMPLE) {
                sample(t);
                __syncthreads();
                prof.mark(9);
            } else if (op.post == PL_POST_REWARD) {
                if (tid < kTile) {  // model_env.py:186-188 with no_termination
                    const float r = Ds[tid * ld];
                    tot[tid] += r;
                    const int rid = rowid[tid];
                    if (ra.trace_rewards && rid >= 0) ra.trace_rewards[(size_t)t * ra.B + rid] = r;
                }
                if (t + 1 < ra.H) load_input(t + 1);
                __syncthreads();
                prof.mark(10);
            }
        }
    }
    if (prof.on) {
#pragma unroll
        for (int i = 0; i < 16; ++i) ra.phase_cycles[wave * 16 + i] += prof.slot[i];
    }
    if (tid < kTile && rowid[tid] >= 0) ra.totals[rowid[tid]] = tot[tid];
}

}  // namespace hipets
