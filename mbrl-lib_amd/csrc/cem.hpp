// cem.hpp -- population sampling and elite refit kernels for the trajectory optimizers.
//
// Replaces the torch op sequences of CEMOptimizer._sample_population / _update_population_params /
// optimize (mbrl/planning/trajectory_opt.py:110-188) and mbrl.util.math.truncated_normal_
// (mbrl/util/math.py:69-92, a host-synchronising rejection loop in the reference).
#pragma once
#include <type_traits>

#include "common.hpp"

namespace hipets {

struct CemDev {
    int n_env;         // independent optimisation problems batched in one launch (mu / dispersion / best per env)
    int pop, H, A, D;  // per-env population; D = H*A
    int K;             // elite_num
    float alpha, one_minus_alpha;
    int return_mean, clipped, unbiased;
    // refit only: when set, values[i] is first formed as the mean over the P particle totals of candidate i (model_env.py:190-191;
    // the same sequential sum and division as particle_mean_kernel, so the same bits) -- the fused plan's rollouts leave their
    // per-row totals and this kernel reduces them, instead of a kernel launch of its own in between
    const float* totals = nullptr;
    int P = 0;
    // refit only: when set, the caller has chosen the elites (DEVICE [n_env][K] candidate indices, best first) and the kernel's own
    // selection is skipped.  For the reference-order parity mode of the optimizer classes (sampler='torch'): torch.topk's order
    // among EQUAL values is whatever its partial sort leaves (trajectory_opt.py:179), and the 0 / 1 rewards of the cartpole family
    // tie dozens of candidates at the elite boundary -- no closed-form tie rule reproduces that, the host's own topk does.
    const int* elite_in = nullptr;
};

// Standard normal truncated to [-2, 2] by rejection: the stationary law of the reference's
// redraw-until-inside loop (util/math.py:83-91).  Counter = (element, attempt).
__device__ __forceinline__ float philox_trunc_normal(uint32_t idx_lo, uint32_t idx_hi, uint64_t seed, uint64_t stream) {
    for (uint32_t attempt = 0; attempt < 64; ++attempt) {
        const Philox4 r = philox4x32_10(idx_lo, idx_hi, attempt, (uint32_t)stream, (uint32_t)seed,
                                        (uint32_t)(seed >> 32) ^ (uint32_t)(stream >> 32) ^ 0x5EED5EEDu);
        float n0, n1, n2, n3;
        box_muller(r.x, r.y, n0, n1);
        if (fabsf(n0) <= 2.0f) return n0;
        if (fabsf(n1) <= 2.0f) return n1;
        box_muller(r.z, r.w, n2, n3);
        if (fabsf(n2) <= 2.0f) return n2;
        if (fabsf(n3) <= 2.0f) return n3;
    }
    return 0.0f;  // P(reached) = 0.0455^256
}

__device__ __forceinline__ float philox_normal(uint32_t idx_lo, uint32_t idx_hi, uint64_t seed, uint64_t stream) {
    const Philox4 r = philox4x32_10(idx_lo, idx_hi, 0u, (uint32_t)stream, (uint32_t)seed,
                                    (uint32_t)(seed >> 32) ^ (uint32_t)(stream >> 32) ^ 0x5EED5EEDu);
    float n0, n1;
    box_muller(r.x, r.y, n0, n1);
    return n0;
}

// trajectory_opt.py:110-128
__global__ void cem_sample_kernel(const CemDev p, const float* __restrict__ mu, const float* __restrict__ disp,
                                  const float* __restrict__ lower, const float* __restrict__ upper,
                                  const float* __restrict__ z_in, unsigned long long seed, unsigned long long stream,
                                  float* __restrict__ population) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)p.n_env * p.pop * p.D) return;
    const int d = (int)(i % p.D);
    const int env = (int)(i / ((long long)p.pop * p.D));
    const float m = mu[env * p.D + d], v = disp[env * p.D + d], lb = lower[d], ub = upper[d];
    float x;
    if (p.clipped) {  // :116-120
        const float z = z_in ? z_in[i] : philox_normal((uint32_t)i, (uint32_t)(i >> 32), seed, stream);
        x = m + v * z;
        x = x > lb ? x : lb;
        x = x < ub ? x : ub;
    } else {  // :122-128
        const float lbd = (m - lb) / 2.0f, ubd = (ub - m) / 2.0f;
        const float mv = fminf(lbd * lbd, ubd * ubd);
        const float cv = fminf(mv, v);
        const float z = z_in ? z_in[i] : philox_trunc_normal((uint32_t)i, (uint32_t)(i >> 32), seed, stream);
        x = z * sqrtf(cv) + m;
    }
    population[i] = x;
}

// NaN filter + top-k + refit + best-so-far, the selection in ONE workgroup (pop <= kMaxPop): the K best values are found in LDS
// (descending, ties broken by lower index: rank counting for small populations, a radix select + a sort of the elites for
// larger ones, a full bitonic network when K > kSelectMaxK), then threads share the dimensions d of the [H,A] plan and
// reduce the K elites in f64.
#ifndef HIPETS_REFIT_SKIP
#define HIPETS_REFIT_SKIP 0  // TIMING-ONLY builds (results are wrong on purpose; profiles/r6_refit_phases.json): bit 0 = no particle means, bit 1 = no selection, bit 2 = no mean / variance sweeps
#endif
constexpr int kRefitThreads = 1024;
constexpr int kMaxPop = 8192;
constexpr int kRefitScratchBytes = 12288;  // LDS behind the key / index arrays: f64 partial sums, or the selection's work space
constexpr int kSelectMaxK = 1024;          // radix top-k below handles elite counts up to here (else: full bitonic sort)
constexpr int kRankSortMax = 640;   // populations up to here: O(n^2 / threads) rank counting (14 us at pop 500); above, the
                                    // radix select (22 / 36 / 64 / 120 us at pop 1036 / 2000 / 4000 / 8000; the bitonic
                                    // network it replaced took 45 / 52 / 105 us at 1036 / 2000 / 4000)

__global__ __launch_bounds__(kRefitThreads) void cem_refit_kernel(const CemDev p, float* values, const float* population, float* mu,
                                                                 float* disp, float* best_value, float* best_solution,
                                                                 int* elite_idx_out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    {  // gridDim.y environments (gridDim.x workgroups each): rebase every pointer to this environment's slice
        const int env = blockIdx.y;
        values += (size_t)env * p.pop;
        if (p.totals && !(HIPETS_REFIT_SKIP & 1)) {
            const float* tot = p.totals + (size_t)env * p.pop * p.P;
            for (int i = threadIdx.x; i < p.pop; i += kRefitThreads) {
                float s = 0.f;  // the same sequential sum; a candidate's particle totals fetched eight at a time
                for (int q0 = 0; q0 < p.P; q0 += 8) {
                    float tv[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) tv[u] = tot[(size_t)i * p.P + min(q0 + u, p.P - 1)];
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        if (q0 + u < p.P) s += tv[u];
                }
                values[i] = s / (float)p.P;
            }
        }
        population += (size_t)env * p.pop * p.D;
        mu += (size_t)env * p.D;
        disp += (size_t)env * p.D;
        best_value += env;
        best_solution += (size_t)env * p.D;
        if (elite_idx_out) elite_idx_out += (size_t)env * p.K;
    }
    const int* const elite_in = p.elite_in ? p.elite_in + (size_t)blockIdx.y * p.K : nullptr;
    int n2 = 1;
    while (n2 < p.pop) n2 <<= 1;
    float* key = reinterpret_cast<float*>(smem);
    int* idx = reinterpret_cast<int*>(smem + (size_t)n2 * 4);
    const int tid = threadIdx.x;
    for (int i = tid; i < n2; i += kRefitThreads) {
        float v = -INFINITY;
        if (i < p.pop) {
            v = values[i];
            if (v != v) { v = -1e-10f; values[i] = v; }  // trajectory_opt.py:178
        }
        key[i] = v;
        idx[i] = i < p.pop ? i : 0x7FFFFFFF;
    }
    __syncthreads();
    // order: "a before b" iff (key_a > key_b) or (equal and idx_a < idx_b)
    if (elite_in) {  // the caller's elites, in the caller's order
        // (indices are clamped into [0, pop): the Python wrapper rejects out-of-range ones with a ValueError -- Engine.cem_refit -- and a
        // raw-ABI caller that passes garbage gets a wrong refit, never an out-of-bounds LDS / global read)
        auto in_range = [&](const int e) { return e < 0 ? 0 : (e >= p.pop ? p.pop - 1 : e); };
        const float top_key = key[in_range(elite_in[0])];
        __syncthreads();
        for (int k = tid; k < p.K; k += kRefitThreads) idx[k] = in_range(elite_in[k]);
        if (tid == 0) key[0] = top_key;  // key[] is only consulted at position 0 from here on
        __syncthreads();
    } else if (HIPETS_REFIT_SKIP & 2) {  // (timing-only: no selection, the identity order)
    } else if (p.pop <= kRankSortMax) {
        // small populations (every PETS config): rank by counting -- element i sits at position #{j before i}.  One pass of
        // pop LDS broadcast reads per element and a single barrier instead of the ~log^2(n)/2 barrier stages of the
        // network below (45 for pop 500); only the first K positions (the elites) and position 0 (the best) are consumed.
        for (int i = tid; i < p.pop; i += kRefitThreads) {
            const float ki = key[i];
            int rank = 0;  // one compare per key: ties go to the lower index, so keys before i count with >=, after i with >
#pragma unroll 8
            for (int j = 0; j < i; ++j) rank += key[j] >= ki;
#pragma unroll 8
            for (int j = i + 1; j < p.pop; ++j) rank += key[j] > ki;
            idx[i] = rank;  // idx[] holds ranks for now (it held the identity so far)
        }
        __syncthreads();
        // scatter: sorted position -> element; ranks are a permutation of 0..pop-1
        int* pos = reinterpret_cast<int*>(smem + (size_t)n2 * 8);  // the scratch region (free until the barrier below)
        for (int i = tid; i < p.pop; i += kRefitThreads) pos[idx[i]] = i;
        __syncthreads();
        float top_key = key[pos[0]];
        __syncthreads();
        for (int i = tid; i < p.pop; i += kRefitThreads) idx[i] = pos[i];
        if (tid == 0) key[0] = top_key;  // key[] is only consulted at position 0 from here on
        __syncthreads();
    } else if (p.K <= kSelectMaxK) {
        // larger populations (sharded multi-GPU plans, iCEM's first iterations): only the K best are needed, so select
        // instead of sorting.  Four 8-bit radix passes over order-preserving integer keys find the K-th largest key T
        // (LDS histogram + a 256-bin scan per pass); the elites are {key > T} plus the lowest-index elements with
        // key == T; they are then ordered among themselves by rank counting (K^2 / threads compares).
        int* hist = reinterpret_cast<int*>(smem + (size_t)n2 * 8);  // [256]
        int* ctl = hist + 256;                                      // prefix, mask, still needed, elite counter
        int* elist = ctl + 8;                                       // [K] element ids, unordered
        float* ekey = reinterpret_cast<float*>(elist + kSelectMaxK);  // [K] their keys
        auto okey = [](float v) {  // monotone increasing in v
            const unsigned u = __float_as_uint(v + 0.0f);  // -0.0 -> +0.0: equal floats get equal keys
            return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
        };
        if (tid == 0) { ctl[0] = 0; ctl[1] = 0; ctl[2] = p.K; ctl[3] = 0; }
        for (int pass = 3; pass >= 0; --pass) {
            for (int b = tid; b < 256; b += kRefitThreads) hist[b] = 0;
            __syncthreads();
            const unsigned prefix = (unsigned)ctl[0], mask = (unsigned)ctl[1];
            for (int i = tid; i < p.pop; i += kRefitThreads) {
                const unsigned u = okey(key[i]);
                if ((u & mask) == prefix) atomicAdd(&hist[(u >> (8 * pass)) & 255u], 1);
            }
            __syncthreads();
            if (tid < 64) {  // wave 0: the bin holding the (still needed)-th largest candidate.  Lane l owns bins 4l .. 4l+3.
                const int h0 = hist[4 * tid], h1 = hist[4 * tid + 1], h2 = hist[4 * tid + 2], h3 = hist[4 * tid + 3];
                const int mine = h0 + h1 + h2 + h3;
                int above = mine;  // inclusive suffix sum over lanes (bins above and including mine)
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const int o = __shfl_down(above, d, 64);
                    if (tid + d < 64) above += o;
                }
                const int need = ctl[2];
                const int higher = above - mine;  // candidates in bins of higher lanes
                if (higher < need && need <= above) {  // exactly one lane: the K-th largest falls into my four bins
                    int n = need - higher, b = 4 * tid + 3;
                    if (n > h3) { n -= h3; b = 4 * tid + 2;
                        if (n > h2) { n -= h2; b = 4 * tid + 1;
                            if (n > h1) { n -= h1; b = 4 * tid; } } }
                    ctl[0] = (int)(prefix | ((unsigned)b << (8 * pass)));
                    ctl[1] = (int)(mask | (255u << (8 * pass)));
                    ctl[2] = n;  // how many elements of this bin (finally: with key == T) are elites
                }
            }
            __syncthreads();
        }
        const unsigned T = (unsigned)ctl[0];
        const int need_eq = ctl[2];
        const bool all_eq = hist[T & 255u] == need_eq;  // the last pass's bin of T counted the elements with key == T
        for (int i = tid; i < p.pop; i += kRefitThreads) {
            const unsigned u = okey(key[i]);
            bool elite = u > T || (u == T && all_eq);
            if (u == T && !all_eq) {  // more ties at the threshold than places: the lowest indices win (as in the full sorts)
                int before = 0;
                for (int j = 0; j < i; ++j) before += okey(key[j]) == T;
                elite = before < need_eq;
            }
            if (elite) {
                const int slot = atomicAdd(&ctl[3], 1);
                elist[slot] = i;
                ekey[slot] = key[i];
            }
        }
        __syncthreads();
        // order the K elites: position = #{elites before me}
        int my_pos = -1, my_id = 0;
        float my_key = 0.f;
        if (tid < p.K) {
            my_id = elist[tid];
            my_key = ekey[tid];
            int rank = 0;
            for (int f = 0; f < p.K; ++f) {
                const float kf = ekey[f];
                rank += (kf > my_key) || (kf == my_key && elist[f] < my_id);
            }
            my_pos = rank;
        }
        __syncthreads();
        if (my_pos >= 0) idx[my_pos] = my_id;
        if (my_pos == 0) key[0] = my_key;  // key[] is only consulted at position 0 from here on
        __syncthreads();
    } else
    for (int k = 2; k <= n2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < n2; i += kRefitThreads) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const float ka = key[i], kb = key[ixj];
                    const int ia = idx[i], ib = idx[ixj];
                    const bool a_first = (ka > kb) || (ka == kb && ia < ib);
                    const bool up = (i & k) == 0;  // descending block
                    if (up ? !a_first : a_first) {
                        key[i] = kb; key[ixj] = ka;
                        idx[i] = ib; idx[ixj] = ia;
                    }
                }
            }
            __syncthreads();
        }
    }
    if (elite_idx_out && blockIdx.x == 0)
        for (int k = tid; k < p.K; k += kRefitThreads) elite_idx_out[k] = idx[k];

    // best-so-far (trajectory_opt.py:184-186)
    const float top = key[0];
    const bool improved = blockIdx.x == 0 && top > best_value[0];
    const int top_i = idx[0];
    __syncthreads();

    // elite mean / variance: `parts` threads share a dimension (each owns every parts-th elite), partial sums meet in LDS; consecutive
    // threads own consecutive dimensions (coalesced rows).  f64 throughout.  Round 6: the D dimensions of the plan are dealt to the
    // gridDim.x workgroups of the environment (each repeats the selection above -- same inputs, same result -- and refits its own slice;
    // workgroup 0 also keeps the best-so-far), so that a slice is at most 128 dimensions wide and eight threads share each: a thread's
    // share of the K elites (cfg2: 7 of 50, cfg4: 13 of 103) is fetched in ONE round trip and held in registers for the variance pass
    // (the single-workgroup version walked a 103-load chain twice per thread for cfg4's 680 dimensions: 20 of its 48 us).
    double* red = reinterpret_cast<double*>(smem + (size_t)n2 * 8);  // [kRefitThreads]
    const int nb = (int)gridDim.x, bx = (int)blockIdx.x;
    const int Dslice = (((p.D + nb - 1) / nb + 63) / 64) * 64;  // dimensions per workgroup (a multiple of the wave width)
    const int d_lo = bx * Dslice, d_hi = min(p.D, d_lo + Dslice);
    int parts = kRefitThreads / Dslice;
    parts = parts < 1 ? 1 : (parts > 8 ? 8 : parts);
    const int G = kRefitThreads / parts;  // dimensions per sweep
    const int part = tid / G, dloc = tid % G;
    const int nu = (p.K + parts - 1) / parts;  // elites per thread (workgroup-uniform)
    // One sweep of G dimensions with NU >= nu elites per thread held in registers (HOLD), or walked NU at a time and fetched again for the
    // variance (more than 16 per thread: K > 128).  Every load of a batch is unconditional, its indices clamped: a guarded load is a
    // branch per element, and the merge behind it waits for the load -- the round trips would go out one after the other.
    auto sweep = [&](const int base, auto nu_tag, auto hold_tag) __attribute__((always_inline)) {
        constexpr int NU = decltype(nu_tag)::value;
        constexpr bool HOLD = decltype(hold_tag)::value;
        const int d = base + dloc;
        const bool live = d < d_hi && part < parts;
        float mu_old = 0.f, disp_old = 0.f;
        float xv[NU];
        double s = 0.0;
        if (live) {
            if (part == 0) { mu_old = mu[d]; disp_old = disp[d]; }  // (in flight beside the elites' rows)
            for (int k0 = part; k0 < (HOLD ? part + 1 : p.K); k0 += parts * NU) {
#pragma unroll
                for (int u = 0; u < NU; ++u) xv[u] = population[(size_t)idx[min(k0 + u * parts, p.K - 1)] * p.D + d];
#pragma unroll
                for (int u = 0; u < NU; ++u)
                    if (k0 + u * parts < p.K) s += (double)xv[u];
            }
        }
        red[tid] = s;
        __syncthreads();
        double mean = 0.0;
        if (live) {
            for (int q = 0; q < parts; ++q) mean += red[q * G + dloc];
            mean /= (double)p.K;
        }
        __syncthreads();
        double ss = 0.0;
        if (live) {
            for (int k0 = part; k0 < (HOLD ? part + 1 : p.K); k0 += parts * NU) {
                if constexpr (!HOLD) {
#pragma unroll
                    for (int u = 0; u < NU; ++u) xv[u] = population[(size_t)idx[min(k0 + u * parts, p.K - 1)] * p.D + d];
                }
#pragma unroll
                for (int u = 0; u < NU; ++u)
                    if (k0 + u * parts < p.K) {
                        const double dv = (double)xv[u] - mean;
                        ss += dv * dv;
                    }
            }
        }
        red[tid] = ss;
        __syncthreads();
        if (live && part == 0) {
            double var = 0.0;
            for (int q = 0; q < parts; ++q) var += red[q * G + dloc];
            var /= (double)(p.unbiased ? (p.K - 1) : p.K);
            const float new_mu = (float)mean;
            const float new_disp = p.clipped ? (float)sqrt(var) : (float)var;  // :134-137
            mu[d] = p.alpha * mu_old + p.one_minus_alpha * new_mu;              // :138
            disp[d] = p.alpha * disp_old + p.one_minus_alpha * new_disp;        // :139
        }
        __syncthreads();
    };
    // the new best plan (workgroup 0, every dimension of it): its row is requested now, beside the elites' rows, and stored behind the sweeps
    constexpr int kBestHold = 4;
    float best_row[kBestHold];
    const bool best_held = p.D <= kBestHold * kRefitThreads;
    if (improved && best_held) {
#pragma unroll
        for (int u = 0; u < kBestHold; ++u) best_row[u] = population[(size_t)top_i * p.D + min(tid + u * kRefitThreads, p.D - 1)];
    }
    for (int base = d_lo; base < ((HIPETS_REFIT_SKIP & 4) ? d_lo : d_hi); base += G) {
        using T = std::true_type;
        using F = std::false_type;
        if (nu <= 2) sweep(base, std::integral_constant<int, 2>{}, T{});
        else if (nu <= 4) sweep(base, std::integral_constant<int, 4>{}, T{});
        else if (nu <= 8) sweep(base, std::integral_constant<int, 8>{}, T{});
        else if (nu <= 16) sweep(base, std::integral_constant<int, 16>{}, T{});
        else sweep(base, std::integral_constant<int, 16>{}, F{});
    }
    // best-so-far (:184-186): workgroup 0 alone reads and writes it (a workgroup that started late must not see the new value)
    if (improved) {  // (workgroup 0 only: see `improved`)
        if (best_held) {
#pragma unroll
            for (int u = 0; u < kBestHold; ++u)
                if (tid + u * kRefitThreads < p.D) best_solution[tid + u * kRefitThreads] = best_row[u];
        } else {
            for (int d = tid; d < p.D; d += kRefitThreads) best_solution[d] = population[(size_t)top_i * p.D + d];
        }
        if (tid == 0) best_value[0] = top;
    }
}

// workgroups per environment of a refit launch (the dimensions of the plan in slices of at most 128)
inline int refit_blocks(const int D) { return D <= 128 ? 1 : (D + 127) / 128 > 8 ? 8 : (D + 127) / 128; }

// trajectory_opt.py:103-108: dispersion0 = ones (clipped) or ((ub - lb)^2) / 16; mu0 = x0
__global__ void cem_init_kernel(const CemDev p, const float* x0, const float* lower, const float* upper, float* mu,
                                float* disp, float* best_value) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < p.n_env * p.D) {
        const int d = i % p.D;
        mu[i] = x0[i];
        const float w = upper[d] - lower[d];
        disp[i] = p.clipped ? 1.0f : (w * w) / 16.0f;
    }
    if (i < p.n_env) best_value[i] = -INFINITY;
}

}  // namespace hipets
