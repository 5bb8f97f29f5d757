// optim.hpp -- device kernels of the MPPI and iCEM trajectory optimizers.
//
//   MPPIOptimizer.optimize   mbrl/planning/trajectory_opt.py:238-311
//   ICEMOptimizer.optimize   mbrl/planning/trajectory_opt.py:391-487
//   powerlaw_psd_gaussian    mbrl/util/math.py:318-396 (Timmer & Koenig coloured noise)
//
// These are HBM/latency-bound elementwise kernels (populations are a few hundred KB); they exist so that a
// plan never leaves the device, not because they are hot.
#pragma once
#include <algorithm>

#include "cem.hpp"
#include "common.hpp"

namespace hipets {

// ---- MPPI ---------------------------------------------------------------------------------------------------
// trajectory_opt.py:262-295.  One thread per (candidate, action dim) walks the horizon: the beta-smoothing
// recurrence (:279-287) runs on UNCLIPPED values, clipping (:290-295) is applied to what is stored.
// Appendix B5: sigma never reaches the population (the scaled noise of :276 is fully overwritten).
__global__ void mppi_sample_kernel(int n_env, int pop, int H, int A, float beta, const float* __restrict__ mean,
                                   const float* __restrict__ past_action, const float* __restrict__ lower,
                                   const float* __restrict__ upper, const float* __restrict__ z_in, unsigned long long seed,
                                   unsigned long long stream, float* __restrict__ population) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_env * pop * A) return;
    const int c = i / A, a = i % A;  // c: candidate over all environments (environment c / pop owns mean / past_action)
    const int env = c / pop;
    mean += (size_t)env * H * A;
    past_action += (size_t)env * A;
    const float omb = 1.0f - beta;
    float prev = past_action[a];
    for (int h = 0; h < H; ++h) {
        const long long idx = ((long long)c * H + h) * A + a;
        const float z = z_in ? z_in[idx] : philox_trunc_normal((uint32_t)idx, (uint32_t)(idx >> 32), seed, stream);
        const float x = beta * (mean[h * A + a] + z) + omb * prev;  // :279-287
        prev = x;
        const float ub = upper[h * A + a], lb = lower[h * A + a];
        float y = x > ub ? ub : x;  // torch.where(population > upper, upper, population)
        y = y < lb ? lb : y;
        population[idx] = y;
    }
}

// The same population, dealt differently (round 6): the kernel above walks a series' horizon with a rejection-sampled draw inside every step of
// the recurrence and scatters its stores (28.5 us for pop 2 000 x H 50 x A 6).  Here a workgroup takes G whole candidates: every thread draws
// for the elements tid, tid + 256, .. of their [G][H][A] block (the draws do not depend on each other: same counters, same values), the
// recurrence -- now a multiply-add and two compares per step -- runs over the block in LDS, one thread per series, and the block leaves in
// coalesced stores.  Same expressions in the same order: same bits.
constexpr int kMppiSampleThreads = 256;
constexpr int kMppiSampleMaxD = 12288;  // [H x A] floats of one candidate that this form stages (48 KB); beyond: the kernel above
inline int mppi_sample_group(const long long npop, const int D) {  // candidates per workgroup: >= ~512 workgroups, <= 48 KB of LDS
    return (int)std::max<long long>(1, std::min<long long>(npop / 512, kMppiSampleMaxD / D));
}
__global__ __launch_bounds__(kMppiSampleThreads) void mppi_sample_staged_kernel(int n_env, int pop, int H, int A, int G, float beta, const float* __restrict__ mean,
                                                                                const float* __restrict__ past_action, const float* __restrict__ lower,
                                                                                const float* __restrict__ upper, const float* __restrict__ z_in,
                                                                                unsigned long long seed, unsigned long long stream, float* __restrict__ population) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* zs = reinterpret_cast<float*>(smem);  // [G][H][A]
    const int tid = threadIdx.x, D = H * A;
    const long long c_first = (long long)blockIdx.x * G;
    const int nc = (int)std::min<long long>(G, (long long)n_env * pop - c_first);
    const int n = nc * D;
    const long long base = c_first * D;
    for (int e = tid; e < n; e += kMppiSampleThreads) {
        const long long idx = base + e;
        zs[e] = z_in ? z_in[idx] : philox_trunc_normal((uint32_t)idx, (uint32_t)(idx >> 32), seed, stream);
    }
    __syncthreads();
    const float omb = 1.0f - beta;
    for (int q = tid; q < nc * A; q += kMppiSampleThreads) {
        const int cl = q / A, a = q % A;
        const int env = (int)((c_first + cl) / pop);
        const float* const mn = mean + (size_t)env * D;
        float prev = past_action[(size_t)env * A + a];
        float* const ser = zs + (size_t)cl * D + a;
        for (int h = 0; h < H; ++h) {
            const float x = beta * (mn[h * A + a] + ser[h * A]) + omb * prev;  // :279-287
            prev = x;
            const float ub = upper[h * A + a], lb = lower[h * A + a];
            float y = x > ub ? ub : x;  // torch.where(population > upper, upper, population)
            y = y < lb ? lb : y;
            ser[h * A] = y;
        }
    }
    __syncthreads();
    for (int e = tid; e < n; e += kMppiSampleThreads) population[base + e] = zs[e];
}

// trajectory_opt.py:296-309: NaN -> -1e-10; w = exp(gamma (v - max v)); mean = sum(w pop) / (sum w + 1e-10)
constexpr int kMppiThreads = 1024;
constexpr int kMppiTileMax = 256;  // candidates per staged tile of the weighted sum (16 loads per thread); 128 where the weights of a large population leave less LDS
// dynamic LDS of a launch: the weights [pop] (padded to 16 bytes), the reduction buffer, two tiles of `tile_c` candidates x 64 dimensions
inline size_t mppi_update_smem(const int pop, const int tile_c) { return ((size_t)((pop + 3) & ~3) + kMppiThreads) * 4 + 2 * (size_t)tile_c * 64 * 4; }
// candidates per tile (a compile-time fact of the kernel instance: its loads are unconditional and all in flight -- a first version with a
// run-time count compiled to one `s_waitcnt vmcnt(0)` per load, sixteen dependent round trips per tile: 74 us)
inline int mppi_update_tile(const int pop, const size_t lds_max) { return mppi_update_smem(pop, kMppiTileMax) + 1024 <= lds_max ? kMppiTileMax : kMppiTileMax / 2; }
template <int PER>  // tile = 16 PER candidates
__global__ __launch_bounds__(kMppiThreads) void mppi_update_kernel(int pop, int D, float gamma, float* __restrict__ values,
                                                                   const float* __restrict__ population, float* __restrict__ mean) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* w = reinterpret_cast<float*>(smem);                // [pop]
    float* red = w + ((pop + 3) & ~3);                        // [kMppiThreads]
    float* tile = red + kMppiThreads;                         // [2][tile_c][64]
    constexpr int tile_c = 16 * PER;
    const int tid = threadIdx.x;
    {  // gridDim.y environments
        const int env = blockIdx.y;
        values += (size_t)env * pop;
        population += (size_t)env * pop * D;
        mean += (size_t)env * D;
    }
    float m = -INFINITY;
    for (int i = tid; i < pop; i += kMppiThreads) {
        float v = values[i];
        if (v != v) { v = -1e-10f; values[i] = v; }
        w[i] = v;
        m = fmaxf(m, v);
    }
    red[tid] = m;
    __syncthreads();
    for (int s = kMppiThreads / 2; s > 0; s >>= 1) {
        if (tid < s) red[tid] = fmaxf(red[tid], red[tid + s]);
        __syncthreads();
    }
    const float vmax = red[0];
    __syncthreads();
    float part = 0.f;
    for (int i = tid; i < pop; i += kMppiThreads) {
        const float e = expf(gamma * (w[i] - vmax));
        w[i] = e;
        part += e;
    }
    red[tid] = part;
    __syncthreads();
    for (int s = kMppiThreads / 2; s > 0; s >>= 1) {
        if (tid < s) red[tid] += red[tid + s];
        __syncthreads();
    }
    const float norm = red[0] + 1e-10f;
    // the weighted sum of a dimension is ONE f32 chain over the candidates (the order the reference's sum leaves is not defined; ours is
    // fixed: c ascending), and the dimensions are dealt to the gridDim.x workgroups of the environment 64 at a time (every workgroup
    // repeats the weights above: same inputs, same values).  The chain is one wave's; what it waits for is the population -- round 6,
    // first version: the chain's own loads went out 32 at a time, 63 dependent round trips for pop 2 000 (146 -> 102 us).  Now ALL sixteen
    // waves fetch: a tile of `tile_c` candidates x 64 dimensions per round trip (wave v the rows v, v + 16, ..: 256 coalesced bytes each),
    // staged in LDS, the next tile in flight while wave 0 walks this one -- the same products added in the same order.
    const int lane_d = tid & 63, row0 = tid >> 6;
    for (int d0 = (int)blockIdx.x * 64; d0 < D; d0 += (int)gridDim.x * 64) {
        const int nd = min(64, D - d0);
        const float* const col = population + d0 + min(lane_d, nd - 1);
        float r[PER];
        auto fetch = [&](const int c0) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < PER; ++j) r[j] = col[(size_t)min(c0 + row0 + 16 * j, pop - 1) * D];  // (unconditional: a guarded load is a branch and a wait)
        };
        auto stash = [&](float* const buf) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < PER; ++j) buf[(row0 + 16 * j) * 64 + lane_d] = r[j];
        };
        fetch(0);
        stash(tile);
        __syncthreads();
        float acc = 0.f;
        int b = 0;
        for (int c0 = 0; c0 < pop; c0 += tile_c) {
            const bool more = c0 + tile_c < pop;  // (uniform)
            if (more) fetch(c0 + tile_c);
            if (tid < 64) {
                const float* const t = tile + (size_t)b * tile_c * 64 + tid;
                const float* const wc = w + c0;
                const int n = min(tile_c, pop - c0);
                int u = 0;
                for (; u + 8 <= n; u += 8) {  // eight products' operands requested together, added in order
                    float tv[8], wv[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) { tv[k] = t[(u + k) * 64]; wv[k] = wc[u + k]; }
#pragma unroll
                    for (int k = 0; k < 8; ++k) acc += tv[k] * wv[k];
                }
                for (; u < n; ++u) acc += t[u * 64] * wc[u];
            }
            if (more) stash(tile + (size_t)(b ^ 1) * tile_c * 64);
            __syncthreads();
            b ^= 1;
        }
        if (tid < nd) mean[d0 + tid] = acc / norm;
    }
}

// workgroups per environment of an MPPI update launch: one wave of dimensions each
inline int mppi_update_blocks(const int D) { return D <= 64 ? 1 : ((D + 63) / 64 > 32 ? 32 : (D + 63) / 64); }

// ---- iCEM ---------------------------------------------------------------------------------------------------
// Coloured noise (util/math.py:318-396) + scale / clip (trajectory_opt.py:433-441).  A thread of series (candidate, action dim)
// draws the series' H/2+1 Fourier coefficients (unit normals scaled by f^(-exponent/2), DC and Nyquist imaginary parts zero),
// inverts them with a direct real DFT for its share of the steps (H <= 128: O(H^2) per series is a few thousand FMAs) and
// normalises by the theoretical std so the series has unit variance.
constexpr int kMaxHorizon = 128;
// Round 6: the coefficients live in REGISTERS.  The first version kept re[] / im[] as run-time indexed local arrays, which the compiler
// demotes to scratch memory (the only kernel of the library with scratch: 72 us per cfg4 iteration for 1 036 x 17 series of 40 steps).
// Here the inverse transform's frequency loop is fully unrolled over NFMAX >= H / 2 + 1 (six instances: horizons up to 16 / 32 / 48 / 64 / 96 /
// 128), branch-free.  A workgroup serves kSeries series with SPLIT threads each: the series' H / 2 + 1 coefficient pairs are drawn ONCE,
// the draws dealt to its SPLIT threads, and parked in LDS columns; every thread then fetches all of them into registers (static
// indices) and inverts its own share of the time steps.  Same draws, same arithmetic per element in the same order: same bits as the
// first version (the plans' oracle replays and the reference goldens hold it).
template <int NFMAX>
__global__ __launch_bounds__((NFMAX > 33 ? 64 : 128) * 4) void icem_sample_kernel(int n_env, int row_stride, int n, int H, int A, float exponent,
                                   const float* __restrict__ mu, const float* __restrict__ var, const float* __restrict__ lower,
                                   const float* __restrict__ upper, const float* __restrict__ normals /* [2, n, A, H/2+1] or null */,
                                   unsigned long long seed, unsigned long long stream, float* __restrict__ population) {
    constexpr int kSeries = NFMAX > 33 ? 64 : 128;  // series per workgroup (the widest instance: 33 KB of coefficient columns)
    __shared__ float cs[kMaxHorizon], sn[kMaxHorizon], scale[kMaxHorizon / 2 + 1];
    __shared__ float sigma_s;
    extern __shared__ float coef[];  // [2][NF][kSeries]
    const int NF = H / 2 + 1;
    const int split = (int)blockDim.x / kSeries;  // threads per series
    const int sl = (int)threadIdx.x % kSeries, q = (int)threadIdx.x / kSeries;
    for (int m = threadIdx.x; m < H; m += blockDim.x) {
        float s, c;
        sincosf(6.28318530717958647692f * (float)m / (float)H, &s, &c);
        cs[m] = c;
        sn[m] = s;
    }
    for (int f = threadIdx.x; f < NF; f += blockDim.x) {  // util/math.py:352-358
        float fr = (float)f / (float)H;
        const float fmin = 1.0f / (float)H;  // max(fmin = 0, 1 / samples)
        if (fr < fmin) fr = fmin;            // s_scale[:ix] = s_scale[ix], ix = 1
        scale[f] = powf(fr, -exponent / 2.0f);
    }
    __syncthreads();
    if (threadIdx.x == 0) {  // :361-363  sigma = 2 sqrt(sum w^2) / samples, w[-1] *= (1 + samples % 2) / 2
        float acc = 0.f;
        for (int f = 1; f < NF; ++f) {
            float wv = scale[f];
            if (f == NF - 1) wv *= (1.0f + (float)(H % 2)) / 2.0f;
            acc += wv * wv;
        }
        sigma_s = 2.0f * sqrtf(acc) / (float)H;
    }
    // n_env environments sample n rows each into population [n_env][row_stride][H][A] from their own mu / var [n_env][H][A]
    const int i = blockIdx.x * kSeries + sl;  // the series: (environment, candidate, action dim)
    const bool valid = i < n_env * n * A;
    const int env = valid ? i / (n * A) : 0;
    const int c = valid ? (i % (n * A)) / A : 0, a = valid ? i % A : 0;
    mu += (size_t)env * H * A;
    var += (size_t)env * H * A;
    population += (size_t)env * row_stride * H * A;
    const bool even = (H & 1) == 0;
    // the draws of series i: frequency f by the thread with q = f % split (a real loop: unrolled, the compiler interleaves the Philox
    // blocks and spills hundreds of registers)
    if (valid) {
#pragma nounroll
        for (int f = q; f < NF; f += split) {
            float nr, ni;
            if (normals) {
                nr = normals[((size_t)c * A + a) * NF + f];
                ni = normals[((size_t)n * A + (size_t)c * A + a) * NF + f];
            } else {
                const Philox4 r = philox4x32_10((uint32_t)i, (uint32_t)f, 0x1CE3u, (uint32_t)stream, (uint32_t)seed,
                                                (uint32_t)(seed >> 32) ^ (uint32_t)(stream >> 32));
                box_muller(r.x, r.y, nr, ni);
            }
            float vr = nr * scale[f], vi = ni * scale[f];
            if (f == 0) vi = 0.f;                // :386
            if (even && f == NF - 1) vi = 0.f;   // :382-383
            coef[f * kSeries + sl] = vr;
            coef[(NF + f) * kSeries + sl] = vi;
        }
    }
    __syncthreads();  // (also publishes sigma_s)
    if (!valid) return;
    float re[NFMAX], im[NFMAX];
#pragma unroll
    for (int f = 0; f < NFMAX; ++f) {
        const int ff = f < NF ? f : NF - 1;  // (clamped: the select below drops what the copies contribute)
        re[f] = coef[ff * kSeries + sl];
        im[f] = coef[(NF + ff) * kSeries + sl];
    }
    const float re_last = coef[(NF - 1) * kSeries + sl];  // re[NF - 1]
    const float inv = 1.0f / ((float)H * sigma_s);
    const int fl = (H & 1) ? NF : NF - 1;  // frequencies with a distinct conjugate partner: 1 .. fl-1
    // this thread's share of the time steps: [t0, t1) of `split` equal shares
    const int per = (H + split - 1) / split;
    const int t0 = q * per, t1 = min(H, t0 + per);
    for (int t = t0; t < t1; ++t) {  // irfft (backward norm 1/H), real output
        const int d = t * A + a;
        const float v_d = var[d], mu_d = mu[d], ub_d = upper[d], lb_d = lower[d];  // (requested before the transform, used behind it)
        float y = re[0];
        int ph = 0;
        // ONE basic block over the NFMAX - 1 frequencies: the table reads of all of them go out together.  (With a branch per
        // frequency every cs / sn pair was an LDS round trip of its own.)  Frequencies beyond fl - 1 compute on clamped copies and are
        // dropped by the select: y's sum is the same chain of additions.
#pragma unroll
        for (int f = 1; f < NFMAX; ++f) {
            ph += t;
            if (ph >= H) ph -= H;  // (f * t) mod H
            const float yn = y + 2.0f * (re[f] * cs[ph] - im[f] * sn[ph]);
            y = f < fl ? yn : y;
        }
        if (even) y += re_last * ((t & 1) ? -1.0f : 1.0f);
        const float noise = y * inv;
        float x = noise * sqrtf(v_d) + mu_d;               // trajectory_opt.py:438-439
        x = fminf(x, ub_d);                                // torch.minimum(.., upper)
        x = fmaxf(x, lb_d);                                // torch.maximum(.., lower)
        population[((size_t)c * H + t) * A + a] = x;
    }
}

// host side: the instance for the horizon; a series' steps (and draws) dealt to 1 / 2 / 4 threads so that the launch fills the chip
inline void launch_icem_sample(hipStream_t st, int n_env, int row_stride, int n, int H, int A, float exponent, const float* mu, const float* var,
                               const float* lower, const float* upper, const float* normals, unsigned long long seed, unsigned long long stream,
                               float* population) {
    const long long total = (long long)n_env * n * A;
    const int split = total >= 65536 ? 1 : (total >= 32768 ? 2 : 4);  // (1 024 SIMDs x 64 lanes)
    const int NF = H / 2 + 1;
    const unsigned series = NF > 33 ? 64u : 128u;
    const dim3 grid((unsigned)((total + series - 1) / series)), block(series * (unsigned)split);
    const size_t lds = (size_t)2 * NF * series * sizeof(float);
#define HIPETS_ICEM_LAUNCH(NFM) \
    hipLaunchKernelGGL(icem_sample_kernel<NFM>, grid, block, lds, st, n_env, row_stride, n, H, A, exponent, mu, var, lower, upper, normals, seed, stream, population)
    if (NF <= 9) HIPETS_ICEM_LAUNCH(9);
    else if (NF <= 17) HIPETS_ICEM_LAUNCH(17);
    else if (NF <= 25) HIPETS_ICEM_LAUNCH(25);
    else if (NF <= 33) HIPETS_ICEM_LAUNCH(33);
    else if (NF <= 49) HIPETS_ICEM_LAUNCH(49);
    else HIPETS_ICEM_LAUNCH(kMaxHorizon / 2 + 1);
#undef HIPETS_ICEM_LAUNCH
}

// trajectory_opt.py:450-462: kept elites shifted one step, tail action ~ N(mu[-1], sqrt(var[-1]))
__global__ void icem_shift_kernel(int n_env, int row_stride, int keep, int H, int A, const float* __restrict__ kept,
                                  const float* __restrict__ mu, const float* __restrict__ var,
                                  const float* __restrict__ end_noise /* [keep, A] or null */, unsigned long long seed,
                                  unsigned long long stream, float* __restrict__ out) {
    // kept [n_env][keep][H][A]; out: `keep` rows per environment, environments row_stride rows apart
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_env * keep * H * A) return;
    const int a = i % A, h = (i / A) % H, jg = i / (A * H);  // jg: kept row over all environments
    const int env = jg / keep, j = jg % keep;
    const size_t o = ((size_t)env * row_stride + j) * H * A + (size_t)h * A + a;
    if (h < H - 1) {
        out[o] = kept[((size_t)jg * H + h + 1) * A + a];
    } else {
        const int d = (H - 1) * A + a;
        const float z = end_noise ? end_noise[jg * A + a]
                                  : philox_normal((uint32_t)(jg * A + a), 0x5E1Fu, seed, stream ^ 0x9E3779B97F4A7C15ull);
        out[o] = mu[(size_t)env * H * A + d] + sqrtf(var[(size_t)env * H * A + d]) * z;
    }
}

// MPPIOptimizer.optimize prologue (trajectory_opt.py:257-258): mean[:-1] = mean[1:] (the last row stays) and
// past_action = the ALREADY shifted mean[0] (Appendix B5).  `src` is a private copy of the caller's mean.
__global__ void mppi_shift_kernel(int n_env, int H, int A, const float* __restrict__ src, float* __restrict__ mean,
                                  float* __restrict__ past_action) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_env * H * A) return;
    const int env = i / (H * A), r = i % (H * A);
    const int t = r / A, a = r % A;
    const float v = src[(size_t)env * H * A + (t + 1 < H ? t + 1 : t) * A + a];
    mean[i] = v;
    if (t == 0) past_action[env * A + a] = v;
}

// `keep` distinct indices drawn uniformly from [0, K) in random order: the law of torch.randperm(K)[:keep]
// (trajectory_opt.py:446-448).  One workgroup: every index gets a Philox key, its rank among the keys is its
// position in the permutation.
__global__ __launch_bounds__(256) void icem_keep_select_kernel(int K, int keep, unsigned long long seed, unsigned long long stream,
                                                              int* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(smem);
    out += (size_t)blockIdx.x * keep;  // one workgroup (and one independent draw) per environment
    for (int j = threadIdx.x; j < K; j += blockDim.x) {
        const Philox4 r = philox4x32_10((uint32_t)j, 0x4B454550u, (uint32_t)blockIdx.x, (uint32_t)stream, (uint32_t)seed,
                                        (uint32_t)(seed >> 32) ^ (uint32_t)(stream >> 32) ^ 0x5EED5EEDu);
        keys[j] = ((unsigned long long)r.x << 32) | r.y;
    }
    __syncthreads();
    for (int j = threadIdx.x; j < K; j += blockDim.x) {
        const unsigned long long kj = keys[j];
        int rank = 0;
        for (int i = 0; i < K; ++i) rank += (keys[i] < kj) || (keys[i] == kj && i < j);
        if (rank < keep) out[rank] = j;
    }
}

// rows of `src` selected by `index` (int64, like torch.index_select) -> dst; used for population[elite_idx]
__global__ void gather_rows_kernel(int rows, int D, const float* __restrict__ src, const int* __restrict__ index, float* __restrict__ dst,
                                   long long src_env_stride, long long dst_env_stride) {
    // blockIdx.y = environment: src / dst advance by their environment strides (floats), index by `rows`
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * D) return;
    const int env = blockIdx.y;
    dst[(size_t)env * dst_env_stride + i] = src[(size_t)env * src_env_stride + (size_t)index[(size_t)env * rows + i / D] * D + (i % D)];
}

// trajectory_opt.py:463-464 for every environment: the extra candidate of the last iteration is the current mean
__global__ void icem_append_mu_kernel(int n_env, int row_stride, int row, int D, const float* __restrict__ mu, float* __restrict__ population) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_env * D) return;
    const int env = i / D, d = i % D;
    population[((size_t)env * row_stride + row) * D + d] = mu[i];
}

}  // namespace hipets
