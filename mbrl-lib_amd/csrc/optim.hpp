// optim.hpp -- device kernels of the MPPI and iCEM trajectory optimizers.
//
//   MPPIOptimizer.optimize   mbrl/planning/trajectory_opt.py:238-311
//   ICEMOptimizer.optimize   mbrl/planning/trajectory_opt.py:391-487
//   powerlaw_psd_gaussian    mbrl/util/math.py:318-396 (Timmer & Koenig coloured noise)
//
// These are HBM/latency-bound elementwise kernels (populations are a few hundred KB); they exist so that a
// plan never leaves the device, not because they are hot.
#pragma once
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

// trajectory_opt.py:296-309: NaN -> -1e-10; w = exp(gamma (v - max v)); mean = sum(w pop) / (sum w + 1e-10)
constexpr int kMppiThreads = 1024;
__global__ __launch_bounds__(kMppiThreads) void mppi_update_kernel(int pop, int D, float gamma, float* __restrict__ values,
                                                                   const float* __restrict__ population, float* __restrict__ mean) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* w = reinterpret_cast<float*>(smem);                // [pop]
    float* red = w + pop;                                     // [kMppiThreads]
    const int tid = threadIdx.x;
    {  // one workgroup per environment
        const int env = blockIdx.x;
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
    for (int d = tid; d < D; d += kMppiThreads) {
        float acc = 0.f;
        for (int c = 0; c < pop; ++c) acc += population[(size_t)c * D + d] * w[c];
        mean[d] = acc / norm;
    }
}

// ---- iCEM ---------------------------------------------------------------------------------------------------
// Coloured noise (util/math.py:318-396) + scale / clip (trajectory_opt.py:433-441).  One thread per
// (candidate, action dim) draws the H/2+1 Fourier coefficients (unit normals scaled by f^(-exponent/2), DC and
// Nyquist imaginary parts zero), inverts them with a direct real DFT (H <= 64 here: O(H^2) per series is a few
// thousand FMAs) and normalises by the theoretical std so the series has unit variance.
constexpr int kMaxHorizon = 128;
__global__ void icem_sample_kernel(int n_env, int row_stride, int n, int H, int A, float exponent, const float* __restrict__ mu,
                                   const float* __restrict__ var, const float* __restrict__ lower, const float* __restrict__ upper,
                                   const float* __restrict__ normals /* [2, n, A, H/2+1] or null */, unsigned long long seed,
                                   unsigned long long stream, float* __restrict__ population) {
    __shared__ float cs[kMaxHorizon], sn[kMaxHorizon], scale[kMaxHorizon / 2 + 1];
    __shared__ float sigma_s;
    const int NF = H / 2 + 1;
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
    __syncthreads();
    // n_env environments sample n rows each into population [n_env][row_stride][H][A] from their own mu / var [n_env][H][A]
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_env * n * A) return;
    const int env = i / (n * A);
    const int c = (i % (n * A)) / A, a = i % A;
    mu += (size_t)env * H * A;
    var += (size_t)env * H * A;
    population += (size_t)env * row_stride * H * A;
    float re[kMaxHorizon / 2 + 1], im[kMaxHorizon / 2 + 1];
    for (int f = 0; f < NF; ++f) {
        float nr, ni;
        if (normals) {
            nr = normals[((size_t)c * A + a) * NF + f];
            ni = normals[((size_t)n * A + (size_t)c * A + a) * NF + f];
        } else {
            const Philox4 r = philox4x32_10((uint32_t)i, (uint32_t)f, 0x1CE3u, (uint32_t)stream, (uint32_t)seed,
                                            (uint32_t)(seed >> 32) ^ (uint32_t)(stream >> 32));
            box_muller(r.x, r.y, nr, ni);
        }
        re[f] = nr * scale[f];
        im[f] = ni * scale[f];
    }
    im[0] = 0.f;                       // :386
    if ((H & 1) == 0) im[NF - 1] = 0.f;  // :382-383
    const float inv = 1.0f / ((float)H * sigma_s);
    for (int t = 0; t < H; ++t) {  // irfft (backward norm 1/H), real output
        float y = re[0];
        const int fl = (H & 1) ? NF : NF - 1;  // frequencies with a distinct conjugate partner: 1 .. fl-1
        int ph = 0;
        for (int f = 1; f < fl; ++f) {
            ph += t;
            if (ph >= H) ph -= H;  // (f * t) mod H
            y += 2.0f * (re[f] * cs[ph] - im[f] * sn[ph]);
        }
        if ((H & 1) == 0) y += re[NF - 1] * ((t & 1) ? -1.0f : 1.0f);
        const float noise = y * inv;
        const int d = t * A + a;
        float x = noise * sqrtf(var[d]) + mu[d];           // trajectory_opt.py:438-439
        x = fminf(x, upper[d]);                            // torch.minimum(.., upper)
        x = fmaxf(x, lower[d]);                            // torch.maximum(.., lower)
        population[((size_t)c * H + t) * A + a] = x;
    }
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
