// rollout.hpp -- the fused PETS rollout kernel for gfx950 (MI355X).
//
// Replaces, per planning step, the ~55 ATen launches of
//   ModelEnv.evaluate_action_sequences   (mbrl/models/model_env.py:145-191)
//   OneDTransitionRewardModel.sample     (mbrl/models/one_dim_tr_model.py:245-289, :103-116)
//   GaussianMLP._forward_ensemble        (mbrl/models/gaussian_mlp.py:129-216)
//   EnsembleLinearLayer.forward          (mbrl/models/util.py:53-65)
//   Ensemble.sample_1d                   (mbrl/models/model.py:426-473)
//   reward / termination fns             (mbrl/env/reward_fns.py, termination_fns.py)
// with one kernel.  A workgroup (4 waves, one per SIMD) owns R row tiles of 16 rollout rows that
// all use the SAME ensemble member in a given step, keeps their activations in LDS (ping-pong
// [rows][ld] f32 buffers, ld == 8 mod 64 so ds_read_b128 A-fragment reads are conflict free) and
// streams that member's weights from L2 as pre-packed v_mfma_f32_16x16x4_f32 B fragments
// (one coalesced 1 KiB global_load_dwordx4 per 16x16 k-chunk, reused by all R row tiles).
//
//   EXACT mode: one launch per step; rows are gathered through the reference's randperm so that
//               workgroup (member m, chunk) sees exactly rows perm[m*B/M + ...] (bit-for-bit the
//               reference's row->member map); state lives in HBM between launches.
//   FAST  mode: one launch for the whole horizon; workgroup (particle p, candidate group g) owns
//               its rows for all H steps, state stays in LDS, the member is drawn per
//               (workgroup, step) from a balanced schedule, eps comes from Philox.
#pragma once
#include "common.hpp"

namespace hipets {

struct LayerMeta {
    int Kp, Np;          // K, N padded to multiples of 16
    int boff;            // float offset of the layer's bias inside a member block
    int pad_;
    long long woff;      // float offset of the layer's packed weights inside a member block
};

struct Extras {  // up to 3 leftover (column tile, row tile) units of one wave
    int c0, c1, c2, r0, r1, r2;
};

struct ModelDev {
    int obs_dim, act_dim, in_dim, out_dim, out_total, hid, n_layers, M;
    int obs_in;  // width of obs_process_fn(obs) = in_dim - act_dim
    int activation;
    float slope;
    int propagation, deterministic, obs_process, reward_fn, term_fn, target_is_delta, learned_rewards, normalizer;
    const LayerMeta* layers;  // DEVICE [n_layers] (a table in memory: runtime-indexed kernargs would go to scratch)
    int Kp0;                  // padded input width of layer 0
    int hidC;                 // column tiles of a hidden layer (cost model)
    long long wmember;  // floats per member (packed weights)
    int bmember;        // floats per member (padded biases)
    int ld;             // LDS activation row stride in floats (== 8 mod 64)
    const float* w;
    const float* b;
    const double* norm_mean;
    const double* norm_std;
    const float* min_lv;
    const float* max_lv;
    const unsigned char* no_delta;  // [obs_dim]
};

struct RolloutArgs {
    int pop, P, H, B;
    int mode;
    int t_begin, t_end;
    int groups;           // FAST: candidate groups per particle; EXACT: workgroups per member domain
    int rows_per_domain;  // EXACT: B / M (or B for expectation)
    const float* actions;  // [pop,H,A]
    const float* s0;       // [obs]
    float* state;          // EXACT: [B,obs] in/out
    float* totals;         // [B] (EXACT in/out; FAST out)
    unsigned char* term;   // EXACT: [B] in/out
    const long long* perm; // EXACT: [H,B] / [B] / null
    long long perm_step;   // stride between steps (0 for fixed_model)
    const float* eps;      // [H,B,out] or null
    int use_philox;        // FAST without eps override
    unsigned long long seed, stream_id;
    const int* schedule;   // FAST: [H, nWG] member slot per (step, workgroup)
    float* trace_next_obs;
    float* trace_rewards;
};

__device__ __forceinline__ f32x4 mfma16x16x4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// One wave's share of a layer: CT strided column tiles (c_first + 4*ct) for all R row tiles, plus EX
// "extra" (column tile, row tile) units taken from the C % 4 leftover column tiles, all accumulated
// in the same k loop so the MFMA pipe always has >= 2 independent accumulators in flight.
template <int R, int CT, int EX>
__device__ __forceinline__ void wave_gemm(const float* __restrict__ in, float* __restrict__ out, const int ld,
                                          const float* __restrict__ W, const float* __restrict__ bias, const int KC,
                                          const int c_first, const Extras ex,
                                          const bool apply_act, const int act, const float slope, const int lane) {
    constexpr int CTn = CT > 0 ? CT : 1;
    constexpr int EXn = EX > 0 ? EX : 1;
    f32x4 acc[CTn][R];
    f32x4 accx[EXn];
#pragma unroll
    for (int ct = 0; ct < CTn; ++ct)
#pragma unroll
        for (int r = 0; r < R; ++r) acc[ct][r] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < EXn; ++e) accx[e] = f32x4{0.f, 0.f, 0.f, 0.f};

    const f32x4* wp[CTn];
    const f32x4* wx[EXn];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) wp[ct] = reinterpret_cast<const f32x4*>(W) + (size_t)(c_first + 4 * ct) * KC * 64 + lane;
    const int exc[3] = {ex.c0, ex.c1, ex.c2};
    const int exr[3] = {ex.r0, ex.r1, ex.r2};
#pragma unroll
    for (int e = 0; e < EX; ++e) wx[e] = reinterpret_cast<const f32x4*>(W) + (size_t)exc[e] * KC * 64 + lane;
    const float* ap = in + (lane & 15) * ld + 4 * (lane >> 4);

// (k loop left rolled: the body is already 4*(CT*R+EX) MFMAs)
    for (int kk = 0; kk < KC; ++kk) {
        f32x4 b[CTn], bx[EXn], a[R];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) b[ct] = wp[ct][(size_t)kk * 64];
#pragma unroll
        for (int e = 0; e < EX; ++e) bx[e] = wx[e][(size_t)kk * 64];
#pragma unroll
        for (int r = 0; r < R; ++r) a[r] = *reinterpret_cast<const f32x4*>(ap + r * 16 * ld + kk * 16);
        f32x4 ax[EXn];
#pragma unroll
        for (int e = 0; e < EX; ++e)  // own LDS read (a runtime-indexed register array would go to scratch)
            ax[e] = *reinterpret_cast<const f32x4*>(ap + exr[e] * 16 * ld + kk * 16);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int r = 0; r < R; ++r) acc[ct][r] = mfma16x16x4(a[r][s], b[ct][s], acc[ct][r]);
#pragma unroll
            for (int e = 0; e < EX; ++e) accx[e] = mfma16x16x4(ax[e][s], bx[e][s], accx[e]);
        }
    }

    // epilogue: D[row = 4*(lane>>4)+i][col = lane&15] -> bias, activation, next layer's A image
    const int j = lane & 15, g4 = 4 * (lane >> 4);
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int col = (c_first + 4 * ct) * 16 + j;
        const float bv = bias[col];
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float v = acc[ct][r][i] + bv;
                if (apply_act) v = activate(v, act, slope);
                out[(r * 16 + g4 + i) * ld + col] = v;
            }
    }
#pragma unroll
    for (int e = 0; e < EX; ++e) {
        const int col = exc[e] * 16 + j;
        const float bv = bias[col];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float v = accx[e][i] + bv;
            if (apply_act) v = activate(v, act, slope);
            out[(exr[e] * 16 + g4 + i) * ld + col] = v;
        }
    }
}

template <int R, int CT>
__device__ __forceinline__ void wave_gemm_ex(int nex, const float* in, float* out, int ld, const float* W,
                                             const float* bias, int KC, int c_first, const Extras ex,
                                             bool apply_act, int act, float slope, int lane) {
    switch (nex) {
        case 0:
            if constexpr (CT > 0) wave_gemm<R, CT, 0>(in, out, ld, W, bias, KC, c_first, ex, apply_act, act, slope, lane);
            break;
        case 1: wave_gemm<R, CT, 1>(in, out, ld, W, bias, KC, c_first, ex, apply_act, act, slope, lane); break;
        case 2: wave_gemm<R, CT, 2>(in, out, ld, W, bias, KC, c_first, ex, apply_act, act, slope, lane); break;
        default: wave_gemm<R, CT, 3>(in, out, ld, W, bias, KC, c_first, ex, apply_act, act, slope, lane); break;
    }
}

// One linear layer (+activation) for the workgroup's 16*R rows: in (LDS) -> out (LDS).
template <int R>
__device__ __forceinline__ void mlp_layer(const ModelDev& md, const int l, const int member, const float* in,
                                          float* out, const int wave, const int lane) {
    const LayerMeta lm = md.layers[l];
    const int KC = lm.Kp / kKChunk;
    const int C = lm.Np / kTile;
    const float* W = md.w + (size_t)member * md.wmember + lm.woff;
    const float* bias = md.b + (size_t)member * md.bmember + lm.boff;
    const bool apply_act = l < md.n_layers - 1;
    const int full = C / kWaves, rem = C % kWaves;
    // leftover units u = (column tile kWaves*full + u / R, row tile u % R), dealt round-robin to waves
    const int nu = rem * R;
    Extras ex;
    ex.c0 = kWaves * full + wave / R;              ex.r0 = wave % R;
    ex.c1 = kWaves * full + (wave + kWaves) / R;     ex.r1 = (wave + kWaves) % R;
    ex.c2 = kWaves * full + (wave + 2 * kWaves) / R; ex.r2 = (wave + 2 * kWaves) % R;
    const int nex = wave < nu ? (nu - wave + kWaves - 1) / kWaves : 0;
    int done = 0;
    while (full - done > 3) {
        wave_gemm<R, 3, 0>(in, out, md.ld, W, bias, KC, wave + kWaves * done, ex, apply_act, md.activation, md.slope, lane);
        done += 3;
    }
    const int c_first = wave + kWaves * done;
    switch (full - done) {
        case 0: wave_gemm_ex<R, 0>(nex, in, out, md.ld, W, bias, KC, c_first, ex, apply_act, md.activation, md.slope, lane); break;
        case 1: wave_gemm_ex<R, 1>(nex, in, out, md.ld, W, bias, KC, c_first, ex, apply_act, md.activation, md.slope, lane); break;
        case 2: wave_gemm_ex<R, 2>(nex, in, out, md.ld, W, bias, KC, c_first, ex, apply_act, md.activation, md.slope, lane); break;
        default: wave_gemm_ex<R, 3>(nex, in, out, md.ld, W, bias, KC, c_first, ex, apply_act, md.activation, md.slope, lane); break;
    }
}

// obs_process_fn(obs)[i] (mbrl/env/pets_halfcheetah.py:91-113, pets_cartpole.py:78-101)
__device__ __forceinline__ float processed_obs(const float* s, int i, int mode) {
    if (mode == HIPETS_OBS_HALFCHEETAH) {  // [s1, sin s2, cos s2, s3:]
        if (i == 0) return s[1];
        if (i == 1) return sinf(s[2]);
        if (i == 2) return cosf(s[2]);
        return s[i];
    }
    if (mode == HIPETS_OBS_CARTPOLE_PETS) {  // [sin s1, cos s1, s0, s2:]
        if (i == 0) return sinf(s[1]);
        if (i == 1) return cosf(s[1]);
        if (i == 2) return s[0];
        return s[i - 1];
    }
    return s[i];
}

__device__ __forceinline__ bool term_eval(const float* s, int obs_dim, int fn) {
    switch (fn) {
        case HIPETS_TERM_CARTPOLE: {  // termination_fns.py:29-44
            const float x = s[0], th = s[2], thr = (float)(12.0 * 2.0 * 3.14159265358979323846 / 360.0);
            return !((x > -2.4f) && (x < 2.4f) && (th > -thr) && (th < thr));
        }
        case HIPETS_TERM_INVERTED_PENDULUM: {  // :47-55
            bool fin = true;
            for (int d = 0; d < obs_dim; ++d) fin = fin && isfinite(s[d]);
            return !(fin && (fabsf(s[1]) <= 0.2f));
        }
        case HIPETS_TERM_HOPPER: {  // :12-26
            bool ok = true;
            for (int d = 0; d < obs_dim; ++d) ok = ok && isfinite(s[d]);
            for (int d = 1; d < obs_dim; ++d) ok = ok && (fabsf(s[d]) < 100.0f);
            return !(ok && (s[0] > 0.7f) && (fabsf(s[1]) < 0.2f));
        }
        case HIPETS_TERM_WALKER2D:  // :66-74
            return !((s[0] > 0.8f) && (s[0] < 2.0f) && (s[1] > -1.0f) && (s[1] < 1.0f));
        case HIPETS_TERM_ANT: {  // :77-85
            bool fin = true;
            for (int d = 0; d < obs_dim; ++d) fin = fin && isfinite(s[d]);
            return !(fin && (s[0] >= 0.2f) && (s[0] <= 1.0f));
        }
        case HIPETS_TERM_HUMANOID:  // :88-95
            return (s[0] < 1.0f) || (s[0] > 2.0f);
        default: return false;  // no_termination :58-63
    }
}

__device__ __forceinline__ float reward_eval(const float* s, const float* a, int obs_dim, int act_dim, int fn,
                                             float learned) {
    switch (fn) {
        case HIPETS_REW_CARTPOLE: return term_eval(s, obs_dim, HIPETS_TERM_CARTPOLE) ? 0.0f : 1.0f;  // reward_fns.py:10-13
        case HIPETS_REW_INVERTED_PENDULUM: return term_eval(s, obs_dim, HIPETS_TERM_INVERTED_PENDULUM) ? 0.0f : 1.0f;
        case HIPETS_REW_CARTPOLE_PETS: {  // :16-24
            const float e0 = (s[0] - 0.6f * sinf(s[1])) - 0.0f, e1 = (-0.6f * cosf(s[1])) - 0.6f;
            const float obs_cost = expf(-(e0 * e0 + e1 * e1) / (float)(0.6 * 0.6));
            float sq = 0.f;
            for (int i = 0; i < act_dim; ++i) sq += a[i] * a[i];
            return obs_cost + (-0.01f * sq);
        }
        case HIPETS_REW_HALFCHEETAH: {  // :33-38
            float sq = 0.f;
            for (int i = 0; i < act_dim; ++i) sq += a[i] * a[i];
            const float run = s[0] - 0.0f * (s[2] * s[2]);
            return run + (-0.1f * sq);
        }
        case HIPETS_REW_PUSHER: {  // :41-53
            const float g0 = 0.45f, g1 = -0.05f, g2 = -0.323f;
            const float tip_obj = fabsf(s[14] - s[17]) + fabsf(s[15] - s[18]) + fabsf(s[16] - s[19]);
            const float obj_goal = fabsf(g0 - s[17]) + fabsf(g1 - s[18]) + fabsf(g2 - s[19]);
            const float obs_cost = 0.5f * tip_obj + 1.25f * obj_goal;
            float sq = 0.f;
            for (int i = 0; i < act_dim; ++i) sq += a[i] * a[i];
            return -(obs_cost + 0.1f * sq);
        }
        default: return learned;  // model_env.py:124-128 with reward_fn None
    }
}

// the 4 standard normals of (row, step, dim block): counter = (row, step, block, stream), key = seed
__device__ __forceinline__ void rollout_normals4(int rid, int t, int blk, unsigned long long seed,
                                                 unsigned long long stream_id, float (&nrm)[4]) {
    const Philox4 r4 = philox4x32_10((uint32_t)rid, (uint32_t)t, (uint32_t)blk, (uint32_t)stream_id, (uint32_t)seed,
                                     (uint32_t)(seed >> 32) ^ (uint32_t)(stream_id >> 32));
    box_muller(r4.x, r4.y, nrm[0], nrm[1]);
    box_muller(r4.z, r4.w, nrm[2], nrm[3]);
}

struct RolloutSmem {
    float* buf0;
    float* buf1;
    float* state;   // [ROWS][obs_dim]
    float* actn;    // [ROWS][act_dim]
    float* tot;     // [ROWS]
    float* lrew;    // [ROWS] learned reward of the current step
    int* term;      // [ROWS]
    int* rowid;     // [ROWS] global row id (candidate*P + particle) or -1
    float* expacc;  // [ROWS][out_total] (expectation propagation only)
};

__host__ __device__ inline size_t align16(size_t x) { return (x + 15) & ~(size_t)15; }

__host__ __device__ inline size_t rollout_smem_bytes(int rows, int ld, int obs_dim, int act_dim, int out_total,
                                                     bool expectation) {
    size_t n = 0;
    n += 2 * align16((size_t)rows * ld * 4);
    n += align16((size_t)rows * obs_dim * 4);
    n += align16((size_t)rows * act_dim * 4);
    n += 4 * align16((size_t)rows * 4);
    if (expectation) n += align16((size_t)rows * out_total * 4);
    return n;
}

template <int R>
__global__ __launch_bounds__(kThreads) void rollout_kernel(const ModelDev md, const RolloutArgs ra) {
    constexpr int ROWS = kTile * R;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    RolloutSmem sm;
    {
        char* p = smem;
        sm.buf0 = reinterpret_cast<float*>(p); p += align16((size_t)ROWS * md.ld * 4);
        sm.buf1 = reinterpret_cast<float*>(p); p += align16((size_t)ROWS * md.ld * 4);
        sm.state = reinterpret_cast<float*>(p); p += align16((size_t)ROWS * md.obs_dim * 4);
        sm.actn = reinterpret_cast<float*>(p); p += align16((size_t)ROWS * md.act_dim * 4);
        sm.tot = reinterpret_cast<float*>(p); p += align16((size_t)ROWS * 4);
        sm.lrew = reinterpret_cast<float*>(p); p += align16((size_t)ROWS * 4);
        sm.term = reinterpret_cast<int*>(p); p += align16((size_t)ROWS * 4);
        sm.rowid = reinterpret_cast<int*>(p); p += align16((size_t)ROWS * 4);
        sm.expacc = reinterpret_cast<float*>(p);
    }
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool fast = ra.mode == HIPETS_MODE_FAST;
    const bool expectation = md.propagation == HIPETS_PROP_EXPECTATION;
    const int wg = blockIdx.x;

    // ---- which rollout rows does this workgroup own -----------------------------------------
    int domain = 0;
    if (fast) {
        const int p = wg % ra.P, grp = wg / ra.P;
        for (int s = tid; s < ROWS; s += kThreads) {
            const int c = grp * ROWS + s;
            sm.rowid[s] = c < ra.pop ? c * ra.P + p : -1;
        }
    } else {
        domain = wg / ra.groups;
        const int j0 = (wg % ra.groups) * ROWS;
        const long long* perm = ra.perm ? ra.perm + (long long)ra.t_begin * ra.perm_step : nullptr;
        for (int s = tid; s < ROWS; s += kThreads) {
            const int j = j0 + s;
            int rid = -1;
            if (j < ra.rows_per_domain) {
                const int jj = domain * ra.rows_per_domain + j;
                rid = perm ? (int)perm[jj] : jj;
            }
            sm.rowid[s] = rid;
        }
    }
    __syncthreads();

    // ---- initial state ------------------------------------------------------------------------
    for (int i = tid; i < ROWS * md.obs_dim; i += kThreads) {
        const int s = i / md.obs_dim, d = i % md.obs_dim;
        const int rid = sm.rowid[s];
        float v = 0.f;
        if (fast) v = ra.s0[d];
        else if (rid >= 0) v = ra.state[(size_t)rid * md.obs_dim + d];
        sm.state[i] = v;
    }
    for (int s = tid; s < ROWS; s += kThreads) {
        const int rid = sm.rowid[s];
        sm.tot[s] = (!fast && rid >= 0) ? ra.totals[rid] : 0.f;
        sm.term[s] = (!fast && rid >= 0) ? (int)ra.term[rid] : 0;
        sm.lrew[s] = 0.f;
    }
    __syncthreads();

    const int nblk = (md.out_dim + 3) / 4;
    const int Kp0 = md.Kp0;

    for (int t = ra.t_begin; t < ra.t_end; ++t) {
        // ---- actions of this step (model_env.py:179-182: row r uses candidate r // P) ----------
        for (int i = tid; i < ROWS * md.act_dim; i += kThreads) {
            const int s = i / md.act_dim, a = i % md.act_dim;
            const int rid = sm.rowid[s];
            sm.actn[i] = rid >= 0 ? ra.actions[((size_t)(rid / ra.P) * ra.H + t) * md.act_dim + a] : 0.f;
        }
        __syncthreads();

        const int n_run = expectation ? md.M : 1;
        float* result = nullptr;
        for (int mi = 0; mi < n_run; ++mi) {
            int member;
            if (expectation) member = mi;
            else if (fast) member = ra.schedule[(size_t)t * gridDim.x + wg];
            else member = domain;

            // ---- model input: cat(obs_process(obs), act), normalised (one_dim_tr_model.py:103-116)
            for (int i = tid; i < ROWS * Kp0; i += kThreads) {
                const int s = i / Kp0, c = i % Kp0;
                float v = 0.f;
                if (c < md.in_dim && sm.rowid[s] >= 0) {
                    v = c < md.obs_in ? processed_obs(sm.state + s * md.obs_dim, c, md.obs_process)
                                      : sm.actn[s * md.act_dim + (c - md.obs_in)];
                    if (md.normalizer == HIPETS_NORM_F64) v = (float)(((double)v - md.norm_mean[c]) / md.norm_std[c]);
                    else if (md.normalizer == HIPETS_NORM_F32) v = (v - (float)md.norm_mean[c]) / (float)md.norm_std[c];
                }
                sm.buf0[s * md.ld + c] = v;
            }
            __syncthreads();

            // ---- the MLP: ping-pong through LDS ------------------------------------------------
            float* cur = sm.buf0;
            float* nxt = sm.buf1;
            for (int l = 0; l < md.n_layers; ++l) {
                mlp_layer<R>(md, l, member, cur, nxt, wave, lane);
                __syncthreads();
                float* tmp = cur; cur = nxt; nxt = tmp;
            }
            result = cur;

            if (expectation) {  // gaussian_mlp.py:213-215: mean over members of mean AND (clamped) logvar
                for (int i = tid; i < ROWS * md.out_total; i += kThreads) {
                    const int s = i / md.out_total, c = i % md.out_total;
                    float v = result[s * md.ld + c];
                    if (!md.deterministic && c >= md.out_dim) {
                        const int d = c - md.out_dim;
                        v = md.max_lv[d] - softplus_f(md.max_lv[d] - v);
                        v = md.min_lv[d] + softplus_f(v - md.min_lv[d]);
                    }
                    sm.expacc[i] = mi == 0 ? v : sm.expacc[i] + v;
                }
                __syncthreads();
            }
        }

        // ---- sample, delta, next obs (model.py:458-473, one_dim_tr_model.py:280-288) -----------
        for (int item = tid; item < ROWS * nblk; item += kThreads) {
            const int s = item / nblk, blk = item % nblk;
            const int rid = sm.rowid[s];
            if (rid < 0) continue;
            float nrm[4] = {0.f, 0.f, 0.f, 0.f};
            if (!md.deterministic) {
                if (ra.eps) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int d = blk * 4 + q;
                        if (d < md.out_dim) nrm[q] = ra.eps[((size_t)t * ra.B + rid) * md.out_dim + d];
                    }
                } else if (ra.use_philox) {
                    rollout_normals4(rid, t, blk, ra.seed, ra.stream_id, nrm);
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int d = blk * 4 + q;
                if (d >= md.out_dim) break;
                float mean, lv = 0.f;
                if (expectation) {
                    mean = sm.expacc[s * md.out_total + d] / (float)md.M;
                    if (!md.deterministic) lv = sm.expacc[s * md.out_total + md.out_dim + d] / (float)md.M;
                } else {
                    mean = result[s * md.ld + d];
                    if (!md.deterministic) {
                        lv = result[s * md.ld + md.out_dim + d];
                        lv = md.max_lv[d] - softplus_f(md.max_lv[d] - lv);  // gaussian_mlp.py:152
                        lv = md.min_lv[d] + softplus_f(lv - md.min_lv[d]);  // :153
                    }
                }
                float pred = mean;
                if (!md.deterministic && (ra.eps || ra.use_philox)) pred = mean + sqrtf(expf(lv)) * nrm[q];
                if (d < md.obs_dim) {
                    float nobs = pred;
                    if (md.target_is_delta && !md.no_delta[d]) nobs = pred + sm.state[s * md.obs_dim + d];
                    sm.state[s * md.obs_dim + d] = nobs;
                    if (ra.trace_next_obs) ra.trace_next_obs[((size_t)t * ra.B + rid) * md.obs_dim + d] = nobs;
                } else {
                    sm.lrew[s] = pred;  // learned reward = last output (one_dim_tr_model.py:287)
                }
            }
        }
        __syncthreads();

        // ---- reward, termination, masked accumulation (model_env.py:124-129, :186-188) ---------
        for (int s = tid; s < ROWS; s += kThreads) {
            const int rid = sm.rowid[s];
            if (rid < 0) continue;
            const float* st = sm.state + s * md.obs_dim;
            const float* ac = sm.actn + s * md.act_dim;
            float r = reward_eval(st, ac, md.obs_dim, md.act_dim, md.reward_fn, sm.lrew[s]);
            const bool done = term_eval(st, md.obs_dim, md.term_fn);
            if (ra.trace_rewards) ra.trace_rewards[(size_t)t * ra.B + rid] = r;
            if (sm.term[s]) r = 0.f;
            sm.term[s] = sm.term[s] | (done ? 1 : 0);
            sm.tot[s] += r;
        }
        __syncthreads();
    }

    // ---- write back -------------------------------------------------------------------------------
    for (int s = tid; s < ROWS; s += kThreads) {
        const int rid = sm.rowid[s];
        if (rid < 0) continue;
        ra.totals[rid] = sm.tot[s];
        if (!fast) ra.term[rid] = (unsigned char)sm.term[s];
    }
    if (!fast) {
        for (int i = tid; i < ROWS * md.obs_dim; i += kThreads) {
            const int s = i / md.obs_dim, d = i % md.obs_dim;
            const int rid = sm.rowid[s];
            if (rid >= 0) ra.state[(size_t)rid * md.obs_dim + d] = sm.state[i];
        }
    }
}

// ---- small helper kernels ---------------------------------------------------------------------------

// model_env.py:170-176: tile s0, zero the accumulators (EXACT mode state lives in HBM between steps)
__global__ void init_state_kernel(float* state, float* totals, unsigned char* term, const float* s0, int B, int obs_dim) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B * obs_dim) state[i] = s0[i % obs_dim];
    if (i < B) { totals[i] = 0.f; term[i] = 0; }
}

// model_env.py:190-191: total_rewards.reshape(-1, P).mean(dim=1)
__global__ void particle_mean_kernel(const float* totals, float* returns, int pop, int P) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= pop) return;
    float s = 0.f;
    for (int p = 0; p < P; ++p) s += totals[(size_t)c * P + p];
    returns[c] = s / (float)P;
}

// FAST-mode member schedule: per step a balanced random assignment of workgroups to member slots
// (every slot gets floor/ceil(nWG/M) workgroups -- the reference's "each model gets exactly the same
// number of samples", gaussian_mlp.py:267-275, at 16*R-row granularity).  fixed_model: one draw
// for all steps (TS-infinity).  One block per step.
__global__ void member_schedule_kernel(int* sched, int nwg, int M, int fixed, unsigned long long seed,
                                       unsigned long long stream_id) {
    const int t = blockIdx.x;
    const unsigned long long tk = fixed ? 0xFFFFFFFFull : (unsigned long long)t;
    const unsigned long long base = mix64(seed ^ mix64(stream_id * 0x9E3779B97F4A7C15ull + tk));
    for (int me = threadIdx.x; me < nwg; me += blockDim.x) {
        const unsigned long long kme = mix64(base + (unsigned long long)me);
        int rank = 0;
        for (int i = 0; i < nwg; ++i) {
            const unsigned long long ki = mix64(base + (unsigned long long)i);
            rank += (ki < kme) || (ki == kme && i < me);
        }
        sched[(size_t)t * nwg + me] = (int)(((long long)rank * M) / nwg);
    }
}

// export of the FAST-mode normals (hipets_fast_normals): out[t][rid][d]
__global__ void export_normals_kernel(float* out, int H, int B, int out_dim, unsigned long long seed,
                                      unsigned long long stream_id) {
    const int nblk = (out_dim + 3) / 4;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)H * B * nblk) return;
    const int blk = (int)(i % nblk);
    const int rid = (int)((i / nblk) % B);
    const int t = (int)(i / ((long long)nblk * B));
    float nrm[4];
    rollout_normals4(rid, t, blk, seed, stream_id, nrm);
    for (int q = 0; q < 4; ++q) {
        const int d = blk * 4 + q;
        if (d < out_dim) out[((size_t)t * B + rid) * out_dim + d] = nrm[q];
    }
}

// Re-pack [E, K, N] row-major weights of the active members into MFMA B-fragment order:
//   dst[m][l][c][kk][lane][s] = W_l[members[m]][16*kk + 4*(lane>>4) + s][16*c + (lane&15)]   (0 outside K x N)
__global__ void pack_weights_kernel(float* dst, const float* src, const int* members, int M, int K, int N, int Kp,
                                    int Np, long long member_stride, long long layer_off) {
    const long long per_member = (long long)Kp * Np;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= per_member * M) return;
    const int m = (int)(i / per_member);
    long long r = i % per_member;
    const int s = (int)(r & 3); r >>= 2;
    const int lane = (int)(r & 63); r >>= 6;
    const int KC = Kp / 16;
    const int kk = (int)(r % KC);
    const int c = (int)(r / KC);
    const int k = 16 * kk + 4 * (lane >> 4) + s;
    const int n = 16 * c + (lane & 15);
    float v = 0.f;
    if (k < K && n < N) v = src[((size_t)members[m] * K + k) * N + n];
    dst[(size_t)m * member_stride + layer_off + (i % per_member)] = v;
}

__global__ void pack_bias_kernel(float* dst, const float* src, const int* members, int M, int N, int Np, int member_stride,
                                 int layer_off) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M * Np) return;
    const int m = i / Np, n = i % Np;
    dst[(size_t)m * member_stride + layer_off + n] = n < N ? src[(size_t)members[m] * N + n] : 0.f;
}

}  // namespace hipets
