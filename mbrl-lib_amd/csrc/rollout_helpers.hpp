// rollout_helpers.hpp -- the small kernels around the rollout kernel (state init, particle mean, member schedules, exports of
// the device-side randomness, weight / bias packing).  Included by hipets.hip only (non-template kernels: one definition).
#pragma once
#include "rollout.hpp"

namespace hipets {

// ---- small helper kernels ---------------------------------------------------------------------------

// model_env.py:170-176: tile s0, zero the accumulators (EXACT mode state lives in HBM between steps).  pop_env > 0 (batched planning):
// row i belongs to candidate i / P, which plans for environment (i / P) / pop_env and starts from that environment's s0
__global__ void init_state_kernel(float* state, float* totals, unsigned char* term, const float* s0, int B, int obs_dim, int P, int pop_env) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B * obs_dim) {
        const int row = i / obs_dim, d = i - row * obs_dim;
        state[i] = s0[(pop_env > 0 ? (size_t)((row / P) / pop_env) * obs_dim : 0) + d];
    }
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

// Export of the FAST-mode member schedule (hipets_fast_schedule): sched[t][w] = the member slot workgroup w of nwg draws for step t
// in its own prologue (common.hpp fast_member: one keyed bijection of the workgroup indices per step, cut into M equal runs -- the
// reference's "each model gets exactly the same number of samples", gaussian_mlp.py:267-275, at 16 R-row granularity; fixed_model:
// one draw for all steps; BasicEnsemble, iid != 0: independent uniform draws, basic_ensemble.py:122-129).  (a, b) = perm_radices(nwg).
__global__ void member_schedule_kernel(int* sched, int nwg, unsigned a, unsigned b, int M, int fixed, int iid, unsigned long long seed,
                                       unsigned long long stream_id) {
    const int w = blockIdx.x * blockDim.x + threadIdx.x, t = blockIdx.y;
    if (w < nwg) sched[(size_t)t * nwg + w] = fast_member((unsigned)w, (unsigned)nwg, a, b, M, iid, seed, stream_id, fixed ? 0xFFFFFFFFu : (unsigned)t);
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

// export of the DEVICE-mode permutations (hipets_device_perms): out[t][j] = row that slot j holds at step t, i.e. the
// tensor the reference would have drawn with torch.randperm(B) at that step (int64 like torch)
__global__ void export_perms_kernel(long long* out, int H, unsigned n, unsigned a, unsigned b, int fixed, unsigned long long seed,
                                    unsigned long long stream_id) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)H * n) return;
    const int t = (int)(i / n);
    const unsigned j = (unsigned)(i % n);
    out[i] = (long long)perm_apply(j, n, a, b, perm_round_keys(perm_key(seed, stream_id, fixed ? 0xFFFFFFFFu : (unsigned)t)));
}

// round keys of the H per-step permutations of a DEVICE-mode rollout (persistent form reads them from memory)
// blockIdx.y: consecutive rollouts of one plan (stream ids stream_id, stream_id + 1, ...), key tables back to back
__global__ void step_keys_kernel(PermKeys* keys, int H, unsigned long long seed, unsigned long long stream_id) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < H) keys[(size_t)blockIdx.y * H + t] = perm_round_keys(perm_key(seed, stream_id + blockIdx.y, (unsigned)t));
}

// Re-pack [E, K, N] row-major weights of the active members into MFMA B-fragment order:
//   dst[m][l][c][kk][lane][s] = W_l[members[m]][16*kk + 4*s + (lane>>4)][16*c + colperm(lane&15)]   (0 outside K x N)
// so that k-step s of a chunk holds 4 CONSECUTIVE k (the tail chunk's all-padding steps can be skipped).
// src_nk != 0: the source is [E, N, K] row-major (nn.Linear's [out, in]) instead of [E, K, N] (EnsembleLinearLayer).
// permute_cols: 0 natural columns, 1 hidden layers (lds_col inside every 16), 2 the output layer's "head pair" order
// (head_pair_col(p, head_dim): means and log-variances of two output dims per lane group; N = 2 head_dim)
__global__ void pack_weights_kernel(float* dst, const float* src, const int* members, int M, int K, int N, int Kp,
                                    int Np, long long member_stride, long long layer_off, int permute_cols, int src_nk, int head_dim = 0) {
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
    const int k = 16 * kk + 4 * s + (lane >> 4);  // MFMA k-step s of a chunk covers k = 16 kk + 4 s + {0,1,2,3}
    // fragment row (lane & 15) = index m of the transposed product D^T[m][batch row]; hidden layers map it to the
    // real column lds_col(m) so that accumulator register i of lane group g lands on LDS position 4g + i
    int n = 16 * c + (permute_cols == 1 ? lds_col(lane & 15) : (lane & 15));
    if (permute_cols == 2) n = head_pair_col(n, head_dim);
    float v = 0.f;
    if (k < K && n >= 0 && n < N) v = src_nk ? src[((size_t)members[m] * N + n) * K + k] : src[((size_t)members[m] * K + k) * N + n];
    dst[(size_t)m * member_stride + layer_off + (i % per_member)] = v;
}

// bf16x3 precision mode: the same weights as three bf16 pieces, one A-operand fragment of v_mfma_f32_16x16x32_bf16 per
// (column tile c, 32-wide k chunk kk, piece p): dst unit (16 bytes) index ((c * KC32 + kk) * 3 + p) * 64 + lane holds
// piece p of W[members[m]][32 kk + 8 (lane >> 4) + j][16 c + (lane & 15)], j = 0..7 (zero outside K x N; natural columns)
__global__ void pack_weights_b3_kernel(uint4* dst, const float* src, const int* members, int M, int K, int N, int Kp32, int Np,
                                       long long member_stride, long long layer_off, int src_nk) {
    const int KC32 = Kp32 / 32, C = Np / 16;
    const long long per_member = (long long)C * KC32 * 3 * 64;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= per_member * M) return;
    const int m = (int)(i / per_member);
    long long r = i % per_member;
    const int lane = (int)(r & 63); r >>= 6;
    const int p = (int)(r % 3); r /= 3;
    const int kk = (int)(r % KC32);
    const int c = (int)(r / KC32);
    const int n = 16 * c + (lane & 15);
    unsigned w[4];
    for (int jj = 0; jj < 4; ++jj) {
        unsigned pair = 0;
        for (int h = 0; h < 2; ++h) {
            const int k = 32 * kk + 8 * (lane >> 4) + 2 * jj + h;
            float v = 0.f;
            if (k < K && n < N) v = src_nk ? src[((size_t)members[m] * N + n) * K + k] : src[((size_t)members[m] * K + k) * N + n];
            unsigned pc[3];
            split3(v, pc);
            pair |= h ? pc[p] : (pc[p] >> 16);
        }
        w[jj] = pair;
    }
    dst[(size_t)m * member_stride + layer_off + (i % per_member)] = make_uint4(w[0], w[1], w[2], w[3]);
}

__global__ void pack_bias_kernel(float* dst, const float* src, const int* members, int M, int N, int Np, int member_stride,
                                 int layer_off, int permute_cols, int head_dim = 0) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M * Np) return;
    const int m = i / Np, np_ = i % Np;
    int n = permute_cols == 1 ? lds_col(np_) : np_;  // same permutation as the weight columns
    if (permute_cols == 2) n = head_pair_col(np_, head_dim);
    dst[(size_t)m * member_stride + layer_off + np_] = (n >= 0 && n < N) ? src[(size_t)members[m] * N + n] : 0.f;
}

}  // namespace hipets
