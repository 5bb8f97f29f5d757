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
#include <type_traits>

#include "common.hpp"

namespace hipets {

struct LayerMeta {
    int Kp, Np;          // K, N padded to multiples of 16
    int boff;            // float offset of the layer's bias inside a member block
    int tail_steps;      // MFMA k-steps (of 4) of the last chunk that hold real weights: ceil((K - (Kp - 16)) / 4)
    long long woff;      // float offset of the layer's packed weights inside a member block
    // bf16x3 precision mode (fp32 operands as three bf16 pieces on the bf16 matrix pipe):
    int Kp32;            // K padded to a multiple of 32 (one v_mfma_f32_16x16x32_bf16 k-chunk)
    int pad_;
    long long woff3;     // 16-byte-unit offset of the layer's packed bf16 planes inside a member block
    // output layer of a stochastic model only: a SECOND pack of its weights / biases with the columns in "head pair" order
    // (head_pair_col below) for the kernel instances that sample straight from the accumulators (KSpec::FUSE); -1 = none
    long long woff_pairs;
    int boff_pairs;
    int pad2_;
};

// "Head pair" column order of the output layer (mean_and_logvar, gaussian_mlp.py:107-112): packed column p = 16 c + 4 g + i
// holds, for the output dim d = 8 c + 2 g + (i & 1), its mean (i < 2) or its log-variance (i >= 2).  Formed transposed, the
// product leaves lane group g of column tile c with {mean d, mean d+1, logvar d, logvar d+1} of one batch row in ONE
// accumulator: everything the sampling of those two dims needs (model.py:471-473), no LDS round trip.  -1 = zero padding.
__host__ __device__ __forceinline__ int head_pair_col(int p, int out_dim) {
    const int c = p >> 4, g = (p >> 2) & 3, i = p & 3;
    const int d = 8 * c + 2 * g + (i & 1);
    if (d >= out_dim) return -1;
    return i < 2 ? d : out_dim + d;
}

struct Extras {  // up to kMaxExtras leftover (column tile, row tile) units of one wave
    int c0, c1, c2, c3, r0, r1, r2, r3;
};

struct ModelDev {
    int obs_dim, act_dim, in_dim, out_dim, out_total, hid, n_layers, M;
    int obs_in;  // width of obs_process_fn(obs) = in_dim - act_dim
    int activation;
    float slope;
    int propagation, deterministic, obs_process, reward_fn, term_fn, target_is_delta, learned_rewards, normalizer;
    const LayerMeta* layers;  // DEVICE [n_layers] (a table in memory: runtime-indexed kernargs would go to scratch)
    int Kp0;                  // padded input width of layer 0
    int hidC;                 // column tiles of a hidden layer (cost model; shape of the lean kernel instances)
    int outC;                 // column tiles of the output layer
    long long wmember;  // floats per member (packed weights)
    int bmember;        // floats per member (padded biases)
    int ld;             // LDS activation row stride in floats (== 8 mod 64)
    int ld_in;          // KSpec::WIDE instances: row stride of the model-input image (>= Kp0, == 8 mod 64); the hidden activations use KSpec::LD
    const float* w;
    const float* b;
    const double* norm_mean;
    const double* norm_std;
    const float* min_lv;  // [lv_rows][out_dim]
    const float* max_lv;
    int lv_rows;          // 1 (bounds shared by the members) or M (BasicEnsemble: one row per member)
    int iid_members;      // BasicEnsemble: members are drawn independently (no balanced shuffle, no batch % M rule)
    const unsigned char* no_delta;  // [obs_dim]
    int precision;            // HIPETS_PREC_*
    long long w3member;       // 16-byte units per member (bf16x3 planes)
    const uint4* w3;          // packed bf16 planes: [member][layer][col tile][k chunk of 32][plane 0..2][lane][8 x bf16]
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
    unsigned perm_n, perm_a, perm_b;  // DEVICE mode: row of slot j = perm_apply(j) over [0, perm_n), radices a x b (perm_n = 0: none)
    PermKeys perm_keys;    // DEVICE mode: round keys of THIS launch's permutation (perm_round_keys(perm_key(seed, stream, step)), host side)
    const float* eps;      // [H,B,out] or null
    int use_philox;        // FAST without eps override
    unsigned long long seed, stream_id;
    const int* schedule;   // FAST: [H, nWG] member slot per (step, workgroup), injected by the caller; null = every workgroup draws its own
                           //   entries in its prologue (common.hpp fast_member, radices fm_a x fm_b = perm_radices(gridDim.x))
    unsigned fm_a, fm_b;
    int fast_members;      // the NON-fast kernel path (rows by identity / permutation, state in HBM around the launch) with the workgroup's
                           //   member chosen as in FAST mode (`schedule`, or the in-kernel draw for step t_begin): hipets_step in FAST mode --
                           //   for one step the per-step launch form IS the FAST form, and it exists in every shape-specialised instance
    float* trace_next_obs;
    float* trace_rewards;
    long long* phase_cycles;  // optional [kWaves][16 phases] cycle counters of workgroup 0 (profiling aid)
    int pop_env;               // FAST batched planning: candidates per environment (candidate c starts from s0[c / pop_env]); 0 = one env
    const float* init_states;  // FAST: optional per-row initial states [B,obs] (ModelEnv.step path) instead of tiling s0
    int write_back;            // FAST: also write the final state [B,obs] to `state` and the done flags to `term`
    int generic_only;          // hipets_rollout_opts.generic_kernel: 1 = only the fully generic kernel instance; 2 = no shape-specialised
                               // (lean) instance, but the hidden-static one (KSpec::HID_STATIC) where the model has its width
    int wide_lds;              // the host sized the LDS (and chose R) for the KSpec::WIDE layout: the launcher runs that instance or fails
    // DEVICE mode, persistent form (all workgroups co-resident, ONE launch for the horizon): rows change workgroups every step
    // through `exchange`, a [B][obs_dim + 2] table of 8-byte {value bits, step tag} granules (state dims, running total,
    // terminated flag).  A granule is written by ONE write-through (sc1) 8-byte store and polled with sc1 loads until its
    // tag is the awaited step: self-validating, so no fence, flag or grid barrier is involved (MI355X_MICROARCH.md R2).
    unsigned long long* exchange;  // null: per-step launches
    unsigned tag_base;             // step t's hand-over carries tag tag_base + t + 1 (the engine advances it by H per launch: no clearing)
    int* capacity_out;             // HOST pointer, launcher only: when set, no launch -- the resident capacity (workgroups) is stored here
    int n_logical;                 // persistent form: logical workgroups (member domain x row group); a launched workgroup serves the
                                   // logical ones wg, wg + gridDim.x, ... one after the other within every step (batches larger than the chip)
    int ragged_last_turn;          // persistent form, KSpec::WIDE two-tile instances: when the row tiles the LAST turn of a step would serve
                                   // fit one per launched workgroup, that turn is dealt in ONE-tile logical workgroups (rollout_kernel:
                                   // "ragged last turn"); 0 = always two-tile turns (A/B measurements: HIPETS_RAGGED_LAST_TURN=0)
    const PermKeys* step_keys;     // DEVICE [H]: round keys of every step's permutation
    int* error_flag;               // HOST-mapped: set to 1 when a poll exceeds its bound (another workgroup was not resident); once it is
                                   // set every later poll of the launch gives up after <= 64 spins, so a stranded grid drains in
                                   // milliseconds instead of waiting out the bound at every step and turn
    long long poll_ticks;          // bound of one hand-over poll in 100 MHz wall-clock ticks (hipets_set_handover_timeout; default 0.2 s)
    unsigned lds_bytes;            // the dynamic LDS size the launch was given (debug builds check every LDS section against it: HIPETS_DEBUG_BOUNDS)
    int* census;                   // DEVICE [2], launcher only: when set the launch is the co-residency SELF-TEST of this kernel instance at
                                   // this grid, not a rollout -- every workgroup arrives at census[0] and waits (bounded by poll_ticks)
                                   // until all gridDim.x have; those that saw everybody count themselves in census[1]
};

// D = A(16x4) * B(4x16) + C, exact f32.  Issued through inline asm with the accumulator tied in place
// ("+v"): with the builtin, hipcc's register allocator rotates the accumulators through fresh registers in
// the unrolled k loop and pays ~45 v_accvgpr_mov/read/write per iteration to undo it at the back edge.
// Hazards: A/B come from loads (the compiler's s_waitcnt covers asm inputs); back-to-back MFMAs that take
// the previous D whole as C need no wait states; the first non-MFMA reader of D is fenced by mfma_drain().
__device__ __forceinline__ void mfma16x16x4(const float a, const float b, f32x4& c) {
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
// >= 12 wait states between the last 8-pass MFMA and a VALU read of its result (cdna4 ISA, XDL write -> VALU read)
__device__ __forceinline__ void mfma_drain() { asm volatile("s_nop 15" ::: "memory"); }

// Column c of an activation row lives at LDS position lds_col(c): inside every 16-wide k chunk the 4x4 block
// (k-step s, lane group g) is stored transposed, so the lane group g of the A fragment reads its 4 k-steps
// {16kk + 4s + g : s = 0..3} with ONE ds_read_b128 at [16kk + 4g, +3].
__device__ __forceinline__ int lds_col(int c) { return (c & ~15) | ((c & 3) << 2) | ((c >> 2) & 3); }

// phase profiler: lane 0 of every wave of workgroup 0 accumulates s_memtime deltas per phase in LDS (a mark is one LDS
// read-modify-write on one lane, ~100 cycles; accumulating straight into global memory cost a ~800-cycle round trip per
// mark and dominated the short phases it measured) and flushes them into RolloutArgs::phase_cycles[wave][phase] at the end
// of the launch (a profiling aid, off unless the caller passes a buffer).  An accumulator holds the cycles in its low 44 bits and
// the NUMBER of marks that fed it above them (round 6): a reader divides both by the step count and can subtract what the marks
// themselves cost (profiles/one_tile_phase_profile.py calibrates that against the unprofiled launch duration).
constexpr int kProfCountShift = 44;
struct Prof {
    long long* slot;  // LDS: this wave's 16 accumulators
    long long t;
    bool on;
    __device__ __forceinline__ void mark(int phase) {
        if (on) {
            const long long now = clock64();
            slot[phase] += (now - t) + (1ll << kProfCountShift);
            t = now;
        }
    }
};

// One wave's share of a layer: CT strided column tiles (c_first + kWaves*ct) for all R row tiles, plus EX
// "extra" (column tile, row tile) units taken from the C % 4 leftover column tiles, all accumulated
// in the same k loop so the MFMA pipe always has >= 2 independent accumulators in flight.
// The k loop is software pipelined by hand with two register buffers: the B fragments (global, L2
// resident) and A fragments (LDS) of chunk kk+1 are in flight while the 4*(CT*R+EX) MFMAs of chunk kk
// issue (one wave per SIMD, so nothing else hides the load latency).
template <int R, int CT, int EX>
struct GemmFrags {
    f32x4 b[CT > 0 ? CT : 1];
    f32x4 bx[EX > 0 ? EX : 1];
    f32x4 a[R];
    f32x4 ax[EX > 0 ? EX : 1];
};

// Cross-layer prefetch (shape-specialised kernels): a wave's chunk-0 weight fragments and biases of the NEXT linear op are
// requested from L2 right after the current op's k loop, so that their ~800-cycle latency passes behind the epilogue
// (activation, LDS stores), the layer barrier and the next op's address set-up instead of stalling the first MFMA.
constexpr int kPreCT = 3;  // >= the column tiles a wave carries through one wave_gemm (kMaxCT in linear_op)
struct Pre {
    f32x4 b[kPreCT], bx[kMaxExtras];    // chunk-0 weight fragments of the strided / extra units
    f32x4 bv[kPreCT], bvx[kMaxExtras];  // their biases
};
struct NextOp {  // this wave's share of the op to prefetch for (wave-uniform)
    const float* W;
    const float* bias;
    int KC, c_first, ct, ex_n;  // ct strided column tiles from c_first, ex_n extra units
    int tail_steps;             // LayerMeta::tail_steps of the op
    Extras ex;
    bool valid;
};
// workgroup barrier for data exchanged through LDS only: waits for this wave's LDS traffic, NOT for global loads in flight
// (__syncthreads() also drains vmcnt, which would expose the latency of the prefetched weight fragments at every barrier)
// 16-byte write-through store / load of a PAIR of hand-over granules {value, tag, value, tag} (persistent DEVICE form).  sc1 =
// device scope: the store leaves the XCD's L2, the load never returns a stale L1 line (MI355X_MICROARCH.md: 8-byte sc1 stores
// cost 2.7x the 16-byte ones per byte, and a workgroup's polls queue behind its own stores).  The load is asynchronous:
// pair_wait() is the s_waitcnt, tied to the destination registers so nothing reads them earlier.
using u32x4g = __attribute__((ext_vector_type(4))) unsigned;
// (s_nop 1 behind the store: a VMEM store of more than 8 bytes reads its data registers late, and on gfx940+ a VALU write of one
// of them needs TWO wait states behind it.  The compiler keeps that distance for its own stores; inside an asm statement it does not
// know there is a store.  Round 5: after an unrelated change the next instruction but one rewrote the first data register, and under
// load -- two workgroups per CU -- quads of lanes published a scratch value instead of a state dim, or a corrupt tag that nobody could
// ever match: returns off in the 5th digit, once a time-out.  __graft_entry__.scan_isa_hazards checks the emitted code for it now.)
__device__ __forceinline__ void pair_store(unsigned long long* p, const unsigned v0, const unsigned v1, const unsigned tag) {
    const u32x4g d = {v0, tag, v1, tag};
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(d) : "memory");
}
__device__ __forceinline__ void pair_load_issue(u32x4g& d, const unsigned long long* p) {
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=&v"(d) : "v"(p) : "memory");
}
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// LDS-DMA of 64 hand-over pairs (one per lane, device-scope loads like pair_load_issue) straight into LDS: lane l's 16 bytes land at
// LDS byte address lds_dst + 16 l (lds_dst wave-uniform), no destination registers.  Counted on vmcnt; the compiler does not know
// (cdna_hip_programming.md: M0 is written and restored inside the statement that reads it; the s_nop is the SALU-write -> M0-read
// state).  Data is visible to a ds_read of the ISSUING wave after its own s_waitcnt vmcnt(0) (MI355X_MICROARCH.md item 7).
__device__ __forceinline__ void pair_dma_issue(const unsigned long long* gsrc, const unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off sc1\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void vmem_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void prefetch_issue(const NextOp& n, const int lane, Pre& pre) {
    if (!n.valid) return;
    const char* Wb = reinterpret_cast<const char*>(n.W);
    const int exc[kMaxExtras] = {n.ex.c0, n.ex.c1, n.ex.c2, n.ex.c3};
#pragma unroll
    for (int ct = 0; ct < kPreCT; ++ct)
        if (ct < n.ct) {
            pre.b[ct] = *reinterpret_cast<const f32x4*>(Wb + (size_t)(unsigned)(((n.c_first + kWaves * ct) * n.KC * 64 + lane) * 16));
            pre.bv[ct] = *reinterpret_cast<const f32x4*>(n.bias + (n.c_first + kWaves * ct) * 16 + 4 * (lane >> 4));
        }
#pragma unroll
    for (int e = 0; e < kMaxExtras; ++e)
        if (e < n.ex_n) {
            pre.bx[e] = *reinterpret_cast<const f32x4*>(Wb + (size_t)(unsigned)((exc[e] * n.KC * 64 + lane) * 16));
            pre.bvx[e] = *reinterpret_cast<const f32x4*>(n.bias + exc[e] * 16 + 4 * (lane >> 4));
        }
    __builtin_amdgcn_sched_barrier(0);  // keep the requests HERE (the scheduler would sink them to their first use)
}

// Minimum waves per SIMD the register allocation must leave room for (= workgroups of 4 waves per CU).  R <= 2 keeps two
// workgroups per CU resident (their barrier / latency phases overlap); R = 3, 4 need the registers.
#ifndef HIPETS_MINWAVES_R1
#define HIPETS_MINWAVES_R1 2
#endif
#ifndef HIPETS_MINWAVES_R2
#define HIPETS_MINWAVES_R2 2
#endif
template <int R> struct MinWavesOf { static constexpr int value = R == 1 ? HIPETS_MINWAVES_R1 : (R == 2 ? HIPETS_MINWAVES_R2 : 1); };

// Debug build (__graft_entry__.build_debug: -O1 -g -DHIPETS_DEBUG_BOUNDS=1, host side under AddressSanitizer): every LDS section of
// the rollout kernel is checked against the dynamic LDS size of the launch, and the indexed LDS accesses of the elementwise phases
// against their section.  A violated bound aborts the kernel (device assert -> the next HIP call reports it).  Off in the shipped
// library: the checks cost registers in kernels that sit at the limit.
#ifndef HIPETS_DEBUG_BOUNDS
#define HIPETS_DEBUG_BOUNDS 0
#endif
#if HIPETS_DEBUG_BOUNDS
// (not <cassert>'s assert: the generic lambdas of wave_gemm are implicitly __host__ __device__, where the host's __assert_fail is not callable)
__host__ __device__ inline void hipets_bound_fail(const int line) {
#if defined(__HIP_DEVICE_COMPILE__)
    printf("hipets: bound violated at rollout.hpp:%d (workgroup %d, thread %d)\n", line, (int)blockIdx.x, (int)threadIdx.x);
    __builtin_trap();
#else
    (void)line;
#endif
}
#define HIPETS_BOUND(cond) do { if (!(cond)) hipets_bound_fail(__LINE__); } while (0)
#else
#define HIPETS_BOUND(cond) ((void)0)
#endif

struct NoTail {};  // wave_gemm's TL: the ordinary epilogue (activation, store as the next op's LDS image)
// A fused tail = four stages over the accumulators (units) a wave finished: prep(slot, c, r) for EVERY unit of a group first -- it
// only loads (LDS) what the unit will need into its slot, so the round trips of all units overlap --, then draw(slot a, slot b, c_a,
// c_b, two) for every PAIR of units (the pair's standard normals: one Philox block per lane for the two units together, see
// rollout_kernel's tail_draw), then unit(slot, acc, c, r) for every unit (arithmetic + stores), then finish() once per wave.
struct FusedSlot {  // what one unit's lane reads from LDS (rollout_kernel, KSpec::FUSE)
    float mxA, mxB, mnA, mnB, pA, pB;
    double nmA, nmB, nsA, nsB;
    int rid, ndA, ndB;
    float n0, n1;  // the two standard normals of the lane's dims (draw stage)
};
template <class P, class D, class F, class G>
struct TailStages {
    P prep;
    D draw;
    F unit;
    G finish;
};
template <class P, class D, class F, class G>
__device__ __forceinline__ TailStages<P, D, F, G> make_tail(P p, D d, F f, G g) { return TailStages<P, D, F, G>{p, d, f, g}; }

// ACT >= 0: the activation is a compile-time fact (one epilogue in the code); ACT < 0: `act` selects it at run time.
// PRE: chunk 0's weight fragments and the biases are already in `pre` (prefetch_issue by the previous op); after the k loop
// the next op's are requested into `pre` again (`nxt`).
// TL != NoTail: instead of the epilogue every finished accumulator is handed to (*tl)(acc, column tile, row tile) -- the fused
// per-step tail of the output layer (KSpec::FUSE: sampling, next state, reward, next input straight from the registers).
// LD > 0: the LDS row stride is a compile-time fact (shape-specialised instances): the A-fragment reads of the R row tiles become
// ONE base register + immediate offsets (ds_read_b128 ... offset:r * 16 * LD * 4), no per-row address arithmetic in the k loop.
#ifndef HIPETS_BUFFER_LOADS
#define HIPETS_BUFFER_LOADS 1  // weight fragments through buffer_load ... s_off offen: the chunk offset rides in an SGPR
#endif
#ifndef HIPETS_KSTEP_NOP
#define HIPETS_KSTEP_NOP 1  // s_nop 1 in front of every k-step (hazard guard, see kstep below)
#endif
// SPL: every unit sums its even and its odd k-steps in two accumulators and adds them at the end -- the order a wave whose whole
// share is ONE unit uses anyway (hazard (2) below).  The OUTPUT layer runs with SPL in every instance: its columns are dealt to
// the waves differently by the natural and the head-pair packs, and with SPL a column's sum does not depend on whether its wave
// holds one unit or several -- shape-specialised and generic instances keep returning the same bits.
// KCS > 0: the number of k chunks is a compile-time fact (ops whose K is the hidden width of a shape-specialised instance): the
// k loop is fully unrolled -- straight-line code, no loop control, no accumulator copies where blocks meet.
#ifndef HIPETS_UNROLL_K
#define HIPETS_UNROLL_K 1
#endif
#ifndef HIPETS_INTERLEAVE
#define HIPETS_INTERLEAVE 1  // the next chunk's fragment loads inside the MFMAs' shadows (compute_il in wave_gemm)
#endif
#ifndef HIPETS_SHARED_DRAWS
#define HIPETS_SHARED_DRAWS 1  // fused tail: one Philox block per lane and PAIR of units (rollout_kernel tail_draw); 0 = every lane computes the whole block of every unit (A/B measurements)
#endif
#ifndef HIPETS_KS_TRIPLE
#define HIPETS_KS_TRIPLE 1  // k-split (one-tile) instances: fragments fetched TWO chunks ahead (three register sets, wave_gemm kTriple)
#endif
// x * rcp(1 + exp2(-x log2 e)) on the 4 accumulator values of a lane: the two multiplies and the add as packed 2 x f32 ops
__device__ __forceinline__ f32x4 silu4(const f32x4 a) {
    using f32x2 = __attribute__((ext_vector_type(2))) float;
    const f32x2 k = {-1.44269504088896340736f, -1.44269504088896340736f}, one = {1.0f, 1.0f};
    const f32x2 lo = {a[0], a[1]}, hi = {a[2], a[3]};
    f32x2 tl = lo * k, th = hi * k;
    tl[0] = __builtin_amdgcn_exp2f(tl[0]); tl[1] = __builtin_amdgcn_exp2f(tl[1]);
    th[0] = __builtin_amdgcn_exp2f(th[0]); th[1] = __builtin_amdgcn_exp2f(th[1]);
    tl = tl + one; th = th + one;
    tl[0] = __builtin_amdgcn_rcpf(tl[0]); tl[1] = __builtin_amdgcn_rcpf(tl[1]);
    th[0] = __builtin_amdgcn_rcpf(th[0]); th[1] = __builtin_amdgcn_rcpf(th[1]);
    const f32x2 yl = lo * tl, yh = hi * th;
    return f32x4{yl[0], yl[1], yh[0], yh[1]};
}

// K-SPLIT of the leftover column tile (one-tile workgroups, round 5; KSpec::KSPLIT).  A hidden layer of 13 column tiles deals 4-3-3-3
// tiles to the four waves: the wave with four sets the pace of every layer (200 of its 4 x 50 MFMA k-steps against 150 of the
// others), and at R = 1 nothing else runs on the CU.  Instead the 13th tile's K RANGE is dealt to the waves -- wave w takes the k
// chunks [w KC / 4, (w + 1) KC / 4) of it, at most kKsSlots -- so every wave issues 3 x 50 + 16 k-steps, and the four partial sums
// meet LAZILY: a wave leaves its partial (pre-activation; wave 0's starts at the bias) in LDS as the f32x4 its lanes hold -- which,
// formed transposed, is exactly the B-operand fragment layout of the NEXT op's last k chunk (`lds_col`) -- and after the layer's
// ordinary barrier every wave of the next op reads the four partials of its lane, adds them in one fixed order ((P0 + P1) + (P2 + P3))
// and applies the activation: that IS its fragment of the last chunk.  No extra barrier, no extra pass; 4 ds_read_b128 + ~20 VALU
// instructions per wave and layer against 34 k-steps (~1.1 k cycles) fewer on the critical wave.  The hidden columns 192..207 are
// summed in another order than in the other instances: KSPLIT instances agree with them to rounding (tests: T2 against the oracle),
// not bit for bit.  Two partial buffers alternate by layer parity (a fast wave may finish layer l + 1 while a slow one still reads
// layer l's partials).
constexpr int kKsSlots = 4;  // k chunks of the split tile per wave (KC <= 16: hidden widths up to 256, inputs up to 256 columns)
struct KsArgs {
    const float* part_in;  // KSI: [kWaves][64][4] the producer's partial sums of this op's LAST k chunk
    float* part_out;       // KSO: [kWaves][64][4] this op's partial sums of its split column tile
    int tile;              // KSO: the split column tile (the op's last)
    int k0, n;             // KSO: this wave's chunks [k0, k0 + n) of it
    int wave;
};

// KS bit 0 (KSI): the input image's last k chunk is NOT in LDS -- it is rebuilt from ks->part_in; bit 1 (KSO): see above;
// bit 2: no k-split, only the one-tile k loop that fetches two chunks ahead (kTriple: ops with a static chunk count, planet.hpp)
template <int R, int CT, int EX, int ACT, bool PRE = false, class TL = NoTail, int LD = -1, bool SPL = false, int KCS = -1, int KS = 0>
__device__ __forceinline__ void wave_gemm(const float* __restrict__ in, float* __restrict__ out, const int ld_rt,
                                          const float* __restrict__ W, const float* __restrict__ bias, const int KC_rt,
                                          const int tail_steps, const int c_first, const Extras ex,
                                          const bool apply_act, const int act, const float slope, const int lane,
                                          Prof& prof, Pre* pre = nullptr, const NextOp* nxt = nullptr, const TL* tl = nullptr,
                                          const int ldi_rt = 0, const KsArgs* ks = nullptr) {
    constexpr int CTn = CT > 0 ? CT : 1;
    constexpr int EXn = EX > 0 ? EX : 1;
    constexpr bool KSI = (KS & 1) != 0, KSO = (KS & 2) != 0;
    static_assert(!KS || (R == 1 && LD > 0 && !PRE && HIPETS_BUFFER_LOADS), "k-split: one-tile shape-specialised instances, rolled k loop");
    static_assert(!KSO || (std::is_same<TL, NoTail>::value && EX == 0 && !SPL), "k-split producer: a hidden op");
    f32x4 acc[CTn][R];
    f32x4 accx[EXn];
    const int ld = LD > 0 ? LD : ld_rt;
    const int ldi = ldi_rt > 0 ? ldi_rt : ld;  // row stride of `in` when it differs from the output's (KSpec::WIDE: the model-input image)
    const int KC = KCS > 0 ? KCS : KC_rt;

    const int exc[kMaxExtras] = {ex.c0, ex.c1, ex.c2, ex.c3};
    const int exr[kMaxExtras] = {ex.r0, ex.r1, ex.r2, ex.r3};
    // per-lane BYTE offsets from the (wave-uniform) chunk base W + 256 kk floats: loop invariant, unsigned 32 bit, so the
    // loads take the scalar-base form (global_load v, v_off, s[base]) and the k loop carries no 64-bit address VALU work
    // (a wave's own VALU instructions do not overlap its MFMAs, profiles/microbench)
    unsigned woff[CTn], wxoff[EXn];
    int axoff[EXn];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) woff[ct] = (unsigned)(((c_first + kWaves * ct) * KC * 64 + lane) * 16);
#pragma unroll
    for (int e = 0; e < EX; ++e) {
        wxoff[e] = (unsigned)((exc[e] * KC * 64 + lane) * 16);
        axoff[e] = exr[e] * 16 * ldi;
    }
    const char* Wb = reinterpret_cast<const char*>(W);
    const float* ap = in + (lane & 15) * ldi + 4 * (lane >> 4);
    // biases of this lane's columns: loaded before the k loop so their latency hides behind it
    f32x4 bv[CTn], bvx[EXn];
    if constexpr (PRE) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) bv[ct] = pre->bv[ct];
#pragma unroll
        for (int e = 0; e < EX; ++e) bvx[e] = pre->bvx[e];
    } else {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) bv[ct] = *reinterpret_cast<const f32x4*>(bias + (c_first + kWaves * ct) * 16 + 4 * (lane >> 4));
#pragma unroll
        for (int e = 0; e < EX; ++e) bvx[e] = *reinterpret_cast<const f32x4*>(bias + exc[e] * 16 + 4 * (lane >> 4));
    }
#if HIPETS_BUFFER_LOADS
    // The weight block of this op as a raw buffer (base = W, wave-uniform): a fragment load is buffer_load_dwordx4 v, v_off, s[rsrc],
    // s_chunk offen -- the loop-invariant per-lane offset in a VGPR, the chunk offset (kk KiB) in an SGPR, NO address VALU work in
    // the k loop (fp32 MFMAs and VALU instructions exclude each other on a SIMD: every v_lshl_add_u64 there is MFMA-pipe idle time)
    // (W is wave-uniform by construction -- member and layer are -- but parts of it came through LDS, which the compiler's divergence
    // analysis cannot see: without the readfirstlane it wraps every buffer_load in a waterfall loop)
    const unsigned long long wbits = reinterpret_cast<unsigned long long>(W);
    const unsigned long long wuni = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(wbits >> 32)) << 32) |
                                    (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)wbits);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(wuni), 0, 0x7FFFFFFF, 0x00020000);
    auto wload = [&](const unsigned voff, const int kk) __attribute__((always_inline)) {
        const u32x4g v = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, (int)voff, kk * 1024, 0);
        f32x4 r;
        __builtin_memcpy(&r, &v, 16);
        return r;
    };
#endif
    auto load = [&](GemmFrags<R, CT, EX>& f, const int kk) __attribute__((always_inline)) {
#if HIPETS_BUFFER_LOADS
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) f.b[ct] = wload(woff[ct], kk);
#pragma unroll
        for (int e = 0; e < EX; ++e) f.bx[e] = wload(wxoff[e], kk);
#else
        const char* Wk = Wb + (size_t)kk * 1024;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) f.b[ct] = *reinterpret_cast<const f32x4*>(Wk + woff[ct]);
#pragma unroll
        for (int e = 0; e < EX; ++e) f.bx[e] = *reinterpret_cast<const f32x4*>(Wk + wxoff[e]);
#endif
#pragma unroll
        for (int r = 0; r < R; ++r) f.a[r] = *reinterpret_cast<const f32x4*>(ap + r * 16 * ldi + kk * 16);
#pragma unroll
        for (int e = 0; e < EX; ++e) f.ax[e] = *reinterpret_cast<const f32x4*>(ap + axoff[e] + kk * 16);
    };
    // Hazards the compiler cannot see inside asm: (1) a VALU write (e.g. a phi copy of an accumulator) must be
    // >= 2 wait states ahead of the MFMA that reads it -> s_nop 1 opens every k-step; (2) an MFMA that takes the
    // previous MFMA's D as C back-to-back (issue interval 32 < dependent latency 40 cycles) reads a stale C on
    // VGPR accumulators -> a wave whose whole share is ONE unit alternates two accumulators (even / odd k-steps).
    constexpr bool kSplit = (CT * R + EX) == 1;
    constexpr bool kSplitAll = SPL && !kSplit;
    f32x4 acc_odd = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 acco[CTn][R], accxo[EXn];  // kSplitAll: the odd k-steps of every unit
#pragma unroll
    for (int ct = 0; ct < CTn; ++ct)
#pragma unroll
        for (int r = 0; r < R; ++r) acco[ct][r] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < EXn; ++e) accxo[e] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto kstep = [&](const GemmFrags<R, CT, EX>& f, const int s) __attribute__((always_inline)) {
#if HIPETS_KSTEP_NOP
        asm volatile("s_nop 1");
#else
        if (s == 0) asm volatile("s_nop 1");  // experiment: the guard once per chunk (after the fragment loads' register writes) only
#endif
        if constexpr (kSplit) {
            f32x4& dst = (s & 1) ? acc_odd : (CT ? acc[0][0] : accx[0]);
            if constexpr (CT) mfma16x16x4(f.b[0][s], f.a[0][s], dst);
            else mfma16x16x4(f.bx[0][s], f.ax[0][s], dst);
        } else if constexpr (kSplitAll) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int r = 0; r < R; ++r) mfma16x16x4(f.b[ct][s], f.a[r][s], (s & 1) ? acco[ct][r] : acc[ct][r]);
#pragma unroll
            for (int e = 0; e < EX; ++e) mfma16x16x4(f.bx[e][s], f.ax[e][s], (s & 1) ? accxo[e] : accx[e]);
        } else {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int r = 0; r < R; ++r) mfma16x16x4(f.b[ct][s], f.a[r][s], acc[ct][r]);
#pragma unroll
            for (int e = 0; e < EX; ++e) mfma16x16x4(f.bx[e][s], f.ax[e][s], accx[e]);
        }
    };
    auto compute = [&](const GemmFrags<R, CT, EX>& f) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < 4; ++s) kstep(f, s);
    };
    // Interleaved form of "load the next chunk, then compute this one" (HIPETS_INTERLEAVE): the kNL fragment loads of chunk
    // kk_next are issued ONE AT A TIME, evenly spread behind the MFMAs of the current chunk, instead of as a clump in front of it.
    // Measured stand-alone (profiles/microbench/kloop_probe.hip, this wave's 3 x 3 + 1 tiling, 220 workgroups): a VMEM / LDS
    // instruction issued while no MFMA is executing costs ~12 cycles of matrix-pipe idle time (8 per 40 MFMAs: 34.46 cycles per
    // MFMA); issued inside an MFMA's 32-cycle shadow it is free (32.98).  Weight fragments first: they have the L2 round trip
    // ahead of them and are needed >= 30 MFMAs (~1 000 cycles) later; the LDS fragments follow in the order the next chunk's first
    // MFMAs consume them.  sched_barrier(0) on both sides pins each load where it is written.
    constexpr bool kIL = HIPETS_INTERLEAVE && LD > 0;  // shape-specialised instances only: in the generic ones (every shape x activation in
                                                        // one kernel, at the 256-VGPR limit) the longer live ranges spill 16-20 VGPRs to scratch
    constexpr int kNU = CT * R + EX;                              // MFMA units of this wave
    constexpr int kNL = CT + EX + (CT > 0 ? R : 0) + EX;          // fragment loads per chunk
    static_assert(4 * kNU >= kNL + 1, "every load needs its own slot behind an MFMA");
    auto load_one = [&](GemmFrags<R, CT, EX>& g, const int kk, const int i) __attribute__((always_inline)) {
        __builtin_amdgcn_sched_barrier(0);
#if HIPETS_BUFFER_LOADS
        if (i < CT) g.b[i < CT ? i : 0] = wload(woff[i < CT ? i : 0], kk);
        else if (i < CT + EX) g.bx[i - CT] = wload(wxoff[i - CT], kk);
#else
        if (i < CT) g.b[i < CT ? i : 0] = *reinterpret_cast<const f32x4*>(Wb + (size_t)kk * 1024 + woff[i < CT ? i : 0]);
        else if (i < CT + EX) g.bx[i - CT] = *reinterpret_cast<const f32x4*>(Wb + (size_t)kk * 1024 + wxoff[i - CT]);
#endif
        else if (CT > 0 && i < CT + EX + R) g.a[i - CT - EX] = *reinterpret_cast<const f32x4*>(ap + (i - CT - EX) * 16 * ldi + kk * 16);
        else {
            const int e = i - CT - EX - (CT > 0 ? R : 0);
            g.ax[e] = *reinterpret_cast<const f32x4*>(ap + axoff[e] + kk * 16);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto mfma_unit = [&](const GemmFrags<R, CT, EX>& f, const int s, const int u) __attribute__((always_inline)) {
        if constexpr (kSplit) {
            f32x4& dst = (s & 1) ? acc_odd : (CT ? acc[0][0] : accx[0]);
            if constexpr (CT) mfma16x16x4(f.b[0][s], f.a[0][s], dst);
            else mfma16x16x4(f.bx[0][s], f.ax[0][s], dst);
        } else if (u < CT * R) {
            const int ct = u / R, r = u - ct * R;
            if constexpr (kSplitAll) mfma16x16x4(f.b[ct][s], f.a[r][s], (s & 1) ? acco[ct][r] : acc[ct][r]);
            else mfma16x16x4(f.b[ct][s], f.a[r][s], acc[ct][r]);
        } else {
            const int e = u - CT * R;
            if constexpr (kSplitAll) mfma16x16x4(f.bx[e][s], f.ax[e][s], (s & 1) ? accxo[e] : accx[e]);
            else mfma16x16x4(f.bx[e][s], f.ax[e][s], accx[e]);
        }
    };
    // after MFMA number m1 (1-based) of the chunk: the loads whose slot this is.  Load j goes behind MFMA (j + 1) * total / (kNL + 1):
    // evenly spread, the last one still several MFMAs ahead of the chunk's end (the next chunk's first MFMAs want its data).
    // Plain nested loops with compile-time bounds and an explicit `#pragma unroll` each: every array index must be a constant
    // after unrolling (a dynamically indexed fragment array is demoted to scratch memory -- measured: 28 ms per rollout).
    auto loads_behind = [&](GemmFrags<R, CT, EX>& g, const int kk_next, const int m1) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < kNL; ++j)
            if (((j + 1) * 4 * kNU) / (kNL + 1) == m1) load_one(g, kk_next, j);
    };
    auto compute_il = [&](const GemmFrags<R, CT, EX>& f, GemmFrags<R, CT, EX>& g, const int kk_next) __attribute__((always_inline)) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            asm volatile("s_nop 1");  // hazard guard of kstep above
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    mfma_unit(f, ks, ct * R + r);
                    loads_behind(g, kk_next, ks * kNU + ct * R + r + 1);
                }
#pragma unroll
            for (int e = 0; e < EX; ++e) {
                mfma_unit(f, ks, CT * R + e);
                loads_behind(g, kk_next, ks * kNU + CT * R + e + 1);
            }
        }
    };
    // The compiler models an asm MFMA as an ordinary instruction whose result is ready immediately, so any VALU
    // copy of an accumulator it places right behind one (phi copies where control flow merges) would read the
    // register before the matrix pipe has written it.  drain_all() = wait out the pipe, then re-define every
    // accumulator through an empty asm so such copies can only be scheduled after the wait.  It ends every
    // conditional arm below and follows the main loop; the loop body itself is branch-free and in place.
    auto drain_all = [&]() __attribute__((always_inline)) {
        mfma_drain();
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int r = 0; r < R; ++r) asm volatile("" : "+v"(acc[ct][r]));
#pragma unroll
        for (int e = 0; e < EX; ++e) asm volatile("" : "+v"(accx[e]));
        asm volatile("" : "+v"(acc_odd));
        if constexpr (kSplitAll) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int r = 0; r < R; ++r) asm volatile("" : "+v"(acco[ct][r]));
#pragma unroll
            for (int e = 0; e < EX; ++e) asm volatile("" : "+v"(accxo[e]));
        }
    };
    // last chunk: only the k-steps that hold real (non-padding) weights, e.g. 2 of 4 for K = 200
    auto compute_tail = [&](const GemmFrags<R, CT, EX>& f) __attribute__((always_inline)) {
        switch (tail_steps) {
            case 1: kstep(f, 0); drain_all(); break;
            case 2: kstep(f, 0); kstep(f, 1); drain_all(); break;
            case 3: kstep(f, 0); kstep(f, 1); kstep(f, 2); drain_all(); break;
            default: kstep(f, 0); kstep(f, 1); kstep(f, 2); kstep(f, 3); drain_all(); break;
        }
    };

    // sched_barrier(0) pins "issue the next chunk's loads, THEN this chunk's MFMAs": without it the machine
    // scheduler sinks each load group down to its first use and the pipeline degenerates to load->wait->compute.
    // k-split: the partials of the input's last chunk (KSI) and this wave's share of the split tile (KSO: its weight and activation
    // fragments, ALL requested up front -- unused slots read chunk 0 and are zeroed below: straight-line code, no branch around an MFMA)
    f32x4 ks_p[kWaves], ks_b[kKsSlots], ks_a[kKsSlots], ks_bias = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (KSI) {
#pragma unroll
        for (int w = 0; w < kWaves; ++w) ks_p[w] = *reinterpret_cast<const f32x4*>(ks->part_in + (w * 64 + lane) * 4);
    }
    if constexpr (KSO) {
        const unsigned xoff = (unsigned)((ks->tile * KC * 64 + lane) * 16);
        ks_bias = *reinterpret_cast<const f32x4*>(bias + ks->tile * 16 + 4 * (lane >> 4));
#pragma unroll
        for (int j = 0; j < kKsSlots; ++j) {
            const int c = j < ks->n ? ks->k0 + j : 0;
            ks_b[j] = wload(xoff, c);
            ks_a[j] = *reinterpret_cast<const f32x4*>(ap + c * 16);
        }
    }
    GemmFrags<R, CT, EX> f0, f1;
    if constexpr (PRE) {  // chunk 0: weights are in registers already, only the activation fragments come from LDS
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) f0.b[ct] = pre->b[ct];
#pragma unroll
        for (int e = 0; e < EX; ++e) f0.bx[e] = pre->bx[e];
#pragma unroll
        for (int r = 0; r < R; ++r) f0.a[r] = *reinterpret_cast<const f32x4*>(ap + r * 16 * ldi);
#pragma unroll
        for (int e = 0; e < EX; ++e) f0.ax[e] = *reinterpret_cast<const f32x4*>(ap + axoff[e]);
    } else {
        load(f0, 0);
    }
    // One-tile k-split instances fetch TWO chunks ahead (kTriple below): a wave's 12 MFMAs per chunk (384 cycles) are no cover for an
    // L2 round trip issued somewhere inside the previous chunk
    // (ops whose chunk count is a compile-time fact -- KCS: everything fed by a hidden layer -- so that the loop's remainder is no run-time
    // branch: the allocator copies accumulators where such arms begin and sinks the copies to just in front of their first MFMA)
    constexpr bool kTriple = HIPETS_KS_TRIPLE && KS != 0 && KCS > 0 && HIPETS_INTERLEAVE && LD > 0;
    static_assert(!KS || KCS <= 0 || kTriple, "k-split ops with a static chunk count run the three-set loop");
    if constexpr (kTriple) load(f1, KCS > 1 ? 1 : 0);
    // accumulators start at the bias (C input of the first MFMA) instead of zero: no add in the epilogue.  Initialised AFTER
    // chunk 0's fragment loads were issued: the bias loads are older, so waiting for them leaves the fragments in flight
    // (initialising first serialised two L2 round trips per layer: ~1.2k cycles of "set-up" per layer in the phase profile)
#pragma unroll
    for (int ct = 0; ct < CTn; ++ct)
#pragma unroll
        for (int r = 0; r < R; ++r) acc[ct][r] = CT > 0 ? bv[ct] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < EXn; ++e) accx[e] = EX > 0 ? bvx[e] : f32x4{0.f, 0.f, 0.f, 0.f};
    // Pin every accumulator's initial value HERE: to the compiler an asm MFMA is an ordinary reader of its C operand, so it may
    // sink the (VALU) initialisation -- a copy of the bias, the zeros of the odd-k-step accumulators -- down to just in front of
    // the first MFMA that uses the register, inside a k-step, behind that k-step's s_nop: a VALU write followed at once by an MFMA
    // reading it as SrcC (hazard (1) below; found in the ISA of the cfg4 instances by __graft_entry__.scan_isa_hazards (tests/test_abi.py), where it returned
    // wrong sums).  An empty asm that "modifies" the register makes the value opaque: it must be complete before this point.
    auto pin = [](f32x4& v) __attribute__((always_inline)) { asm volatile("" : "+v"(v)); };
#pragma unroll
    for (int ct = 0; ct < CTn; ++ct)
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if constexpr (CT > 0) pin(acc[ct][r]);
            if constexpr (CT > 0 && kSplitAll) pin(acco[ct][r]);
        }
#pragma unroll
    for (int e = 0; e < EXn; ++e) {
        if constexpr (EX > 0) pin(accx[e]);
        if constexpr (EX > 0 && kSplitAll) pin(accxo[e]);
    }
    if constexpr (kSplit) pin(acc_odd);
    prof.mark(14);
    f32x4 a_last = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (KSI) {  // this lane's fragment of the input's last chunk: the activation of the four partial sums, in ONE fixed order
        static_assert(ACT == HIPETS_ACT_SILU, "k-split instances are SiLU instances");
        a_last = silu4((ks_p[0] + ks_p[1]) + (ks_p[2] + ks_p[3]));
    }
    f32x4 ks_e = f32x4{0.f, 0.f, 0.f, 0.f}, ks_o = f32x4{0.f, 0.f, 0.f, 0.f};  // even / odd k-steps of the split tile (hazard (2) above)
    if constexpr (KSO) {
        // Wave-uniform choices as ARITHMETIC (a multiply by 1.0f or 0.0f from an SGPR: exact on finite values), never as control flow:
        // written as `if`s the compiler built a web of ~40 scalar branches around these 20 register writes
        ks_e = ks_bias * (ks->wave == 0 ? 1.0f : 0.0f);  // the sum of the four partials carries the bias once
#pragma unroll
        for (int j = 0; j < kKsSlots; ++j) ks_b[j] = ks_b[j] * (j < ks->n ? 1.0f : 0.0f);  // unused slot: 0 x (a valid activation of chunk 0)
        if constexpr (KSI) {
            // the input's last chunk lives in registers (a_last), not in LDS.  In an op fed by a hidden layer it is the LAST slot of the
            // LAST wave (KC = 13 .. 16: wave 3 holds chunks [3 KC / 4, KC), four of them); on the other waves that slot is unused (zero
            // weights), so it may hold the same finite values there: no selection at all
            HIPETS_BOUND(KC >= 13 && KC <= 16);
            ks_a[kKsSlots - 1] = a_last;
        }
        pin(ks_e);
        pin(ks_o);
#pragma unroll
        for (int j = 0; j < kKsSlots; ++j) {
            pin(ks_b[j]);
            pin(ks_a[j]);
        }
#pragma unroll
        for (int j = 0; j < kKsSlots; ++j)
#pragma unroll
            for (int s_ = 0; s_ < 4; ++s_) {
                asm volatile("s_nop 1");
                mfma16x16x4(ks_b[j][s_], ks_a[j][s_], (s_ & 1) ? ks_o : ks_e);
            }
        prof.mark(7);  // (profiling builds) the k-split share: its loads' round trip + 16 MFMAs
    }
    if constexpr (KCS > 0 && !kTriple) {
        constexpr int kEnd = KCS >= 2 ? ((KCS - 1) / 2) * 2 : 0;  // the loop below leaves kk at the smallest even number >= KCS - 2
#pragma unroll
        for (int kk = 0; kk + 2 < KCS; kk += 2) {
            if constexpr (kIL) {
                compute_il(f0, f1, kk + 1);
                compute_il(f1, f0, kk + 2);
            } else {
                load(f1, kk + 1);
                __builtin_amdgcn_sched_barrier(0);
                compute(f0);
                __builtin_amdgcn_sched_barrier(0);
                load(f0, kk + 2);
                __builtin_amdgcn_sched_barrier(0);
                compute(f1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if constexpr (kEnd + 1 < KCS) {
            if constexpr (kIL) {
                compute_il(f0, f1, kEnd + 1);
            } else {
                load(f1, kEnd + 1);
                __builtin_amdgcn_sched_barrier(0);
                compute(f0);
                __builtin_amdgcn_sched_barrier(0);
            }
            compute_tail(f1);
        } else {
            compute_tail(f0);
        }
    } else if constexpr (kTriple) {
        // three fragment sets in rotation: chunk kk is computed from one while chunk kk + 2 is being fetched into another (measured
        // on MI355X, one-tile workgroups, round 5: with one chunk of lead the k loop of a wave with 3 column tiles ran at the pace of
        // the L2 round trips, not of its MFMAs -- profiles/r5_small_batches.json)
        GemmFrags<R, CT, EX> f2;
        auto last_frag = [&](GemmFrags<R, CT, EX>& f) __attribute__((always_inline)) {  // KSI: see the two-set loop below
            if constexpr (KSI) {
                if constexpr (CT > 0) f.a[0] = a_last;
#pragma unroll
                for (int e = 0; e < EX; ++e) f.ax[e] = a_last;
                if constexpr (CT > 0) pin(f.a[0]);
#pragma unroll
                for (int e = 0; e < EX; ++e) pin(f.ax[e]);
            }
        };
        constexpr int kEnd3 = KCS >= 3 ? ((KCS - 1) / 3) * 3 : 0;  // where the loop below leaves kk
        constexpr int kRem = KCS - kEnd3;                          // 1 .. 3 chunks left then, the last of them the tail chunk
#pragma nounroll
        for (int kk = 0; kk + 3 < KCS; kk += 3) {  // chunks kk .. kk + 2 are full ones; f0 = chunk kk, f1 = chunk kk + 1 on entry
            compute_il(f0, f2, kk + 2);
            compute_il(f1, f0, kk + 3);
            compute_il(f2, f1, min(kk + 4, KCS - 1));  // (past the end: the last chunk once more, never used)
        }
        drain_all();
        if constexpr (kRem == 1) {
            last_frag(f0);
            compute_tail(f0);
        } else if constexpr (kRem == 2) {
            compute(f0);
            drain_all();
            last_frag(f1);
            compute_tail(f1);
        } else {
            compute_il(f0, f2, kEnd3 + 2);
            compute(f1);
            drain_all();
            last_frag(f2);
            compute_tail(f2);
        }
    } else {
        int kk = 0;
        for (; kk + 2 < KC; kk += 2) {  // chunks kk, kk+1 are not the last one
            if constexpr (kIL) {
                compute_il(f0, f1, kk + 1);
                compute_il(f1, f0, kk + 2);
            } else {
                load(f1, kk + 1);
                __builtin_amdgcn_sched_barrier(0);
                compute(f0);
                __builtin_amdgcn_sched_barrier(0);
                load(f0, kk + 2);
                __builtin_amdgcn_sched_barrier(0);
                compute(f1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // (k-split instances drain unconditionally: their register allocation copies the accumulators where the tail's arms begin, and
        // the build's ISA scan cannot know that the path "loop left with results in flight AND kk == 0" does not exist)
        if (KS != 0 || kk > 0) drain_all();
        // KSI: the last chunk's activation fragment is a_last (what the loads above fetched from that LDS position is unwritten space)
        auto last_frag = [&](GemmFrags<R, CT, EX>& f) __attribute__((always_inline)) {
            if constexpr (KSI) {
                if constexpr (CT > 0) f.a[0] = a_last;
#pragma unroll
                for (int e = 0; e < EX; ++e) f.ax[e] = a_last;
                if constexpr (CT > 0) pin(f.a[0]);
#pragma unroll
                for (int e = 0; e < EX; ++e) pin(f.ax[e]);
            }
        };
        if (kk + 1 < KC) {  // two chunks left: kk (full) and kk+1 (tail)
            if constexpr (kIL) {
                compute_il(f0, f1, kk + 1);
            } else {
                load(f1, kk + 1);
                __builtin_amdgcn_sched_barrier(0);
                compute(f0);
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (KS != 0) drain_all();  // (the allocator copies the accumulators where the tail's arms begin: see above)
            last_frag(f1);
            compute_tail(f1);
        } else {  // one chunk left
            last_frag(f0);
            compute_tail(f0);
        }
    }
    // the fused tail runs over the wave's units in groups of at most kTailGroup: within a group the LDS loads of ALL its units are
    // issued first (prep), then the arithmetic (unit); the slots of one group are dead before the next starts (a wave of a
    // 47-tile output layer holds 6-8 units per pass: all their slots at once would not fit the arch VGPRs)
    constexpr int kNUt = CT * R + EX;
    constexpr int kTailGroup = 4;
    FusedSlot slots[kTailGroup];
    auto unit_c = [&](const int u) __attribute__((always_inline)) { return u < CT * R ? c_first + kWaves * (u / R) : exc[(u >= CT * R && u < kNUt) ? u - CT * R : 0]; };
    auto unit_r = [&](const int u) __attribute__((always_inline)) { return u < CT * R ? u % R : exr[(u >= CT * R && u < kNUt) ? u - CT * R : 0]; };
    if constexpr (!std::is_same<TL, NoTail>::value) {  // the first group's LDS loads: in flight while the matrix pipe drains
#pragma unroll
        for (int k = 0; k < kTailGroup; ++k)
            if (k < kNUt) tl->prep(slots[k < kNUt ? k : 0], unit_c(k), unit_r(k));
    }
    if constexpr (kSplit) {
        mfma_drain();
        if constexpr (CT) acc[0][0] += acc_odd;
        else accx[0] += acc_odd;
    }
    if constexpr (kSplitAll) {
        drain_all();
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int r = 0; r < R; ++r) acc[ct][r] += acco[ct][r];
#pragma unroll
        for (int e = 0; e < EX; ++e) accx[e] += accxo[e];
    }
    mfma_drain();
    __builtin_amdgcn_sched_barrier(0);
    prof.mark(11);
    if constexpr (PRE) prefetch_issue(*nxt, lane, *pre);
    if constexpr (!std::is_same<TL, NoTail>::value) {
        // (constant trip counts on both levels: every slot / accumulator index must be a constant after unrolling)
#pragma unroll
        for (int g = 0; g < (kNUt + kTailGroup - 1) / kTailGroup; ++g) {
            if (g > 0) {
#pragma unroll
                for (int k = 0; k < kTailGroup; ++k) {
                    const int u = g * kTailGroup + k;
                    if (u < kNUt) tl->prep(slots[k], unit_c(u), unit_r(u));
                }
            }
            static_assert(kTailGroup % 2 == 0, "the draw stage pairs the units of a group");
            // (pair by pair -- draw, unit, unit -- so that only one pair's normals are live at a time: with the whole group's drawn up front
            // three two-workgroups-per-CU instances spilt to scratch memory)
#pragma unroll
            for (int k = 0; k < kTailGroup; k += 2) {
                const int u = g * kTailGroup + k;
                if (u + 1 < kNUt) tl->draw(slots[k], slots[k + 1], unit_c(u), unit_c(u + 1), true);
                else if (u < kNUt) tl->draw(slots[k], slots[k], unit_c(u), unit_c(u), false);
#pragma unroll
                for (int kk = k; kk < k + 2; ++kk) {
                    const int uu = g * kTailGroup + kk;
                    if (uu < CT * R) tl->unit(slots[kk], acc[(uu < CT * R ? uu : 0) / R][(uu < CT * R ? uu : 0) % R], unit_c(uu), unit_r(uu));
                    else if (uu < kNUt) tl->unit(slots[kk], accx[(uu >= CT * R && uu < kNUt) ? uu - CT * R : 0], unit_c(uu), unit_r(uu));
                }
            }
        }
        tl->finish();
        prof.mark(9);  // the fused tail is booked as the "sample" phase
        return;
    }

    if constexpr (KSO) {
        // this wave's partial sum of the split tile -> LDS, as the f32x4 the lane holds (= the next op's fragment layout).  The two
        // accumulators are re-defined behind the drain above: to the compiler an asm MFMA's result is ready at once, and it would
        // otherwise be free to form this sum right behind the mini-loop, while the matrix pipe still writes the registers
        asm volatile("" : "+v"(ks_e));
        asm volatile("" : "+v"(ks_o));
        *reinterpret_cast<f32x4*>(ks->part_out + (ks->wave * 64 + lane) * 4) = ks_e + ks_o;
    }
    // epilogue: D[row = 4*(lane>>4)+i][col = lane&15] -> bias, activation, next layer's A image.
    // The activation switch is hoisted OUT of the element loops: one compact straight-line body per
    // activation (a per-element switch made the hot path stream ~12 KB of mostly-skipped code per layer
    // through the instruction cache: 11k cycles per epilogue instead of ~2k).
    // The product is formed transposed (weights are the MFMA A operand, activations the B operand), so a lane's
    // accumulator holds 4 CONSECUTIVE LDS columns (16c + 4g .. +3; the weight / bias packing pre-permutes the real
    // columns so that this holds in the chunk-transposed layout too) of batch row 16r + (lane & 15): one
    // ds_write_b128 per accumulator instead of four ds_write_b32.
    // actfn maps the 4 accumulator values of a lane at once (lets an activation use packed 2 x f32 VALU instructions:
    // a wave's VALU work is not hidden behind anything here, so the instruction count is the cost)
    auto store = [&](auto actfn) __attribute__((always_inline)) {
        const int j = lane & 15, g4 = 4 * (lane >> 4);
        HIPETS_BOUND(c_first >= 0 && (CT == 0 || (c_first + kWaves * (CT - 1)) * 16 + g4 + 3 < ld) && KC >= 1 && KC * 16 <= ldi);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const int col = (c_first + kWaves * ct) * 16 + g4;
#pragma unroll
            for (int r = 0; r < R; ++r) *reinterpret_cast<f32x4*>(out + (r * 16 + j) * ld + col) = actfn(acc[ct][r]);
        }
#pragma unroll
        for (int e = 0; e < EX; ++e) *reinterpret_cast<f32x4*>(out + (exr[e] * 16 + j) * ld + exc[e] * 16 + g4) = actfn(accx[e]);
    };
    auto each = [](auto f) {  // lift a scalar activation to the 4 values
        return [f](const f32x4 a) {
            f32x4 v;
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = f(a[i]);
            return v;
        };
    };
    if (!apply_act) {
        store([](const f32x4 a) { return a; });
    } else {
        switch (ACT >= 0 ? ACT : act) {
            case HIPETS_ACT_SILU: store([](const f32x4 a) { return silu4(a); }); break;
            case HIPETS_ACT_RELU: store(each([](float x) { return x < 0.0f ? 0.0f : x; })); break;  // NOT fmaxf: v_max_f32 returns 0 for a NaN input, torch.relu returns NaN
            case HIPETS_ACT_LEAKY_RELU: store(each([slope](float x) { return x > 0.0f ? x : slope * x; })); break;
            case HIPETS_ACT_TANH: store(each([](float x) { return tanhf(x); })); break;
            default: store(each([](float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.44269504088896340736f)); })); break;
        }
    }
    prof.mark(13);
}

template <int R, int CT, int ACT, bool SPL = false>
__device__ __forceinline__ void wave_gemm_ex(int nex, const float* in, float* out, int ld, const float* W,
                                             const float* bias, int KC, int tail_steps, int c_first, const Extras ex,
                                             bool apply_act, int act, float slope, int lane, Prof& prof) {
    // a wave holds at most ceil(3 R / 4) leftover units (C % 4 <= 3 leftover column tiles x R row tiles dealt to 4 waves): only those
    // counts are instantiated (every instance adds to the kernel's register maximum, and the generic kernels sit at the limit)
    constexpr int kMaxEx = (3 * R + kWaves - 1) / kWaves;
    switch (nex) {
        case 0:
            if constexpr (CT > 0) wave_gemm<R, CT, 0, ACT, false, NoTail, -1, SPL>(in, out, ld, W, bias, KC, tail_steps, c_first, ex, apply_act, act, slope, lane, prof);
            break;
        case 1: wave_gemm<R, CT, 1, ACT, false, NoTail, -1, SPL>(in, out, ld, W, bias, KC, tail_steps, c_first, ex, apply_act, act, slope, lane, prof); break;
        case 2:
            if constexpr (kMaxEx >= 2) wave_gemm<R, CT, 2, ACT, false, NoTail, -1, SPL>(in, out, ld, W, bias, KC, tail_steps, c_first, ex, apply_act, act, slope, lane, prof);
            break;
        default:
            if constexpr (kMaxEx >= 3) wave_gemm<R, CT, 3, ACT, false, NoTail, -1, SPL>(in, out, ld, W, bias, KC, tail_steps, c_first, ex, apply_act, act, slope, lane, prof);
            break;
    }
}

// One linear op (+activation) for the workgroup's 16*R rows: in (LDS) -> out (LDS), both with row stride ld.
// W / bias point at the packed fragments / padded biases of this op (pack_weights_kernel / pack_bias_kernel).
// CS >= 0: the number of column tiles is a compile-time fact (shape-specialised kernels): every wave's (CT, EX) follows
// from it and the wave index through ONE branch, and only the two wave_gemm instances the shape needs are compiled;
// CS < 0: it is read from the layer table and dispatched through the (full, nex) switches.
// KS (shape-specialised ops of one-tile workgroups): wave_gemm's k-split bits; part_in / part_out: the partial-sum buffers (KsArgs)
template <int R, int ACT = -1, int CS = -1, bool PRE = false, class TL = NoTail, int LD = -1, bool SPL = false, int KCS = -1, int KS = 0>
__device__ __forceinline__ void linear_op(const float* W, const float* bias, const LayerMeta lm, const int ld, const bool apply_act,
                                          const int activation, const float slope, const float* in, float* out, const int wave,
                                          const int lane, Prof& prof, Pre* pre = nullptr, const NextOp* nxt = nullptr, const TL* tl = nullptr,
                                          const int ldi = 0, const float* part_in = nullptr, float* part_out = nullptr) {
    static_assert(std::is_same<TL, NoTail>::value || CS >= 0, "a fused tail needs a shape-specialised op");
    static_assert(!KS || CS >= 0, "k-split needs a shape-specialised op");
    const int KC = lm.Kp / kKChunk;
    if constexpr ((KS & 2) != 0) {
        // k-split producer: the CS - 1 strided tiles as usual (CS / 4 per wave, no leftover units), the last tile's k range dealt to the waves
        static_assert(CS % kWaves == 1 && CS / kWaves >= 1 && CS / kWaves <= 3, "k-split: one leftover column tile");
        KsArgs ks;
        ks.part_in = part_in; ks.part_out = part_out; ks.tile = CS - 1; ks.wave = wave;
        ks.k0 = (wave * KC) / kWaves;
        ks.n = ((wave + 1) * KC) / kWaves - ks.k0;
        Extras ex0;
        ex0.c0 = ex0.c1 = ex0.c2 = ex0.c3 = 0; ex0.r0 = ex0.r1 = ex0.r2 = ex0.r3 = 0;
        wave_gemm<R, CS / kWaves, 0, ACT, false, NoTail, LD, false, KCS, KS>(in, out, ld, W, bias, KC, lm.tail_steps, wave, ex0, apply_act, activation, slope, lane, prof,
                                                                            nullptr, nullptr, nullptr, ldi, &ks);
    } else {
    KsArgs ks;  // (KS == 1: a consumer only -- the output layer)
    ks.part_in = part_in; ks.part_out = nullptr; ks.tile = 0; ks.k0 = 0; ks.n = 0; ks.wave = wave;
    // a wave's strided column tiles go through in passes of at most kMaxCT tiles (accumulator + double-buffered
    // fragment registers must fit the 256 VGPRs two waves per SIMD leave each wave)
    constexpr int kMaxCT = kWaves >= 8 ? 2 : 3;
    if constexpr (CS >= 0) {
        constexpr int full = CS / kWaves, rem = CS % kWaves, nu = rem * R;
        Extras ex;
        ex.c0 = kWaves * full + wave / R;                ex.r0 = wave % R;
        ex.c1 = kWaves * full + (wave + kWaves) / R;     ex.r1 = (wave + kWaves) % R;
        ex.c2 = kWaves * full + (wave + 2 * kWaves) / R; ex.r2 = (wave + 2 * kWaves) % R;
        ex.c3 = kWaves * full + (wave + 3 * kWaves) / R; ex.r3 = (wave + 3 * kWaves) % R;
        constexpr int passes = full > kMaxCT ? (full - 1) / kMaxCT : 0;  // whole passes of kMaxCT tiles before the last one
        constexpr int last = full - passes * kMaxCT;                      // 0 .. kMaxCT column tiles ride with the extras
        if constexpr (std::is_same<TL, NoTail>::value) {
#pragma unroll
            for (int p = 0; p < passes; ++p)
                wave_gemm<R, kMaxCT, 0, ACT, false, TL, LD, SPL, KCS, (KS & 4)>(in, out, ld, W, bias, KC, lm.tail_steps, wave + kWaves * kMaxCT * p, ex, apply_act, activation, slope, lane, prof, nullptr, nullptr, tl, ldi);
        } else {  // with a fused tail inlined per unit the body is large: ONE copy, a real loop over the passes
#pragma nounroll
            for (int p = 0; p < passes; ++p)
                wave_gemm<R, kMaxCT, 0, ACT, false, TL, LD, SPL, KCS, (KS & 4)>(in, out, ld, W, bias, KC, lm.tail_steps, wave + kWaves * kMaxCT * p, ex, apply_act, activation, slope, lane, prof, nullptr, nullptr, tl, ldi);
        }
        const int c_first = wave + kWaves * kMaxCT * passes;
        // the nu leftover units are dealt round-robin: waves below nu % kWaves hold one more than the others
        constexpr int lo = nu / kWaves, hi = (nu + kWaves - 1) / kWaves;
        static_assert(!PRE || passes == 0, "cross-layer prefetch needs the op to fit one wave_gemm per wave");
        if constexpr (lo == hi) {
            if constexpr (last > 0 || lo > 0)
                wave_gemm<R, last, lo, ACT, PRE, TL, LD, SPL, KCS, KS>(in, out, ld, W, bias, KC, lm.tail_steps, c_first, ex, apply_act, activation, slope, lane, prof, pre, nxt, tl, ldi, &ks);
            else if constexpr (PRE) prefetch_issue(*nxt, lane, *pre);  // nothing to compute here: still fetch for the next op
        } else {
            if (wave < nu % kWaves) {
                wave_gemm<R, last, hi, ACT, PRE, TL, LD, SPL, KCS, KS>(in, out, ld, W, bias, KC, lm.tail_steps, c_first, ex, apply_act, activation, slope, lane, prof, pre, nxt, tl, ldi, &ks);
            } else {
                if constexpr (last > 0 || lo > 0)
                    wave_gemm<R, last, lo, ACT, PRE, TL, LD, SPL, KCS, KS>(in, out, ld, W, bias, KC, lm.tail_steps, c_first, ex, apply_act, activation, slope, lane, prof, pre, nxt, tl, ldi, &ks);
                else if constexpr (PRE) prefetch_issue(*nxt, lane, *pre);
            }
        }
    } else {
        const int C = lm.Np / kTile;
        const int full = C / kWaves, rem = C % kWaves;
        // leftover units u = (column tile kWaves*full + u / R, row tile u % R), dealt round-robin to waves
        const int nu = rem * R;
        Extras ex;
        ex.c0 = kWaves * full + wave / R;                ex.r0 = wave % R;
        ex.c1 = kWaves * full + (wave + kWaves) / R;     ex.r1 = (wave + kWaves) % R;
        ex.c2 = kWaves * full + (wave + 2 * kWaves) / R; ex.r2 = (wave + 2 * kWaves) % R;
        ex.c3 = kWaves * full + (wave + 3 * kWaves) / R; ex.r3 = (wave + 3 * kWaves) % R;
        const int nex = wave < nu ? (nu - wave + kWaves - 1) / kWaves : 0;  // <= kMaxExtras since rem < kWaves, R <= 4
        int done = 0;
        while (full - done > kMaxCT) {
            wave_gemm<R, kMaxCT, 0, ACT, false, NoTail, -1, SPL>(in, out, ld, W, bias, KC, lm.tail_steps, wave + kWaves * done, ex, apply_act, activation, slope, lane, prof);
            done += kMaxCT;
        }
        const int c_first = wave + kWaves * done;
        switch (full - done) {
            case 0: wave_gemm_ex<R, 0, ACT, SPL>(nex, in, out, ld, W, bias, KC, lm.tail_steps, c_first, ex, apply_act, activation, slope, lane, prof); break;
            case 1: wave_gemm_ex<R, 1, ACT, SPL>(nex, in, out, ld, W, bias, KC, lm.tail_steps, c_first, ex, apply_act, activation, slope, lane, prof); break;
            case 2: wave_gemm_ex<R, 2, ACT, SPL>(nex, in, out, ld, W, bias, KC, lm.tail_steps, c_first, ex, apply_act, activation, slope, lane, prof); break;
            default:
                // (SPL ops have at most 8 column tiles, i.e. at most 2 strided tiles per wave: mlp_layer)
                if constexpr (kMaxCT >= 3 && !SPL)
                    wave_gemm_ex<R, 3, ACT, SPL>(nex, in, out, ld, W, bias, KC, lm.tail_steps, c_first, ex, apply_act, activation, slope, lane, prof);
                break;
        }
    }
    }  // (not a k-split producer)
}

// ---------------------------------------------------------------------------------------------------------------------
// bf16x3 precision mode ("f32 on the bf16 matrix pipe").  An fp32 operand x is carried as three bf16 pieces x0 + x1 + x2
// (x0 = bf16(x), x1 = bf16(x - x0), x2 = bf16(x - x0 - x1): |x - x0 - x1 - x2| <= 2^-24 |x|), and a product a b is formed from
// the six partial products of weight <= 2^-16: a0 b0 + (a0 b1 + a1 b0) + (a0 b2 + a1 b1 + a2 b0), each EXACT in fp32
// (8 x 8 significand bits), accumulated in fp32 by v_mfma_f32_16x16x32_bf16.  The dropped terms (a1 b2, a2 b1, a2 b2) are
// <= 2^-23 |a b|: the result is fp32-accurate to a few ulps of the products, at 6 x 17 cycles per 16x16x32 block instead of
// 8 x 32 cycles for the fp32 MFMAs -- and the bf16 matrix pipe, unlike the fp32 one, runs beside the VALU.
// Layouts: weights packed per (column tile, 32-wide k chunk, piece) as one A-operand fragment (lane l: output column
// l & 15, k = 8 (l >> 4) .. + 7); activations in LDS per row as [k chunk][piece][4 groups][8 x bf16] so that a lane's
// B-operand fragment of a piece is ONE ds_read_b128.  The last layer's results stay fp32 (sampling reads them).
// ---------------------------------------------------------------------------------------------------------------------
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

// round-to-nearest-even bf16 of x as the upper 16 bits of a word (finite x)
__host__ __device__ __forceinline__ unsigned bf16_rne_bits(float x) {
    unsigned u;
    __builtin_memcpy(&u, &x, 4);
    return (u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u;
}
__host__ __device__ __forceinline__ float bits_to_float(unsigned u) {
    float f;
    __builtin_memcpy(&f, &u, 4);
    return f;
}
// the three pieces of x as bf16 bit patterns (in the UPPER halves of h[0..2])
__host__ __device__ __forceinline__ void split3(float x, unsigned (&h)[3]) {
    h[0] = bf16_rne_bits(x);
    const float r1 = x - bits_to_float(h[0]);
    h[1] = bf16_rne_bits(r1);
    const float r2 = r1 - bits_to_float(h[1]);
    h[2] = bf16_rne_bits(r2);
}
// four consecutive values -> per piece one 8-byte word pair (4 x bf16, little endian: value 0 in the low half of word 0).
// v_cvt_pk_bf16_f32 rounds two floats to nearest-even and packs them in exactly that order: one conversion, two bit
// operations and one packed subtract per pair and piece (the host-side split3 above states the same arithmetic bit by bit).
__device__ __forceinline__ void split3x4(const f32x4 v, u32x2 (&out)[3]) {
    using f32p = __attribute__((ext_vector_type(2))) float;
    using bf16p = __attribute__((ext_vector_type(2))) __bf16;
    f32p lo = {v[0], v[1]}, hi = {v[2], v[3]};
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        const bf16p hl = __builtin_convertvector(lo, bf16p), hh = __builtin_convertvector(hi, bf16p);
        unsigned ul, uh;
        __builtin_memcpy(&ul, &hl, 4);
        __builtin_memcpy(&uh, &hh, 4);
        out[p][0] = ul;
        out[p][1] = uh;
        if (p < 2) {
            lo = lo - f32p{bits_to_float(ul << 16), bits_to_float(ul & 0xFFFF0000u)};
            hi = hi - f32p{bits_to_float(uh << 16), bits_to_float(uh & 0xFFFF0000u)};
        }
    }
}
__device__ __forceinline__ bf16x8 as_bf16x8(const u32x4 v) {
    bf16x8 r;
    __builtin_memcpy(&r, &v, 16);
    return r;
}
// byte offset inside an activation row of the 4 consecutive columns k0 .. k0 + 3 (k0 % 4 == 0) of piece p
__device__ __forceinline__ int b3_offset(int k0, int p) { return (k0 >> 5) * 192 + p * 64 + ((k0 & 31) >> 3) * 16 + (k0 & 7) * 2; }

template <int R, int CT, int EX>
struct GemmFragsB3 {
    u32x4 w[CT > 0 ? CT : 1][3];   // weight pieces (A operand) of the strided column tiles
    u32x4 wx[EX > 0 ? EX : 1][3];  // ... of the extra units
    u32x4 a[R][3];                 // activation pieces (B operand) of the row tiles
    u32x4 ax[EX > 0 ? EX : 1][3];
};

// One wave's share of a linear op in bf16x3 arithmetic: same unit decomposition as wave_gemm (CT strided column tiles x R row
// tiles + EX extra units).  `in`: LDS activation pieces (byte stride ldb); hidden ops write the activated result as pieces into
// `out`, the last op writes fp32 (float stride ldb / 4) for the sampling phase.
template <int R, int CT, int EX, int ACT>
__device__ __forceinline__ void wave_gemm_b3(const char* __restrict__ in, char* __restrict__ out, const int ldb, const uint4* __restrict__ W3,
                                             const float* __restrict__ bias, const int KC32, const int c_first, const Extras ex,
                                             const bool last_op, const int lane) {
    constexpr int CTn = CT > 0 ? CT : 1;
    constexpr int EXn = EX > 0 ? EX : 1;
    f32x4 acc[CTn][R];
    f32x4 accx[EXn];
    const int exc[kMaxExtras] = {ex.c0, ex.c1, ex.c2, ex.c3};
    const int exr[kMaxExtras] = {ex.r0, ex.r1, ex.r2, ex.r3};
    // weights: 16-byte units; (column tile c, chunk kk, piece p, lane) -> ((c * KC32 + kk) * 3 + p) * 64 + lane
    unsigned woff[CTn], wxoff[EXn];
    int axoff[EXn];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) woff[ct] = (unsigned)((c_first + kWaves * ct) * KC32 * 192 + lane);
#pragma unroll
    for (int e = 0; e < EX; ++e) {
        wxoff[e] = (unsigned)(exc[e] * KC32 * 192 + lane);
        axoff[e] = exr[e] * 16 * ldb;
    }
    const char* ap = in + (lane & 15) * ldb + (lane >> 4) * 16;
    // biases: the packed bias arrays serve the fp32 kernels, where hidden layers keep their columns permuted inside every group
    // of 16 (position lds_col(n) holds column n; lds_col is an involution); here columns are natural
    auto bias4 = [&](const int c) __attribute__((always_inline)) {
        const int g = lane >> 4;
        if (last_op) return *reinterpret_cast<const f32x4*>(bias + c * 16 + 4 * g);
        const float* bp = bias + c * 16 + g;  // natural column 4 g + i sits at position 4 i + g
        return f32x4{bp[0], bp[4], bp[8], bp[12]};
    };
#pragma unroll
    for (int ct = 0; ct < CTn; ++ct) {
        const f32x4 b = CT > 0 ? bias4(c_first + kWaves * ct) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < R; ++r) acc[ct][r] = b;
    }
#pragma unroll
    for (int e = 0; e < EXn; ++e) accx[e] = EX > 0 ? bias4(exc[e]) : f32x4{0.f, 0.f, 0.f, 0.f};

    auto load = [&](GemmFragsB3<R, CT, EX>& f, const int kk) __attribute__((always_inline)) {
        const uint4* Wk = W3 + (size_t)kk * 192;
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int p = 0; p < 3; ++p) f.w[ct][p] = *reinterpret_cast<const u32x4*>(Wk + woff[ct] + p * 64);
#pragma unroll
        for (int e = 0; e < EX; ++e)
#pragma unroll
            for (int p = 0; p < 3; ++p) f.wx[e][p] = *reinterpret_cast<const u32x4*>(Wk + wxoff[e] + p * 64);
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int p = 0; p < 3; ++p) f.a[r][p] = *reinterpret_cast<const u32x4*>(ap + r * 16 * ldb + kk * 192 + p * 64);
#pragma unroll
        for (int e = 0; e < EX; ++e)
#pragma unroll
            for (int p = 0; p < 3; ++p) f.ax[e][p] = *reinterpret_cast<const u32x4*>(ap + axoff[e] + kk * 192 + p * 64);
    };
    // the six partial products of one unit, smallest weights first
    auto unit = [&](const u32x4 (&w)[3], const u32x4 (&a)[3], f32x4& c) __attribute__((always_inline)) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(w[2]), as_bf16x8(a[0]), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(w[1]), as_bf16x8(a[1]), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(w[0]), as_bf16x8(a[2]), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(w[1]), as_bf16x8(a[0]), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(w[0]), as_bf16x8(a[1]), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(w[0]), as_bf16x8(a[0]), c, 0, 0, 0);
    };
    auto compute = [&](const GemmFragsB3<R, CT, EX>& f) __attribute__((always_inline)) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int r = 0; r < R; ++r) unit(f.w[ct], f.a[r], acc[ct][r]);
#pragma unroll
        for (int e = 0; e < EX; ++e) unit(f.wx[e], f.ax[e], accx[e]);
    };
    GemmFragsB3<R, CT, EX> f0, f1;
    load(f0, 0);
    int kk = 0;
    for (; kk + 1 < KC32; kk += 2) {
        load(f1, kk + 1);
        __builtin_amdgcn_sched_barrier(0);
        compute(f0);
        __builtin_amdgcn_sched_barrier(0);
        if (kk + 2 < KC32) load(f0, kk + 2);
        __builtin_amdgcn_sched_barrier(0);
        compute(f1);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (kk < KC32) compute(f0);

    const int j = lane & 15, g4 = 4 * (lane >> 4);
    auto silu4 = [](const f32x4 a) {
        using f32x2 = __attribute__((ext_vector_type(2))) float;
        const f32x2 k = {-1.44269504088896340736f, -1.44269504088896340736f}, one = {1.0f, 1.0f};
        const f32x2 lo = {a[0], a[1]}, hi = {a[2], a[3]};
        f32x2 tl = lo * k, th = hi * k;
        tl[0] = __builtin_amdgcn_exp2f(tl[0]); tl[1] = __builtin_amdgcn_exp2f(tl[1]);
        th[0] = __builtin_amdgcn_exp2f(th[0]); th[1] = __builtin_amdgcn_exp2f(th[1]);
        tl = tl + one; th = th + one;
        tl[0] = __builtin_amdgcn_rcpf(tl[0]); tl[1] = __builtin_amdgcn_rcpf(tl[1]);
        th[0] = __builtin_amdgcn_rcpf(th[0]); th[1] = __builtin_amdgcn_rcpf(th[1]);
        const f32x2 yl = lo * tl, yh = hi * th;
        return f32x4{yl[0], yl[1], yh[0], yh[1]};
    };
    static_assert(ACT == HIPETS_ACT_SILU, "bf16x3 instances exist for SiLU models");
    auto store = [&](const f32x4 v, const int row, const int col0) __attribute__((always_inline)) {
        if (last_op) {
            *reinterpret_cast<f32x4*>(out + (size_t)row * ldb + col0 * 4) = v;  // fp32, natural columns (float stride ldb / 4)
        } else {
            u32x2 pc[3];
            split3x4(silu4(v), pc);
#pragma unroll
            for (int p = 0; p < 3; ++p) *reinterpret_cast<u32x2*>(out + (size_t)row * ldb + b3_offset(col0, p)) = pc[p];
        }
    };
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int r = 0; r < R; ++r) store(acc[ct][r], r * 16 + j, (c_first + kWaves * ct) * 16 + g4);
#pragma unroll
    for (int e = 0; e < EX; ++e) store(accx[e], exr[e] * 16 + j, exc[e] * 16 + g4);
}

// linear op with CS column tiles in bf16x3 arithmetic (static shapes only: the lean instances)
template <int R, int ACT, int CS>
__device__ __forceinline__ void linear_op_b3(const uint4* W3, const float* bias, const int KC32, const int ldb, const bool last_op, const char* in,
                                             char* out, const int wave, const int lane) {
    constexpr int kMaxCT = 3;
    constexpr int full = CS / kWaves, rem = CS % kWaves, nu = rem * R;
    static_assert(full <= kMaxCT, "bf16x3 instances cover ops of at most 15 column tiles");
    Extras ex;
    ex.c0 = kWaves * full + wave / R;                ex.r0 = wave % R;
    ex.c1 = kWaves * full + (wave + kWaves) / R;     ex.r1 = (wave + kWaves) % R;
    ex.c2 = kWaves * full + (wave + 2 * kWaves) / R; ex.r2 = (wave + 2 * kWaves) % R;
    ex.c3 = kWaves * full + (wave + 3 * kWaves) / R; ex.r3 = (wave + 3 * kWaves) % R;
    constexpr int lo = nu / kWaves, hi = (nu + kWaves - 1) / kWaves;
    if constexpr (lo == hi) {
        if constexpr (full > 0 || lo > 0) wave_gemm_b3<R, full, lo, ACT>(in, out, ldb, W3, bias, KC32, wave, ex, last_op, lane);
    } else {
        if (wave < nu % kWaves) {
            wave_gemm_b3<R, full, hi, ACT>(in, out, ldb, W3, bias, KC32, wave, ex, last_op, lane);
        } else {
            if constexpr (full > 0 || lo > 0) wave_gemm_b3<R, full, lo, ACT>(in, out, ldb, W3, bias, KC32, wave, ex, last_op, lane);
        }
    }
}

// Compile-time facts of a rollout-kernel instance; -1 = decided at run time.  Instances with HIDC >= 0 are the
// SHAPE-SPECIALISED ("lean") kernels of the BASELINE configurations: hidden / output column-tile counts, normaliser kind,
// obs preprocessing, reward and termination functions and the launch mode are template arguments, and everything those
// shapes never use (expectation propagation, injected eps, traces, the phase profiler, batched / per-row initial states,
// per-member logvar bounds) is compiled out.  The host picks an instance only when the model and the call match ALL of its
// facts (launch.hpp select_*); anything else runs the generic instance.  Same arithmetic, instruction for instruction, in
// the parts both execute: tests compare the two bit for bit.
// LDS row stride the host derives for a model whose widest layer has `tiles` column tiles (hipets_set_model: >= the widest
// activation, == 8 mod 64); a shape-specialised instance runs only when the model's stride is this one (launch: lean_shape_ok)
// Output layers of up to this many column tiles sum every unit's even / odd k-steps separately (wave_gemm SPL) and, in the
// shape-specialised fp32 instances, feed the fused tail (KSpec::FUSE).  Wider ones (cfg4': 47) keep the plain order: a wave's share
// is a dozen units there, and twice the accumulators (or a dozen inlined tails) do not fit the register file.
constexpr int kSplMaxTiles = 8;

constexpr int lean_ld(int hidc, int outc) {
    int m = (hidc > outc ? hidc : outc) * 16;
    while (m % 64 != 8) m += 4;
    return m;
}

template <int ACT_, int HIDC_ = -1, int OUTC_ = -1, int NORM_ = -1, int OBSP_ = -1, int REW_ = -1, int TERM_ = -1, int KMODE_ = -1, int PREC_ = 0,
          int FUSE_ = 0>
struct KSpec {
    // WIDE (fused fp32 instances whose output layer is wider than kSplMaxTiles column tiles -- cfg4': 47): no LDS image of the outputs
    // exists at all, so the two activation buffers hold hidden activations only (row stride for HIDC tiles) and the model-input
    // image -- wider than a hidden layer there: 393 columns -- lives in buf0 with its own run-time stride (ModelDev::ld_in).
    // 4.4 KB of LDS per row instead of 7.6: two row tiles per workgroup fit where one did.
#ifndef HIPETS_WIDE_FUSE
#define HIPETS_WIDE_FUSE 1
#endif
    static constexpr bool WIDE = HIPETS_WIDE_FUSE && FUSE_ != 0 && HIDC_ >= 0 && PREC_ == HIPETS_PREC_F32 && OUTC_ > kSplMaxTiles;
    static constexpr int LD = (HIDC_ >= 0 && PREC_ == HIPETS_PREC_F32) ? lean_ld(HIDC_, WIDE ? HIDC_ : OUTC_) : -1;  // compile-time LDS row stride (fp32 lean instances)
    static constexpr int ACT = ACT_, HIDC = HIDC_, OUTC = OUTC_, NORM = NORM_, OBSP = OBSP_, REW = REW_, TERM = TERM_, KMODE = KMODE_;
    static constexpr int PREC = PREC_;  // HIPETS_PREC_F32 (fp32 MFMA) or HIPETS_PREC_BF16X3 (lean instances only)
    static constexpr bool LEAN = HIDC_ >= 0 && OUTC_ >= 0;
    // HIDDEN-STATIC instances (HIDC_ >= 0, everything else decided at run time): what ANY model with that hidden width gets --
    // the reference's default is 200 = 13 column tiles (conf/dynamics_model/gaussian_mlp_ensemble.yaml:8), whatever its
    // environment's obs preprocessing, reward / termination functions, normaliser, output width or propagation method.  The ops
    // that carry > 90 % of a step's FLOPs (every op whose N is the hidden width) run exactly like in the shape-specialised
    // instances: per-wave (CT, EX) through one branch, compile-time LDS stride, interleaved fragment loads, unrolled k loops where
    // the register file allows; the output layer and every elementwise phase stay the generic kernel's.
    static constexpr bool HID_STATIC = HIDC_ >= 0 && OUTC_ < 0;
    // FUSE (lean fp32 instances): the output layer runs on the "head pair" pack and its accumulators go straight into the
    // step's tail -- sampling, delta, next state, hand-over publication, reward / termination / totals and the next step's
    // normalised model input happen in registers in the output layer's own barrier interval (5 barriers per step instead
    // of 7, no LDS round trip of the 2 x out_dim outputs).  Needs reward / termination forms that read state dims 0..3 only.
    // (output layers of up to 8 column tiles: beyond that -- cfg4' has 47 -- a wave's tail covers a dozen units and the instance spills)
    static constexpr bool FUSE = FUSE_ != 0 && LEAN && PREC_ == HIPETS_PREC_F32 && (OUTC_ <= kSplMaxTiles || WIDE);
    static constexpr bool SPL_OUT = OUTC_ >= 0 && OUTC_ <= kSplMaxTiles;  // the output layer sums even / odd k-steps separately (wave_gemm SPL)
    // K-split of the leftover hidden column tile (KsArgs above): the fused fp32 instances whose hidden layers leave ONE column tile over
    // (13 = 3 x 4 + 1), used by the kernel for ONE-TILE workgroups only (R = 1: rollout_kernel's kKS)
#ifndef HIPETS_KSPLIT
#define HIPETS_KSPLIT 1
#endif
    // (13 column tiles only: the consumer side, wave_gemm KSI inside KSO, rebuilds the last k chunk from slot kKsSlots - 1 of the last wave,
    // which is where a 13-chunk range -- 3 + 3 + 3 + 4 chunks -- ends; a 5- or 9-tile shape would end in another slot and read unwritten LDS)
    static constexpr bool KSPLIT = HIPETS_KSPLIT && FUSE && !WIDE && kWaves == 4 && HIDC_ == 13;
    // termination functions that test EVERY state dim (inverted_pendulum: isfinite(next_obs).all(), termination_fns.py:47-55) are fused for
    // models with obs_dim <= 4 only -- then dims 0..3 ARE every dim (launch.hpp fused_term_ok checks the model)
    static_assert(!FUSE || ((REW_ == HIPETS_REW_HALFCHEETAH || REW_ == HIPETS_REW_CARTPOLE || REW_ == HIPETS_REW_CARTPOLE_PETS || REW_ == HIPETS_REW_LEARNED) &&
                            (TERM_ == HIPETS_TERM_NONE || TERM_ == HIPETS_TERM_CARTPOLE || TERM_ == HIPETS_TERM_HUMANOID || TERM_ == HIPETS_TERM_INVERTED_PENDULUM ||
                             TERM_ == HIPETS_TERM_HOPPER)),
                  "fused tail: the reward / termination lane sees dims 0..3 of its row");
    // hopper (termination_fns.py:12-26) tests EVERY state dim of a model whose dims span several column tiles, i.e. several waves: every
    // tail lane judges its own two dims and raises a per-row flag in LDS; the flag of step t is complete at the barrier that ends the
    // step and is folded into the row's `terminated` by the tail of step t + 1 -- which is when it first matters (model_env.py:186-188:
    // the reward of the terminating step itself still counts).  The row must still be HERE then: FAST instances only (in the persistent
    // DEVICE form it has moved to another workgroup, which would need the flag through the hand-over table); learned rewards only.
    // Round 5: DEVICE-mode instances too.  One launch per step: the flag of the launch's step is folded into `terminated` behind the
    // step loop, before the write-back.  Persistent form: the row has moved on -- and its NEXT owner holds every dim of the state it
    // receives: the threads that collect a pair of dims judge them exactly like the tail lanes would have and raise the flag in the
    // new owner's LDS (hop_flags below); nothing more travels through the hand-over table.
    static_assert(!FUSE || TERM_ != HIPETS_TERM_HOPPER || (REW_ == HIPETS_REW_LEARNED && !WIDE),
                  "fused tail with an all-dims termination function: instances with a learned reward");
    // learned rewards (round 4): the reward is the sampled LAST output column.  Without a termination function (pets_pusher / pets_reacher /
    // pets_mppi_halfcheetah) the lane that holds that column keeps the row's running total and needs no state dim at all; with one
    // (pets_inv_pendulum) the lane with dims 0, 1 keeps it and fetches the reward from the column's lane of the SAME accumulator, i.e. the
    // column must sit in column tile 0: obs_dim < 8 (fused_term_ok)
    static_assert(!FUSE || REW_ != HIPETS_REW_LEARNED || !WIDE, "fused tail with learned rewards: no WIDE instance");
    // obs preprocessing in the fused tail (round 4): the lane that holds the trig dim writes its sin and cos columns (ObsMap)
    static_assert(!FUSE || (NORM_ == HIPETS_NORM_F64 && (OBSP_ == HIPETS_OBS_NONE || !WIDE)), "fused tail: f64 normaliser; WIDE instances: no obs preprocessing");
    static_assert(!FUSE || OBSP_ == HIPETS_OBS_NONE || OBSP_ == HIPETS_OBS_HALFCHEETAH || OBSP_ == HIPETS_OBS_CARTPOLE_PETS, "unknown obs preprocessing");
};

// can an op with CS column tiles take part in the cross-layer prefetch? (one wave_gemm per wave, see linear_op)
template <int CS> struct PreOk { static constexpr bool value = CS >= 0 && CS / kWaves <= (kWaves >= 8 ? 2 : 3); };

// this wave's share of an op with CS column tiles (the (CT, EX, c_first, extras) linear_op<.., CS> derives), for prefetch_issue
template <int R, int CS>
__device__ __forceinline__ NextOp describe_op(const float* W, const float* bias, const int KC, const int tail_steps, const int wave) {
    constexpr int full = CS / kWaves, rem = CS % kWaves, nu = rem * R;
    constexpr int lo = nu / kWaves, hi = (nu + kWaves - 1) / kWaves;
    NextOp n;
    n.W = W; n.bias = bias; n.KC = KC; n.c_first = wave; n.ct = full; n.tail_steps = tail_steps;
    n.ex_n = lo == hi ? lo : (wave < nu % kWaves ? hi : lo);
    n.ex.c0 = kWaves * full + wave / R;                n.ex.r0 = wave % R;
    n.ex.c1 = kWaves * full + (wave + kWaves) / R;     n.ex.r1 = (wave + kWaves) % R;
    n.ex.c2 = kWaves * full + (wave + 2 * kWaves) / R; n.ex.r2 = (wave + 2 * kWaves) % R;
    n.ex.c3 = kWaves * full + (wave + 3 * kWaves) / R; n.ex.r3 = (wave + 3 * kWaves) % R;
    n.valid = true;
    return n;
}

// the op `l` of member `member` as a prefetch target (lean kernels: hidden ops have S::HIDC column tiles, the last S::OUTC)
template <int R, class S>
__device__ __forceinline__ NextOp describe_layer(const ModelDev& md, const LayerMeta* lmeta, const int l, const int member, const int wave) {
    const LayerMeta lm = lmeta[l];
    const float* W = md.w + (size_t)member * md.wmember + lm.woff;
    const float* bias = md.b + (size_t)member * md.bmember + lm.boff;
    if (l < md.n_layers - 1) return describe_op<R, S::HIDC>(W, bias, lm.Kp / kKChunk, lm.tail_steps, wave);
    return describe_op<R, S::OUTC>(W, bias, lm.Kp / kKChunk, lm.tail_steps, wave);
}

// An op with the cross-layer prefetch: runs from the descriptor `cur` that was computed (and prefetched for) one op earlier --
// every op's pointers and shares are derived exactly once --, consumes `pre`, refills it for `nxt`
template <int R, class S>
__device__ __forceinline__ void mlp_layer_pre(const ModelDev& md, const bool last_op, const NextOp& cur, const float* in, float* out,
                                              const int wave, const int lane, Prof& prof, Pre& pre, const NextOp& nxt) {
    LayerMeta lm;
    lm.Kp = cur.KC * kKChunk;
    lm.tail_steps = cur.tail_steps;
    if (!last_op) linear_op<R, S::ACT, S::HIDC, true>(cur.W, cur.bias, lm, md.ld, true, md.activation, md.slope, in, out, wave, lane, prof, &pre, &nxt);
    else linear_op<R, S::ACT, S::OUTC, true>(cur.W, cur.bias, lm, md.ld, false, md.activation, md.slope, in, out, wave, lane, prof, &pre, &nxt);
}

// Layer l in bf16x3 arithmetic
template <int R, class S>
__device__ __forceinline__ void mlp_layer_b3(const ModelDev& md, const LayerMeta* lmeta, const int l, const int member, const float* in, float* out,
                                             const int wave, const int lane) {
    const LayerMeta lm = lmeta[l];
    const uint4* W3 = md.w3 + (size_t)member * md.w3member + lm.woff3;
    const float* bias = md.b + (size_t)member * md.bmember + lm.boff;
    const char* inb = reinterpret_cast<const char*>(in);
    char* outb = reinterpret_cast<char*>(out);
    if (l < md.n_layers - 1) linear_op_b3<R, S::ACT, S::HIDC>(W3, bias, lm.Kp32 / 32, md.ld * 4, false, inb, outb, wave, lane);
    else linear_op_b3<R, S::ACT, S::OUTC>(W3, bias, lm.Kp32 / 32, md.ld * 4, true, inb, outb, wave, lane);
}

// Layer l of the ensemble MLP with member `member`'s weights.
// part: the two k-split partial-sum buffers of a KSpec::KSPLIT one-tile workgroup ([2][kWaves][64][4] floats, alternating by layer)
template <int R, class S>
__device__ __forceinline__ void mlp_layer(const ModelDev& md, const LayerMeta* lmeta, const int l, const int member,
                                          const float* in, float* out, const int wave, const int lane, Prof& prof, float* part = nullptr) {
    const LayerMeta lm = lmeta[l];  // staged in LDS once per launch (a global scalar load here cost ~400 cycles per layer)
    const float* W = md.w + (size_t)member * md.wmember + lm.woff;
    const float* bias = md.b + (size_t)member * md.bmember + lm.boff;
    if constexpr (S::KSPLIT && R == 1) {
        // hidden ops of a one-tile workgroup: every wave 3 column tiles + its quarter of the 13th tile's k range (wave_gemm KSO); ops fed
        // by a hidden layer rebuild their last k chunk from the previous op's partial sums (KSI)
        float* const po = part + (l & 1) * (kWaves * 64 * 4);
        const float* const pi = part + ((l & 1) ^ 1) * (kWaves * 64 * 4);
        if (l == 0) linear_op<R, S::ACT, S::HIDC, false, NoTail, S::LD, false, -1, 2>(W, bias, lm, md.ld, true, md.activation, md.slope, in, out, wave, lane, prof, nullptr, nullptr, nullptr, 0, nullptr, po);
        else linear_op<R, S::ACT, S::HIDC, false, NoTail, S::LD, false, HIPETS_KS_TRIPLE ? S::HIDC : -1, 3>(W, bias, lm, md.ld, true, md.activation, md.slope, in, out, wave, lane, prof, nullptr, nullptr, nullptr, 0, pi, po);
    } else if constexpr (S::LEAN) {
        // ops fed by a hidden layer have K = hid: HIDC chunks, a compile-time count (the input layer's K is the model's input width)
        // Unrolled only where the register file is not the constraint (R >= 3: one workgroup per CU, 512 registers per lane).  At R = 2
        // (two workgroups per CU, 256-register cap) the allocator splits accumulator live ranges inside the unrolled stream and
        // puts v_mov copies straight behind asm MFMAs -- which it believes complete at once (wave_gemm, "drain_all") -- and the
        // interleaved + unrolled build returned wrong sums (caught by the cfg5 parity tests); R = 1 measured 1 % slower unrolled.
        constexpr int kHidChunks = (HIPETS_UNROLL_K && MinWavesOf<R>::value == 1) ? S::HIDC : -1;
        if (l == 0) linear_op<R, S::ACT, S::HIDC, false, NoTail, S::LD>(W, bias, lm, md.ld, true, md.activation, md.slope, in, out, wave, lane, prof, nullptr, nullptr, nullptr, S::WIDE ? md.ld_in : 0);
        else if (l < md.n_layers - 1) linear_op<R, S::ACT, S::HIDC, false, NoTail, S::LD, false, kHidChunks>(W, bias, lm, md.ld, true, md.activation, md.slope, in, out, wave, lane, prof);
        else linear_op<R, S::ACT, S::OUTC, false, NoTail, S::LD, (S::OUTC <= kSplMaxTiles), kHidChunks>(W, bias, lm, md.ld, false, md.activation, md.slope, in, out, wave, lane, prof);
    } else if constexpr (S::HID_STATIC) {
        // (unrolled up to 13 MFMA units per wave -- the widest the shape-specialised instances run: at 16 units, hid 256 with R = 4,
        // the allocator splits accumulator live ranges inside the unrolled stream again and the build's ISA scan finds a v_mov of an
        // accumulator behind an MFMA still in flight; the rolled loop ends every block with drain_all)
        constexpr int kHidChunks = (HIPETS_UNROLL_K && MinWavesOf<R>::value == 1 && ((S::HIDC + kWaves - 1) / kWaves) * R <= 13) ? S::HIDC : -1;
        if (l == 0) linear_op<R, S::ACT, S::HIDC, false, NoTail, S::LD>(W, bias, lm, md.ld, true, md.activation, md.slope, in, out, wave, lane, prof);
        else if (l < md.n_layers - 1) linear_op<R, S::ACT, S::HIDC, false, NoTail, S::LD, false, kHidChunks>(W, bias, lm, md.ld, true, md.activation, md.slope, in, out, wave, lane, prof);
        else if (lm.Np / kTile <= kSplMaxTiles)  // the output layer: the generic instance's dispatch, the SAME summation rule (SPL)
            linear_op<R, S::ACT, -1, false, NoTail, -1, true>(W, bias, lm, md.ld, false, md.activation, md.slope, in, out, wave, lane, prof);
        else linear_op<R, S::ACT>(W, bias, lm, md.ld, false, md.activation, md.slope, in, out, wave, lane, prof);
    } else {
        // the output layer of up to kSplMaxTiles column tiles: SPL (the SAME rule in the shape-specialised branch above)
        if (l == md.n_layers - 1 && lm.Np / kTile <= kSplMaxTiles)
            linear_op<R, S::ACT, -1, false, NoTail, -1, true>(W, bias, lm, md.ld, false, md.activation, md.slope, in, out, wave, lane, prof);
        else linear_op<R, S::ACT>(W, bias, lm, md.ld, l < md.n_layers - 1, md.activation, md.slope, in, out, wave, lane, prof);
    }
}

// The OUTPUT layer of a KSpec::FUSE instance: the "head pair" pack, accumulators handed to `tl` (no LDS image of the outputs)
template <int R, class S, class TL>
__device__ __forceinline__ void mlp_output_layer_fused(const ModelDev& md, const LayerMeta* lmeta, const int member, const float* in,
                                                       const int wave, const int lane, Prof& prof, const TL& tl, const float* part = nullptr) {
    const LayerMeta lm = lmeta[md.n_layers - 1];
    const float* W = md.w + (size_t)member * md.wmember + lm.woff_pairs;
    const float* bias = md.b + (size_t)member * md.bmember + lm.boff_pairs;
    if constexpr (S::KSPLIT && R == 1) {  // the last hidden layer (index n_layers - 2) left its 13th tile as partial sums
        const float* const pi = part + ((md.n_layers - 2) & 1) * (kWaves * 64 * 4);
        linear_op<R, S::ACT, S::OUTC, false, TL, S::LD, S::SPL_OUT, HIPETS_KS_TRIPLE ? S::HIDC : -1, 1>(W, bias, lm, md.ld, false, md.activation, md.slope, in, nullptr, wave, lane, prof, nullptr, nullptr, &tl, 0, pi);
    } else {
        linear_op<R, S::ACT, S::OUTC, false, TL, S::LD, S::SPL_OUT, (HIPETS_UNROLL_K && MinWavesOf<R>::value == 1) ? S::HIDC : -1>(W, bias, lm, md.ld, false, md.activation, md.slope, in, nullptr, wave, lane, prof, nullptr, nullptr, &tl);
    }
}

// obs_process_fn seen from the PRODUCER of an observation dim (the fused tail, the straight form's collect phase: both hold a pair
// of raw dims in registers and write the next step's input image themselves): which input column does dim d feed, and which dim
// enters as sin / cos?  halfcheetah (env/pets_halfcheetah.py:91-113): [s1, sin s2, cos s2, s3:] -- dim 0 feeds nothing;
// cartpole_pets (env/pets_cartpole.py:78-101): [sin s1, cos s1, s0, s2:] -- one column more than dims.
template <int OBSP>
struct ObsMap {
    static constexpr int kTrigDim = OBSP == HIPETS_OBS_HALFCHEETAH ? 2 : (OBSP == HIPETS_OBS_CARTPOLE_PETS ? 1 : -1);  // enters as sin and cos
    static constexpr int kSinCol = OBSP == HIPETS_OBS_HALFCHEETAH ? 1 : 0;
    static constexpr int kCosCol = OBSP == HIPETS_OBS_HALFCHEETAH ? 2 : 1;
    // column of dim d (the sin column for the trig dim), -1 = the dim is not a model input
    __device__ static __forceinline__ int col(const int d) {
        if constexpr (OBSP == HIPETS_OBS_HALFCHEETAH) return d == 0 ? -1 : (d == 1 ? 0 : (d == 2 ? 1 : d));
        else if constexpr (OBSP == HIPETS_OBS_CARTPOLE_PETS) return d == 0 ? 2 : (d == 1 ? 0 : d + 1);
        else return d;
    }
};

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
        case HIPETS_REW_NONE: return 0.0f;  // the caller evaluates its own reward_fn on the returned next_obs
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

// termination_fns.hopper (:12-26) seen from ONE pair of state dims (d, d + 1): all finite, |dims 1..| < 100, height (dim 0) > 0.7,
// |angle (dim 1)| < 0.2.  The row is unhealthy iff any of its pairs says so (fused tail lanes / the collecting threads of the
// persistent DEVICE form, KSpec above).
__device__ __forceinline__ bool hopper_pair_bad(const int d, const float vA, const float vB, const bool hasA, const bool hasB) {
    bool bad = false;
    if (hasA) bad = !isfinite(vA) || (d >= 1 ? !(fabsf(vA) < 100.0f) : !(vA > 0.7f));
    if (hasB) bad = bad || !isfinite(vB) || !(fabsf(vB) < 100.0f) || (d == 0 && !(fabsf(vB) < 0.2f));
    return bad;
}

struct RolloutSmem {
    float* buf0;
    float* buf1;
    float* state;    // [ROWS][obs_dim]
    float* actn;     // [2][ROWS][act_dim]  (double buffered: reward(t) reads while input(t+1) is built)
    float* tot;      // [ROWS]
    float* lrew;     // [ROWS] learned reward of the current step
    int* term;       // [ROWS]
    int* rowid;      // [ROWS] global row id (candidate*P + particle) or -1
    int* pend;       // [2][ROWS] persistent DEVICE form: running total / flag granule of the row not yet collected
    double* nmean;   // [in_dim] normaliser stats (f64 like the reference)
    double* nstd;    // [in_dim] (f64 normaliser: holds 1 / std)
    float* minlv;    // [lv_rows][out_dim]
    float* maxlv;    // [lv_rows][out_dim]
    int* nodelta;    // [obs_dim]
    int* sched;      // [H] member slot of this workgroup per step (FAST)
    LayerMeta* lmeta;  // [HIPETS_MAX_LAYERS]
    long long* prof;   // [kWaves][16] phase-cycle accumulators (profiling aid)
    float* dump;     // [4] sink of the fused tail's masked-off LDS stores (branch-free: an inactive lane stores here)
    float* part;     // one-tile workgroups: [2][kWaves][64][4] k-split partial sums (KsArgs)
    float* expacc;   // [ROWS][out_total] (expectation propagation only)
    char* stage_x;   // KSpec::WIDE: the chunks of the hand-over staging area that do not fit buf0 + buf1 (dma_stage_extra_bytes; last section)
};

__host__ __device__ inline size_t align16(size_t x) { return (x + 15) & ~(size_t)15; }

// ld0 > 0: buf0 has its own row stride (KSpec::WIDE: the model-input image), buf1 uses `ld`
// KSpec::WIDE instances collect a turn's rows by LDS-DMA (rollout_kernel, "dma_collect"): rows x pairs 16-byte hand-over pairs staged
// in 1 KiB chunks of 64.  The two activation buffers are idle then and hold most of them; what does not fit gets a section of its own.
__host__ __device__ inline size_t dma_stage_extra_bytes(int rows, int ld, int ld0, int obs_dim) {
    const size_t nvp = (size_t)(obs_dim + 1) / 2 + 1;          // pairs per row
    const size_t need = (size_t)rows * ((nvp + 63) / 64) * 1024;  // every row's pairs in whole chunks of 64 (dma_collect)
    const size_t have = ((size_t)rows * ld0 * 4 + 15) / 16 * 16 + ((size_t)rows * ld * 4 + 15) / 16 * 16;
    return need > have ? need - have : 0;
}

__host__ __device__ inline size_t rollout_smem_bytes(int rows, int ld, int obs_dim, int act_dim, int in_dim, int out_dim,
                                                     int out_total, int horizon, bool expectation, int lv_rows = 1, int ld0 = 0) {
    size_t n = 0;
    if (ld0 > 0) n += dma_stage_extra_bytes(rows, ld, ld0, obs_dim);  // (ld0 > 0: the KSpec::WIDE layout)
    n += align16((size_t)rows * (ld0 > 0 ? ld0 : ld) * 4) + align16((size_t)rows * ld * 4);
    n += align16((size_t)rows * obs_dim * 4);
    n += align16((size_t)2 * rows * act_dim * 4);
    n += 4 * align16((size_t)rows * 4) + align16((size_t)2 * rows * 4);
    n += 2 * align16((size_t)in_dim * 8);
    n += 2 * align16((size_t)lv_rows * out_dim * 4);
    n += align16((size_t)obs_dim * 4);
    n += align16((size_t)horizon * 4);
    n += align16(sizeof(LayerMeta) * HIPETS_MAX_LAYERS);
    n += align16((size_t)kWaves * 16 * 8);
    n += 16;  // dump slot
    if (rows == kTile) n += (size_t)2 * kWaves * 64 * 16;  // one-tile workgroups: the k-split partial sums (RolloutSmem::part)
    if (expectation) n += align16((size_t)rows * out_total * 4);
    return n;
}

// Raw hardware transcendentals (v_exp_f32 / v_log_f32 are base 2, ~1 ulp, no denormal fix-up sequences).
__device__ __forceinline__ float exp_hw(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }
__device__ __forceinline__ float log_hw(float x) { return __builtin_amdgcn_logf(x) * 0.69314718055994530942f; }
// log(1 + e^x) (abs error ~1e-7; F.softplus' threshold-20 branch kept as a select)
__device__ __forceinline__ float softplus_fast(float x) {
    const float y = log_hw(1.0f + exp_hw(fminf(x, 20.0f)));
    return x > 20.0f ? x : y;
}

template <int R, class S> struct MinWaves { static constexpr int value = S::WIDE ? 1 : MinWavesOf<R>::value; };  // (WIDE: the LDS admits one workgroup per CU anyway)

// S = KSpec<...>: the compile-time facts of this instance (generic: only the activation may be fixed; lean: the whole shape).
// Profiling build only (-DHIPETS_STEP_TRACE, profiles/handover_trace.py): wall-clock stamps (100 MHz, chip-wide) of every
// workgroup at four points of every step of the persistent DEVICE form, behind the phase-cycle table of the caller's buffer.
#ifdef HIPETS_STEP_TRACE
#define HIPETS_STAMP(k, step)                                                                                                     \
    do {  /* one record per (launched workgroup, step, turn): stamp_seq = the (step, turn) sequence index of the step loop */     \
        (void)(step);                                                                                                             \
        if (persist && ra.phase_cycles && tid == 0 && n_serve <= 4)                                                                \
            ra.phase_cycles[128 + ((size_t)blockIdx.x * (ra.H * 4) + stamp_seq) * 4 + (k)] = wall_clock64();  /* stride: <= 4 turns per step */ \
    } while (0)
#else
#define HIPETS_STAMP(k, step) do {} while (0)
#endif

template <int R, class S>
__global__ __launch_bounds__(kThreads, (MinWaves<R, S>::value)) void rollout_kernel(const ModelDev md, const RolloutArgs ra) {
    constexpr int ROWS = kTile * R;
    constexpr bool kLean = S::LEAN;
    constexpr bool kB3 = S::PREC == HIPETS_PREC_BF16X3;  // operands as three bf16 pieces on the bf16 matrix pipe (lean instances)
    static_assert(!kB3 || kLean, "bf16x3 arithmetic exists for the shape-specialised instances");
    // cross-layer weight prefetch (wave_gemm PRE): measured on MI355X and left OFF -- cfg2 FAST 1.054 ms with it (1.073 before
    // the descriptors were derived only once and the layer barriers stopped draining vmcnt) against 1.021 ms without: the
    // ~800-cycle first-fragment latency it hides is outweighed by 56 more live VGPRs and the extra scalar work between layers
#ifndef HIPETS_CROSS_LAYER_PREFETCH
#define HIPETS_CROSS_LAYER_PREFETCH 0
#endif
    constexpr bool kPre = HIPETS_CROSS_LAYER_PREFETCH && kLean && PreOk<S::HIDC>::value && PreOk<S::OUTC>::value;
    constexpr bool kFuse = S::FUSE && !kPre;  // the output layer's accumulators feed the step's tail directly (KSpec::FUSE)
    // the k-split ops of one-tile workgroups (mlp_layer: S::KSPLIT && R == 1) exist in the fused step flow only -- the generic flow
    // passes no partial-sum buffer (a build with HIPETS_CROSS_LAYER_PREFETCH=1 turns kFuse off: it must not keep KSPLIT on)
    static_assert(!(S::KSPLIT && R == 1) || kFuse, "k-split one-tile ops need the fused step flow (sm.part)");
    // facts that are template arguments in a lean instance and model / call fields in the generic one
    const int normalizer = S::NORM >= 0 ? S::NORM : md.normalizer;
    const int obs_process = S::OBSP >= 0 ? S::OBSP : md.obs_process;
    const int reward_fn = S::REW >= 0 ? S::REW : md.reward_fn;
    const int term_fn = S::TERM >= 0 ? S::TERM : md.term_fn;
    const int lv_rows = kLean ? 1 : md.lv_rows;
    const bool deterministic = kLean ? false : md.deterministic != 0;
    float* const trace_next_obs = kLean ? nullptr : ra.trace_next_obs;
    float* const trace_rewards = kLean ? nullptr : ra.trace_rewards;
    const int ld_k = S::LD > 0 ? S::LD : md.ld;  // the LDS row stride: a compile-time fact in the shape-specialised fp32 instances
    constexpr bool kWide = S::WIDE;
    const int ld_in = kWide ? md.ld_in : ld_k;    // row stride of the model-input image (buf0 in the WIDE instances, see KSpec)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    RolloutSmem sm;
    {
        // sections whose size follows from ROWS and the row stride first: in a shape-specialised instance (compile-time stride) their
        // addresses are constants -- immediate offsets in the LDS instructions instead of a live SGPR each (the DEVICE instance
        // of cfg2 spills > 200 scalars); the sections sized by the model's run-time dimensions follow
        char* p = smem;
        sm.buf0 = reinterpret_cast<float*>(p); p += align16((size_t)ROWS * (kWide ? md.ld_in : ld_k) * 4);
        sm.buf1 = reinterpret_cast<float*>(p); p += align16((size_t)ROWS * ld_k * 4);
        sm.tot = reinterpret_cast<float*>(p); p += align16((size_t)ROWS * 4);
        sm.lrew = reinterpret_cast<float*>(p); p += align16((size_t)ROWS * 4);
        sm.term = reinterpret_cast<int*>(p); p += align16((size_t)ROWS * 4);
        sm.rowid = reinterpret_cast<int*>(p); p += align16((size_t)ROWS * 4);
        sm.pend = reinterpret_cast<int*>(p); p += align16((size_t)2 * ROWS * 4);
        sm.lmeta = reinterpret_cast<LayerMeta*>(p); p += align16(sizeof(LayerMeta) * HIPETS_MAX_LAYERS);
        sm.prof = reinterpret_cast<long long*>(p); p += align16((size_t)kWaves * 16 * 8);
        sm.dump = reinterpret_cast<float*>(p); p += 16;
        sm.part = reinterpret_cast<float*>(p); p += (ROWS == kTile) ? (size_t)2 * kWaves * 64 * 16 : 0;
        sm.state = reinterpret_cast<float*>(p); p += align16((size_t)ROWS * md.obs_dim * 4);
        sm.actn = reinterpret_cast<float*>(p); p += align16((size_t)2 * ROWS * md.act_dim * 4);
        sm.nmean = reinterpret_cast<double*>(p); p += align16((size_t)md.in_dim * 8);
        sm.nstd = reinterpret_cast<double*>(p); p += align16((size_t)md.in_dim * 8);
        sm.minlv = reinterpret_cast<float*>(p); p += align16((size_t)md.lv_rows * md.out_dim * 4);
        sm.maxlv = reinterpret_cast<float*>(p); p += align16((size_t)md.lv_rows * md.out_dim * 4);
        sm.nodelta = reinterpret_cast<int*>(p); p += align16((size_t)md.obs_dim * 4);
        sm.sched = reinterpret_cast<int*>(p); p += align16((size_t)ra.H * 4);
        sm.expacc = reinterpret_cast<float*>(p);
        sm.stage_x = p;  // (KSpec::WIDE instances have no expectation accumulator: the extra staging chunks are the last section)
#if HIPETS_DEBUG_BOUNDS
        {   // every section starts inside the launch's dynamic LDS, 16-byte aligned, in layout order; the last one ends inside it
            const char* const secs[] = {(char*)sm.buf0, (char*)sm.buf1, (char*)sm.tot, (char*)sm.lrew, (char*)sm.term, (char*)sm.rowid, (char*)sm.pend,
                                        (char*)sm.lmeta, (char*)sm.prof, (char*)sm.dump, (char*)sm.part, (char*)sm.state, (char*)sm.actn, (char*)sm.nmean, (char*)sm.nstd,
                                        (char*)sm.minlv, (char*)sm.maxlv, (char*)sm.nodelta, (char*)sm.sched, (char*)sm.expacc};
            constexpr int kSecs = (int)(sizeof(secs) / sizeof(secs[0]));
            for (int i = 0; i < kSecs; ++i) {
                HIPETS_BOUND(secs[i] >= smem && secs[i] <= smem + ra.lds_bytes);
                HIPETS_BOUND(((size_t)(secs[i] - smem) & 15) == 0);
                HIPETS_BOUND(i == 0 || secs[i] >= secs[i - 1]);
            }
            const bool expect = !S::LEAN && md.propagation == HIPETS_PROP_EXPECTATION;
            HIPETS_BOUND((char*)sm.expacc + (expect ? align16((size_t)ROWS * md.out_total * 4) : 0) <= smem + ra.lds_bytes);
            HIPETS_BOUND(md.Kp0 <= (kWide ? md.ld_in : ld_k) && md.obs_in + md.act_dim == md.in_dim && md.in_dim <= md.Kp0);
        }
#endif
    }
    const int tid = threadIdx.x;
    if (ra.census) {
        // Co-residency self-test (launch.hpp / rollout_inst.inc launch_one): the persistent DEVICE form is only correct when every
        // launched workgroup is resident at once, and the occupancy the host computes (API answer, register and LDS arithmetic) is
        // an estimate.  Same kernel, same launch configuration, so the same footprint: if all gridDim.x workgroups can meet here,
        // they can wait for each other's rows.  A workgroup that is not admitted never arrives and the others time out.
        if (tid == 0) {
            __hip_atomic_fetch_add(ra.census, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const long long t0 = wall_clock64();
            bool all = true;
            while (__hip_atomic_load(ra.census, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (int)gridDim.x) {
                if (wall_clock64() - t0 > ra.poll_ticks) { all = false; break; }
                __builtin_amdgcn_s_sleep(8);
            }
            if (all) __hip_atomic_fetch_add(ra.census + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool fast = S::KMODE >= 0 ? S::KMODE == HIPETS_MODE_FAST : ra.mode == HIPETS_MODE_FAST;
    const bool expectation = kLean ? false : md.propagation == HIPETS_PROP_EXPECTATION;
    // Modes whose workgroups are bound to a member (EXACT / DEVICE: member = wg / groups): hardware deals block b to XCD b % 8
    // (observed, used for speed only), so the logical workgroup index is taken XCD-major -- XCD x runs a CONTIGUOUS range of
    // logical workgroups, i.e. the workgroups of at most two members, and its 4 MiB L2 streams two members' weights instead of
    // all of them (bf16x3: 0.83 MB per member, all five = 4.2 MB do not fit one L2).  Any bijection is correct.
    int wg = blockIdx.x;
    if (!fast) {
#ifdef HIPETS_XCD_ROT  // profiling builds, grids that are multiples of 8 only: which XCD hosts which logical range (is a slow range slow because of the XCD or the member?)
        const int nwg = gridDim.x, x = ((wg & 7) + HIPETS_XCD_ROT) & 7, slot = wg >> 3;
#else
        const int nwg = gridDim.x, x = wg & 7, slot = wg >> 3;
#endif
        wg = x * (nwg >> 3) + min(x, nwg & 7) + slot;
    }

    // ---- which rollout rows does this workgroup own; per-dimension constants into LDS -------------
    int domain = 0;
    if (fast) {
        // the B = pop * P rows in ONE run, particle-major (run index g = p * pop + c), dealt ROWS at a time: ceil(ceil(B / 16) / R)
        // workgroups, whatever pop is (until round 5 every particle's pop rows were dealt on their own: cfg4''s second iCEM iteration,
        // pop 805 -> 51 tiles per particle, took 26 x 20 = 520 two-tile workgroups -- a third round on 256 CUs for 8 of them)
        for (int s = tid; s < ROWS; s += kThreads) {
            const int g = wg * ROWS + s;  // (<= 8000 workgroups x 64 rows)
            const int p = g / ra.pop, c = g - p * ra.pop;
            sm.rowid[s] = g < ra.B ? c * ra.P + p : -1;
        }
        if (!expectation) {
            // this workgroup's member slot of every step: the caller's schedule, or drawn here (one keyed bijection of the workgroup
            // indices per step, thread t evaluates step t's: common.hpp fast_member)
            if (ra.schedule) {
                for (int t = tid; t < ra.H; t += kThreads) sm.sched[t] = ra.schedule[(size_t)t * gridDim.x + wg];
            } else {
                const bool fixed = md.propagation == HIPETS_PROP_FIXED_MODEL;
                for (int t = tid; t < ra.H; t += kThreads)
                    sm.sched[t] = fast_member((unsigned)wg, gridDim.x, ra.fm_a, ra.fm_b, md.M, md.iid_members, ra.seed, ra.stream_id, fixed ? 0xFFFFFFFFu : (unsigned)t);
            }
        }
    } else {
        domain = wg / ra.groups;
        const int j0 = (wg % ra.groups) * ROWS;
        const long long* perm = ra.perm ? ra.perm + (long long)ra.t_begin * ra.perm_step : nullptr;
        // unbalanced member maps (BasicEnsemble) pad every member's slots with -1 at the tail: nothing to do here
        if (perm && perm[(long long)domain * ra.rows_per_domain + j0] < 0) return;
        // DEVICE mode: the step's permutation is a keyed bijection evaluated on the fly (common.hpp perm_apply)
        const PermKeys keys0 = ra.exchange ? ra.step_keys[ra.t_begin] : ra.perm_keys;
        for (int s = tid; s < ROWS; s += kThreads) {
            const int j = j0 + s;
            int rid = -1;
            if (j < ra.rows_per_domain) {
                const int jj = domain * ra.rows_per_domain + j;
                rid = perm ? (int)perm[jj] : (ra.perm_n ? (int)perm_apply((unsigned)jj, ra.perm_n, ra.perm_a, ra.perm_b, keys0) : jj);
            }
            sm.rowid[s] = rid;
        }
    }
    // the ensemble member this workgroup runs: its row domain's (reference semantics: slot j -> member j / (B / M)) -- or, for
    // RolloutArgs::fast_members launches (one step of B independent rows, hipets_step in FAST mode), the FAST rule's
    int member_dom = domain;
    if (!fast && ra.fast_members && !expectation)
        member_dom = __builtin_amdgcn_readfirstlane(
            ra.schedule ? ra.schedule[(size_t)ra.t_begin * gridDim.x + wg]
                        : fast_member((unsigned)wg, gridDim.x, ra.fm_a, ra.fm_b, md.M, md.iid_members, ra.seed, ra.stream_id,
                                      md.propagation == HIPETS_PROP_FIXED_MODEL ? 0xFFFFFFFFu : (unsigned)ra.t_begin));
    const bool persist = !fast && ra.exchange != nullptr;  // DEVICE mode in ONE launch: rows are handed over through `exchange`
    // ---- Ragged last turn (round 6; KSpec::WIDE two-tile instances in the turn-based persistent form) -------------------------------
    // A batch of n2 two-tile logical workgroups on G launched ones is served in ceil(n2 / G) turns per step, and a step is as long as
    // its turns: the last turn costs a whole two-tile turn however few rows it holds (cfg4' iCEM, 497 candidates: 315 logical
    // workgroups = 256 + 59 -- the second turn is 23 % full and the rollout takes exactly as long as the 805-candidate one).  When
    // the row tiles left for the last turn fit ONE per launched workgroup, that turn is dealt in one-tile logical workgroups instead
    // -- v >= half_from: (member domain, tile) in domain-major order behind the last full turn -- and runs the R = 1 bodies of the
    // MLP ops on row tile 0 of the same LDS layout (the turn costs 0.69 of a two-tile one).  Which workgroup holds a row never
    // enters the arithmetic (the step's permutation decides the member, every (column tile, row tile) unit sums in the same k
    // order whatever R is): same bits as the two-tile dealing, the per-step launches and the generic kernel (tested).
    constexpr bool kRagged = S::WIDE && R == 2;
    int half_from = 0x7FFFFFFF;     // first one-tile logical workgroup
    int n_logical = ra.n_logical;   // logical workgroups of this launch
    int hf_dom = 0, hf_tile = 0, hf_tpd = 1;  // where the one-tile range starts (member domain, tile in it); tiles per domain
    if constexpr (kRagged) {
        if (persist && ra.ragged_last_turn) {
            const int G = (int)gridDim.x, n2 = ra.n_logical;
            const int turns2 = (n2 + G - 1) / G, full = (turns2 - 1) * G;  // two-tile logical workgroups of the full turns
            if (turns2 >= 2) {
                hf_tpd = (ra.rows_per_domain + kTile - 1) / kTile;
                hf_dom = full / ra.groups;
                hf_tile = 2 * (full - hf_dom * ra.groups);
                const int rem = (hf_tpd - hf_tile) + (n2 / ra.groups - 1 - hf_dom) * hf_tpd;  // row tiles behind the full turns
                if (rem <= G) {
                    half_from = full;
                    n_logical = full + rem;
                }
            }
        }
    }
    // logical workgroup v -> its member domain, first row slot in the domain, rows it holds
    auto logical_rows = [&](const int v, int& dom, int& j0, int& live) __attribute__((always_inline)) {
        if (!kRagged || v < half_from) {
            dom = v / ra.groups;
            j0 = (v - dom * ra.groups) * ROWS;
            live = ROWS;
        } else {
            int w = v - half_from + hf_tile;  // tile index counted from the start of domain hf_dom
            const int dd = w / hf_tpd;
            dom = hf_dom + dd;
            j0 = (w - dd * hf_tpd) * kTile;
            live = kTile;
        }
    };
    const bool poll_every = ra.poll_ticks < 1000;  // bounds below 10 us (tests of the time-out path): look at the clock on every spin, not every 64th
    // hand-over table row = NVP pairs of 8-byte granules: the state dims (padded to an even count), then {running total, flag}.
    // exchange item i = (row slot i / NVP, pair i % NVP): items tid + q * kThreads of a thread are the same every step
    constexpr int kG = 2;  // 16-byte pair loads in flight per thread and round (cfg2: 48 rows x 10 pairs = 480 items, one round)
    const int NVP = (md.obs_dim + 1) / 2 + 1;  // pairs per row
    const int NV = 2 * NVP;                    // granules per row
    const unsigned nvp_magic = (unsigned)(0x100000000ull / (unsigned)NVP) + 1u;  // NVP >= 2
    // rows wider than a few pairs (cfg4: 24 pairs per row, cfg4': 189) are collected in rounds of kGT pairs per thread, each round one
    // round trip to the table (>= 1 us even when the rows are long there: the later turns of a step): 4 / 8 in flight instead of 2
    // cut cfg4''s 12 rounds per turn to 3.  (Only the collect phase holds these registers; the straight form's cfg2 needs one round.)
    // (round 4, step trace of the turn-based form, profiles/turn_trace.py + r4_turn_trace.json: cfg4' spends 9.5 us per turn in its
    // three rounds of 8 and 8.3 us between "my rows arrived" and "input built"; MORE pairs in flight measured SLOWER -- 16 per thread
    // for the WIDE instances: 9.84 -> 10.36 ms per cfg4' rollout (the per-item state of the retry loop pushes the kernel's
    // accumulators into AccVGPR spill space: 98 -> 126), 8 instead of 4 for cfg4: 3.37 -> 3.41 ms -- and HALVING the hand-over
    // traffic changed nothing (9.84 vs 9.89 ms): the phase is bound by round-trip latency and its own bookkeeping, not by bandwidth)
#ifndef HIPETS_COLLECT_WIDE
#define HIPETS_COLLECT_WIDE 8
#endif
#ifndef HIPETS_COLLECT_OUT4
#define HIPETS_COLLECT_OUT4 4
#endif
    constexpr int kGT = S::WIDE ? HIPETS_COLLECT_WIDE : ((kLean && S::OUTC >= 4) ? HIPETS_COLLECT_OUT4 : kG);
    // Round 6, KSpec::WIDE (cfg4', Humanoid-v4: 189 pairs per row, 6 048 per two-tile workgroup and turn): the rows of a turn are fetched
    // by LDS-DMA -- no destination registers, so ALL of a wave's ~24 KiB are in flight at once (the register path above manages 8 pairs
    // per thread and needs three round trips of ~3 us) -- into the activation buffers, which are idle between two turns, and validated
    // LDS -> LDS by the wave that issued them (dma_collect below).  Every persistent launch of the instance takes the turn-based flow
    // then, also when each workgroup serves one logical workgroup (the straight form's collect writes the input image while it polls:
    // the image IS the staging area here).
#ifndef HIPETS_DMA_COLLECT
#define HIPETS_DMA_COLLECT 1
#endif
    constexpr bool kDmaCollect = S::WIDE && HIPETS_DMA_COLLECT;
    int xs[kG], xv[kG];
#pragma unroll
    for (int q = 0; q < kG; ++q) {
        const int i = tid + q * kThreads;
        xs[q] = i < ROWS * NVP ? i / NVP : -1;
        xv[q] = i < ROWS * NVP ? i - (i / NVP) * NVP : 0;
    }
    if constexpr (kB3) {
        // bf16x3: the k chunks are 32 wide, the column tiles 16: the last chunk of a 13-tile layer ends in 16 columns no epilogue
        // ever writes.  Their weights are zero, but 0 x (whatever bits LDS holds) may be NaN: clear both activation buffers once.
        f32x4* z = reinterpret_cast<f32x4*>(sm.buf0);
        const int n16 = (int)(2 * align16((size_t)ROWS * ld_k * 4) / 16);
        for (int i = tid; i < n16; i += kThreads) z[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // ---- per-dimension constants -> LDS, and the rows' initial state / totals / flags / first actions: EVERY global load of the
    // prologue is issued before the first dependent LDS store, so the whole prologue costs ONE global round trip (the per-step
    // launches of EXACT / DEVICE mode pay it once per step and workgroup; element i of every table is fetched by thread i,
    // tables longer than the workgroup loop on afterwards)
    const int nlv = deterministic ? 0 : lv_rows * md.out_dim;
    const bool norm = normalizer != HIPETS_NORM_NONE;
    double c_nm = 0.0, c_ns = 1.0;
    float c_lo = 0.f, c_hi = 0.f;
    int c_nd = 0, c_lm = 0;
    constexpr int kMetaWords = (int)(sizeof(LayerMeta) / sizeof(int));  // the layer table travels as plain 32-bit words
    const int n_meta = md.n_layers * kMetaWords;                        // (<= 48: one word per thread, no private copy)
    if (norm && tid < md.in_dim) { c_nm = md.norm_mean[tid]; c_ns = md.norm_std[tid]; }
    if (tid < nlv) { c_lo = md.min_lv[tid]; c_hi = md.max_lv[tid]; }
    if (tid < md.obs_dim) c_nd = md.no_delta[tid];
    if (tid < n_meta) c_lm = reinterpret_cast<const int*>(md.layers)[tid];
    auto commit_constants = [&]() __attribute__((always_inline)) {
        if (norm && tid < md.in_dim) { sm.nmean[tid] = c_nm; sm.nstd[tid] = normalizer == HIPETS_NORM_F64 ? 1.0 / c_ns : c_ns; }  // f64: 1 / std, see build_input
        if (tid < nlv) { sm.minlv[tid] = c_lo; sm.maxlv[tid] = c_hi; }
        if (tid < md.obs_dim) sm.nodelta[tid] = c_nd;
        if (tid < n_meta) reinterpret_cast<int*>(sm.lmeta)[tid] = c_lm;
        if (norm)
            for (int i = tid + kThreads; i < md.in_dim; i += kThreads) { sm.nmean[i] = md.norm_mean[i]; sm.nstd[i] = normalizer == HIPETS_NORM_F64 ? 1.0 / md.norm_std[i] : md.norm_std[i]; }
        for (int i = tid + kThreads; i < nlv; i += kThreads) { sm.minlv[i] = md.min_lv[i]; sm.maxlv[i] = md.max_lv[i]; }
        for (int i = tid + kThreads; i < md.obs_dim; i += kThreads) sm.nodelta[i] = md.no_delta[i];
    };
    lds_barrier();  // sm.rowid (and the B3 clear) visible; the constants' loads stay in flight

    // ---- initial state, totals, flags: loads issued in batches of kStage per thread (one round trip, not one per element) ----
    auto load_initial_state = [&]() __attribute__((always_inline)) {
        constexpr int kStage = 4;
        constexpr int kRowStage = (ROWS + kThreads - 1) / kThreads;
        const int n_st = ROWS * md.obs_dim;
        float tot0[kRowStage];
        int term0[kRowStage];
#pragma unroll
        for (int q = 0; q < kRowStage; ++q) {
            const int s = tid + q * kThreads;
            const int rid = s < ROWS ? sm.rowid[s] : -1;
            tot0[q] = (!fast && !persist && rid >= 0) ? ra.totals[rid] : 0.f;
            term0[q] = (!fast && !persist && rid >= 0) ? (int)ra.term[rid] : 0;
        }
        bool first = true;
        for (int base = tid; base < n_st || first; base += kStage * kThreads) {
            float v[kStage];
#pragma unroll
            for (int q = 0; q < kStage; ++q) {
                const int i = base + q * kThreads;
                v[q] = 0.f;
                if (i < n_st) {
                    const int s = i / md.obs_dim, d = i - s * md.obs_dim;
                    const int rid = sm.rowid[s];
                    if (fast) {
                        if (!kLean && ra.init_states) v[q] = rid >= 0 ? ra.init_states[(size_t)rid * md.obs_dim + d] : 0.f;
                        else if (!kLean && ra.pop_env > 0) v[q] = rid >= 0 ? ra.s0[(size_t)((rid / ra.P) / ra.pop_env) * md.obs_dim + d] : 0.f;
                        else v[q] = ra.s0[d];
                    } else if (rid >= 0) {
                        // persistent form: starts from the tiled s0 itself (batched planning: the s0 of the row's environment)
                        if (!persist) v[q] = ra.state[(size_t)rid * md.obs_dim + d];
                        else if (!kLean && ra.pop_env > 0) v[q] = ra.s0[(size_t)((rid / ra.P) / ra.pop_env) * md.obs_dim + d];
                        else v[q] = ra.s0[d];
                    }
                }
            }
            if (first) {  // everything the prologue needs from global memory has been requested: now the dependent stores
                commit_constants();
#pragma unroll
                for (int q = 0; q < kRowStage; ++q) {
                    const int s = tid + q * kThreads;
                    if (s < ROWS) {
                        sm.tot[s] = tot0[q];
                        sm.term[s] = term0[q];
                        sm.lrew[s] = 0.f;
                        sm.pend[s] = 0;
                        sm.pend[ROWS + s] = 0;
                    }
                }
                first = false;
            }
#pragma unroll
            for (int q = 0; q < kStage; ++q) {
                const int i = base + q * kThreads;
                if (i < n_st) sm.state[i] = v[q];
            }
        }
    };

    const int nblk = (md.out_dim + 3) / 4;
    const int Kp0 = md.Kp0;
    Prof prof;
#ifndef HIPETS_LEAN_PROF
#define HIPETS_LEAN_PROF 0  // profiling builds: the phase profiler also in the shape-specialised instances (profiles/kernel_variants.py)
#endif
    prof.on = (!kLean || HIPETS_LEAN_PROF) && ra.phase_cycles != nullptr && wg == 0 && lane == 0;
    prof.slot = sm.prof + wave * 16;
    if (prof.on) {
#pragma unroll
        for (int i = 0; i < 15; ++i) prof.slot[i] = 0;
        prof.slot[15] = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (15 << 11));  // where this wave landed (HW_REG_HW_ID: simd [5:4], cu [11:8])
    }
    prof.t = prof.on ? clock64() : 0;

    // the step's actions (model_env.py:179-182: row r uses candidate r // P) come from HBM.  Each thread owns up to
    // kPrefetch (row, action-dim) elements; their addresses are fixed for the whole horizon up to the t * A term, and
    // the loads for step t+1 are issued at the top of step t's sampling phase so their latency hides behind it.
    constexpr int kPrefetch = 4;
    const int n_act = ROWS * md.act_dim;
    long long act_base[kPrefetch];  // element offset of (candidate, t = 0, a); -1 = nothing to fetch
    // (row slot, action dim) of this thread's elements never change; the candidate of a row id (rid / P) by multiply-high with
    // ceil(2^32 / P): exact while rid * P < 2^32 (B * P: 2e5 x 20 at cfg2), else the division itself
    int act_s[kPrefetch], act_a[kPrefetch];
#pragma unroll
    for (int q = 0; q < kPrefetch; ++q) {
        const int i = tid + q * kThreads;
        act_s[q] = i < n_act ? i / md.act_dim : -1;
        act_a[q] = i < n_act ? i - (i / md.act_dim) * md.act_dim : 0;
    }
#ifdef HIPETS_DBG_NOMAGIC
    const bool magic_ok = false;
#else
    const bool magic_ok = ra.P > 1 && (unsigned long long)ra.B * (unsigned)ra.P < 0x100000000ull;  // (P = 1: the constant would be 2^32)
#endif
    const unsigned magic_p = (unsigned)(0x100000000ull / (unsigned)max(ra.P, 2)) + 1u;
    const int act_stride = ra.H * md.act_dim;
    // whose rows the action fetch / action columns are for: the slot's current rows -- except in the straight persistent form
    // (below), where the actions of step t + 1 are fetched during step t for the rows the slot will hold THEN
    const int* act_rows = sm.rowid;
    auto compute_act_base = [&]() __attribute__((always_inline)) {  // from act_rows (again whenever the workgroup's rows change)
#pragma unroll
        for (int q = 0; q < kPrefetch; ++q) {
            act_base[q] = -1;
            if (act_s[q] >= 0) {
                const int rid = act_rows[act_s[q]];
                const int cand = magic_ok ? (int)__umulhi((unsigned)rid, magic_p) : rid / ra.P;
                if (rid >= 0) act_base[q] = (long long)cand * act_stride + act_a[q];
            }
        }
    };
    compute_act_base();
    auto fetch_actions_rest = [&](const int t) __attribute__((always_inline)) {  // elements beyond kPrefetch per thread (very wide action spaces)
        float* actn_t = sm.actn + (t & 1) * n_act;
        for (int i = tid + kPrefetch * kThreads; i < n_act; i += kThreads) {
            const int s = i / md.act_dim, a = i % md.act_dim;
            const int rid = act_rows[s];
            actn_t[i] = rid >= 0 ? ra.actions[((size_t)(rid / ra.P) * ra.H + t) * md.act_dim + a] : 0.f;
        }
    };
    auto fetch_actions_issue = [&](const int t, float (&av)[kPrefetch]) {
#pragma unroll
        for (int q = 0; q < kPrefetch; ++q) av[q] = act_base[q] >= 0 ? ra.actions[act_base[q] + (long long)t * md.act_dim] : 0.f;
    };
    auto fetch_actions_commit = [&](const int t, const float (&av)[kPrefetch]) {
        float* actn_t = sm.actn + (t & 1) * n_act;
#pragma unroll
        for (int q = 0; q < kPrefetch; ++q) {
            const int i = tid + q * kThreads;
            if (i < n_act) actn_t[i] = av[q];
        }
        fetch_actions_rest(t);
    };

    // model input of step t: cat(obs_process(obs), act), normalised (one_dim_tr_model.py:103-116), into buf0.
    // One item = (row, 4 consecutive columns): four independent LDS-read -> f64 normalise -> LDS-write chains.
    int kq = Kp0 >> 2;  // column quads per row (the padded input width is a multiple of 16; bf16x3: of 32, set once the layer table is in LDS)
    // The (wave-uniform) normaliser / obs-preprocess switches are resolved ONCE per call into a compile-time variant:
    // with the switches inside, each of the four elements became its own chain of scalar branches and waits.
    // Item -> columns.  The fp32 image stores, inside every 16-wide k chunk, the 4 x 4 block (k-step, lane group) transposed
    // (lds_col): the four columns {16 kk + 4 q + g : q = 0..3} of lane group g sit at the CONSECUTIVE positions 16 kk + 4 g .. + 3.
    // An item is therefore (row, chunk kk, group g) with those four columns (round 4): ONE ds_write_b128 instead of four scattered
    // ds_write_b32, and consecutive threads read consecutive state floats / normaliser doubles (conflict free) where the old item
    // (four CONSECUTIVE columns) read with a stride of four (measured on cfg4', 393 columns x 32 rows: 8.3 us per turn in the step
    // trace, profiles/turn_trace.py).  Same arithmetic per element.  bf16x3 images keep consecutive columns (split3x4's layout).
#ifndef HIPETS_INPUT_BY_GROUP
#define HIPETS_INPUT_BY_GROUP 1
#endif
    auto build_input_impl = [&](const int t, float* const dst, auto norm_tag, auto plain_tag) __attribute__((always_inline)) {
        constexpr int NORM = decltype(norm_tag)::value;
        constexpr bool PLAIN = decltype(plain_tag)::value;
        // (shape-specialised instances WITH obs preprocessing keep the old items: their input is narrow, this function runs in their
        // prologue and turn-based flows only, and the R = 2 halfcheetah DEVICE instance -- at the 256-register limit of two waves per
        // SIMD -- spilt one VGPR to scratch with four sinf / cosf-bearing elements held for one store)
        constexpr bool kByGroup = HIPETS_INPUT_BY_GROUP && !kB3 && (PLAIN || !kLean);
        const float* actn_t = sm.actn + (t & 1) * n_act;
#ifndef HIPETS_INPUT_BATCHED
#define HIPETS_INPUT_BATCHED 1
#endif
        if constexpr (HIPETS_INPUT_BATCHED && kByGroup && PLAIN && NORM != HIPETS_NORM_F32) {
            // Round 5 (step trace of the turn-based DEVICE form, profiles/r5_turn_trace.json: 8.3 us per turn for cfg4''s 32 rows x 400
            // columns = 20 k cycles for 12.5 items per thread): the loop below is a chain of LDS round trips -- row id, then the value
            // (from the state or from the actions, behind a branch), then the two normaliser doubles, element after element.  Here an
            // item's twelve LDS reads are independent of each other (the value's source is a selected ADDRESS, not a branch) and issued
            // together, the (row, quad) of an item follows from the previous item's by addition; same arithmetic per element.
            const int d_s = kThreads / kq, d_c = kThreads - d_s * kq;
            int s = tid / kq, cq = tid - s * kq;
            for (int i = tid; i < ROWS * kq; i += kThreads) {
                const bool valid = sm.rowid[s] >= 0;
                float xv[4];
                double nm[4], ns[4];
                bool inb[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c = ((cq >> 2) << 4) + 4 * q + (cq & 3);
                    const int cc = min(c, md.in_dim - 1);
                    const float* src = cc < md.obs_in ? sm.state + s * md.obs_dim + cc : actn_t + s * md.act_dim + (cc - md.obs_in);
                    xv[q] = *src;
                    inb[q] = c < md.in_dim;
                    if constexpr (NORM == HIPETS_NORM_F64) { nm[q] = sm.nmean[cc]; ns[q] = sm.nstd[cc]; }
                }
                f32x4 v;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float x = xv[q];
                    if constexpr (NORM == HIPETS_NORM_F64) x = (float)(((double)x - nm[q]) * ns[q]);
                    v[q] = (inb[q] && valid) ? x : 0.f;
                }
                HIPETS_BOUND(s >= 0 && s < ROWS && 4 * cq + 3 < ld_in);
                *reinterpret_cast<f32x4*>(dst + s * ld_in + 4 * cq) = v;
                s += d_s;
                cq += d_c;
                if (cq >= kq) { cq -= kq; ++s; }
            }
            return;
        }
        for (int i = tid; i < ROWS * kq; i += kThreads) {
            const int s = i / kq, cq = i % kq;
            const bool valid = sm.rowid[s] >= 0;
            float v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = kByGroup ? ((cq >> 2) << 4) + 4 * q + (cq & 3) : 4 * cq + q;
                const int cc = min(c, md.in_dim - 1);  // clamped index: loads stay in bounds, result masked below
                float x;
                if (cc < md.obs_in) {
                    if constexpr (PLAIN) x = sm.state[s * md.obs_dim + cc];
                    else x = processed_obs(sm.state + s * md.obs_dim, cc, obs_process);
                } else {
                    x = actn_t[s * md.act_dim + (cc - md.obs_in)];
                }
                // f64 normaliser: (x - mean) * (1 / std) with the reciprocal formed once per launch in f64.  Against the reference's
                // f64 division the product is off by <= 1.5 ulp OF F64 before the rounding to f32: the f32 value differs (by one
                // f32 ulp) only when the quotient sits within ~2^-29 of a rounding boundary, ~1e-8 of the elements -- far inside
                // T1 -- and the per-element f64 division sequence (~12 f64 instructions) leaves the per-step critical path
                if constexpr (NORM == HIPETS_NORM_F64) x = (float)(((double)x - sm.nmean[cc]) * sm.nstd[cc]);
                else if constexpr (NORM == HIPETS_NORM_F32) x = (x - (float)sm.nmean[cc]) / (float)sm.nstd[cc];
                v[q] = (c < md.in_dim && valid) ? x : 0.f;
            }
            HIPETS_BOUND(s >= 0 && s < ROWS && 4 * cq + 3 < ld_in);
            if constexpr (kB3) {  // three bf16 pieces per value, in the B-operand layout of wave_gemm_b3
                u32x2 pc[3];
                split3x4(f32x4{v[0], v[1], v[2], v[3]}, pc);
                char* row = reinterpret_cast<char*>(dst) + (size_t)s * ld_in * 4;
#pragma unroll
                for (int p = 0; p < 3; ++p) *reinterpret_cast<u32x2*>(row + b3_offset(4 * cq, p)) = pc[p];
            } else if constexpr (kByGroup) {  // columns 16 kk + 4 q + g live at positions 16 kk + 4 g + q: item cq = 4 kk + g writes [4 cq, 4 cq + 3]
                *reinterpret_cast<f32x4*>(dst + s * ld_in + 4 * cq) = f32x4{v[0], v[1], v[2], v[3]};
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) dst[s * ld_in + lds_col(4 * cq + q)] = v[q];
            }
        }
    };
    // KSpec::WIDE (cfg4': 32 rows x 400 input columns per pass; f64 normaliser, no obs preprocessing -- facts of the shape): the pass
    // by COLUMN.  A thread owns columns tid, tid + 256, ... for every row: its normaliser constants and LDS position are loaded once
    // per column, consecutive lanes read consecutive state floats (no bank conflict) and write the 64 positions of four whole k chunks
    // (lds_col permutes inside a chunk: no conflict either).  The by-group items above read the four columns {16 kk + 4 q + g} per
    // lane: lanes 16 columns apart meet on a bank -- 4-way conflicts on the state reads, 8-way on the f64 constants -- and the pass
    // measured 7-8 us per turn in the step trace (profiles/r6_turn_trace.json: "arrived -> built"), a tenth of a cfg4' turn in BOTH
    // modes.  Same arithmetic per element: same bits.
#ifndef HIPETS_INPUT_BY_COLUMN
#define HIPETS_INPUT_BY_COLUMN 1
#endif
    int in_rows = ROWS;  // rows of the input image the next MLP pass reads (kTile in a one-tile turn: "ragged last turn" above)
    auto build_input_cols = [&](const int t, float* const dst) __attribute__((always_inline)) {
        const float* actn_t = sm.actn + (t & 1) * n_act;
        constexpr int kU = 8;  // rows per batch: the batch's LDS reads are issued together
        static_assert(ROWS % kU == 0, "row batches");
        for (int c = tid; c < Kp0; c += kThreads) {
            const bool inb = c < md.in_dim;
            const int cc = min(c, md.in_dim - 1);
            const double nm = sm.nmean[cc], ns = sm.nstd[cc];
            const bool from_state = cc < md.obs_in;
            const float* const src = from_state ? sm.state + cc : actn_t + (cc - md.obs_in);
            const int stride = from_state ? md.obs_dim : md.act_dim;
            float* const out = dst + lds_col(c);
            for (int s0_ = 0; s0_ < in_rows; s0_ += kU) {  // (a one-tile turn of the ragged last turn: row tile 0 only)
                float x[kU];
                int rid[kU];
#pragma unroll
                for (int u = 0; u < kU; ++u) {
                    x[u] = src[(s0_ + u) * stride];
                    rid[u] = sm.rowid[s0_ + u];
                }
#pragma unroll
                for (int u = 0; u < kU; ++u) out[(s0_ + u) * ld_in] = (inb && rid[u] >= 0) ? (float)(((double)x[u] - nm) * ns) : 0.f;
            }
        }
    };
    auto build_input = [&](const int t, float* const dst) __attribute__((always_inline)) {
        using T = std::true_type;
        using F = std::false_type;
        if constexpr (kWide && HIPETS_INPUT_BY_COLUMN) {  // (KSpec static_assert: WIDE instances have the f64 normaliser and no obs preprocessing)
            build_input_cols(t, dst);
            return;
        }
        const bool plain = obs_process == HIPETS_OBS_NONE;
        switch (normalizer) {
            case HIPETS_NORM_F64:
                if (plain) build_input_impl(t, dst, std::integral_constant<int, HIPETS_NORM_F64>{}, T{});
                else build_input_impl(t, dst, std::integral_constant<int, HIPETS_NORM_F64>{}, F{});
                break;
            case HIPETS_NORM_F32:
                if (plain) build_input_impl(t, dst, std::integral_constant<int, HIPETS_NORM_F32>{}, T{});
                else build_input_impl(t, dst, std::integral_constant<int, HIPETS_NORM_F32>{}, F{});
                break;
            default:
                if (plain) build_input_impl(t, dst, std::integral_constant<int, HIPETS_NORM_NONE>{}, T{});
                else build_input_impl(t, dst, std::integral_constant<int, HIPETS_NORM_NONE>{}, F{});
                break;
        }
    };
    // KSpec::FUSE, FAST form: the obs columns of the next step's input are written by the output layer's tail; the remaining
    // columns of the padded input -- the normalised actions of step t and the zero padding up to Kp0 -- come from here, element
    // by element (column obs_in - 1 and column obs_in may share a quad).  Same arithmetic as build_input_impl.
    // This thread's column of those (<= 16 action + padding columns, the usual case): column obs_in + (tid & 15), rows tid / 16 + 16 q --
    // no division, the column's normaliser constants and LDS position fixed for the launch.
    const int bac_c = md.obs_in + (tid & 15);
    const bool bac_fast = kFuse && !kWide && Kp0 - md.obs_in <= 16;
    const bool bac_live = bac_fast && bac_c < md.in_dim;  // an action column (else zero padding, or beyond Kp0: nothing to write)
    double bac_nm = 0.0, bac_ns = 0.0;  // read from LDS once the prologue has put the constants there (below)
    const int bac_pos = lds_col(min(bac_c, Kp0 - 1));
    // (fast path: by the waves 1 .. kWaves - 1 only -- the output layer deals its leftover units to wave 0 first, so the others
    // reach the layer's barrier early by at least one unit's k loop and this work disappears in that slack; 12 rows per pass)
    auto build_action_columns = [&](const int t, float* const dst) __attribute__((always_inline)) {
        const float* actn_t = sm.actn + (t & 1) * n_act;
        if (bac_fast) {
            if (wave != 0 && bac_c < Kp0) {
                constexpr int kRowsPerPass = (kThreads - 64) / 16;
                constexpr int kPasses = (ROWS + kRowsPerPass - 1) / kRowsPerPass;
                const int r0 = (tid - 64) >> 4;
                float x[kPasses];
                int rid[kPasses];
#pragma unroll
                for (int q = 0; q < kPasses; ++q) {  // all LDS reads first: one round trip for the thread's rows
                    const int s = r0 + kRowsPerPass * q;
                    rid[q] = s < ROWS ? act_rows[s] : -1;
                    x[q] = (bac_live && s < ROWS) ? actn_t[s * md.act_dim + (bac_c - md.obs_in)] : 0.f;
                }
#pragma unroll
                for (int q = 0; q < kPasses; ++q) {
                    const int s = r0 + kRowsPerPass * q;
                    if (s < ROWS) dst[s * ld_in + bac_pos] = (bac_live && rid[q] >= 0) ? (float)(((double)x[q] - bac_nm) * bac_ns) : 0.f;
                }
            }
            return;
        }
        const int ntc = Kp0 - md.obs_in;
        for (int i = tid; i < ROWS * ntc; i += kThreads) {
            const int s = i / ntc, c = md.obs_in + (i - s * ntc);
            float v = 0.f;
            if (c < md.in_dim && act_rows[s] >= 0) {
                const float x = actn_t[s * md.act_dim + (c - md.obs_in)];
                v = (float)(((double)x - sm.nmean[c]) * sm.nstd[c]);  // KSpec::FUSE instances: f64 normaliser (static_assert in KSpec)
            }
            dst[s * ld_in + lds_col(c)] = v;
        }
    };

    {   // the first step's actions are in flight while the state / totals / flags are fetched: one round trip for all of it
        // (the per-step launches of EXACT / DEVICE mode pay this prologue every step)
        float av[kPrefetch];
        fetch_actions_issue(ra.t_begin, av);
        load_initial_state();
        fetch_actions_commit(ra.t_begin, av);
    }
    __syncthreads();
    if (bac_live) { bac_nm = sm.nmean[bac_c]; bac_ns = sm.nstd[bac_c]; }
    if constexpr (kB3) kq = sm.lmeta[0].Kp32 >> 2;
    Pre pre;      // chunk-0 weight fragments + biases of the NEXT linear op of this wave (kPre kernels)
    NextOp cur_op;  // ... and that op's descriptor
    cur_op.valid = false;
    if constexpr (kPre) {
        const int m0 = fast ? __builtin_amdgcn_readfirstlane(sm.sched[ra.t_begin]) : member_dom;
        cur_op = describe_layer<R, S>(md, sm.lmeta, 0, m0, wave);
        prefetch_issue(cur_op, lane, pre);
    }
    build_input(ra.t_begin, sm.buf0);
    __syncthreads();
    prof.mark(0);
    float* step_in = sm.buf0;  // LDS image of the current step's model input (KSpec::FUSE: alternates, see the output layer below)

    // Persistent DEVICE form with more logical workgroups than launched ones: within every step this workgroup serves its
    // logical workgroups in turn (sequence index q = step * n_serve + turn); each turn collects its rows from the hand-over
    // table, runs the step, publishes.  Only a step's first turn can find rows missing (published by other workgroups' last
    // turns of the previous step); the later turns' rows arrived while the earlier ones computed.
    const int n_serve = persist ? (n_logical - wg + (int)gridDim.x - 1) / (int)gridDim.x : 1;
    const int n_seq = (ra.t_end - ra.t_begin) * n_serve;
    int stamp_seq = 0;  // (profiling builds: index of the current (step, turn) for HIPETS_STAMP)
    (void)stamp_seq;
    // Straight persistent form (KSpec::FUSE instances, every launched workgroup serving exactly one logical workgroup, >= 3 hidden
    // layers): see the step loop.  The slot's rows of the current and of the next step live in two LDS arrays that swap roles.
#ifdef HIPETS_DBG_NOSTRAIGHT
    const bool straight = false;
#else
    const bool straight = kFuse && !kDmaCollect && persist && ra.n_logical == (int)gridDim.x && md.n_layers >= 4;
#endif
    int* const rows_a = sm.rowid;
    int* const rows_b = sm.pend + ROWS;
    // fused hopper termination, persistent DEVICE form: per-row "the state this row arrived with is unhealthy" flags, raised by the
    // collecting threads, consumed by the next tail (the fused instances never use sm.lrew: the learned reward stays in registers)
    int* const hop_flags = reinterpret_cast<int*>(sm.lrew);
    // Collect the rows `rows` holds (published by their previous owners in step t_next - 1): every thread polls its (row slot, pair)
    // items as in the general form below, and the thread that receives a pair of state dims also writes them -- normalised exactly
    // like build_input_impl's f64 form -- into the next step's input image: no separate input pass, two barriers less per step.
    auto collect_straight = [&](const int t_next, const int* const rows, float* const dst) __attribute__((always_inline)) {
        const unsigned want = ra.tag_base + (unsigned)t_next;
        for (int base = 0; base < ROWS * NVP; base += kG * kThreads) {
            const unsigned long long* src[kG];
            u32x4g g[kG];
            int gs[kG], gv[kG];
            bool soft[kG], live[kG];
            double nm[kG][2], ns[kG][2];
#pragma unroll
            for (int q = 0; q < kG; ++q) {
                const int i = base + tid + q * kThreads;
                gs[q] = base == 0 ? xs[q] : (i < ROWS * NVP ? i / NVP : -1);
                gv[q] = base == 0 ? xv[q] : (i < ROWS * NVP ? i - (i / NVP) * NVP : 0);
                soft[q] = gv[q] == NVP - 1;
                src[q] = nullptr;
                live[q] = false;
                g[q] = u32x4g{0u, want, 0u, want};
                if (gs[q] >= 0) {
                    const int rid = rows[gs[q]];
                    if (rid >= 0) { src[q] = ra.exchange + (size_t)rid * NV + 2 * gv[q]; live[q] = true; }
                }
                // the normaliser constants of the input columns the pair's two dims feed (ObsMap): requested now, used when the pair has arrived
                const int d = soft[q] ? 0 : 2 * gv[q];
                const int i0 = max(0, min(ObsMap<S::OBSP>::col(d), md.obs_in - 1)), i1 = max(0, min(ObsMap<S::OBSP>::col(d + 1), md.obs_in - 1));
                nm[q][0] = sm.nmean[i0]; nm[q][1] = sm.nmean[i1];
                ns[q][0] = sm.nstd[i0]; ns[q][1] = sm.nstd[i1];
            }
            const long long t_poll = wall_clock64();
            for (int spins = 0;; ++spins) {
                static_assert(kG == 2, "the wait below names the two destinations");
                u32x4g got[kG];
                pair_load_issue(got[0], src[0] ? src[0] : ra.exchange);
                pair_load_issue(got[1], src[1] ? src[1] : ra.exchange);
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(got[0]), "+v"(got[1])::"memory");
                bool ready = true;
#pragma unroll
                for (int q = 0; q < kG; ++q)
                    if (src[q]) {
                        g[q] = got[q];
                        if (got[q][1] == want && got[q][3] == want) src[q] = nullptr;
                        else if (!soft[q]) ready = false;
                    }
                if (ready) break;
                if ((poll_every || (spins & 63) == 63) && (wall_clock64() - t_poll > ra.poll_ticks || *(volatile int*)ra.error_flag)) {
                    *ra.error_flag = 1;
                    break;
                }
                __builtin_amdgcn_s_sleep(8);
            }
#pragma unroll
            for (int q = 0; q < kG; ++q)
                if (gs[q] >= 0) {
                    HIPETS_BOUND(gs[q] < ROWS && gv[q] >= 0 && gv[q] < NVP);
                    if (!soft[q]) {
                        using OM = ObsMap<S::OBSP>;
                        const int d = 2 * gv[q];
                        const float v0 = __uint_as_float(g[q][0]), v1 = __uint_as_float(g[q][2]);
                        float x0 = v0, x1 = v1;
                        if constexpr (OM::kTrigDim >= 0) {
                            if (d == OM::kTrigDim || d + 1 == OM::kTrigDim) {  // this thread holds the dim that enters as sin and cos (processed_obs's sinf / cosf)
                                const float tv = d == OM::kTrigDim ? v0 : v1;
                                const float sv = sinf(tv), cv = cosf(tv);
                                if (d == OM::kTrigDim) x0 = sv; else x1 = sv;
                                dst[gs[q] * ld_in + lds_col(OM::kCosCol)] = live[q] ? (float)(((double)cv - sm.nmean[OM::kCosCol]) * sm.nstd[OM::kCosCol]) : 0.f;
                            }
                        }
                        if constexpr (S::TERM == HIPETS_TERM_HOPPER) {
                            if (live[q] && hopper_pair_bad(d, v0, v1, true, d + 1 < md.obs_dim)) hop_flags[gs[q]] = 1;
                        }
                        const int c0 = OM::col(d), c1 = OM::col(d + 1);
                        sm.state[gs[q] * md.obs_dim + d] = v0;
                        if (c0 >= 0) dst[gs[q] * ld_in + lds_col(c0)] = live[q] ? (float)(((double)x0 - nm[q][0]) * ns[q][0]) : 0.f;
                        if (d + 1 < md.obs_dim) {
                            sm.state[gs[q] * md.obs_dim + d + 1] = v1;
                            if (c1 >= 0) dst[gs[q] * ld_in + lds_col(c1)] = live[q] ? (float)(((double)x1 - nm[q][1]) * ns[q][1]) : 0.f;
                        }
                    } else if (src[q]) {
                        sm.pend[gs[q]] = 1;  // not there yet: the next tail fetches the pair
                    } else {
                        sm.tot[gs[q]] = __uint_as_float(g[q][0]);
                        sm.term[gs[q]] = (int)g[q][2];
                    }
                }
        }
        HIPETS_STAMP(2, t_next - 1);  // this thread's rows have arrived
    };
    // ---- KSpec::WIDE: collect the rows `sm.rowid` names (published with tag `want`) by LDS-DMA ---------------------------------------
    // Wave w owns the row slots w, w + kWaves, ... from issue to commit: a row's NVP pairs are CPR = ceil(NVP / 64) chunks of 64 (1 KiB
    // of staging each; slot s, chunk k sits at staging chunk s CPR + k), the row id is wave-uniform, a lane's source is the row's base +
    // its pair -- no division, no per-lane row lookup (the first version dealt chunks of 64 CONSECUTIVE items to the waves and paid an
    // LDS round trip for the row id and a multiply-high per chunk and lane: 12.2 us per collect against the register path's 9.6,
    // profiles/r6_turn_trace.json).  The wave's own s_waitcnt vmcnt(0) is all that orders its ds_reads behind its DMAs (no barrier).  A
    // chunk with a lane whose pair has not been published yet (its tags are an earlier step's) is fetched again; only a step's first
    // turn can see that.  The {running total, flag} pair of a row gets one look per pass and is left to the next tail if late
    // (sm.pend), exactly like the register path.
    auto dma_collect = [&](const unsigned want) __attribute__((always_inline)) {
        // chunks per row: a compile-time fact of the WIDE shapes (47 output column tiles <=> obs 369..376 <=> 186..189 pairs <=> 3 chunks)
        constexpr int CPR = kWide ? (((S::OUTC > 0 ? S::OUTC : 1) * 8 + 1) / 2 + 1 + 63) / 64 : 1;
        constexpr int kRowsPerWave = ROWS / kWaves;
        static_assert(ROWS % kWaves == 0 && kRowsPerWave * CPR <= 32, "row slots are dealt to the waves; one pending bit per (row, chunk)");
        const int n_main = (int)((align16((size_t)ROWS * ld_in * 4) + align16((size_t)ROWS * ld_k * 4)) >> 10);  // chunks that fit buf0 + buf1 (contiguous)
        char* const stage0 = reinterpret_cast<char*>(sm.buf0);
        HIPETS_BOUND(CPR == (NVP + 63) / 64 && (size_t)max(ROWS * CPR - n_main, 0) * 1024 <= dma_stage_extra_bytes(ROWS, ld_k, ld_in, md.obs_dim));
        auto chunk_ptr = [&](const int c) __attribute__((always_inline)) { return c < n_main ? stage0 + ((size_t)c << 10) : sm.stage_x + ((size_t)(c - n_main) << 10); };
        auto lds_of = [&](const int c) __attribute__((always_inline)) { return (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)chunk_ptr(c)); };
        unsigned pending = 0;  // bit j CPR + k: chunk k of this wave's j-th row still has a lane waiting
        int rids[kRowsPerWave];
#pragma unroll
        for (int j = 0; j < kRowsPerWave; ++j) rids[j] = sm.rowid[wave + kWaves * j];  // (all of the wave's row ids in one LDS round trip)
#pragma unroll
        for (int j = 0; j < kRowsPerWave; ++j) {
            const int s = wave + kWaves * j;
            const int rid = __builtin_amdgcn_readfirstlane(rids[j]);
            rids[j] = rid;
            if (rid >= 0) {
                const unsigned long long* const row = ra.exchange + (size_t)rid * NV;
#pragma unroll
                for (int k = 0; k < CPR; ++k) {
                    const int v = min(64 * k + lane, NVP - 1);  // (lanes beyond the row's last pair fetch it again and ignore it)
                    pair_dma_issue(row + 2 * v, lds_of(s * CPR + k));
                }
                pending |= ((1u << CPR) - 1u) << (j * CPR);
            } else {  // a row of the padding: zero state, total, flag
                for (int d = lane; d < md.obs_dim; d += 64) sm.state[s * md.obs_dim + d] = 0.f;
                if (lane == 0) { sm.tot[s] = 0.f; sm.term[s] = 0; sm.pend[s] = 0; }
            }
        }
        const long long t_poll = wall_clock64();
        for (int spins = 0; pending; ++spins) {
            vmem_drain();  // this wave's DMAs have landed
            unsigned still = 0;
#pragma unroll
            for (int j = 0; j < kRowsPerWave; ++j) {
                if (!((pending >> (j * CPR)) & ((1u << CPR) - 1u))) continue;  // (wave-uniform) nothing of this row is waiting
                const int s = wave + kWaves * j;
                u32x4g g[CPR];
#pragma unroll
                for (int k = 0; k < CPR; ++k) g[k] = *reinterpret_cast<const u32x4g*>(chunk_ptr(s * CPR + k) + lane * 16);  // the row's chunks in one LDS round trip
#pragma unroll
                for (int k = 0; k < CPR; ++k) {
                    if (!((pending >> (j * CPR + k)) & 1u)) continue;  // (wave-uniform)
                    const int v = 64 * k + lane;
                    const bool mine = v < NVP, soft = v == NVP - 1;
                    const bool ok = g[k][1] == want && g[k][3] == want;
                    HIPETS_BOUND(s < ROWS);
                    if (mine && !soft) {
                        if (ok) {
                            const int d = 2 * v;
                            sm.state[s * md.obs_dim + d] = __uint_as_float(g[k][0]);
                            if (d + 1 < md.obs_dim) sm.state[s * md.obs_dim + d + 1] = __uint_as_float(g[k][2]);
                        }
                    } else if (soft) {
                        if (ok) {
                            sm.tot[s] = __uint_as_float(g[k][0]);
                            sm.term[s] = (int)g[k][2];
                        }
                        sm.pend[s] = ok ? 0 : 1;  // not there yet: the next tail fetches the pair
                    }
                    if (__builtin_amdgcn_ballot_w64(mine && !soft && !ok) != 0) {  // somebody's state pair is still an earlier step's: fetch the chunk again
                        still |= 1u << (j * CPR + k);
                        pair_dma_issue(ra.exchange + (size_t)rids[j] * NV + 2 * min(v, NVP - 1), lds_of(s * CPR + k));
                    }
                }
            }
            pending = still;
            if (!pending) break;
            if ((poll_every || (spins & 63) == 63) && (wall_clock64() - t_poll > ra.poll_ticks || *(volatile int*)ra.error_flag)) {
                *ra.error_flag = 1;
                vmem_drain();  // nothing of this wave may still be landing in the activation buffers when the flow goes on
                break;
            }
            __builtin_amdgcn_s_sleep(8);
        }
    };
    (void)dma_collect;

    bool one_tile = false;  // this turn serves a one-tile logical workgroup (kRagged; workgroup-uniform)
    for (int q_seq = 0; q_seq < n_seq; ++q_seq) {
        stamp_seq = q_seq;
        const int t = ra.t_begin + q_seq / n_serve;
        const bool more = t + 1 < ra.t_end;
        const bool has_next = q_seq + 1 < n_seq;  // persistent form: another (step, turn) follows
        const int t_next = ra.t_begin + (q_seq + 1) / n_serve;
        const int v_next = wg + ((q_seq + 1) - ((q_seq + 1) / n_serve) * n_serve) * (int)gridDim.x;  // its logical workgroup
        float av[kPrefetch];
        if (more && !persist) fetch_actions_issue(t + 1, av);  // consumed after the sampling phase: the HBM / L2 latency hides behind the MLP
        unsigned long long* const handover = (more && persist) ? ra.exchange : nullptr;
        const unsigned long long handover_tag = (unsigned long long)(ra.tag_base + (unsigned)t + 1u) << 32;  // tags never repeat across launches
        if constexpr (kFuse) {
            // ---- KSpec::FUSE: hidden layers as usual; the OUTPUT layer's accumulators go straight into the step's tail ----------
            const int member = fast ? __builtin_amdgcn_readfirstlane(sm.sched[t]) : member_dom;  // wave-uniform
            float* cur = step_in;
            float* nxt = step_in == sm.buf0 ? sm.buf1 : sm.buf0;
            const int L = md.n_layers;
            const bool write_input = more && !persist && !kWide;  // FAST form: rows stay here, the next step's input is built in place
            const bool wide_fast_next = kWide && more && !persist;  // WIDE: buf0 is the input image AND an activation buffer: the input is built after the step
            // Straight persistent form (every launched workgroup serves ONE logical workgroup): everything of step t + 1 that does
            // not depend on the rows' states is prepared while step t computes -- which rows the slot holds then (the step's keyed
            // permutation, evaluated beside layer 1 by the wave with the lightest GEMM share), their actions (fetched behind layer 2, in LDS before
            // the output layer, their input columns built beside it) -- so that between the output layer and the next step there
            // is only: barrier, wait for the rows, normalise them into the input image as they arrive, barrier.
            const bool prep_next = straight && more;
            int* const rows_nxt = sm.rowid == rows_a ? rows_b : rows_a;
            for (int l = 0; l + 1 < L; ++l) {
                // the raw actions of step t + 1 (requested at the top of the step) go to their LDS buffer now: visible to every
                // thread after this layer's barrier, i.e. when the output layer starts
                prof.mark(12);
                if (l == 1 && prep_next && wave == kWaves - 1) {  // this wave's share of a hidden layer is one unit short: room for the permutation
                    for (int s = lane; s < ROWS; s += 64) {
                        const int j = (wg % ra.groups) * ROWS + s;
                        rows_nxt[s] = j < ra.rows_per_domain
                                          ? (int)perm_apply((unsigned)(domain * ra.rows_per_domain + j), ra.perm_n, ra.perm_a, ra.perm_b, ra.step_keys[t + 1]) : -1;
                    }
                }
                if (l == 2 && prep_next) {  // rows_nxt is visible since layer 1's barrier
                    act_rows = rows_nxt;
                    compute_act_base();
                    fetch_actions_issue(t + 1, av);
                }
                if (l == L - 2 && (write_input || prep_next || wide_fast_next)) fetch_actions_commit(t + 1, av);
                if constexpr (kRagged) {
                    if (one_tile) mlp_layer<1, S>(md, sm.lmeta, l, member, cur, nxt, wave, lane, prof, sm.part);
                    else mlp_layer<R, S>(md, sm.lmeta, l, member, cur, nxt, wave, lane, prof, sm.part);
                } else {
                    mlp_layer<R, S>(md, sm.lmeta, l, member, cur, nxt, wave, lane, prof, sm.part);
                }
                __syncthreads();
                prof.mark(8);
                float* tmp = cur; cur = nxt; nxt = tmp;
            }
            // `nxt` (the output layer's would-be LDS image) is read by nobody while the output layer runs: it receives the next
            // step's model input -- action columns and zero padding from all threads here, the obs columns from the tail lanes
            // (FAST) / from the threads that receive the rows (straight persistent form)
            if (write_input || (prep_next && !kWide)) build_action_columns(t + 1, nxt);
            const float* const actn_t = sm.actn + (t & 1) * n_act;
            const unsigned handover_tg = (unsigned)(handover_tag >> 32);
            // One finished accumulator = column tile c, row tile r of the head-pair pack: this lane (group g, row j) holds
            // {mean d0, mean d0 + 1, logvar d0, logvar d0 + 1} of batch row s = 16 r + j for d0 = 8 c + 2 g.  Same arithmetic, op for
            // op, as the LDS-based phases of the other instances (sample_impl / reward phase / build_input_impl below): the
            // shape-specialised and the generic kernels return the same bits.  BRANCH-FREE on purpose: every lane computes, loads
            // use clamped indices, stores of inactive lanes go to a dump slot -- so the two or three tails of a wave are one
            // basic block and the scheduler overlaps their LDS round trips instead of paying them one after the other.
            auto tail_prep = [&](FusedSlot& q, const int c, const int r) __attribute__((always_inline)) {
                const int s = r * kTile + (lane & 15);
                const int d0 = 8 * c + 2 * (lane >> 4);
                const int dA = min(d0, md.out_dim - 1), dB = min(d0 + 1, md.out_dim - 1);  // clamped: loads stay in bounds
                const int oA = min(d0, md.obs_dim - 1), oB = min(d0 + 1, md.obs_dim - 1);
                q.rid = sm.rowid[s];
                q.mxA = sm.maxlv[dA]; q.mxB = sm.maxlv[dB]; q.mnA = sm.minlv[dA]; q.mnB = sm.minlv[dB];
                q.pA = sm.state[s * md.obs_dim + oA]; q.pB = sm.state[s * md.obs_dim + oB];
                q.ndA = sm.nodelta[oA]; q.ndB = sm.nodelta[oB];
                // the normaliser constants of the input COLUMNS the two dims feed (obs preprocessing moves them: ObsMap)
                using OM = ObsMap<S::OBSP>;
                const int iA = max(0, min(OM::col(d0), md.obs_in - 1)), iB = max(0, min(OM::col(d0 + 1), md.obs_in - 1));
                q.nmA = sm.nmean[iA]; q.nmB = sm.nmean[iB]; q.nsA = sm.nstd[iA]; q.nsB = sm.nstd[iB];
            };
            // The standard normals of two units at once.  A lane's two dims (d0 = 8 c + 2 g, d0 + 1) take HALF of Philox block (row, step,
            // d0 / 4) -- x, y for an even lane group g, z, w for an odd one (exactly rollout_normals4's assignment) -- and the other half
            // belongs to the lane 16 further (g ^ 1: same row, the block's other two dims).  Until round 6 every lane computed the whole
            // block of every unit and dropped half of it (the ten rounds are half of a unit's VALU time).  Now, for a PAIR of units (a, b),
            // the even lane groups compute a's block and the odd ones b's, and each lane fetches the half it lacks from its partner's
            // registers (two ds_bpermute): one block per lane and pair instead of two.  Same blocks, same halves, same Box-Muller: same bits.
            // A group's odd unit out (`two` false) draws as before.  Three instances keep the draw inside tail_unit: the two-tile DEVICE-mode
            // instances with obs preprocessing or a learned reward (pets_halfcheetah, pets_pusher / pets_reacher, pets_mppi_halfcheetah in
            // DEVICE mode) sit at the 256-register limit of two workgroups per CU, and with the pair's exchange they spilt 3-14 registers to
            // scratch memory (the build's resource report; pets_halfcheetah 0.498 -> 0.487 of peak; tests/test_abi.py allows no kernel any).
            constexpr bool kPairDraws = HIPETS_SHARED_DRAWS != 0 &&
                                        !(MinWaves<R, S>::value == 2 && R == 2 && S::KMODE != HIPETS_MODE_FAST && (S::OBSP != HIPETS_OBS_NONE || S::REW == HIPETS_REW_LEARNED));
            auto unit_draw = [&](const int rid, const int c, float& n0, float& n1) __attribute__((always_inline)) {  // one unit, the whole block per lane
                const int g = lane >> 4;
                const bool odd = (g & 1) != 0;
                const Philox4 r4 = philox4x32_10((uint32_t)rid, (uint32_t)t, (uint32_t)((8 * c + 2 * g) >> 2), (uint32_t)ra.stream_id, (uint32_t)ra.seed,
                                                 (uint32_t)(ra.seed >> 32) ^ (uint32_t)(ra.stream_id >> 32));
                box_muller(odd ? r4.z : r4.x, odd ? r4.w : r4.y, n0, n1);
            };
            auto tail_draw = [&](FusedSlot& qa, FusedSlot& qb, const int ca, const int cb, const bool two) __attribute__((always_inline)) {
#ifndef HIPETS_TIMING_NO_DRAWS
#define HIPETS_TIMING_NO_DRAWS 0  // 1 = TIMING-ONLY builds (results are wrong on purpose): the tail draws nothing -- an upper bound of what moving
#endif                            // the draws off the step's critical path (e.g. into the hand-over wait) could gain; profiles/headline_probe.py
                if constexpr (HIPETS_TIMING_NO_DRAWS) {
                    qa.n0 = 0.37f + 1e-3f * (float)(qa.rid & 7);
                    qa.n1 = -0.81f;
                    qb.n0 = 0.37f + 1e-3f * (float)(qb.rid & 7);
                    qb.n1 = -0.81f;
                    return;
                }
                if constexpr (!kPairDraws) return;  // (drawn by tail_unit)
                const int g = lane >> 4;
                const bool odd = (g & 1) != 0;
                const uint32_t k0 = (uint32_t)ra.seed, k1 = (uint32_t)(ra.seed >> 32) ^ (uint32_t)(ra.stream_id >> 32);
                if (!two) {
                    unit_draw(qa.rid, ca, qa.n0, qa.n1);
                    return;
                }
                const Philox4 r4 = philox4x32_10((uint32_t)(odd ? qb.rid : qa.rid), (uint32_t)t, (uint32_t)((8 * (odd ? cb : ca) + 2 * g) >> 2),
                                                 (uint32_t)ra.stream_id, k0, k1);
                // what the partner lacks of this lane's block: an even lane holds a's block, its (odd) partner wants z, w; an odd lane
                // holds b's block, its (even) partner wants x, y
                const int partner = ((lane ^ 16) & 63) << 2;
                const uint32_t t0 = (uint32_t)__builtin_amdgcn_ds_bpermute(partner, (int)(odd ? r4.x : r4.z));
                const uint32_t t1 = (uint32_t)__builtin_amdgcn_ds_bpermute(partner, (int)(odd ? r4.y : r4.w));
                box_muller(odd ? t0 : r4.x, odd ? t1 : r4.y, qa.n0, qa.n1);  // unit a: (x, y) of a's block on even lanes, (z, w) -- from the partner -- on odd ones
                box_muller(odd ? r4.z : t0, odd ? r4.w : t1, qb.n0, qb.n1);  // unit b: (x, y) of b's block -- from the partner -- on even lanes, (z, w) on odd ones
            };
            auto tail_unit = [&](const FusedSlot& q, const f32x4 a, const int c, const int r) __attribute__((always_inline)) {
                const int g = lane >> 4, j = lane & 15;
                const int s = r * kTile + j;
                const int d0 = 8 * c + 2 * g;
                const int rid = q.rid;
                HIPETS_BOUND(s >= 0 && s < ROWS && r >= 0 && r < R && c >= 0 && 16 * c < 2 * md.out_dim + 16 && rid < ra.B);
                const bool okA = rid >= 0 && d0 < md.obs_dim, okB = rid >= 0 && d0 + 1 < md.obs_dim;
                const float mxA = q.mxA, mxB = q.mxB, mnA = q.mnA, mnB = q.mnB, pA = q.pA, pB = q.pB;
                const bool addA = md.target_is_delta && !q.ndA, addB = md.target_is_delta && !q.ndB;
                const double nmA = q.nmA, nmB = q.nmB, nsA = q.nsA, nsB = q.nsB;
                float n0 = q.n0, n1 = q.n1;  // the two normals of (row, step, dims d0, d0 + 1): tail_draw ...
                if constexpr (!kPairDraws && !HIPETS_TIMING_NO_DRAWS) unit_draw(rid, c, n0, n1);  // ... or drawn here (two workgroups per CU)
                float lvA = a[2], lvB = a[3];
                lvA = mxA - softplus_fast(mxA - lvA);  // gaussian_mlp.py:152
                lvB = mxB - softplus_fast(mxB - lvB);
                lvA = mnA + softplus_fast(lvA - mnA);  // :153
                lvB = mnB + softplus_fast(lvB - mnB);
                const float predA = a[0] + __builtin_amdgcn_sqrtf(exp_hw(lvA)) * n0;  // model.py:471-473
                const float predB = a[1] + __builtin_amdgcn_sqrtf(exp_hw(lvB)) * n1;
                const float vA = predA + (addA ? pA : 0.f);  // one_dim_tr_model.py:281-286 (0 for no_delta dims)
                const float vB = predB + (addB ? pB : 0.f);
                sm.state[okA ? s * md.obs_dim + d0 : (int)(sm.dump - sm.state)] = vA;
                sm.state[okB ? s * md.obs_dim + d0 + 1 : (int)(sm.dump - sm.state) + 1] = vB;
                if (write_input) {  // wave-uniform; build_input_impl's f64 form (the columns ObsMap names: input column d = obs dim d without preprocessing)
                    using OM = ObsMap<S::OBSP>;
                    const int cA = OM::col(d0), cB = OM::col(d0 + 1);
                    float xA = vA, xB = vB;
                    if constexpr (OM::kTrigDim >= 0) {
                        if (c == 0) {  // wave-uniform: the trig dim lives in column tile 0.  Same sinf / cosf as processed_obs: same bits as the generic kernel
                            const bool trigA = d0 == OM::kTrigDim, trigB = d0 + 1 == OM::kTrigDim;
                            const float tv = trigA ? vA : vB;
                            const float sv = sinf(tv), cv = cosf(tv);
                            xA = trigA ? sv : xA;
                            xB = trigB ? sv : xB;
                            const bool okT = (trigA && okA) || (trigB && okB);
                            nxt[okT ? s * ld_k + lds_col(OM::kCosCol) : (int)(sm.dump - nxt) + 1] = (float)(((double)cv - sm.nmean[OM::kCosCol]) * sm.nstd[OM::kCosCol]);
                        }
                    }
                    nxt[(okA && cA >= 0) ? s * ld_k + lds_col(max(cA, 0)) : (int)(sm.dump - nxt) + 2] = (float)(((double)xA - nmA) * nsA);
                    nxt[(okB && cB >= 0) ? s * ld_k + lds_col(max(cB, 0)) : (int)(sm.dump - nxt) + 3] = (float)(((double)xB - nmB) * nsB);
                }
                const unsigned pubA = okA ? __float_as_uint(vA) : 0u, pubB = okB ? __float_as_uint(vB) : 0u;
                // persistent DEVICE form: the row's next owner waits for these values.  Under a real (divergent) predicate: redirecting
                // the inactive lanes' stores to one spare row instead made every workgroup's write-through stores queue on ONE
                // address (measured: 1.375 vs 1.081 ms per cfg2 rollout)
                if (handover && okA) pair_store(handover + (size_t)rid * NV + d0, pubA, pubB, handover_tg);
                // Reward, termination, masked accumulation (model_env.py:124-129, :186-188) by ONE lane per row: closed forms -- the lane
                // that holds dims 0, 1 of the row (dims 2, 3 sit in the next lane group of the same accumulator); learned rewards -- the
                // lane that holds output column obs_dim, whose sampled value IS the reward (one_dim_tr_model.py:287).  (A wave can hold
                // several such units -- one per row tile when the output layer has >= 4 column tiles -- hence here, per unit.)
                constexpr bool kLearnedRew = S::REW == HIPETS_REW_LEARNED;
                constexpr bool kAllDims = S::TERM == HIPETS_TERM_HOPPER;  // every lane judges its own dims; flags through LDS, folded in one step later (KSpec)
                constexpr bool kRewLane = kLearnedRew && (S::TERM == HIPETS_TERM_NONE || kAllDims);  // the reward column's lane keeps the total (else: the lane with dims 0, 1)
                if constexpr (kAllDims) {  // hopper: all dims finite, |dims 1..| < 100, height (dim 0) > 0.7, |angle (dim 1)| < 0.2
                    // (persistent form: the row's next owner judges the dims it receives -- collect phases below)
                    if (!persist && hopper_pair_bad(d0, vA, vB, okA, okB)) sm.pend[(t & 1) * ROWS + s] = 1;  // (every writer stores the same value; read after this step's barrier)
                }
                const int c_rew = kRewLane ? (md.obs_dim >> 3) : 0, g_rew = kRewLane ? ((md.obs_dim & 7) >> 1) : 0;
                if (c == c_rew) {  // wave-uniform
                    float st[4] = {0.f, 0.f, 0.f, 0.f};
                    float lrew = (md.obs_dim & 1) ? predB : predA;  // the learned reward, on the lane that holds output column obs_dim (= sample_impl's sm.lrew[s])
                    if constexpr (!kRewLane) {
                        st[0] = okA ? vA : 0.f;
                        st[1] = okB ? vB : 0.f;
                        st[2] = __uint_as_float((unsigned)__builtin_amdgcn_ds_bpermute(((lane + 16) & 63) << 2, (int)pubA));
                        st[3] = __uint_as_float((unsigned)__builtin_amdgcn_ds_bpermute(((lane + 16) & 63) << 2, (int)pubB));
                        if constexpr (kLearnedRew)  // column obs_dim sits (md.obs_dim & 7) / 2 lane groups further in this accumulator
                            lrew = __uint_as_float((unsigned)__builtin_amdgcn_ds_bpermute(((lane + 16 * ((md.obs_dim & 7) >> 1)) & 63) << 2, (int)__float_as_uint(lrew)));
                    }
                    if (g == g_rew && rid >= 0) {
                        float tot = sm.tot[s];
                        int trm = sm.term[s];
                        if (persist && sm.pend[s]) {  // collected late: the previous owner published them after ITS output layer (normally long arrived)
                            sm.pend[s] = 0;
                            const unsigned long long* const src = ra.exchange + (size_t)rid * NV + (NV - 2);
                            const unsigned want = ra.tag_base + (unsigned)t;
                            const long long t_poll = wall_clock64();
                            u32x4g gq = {0u, 0u, 0u, 0u};
                            for (int spins = 0;; ++spins) {
                                pair_load_issue(gq, src);
                                asm volatile("s_waitcnt vmcnt(0)" : "+v"(gq)::"memory");
                                if (gq[1] == want && gq[3] == want) break;
                                if ((poll_every || (spins & 63) == 63) && (wall_clock64() - t_poll > ra.poll_ticks || *(volatile int*)ra.error_flag)) {
                                    *ra.error_flag = 1;
                                    break;
                                }
                                __builtin_amdgcn_s_sleep(8);
                            }
                            tot = __uint_as_float(gq[0]);
                            trm = (int)gq[2];
                        }
                        float rwd;
                        if constexpr (kLearnedRew) rwd = lrew;
                        else rwd = reward_eval(st, actn_t + s * md.act_dim, 4, md.act_dim, S::REW, 0.f);
                        bool done = false;
                        if constexpr (kAllDims) {  // `terminated` up to and including step t - 1: that step's flag is complete since its barrier
                            if (persist) {  // raised by the threads that collected this row's state (the state step t - 1 left)
                                trm = trm | hop_flags[s];
                                hop_flags[s] = 0;  // (raised again by the next collect phase: a barrier away)
                            } else {
                                int* const flag = sm.pend + ((t & 1) ^ 1) * ROWS + s;
                                trm = trm | (t > ra.t_begin ? *flag : 0);
                                *flag = 0;  // (raised again in step t + 1 at the earliest: two barriers away)
                            }
                        } else {
                            done = term_eval(st, 4, S::TERM);
                        }
                        if (trm) rwd = 0.f;
                        trm = trm | (done ? 1 : 0);
                        tot += rwd;
                        sm.term[s] = trm;
                        sm.tot[s] = tot;
                        if (handover) pair_store(handover + (size_t)rid * NV + (NV - 2), __float_as_uint(tot), (unsigned)trm, handover_tg);
                        else if (persist) ra.totals[rid] = tot;  // last step: the row's return
                    }
                }
            };
            auto tail_finish = [&]() __attribute__((always_inline)) {};
            const auto tail = make_tail(tail_prep, tail_draw, tail_unit, tail_finish);
            prof.mark(12);
            if constexpr (kRagged) {
                if (one_tile) mlp_output_layer_fused<1, S>(md, sm.lmeta, member, cur, wave, lane, prof, tail, sm.part);
                else mlp_output_layer_fused<R, S>(md, sm.lmeta, member, cur, wave, lane, prof, tail, sm.part);
            } else {
                mlp_output_layer_fused<R, S>(md, sm.lmeta, member, cur, wave, lane, prof, tail, sm.part);
            }
            // straight persistent form: what the two sides of this barrier exchange goes through LDS; the tail's write-through
            // hand-over stores need not have been acknowledged (__syncthreads() would wait for that -- about a microsecond --
            // before the first poll for the incoming rows is even issued; this way the two round trips overlap)
            if (prep_next) lds_barrier();
            else __syncthreads();
            prof.mark(8);
            HIPETS_STAMP(0, t);  // the MLP and the step's tail are done
            step_in = kWide ? sm.buf0 : nxt;
            if (wide_fast_next) {  // the tail wrote the new states; the input image of step t + 1 from them (buf0 is free again)
                build_input(t + 1, step_in);
                __syncthreads();
            }
            if (prep_next) {
                HIPETS_STAMP(1, t);
                if (kWide) build_action_columns(t + 1, step_in);  // (not beside the output layer: buf0 may be the buffer it reads)
                collect_straight(t + 1, rows_nxt, step_in);
                sm.rowid = rows_nxt;  // the slot's rows from here on
                __syncthreads();
                HIPETS_STAMP(3, t);
                continue;
            }
            if (persist && has_next) {  // the slot's row in the next turn (the tail above was the last reader of this turn's rowid)
                int nd, nj0, nlive;
                logical_rows(v_next, nd, nj0, nlive);
                if constexpr (kRagged) {
                    one_tile = nlive < ROWS;
                    in_rows = nlive;
                }
                for (int s = tid; s < ROWS; s += kThreads) {
                    const int j = nj0 + s;
                    sm.rowid[s] = (s < nlive && j < ra.rows_per_domain)
                                      ? (int)perm_apply((unsigned)(nd * ra.rows_per_domain + j), ra.perm_n, ra.perm_a, ra.perm_b, ra.step_keys[t_next]) : -1;
                    sm.lrew[s] = 0.f;
                }
                __syncthreads();
            }
        } else {
            const int n_run = expectation ? md.M : 1;
            float* result = nullptr;
            int member = 0;
            for (int mi = 0; mi < n_run; ++mi) {
                if (expectation) member = mi;
                else if (fast) member = __builtin_amdgcn_readfirstlane(sm.sched[t]);  // wave-uniform: weight pointers stay in SGPRs
                else member = member_dom;
                if (mi > 0) {  // expectation: layer 1 overwrote buf0, rebuild the same input for the next member
                    build_input(t, sm.buf0);
                    __syncthreads();
                }
                // ---- the MLP: ping-pong through LDS ------------------------------------------------
                float* cur = sm.buf0;
                float* nxt = sm.buf1;
                for (int l = 0; l < md.n_layers; ++l) {
                    prof.mark(12);
                    if constexpr (kPre) {
                        NextOp nop;
                        nop.valid = false;
                        if (l + 1 < md.n_layers) nop = describe_layer<R, S>(md, sm.lmeta, l + 1, member, wave);
                        else if (t + 1 < ra.t_end)  // the next step's first op (its member: the schedule's next entry / the same domain)
                            nop = describe_layer<R, S>(md, sm.lmeta, 0, fast ? __builtin_amdgcn_readfirstlane(sm.sched[t + 1]) : member_dom, wave);
                        mlp_layer_pre<R, S>(md, l + 1 == md.n_layers, cur_op, cur, nxt, wave, lane, prof, pre, nop);
                        cur_op = nop;
                        lds_barrier();
                    } else if constexpr (kB3) {
                        mlp_layer_b3<R, S>(md, sm.lmeta, l, member, cur, nxt, wave, lane);
                        __syncthreads();
                    } else {
                        mlp_layer<R, S>(md, sm.lmeta, l, member, cur, nxt, wave, lane, prof);
                        __syncthreads();
                    }
                    prof.mark(8);
                    float* tmp = cur; cur = nxt; nxt = tmp;
                }
                result = cur;

                if (expectation) {  // gaussian_mlp.py:213-215: mean over members of mean AND (clamped) logvar
                    for (int i = tid; i < ROWS * md.out_total; i += kThreads) {
                        const int s = i / md.out_total, c = i % md.out_total;
                        float v = result[s * ld_k + c];
                        if (!deterministic && c >= md.out_dim) {
                            const int d = c - md.out_dim;
                            const int bd = (lv_rows > 1 ? member * md.out_dim : 0) + d;
                            v = sm.maxlv[bd] - softplus_fast(sm.maxlv[bd] - v);
                            v = sm.minlv[bd] + softplus_fast(v - sm.minlv[bd]);
                        }
                        sm.expacc[i] = mi == 0 ? v : sm.expacc[i] + v;
                    }
                    __syncthreads();
                }
            }

            // ---- sample, delta, next obs (model.py:458-473, one_dim_tr_model.py:280-288) -----------
            // MODE 0: prediction = mean (deterministic model, or no eps given); 1: injected eps; 2: in-kernel Philox.
            // Wave-uniform switches are hoisted into compile-time variants so the four per-dimension chains
            // (LDS read -> 2 softplus -> exp -> sqrt -> fma) are straight-line code and interleave.
            HIPETS_STAMP(0, t);  // the MLP is done
            auto sample_impl = [&](auto expect_tag, auto mode_tag) __attribute__((always_inline)) {
                constexpr bool EXPECT = decltype(expect_tag)::value;
                constexpr int MODE = decltype(mode_tag)::value;
                const float inv_m = 1.0f / (float)md.M;
                const float* lvmin = sm.minlv + (lv_rows > 1 ? member * md.out_dim : 0);  // this step's member owns the bounds
                const float* lvmax = sm.maxlv + (lv_rows > 1 ? member * md.out_dim : 0);
                for (int item = tid; item < ROWS * nblk; item += kThreads) {
                    const int s = item / nblk, blk = item % nblk;
                    const int rid = sm.rowid[s];
                    HIPETS_BOUND(s < ROWS && rid < ra.B && 4 * blk < md.out_dim + 4 && md.out_total <= ld_k);
                    if (rid < 0) continue;
                    float nrm[4] = {0.f, 0.f, 0.f, 0.f};
                    if constexpr (MODE == 1) {
    #pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int d = min(blk * 4 + q, md.out_dim - 1);
                            nrm[q] = ra.eps[((size_t)t * ra.B + rid) * md.out_dim + d];
                        }
                    } else if constexpr (MODE == 2) {
                        rollout_normals4(rid, t, blk, ra.seed, ra.stream_id, nrm);
                    }
                    float pred[4], prev[4];
    #pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int d = min(blk * 4 + q, md.out_dim - 1);
                        float mean, lv = 0.f;
                        if constexpr (EXPECT) {
                            mean = sm.expacc[s * md.out_total + d] / (float)md.M;
                            if constexpr (MODE != 0) lv = sm.expacc[s * md.out_total + md.out_dim + d] / (float)md.M;
                        } else {
                            mean = result[s * ld_k + d];
                            if constexpr (MODE != 0) {
                                lv = result[s * ld_k + md.out_dim + d];
                                lv = lvmax[d] - softplus_fast(lvmax[d] - lv);  // gaussian_mlp.py:152
                                lv = lvmin[d] + softplus_fast(lv - lvmin[d]);  // :153
                            }
                        }
                        if constexpr (MODE != 0) pred[q] = mean + __builtin_amdgcn_sqrtf(exp_hw(lv)) * nrm[q];  // model.py:471-473
                        else pred[q] = mean;
                        const int do_ = min(d, md.obs_dim - 1);
                        prev[q] = (md.target_is_delta && !sm.nodelta[do_]) ? sm.state[s * md.obs_dim + do_] : 0.f;
                    }
                    (void)inv_m;
                    unsigned pub[4] = {0u, 0u, 0u, 0u};
    #pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int d = blk * 4 + q;
                        if (d < md.obs_dim) {
                            const float nobs = pred[q] + prev[q];  // one_dim_tr_model.py:281-286 (prev = 0 for no_delta dims)
                            sm.state[s * md.obs_dim + d] = nobs;
                            pub[q] = __float_as_uint(nobs);
                            if (trace_next_obs) trace_next_obs[((size_t)t * ra.B + rid) * md.obs_dim + d] = nobs;
                        } else if (d < md.out_dim) {
                            sm.lrew[s] = pred[q];  // learned reward = last output (one_dim_tr_model.py:287)
                        }
                    }
                    // persistent DEVICE form: the row's next owner waits for these values -- on their way before the reward phase
                    if (handover) {
                        const unsigned tg = (unsigned)(handover_tag >> 32);
                        if (blk * 4 < md.obs_dim) pair_store(handover + (size_t)rid * NV + blk * 4, pub[0], pub[1], tg);
                        if (blk * 4 + 2 < md.obs_dim) pair_store(handover + (size_t)rid * NV + blk * 4 + 2, pub[2], pub[3], tg);
                    }
                }
            };
            {
                using T = std::true_type;
                using F = std::false_type;
                // lean instances: stochastic model, in-kernel Philox draws (the host selects them only then)
                const int mode = kLean ? 2 : (deterministic ? 0 : (ra.eps != nullptr ? 1 : (ra.use_philox ? 2 : 0)));
                if (expectation) {
                    if (mode == 0) sample_impl(T{}, std::integral_constant<int, 0>{});
                    else if (mode == 1) sample_impl(T{}, std::integral_constant<int, 1>{});
                    else sample_impl(T{}, std::integral_constant<int, 2>{});
                } else {
                    if (mode == 0) sample_impl(F{}, std::integral_constant<int, 0>{});
                    else if (mode == 1) sample_impl(F{}, std::integral_constant<int, 1>{});
                    else sample_impl(F{}, std::integral_constant<int, 2>{});
                }
            }
            if (more && !persist) fetch_actions_commit(t + 1, av);
            __syncthreads();
            prof.mark(9);

            // ---- reward, termination, masked accumulation (model_env.py:124-129, :186-188) of step t, and, in the
            // same barrier interval, the model input of step t+1 (both only READ the new state) ------------------
            // Persistent DEVICE form: every row changes workgroups now.  Its new state left in the sampling phase; its running
            // total and flag follow from here, and the same thread then evaluates which row the slot holds in step t + 1.
            for (int s = tid; s < ROWS; s += kThreads) {
                const int rid = sm.rowid[s];
                if (rid >= 0) {
                    const float* st = sm.state + s * md.obs_dim;
                    const float* ac = sm.actn + (t & 1) * ROWS * md.act_dim + s * md.act_dim;
                    float tot = sm.tot[s];
                    int trm = sm.term[s];
                    if (persist && sm.pend[s]) {  // collected late: the previous owner published them after ITS reward phase (normally long arrived)
                        sm.pend[s] = 0;
                        const unsigned long long* const src = ra.exchange + (size_t)rid * NV + (NV - 2);
                        const unsigned want = ra.tag_base + (unsigned)t;
                        const long long t_poll = wall_clock64();
                        u32x4g g = {0u, 0u, 0u, 0u};
                        for (int spins = 0;; ++spins) {
                            pair_load_issue(g, src);
                            asm volatile("s_waitcnt vmcnt(0)" : "+v"(g)::"memory");
                            if (g[1] == want && g[3] == want) break;
                            if ((poll_every || (spins & 63) == 63) && (wall_clock64() - t_poll > ra.poll_ticks || *(volatile int*)ra.error_flag)) {
                                *ra.error_flag = 1;
                                break;
                            }
                            __builtin_amdgcn_s_sleep(8);
                        }
                        tot = __uint_as_float(g[0]);
                        trm = (int)g[2];
                    }
                    float r = reward_eval(st, ac, md.obs_dim, md.act_dim, reward_fn, sm.lrew[s]);
                    const bool done = term_eval(st, md.obs_dim, term_fn);
                    if (trace_rewards) trace_rewards[(size_t)t * ra.B + rid] = r;
                    if (trm) r = 0.f;
                    trm = trm | (done ? 1 : 0);
                    tot += r;
                    sm.term[s] = trm;
                    sm.tot[s] = tot;
                    if (handover) {
                        pair_store(handover + (size_t)rid * NV + (NV - 2), __float_as_uint(tot), (unsigned)trm, (unsigned)(handover_tag >> 32));
                    } else if (persist) {
                        ra.totals[rid] = tot;  // last step: the row's return
                    }
                }
                if (persist && has_next) {  // the slot's row in the next turn (only this thread reads rowid[s] between the two barriers around here)
                    const int j = (v_next % ra.groups) * ROWS + s;
                    sm.rowid[s] = j < ra.rows_per_domain
                                      ? (int)perm_apply((unsigned)((v_next / ra.groups) * ra.rows_per_domain + j), ra.perm_n, ra.perm_a, ra.perm_b, ra.step_keys[t_next]) : -1;
                    sm.lrew[s] = 0.f;
                }
            }
            if (more && !persist) build_input(t + 1, sm.buf0);
            __syncthreads();
            prof.mark(10);

        }
        HIPETS_STAMP(1, t);  // sampled, rewarded, published
        if (persist && has_next) {
            // ---- collect the rows of the next turn: 8-byte {value bits, step tag} granules, self-validating ----
            {
                int nj0, nlive;
                logical_rows(v_next, domain, nj0, nlive);
            }
            member_dom = domain;
            compute_act_base();
            float av2[kPrefetch];
            fetch_actions_issue(t_next, av2);  // in flight while the rows arrive
            const unsigned long long tag = (unsigned long long)(ra.tag_base + (unsigned)t_next) << 32;  // published in step t_next - 1
            // (Writing the normalised input columns straight from here, as the straight form does, measured SLOWER in this flow: cfg4'
            // 10.5 vs 9.85 ms per rollout, cfg4 3.44 vs 3.37 -- with 12-24 pairs per thread the f64 work serialises behind every poll
            // round, while the separate pass below spreads it over the workgroup.)
            if (t_next == ra.t_begin) {  // a later turn of the FIRST step: the rows start from s0 (nothing was handed over yet)
                for (int i = tid; i < ROWS * md.obs_dim; i += kThreads) {
                    const int s_ = i / md.obs_dim;
                    const int rid_ = sm.rowid[s_];
                    const size_t env_off = (!kLean && ra.pop_env > 0 && rid_ >= 0) ? (size_t)((rid_ / ra.P) / ra.pop_env) * md.obs_dim : 0;
                    sm.state[i] = rid_ >= 0 ? ra.s0[env_off + (i - s_ * md.obs_dim)] : 0.f;
                }
                for (int s_ = tid; s_ < ROWS; s_ += kThreads) {
                    sm.tot[s_] = 0.f;
                    sm.term[s_] = 0;
                }
            } else if constexpr (kDmaCollect) {
                dma_collect((unsigned)(tag >> 32));
            } else
            for (int base = 0; base < ROWS * NVP; base += kGT * kThreads) {
                const unsigned long long* src[kGT];
                u32x4g g[kGT];
                int gs[kGT], gv[kGT];
                bool soft[kGT];  // {running total, flag}: wanted at the NEXT reward phase only -- one look now, the rest there
                const unsigned want = (unsigned)(tag >> 32);
#pragma unroll
                for (int q = 0; q < kGT; ++q) {
                    const int i = base + tid + q * kThreads;
                    // (item -> (row slot, pair) by multiply-high with ceil(2^32 / NVP): exact for i < 2^32 / NVP, and i < 64 * NVP here)
                    const int i_row = (int)__umulhi((unsigned)i, nvp_magic);
                    gs[q] = (base == 0 && q < kG) ? xs[q < kG ? q : 0] : (i < ROWS * NVP ? i_row : -1);
                    gv[q] = (base == 0 && q < kG) ? xv[q < kG ? q : 0] : (i < ROWS * NVP ? i - i_row * NVP : 0);
                    soft[q] = gv[q] == NVP - 1;
                    src[q] = nullptr;
                    g[q] = u32x4g{0u, want, 0u, want};  // rows of the padding: zero state, total, flag
                    if (gs[q] >= 0) {
                        const int rid = sm.rowid[gs[q]];
                        if (rid >= 0) src[q] = ra.exchange + (size_t)rid * NV + 2 * gv[q];
                    }
                }
                const long long t_poll = wall_clock64();  // constant 100 MHz counter
                for (int spins = 0;; ++spins) {
                    // issue, issue, wait as straight-line asm (no branch between a load and its wait: the compiler does not know the
                    // destination registers are still in flight); items with nothing to fetch read the table's first pair and ignore it
                    static_assert(kGT == 2 || kGT == 4 || kGT == 8 || kGT == 16 || kGT == 24, "the wait below names its destinations");
                    u32x4g got[kGT];
#pragma unroll
                    for (int q = 0; q < kGT; ++q) pair_load_issue(got[q], src[q] ? src[q] : ra.exchange);
                    if constexpr (kGT == 2) asm volatile("s_waitcnt vmcnt(0)" : "+v"(got[0]), "+v"(got[1])::"memory");
                    else if constexpr (kGT == 4) asm volatile("s_waitcnt vmcnt(0)" : "+v"(got[0]), "+v"(got[1]), "+v"(got[2]), "+v"(got[3])::"memory");
                    else {  // (every 8 destinations one statement; the first is the wait, the others only tie their registers behind it)
#pragma unroll
                        for (int q8 = 0; q8 < kGT; q8 += 8)
                            asm volatile("s_waitcnt vmcnt(0)" : "+v"(got[q8]), "+v"(got[q8 + 1]), "+v"(got[q8 + 2]), "+v"(got[q8 + 3]), "+v"(got[q8 + 4]), "+v"(got[q8 + 5]), "+v"(got[q8 + 6]), "+v"(got[q8 + 7])::"memory");
                    }
                    bool ready = true;
#pragma unroll
                    for (int q = 0; q < kGT; ++q)
                        if (src[q]) {
                            g[q] = got[q];
                            if (got[q][1] == want && got[q][3] == want) src[q] = nullptr;
                            else if (!soft[q]) ready = false;
                        }
                    if (ready) break;
                    // a hand-over takes microseconds; 0.2 s (poll_ticks) without the producer means it is not running at all (the grid is
                    // not co-resident: another process or stream holds CUs) -- give up loudly instead of spinning on; a flag that is
                    // already up (another workgroup, or an earlier poll, gave up) ends the wait after 64 spins: the results of the
                    // launch are void anyway and the host re-runs the call (hipets_check_async_error)
                    if ((poll_every || (spins & 63) == 63) && (wall_clock64() - t_poll > ra.poll_ticks || *(volatile int*)ra.error_flag)) {
                        *ra.error_flag = 1;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(8);
                }
#pragma unroll
                for (int q = 0; q < kGT; ++q)
                    if (gs[q] >= 0) {
                        HIPETS_BOUND(gs[q] < ROWS && gv[q] >= 0 && gv[q] < NVP);
                        if (!soft[q]) {
                            const int d = 2 * gv[q];
                            const float v0 = __uint_as_float(g[q][0]), v1 = __uint_as_float(g[q][2]);
                            if constexpr (kFuse && S::TERM == HIPETS_TERM_HOPPER) {  // (rows of the padding hold zeros and no row id: never read)
                                if (sm.rowid[gs[q]] >= 0 && hopper_pair_bad(d, v0, v1, true, d + 1 < md.obs_dim)) hop_flags[gs[q]] = 1;
                            }
                            sm.state[gs[q] * md.obs_dim + d] = v0;
                            if (d + 1 < md.obs_dim) sm.state[gs[q] * md.obs_dim + d + 1] = v1;
                        } else if (src[q]) {
                            sm.pend[gs[q]] = 1;  // not there yet: the reward phase fetches the pair
                        } else {
                            sm.tot[gs[q]] = __uint_as_float(g[q][0]);
                            sm.term[gs[q]] = (int)g[q][2];
                        }
                    }
            }
            HIPETS_STAMP(2, t);  // this thread's rows have arrived
            fetch_actions_commit(t_next, av2);
            __syncthreads();
            build_input(t_next, step_in);
            __syncthreads();
            HIPETS_STAMP(3, t);  // the next step's input is built
        }
    }

    // ---- write back -------------------------------------------------------------------------------
    if constexpr (kFuse && S::TERM == HIPETS_TERM_HOPPER) {
        // the fused all-dims termination folds step t's per-row flag into `terminated` one step later (tail_unit): the LAST step's
        // flag is still pending here.  It cannot change a return (model_env.py:186-188: the terminating step's reward counts), but
        // sm.term is what the write-back publishes -- one launch per step in DEVICE mode: the next launch starts from it -- so it is
        // completed before anything reads it (complete since the step's barrier)
        if (ra.t_end > ra.t_begin && !persist)
            for (int s = tid; s < ROWS; s += kThreads) sm.term[s] |= sm.pend[((ra.t_end - 1) & 1) * ROWS + s];
    }
    for (int s = tid; s < ROWS; s += kThreads) {
        const int rid = sm.rowid[s];
        if (rid < 0 || persist) continue;  // persistent form: written in the last step's reward phase
        ra.totals[rid] = sm.tot[s];
        if ((!fast && !persist) || (!kLean && ra.write_back)) ra.term[rid] = (unsigned char)sm.term[s];
    }
    if (prof.on) {  // flush the phase accumulators of this wave
#pragma unroll
        for (int i = 0; i < 16; ++i) ra.phase_cycles[wave * 16 + i] += prof.slot[i];
    }
    if ((!fast && !persist) || (!kLean && ra.write_back)) {
        for (int i = tid; i < ROWS * md.obs_dim; i += kThreads) {
            const int s = i / md.obs_dim, d = i % md.obs_dim;
            const int rid = sm.rowid[s];
            if (rid >= 0) ra.state[(size_t)rid * md.obs_dim + d] = sm.state[i];
        }
    }
}

}  // namespace hipets
