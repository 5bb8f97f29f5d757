/*
 * hipets.h -- C ABI of libhipets.so, the MI355X (gfx950) PETS planning / rollout engine.
 *
 * Drop-in boundary for ONE hot path of facebookresearch/mbrl-lib (v0.2.0):
 *
 *   TrajectoryOptimizerAgent.act -> {CEM,iCEM,MPPI}Optimizer.optimize
 *       -> ModelEnv.evaluate_action_sequences -> GaussianMLP ensemble -> reward/termination
 *
 * The reference is pure Python/PyTorch, so "the reference's FFI for this path" is a ctypes
 * binding; every entry point below names the reference interface it replaces
 * (file:line relative to the reference root).  Plain pointers and sizes only, no torch types.
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on error; hipets_last_error() gives the
 *     thread-local message.  Nothing throws across the ABI.
 *   - pointers marked DEVICE point into caller-owned HBM (e.g. torch storage); pointers marked
 *     HOST are read during the call only.  The library retains no caller pointer past a call,
 *     except hipets_set_model which copies weights into its own packed layout.
 *   - `stream` is a hipStream_t (pass torch.cuda.current_stream().cuda_stream); all work is enqueued asynchronously on it.
 *     An engine's workspace is shared by its calls, so their work executes in call order: a call on another stream than the
 *     previous call's makes its stream wait (device side) for that call's work.  Two entry points synchronise `stream` before
 *     returning, because they read caller-owned DEVICE tensors that may be freed right after the call: hipets_set_model and
 *     hipets_planet_set_model.  hipets_timing_read waits for the events it reports; the one-time co-residency self-test of a
 *     persistent DEVICE-mode kernel instance synchronises once (hipets_set_persistent).  Nothing else synchronises.
 *   - HOST arrays are consumed before the call returns, so temporaries are fine: observations are copied into
 *     pinned staging buffers owned by the engine and uploaded asynchronously from there; model descriptors are
 *     uploaded inside hipets_set_model, which synchronises.
 *   - one engine per device; an engine is not thread-safe (the reference is single-threaded).  Engines of DIFFERENT devices may
 *     be driven from different host threads (the library's per-kernel residency table is locked).
 */
#ifndef HIPETS_H
#define HIPETS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HIPETS_ABI_VERSION 6
#define HIPETS_MAX_LAYERS 8

typedef struct hipets_engine hipets_engine;

/* activation module of GaussianMLP (mbrl/models/gaussian_mlp.py:88-96) */
enum { HIPETS_ACT_RELU = 0, HIPETS_ACT_SILU = 1, HIPETS_ACT_LEAKY_RELU = 2, HIPETS_ACT_TANH = 3, HIPETS_ACT_SIGMOID = 4 };
/* Ensemble.propagation_method (mbrl/models/gaussian_mlp.py:201-216) */
enum { HIPETS_PROP_RANDOM_MODEL = 0, HIPETS_PROP_FIXED_MODEL = 1, HIPETS_PROP_EXPECTATION = 2 };
/* obs_process_fn (mbrl/env/pets_halfcheetah.py:91-113, mbrl/env/pets_cartpole.py:78-101) */
enum { HIPETS_OBS_NONE = 0, HIPETS_OBS_HALFCHEETAH = 1, HIPETS_OBS_CARTPOLE_PETS = 2 };
/* reward_fn (mbrl/env/reward_fns.py:10-53); LEARNED = last model output (model_env.py:124-128) */
enum { HIPETS_REW_LEARNED = 0, HIPETS_REW_CARTPOLE = 1, HIPETS_REW_CARTPOLE_PETS = 2, HIPETS_REW_INVERTED_PENDULUM = 3,
       HIPETS_REW_HALFCHEETAH = 4, HIPETS_REW_PUSHER = 5,
       HIPETS_REW_NONE = 6 /* rewards are computed by the caller (arbitrary Python reward_fn on hipets_step's next_obs) */ };
/* termination_fn (mbrl/env/termination_fns.py:12-95) */
enum { HIPETS_TERM_NONE = 0, HIPETS_TERM_CARTPOLE = 1, HIPETS_TERM_INVERTED_PENDULUM = 2, HIPETS_TERM_HOPPER = 3,
       HIPETS_TERM_WALKER2D = 4, HIPETS_TERM_ANT = 5, HIPETS_TERM_HUMANOID = 6 };
/* Normalizer dtype (mbrl/util/math.py:108-111; normalize_double_precision, one_dim_tr_model.py:87-92) */
enum { HIPETS_NORM_NONE = 0, HIPETS_NORM_F32 = 1, HIPETS_NORM_F64 = 2 };
/* Ensemble container: GAUSSIAN_MLP = one GaussianMLP with E members: balanced random shuffles and the batch % members
 * check (mbrl/models/gaussian_mlp.py:179-216); BASIC_ENSEMBLE = mbrl.models.BasicEnsemble of E single-member
 * GaussianMLPs (conf/dynamics_model/basic_ensemble.yaml): every row draws its member independently
 * (basic_ensemble.py:122-129, 255-260), any batch size, no elites (:262-266), per-member logvar bounds. */
enum { HIPETS_ENSEMBLE_GAUSSIAN_MLP = 0, HIPETS_ENSEMBLE_BASIC = 1 };
/* arithmetic of the ensemble MLP's linear layers */
enum { HIPETS_PREC_F32 = 0,    /* v_mfma_f32_16x16x4_f32: fp32 operands, fp32 accumulate (the graded mode)              */
       HIPETS_PREC_BF16X3 = 1  /* fp32 operands carried as three bf16 pieces, six exact partial products per product     */
                               /* on v_mfma_f32_16x16x32_bf16, fp32 accumulate: fp32-accurate to a few product ulps      */
                               /* (|error| <= 2^-22 |a||b| per product), reported SEPARATELY from the fp32-MFMA mode     */ };
/* randomness source of a rollout */
enum { HIPETS_MODE_EXACT = 0,  /* reference semantics, injected perms / eps (parity mode)                              */
       HIPETS_MODE_FAST = 1,   /* whole-horizon kernel, block-balanced member schedule, in-kernel Philox             */
       HIPETS_MODE_DEVICE = 2  /* reference semantics (ONE balanced permutation of all B rows per step,              */
                               /* gaussian_mlp.py:203-205; iid eps per row and dim, model.py:471-473) with both      */
                               /* drawn in-kernel from (seed, stream_id): a keyed bijection of [0, B) and Philox     */
                               /* normals.  No input tensors; exportable through hipets_device_perms /              */
                               /* hipets_fast_normals for replay through a reference implementation.                 */ };

/*
 * Snapshot of what ModelEnv.evaluate_action_sequences reads from the live objects
 * (OneDTransitionRewardModel mbrl/models/one_dim_tr_model.py:29-116, GaussianMLP
 * mbrl/models/gaussian_mlp.py:69-127, EnsembleLinearLayer mbrl/models/util.py:31-65).
 */
typedef struct {
    int32_t obs_dim;         /* raw observation width                                             */
    int32_t act_dim;         /* action width                                                      */
    int32_t in_dim;          /* model input width = width(obs_process_fn(obs)) + act_dim          */
    int32_t out_dim;         /* model output width = obs_dim + learned_rewards                    */
    int32_t hid;             /* hidden width                                                      */
    int32_t n_layers;        /* number of linear layers = hidden layers + 1 (<= HIPETS_MAX_LAYERS) */
    int32_t ensemble_size;   /* E: leading dim of every weight tensor                             */
    int32_t n_members;       /* M: number of ACTIVE members (elite set, gaussian_mlp.py:161-163)  */
    const int32_t* members;  /* HOST [M] indices into E                                           */
    int32_t activation;      /* HIPETS_ACT_*                                                      */
    float leaky_slope;       /* negative slope for LEAKY_RELU                                     */
    int32_t propagation;     /* HIPETS_PROP_*                                                     */
    int32_t deterministic;   /* model has no logvar head (gaussian_mlp.py:113-114)                */
    int32_t obs_process;     /* HIPETS_OBS_*                                                      */
    int32_t reward_fn;       /* HIPETS_REW_*                                                      */
    int32_t termination_fn;  /* HIPETS_TERM_*                                                     */
    int32_t target_is_delta; /* one_dim_tr_model.py:281-286                                       */
    int32_t learned_rewards;
    int32_t n_no_delta;
    const int32_t* no_delta; /* HOST [n_no_delta] observation dims predicted absolutely           */
    int32_t normalizer;      /* HIPETS_NORM_*                                                     */
    const double* norm_mean; /* HOST [in_dim] (f32 stats widened exactly; arithmetic stays f32)   */
    const double* norm_std;  /* HOST [in_dim]                                                     */
    const float* min_logvar; /* HOST [out_dim] ([M,out_dim] for BASIC_ENSEMBLE: every member owns */
    const float* max_logvar; /*   its bounds) or NULL if deterministic                            */
    const void* const* weights; /* HOST array [n_layers] of DEVICE float [E, in_l, out_l]         */
    const void* const* biases;  /* HOST array [n_layers] of DEVICE float [E, 1, out_l]            */
    int32_t ensemble_kind;   /* HIPETS_ENSEMBLE_*                                                 */
    int32_t precision;       /* HIPETS_PREC_*: BF16X3 runs only where a shape-specialised kernel instance   */
                             /*   exists for the model and the call (else the rollout call fails)           */
} hipets_model_desc;

/* options of one evaluate_action_sequences call */
typedef struct {
    int32_t mode;            /* HIPETS_MODE_*                                                     */
    /* EXACT mode: the reference's random draws, injected                */
    const int64_t* perms;    /* DEVICE: random_model [H,B] (one torch.randperm(B) per step,       */
                             /*   gaussian_mlp.py:205); fixed_model [B] (:375); else NULL         */
    const float* eps;        /* DEVICE [H,B,out_dim] standard normals consumed by torch.normal    */
                             /*   (model.py:471-473); NULL => predictions are the mean            */
    /* FAST and DEVICE modes: counter-based RNG                                                   */
    uint64_t seed;
    uint64_t stream_id;      /* e.g. plan counter * iterations + iteration                        */
    const int32_t* member_schedule; /* DEVICE [H, n_workgroups] optional override of the FAST-mode */
                             /*   member draws (testing; TS-infinity ModelEnv.step); see member_schedule_len */
    const float* fast_eps;   /* DEVICE [H,B,out_dim] optional override of the Philox normals      */
    /* optional debug taps (DEVICE, may be NULL)                                                  */
    float* trace_next_obs;   /* [H,B,obs_dim]                                                     */
    float* trace_rewards;    /* [H,B] reward of the step before termination masking               */
    int32_t rows_per_group;  /* 0 = auto; else force R (row tiles of 16 per workgroup)            */
    int64_t* phase_cycles;   /* DEVICE [8,16] optional: per-wave, per-phase shader-cycle counters of     */
                             /*   workgroup 0, accumulated (profiling aid): cycles in the  */
                             /*   low 44 bits of an entry, the number of marks that fed it above them    */
    int32_t no_sample;       /* FAST: predictions are the mean (no eps), like ModelEnv.step(sample=False) */
    int32_t rows_per_member; /* EXACT, > 0: explicit per-row member maps.  perms is [H, M*rows_per_member]             */
                             /*   ([M*rows_per_member] for fixed_model): slot m*rows_per_member + j = j-th row of        */
                             /*   member m, -1 = padding (members may own unequal row counts).  Required for            */
                             /*   BASIC_ENSEMBLE (randint maps, basic_ensemble.py:122-129); for GAUSSIAN_MLP it          */
                             /*   expresses mbrl.util.math.propagate_from_indices (util/math.py:180-196): any            */
                             /*   row -> member assignment, no batch % members rule                                      */
    int32_t n_env;           /* FAST / DEVICE batched planning: the pop candidates are n_env groups of */
                             /*   pop / n_env, group g starts from s0[g] (s0 is then HOST [n_env, obs_dim]); 0/1 = one.     */
                             /*   DEVICE: ONE balanced permutation per step over the rows of all environments               */
    int32_t generic_kernel;  /* 1 = only the fully generic kernel instance.  (The library also instantiates the rollout     */
                             /*   kernel (a) for hidden widths of 193..208 -- the reference's default 200 -- with the hidden  */
                             /*   layers' shape as a compile-time fact and everything else generic, and (b) for the BASELINE */
                             /*   shapes with all layer shapes / reward / termination fns as compile-time facts; same        */
                             /*   arithmetic -- tests compare them bit for bit.)  2 = (a) allowed, (b) not                   */
    /* ABI v6 */
    int32_t member_schedule_len; /* entries of member_schedule (horizon x n_workgroups of hipets_fast_geometry); 0 = not stated.  */
                             /*   A stated length that does not match the call's geometry fails the call instead of reading   */
                             /*   the schedule with another stride (the geometry of FAST calls changed in ABI v5 -> v6)       */
    uint64_t perm_stream_id; /* hipets_step, DEVICE mode, fixed_model propagation: stream of the TS-infinity permutation (the     */
                             /*   stream of the rollout's reset: model.py:404-407) while stream_id -- the step's -- keys the eps; */
                             /*   0 = stream_id                                                                                   */
} hipets_rollout_opts;

/* ---- lifecycle -------------------------------------------------------------------------- */
int hipets_abi_version(void);
const char* hipets_last_error(void);
/* Class of this thread's last failure, for callers that must tell "the arguments were refused" (deterministic: every rank of
 * a sharded job that passed the same arguments got the same answer) from "the machine failed" (may have hit one rank only):  */
#define HIPETS_ERR_NONE 0
#define HIPETS_ERR_INVALID_ARGUMENT 1 /* a value, shape or configuration the library rejects                                */
#define HIPETS_ERR_RUNTIME 2          /* a HIP or RCCL call, an allocation or a kernel launch failed                         */
#define HIPETS_ERR_TIMEOUT 3          /* a persistent DEVICE-mode rollout gave up waiting for another workgroup's rows       */
int hipets_last_error_kind(void);
int hipets_create(int device, hipets_engine** out);
void hipets_destroy(hipets_engine* e);

/* Re-snapshot weights / normaliser / elite set (call after ModelTrainer.train,
 * mbrl/models/model_trainer.py:288-296).  Packs into the MFMA fragment layout. */
int hipets_set_model(hipets_engine* e, const hipets_model_desc* desc, void* stream);

/* ---- ModelEnv.evaluate_action_sequences (mbrl/models/model_env.py:145-191) ---------------- */
/* actions DEVICE [pop,H,A] f32; s0 HOST [obs_dim] f32; returns DEVICE [pop] f32.              */
int hipets_rollout(hipets_engine* e, const float* actions, const float* s0, int32_t pop, int32_t horizon,
                   int32_t num_particles, const hipets_rollout_opts* opts, float* returns, void* stream);
/* ---- ModelEnv.step (mbrl/models/model_env.py:87-140; MBPO-style one-step model rollouts) ---
 * One transition for B independent rows: obs DEVICE [B,obs_dim], actions DEVICE [B,act_dim] ->
 * next_obs DEVICE [B,obs_dim], rewards DEVICE [B] f32, dones DEVICE [B] uint8.  opts as for hipets_rollout with
 * horizon 1 and one particle per row: EXACT takes perms [B] (one torch.randperm, or the fixed_model indices) and eps
 * [1,B,out_dim] (NULL => the deterministic mean, i.e. ModelEnv.step(sample=False)); DEVICE (the reference's per-row balanced
 * shuffle, gaussian_mlp.py:203-205) and FAST (one member per workgroup of 16 R consecutive rows) draw both in-kernel from
 * (seed, stream_id); set opts->no_sample for the deterministic mean.  DEVICE + fixed_model: see opts->perm_stream_id.          */
int hipets_step(hipets_engine* e, const float* obs, const float* actions, int32_t batch, const hipets_rollout_opts* opts,
                float* next_obs, float* rewards, uint8_t* dones, void* stream);

/* geometry the FAST kernel will use for (pop, P): workgroups and rows per group (for member_schedule).  rows_per_group 0: the
 * library's own choice for a DEFAULT call (hipets_rollout without injected eps / traces, the fused plans); > 0: forced.
 * rows_per_group -1: the choice for calls that run the general kernel layout whatever the model -- hipets_step, rollouts with
 * injected eps or traces.  The two differ only for models with a wide-output shape-specialised instance (Humanoid-v4's 376 obs
 * dims: two row tiles per workgroup there, one in the general layout); a call of the second kind that brings its own
 * member_schedule on such a model must size it with -1 and pass the reported row-tile count as opts->rows_per_group.        */
int hipets_fast_geometry(hipets_engine* e, int32_t pop, int32_t num_particles, int32_t horizon, int32_t rows_per_group,
                         int32_t* n_workgroups, int32_t* row_tiles);

/* Which instance of the rollout kernel a DEFAULT call (hipets_rollout / the fused plans with in-kernel randomness, no injected
 * eps, no traces) of this size runs on the engine's model, and with how many row tiles per workgroup.  mode: HIPETS_MODE_FAST or
 * HIPETS_MODE_DEVICE.  rows_per_group: 0 = the library's own choice (what a default call does); 1..4 = the answer for a call that
 * forces this row-tile count through opts->rows_per_group (ABI v6).  The instances compute the same arithmetic (the GPU suite compares them bit for bit); they differ in how
 * much of the model's shape is a compile-time fact:
 *   GENERIC        everything decided at run time (any widths, activations, propagation, normaliser)
 *   HIDDEN_STATIC  SiLU models whose hidden layers are 193..208 wide -- the reference's default 200
 *                  (conf/dynamics_model/gaussian_mlp_ensemble.yaml:8) --, 113..128 or 241..256 wide, whatever their reward /
 *                  termination / preprocessing / output width
 *   FUSED          hidden AND output layer shapes, reward / termination closed form (or learned reward) and obs preprocessing are
 *                  compile-time facts; the output layer's accumulators feed the step's tail from registers (launch.hpp's tables:
 *                  the BASELINE.json configurations and the conf/overrides/pets_*.yaml workloads without a termination function
 *                  that reads every state dim)
 *   WIDE           FUSED for output layers wider than 8 column tiles (Humanoid-v4)
 * Diagnostic only -- nothing needs to call it; a profile (rocprofv3 --kernel-trace) shows the same thing as a kernel name.      */
enum { HIPETS_KERNEL_GENERIC = 0, HIPETS_KERNEL_HIDDEN_STATIC = 1, HIPETS_KERNEL_FUSED = 2, HIPETS_KERNEL_WIDE = 3 };
int hipets_kernel_class(hipets_engine* e, int32_t pop, int32_t num_particles, int32_t horizon, int32_t mode, int32_t rows_per_group,
                        int32_t* kernel_class, int32_t* row_tiles);

/* FAST-mode randomness, exported so a FAST rollout can be replayed through a reference implementation:
 * schedule DEVICE int32 [H, n_workgroups] = member slot of workgroup w at step t (the B = pop * P rows form one run,
 * particle-major -- run index g = p * pop + c for particle p of candidate c, i.e. row c * P + p of the batch -- and
 * workgroup w owns run indices [w * 16 * row_tiles, (w + 1) * 16 * row_tiles): n_workgroups = ceil(ceil(B / 16) /
 * row_tiles)).  Since ABI v6 the slot is position p_t(w) * M / n_workgroups of the step's keyed bijection p_t of the
 * workgroup indices (the Feistel network DEVICE mode applies to rows): every workgroup evaluates its own entry, there is
 * no schedule kernel or buffer behind a default call, and this export returns exactly those integers.
 * NOTE (small populations): a workgroup's 16 * row_tiles rows are CONSECUTIVE run indices, so when pop < 16 * row_tiles -- or
 * where a workgroup straddles the end of one particle's run -- several particles of the same candidate sit in one workgroup
 * and share a member at every step; the reference's per-row assignment (gaussian_mlp.py:267-275) gives them independent,
 * balanced members.  DEVICE mode (the default of the Python layer) has the reference's semantics at every size.
 * normals DEVICE f32 [H, B, out_dim] = the eps the kernel draws for (step, row, dim) with the same (seed, stream_id).     */
int hipets_fast_schedule(hipets_engine* e, int32_t horizon, int32_t n_workgroups, uint64_t seed, uint64_t stream_id,
                         int32_t* schedule, void* stream);
int hipets_fast_normals(hipets_engine* e, int32_t horizon, int32_t batch, uint64_t seed, uint64_t stream_id,
                        float* normals, void* stream);

/* DEVICE-mode randomness, exported for replay: perms DEVICE int64 [H, B] (random_model; [1, B] for fixed_model) = the
 * permutation a DEVICE rollout of `batch` = pop * particles rows with the same (seed, stream_id) uses at every step, in
 * the reference's convention (slot j holds row perms[t][j]; slot j runs on active member j / (B / M)); the eps of that
 * rollout are hipets_fast_normals(seed, stream_id).                                                              */
int hipets_device_perms(hipets_engine* e, int32_t horizon, int32_t batch, uint64_t seed, uint64_t stream_id, int64_t* perms,
                        void* stream);

/* DEVICE-mode rollouts with a fresh permutation per step (random_model) run as ONE persistent launch: only as many workgroups
 * as are resident at once are launched, rows change workgroups every step through a table of 16-byte {value, tag, value, tag}
 * granule pairs in HBM (write-through stores, polled loads; no grid barrier), and a batch with more logical workgroups than that
 * is served in turns by the launched ones (except where two workgroups fit a CU and the batch still exceeds the chip: those
 * launch once per step).  What makes that safe:
 *   - residency is VERIFIED, not assumed: the first time a kernel instance is to run persistently at a larger grid than before,
 *     the library launches that very instance in a self-test mode in which every workgroup waits for all the others (one extra
 *     launch and ONE synchronisation of `stream`, once per instance, LDS size (model / horizon) and grid size -- so the first plan
 *     of a new shape is not capturable into a hipGraph, later ones are); if they cannot meet, the runtime's smaller
 *     occupancy answer is tried, and failing that the engine launches per step;
 *   - every poll is bounded (hipets_set_handover_timeout, default 0.2 s): if a producer never shows up (another process or
 *     stream took CUs after the self-test) the kernel raises a host-visible flag and drains in milliseconds.  The results of
 *     that launch, and everything computed from them, are INVALID;
 *   - hipets_check_async_error tells: call it once the results have reached the host (after the device-to-host copy of a plan,
 *     or any synchronisation of `stream`), before acting on them.  On *timed_out = 1 the engine has already fallen back to
 *     per-step launches: re-run the call (hipets.planning does exactly this).  A caller that never asks gets the report as an
 *     error from its NEXT rollout / plan call instead.
 * The persistent form assumes what the reference's deployment gives it -- one planning process per GPU; on = 0 forces per-step
 * launches (also: env HIPETS_NO_PERSISTENT=1).  Env HIPETS_MAX_WORKGROUPS=n caps the workgroups of a persistent launch at n (the
 * rest of the batch is served in turns, as when the chip is the limit): processes that share a GPU can leave each other room.
 * (The wide-output instances -- Humanoid-v4: 752 output columns, two row tiles per workgroup -- deal the LAST turn of a step in one-tile
 * workgroups when the row tiles left for it fit one per launched workgroup; same results bit for bit, a shorter step.  Env
 * HIPETS_RAGGED_LAST_TURN=0 keeps two-tile turns throughout: A/B measurements.)                                               */
int hipets_set_persistent(hipets_engine* e, int32_t on);
int hipets_set_handover_timeout(hipets_engine* e, double seconds);
int hipets_check_async_error(hipets_engine* e, int32_t* timed_out);

/* ---- CEMOptimizer pieces (mbrl/planning/trajectory_opt.py:100-188) ------------------------- */
typedef struct {
    int32_t population_size;
    int32_t horizon;
    int32_t act_dim;
    int32_t num_iterations;
    int32_t elite_num;          /* ceil(pop * elite_ratio), trajectory_opt.py:89-91             */
    double alpha;               /* momentum; (1 - alpha) is formed in double like the Python reference */
    int32_t return_mean_elites;
    int32_t clipped_normal;
    int32_t unbiased_var;       /* 1 = CEM (:137), 0 = iCEM (:479)                               */
} hipets_cem_params;

/* _sample_population (:110-128).  z DEVICE [pop,H,A] optional injected N(0,1) draws (already
 * truncated for the truncated-normal branch); NULL => Philox(seed, stream_id).  lower/upper/mu/
 * dispersion DEVICE [H,A]; population DEVICE [pop,H,A] out.                                    */
int hipets_cem_sample(hipets_engine* e, const hipets_cem_params* p, const float* mu, const float* dispersion,
                      const float* lower, const float* upper, const float* z, uint64_t seed, uint64_t stream_id,
                      float* population, void* stream);
/* NaN->-1e-10 (:178), topk (:179), elite mean/var refit with momentum (:130-140), best-so-far
 * (:184-186).  values DEVICE [pop] (modified in place like the reference); mu/dispersion in-out;
 * best_value DEVICE [1] in-out (init -inf); best_solution DEVICE [H,A] in-out;
 * elite_idx DEVICE [elite_num] int32 out (optional).                                           */
int hipets_cem_refit(hipets_engine* e, const hipets_cem_params* p, float* values, const float* population,
                     float* mu, float* dispersion, float* best_value, float* best_solution, int32_t* elite_idx,
                     void* stream);

/* The same refit with the elites CHOSEN BY THE CALLER: elites DEVICE [elite_num] int32 candidate indices, best first (values must
 * hold no index twice; out-of-range indices are the caller's error).  For seed-identical replays of the reference: the order
 * torch.topk (:179) leaves among EQUAL values is an artefact of its partial sort, and 0 / 1 reward functions (cartpole,
 * inverted pendulum: env/reward_fns.py:10-13, 27-30) tie dozens of candidates at the elite boundary -- the host runs the
 * reference's own topk on the returned values and hands the indices in.  NaN -> -1e-10 still happens in place.           */
int hipets_cem_refit_elites(hipets_engine* e, const hipets_cem_params* p, float* values, const float* population,
                            const int32_t* elites, float* mu, float* dispersion, float* best_value, float* best_solution,
                            void* stream);

/* population[elite_idx] -> dst (the persistent ICEMOptimizer.elite, trajectory_opt.py:476): rows of src
 * [n_src, D] selected by index DEVICE int32 [rows].                                                          */
int hipets_gather_rows(hipets_engine* e, int32_t rows, int32_t dim, const float* src, const int32_t* index, float* dst,
                       void* stream);

/* ---- MPPIOptimizer pieces (mbrl/planning/trajectory_opt.py:238-311) -------------------------------------- */
/* noise + beta-smoothing recurrence + clipping (:262-295).  mean DEVICE [H,A] (already shifted, :257-258),
 * past_action DEVICE [A], z DEVICE [pop,H,A] optional injected truncated normals, population DEVICE out.     */
int hipets_mppi_sample(hipets_engine* e, int32_t pop, int32_t horizon, int32_t act_dim, double beta, const float* mean,
                       const float* past_action, const float* lower, const float* upper, const float* z, uint64_t seed,
                       uint64_t stream_id, float* population, void* stream);
/* NaN -> -1e-10, exp(gamma (v - max v)) weights, weighted mean (:296-309).  values DEVICE [pop] in place.     */
int hipets_mppi_update(hipets_engine* e, int32_t pop, int32_t horizon, int32_t act_dim, double gamma, float* values,
                       const float* population, float* mean, void* stream);

/* ---- ICEMOptimizer pieces (mbrl/planning/trajectory_opt.py:391-487, mbrl/util/math.py:318-396) ------------ */
/* coloured-noise population (:433-441): n rows of population DEVICE [>= n, H, A].  normals DEVICE
 * [2, n, A, H/2+1] optional injected unit normals (real, imaginary parts of the spectrum).                   */
int hipets_icem_sample(hipets_engine* e, int32_t n, int32_t horizon, int32_t act_dim, double exponent, const float* mu,
                       const float* var, const float* lower, const float* upper, const float* normals, uint64_t seed,
                       uint64_t stream_id, float* population, void* stream);
/* kept elites shifted one step with a fresh tail action (:450-462).  kept DEVICE [keep,H,A]; end_noise DEVICE
 * [keep, A] optional injected N(0,1); out DEVICE [keep,H,A].                                                   */
int hipets_icem_shift(hipets_engine* e, int32_t keep, int32_t horizon, int32_t act_dim, const float* kept, const float* mu,
                      const float* var, const float* end_noise, uint64_t seed, uint64_t stream_id, float* out, void* stream);

/* ---- fused plans: a whole optimizer.optimize() with the engine's rollout as objective -------- */
/* Randomness mode of the rollouts inside the hipets_plan_* calls: HIPETS_MODE_FAST (default) or HIPETS_MODE_DEVICE
 * (reference propagation semantics).  Iteration i of plan `plan_id` uses stream_id = plan_id * num_iterations + i for
 * both its population noise and its rollout (iCEM: 4 * that, + 0..3 for noise / shifted tail / kept-elite draw / rollout). */
int hipets_set_plan_mode(hipets_engine* e, int32_t mode);

/* Optional per-iteration record of the following fused plans (NULL pointers are skipped, a NULL trace switches recording
 * off).  All DEVICE, written asynchronously on the plan's stream; for a batched plan n_env > 1, else n_env = 1:
 *   populations [iters, max_rows, H, A]  the candidates iteration i evaluated, environment after environment (max_rows
 *                                        >= n_env * the largest per-environment count; rows beyond an iteration's untouched)
 *   values      [iters, max_rows]        their returns after the NaN filter
 *   mus, dispersions [iters, n_env, H, A]  optimizer state after iteration i (MPPI: mus = the refined mean)
 *   elite_idx   [iters, n_env, elite_num]  int32, best first, indices into the environment's own candidates (CEM / iCEM)
 * This is what lets a test replay a fused plan through a reference implementation draw by draw.                      */
typedef struct {
    float* populations;
    float* values;
    float* mus;
    float* dispersions;
    int32_t* elite_idx;
    int32_t max_rows;
} hipets_plan_trace;
int hipets_set_plan_trace(hipets_engine* e, const hipets_plan_trace* trace);

/* Whole CEMOptimizer.optimize with the engine's rollout as objective, no host round trip
 * (replaces trajectory_opt.py:142-188 + the closure at :743-748).  x0/lower/upper DEVICE [H,A];
 * out DEVICE [H,A] = mu if return_mean_elites else best.  Rollouts in the engine's plan mode;
 * workspace is engine-owned.                                                                  */
int hipets_plan_cem(hipets_engine* e, const hipets_cem_params* p, const float* x0, const float* lower,
                    const float* upper, const float* s0, int32_t num_particles, uint64_t seed, uint64_t plan_id,
                    float* out, void* stream);
/* The same for n_env independent environments in ONE set of launches (vectorised envs / many agents): x0 and out
 * DEVICE [n_env,H,A], s0 HOST [n_env,obs_dim], bounds shared.  p->population_size is PER environment.  Fills all 256 CUs
 * where a single plan cannot (cfg2 has 2.4 row tiles per CU).                                                    */
int hipets_plan_cem_batched(hipets_engine* e, const hipets_cem_params* p, int32_t n_env, const float* x0, const float* lower,
                            const float* upper, const float* s0, int32_t num_particles, uint64_t seed, uint64_t plan_id,
                            float* out, void* stream);

/* Whole MPPIOptimizer.optimize (trajectory_opt.py:238-311) with the engine's FAST rollout as objective.
 * mean DEVICE [H,A] in-out = the optimizer's persistent self.mean: shifted one step in place (:257-258), then
 * refined num_iterations times (:262-309); the refined mean is what optimize() returns (:311).              */
int hipets_plan_mppi(hipets_engine* e, int32_t population_size, int32_t horizon, int32_t act_dim, int32_t num_iterations,
                     double gamma, double beta, float* mean, const float* lower, const float* upper, const float* s0,
                     int32_t num_particles, uint64_t seed, uint64_t plan_id, void* stream);

/* The same for n_env environments in ONE set of launches (FAST-mode rollouts): mean DEVICE [n_env,H,A] in-out, s0 HOST
 * [n_env,obs_dim], bounds shared, population_size PER environment.                                               */
int hipets_plan_mppi_batched(hipets_engine* e, int32_t population_size, int32_t horizon, int32_t act_dim, int32_t num_iterations,
                             double gamma, double beta, int32_t n_env, float* mean, const float* lower, const float* upper,
                             const float* s0, int32_t num_particles, uint64_t seed, uint64_t plan_id, void* stream);

typedef struct {
    int32_t population_size;
    int32_t horizon;
    int32_t act_dim;
    int32_t num_iterations;
    int32_t elite_num;              /* ceil(pop * elite_ratio), trajectory_opt.py:368-370                      */
    int32_t keep_elite_size;        /* ceil(keep_elite_frac * elite_num), rounded up to the module (:378-383)  */
    int32_t population_size_module; /* 0 = none (:419-431)                                                     */
    int32_t return_mean_elites;
    double alpha;
    double population_decay_factor;
    double colored_noise_exponent;
} hipets_icem_params;

/* Whole ICEMOptimizer.optimize (trajectory_opt.py:391-487) with the engine's FAST rollout as objective.
 * elite DEVICE [elite_num,H,A] in-out = the optimizer's persistent self.elite (read only when has_elite != 0,
 * always written: it holds the last iteration's elites on return).  keep_idx DEVICE int32 [num_iterations,
 * keep_elite_size] optional injected randperm(elite_num)[:keep] per iteration (:446-448); NULL => Philox.
 * out DEVICE [H,A] = mu if return_mean_elites else best.                                                      */
int hipets_plan_icem(hipets_engine* e, const hipets_icem_params* p, const float* x0, const float* lower, const float* upper,
                     float* elite, int32_t has_elite, const int32_t* keep_idx, const float* s0, int32_t num_particles,
                     uint64_t seed, uint64_t plan_id, float* out, void* stream);

/* The same for n_env environments in ONE set of launches (FAST-mode rollouts): x0 / out DEVICE [n_env,H,A], elite DEVICE
 * [n_env,elite_num,H,A] in-out, keep_idx DEVICE int32 [num_iterations, n_env, keep_elite_size] or NULL, s0 HOST
 * [n_env,obs_dim]; bounds and the per-iteration population sizes are shared by the environments.                   */
int hipets_plan_icem_batched(hipets_engine* e, const hipets_icem_params* p, int32_t n_env, const float* x0, const float* lower,
                             const float* upper, float* elite, int32_t has_elite, const int32_t* keep_idx, const float* s0,
                             int32_t num_particles, uint64_t seed, uint64_t plan_id, float* out, void* stream);

/* ---- multi-GPU: population sharding with ONE all-gather of returns per iteration --------------------- */
/* The library talks to RCCL itself (librccl is loaded on first use, no link-time dependency): rank 0 makes an id, the
 * host broadcasts its HIPETS_COMM_ID_BYTES bytes by any means (MPI, torch.distributed, a file), every rank joins.        */
#define HIPETS_COMM_ID_BYTES 128
int hipets_comm_unique_id(void* id_out);
int hipets_comm_init(hipets_engine* e, const void* unique_id, int32_t rank, int32_t world_size);
int hipets_comm_destroy(hipets_engine* e);
/* rank and size as the communicator itself reports them (ncclCommUserRank / ncclCommCount); 0 / 1 without a communicator.
 * Env HIPETS_RCCL_LIB=<path> makes the library load that shared object instead of librccl (tests/fake_rccl builds a stand-in
 * that runs N ranks as N processes on ONE GPU, so the world > 1 path of hipets_plan_cem_sharded is testable on a 1-GPU box). */
int hipets_comm_info(hipets_engine* e, int32_t* rank, int32_t* world_size);
/* hipets_plan_cem over all ranks of the communicator as one device-side loop per rank: every rank passes IDENTICAL
 * arguments; the population is sampled identically everywhere (counter-based RNG), rank r rolls out candidates
 * [r * pop / world ...) (first pop % world ranks hold one more), one ncclAllGather of the per-candidate returns per
 * iteration over xGMI, then the same refit on the same data everywhere (bit-identical mu / dispersion, no broadcast).
 * With world_size == 1 it equals hipets_plan_cem.  Shard sizes that the rollout mode cannot take (DEVICE mode: rows % members)
 * are refused on every rank before the first collective; a rank on which an iteration cannot be enqueued keeps taking part in
 * the collectives and returns its error at the end (no peer is left waiting) -- agree on the outcome across ranks before using
 * a plan (hipets.dist.plan_cem_sharded does).  hipets_set_plan_trace records the gathered values like hipets_plan_cem's.   */
int hipets_plan_cem_sharded(hipets_engine* e, const hipets_cem_params* p, const float* x0, const float* lower, const float* upper,
                            const float* s0, int32_t num_particles, uint64_t seed, uint64_t plan_id, float* out, void* stream);

/* The same scheme for the other two optimizers ("MPPI: same all-gather of values (weights need global max and
 * sum)"; "iCEM: kept elites are replicated state, so same scheme"): arguments as hipets_plan_mppi / hipets_plan_icem, identical
 * on every rank; sampling (MPPI's smoothed noise; iCEM's coloured noise, kept / shifted elites, the +1 mu row) is replicated,
 * every iteration's population -- iCEM's shrinks from iteration to iteration, the shards with it -- is rolled out shard-wise and
 * its returns all-gathered, the importance-weighted mean (:297-311) / the elite refit (:474-485) then runs on identical data
 * everywhere: bit-identical persistent state (`mean`, `elite`) on all ranks.  Same refusal / failure rules as above.       */
int hipets_plan_mppi_sharded(hipets_engine* e, int32_t population_size, int32_t horizon, int32_t act_dim, int32_t num_iterations,
                             double gamma, double beta, float* mean, const float* lower, const float* upper, const float* s0,
                             int32_t num_particles, uint64_t seed, uint64_t plan_id, void* stream);
int hipets_plan_icem_sharded(hipets_engine* e, const hipets_icem_params* p, const float* x0, const float* lower, const float* upper,
                             float* elite, int32_t has_elite, const int32_t* keep_idx, const float* s0, int32_t num_particles,
                             uint64_t seed, uint64_t plan_id, float* out, void* stream);

/* ---- PlaNet latent planner (mbrl/models/planet.py) ----------------------------------- */
/* The tensors PlaNetModel.sample reads (planet.py:531-581), DEVICE f32 in nn.Linear layout: weights [out, in]
 * row-major, biases [out].  The library packs private copies; call again after every model update.            */
typedef struct {
    int32_t latent_size;     /* latent_state_size                                                               */
    int32_t action_size;
    int32_t belief_size;
    int32_t hidden_size;     /* hidden_size_fcs                                                                 */
    float min_std;           /* MeanStdCat (planet.py:104-115)                                                  */
    const void* w_embed;     /* belief_model.embedding_layer[0]  [belief, latent + action]   (planet.py:86-88)  */
    const void* b_embed;
    const void* w_ih;        /* belief_model.rnn (GRUCell) weight_ih [3 belief, belief], gates r | z | n  (:89) */
    const void* b_ih;
    const void* w_hh;        /* weight_hh [3 belief, belief]                                                    */
    const void* b_hh;
    const void* w_prior1;    /* prior_transition_model[0] [hidden, belief]                        (:229-234)   */
    const void* b_prior1;
    const void* w_prior2;    /* prior_transition_model[2] [2 latent, hidden]                                    */
    const void* b_prior2;
    const void* w_rew1;      /* reward_model[0] [hidden, belief + latent]                          (:260-266)   */
    const void* b_rew1;
    const void* w_rew2;      /* reward_model[2] [hidden, hidden]                                                */
    const void* b_rew2;
    const void* w_rew3;      /* reward_model[4] [1, hidden]                                                     */
    const void* b_rew3;
} hipets_planet_desc;
int hipets_planet_set_model(hipets_engine* e, const hipets_planet_desc* d, void* stream);

typedef struct {
    const float* eps;        /* DEVICE [H,B,latent] injected N(0,1) draws of planet.py:299-305; NULL => Philox   */
    uint64_t seed;
    uint64_t stream_id;
    int32_t no_sample;       /* latent = prior mean (sample(deterministic=True))                                 */
    float* trace_latent;     /* optional DEVICE taps: [H,B,latent], [H,B,belief], [H,B]                          */
    float* trace_belief;
    float* trace_rewards;
    int64_t* phase_cycles;   /* ABI v6: DEVICE [8,16] phase accumulators of workgroup 0, as hipets_rollout_opts.phase_cycles;    */
                             /*   filled by profiling builds (-DHIPETS_LEAN_PROF=1) only, ignored by the shipped library         */
} hipets_planet_opts;
/* ModelEnv.evaluate_action_sequences on a PlaNetModel with no_termination and the learned reward head
 * (model_env.py:145-191, mbrl/algorithms/planet.py): actions DEVICE [pop,H,A]; latent0 / belief0 DEVICE [latent] /
 * [belief] = the model's saved posterior sample and belief (planet.py:669-672), tiled over the pop * P rows;
 * returns DEVICE [pop] particle-averaged.  One kernel launch for the whole horizon.                              */
int hipets_planet_rollout(hipets_engine* e, const float* actions, const float* latent0, const float* belief0, int32_t pop,
                          int32_t horizon, int32_t num_particles, const hipets_planet_opts* opts, float* returns, void* stream);

/* The PlaNet planner as one call: CEMOptimizer.optimize (trajectory_opt.py:142-188; the PlaNet configs use the clipped-normal
 * branch :116-120, conf/overrides/planet_cheetah_run.yaml:29-35) with hipets_planet_rollout as objective, no host round trip.
 * Arguments as hipets_plan_cem, with the latent start state of hipets_planet_rollout instead of an observation.            */
int hipets_plan_planet_cem(hipets_engine* e, const hipets_cem_params* p, const float* x0, const float* lower, const float* upper,
                           const float* latent0, const float* belief0, int32_t num_particles, uint64_t seed, uint64_t plan_id,
                           float* out, void* stream);

/* ---- instrumentation (bench.py roofline leg) ----------------------------------------------- */
/* on = 1: every rollout-kernel launch carries a start / stop hipEvent pair on its dispatch packet; on = k > 1: every
 * k-th launch does (a completion signal per packet costs a few microseconds between back-to-back short launches -- the
 * per-step launches of DEVICE mode -- so bench.py samples them); on = 0: off.                                      */
int hipets_timing_enable(hipets_engine* e, int32_t on);
/* Synchronises the recorded events; returns number of launches and their summed duration.       */
int hipets_timing_read(hipets_engine* e, int64_t* launches, double* total_ms, int32_t reset);

#ifdef __cplusplus
}
#endif
#endif /* HIPETS_H */
