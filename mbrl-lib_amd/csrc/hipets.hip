// hipets.hip -- C-ABI implementation (see include/hipets.h).  gfx950 only; no CPU fallback.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/hipets.h"
#include "cem.hpp"
#include "launch.hpp"
#include "optim.hpp"
#include "rollout_helpers.hpp"

using namespace hipets;

namespace {

thread_local std::string g_err;
thread_local int g_err_kind = HIPETS_ERR_NONE;  // class of the last failure of this thread (hipets_last_error_kind)

// an argument / configuration the library rejects: deterministic, the same on every rank that passes the same arguments
int fail(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    g_err_kind = HIPETS_ERR_INVALID_ARGUMENT;
    return 1;
}

// something the machine did (a HIP / RCCL call, an allocation, a launch, a hand-over time-out): may hit one rank only
int fail_kind(const int kind, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    g_err_kind = kind;
    return 1;
}

#define HCHECK(expr)                                                                              \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess) return fail_kind(HIPETS_ERR_RUNTIME, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return 0;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        hipError_t e = hipMalloc(&p, bytes);
        if (e != hipSuccess) return fail_kind(HIPETS_ERR_RUNTIME, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        cap = bytes;
        return 0;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    template <typename T>
    T* as() const { return reinterpret_cast<T*>(p); }
};

constexpr int kMaxR = 4;

}  // namespace

struct hipets_engine {
    int device = 0;
    int num_cu = 256;
    size_t lds_max = 160 * 1024;
    bool has_model = false;
    ModelDev md{};
    int ensemble_size = 0;
    DevBuf w3pack;  // bf16x3 precision mode: weight pieces
    DevBuf wpack, bpack, layer_meta, norm_mean, norm_std, min_lv, max_lv, no_delta, members;
    // rollout workspace
    DevBuf s0, state, totals, term;
    // DEVICE mode, persistent form: row exchange table, per-step permutation keys, timeout flag (host-mapped)
    DevBuf exchange, step_keys, plan_keys;
    // host -> device staging of the caller's observations: a small ring of pinned buffers owned by the engine, so the async
    // copy never reads caller memory after the call returned (hipets.h: HOST arrays are consumed during the call)
    struct HostStage { void* p = nullptr; size_t cap = 0; hipEvent_t done = nullptr; bool used = false; };
    HostStage stage[4];
    int stage_next = 0;
    uint32_t tag_base = 0;  // hand-over tags handed out so far (exchange granules hold tags <= tag_base)
    // the key tables a fused plan generated up front: rollouts (plan_keys_seed, stream in [first, first + count), H) read them
    uint64_t plan_keys_seed = 0, plan_keys_first = 0;
    int plan_keys_count = 0, plan_keys_H = 0;
    int* error_flag = nullptr;
    bool persistent_ok = true;
    long long poll_ticks = 20000000ll;  // bound of one hand-over poll, 100 MHz ticks (hipets_set_handover_timeout; default 0.2 s)
    DevBuf census;                      // [2] ints of the co-residency self-test (rollout_inst.inc launch_one)
    // The workspace (state / totals / schedules / plan buffers), the hand-over table, its tags and the key tables are
    // engine-global: the work of two calls must execute in the order the calls were made.  A call on another stream than the
    // previous call's first makes its stream wait for that one (an event, device side only), so "any stream per call"
    // (hipets.h) stays true without two launches ever sharing a buffer.
    hipStream_t last_stream = nullptr;
    bool last_stream_set = false;
    hipEvent_t last_done = nullptr;
    // plan workspace
    DevBuf mu, disp, population, values, best_value, best_solution, past_action, kept, elite_idx, keep_idx;
    // RCCL communicator (lazy-loaded librccl)
    void* comm = nullptr;
    int comm_rank = 0, comm_world = 1;
    DevBuf shard_values, gathered;
    // PlaNet latent model
    bool has_planet = false;
    bool planet_static = false;  // the PlaNet model has conf/dynamics_model/planet.yaml's shapes: the STATIC kernel instance (planet_types.hpp)
    PlanetDev pd{};
    DevBuf planet_w, planet_b, planet_member, planet_ops;
    // fused plans: randomness mode of their rollouts, optional per-iteration trace
    int plan_mode = HIPETS_MODE_FAST;
    bool has_trace = false;
    hipets_plan_trace trace{};
    // timing: every timing_stride-th rollout-kernel launch carries a start / stop event pair on its dispatch packet
    bool timing = false;
    int timing_stride = 1;
    unsigned long long launch_counter = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> event_pool;
};

namespace {

int launch_rollout(hipets_engine* e, int R, int grid, size_t lds, const RolloutArgs& ra, hipStream_t st) {
    hipEvent_t a = nullptr, b = nullptr;
    const bool timed = !ra.capacity_out && e->timing && (e->launch_counter++ % (unsigned long long)e->timing_stride) == 0;
    if (timed) {
        if (!e->event_pool.empty()) {
            a = e->event_pool.back().first;
            b = e->event_pool.back().second;
            e->event_pool.pop_back();
        } else {
            HCHECK(hipEventCreate(&a));
            HCHECK(hipEventCreate(&b));
        }
    }
    hipError_t err;
    RolloutArgs rl = ra;
    rl.lds_bytes = (unsigned)lds;  // (debug builds check every LDS section against it)
    switch (R) {
        case 1: err = launch_rollout_r1(grid, (unsigned)lds, (int)e->lds_max, e->md, rl, st, a, b); break;
        case 2: err = launch_rollout_r2(grid, (unsigned)lds, (int)e->lds_max, e->md, rl, st, a, b); break;
        case 3: err = launch_rollout_r3(grid, (unsigned)lds, (int)e->lds_max, e->md, rl, st, a, b); break;
        case 4: err = launch_rollout_r4(grid, (unsigned)lds, (int)e->lds_max, e->md, rl, st, a, b); break;
        default: return fail("unsupported rows_per_group %d (1..%d)", R, kMaxR);
    }
    if (timed) e->events.emplace_back(a, b);  // recorded (or leaked to the pool) either way
    if (err == hipErrorNotSupported && e->md.precision == HIPETS_PREC_BF16X3)
        return fail("precision bf16x3: no shape-specialised kernel instance for this model / call (SiLU, f64 normaliser, no obs "
                    "preprocessing, in-kernel sampling, one of the BASELINE layer shapes, R = %d); use precision f32", R);
    if (err != hipSuccess) return fail_kind(HIPETS_ERR_RUNTIME, "rollout kernel launch failed: %s", hipGetErrorString(err));
    return 0;
}

// wide: the call will run a KSpec::WIDE instance (launch.hpp wide_model + lean_call): hidden-width activation buffers, the input
// image with its own stride in buf0
size_t lds_for(const hipets_engine* e, int R, int horizon, bool wide = false) {
    const ModelDev& md = e->md;
    if (wide)
        return rollout_smem_bytes(kTile * R, lean_ld(md.hidC, md.hidC), md.obs_dim, md.act_dim, md.in_dim, md.out_dim, md.out_total, horizon, false,
                                  md.lv_rows, md.ld_in);
    return rollout_smem_bytes(kTile * R, md.ld, md.obs_dim, md.act_dim, md.in_dim, md.out_dim, md.out_total, horizon,
                              md.propagation == HIPETS_PROP_EXPECTATION, md.lv_rows);
}

// Cost model for the row-tile count R of a workgroup (DESIGN.md "Choosing R"), in units of "one MFMA unit through a layer's k loop"
// (~2.7 us per step at hid 200).  A workgroup's step costs a + units(R): `units` = MFMA units per k-chunk of its busiest wave
// (hid 200: 4, 7, 10, 13 for R = 1..4), a ~ 1.8 = what a step spends outside the k loops (epilogues, tail, set-up, barriers).
// A CU holds two workgroups of an R <= 2 instance at once (256 registers each) and their fixed parts hide behind each other's MFMAs:
// a pair costs a + 2 units; R >= 3 instances own the CU (512 registers) and their workgroups run one after the other.  A (shape, R)
// pair without a shape-specialised instance (launch.hpp lean_shape_exists) runs the hidden-static or the generic kernel: + 8 %.
// `desync`: the workgroups of the launch do not wait for each other (FAST mode: one launch for the horizon, no hand-over).  Two of them
// on a CU then drift apart and their heavy waves stop meeting on a SIMD: a pair costs a + 2 x the AVERAGE units per SIMD (C R / 4: 3.25
// instead of 4 per row tile at hid 200).  Step-synchronous launches (DEVICE / EXACT: hand-over or one launch per step) pay the busiest one.
// Calibrated on MI355X (profiles/r4_stock_workloads.json, r4_learned_reward_workloads.json, r4_device_r_sweep.json: every R forced, 12
// workloads, both modes -- the rule picks the fastest R in 23 of the 24 cases and loses 0.3 % in the other).

// FAST-mode geometry: the B = pop x P rows are one run (rollout.hpp, prologue) of ceil(B / 16) row tiles, R per workgroup
inline long long fast_tiles(long long pop, int P) { return (pop * P + kTile - 1) / kTile; }

int wave_units(int C, int R) {  // MFMA units per k-chunk of the busiest SIMD (waves w and w + 4 share SIMD w % 4)
    const int full = C / kWaves, rem = C % kWaves, nu = rem * R;
    int simd[4] = {0, 0, 0, 0};
    for (int w = 0; w < kWaves; ++w) simd[w % 4] += full * R + (w < nu ? (nu - w + kWaves - 1) / kWaves : 0);
    return std::max(std::max(simd[0], simd[1]), std::max(simd[2], simd[3]));
}

int choose_R(const hipets_engine* e, long long tiles_total_per_slice, int slices, int forced, int horizon, bool wide, bool desync) {
    if (forced > 0) return forced;
    const int C = e->md.hidC;
    // the fixed part scales with the layer width like the units do.  WIDE instances (Humanoid-v4: 47 output column tiles, one workgroup per
    // CU) carry their output layer and its tail in it: a round of two-tile workgroups costs 1.29 x a round of one-tile ones in FAST mode,
    // 1.45 x in the turn-based DEVICE form (profiles/r5_cfg4p_iterations.json: the five population sizes of the cfg4' iCEM plan, both R)
    const double a = (wide ? (desync ? 6.45 : 2.67) : 1.77) * (double)C / 13.0;
    int best = 1;
    double best_cost = 1e300;
    // bf16x3 arithmetic exists in shape-specialised instances only: among the R that have one (if any has: else the launch reports it)
    bool b3_only = false;
    if (e->md.precision == HIPETS_PREC_BF16X3)
        for (int R = 1; R <= kMaxR; ++R) b3_only = b3_only || b3_shape_exists(e->md, R);
    for (int R = 1; R <= kMaxR; ++R) {
        if (lds_for(e, R, horizon, wide) > e->lds_max) break;
        if (wide && R > 2) break;  // WIDE instances exist for R = 1, 2 (rollout_inst.inc)
        if (b3_only && !b3_shape_exists(e->md, R)) continue;
        const long long groups = (tiles_total_per_slice + R - 1) / R;
        const long long nwg = groups * slices;
        const long long n = (nwg + e->num_cu - 1) / e->num_cu;  // workgroups the busiest CU serves
        const int co = (R <= 2 && !wide) ? 2 : 1;               // ... of which it holds this many at once
        const double u = wave_units(C, R);
        const double u_pair = (co == 2 && desync) ? (double)C * R / 4.0 : u;
        const long long full = n / co, rem = n % co;
        double cost = (double)full * (a + co * u_pair) + (rem ? a + (double)rem * u : 0.0);
        if (!lean_shape_exists(e->md, R, desync)) cost *= 1.08;
        if (cost < best_cost - 1e-9) {
            best_cost = cost;
            best = R;
        }
    }
    return best;
}

// MPPI sampling (optim.hpp): whole candidates staged in LDS where one fits, else one thread per series
int launch_mppi_sample(int n_env, int pop, int H, int A, float beta, const float* mean, const float* past_action, const float* lower, const float* upper,
                       const float* z, uint64_t seed, uint64_t stream_id, float* population, hipStream_t st) {
    const long long npop = (long long)n_env * pop;
    const long long D = (long long)H * A;
    if (D <= kMppiSampleMaxD) {
        const int G = mppi_sample_group(npop, (int)D);
        hipLaunchKernelGGL(mppi_sample_staged_kernel, dim3((unsigned)((npop + G - 1) / G)), dim3(kMppiSampleThreads), (size_t)G * D * 4, st, n_env, pop, H, A, G, beta,
                           mean, past_action, lower, upper, z, (unsigned long long)seed, (unsigned long long)stream_id, population);
    } else {
        const long long n = npop * A;
        hipLaunchKernelGGL(mppi_sample_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n_env, pop, H, A, beta, mean, past_action, lower, upper, z,
                           (unsigned long long)seed, (unsigned long long)stream_id, population);
    }
    HCHECK(hipGetLastError());
    return 0;
}

// MPPI update (optim.hpp): the weighted sum stages the population through LDS tiles as large as the CU holds -- the kernel opts in to the
// full LDS once per device, like the rollout kernels
int launch_mppi_update(hipets_engine* e, int n_env, int pop, int D, float gamma, float* values, const float* population, float* mean, hipStream_t st) {
    static std::atomic<bool> attr_set[64] = {};
    const int dev = e->device;
    if (dev < 0 || dev >= 64 || !attr_set[dev].load(std::memory_order_acquire)) {
        HCHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&mppi_update_kernel<kMppiTileMax / 16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)e->lds_max));
        HCHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&mppi_update_kernel<kMppiTileMax / 32>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)e->lds_max));
        if (dev >= 0 && dev < 64) attr_set[dev].store(true, std::memory_order_release);
    }
    const int tile_c = mppi_update_tile(pop, e->lds_max);
    const dim3 grid(mppi_update_blocks(D), n_env);
    if (tile_c == kMppiTileMax)
        hipLaunchKernelGGL(mppi_update_kernel<kMppiTileMax / 16>, grid, dim3(kMppiThreads), mppi_update_smem(pop, tile_c), st, pop, D, gamma, values, population, mean);
    else
        hipLaunchKernelGGL(mppi_update_kernel<kMppiTileMax / 32>, grid, dim3(kMppiThreads), mppi_update_smem(pop, tile_c), st, pop, D, gamma, values, population, mean);
    HCHECK(hipGetLastError());
    return 0;
}

// copy `bytes` of caller HOST memory to `dst` on `st`: memcpy into the next pinned slot of the engine's ring, async copy from
// there.  A slot is reused only after the copy that last read it has executed (its event; normally long complete).
int stage_h2d(hipets_engine* e, void* dst, const void* src, size_t bytes, hipStream_t st) {
    hipets_engine::HostStage& sl = e->stage[e->stage_next];
    e->stage_next = (e->stage_next + 1) % 4;
    if (sl.used) HCHECK(hipEventSynchronize(sl.done));
    if (bytes > sl.cap) {
        if (sl.p) (void)hipHostFree(sl.p);
        sl.p = nullptr;
        sl.cap = 0;
        HCHECK(hipHostMalloc(&sl.p, bytes < 4096 ? 4096 : bytes, hipHostMallocDefault));
        sl.cap = bytes < 4096 ? 4096 : bytes;
    }
    if (!sl.done) HCHECK(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
    std::memcpy(sl.p, src, bytes);
    HCHECK(hipMemcpyAsync(dst, sl.p, bytes, hipMemcpyHostToDevice, st));
    HCHECK(hipEventRecord(sl.done, st));
    sl.used = true;
    return 0;
}

// Every entry point that enqueues work on the engine's workspace calls this first: if `st` is not the stream of the previous
// call, `st` waits (device side) for what that call enqueued.  Same stream: nothing to do.  The event it waits for was recorded at
// the END of the previous call, on that call's own stream, while the caller was still inside the library -- i.e. while the stream
// was certainly alive (StreamScope below); nothing is ever recorded on a stream the caller may have destroyed since.
int enter_stream(hipets_engine* e, hipStream_t st) {
    if (!e->last_done) HCHECK(hipEventCreateWithFlags(&e->last_done, hipEventDisableTiming));
    if (e->last_stream_set && e->last_stream != st) HCHECK(hipStreamWaitEvent(st, e->last_done, 0));
    e->last_stream = st;
    e->last_stream_set = true;
    return 0;
}
// ... and holds one of these until it returns: marks the end of the call's work on its stream
struct StreamScope {
    hipets_engine* e;
    hipStream_t st;
    ~StreamScope() {
        static const bool off = std::getenv("HIPETS_NO_STREAM_SCOPE") != nullptr;  // (A/B measurements only)
        if (e && e->last_done && !off) (void)hipEventRecord(e->last_done, st);
    }
};
#define ENTER_STREAM(e, st)                \
    if (enter_stream((e), (st))) return 1; \
    StreamScope stream_scope_ { (e), (st) }

// Plan-level prologue shared by the fused plans: stage the observation(s) once (the same for every iteration) and, for DEVICE-mode
// plans, generate the per-step permutation keys of ALL `iters` rollouts (stream ids first_stream, +1, ...) in one launch.  (FAST-mode
// rollouts need nothing up front since round 6: every workgroup draws its own member schedule in its prologue, common.hpp fast_member.)
int plan_prologue(hipets_engine* e, const float* s0, int n_env, int H, int iters, uint64_t seed, uint64_t first_stream, hipStream_t st) {
    const ModelDev& md = e->md;
    if (e->s0.ensure((size_t)n_env * md.obs_dim * 4)) return 1;
    if (stage_h2d(e, e->s0.p, s0, (size_t)n_env * md.obs_dim * 4, st)) return 1;
    if (md.propagation == HIPETS_PROP_RANDOM_MODEL && iters >= 1 && e->plan_mode == HIPETS_MODE_DEVICE && e->persistent_ok) {
        // the per-step permutation keys of every rollout of the plan in one launch (rollout_impl finds them by seed / stream id)
        if (e->plan_keys.ensure((size_t)iters * H * sizeof(PermKeys))) return 1;
        hipLaunchKernelGGL(step_keys_kernel, dim3((H + 63) / 64, iters), dim3(64), 0, st, e->plan_keys.as<PermKeys>(), H, (unsigned long long)seed,
                           (unsigned long long)first_stream);
        HCHECK(hipGetLastError());
        e->plan_keys_seed = seed; e->plan_keys_first = first_stream; e->plan_keys_count = iters; e->plan_keys_H = H;
    }
    return 0;
}

// ---- RCCL through dlopen: no link-time dependency, and inside a PyTorch process the already loaded librccl is reused ----
struct RcclId { char internal[HIPETS_COMM_ID_BYTES]; };
struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, RcclId /* ncclUniqueId by value */, int) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    int (*CommCount)(void*, int*) = nullptr;     // optional (hipets_comm_info)
    int (*CommUserRank)(void*, int*) = nullptr;  // optional
};
Rccl g_rccl;

int rccl_load() {
    if (g_rccl.lib) return 0;
    void* lib = nullptr;
    // HIPETS_RCCL_LIB=<path>: load THIS library instead (a site build of RCCL; tests/fake_rccl: a stand-in that implements the
    // five entry points for N processes sharing one GPU, so that the world > 1 path runs on a one-GPU box)
    const char* forced = std::getenv("HIPETS_RCCL_LIB");
    if (forced && forced[0]) {
        lib = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
        if (!lib) return fail_kind(HIPETS_ERR_RUNTIME, "cannot load the RCCL library named by HIPETS_RCCL_LIB (%s): %s", forced, dlerror());
    }
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so"}) {
        if (lib) break;
        lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    }
    if (!lib) return fail_kind(HIPETS_ERR_RUNTIME, "cannot load librccl: %s", dlerror());
    g_rccl.GetUniqueId = reinterpret_cast<int (*)(void*)>(dlsym(lib, "ncclGetUniqueId"));
    g_rccl.CommInitRank = reinterpret_cast<int (*)(void**, int, RcclId, int)>(dlsym(lib, "ncclCommInitRank"));
    g_rccl.CommDestroy = reinterpret_cast<int (*)(void*)>(dlsym(lib, "ncclCommDestroy"));
    g_rccl.AllGather = reinterpret_cast<int (*)(const void*, void*, size_t, int, void*, hipStream_t)>(dlsym(lib, "ncclAllGather"));
    g_rccl.GetErrorString = reinterpret_cast<const char* (*)(int)>(dlsym(lib, "ncclGetErrorString"));
    g_rccl.CommCount = reinterpret_cast<int (*)(void*, int*)>(dlsym(lib, "ncclCommCount"));
    g_rccl.CommUserRank = reinterpret_cast<int (*)(void*, int*)>(dlsym(lib, "ncclCommUserRank"));
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.CommDestroy || !g_rccl.AllGather) return fail_kind(HIPETS_ERR_RUNTIME, "librccl lacks an expected symbol");
    g_rccl.lib = lib;
    return 0;
}
#define NCHECK(x)                                                                                        \
    do {                                                                                                 \
        const int r_ = (x);                                                                              \
        if (r_ != 0) return fail_kind(HIPETS_ERR_RUNTIME, "RCCL error %d (%s) at %s:%d", r_, g_rccl.GetErrorString ? g_rccl.GetErrorString(r_) : "?", __FILE__, __LINE__); \
    } while (0)

// candidates of rank r when pop candidates are dealt to `world` ranks (first pop % world ranks hold one more)
inline void shard_bounds(int pop, int world, int r, int* lo, int* hi) {
    const int base = pop / world, extra = pop % world;
    *lo = r * base + std::min(r, extra);
    *hi = *lo + base + (r < extra ? 1 : 0);
}

// gathered [world, width] (rank r's shard in row r, padded) -> values [pop]
__global__ void unpad_shards_kernel(const float* gathered, float* values, int pop, int world, int width) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= pop) return;
    const int base = pop / world, extra = pop % world;
    const int split = extra * (base + 1);  // candidates held by the ranks with one more
    const int r = i < split ? i / (base + 1) : extra + (i - split) / base;
    const int lo = r * base + min(r, extra);
    values[i] = gathered[(size_t)r * width + (i - lo)];
}

// hipets_set_plan_trace: record iteration i of a fused plan (population as evaluated, values after the NaN filter, refitted
// mean / dispersion) into the caller's buffers.  A no-op unless a trace is set.
int trace_iter(hipets_engine* e, int i, int rows, size_t nd, const float* population, const float* values, const float* mu,
               const float* disp, hipStream_t st, int n_env = 1) {
    if (!e->has_trace) return 0;
    const hipets_plan_trace& t = e->trace;
    if (rows > t.max_rows) return fail("plan trace: iteration %d evaluates %d candidates, trace buffers hold %d", i, rows, t.max_rows);
    const size_t ne = (size_t)n_env;
    if (t.populations && population)
        HCHECK(hipMemcpyAsync(t.populations + (size_t)i * t.max_rows * nd, population, (size_t)rows * nd * 4, hipMemcpyDeviceToDevice, st));
    if (t.values && values) HCHECK(hipMemcpyAsync(t.values + (size_t)i * t.max_rows, values, (size_t)rows * 4, hipMemcpyDeviceToDevice, st));
    if (t.mus && mu) HCHECK(hipMemcpyAsync(t.mus + (size_t)i * ne * nd, mu, ne * nd * 4, hipMemcpyDeviceToDevice, st));
    if (t.dispersions && disp) HCHECK(hipMemcpyAsync(t.dispersions + (size_t)i * ne * nd, disp, ne * nd * 4, hipMemcpyDeviceToDevice, st));
    return 0;
}

CemDev make_cem(const hipets_cem_params* p, int n_env = 1) {
    CemDev c{};
    c.n_env = n_env;
    c.pop = p->population_size;
    c.H = p->horizon;
    c.A = p->act_dim;
    c.D = p->horizon * p->act_dim;
    c.K = p->elite_num;
    c.alpha = (float)p->alpha;
    c.one_minus_alpha = (float)(1.0 - (double)p->alpha);
    c.return_mean = p->return_mean_elites;
    c.clipped = p->clipped_normal;
    c.unbiased = p->unbiased_var;
    return c;
}

int check_cem(const hipets_cem_params* p) {
    if (!p) return fail("null cem params");
    if (p->population_size < 1 || p->population_size > kMaxPop)
        return fail("population_size %d outside [1, %d]", p->population_size, kMaxPop);
    if (p->elite_num < 1 || p->elite_num > p->population_size) return fail("elite_num %d invalid", p->elite_num);
    if (p->unbiased_var && !p->clipped_normal && p->elite_num < 2) {
        // torch.var of one sample is NaN in the reference too; allowed, just flagged by NaN results
    }
    if (p->horizon < 1 || p->act_dim < 1) return fail("bad horizon/act_dim");
    return 0;
}

}  // namespace

extern "C" {

int hipets_abi_version(void) { return HIPETS_ABI_VERSION; }

const char* hipets_last_error(void) { return g_err.c_str(); }

int hipets_last_error_kind(void) { return g_err_kind; }

int hipets_create(int device, hipets_engine** out) {
    if (!out) return fail("null out pointer");
    *out = nullptr;
    int n = 0;
    hipError_t err = hipGetDeviceCount(&n);
    if (err != hipSuccess || n <= 0)
        return fail("no HIP device visible (%s) -- libhipets has no CPU fallback", hipGetErrorString(err));
    if (device < 0 || device >= n) return fail("device %d out of range (%d visible)", device, n);
    HCHECK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HCHECK(hipGetDeviceProperties(&prop, device));
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
        return fail("device %d is %s; libhipets is built for gfx950 (MI355X) only", device, prop.gcnArchName);
    auto* e = new hipets_engine();
    e->device = device;
    e->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    e->lds_max = prop.sharedMemPerBlockOptin > 0 ? (size_t)prop.sharedMemPerBlockOptin : (size_t)prop.sharedMemPerBlock;
    if (e->lds_max > 160 * 1024) e->lds_max = 160 * 1024;
    // timeout flag of the persistent DEVICE-mode kernel: host memory the device can write, read by the host without a sync
    if (hipHostMalloc(reinterpret_cast<void**>(&e->error_flag), sizeof(int), hipHostMallocMapped) != hipSuccess) e->error_flag = nullptr;
    if (e->error_flag) *e->error_flag = 0;
    const char* np = std::getenv("HIPETS_NO_PERSISTENT");
    e->persistent_ok = e->error_flag != nullptr && !(np && np[0] == '1');
    if (e->census.ensure(2 * sizeof(int))) e->persistent_ok = false;
    *out = e;
    return 0;
}

void hipets_destroy(hipets_engine* e) {
    if (!e) return;
    (void)hipSetDevice(e->device);
    if (e->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(e->comm);
    for (DevBuf* b : {&e->w3pack, &e->wpack, &e->bpack, &e->layer_meta, &e->norm_mean, &e->norm_std, &e->min_lv, &e->max_lv, &e->no_delta, &e->members,
                      &e->s0, &e->state, &e->totals, &e->term, &e->exchange, &e->step_keys, &e->plan_keys, &e->mu, &e->disp, &e->population, &e->values,
                      &e->best_value, &e->best_solution, &e->past_action, &e->kept, &e->elite_idx, &e->keep_idx, &e->planet_w, &e->planet_b, &e->planet_member, &e->planet_ops, &e->shard_values, &e->gathered, &e->census})
        b->release();
    for (auto& ev : e->events) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
    for (auto& ev : e->event_pool) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
    if (e->error_flag) (void)hipHostFree(e->error_flag);
    if (e->last_done) (void)hipEventDestroy(e->last_done);
    for (auto& sl : e->stage) {
        if (sl.p) (void)hipHostFree(sl.p);
        if (sl.done) (void)hipEventDestroy(sl.done);
    }
    delete e;
}

int hipets_set_model(hipets_engine* e, const hipets_model_desc* d, void* stream) {
    if (!e || !d) return fail("null argument");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    HCHECK(hipSetDevice(e->device));
    ENTER_STREAM(e, st);
    if (d->n_layers < 2 || d->n_layers > HIPETS_MAX_LAYERS) return fail("n_layers %d outside [2, %d]", d->n_layers, HIPETS_MAX_LAYERS);
    if (d->n_members < 1 || d->n_members > d->ensemble_size) return fail("n_members %d invalid for ensemble_size %d", d->n_members, d->ensemble_size);
    if (d->obs_dim < 1 || d->act_dim < 1 || d->in_dim < d->act_dim + 1 || d->hid < 1) return fail("bad dimensions");
    if (d->out_dim != d->obs_dim + (d->learned_rewards ? 1 : 0)) return fail("out_dim %d != obs_dim %d + learned_rewards %d", d->out_dim, d->obs_dim, d->learned_rewards);
    const int obs_in = d->in_dim - d->act_dim;
    const int expect_in = d->obs_process == HIPETS_OBS_CARTPOLE_PETS ? d->obs_dim + 1 : d->obs_dim;
    if (obs_in != expect_in) return fail("in_dim %d inconsistent with obs_dim %d / obs_process %d / act_dim %d", d->in_dim, d->obs_dim, d->obs_process, d->act_dim);
    if (d->reward_fn == HIPETS_REW_LEARNED && !d->learned_rewards) return fail("reward_fn LEARNED needs learned_rewards");
    if (d->reward_fn == HIPETS_REW_HALFCHEETAH && d->obs_dim < 3) return fail("halfcheetah reward needs obs_dim >= 3");
    if (d->reward_fn == HIPETS_REW_PUSHER && d->obs_dim < 20) return fail("pusher reward needs obs_dim >= 20");
    if ((d->reward_fn == HIPETS_REW_CARTPOLE || d->termination_fn == HIPETS_TERM_CARTPOLE) && d->obs_dim < 3) return fail("cartpole fns need obs_dim >= 3");
    if ((d->obs_process == HIPETS_OBS_HALFCHEETAH) && d->obs_dim < 3) return fail("halfcheetah obs_process needs obs_dim >= 3");
    if (d->obs_dim < 2 && (d->termination_fn != HIPETS_TERM_NONE || d->reward_fn == HIPETS_REW_CARTPOLE_PETS)) return fail("termination/reward fn needs obs_dim >= 2");
    if (!d->deterministic && (!d->min_logvar || !d->max_logvar)) return fail("logvar bounds missing");
    if (d->ensemble_kind != HIPETS_ENSEMBLE_GAUSSIAN_MLP && d->ensemble_kind != HIPETS_ENSEMBLE_BASIC)
        return fail("unknown ensemble_kind %d", d->ensemble_kind);
    if (d->ensemble_kind == HIPETS_ENSEMBLE_BASIC && d->n_members != d->ensemble_size)
        return fail("BasicEnsemble has no elite subset (basic_ensemble.py:262-266): n_members %d != ensemble_size %d", d->n_members,
                    d->ensemble_size);
    if (d->normalizer != HIPETS_NORM_NONE && (!d->norm_mean || !d->norm_std)) return fail("normalizer stats missing");
    for (int i = 0; i < d->n_members; ++i)
        if (d->members[i] < 0 || d->members[i] >= d->ensemble_size) return fail("member index %d out of range", d->members[i]);

    ModelDev md{};
    md.obs_dim = d->obs_dim; md.act_dim = d->act_dim; md.in_dim = d->in_dim; md.out_dim = d->out_dim;
    md.out_total = d->deterministic ? d->out_dim : 2 * d->out_dim;
    md.hid = d->hid; md.n_layers = d->n_layers; md.M = d->n_members; md.obs_in = obs_in;
    md.activation = d->activation; md.slope = d->leaky_slope; md.propagation = d->propagation;
    md.deterministic = d->deterministic; md.obs_process = d->obs_process; md.reward_fn = d->reward_fn;
    md.term_fn = d->termination_fn; md.target_is_delta = d->target_is_delta; md.learned_rewards = d->learned_rewards;
    md.normalizer = d->normalizer;
    md.iid_members = d->ensemble_kind == HIPETS_ENSEMBLE_BASIC ? 1 : 0;
    md.lv_rows = (md.iid_members && !d->deterministic) ? d->n_members : 1;
    auto up16 = [](int x) { return (x + 15) / 16 * 16; };
    long long woff = 0;
    int boff = 0, maxK = 0;
    std::vector<int> Ks(d->n_layers), Ns(d->n_layers);
    std::vector<LayerMeta> lms(d->n_layers);
    for (int l = 0; l < d->n_layers; ++l) {
        Ks[l] = l == 0 ? d->in_dim : d->hid;
        Ns[l] = l == d->n_layers - 1 ? md.out_total : d->hid;
        lms[l].Kp = up16(Ks[l]);
        lms[l].Np = up16(Ns[l]);
        lms[l].woff = woff;
        lms[l].boff = boff;
        lms[l].tail_steps = (Ks[l] - (lms[l].Kp - 16) + 3) / 4;
        lms[l].woff_pairs = -1;
        lms[l].boff_pairs = 0;
        lms[l].pad2_ = 0;
        woff += (long long)lms[l].Kp * lms[l].Np;
        boff += lms[l].Np;
        maxK = std::max(maxK, std::max(lms[l].Kp, lms[l].Np));
    }
    if (!d->deterministic) {  // second pack of the mean / logvar head in "head pair" column order (rollout.hpp head_pair_col, KSpec::FUSE)
        LayerMeta& out = lms[d->n_layers - 1];
        out.woff_pairs = woff;
        out.boff_pairs = boff;
        woff += (long long)out.Kp * out.Np;  // ceil(out_dim / 8) column tiles == Np / 16: the pair order never needs more tiles
        boff += out.Np;
    }
    md.precision = d->precision;
    if (d->precision != HIPETS_PREC_F32 && d->precision != HIPETS_PREC_BF16X3) return fail("unknown precision %d", d->precision);
    long long w3off = 0;  // 16-byte units
    int max_kc32 = 1;
    for (int l = 0; l < d->n_layers; ++l) {
        lms[l].Kp32 = (Ks[l] + 31) / 32 * 32;
        lms[l].pad_ = 0;
        lms[l].woff3 = w3off;
        w3off += (long long)(lms[l].Np / 16) * (lms[l].Kp32 / 32) * 3 * 64;
        max_kc32 = std::max(max_kc32, lms[l].Kp32 / 32);
    }
    md.w3member = w3off;
    md.Kp0 = lms[0].Kp;
    md.hidC = up16(d->hid) / kTile;
    md.outC = up16(md.out_total) / kTile;
    md.wmember = woff;
    md.bmember = boff;
    // row stride: >= widest activation, == 8 (mod 64) floats => conflict-free ds_read_b128 A fragments
    int ld = maxK;
    while (ld % 64 != 8) ld += 4;
    if (d->precision == HIPETS_PREC_BF16X3) {
        // activation rows hold [k chunk of 32][3 pieces][32 x bf16] = 192 bytes per chunk; the last layer's fp32 results share the
        // rows; a byte stride that is an ODD multiple of 16 keeps the ds_read_b128 of 16 consecutive rows on distinct slots
        int ldb = std::max(max_kc32 * 192, lms[d->n_layers - 1].Np * 4);
        ldb = (ldb + 15) / 16 * 16;
        if ((ldb / 16) % 2 == 0) ldb += 16;
        ld = ldb / 4;
    }
    md.ld = ld;
    md.ld_in = md.Kp0;  // KSpec::WIDE instances: the model-input image's own row stride
    while (md.ld_in % 64 != 8) md.ld_in += 4;
    if (rollout_smem_bytes(kTile, md.ld, md.obs_dim, md.act_dim, md.in_dim, md.out_dim, md.out_total, 64,
                           md.propagation == HIPETS_PROP_EXPECTATION, md.lv_rows) > e->lds_max)
        return fail("model too wide for LDS (ld=%d)", md.ld);

    if (d->precision == HIPETS_PREC_BF16X3 && e->w3pack.ensure((size_t)md.w3member * md.M * 16)) return 1;
    if (e->wpack.ensure((size_t)md.wmember * md.M * 4)) return 1;
    if (e->bpack.ensure((size_t)md.bmember * md.M * 4)) return 1;
    if (e->members.ensure((size_t)md.M * 4)) return 1;
    HCHECK(hipMemcpyAsync(e->members.p, d->members, (size_t)md.M * 4, hipMemcpyHostToDevice, st));
    if (e->layer_meta.ensure(sizeof(LayerMeta) * d->n_layers)) return 1;
    HCHECK(hipMemcpyAsync(e->layer_meta.p, lms.data(), sizeof(LayerMeta) * d->n_layers, hipMemcpyHostToDevice, st));
    for (int l = 0; l < d->n_layers; ++l) {
        const long long n = (long long)lms[l].Kp * lms[l].Np * md.M;
        hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, e->wpack.as<float>(),
                           reinterpret_cast<const float*>(d->weights[l]), e->members.as<int>(), md.M, Ks[l], Ns[l], lms[l].Kp,
                           lms[l].Np, md.wmember, lms[l].woff, l < d->n_layers - 1 ? 1 : 0, 0);
        HCHECK(hipGetLastError());
        if (d->precision == HIPETS_PREC_BF16X3) {
            const long long n3 = (long long)(lms[l].Np / 16) * (lms[l].Kp32 / 32) * 3 * 64 * md.M;
            hipLaunchKernelGGL(pack_weights_b3_kernel, dim3((unsigned)((n3 + 255) / 256)), dim3(256), 0, st, e->w3pack.as<uint4>(),
                               reinterpret_cast<const float*>(d->weights[l]), e->members.as<int>(), md.M, Ks[l], Ns[l], lms[l].Kp32, lms[l].Np,
                               md.w3member, lms[l].woff3, 0);
            HCHECK(hipGetLastError());
        }
        const int nb = md.M * lms[l].Np;
        hipLaunchKernelGGL(pack_bias_kernel, dim3((nb + 255) / 256), dim3(256), 0, st, e->bpack.as<float>(),
                           reinterpret_cast<const float*>(d->biases[l]), e->members.as<int>(), md.M, Ns[l], lms[l].Np, md.bmember,
                           lms[l].boff, l < d->n_layers - 1 ? 1 : 0, 0);
        HCHECK(hipGetLastError());
        if (lms[l].woff_pairs >= 0) {
            hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, e->wpack.as<float>(),
                               reinterpret_cast<const float*>(d->weights[l]), e->members.as<int>(), md.M, Ks[l], Ns[l], lms[l].Kp,
                               lms[l].Np, md.wmember, lms[l].woff_pairs, 2, 0, d->out_dim);
            HCHECK(hipGetLastError());
            hipLaunchKernelGGL(pack_bias_kernel, dim3((nb + 255) / 256), dim3(256), 0, st, e->bpack.as<float>(),
                               reinterpret_cast<const float*>(d->biases[l]), e->members.as<int>(), md.M, Ns[l], lms[l].Np, md.bmember,
                               lms[l].boff_pairs, 2, d->out_dim);
            HCHECK(hipGetLastError());
        }
    }
    if (d->normalizer != HIPETS_NORM_NONE) {
        if (e->norm_mean.ensure((size_t)d->in_dim * 8) || e->norm_std.ensure((size_t)d->in_dim * 8)) return 1;
        HCHECK(hipMemcpyAsync(e->norm_mean.p, d->norm_mean, (size_t)d->in_dim * 8, hipMemcpyHostToDevice, st));
        HCHECK(hipMemcpyAsync(e->norm_std.p, d->norm_std, (size_t)d->in_dim * 8, hipMemcpyHostToDevice, st));
    }
    if (!d->deterministic) {
        const size_t nlv = (size_t)md.lv_rows * d->out_dim * 4;
        if (e->min_lv.ensure(nlv) || e->max_lv.ensure(nlv)) return 1;
        HCHECK(hipMemcpyAsync(e->min_lv.p, d->min_logvar, nlv, hipMemcpyHostToDevice, st));
        HCHECK(hipMemcpyAsync(e->max_lv.p, d->max_logvar, nlv, hipMemcpyHostToDevice, st));
    }
    std::vector<unsigned char> nd(d->obs_dim, 0);
    for (int i = 0; i < d->n_no_delta; ++i) {
        if (d->no_delta[i] < 0 || d->no_delta[i] >= d->obs_dim) return fail("no_delta index %d out of range", d->no_delta[i]);
        nd[d->no_delta[i]] = 1;
    }
    if (e->no_delta.ensure((size_t)d->obs_dim)) return 1;
    HCHECK(hipMemcpyAsync(e->no_delta.p, nd.data(), (size_t)d->obs_dim, hipMemcpyHostToDevice, st));
    HCHECK(hipStreamSynchronize(st));  // host staging buffers (nd, caller arrays) may go away after return
    md.layers = e->layer_meta.as<LayerMeta>();
    md.w3 = e->w3pack.as<uint4>();
    md.w = e->wpack.as<float>();
    md.b = e->bpack.as<float>();
    md.norm_mean = e->norm_mean.as<double>();
    md.norm_std = e->norm_std.as<double>();
    md.min_lv = e->min_lv.as<float>();
    md.max_lv = e->max_lv.as<float>();
    md.no_delta = e->no_delta.as<unsigned char>();
    e->md = md;
    e->ensemble_size = d->ensemble_size;
    e->has_model = true;
    return 0;
}

int hipets_fast_geometry(hipets_engine* e, int32_t pop, int32_t P, int32_t horizon, int32_t rows_per_group,
                         int32_t* n_workgroups, int32_t* row_tiles) {
    if (!e || !e->has_model) return fail("engine has no model");
    if (pop < 1 || P < 1) return fail("bad pop/P");
    if (rows_per_group < -1 || rows_per_group > kMaxR) return fail("rows_per_group outside [-1, %d]", kMaxR);
    const long long tiles = fast_tiles(pop, P);
    const bool wide = rows_per_group >= 0 && rows_per_group <= 2 && wide_model(e->md);  // a default call runs the WIDE instance there
    const int R = choose_R(e, tiles, 1, rows_per_group < 0 ? 0 : rows_per_group, horizon, wide, horizon > 1);  // (one step: nothing drifts apart, hipets_step)
    if (lds_for(e, R, horizon, wide) > e->lds_max) return fail("rows_per_group %d does not fit LDS", R);
    const long long groups = (tiles + R - 1) / R;
    if (n_workgroups) *n_workgroups = (int)groups;
    if (row_tiles) *row_tiles = R;
    return 0;
}

int hipets_kernel_class(hipets_engine* e, int32_t pop, int32_t P, int32_t horizon, int32_t mode, int32_t rows_per_group, int32_t* kernel_class,
                        int32_t* row_tiles) {
    if (!e || !e->has_model) return fail("engine has no model");
    if (pop < 1 || P < 1 || horizon < 1) return fail("bad pop/horizon/particles");
    if (rows_per_group < 0 || rows_per_group > kMaxR) return fail("rows_per_group outside [0, %d]", kMaxR);
    if (mode != HIPETS_MODE_FAST && mode != HIPETS_MODE_DEVICE) return fail("hipets_kernel_class: mode must be HIPETS_MODE_FAST or HIPETS_MODE_DEVICE");
    const ModelDev& md = e->md;
    RolloutArgs probe{};  // what a default call's arguments look like to the launcher: in-kernel draws, nothing injected or traced
    probe.use_philox = 1;
    probe.mode = mode;
    const bool call_lean = lean_call(md, probe);
    long long tiles;
    int slices;
    if (mode == HIPETS_MODE_DEVICE) {  // rollout_impl's geometry: one slice of B / M rows per member
        const long long B = (long long)pop * P;
        const int domains = md.propagation == HIPETS_PROP_EXPECTATION ? 1 : md.M;
        if (md.iid_members && domains > 1) return fail("DEVICE mode has no BasicEnsemble (iid member map) variant");
        if (B % domains != 0) return fail("GaussianMLP ensemble requires batch size to be a multiple of the number of models. "
                                          "Current batch size is %lld for %d models.", B, md.M);
        tiles = (B / domains + kTile - 1) / kTile;
        slices = domains;
    } else {  // all B rows in one run
        tiles = fast_tiles(pop, P);
        slices = 1;
    }
    const bool wide = wide_model(md) && call_lean && rows_per_group <= 2;
    const int R = choose_R(e, tiles, slices, rows_per_group, horizon, wide, mode == HIPETS_MODE_FAST && horizon > 1);
    if (lds_for(e, R, horizon, wide) > e->lds_max) return fail("the model does not fit LDS");
    int cls = HIPETS_KERNEL_GENERIC;
    if (md.precision == HIPETS_PREC_BF16X3) {
        if (!(call_lean && b3_shape_exists(md, R))) return fail("bf16x3 arithmetic exists for the shape-specialised instances only");
        cls = HIPETS_KERNEL_FUSED;
    } else if (wide) {
        cls = HIPETS_KERNEL_WIDE;
    } else if (call_lean && !wide_model(md) && lean_shape_exists(md, R, mode == HIPETS_MODE_FAST)) {
        cls = HIPETS_KERNEL_FUSED;
    } else {
#define HIPETS_CLASS_HID(HC) if (hid_static_call(md, probe, HC)) cls = HIPETS_KERNEL_HIDDEN_STATIC;
        HIPETS_HID_STATIC_SHAPES(HIPETS_CLASS_HID)
#undef HIPETS_CLASS_HID
    }
    if (kernel_class) *kernel_class = cls;
    if (row_tiles) *row_tiles = R;
    return 0;
}

}  // extern "C"

namespace {
// hipets_rollout with a plan-level shortcut: s0 == nullptr means the initial state(s) are already staged in e->s0
// (the observation is the same for every iteration of a plan).
int rollout_impl(hipets_engine* e, const float* actions, const float* s0, int32_t pop, int32_t H, int32_t P,
                 const hipets_rollout_opts* o, float* returns, void* stream);
}

extern "C" {

int hipets_rollout(hipets_engine* e, const float* actions, const float* s0, int32_t pop, int32_t H, int32_t P,
                   const hipets_rollout_opts* o, float* returns, void* stream) {
    if (!e || !s0) return fail("null argument");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    HCHECK(hipSetDevice(e->device));
    ENTER_STREAM(e, st);
    return rollout_impl(e, actions, s0, pop, H, P, o, returns, stream);
}

}  // extern "C"

namespace {
int rollout_impl(hipets_engine* e, const float* actions, const float* s0, int32_t pop, int32_t H, int32_t P,
                 const hipets_rollout_opts* o, float* returns, void* stream) {
    if (!e || !e->has_model) return fail("engine has no model (call hipets_set_model)");
    if (!actions || !o) return fail("null argument");  // (returns == nullptr: internal callers that fold the particle mean)
    if (pop < 1 || H < 1 || P < 1) return fail("bad pop/horizon/particles");
    if (e->error_flag && *e->error_flag) {  // raised by an EARLIER launch: its returns were garbage
        *e->error_flag = 0;
        e->persistent_ok = false;  // fall back to one launch per step from now on
        return fail_kind(HIPETS_ERR_TIMEOUT, "a persistent DEVICE-mode rollout timed out waiting for rows of another workgroup (its workgroups were not all "
                    "resident) and nobody asked (hipets_check_async_error after the results were read): the results of that earlier "
                    "call are invalid.  Persistent launches are now disabled for this engine.");
    }
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);  // (the public entry point that led here holds the StreamScope)
    HCHECK(hipSetDevice(e->device));
    const ModelDev& md = e->md;
    const long long B = (long long)pop * P;
    if (B > 0x7FFFFFFF / std::max(md.obs_dim, md.out_dim)) return fail("batch too large");
    if (o->rows_per_group < 0 || o->rows_per_group > kMaxR) return fail("rows_per_group outside [0, %d]", kMaxR);

    const int n_env = o->n_env > 1 ? o->n_env : 1;
    if (n_env > 1 && ((o->mode != HIPETS_MODE_FAST && o->mode != HIPETS_MODE_DEVICE) || pop % n_env != 0))
        return fail("n_env %d needs FAST or DEVICE mode and a population (%d) divisible by it", n_env, pop);
    if (e->s0.ensure((size_t)n_env * md.obs_dim * 4)) return 1;
    if (e->totals.ensure((size_t)B * 4)) return 1;
    if (s0 && stage_h2d(e, e->s0.p, s0, (size_t)n_env * md.obs_dim * 4, st)) return 1;

    RolloutArgs ra{};
    ra.pop = pop; ra.P = P; ra.H = H; ra.B = (int)B;
    ra.mode = o->mode;
    ra.actions = actions;
    ra.s0 = e->s0.as<float>();
    ra.totals = e->totals.as<float>();
    ra.seed = o->seed;
    ra.stream_id = o->stream_id;
    ra.trace_next_obs = o->trace_next_obs;
    ra.trace_rewards = o->trace_rewards;
    ra.phase_cycles = reinterpret_cast<long long*>(o->phase_cycles);
    ra.pop_env = n_env > 1 ? pop / n_env : 0;
    ra.generic_only = o->generic_kernel;

    if (o->mode == HIPETS_MODE_EXACT || o->mode == HIPETS_MODE_DEVICE) {
        // Reference propagation semantics: per step ONE balanced permutation of all B rows, slot j -> member j / (B / M)
        // (gaussian_mlp.py:164-166, 203-205).  EXACT takes the permutations and eps from the caller (the reference's own
        // draws); DEVICE evaluates a keyed bijection and Philox normals in-kernel (no input tensors, no host work).
        const bool device = o->mode == HIPETS_MODE_DEVICE;
        const bool expectation = md.propagation == HIPETS_PROP_EXPECTATION;
        const int domains = expectation ? 1 : md.M;
        // explicit per-row member maps (padded member slots, opts.rows_per_member): what BasicEnsemble draws with randint
        // (basic_ensemble.py:122-129) and what mbrl.util.math.propagate_from_indices expresses (util/math.py:180-196);
        // accepted for GaussianMLP models too (any batch size, members may own unequal row counts)
        const bool slots = !device && !expectation && o->rows_per_member > 0;
        if (!md.iid_members && !slots && B % md.M != 0)  // the reference's ValueError (gaussian_mlp.py:195-200), raised for every propagation method
            return fail("GaussianMLP ensemble requires batch size to be a multiple of the number of models. "
                        "Current batch size is %lld for %d models.", B, md.M);
        if (!expectation) {
            if (device && md.iid_members)
                return fail("DEVICE mode has no BasicEnsemble (iid member map) variant: use FAST, or EXACT with injected maps");
            if (!device && !o->perms) return fail("EXACT mode with random_model/fixed_model propagation needs opts.perms");
            if ((md.iid_members && !device && !slots) || (slots && o->rows_per_member > B))
                return fail("EXACT mode with per-row member maps needs opts.rows_per_member in [1, B] (padded member slots)");
        }
        const int rpd = expectation ? (int)B : (slots ? o->rows_per_member : (int)(B / domains));
        const long long tiles = (rpd + kTile - 1) / kTile;
        bool wide = false;
        if (device && wide_model(md)) {  // will the launcher pick the WIDE instance?  (its lean_call on what `ra` is going to hold)
            RolloutArgs probe = ra;
            probe.eps = nullptr;
            probe.use_philox = o->no_sample ? 0 : 1;
            wide = lean_call(md, probe) && o->rows_per_group <= 2;
        }
        ra.wide_lds = wide ? 1 : 0;
        const int R = choose_R(e, tiles, domains, o->rows_per_group, H, wide, false);
        const size_t lds = lds_for(e, R, H, wide);
        if (lds > e->lds_max) return fail("rows_per_group %d does not fit LDS", R);
        const int groups = (int)((tiles + R - 1) / R);
        // rows change workgroups between steps only when a fresh permutation is drawn per step: one launch per step then
        // (state through HBM); TS-infinity / expectation rollouts of DEVICE mode keep their rows and run as ONE launch
        const bool per_step = !device || md.propagation == HIPETS_PROP_RANDOM_MODEL;
        if (e->state.ensure((size_t)B * md.obs_dim * 4) || e->term.ensure((size_t)B)) return 1;
        ra.groups = groups;
        ra.rows_per_domain = rpd;
        ra.state = e->state.as<float>();
        ra.term = e->term.as<unsigned char>();
        if (device) {
            ra.perm = nullptr;
            ra.eps = nullptr;
            ra.use_philox = o->no_sample ? 0 : 1;
            if (!expectation) {
                ra.perm_n = (unsigned)B;
                perm_radices((uint32_t)B, &ra.perm_a, &ra.perm_b);
                ra.perm_keys = perm_round_keys(perm_key(o->seed, o->stream_id, 0xFFFFFFFFu));  // fixed_model; random_model: per step below
            }
        } else {
            ra.perm = expectation ? nullptr : reinterpret_cast<const long long*>(o->perms);
            ra.perm_step = md.propagation == HIPETS_PROP_RANDOM_MODEL ? (long long)domains * rpd : 0;
            ra.eps = o->eps;
            ra.use_philox = 0;
        }
        // DEVICE + random_model: ONE launch for the horizon, rows handed over between workgroups through the tagged-granule
        // table.  Only as many workgroups as are resident at once are launched; a batch with more logical workgroups (cfg4: 435)
        // is served in turns, workgroup b taking b, b + grid, ... every step.
        bool persistent = device && per_step && e->persistent_ok && H > 1;
        if (persistent) {
            // How many workgroups of this kernel instance can wait for each other (be resident at once)?  The launcher answers
            // from the occupancy arithmetic AND a one-time self-test of the instance at this grid size (a census launch + one
            // stream synchronisation the first time a larger grid is asked for); 0 = the self-test failed (CUs held by another
            // process, a masked device, an occupancy estimate that does not hold on this ROCm build): launch per step then.
            int capacity = 0;
            RolloutArgs q = ra;
            q.exchange = reinterpret_cast<unsigned long long*>(&capacity);  // marks the persistent form for the launcher; never dereferenced
            q.capacity_out = &capacity;
            q.census = e->census.as<int>();
            q.poll_ticks = std::max(e->poll_ticks, 20000000ll);  // the self-test keeps its 0.2 s whatever bound the hand-over polls were given
            if (launch_rollout(e, R, domains * groups, lds, q, st)) return 1;
            if (capacity <= 0) {
                e->persistent_ok = false;  // stays off until hipets_set_persistent(e, 1)
                persistent = false;
            } else {
                // Turns pay off when a CU holds ONE workgroup of this instance (cfg4: 4.51 -> 4.11 ms per rollout, cfg4' 16.4 -> 15.0).
                // Where two are resident, one launch per step lets the hardware deal 1 250 workgroups to 512 slots as they free up;
                // fixed turns (3 for some workgroups, 2 for the rest) measured slower there (cfg5: 7.3 vs 6.6 ms).
                persistent = domains * groups <= capacity || capacity <= e->num_cu;
            }
        }
        if (!persistent) {  // the persistent form starts from s0 itself and writes every row's total at the end
            hipLaunchKernelGGL(init_state_kernel, dim3((unsigned)((B * md.obs_dim + 255) / 256)), dim3(256), 0, st,
                               e->state.as<float>(), e->totals.as<float>(), e->term.as<unsigned char>(), e->s0.as<float>(), (int)B,
                               md.obs_dim, P, ra.pop_env);
            HCHECK(hipGetLastError());
        }
        if (persistent) {
            const size_t nv = 2 * ((size_t)(md.obs_dim + 1) / 2 + 1);  // granules per row: the state dims padded to pairs, then {total, flag}
            const size_t cap_before = e->exchange.cap;
            if (e->exchange.ensure((size_t)B * nv * 8)) return 1;
            // hand-over tags grow monotonically across launches (step t of this launch: tag_base + t + 1), so a granule left by an
            // earlier rollout can never pass for this one's: the table is cleared only when it is new or the 32-bit tag would wrap
            if (e->exchange.cap != cap_before || e->tag_base > 0xFFFFFFFFu - 2u * (uint32_t)H - 2u) {
                HCHECK(hipMemsetAsync(e->exchange.p, 0, e->exchange.cap, st));
                e->tag_base = 0;
            }
            ra.tag_base = e->tag_base;
            e->tag_base += (uint32_t)H;
            ra.exchange = e->exchange.as<unsigned long long>();
            const uint64_t sid = o->stream_id;
            if (e->plan_keys.p && o->seed == e->plan_keys_seed && H == e->plan_keys_H && sid >= e->plan_keys_first &&
                sid - e->plan_keys_first < (uint64_t)e->plan_keys_count) {
                ra.step_keys = e->plan_keys.as<PermKeys>() + (size_t)(sid - e->plan_keys_first) * H;  // generated by the plan's prologue
            } else {
                if (e->step_keys.ensure((size_t)H * sizeof(PermKeys))) return 1;
                hipLaunchKernelGGL(step_keys_kernel, dim3((H + 63) / 64), dim3(64), 0, st, e->step_keys.as<PermKeys>(), H, (unsigned long long)o->seed,
                                   (unsigned long long)o->stream_id);
                HCHECK(hipGetLastError());
                ra.step_keys = e->step_keys.as<PermKeys>();
            }
            ra.error_flag = e->error_flag;
            ra.poll_ticks = e->poll_ticks;
            ra.t_begin = 0;
            ra.t_end = H;
            ra.n_logical = domains * groups;
            // KSpec::WIDE two-tile instances deal a ragged last turn in one-tile logical workgroups (rollout.hpp "Ragged last turn": same
            // bits, 0.69 of the turn's time); HIPETS_RAGGED_LAST_TURN=0 keeps two-tile turns throughout (A/B measurements)
            static const bool ragged_ok = [] { const char* v = std::getenv("HIPETS_RAGGED_LAST_TURN"); return !(v && v[0] == '0'); }();
            ra.ragged_last_turn = (wide && R == 2 && ragged_ok) ? 1 : 0;

            if (launch_rollout(e, R, domains * groups, lds, ra, st)) return 1;  // cut to the resident capacity by the launcher
        } else if (per_step) {
            for (int t = 0; t < H; ++t) {
                ra.t_begin = t;
                ra.t_end = t + 1;
                if (ra.perm_n) ra.perm_keys = perm_round_keys(perm_key(o->seed, o->stream_id, (uint32_t)t));  // this step's permutation
                if (launch_rollout(e, R, domains * groups, lds, ra, st)) return 1;
            }
        } else {
            ra.t_begin = 0;
            ra.t_end = H;
            if (launch_rollout(e, R, domains * groups, lds, ra, st)) return 1;
        }
    } else if (o->mode == HIPETS_MODE_FAST) {
        const long long tiles = fast_tiles(pop, P);
        ra.eps = o->fast_eps;
        ra.use_philox = (o->fast_eps || o->no_sample) ? 0 : 1;
        // (a caller-sized member schedule follows hipets_fast_geometry: the default call's geometry -- the WIDE instance's where one
        // will run; rows_per_group = -1 asks for the general layout's, which is what calls with injected eps / traces run)
        const bool wide = wide_model(md) && lean_call(md, ra) && o->rows_per_group <= 2;
        if (!wide && wide_model(md) && o->member_schedule && o->rows_per_group == 0)
            return fail("this call runs the general kernel layout (injected eps / traces / generic_kernel) on a model whose default geometry is "
                        "the wide-output instance's: size member_schedule with hipets_fast_geometry(rows_per_group = -1) and pass its row-tile "
                        "count as opts->rows_per_group");
        ra.wide_lds = wide ? 1 : 0;
        const int R = choose_R(e, tiles, 1, o->rows_per_group, H, wide, H > 1);
        const size_t lds = lds_for(e, R, H, wide);
        if (lds > e->lds_max) return fail("rows_per_group %d does not fit LDS", R);
        const int groups = (int)((tiles + R - 1) / R);
        const int nwg = groups;
        if (nwg > 8000) return fail("FAST mode supports at most 8000 workgroups per launch (got %d); shard the population", nwg);
        ra.groups = groups;
        if (o->member_schedule && o->member_schedule_len != 0 && (long long)o->member_schedule_len != (long long)H * nwg)
            return fail("member_schedule holds %d entries, this call's geometry is horizon %d x %d workgroups (hipets_fast_geometry)",
                        o->member_schedule_len, H, nwg);
        // the caller's schedule, or (null) every workgroup draws its own entries in its prologue (common.hpp fast_member)
        ra.schedule = md.propagation != HIPETS_PROP_EXPECTATION ? o->member_schedule : nullptr;
        perm_radices((uint32_t)nwg, &ra.fm_a, &ra.fm_b);
        ra.t_begin = 0;
        ra.t_end = H;
        if (launch_rollout(e, R, nwg, lds, ra, st)) return 1;
    } else {
        return fail("unknown rollout mode %d", o->mode);
    }
    if (!returns) return 0;  // the caller reduces e->totals over the particles itself (hipets_plan_cem: inside the refit kernel)
    hipLaunchKernelGGL(particle_mean_kernel, dim3((pop + 255) / 256), dim3(256), 0, st, e->totals.as<float>(), returns, pop, P);
    HCHECK(hipGetLastError());
    return 0;
}

}  // namespace

namespace {

// First local failure of a sharded plan on this rank (message + class).  A rank on which something cannot be enqueued must NOT
// leave the plan: its peers are, or will be, waiting in this and the remaining iterations' collectives.  It keeps contributing
// (stale) shards to every ncclAllGather and reports its own error at the end; the peers' plans are then built on garbage, which is
// why hipets.dist agrees on the outcome over all ranks (an all-reduce of the status) before anybody uses a plan.
struct LocalErr {
    std::string msg;
    int kind = HIPETS_ERR_NONE;
    bool ok() const { return kind == HIPETS_ERR_NONE; }
    void note(const int rc) {  // rc of a call that has just set g_err / g_err_kind
        if (rc && ok()) { msg = g_err; kind = g_err_kind == HIPETS_ERR_NONE ? HIPETS_ERR_RUNTIME : g_err_kind; }
    }
    int report() const {
        if (ok()) return 0;
        g_err = msg;
        g_err_kind = kind;
        return 1;
    }
};

// Everything that can fail for ONE rank's shard size must fail on EVERY rank, before the first collective (a rank that returned
// early would leave its peers blocked in ncclAllGather): the shards of `rows` candidates hold rows / world or one more, and
// EXACT / DEVICE-mode rollouts of a GaussianMLP ensemble need shard rows % members == 0 (gaussian_mlp.py:195-200) for both sizes.
int check_shards(const hipets_engine* e, const int rows, const int P) {
    const int world = e->comm_world;
    if (rows < world) return fail("population_size %d < world_size %d", rows, world);
    if (e->plan_mode == HIPETS_MODE_DEVICE && !e->md.iid_members) {
        const int base = rows / world, extra = rows % world;
        for (int n : {base, extra ? base + 1 : base})
            if (((long long)n * P) % e->md.M != 0)
                return fail("GaussianMLP ensemble requires batch size to be a multiple of the number of models. A shard of %d candidates x %d "
                            "particles = %lld rows for %d models (population %d over %d ranks).", n, P, (long long)n * P, e->md.M, rows, world);
    }
    return 0;
}

// The objective of one iteration of a sharded plan: `population` holds ALL `rows` candidates (sampled identically on
// every rank: counter-based RNG), this rank rolls out its shard shard_bounds(rows, world, rank) with all particles, ONE
// ncclAllGather of the per-candidate returns (padded to ceil(rows / world) per rank), and e->values [rows] then holds every
// candidate's return on every rank.  The caller has sized e->values / shard_values / gathered.  Returns non-zero only when the
// collective itself failed (RCCL error: the plan is over for everybody); local failures go to *le and the collective still runs.
int sharded_evaluate(hipets_engine* e, const float* population, const int rows, const int H, const int P, const hipets_rollout_opts* ro,
                     void* stream, LocalErr* le) {
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int world = e->comm_world, rank = e->comm_rank;
    int lo, hi;
    shard_bounds(rows, world, rank, &lo, &hi);
    const int width = (rows + world - 1) / world;
    const size_t nd = (size_t)H * e->md.act_dim;
    float* shard_out = world == 1 ? e->values.as<float>() : e->shard_values.as<float>();
    if (le->ok()) le->note(rollout_impl(e, population + (size_t)lo * nd, nullptr, hi - lo, H, P, ro, shard_out, stream));
    if (world > 1) {  // every rank, every iteration, whatever happened locally
        NCHECK(g_rccl.AllGather(e->shard_values.p, e->gathered.p, (size_t)width, 7 /* ncclFloat32 */, e->comm, st));
        if (le->ok()) {
            hipLaunchKernelGGL(unpad_shards_kernel, dim3((rows + 255) / 256), dim3(256), 0, st, e->gathered.as<float>(), e->values.as<float>(), rows,
                               world, width);
            const hipError_t err = hipGetLastError();
            if (err != hipSuccess) le->note(fail_kind(HIPETS_ERR_RUNTIME, "unpad_shards_kernel launch failed: %s", hipGetErrorString(err)));
        }
    }
    return 0;
}

int plan_mppi_impl(hipets_engine* e, int32_t pop, int32_t H, int32_t A, int32_t num_iterations, double gamma, double beta, int32_t n_env,
                   float* mean, const float* lower, const float* upper, const float* s0, int32_t P, uint64_t seed, uint64_t plan_id,
                   void* stream, bool sharded);
int plan_icem_impl(hipets_engine* e, const hipets_icem_params* p, int32_t n_env, const float* x0, const float* lower, const float* upper,
                   float* elite, int32_t has_elite, const int32_t* keep_idx, const float* s0, int32_t P, uint64_t seed, uint64_t plan_id,
                   float* out, void* stream, bool sharded);

}  // namespace

extern "C" {

int hipets_step(hipets_engine* e, const float* obs, const float* actions, int32_t B, const hipets_rollout_opts* o, float* next_obs,
                float* rewards, uint8_t* dones, void* stream) {
    if (!e || !e->has_model) return fail("engine has no model (call hipets_set_model)");
    if (!obs || !actions || !o || !next_obs || !rewards || !dones) return fail("null argument");
    if (B < 1) return fail("bad batch");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    HCHECK(hipSetDevice(e->device));
    ENTER_STREAM(e, st);
    const ModelDev& md = e->md;
    if (o->rows_per_group < 0 || o->rows_per_group > kMaxR) return fail("rows_per_group outside [0, %d]", kMaxR);
    if (!md.iid_members && B % md.M != 0 && !(o->mode == HIPETS_MODE_EXACT && o->rows_per_member > 0))  // gaussian_mlp.py:195-200
        return fail("GaussianMLP ensemble requires batch size to be a multiple of the number of models. "
                    "Current batch size is %d for %d models.", B, md.M);
    // the kernel updates state / totals / terminated in place: run it on the caller's output buffers
    HCHECK(hipMemcpyAsync(next_obs, obs, (size_t)B * md.obs_dim * 4, hipMemcpyDeviceToDevice, st));
    HCHECK(hipMemsetAsync(rewards, 0, (size_t)B * 4, st));
    HCHECK(hipMemsetAsync(dones, 0, (size_t)B, st));
    RolloutArgs ra{};
    ra.pop = B; ra.P = 1; ra.H = 1; ra.B = B;
    ra.mode = o->mode;
    ra.actions = actions;
    ra.s0 = nullptr;
    ra.state = next_obs;
    ra.totals = rewards;
    ra.term = dones;
    ra.seed = o->seed;
    ra.stream_id = o->stream_id;
    ra.trace_next_obs = nullptr;
    ra.trace_rewards = nullptr;
    ra.phase_cycles = nullptr;
    ra.generic_only = o->generic_kernel;
    ra.t_begin = 0;
    ra.t_end = 1;
    if (o->mode == HIPETS_MODE_EXACT || o->mode == HIPETS_MODE_DEVICE) {
        const bool device = o->mode == HIPETS_MODE_DEVICE;
        const bool expectation = md.propagation == HIPETS_PROP_EXPECTATION;
        const int domains = expectation ? 1 : md.M;
        const bool slots = !device && !expectation && o->rows_per_member > 0;  // explicit per-row member maps, see hipets_rollout
        if (!expectation && device && md.iid_members)
            return fail("DEVICE mode has no BasicEnsemble (iid member map) variant: use FAST, or EXACT with injected maps");
        if (!expectation && !device && !o->perms) return fail("EXACT mode with random_model/fixed_model propagation needs opts.perms");
        if (!expectation && ((md.iid_members && !device && !slots) || (slots && o->rows_per_member > B)))
            return fail("EXACT mode with per-row member maps needs opts.rows_per_member in [1, B] (padded member slots)");
        const int rpd = expectation ? B : (slots ? o->rows_per_member : B / domains);
        const long long tiles = (rpd + kTile - 1) / kTile;
        const int R = choose_R(e, tiles, domains, o->rows_per_group, 1, false, false);
        const size_t lds = lds_for(e, R, 1);
        if (lds > e->lds_max) return fail("rows_per_group %d does not fit LDS", R);
        ra.groups = (int)((tiles + R - 1) / R);
        ra.rows_per_domain = rpd;
        if (device) {
            ra.use_philox = o->no_sample ? 0 : 1;
            if (!expectation) {
                ra.perm_n = (unsigned)B;
                perm_radices((uint32_t)B, &ra.perm_a, &ra.perm_b);
                // random_model: the permutation of (seed, stream_id), step 0.  fixed_model (a ModelEnv.step of a TS-infinity rollout keeps
                // its member map while the eps change): the TS-infinity permutation of (seed, perm_stream_id) -- the stream of the reset --
                // next to eps drawn from (seed, stream_id), the stream of the step
                const bool fixed = md.propagation == HIPETS_PROP_FIXED_MODEL;
                const uint64_t pstream = (fixed && o->perm_stream_id) ? o->perm_stream_id : o->stream_id;
                ra.perm_keys = perm_round_keys(perm_key(o->seed, pstream, fixed ? 0xFFFFFFFFu : 0u));
            }
        } else {
            ra.perm = expectation ? nullptr : reinterpret_cast<const long long*>(o->perms);
            ra.perm_step = 0;
            ra.eps = o->eps;
            ra.use_philox = 0;
        }
        if (launch_rollout(e, R, domains * ra.groups, lds, ra, st)) return 1;
    } else if (o->mode == HIPETS_MODE_FAST) {
        // One step of B independent rows: workgroup w owns rows [w * 16 R, (w + 1) * 16 R) and runs the member the FAST rule gives it
        // (the caller's schedule, else common.hpp fast_member).  For ONE step the per-step launch form -- rows and state in HBM around
        // the launch -- IS the FAST form, and every shape-specialised instance has it: the launch below is a DEVICE-form launch with
        // the identity permutation (one domain of B rows) and RolloutArgs::fast_members.  (Until round 6 this was a FAST-path launch
        // with per-row initial states, which only the generic / hidden-static instances have, behind a member-schedule kernel that
        // ranked all workgroups' sort keys: a 100 000-row call took 0.44 ms where DEVICE mode took 0.34.)
        const long long tiles = (B + kTile - 1) / kTile;
        const int R = choose_R(e, tiles, 1, o->rows_per_group, 1, false, false);  // (a single step: two workgroups on a CU do not drift apart -- priced like DEVICE's)
        const size_t lds = lds_for(e, R, 1);
        if (lds > e->lds_max) return fail("rows_per_group %d does not fit LDS", R);
        const int nwg = (int)((tiles + R - 1) / R);
        if (nwg > 8000) return fail("FAST mode supports at most 8000 workgroups per launch (got %d)", nwg);
        if (o->member_schedule && o->member_schedule_len != 0 && o->member_schedule_len != nwg)
            return fail("member_schedule holds %d entries, this call's geometry is %d workgroups (hipets_fast_geometry with rows_per_group -1)",
                        o->member_schedule_len, nwg);
        ra.mode = HIPETS_MODE_DEVICE;  // the kernel's per-step launch form (see above); nothing else reads the mode
        ra.groups = nwg;
        ra.rows_per_domain = B;
        ra.eps = o->fast_eps;
        ra.use_philox = (o->fast_eps || o->no_sample) ? 0 : 1;
        ra.fast_members = 1;
        ra.schedule = md.propagation != HIPETS_PROP_EXPECTATION ? o->member_schedule : nullptr;
        perm_radices((uint32_t)nwg, &ra.fm_a, &ra.fm_b);
        if (launch_rollout(e, R, nwg, lds, ra, st)) return 1;
    } else {
        return fail("unknown rollout mode %d", o->mode);
    }
    return 0;
}

int hipets_fast_schedule(hipets_engine* e, int32_t H, int32_t nwg, uint64_t seed, uint64_t stream_id, int32_t* schedule,
                         void* stream) {
    if (!e || !e->has_model) return fail("engine has no model");
    if (!schedule || H < 1 || nwg < 1) return fail("bad argument");
    HCHECK(hipSetDevice(e->device));
    uint32_t fa, fb;
    perm_radices((uint32_t)nwg, &fa, &fb);
    hipLaunchKernelGGL(member_schedule_kernel, dim3((unsigned)((nwg + 255) / 256), H), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), schedule, nwg, fa, fb,
                       e->md.M, e->md.propagation == HIPETS_PROP_FIXED_MODEL ? 1 : 0, e->md.iid_members, (unsigned long long)seed,
                       (unsigned long long)stream_id);
    HCHECK(hipGetLastError());
    return 0;
}

int hipets_fast_normals(hipets_engine* e, int32_t H, int32_t B, uint64_t seed, uint64_t stream_id, float* normals, void* stream) {
    if (!e || !e->has_model) return fail("engine has no model");
    if (!normals || H < 1 || B < 1) return fail("bad argument");
    HCHECK(hipSetDevice(e->device));
    const long long n = (long long)H * B * ((e->md.out_dim + 3) / 4);
    hipLaunchKernelGGL(export_normals_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       normals, H, B, e->md.out_dim, (unsigned long long)seed, (unsigned long long)stream_id);
    HCHECK(hipGetLastError());
    return 0;
}

int hipets_device_perms(hipets_engine* e, int32_t H, int32_t B, uint64_t seed, uint64_t stream_id, int64_t* perms, void* stream) {
    if (!e || !e->has_model) return fail("engine has no model");
    if (!perms || H < 1 || B < 1) return fail("bad argument");
    HCHECK(hipSetDevice(e->device));
    uint32_t a, b;
    perm_radices((uint32_t)B, &a, &b);
    const int fixed = e->md.propagation == HIPETS_PROP_FIXED_MODEL ? 1 : 0;
    const int rows = fixed ? 1 : H;
    const long long n = (long long)rows * B;
    hipLaunchKernelGGL(export_perms_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<long long*>(perms), rows, (unsigned)B, a, b, fixed, (unsigned long long)seed,
                       (unsigned long long)stream_id);
    HCHECK(hipGetLastError());
    return 0;
}

int hipets_set_plan_mode(hipets_engine* e, int32_t mode) {
    if (!e) return fail("null engine");
    if (mode != HIPETS_MODE_FAST && mode != HIPETS_MODE_DEVICE) return fail("plan mode must be HIPETS_MODE_FAST or HIPETS_MODE_DEVICE");
    e->plan_mode = mode;
    return 0;
}

int hipets_set_persistent(hipets_engine* e, int32_t on) {
    if (!e) return fail("null engine");
    if (on && !e->error_flag) return fail("persistent DEVICE-mode launches need the host-mapped timeout flag, which could not be allocated");
    e->persistent_ok = on != 0;
    return 0;
}

int hipets_set_handover_timeout(hipets_engine* e, double seconds) {
    if (!e) return fail("null engine");
    if (!(seconds >= 0.0) || seconds > 60.0) return fail("hand-over timeout %g s outside [0, 60]", seconds);
    e->poll_ticks = (long long)(seconds * 1.0e8);  // the kernel's wall clock runs at 100 MHz
    return 0;
}

int hipets_check_async_error(hipets_engine* e, int32_t* timed_out) {
    if (!e || !timed_out) return fail("null argument");
    *timed_out = 0;
    if (e->error_flag && *e->error_flag) {
        *e->error_flag = 0;
        e->persistent_ok = false;  // per-step launches from now on (hipets_set_persistent(e, 1) switches back)
        *timed_out = 1;
        g_err_kind = HIPETS_ERR_TIMEOUT;
        g_err = "a persistent DEVICE-mode rollout gave up waiting for rows of another workgroup (its workgroups were not all resident: "
                "another process or stream held CUs); everything computed from that launch on is invalid -- re-run the call.  "
                "Persistent launches are now disabled for this engine.";
    }
    return 0;
}

int hipets_set_plan_trace(hipets_engine* e, const hipets_plan_trace* t) {
    if (!e) return fail("null engine");
    e->has_trace = t != nullptr;
    if (t) e->trace = *t;
    return 0;
}

int hipets_cem_sample(hipets_engine* e, const hipets_cem_params* p, const float* mu, const float* dispersion,
                      const float* lower, const float* upper, const float* z, uint64_t seed, uint64_t stream_id,
                      float* population, void* stream) {
    if (!e) return fail("null engine");
    if (check_cem(p)) return 1;
    if (!mu || !dispersion || !lower || !upper || !population) return fail("null argument");
    HCHECK(hipSetDevice(e->device));
    const CemDev c = make_cem(p);
    const long long n = (long long)c.pop * c.D;
    hipLaunchKernelGGL(cem_sample_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), c,
                       mu, dispersion, lower, upper, z, (unsigned long long)seed, (unsigned long long)stream_id, population);
    HCHECK(hipGetLastError());
    return 0;
}

int hipets_cem_refit(hipets_engine* e, const hipets_cem_params* p, float* values, const float* population, float* mu,
                     float* dispersion, float* best_value, float* best_solution, int32_t* elite_idx, void* stream) {
    if (!e) return fail("null engine");
    if (check_cem(p)) return 1;
    if (!values || !population || !mu || !dispersion || !best_value || !best_solution) return fail("null argument");
    HCHECK(hipSetDevice(e->device));
    const CemDev c = make_cem(p);
    int n2 = 1;
    while (n2 < c.pop) n2 <<= 1;
    hipLaunchKernelGGL(cem_refit_kernel, dim3(refit_blocks(c.D), 1), dim3(kRefitThreads), (size_t)n2 * 8 + kRefitScratchBytes, reinterpret_cast<hipStream_t>(stream), c,
                       values, population, mu, dispersion, best_value, best_solution, elite_idx);
    HCHECK(hipGetLastError());
    return 0;
}

int hipets_cem_refit_elites(hipets_engine* e, const hipets_cem_params* p, float* values, const float* population, const int32_t* elites,
                            float* mu, float* dispersion, float* best_value, float* best_solution, void* stream) {
    if (!e) return fail("null engine");
    if (check_cem(p)) return 1;
    if (!values || !population || !elites || !mu || !dispersion || !best_value || !best_solution) return fail("null argument");
    HCHECK(hipSetDevice(e->device));
    CemDev c = make_cem(p);
    c.elite_in = elites;
    int n2 = 1;
    while (n2 < c.pop) n2 <<= 1;
    hipLaunchKernelGGL(cem_refit_kernel, dim3(refit_blocks(c.D), 1), dim3(kRefitThreads), (size_t)n2 * 8 + kRefitScratchBytes, reinterpret_cast<hipStream_t>(stream), c,
                       values, population, mu, dispersion, best_value, best_solution, (int*)nullptr);
    HCHECK(hipGetLastError());
    return 0;
}

int hipets_gather_rows(hipets_engine* e, int32_t rows, int32_t dim, const float* src, const int32_t* index, float* dst,
                       void* stream) {
    if (!e || !src || !index || !dst || rows < 1 || dim < 1) return fail("bad argument");
    HCHECK(hipSetDevice(e->device));
    const long long n = (long long)rows * dim;
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), rows,
                       dim, src, index, dst, 0ll, 0ll);
    HCHECK(hipGetLastError());
    return 0;
}

int hipets_mppi_sample(hipets_engine* e, int32_t pop, int32_t H, int32_t A, double beta, const float* mean, const float* past_action,
                       const float* lower, const float* upper, const float* z, uint64_t seed, uint64_t stream_id,
                       float* population, void* stream) {
    if (!e || !mean || !past_action || !lower || !upper || !population) return fail("null argument");
    if (pop < 1 || H < 1 || A < 1) return fail("bad pop/horizon/act_dim");
    HCHECK(hipSetDevice(e->device));
    return launch_mppi_sample(1, pop, H, A, (float)beta, mean, past_action, lower, upper, z, seed, stream_id, population, reinterpret_cast<hipStream_t>(stream));
}

int hipets_mppi_update(hipets_engine* e, int32_t pop, int32_t H, int32_t A, double gamma, float* values, const float* population,
                       float* mean, void* stream) {
    if (!e || !values || !population || !mean) return fail("null argument");
    if (pop < 1 || pop > 12000 || H < 1 || A < 1) return fail("population_size %d outside [1, 12000]", pop);
    HCHECK(hipSetDevice(e->device));
    return launch_mppi_update(e, 1, pop, H * A, (float)gamma, values, population, mean, reinterpret_cast<hipStream_t>(stream));
}

int hipets_icem_sample(hipets_engine* e, int32_t n, int32_t H, int32_t A, double exponent, const float* mu, const float* var,
                       const float* lower, const float* upper, const float* normals, uint64_t seed, uint64_t stream_id,
                       float* population, void* stream) {
    if (!e || !mu || !var || !lower || !upper || !population) return fail("null argument");
    if (n < 1 || A < 1) return fail("bad n/act_dim");
    if (H < 2 || H > kMaxHorizon) return fail("iCEM horizon %d outside [2, %d]", H, kMaxHorizon);
    HCHECK(hipSetDevice(e->device));
    launch_icem_sample(reinterpret_cast<hipStream_t>(stream), 1, n, n, H, A, (float)exponent, mu, var, lower, upper, normals, (unsigned long long)seed,
                       (unsigned long long)stream_id, population);
    HCHECK(hipGetLastError());
    return 0;
}

int hipets_icem_shift(hipets_engine* e, int32_t keep, int32_t H, int32_t A, const float* kept, const float* mu, const float* var,
                      const float* end_noise, uint64_t seed, uint64_t stream_id, float* out, void* stream) {
    if (!e || !kept || !mu || !var || !out) return fail("null argument");
    if (keep < 1 || H < 1 || A < 1) return fail("bad keep/horizon/act_dim");
    HCHECK(hipSetDevice(e->device));
    const int n = keep * H * A;
    hipLaunchKernelGGL(icem_shift_kernel, dim3((n + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), 1, keep, keep, H, A, kept,
                       mu, var, end_noise, (unsigned long long)seed, (unsigned long long)stream_id, out);
    HCHECK(hipGetLastError());
    return 0;
}

int hipets_plan_cem(hipets_engine* e, const hipets_cem_params* p, const float* x0, const float* lower, const float* upper,
                    const float* s0, int32_t P, uint64_t seed, uint64_t plan_id, float* out, void* stream) {
    return hipets_plan_cem_batched(e, p, 1, x0, lower, upper, s0, P, seed, plan_id, out, stream);
}

int hipets_plan_cem_batched(hipets_engine* e, const hipets_cem_params* p, int32_t n_env, const float* x0, const float* lower,
                            const float* upper, const float* s0, int32_t P, uint64_t seed, uint64_t plan_id, float* out,
                            void* stream) {
    if (!e || !e->has_model) return fail("engine has no model (call hipets_set_model)");
    if (check_cem(p)) return 1;
    if (!x0 || !lower || !upper || !s0 || !out) return fail("null argument");
    if (p->act_dim != e->md.act_dim) return fail("act_dim %d != model act_dim %d", p->act_dim, e->md.act_dim);
    if (n_env < 1 || n_env > 4096) return fail("n_env %d outside [1, 4096]", n_env);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    HCHECK(hipSetDevice(e->device));
    ENTER_STREAM(e, st);
    const CemDev c = make_cem(p, n_env);
    const size_t nd = (size_t)n_env * c.D, npop = (size_t)n_env * c.pop;
    if (e->mu.ensure(nd * 4) || e->disp.ensure(nd * 4) || e->best_solution.ensure(nd * 4) || e->best_value.ensure((size_t)n_env * 4 + 16) ||
        e->population.ensure(npop * c.D * 4) || e->values.ensure(npop * 4))
        return 1;
    hipLaunchKernelGGL(cem_init_kernel, dim3((unsigned)((nd + 255) / 256)), dim3(256), 0, st, c, x0, lower, upper, e->mu.as<float>(),
                       e->disp.as<float>(), e->best_value.as<float>());
    HCHECK(hipGetLastError());
    HCHECK(hipMemsetAsync(e->best_solution.p, 0, nd * 4, st));
    hipets_rollout_opts ro{};
    ro.mode = e->plan_mode;
    ro.seed = seed;
    ro.n_env = n_env;
    int n2 = 1;
    while (n2 < c.pop) n2 <<= 1;
    if (plan_prologue(e, s0, n_env, c.H, p->num_iterations, seed, plan_id * (uint64_t)p->num_iterations, st)) return 1;
    for (int i = 0; i < p->num_iterations; ++i) {
        const uint64_t sid = plan_id * (uint64_t)p->num_iterations + (uint64_t)i;
        const long long n = (long long)npop * c.D;
        hipLaunchKernelGGL(cem_sample_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, c, e->mu.as<float>(), e->disp.as<float>(),
                           lower, upper, (const float*)nullptr, (unsigned long long)seed, (unsigned long long)sid,
                           e->population.as<float>());
        HCHECK(hipGetLastError());
        ro.stream_id = sid;
        // the particle mean of the returns (model_env.py:190-191) happens inside the refit kernel: one launch less per iteration
        if (rollout_impl(e, e->population.as<float>(), nullptr, (int32_t)npop, c.H, P, &ro, nullptr, stream)) return 1;
        int* eidx = (e->has_trace && e->trace.elite_idx) ? e->trace.elite_idx + (size_t)i * n_env * c.K : nullptr;
        CemDev cr = c;
        cr.totals = e->totals.as<float>();
        cr.P = P;
        hipLaunchKernelGGL(cem_refit_kernel, dim3(refit_blocks(cr.D), n_env), dim3(kRefitThreads), (size_t)n2 * 8 + kRefitScratchBytes, st, cr, e->values.as<float>(),
                           e->population.as<float>(), e->mu.as<float>(), e->disp.as<float>(), e->best_value.as<float>(),
                           e->best_solution.as<float>(), eidx);
        HCHECK(hipGetLastError());
        if (trace_iter(e, i, (int)npop, (size_t)c.D, e->population.as<float>(), e->values.as<float>(), e->mu.as<float>(), e->disp.as<float>(), st, n_env))
            return 1;
    }
    HCHECK(hipMemcpyAsync(out, p->return_mean_elites ? e->mu.p : e->best_solution.p, nd * 4, hipMemcpyDeviceToDevice, st));
    return 0;
}

int hipets_plan_mppi(hipets_engine* e, int32_t pop, int32_t H, int32_t A, int32_t num_iterations, double gamma, double beta,
                     float* mean, const float* lower, const float* upper, const float* s0, int32_t P, uint64_t seed,
                     uint64_t plan_id, void* stream) {
    return hipets_plan_mppi_batched(e, pop, H, A, num_iterations, gamma, beta, 1, mean, lower, upper, s0, P, seed, plan_id, stream);
}

int hipets_plan_mppi_batched(hipets_engine* e, int32_t pop, int32_t H, int32_t A, int32_t num_iterations, double gamma, double beta,
                             int32_t n_env, float* mean, const float* lower, const float* upper, const float* s0, int32_t P,
                             uint64_t seed, uint64_t plan_id, void* stream) {
    return plan_mppi_impl(e, pop, H, A, num_iterations, gamma, beta, n_env, mean, lower, upper, s0, P, seed, plan_id, stream, false);
}

int hipets_plan_mppi_sharded(hipets_engine* e, int32_t pop, int32_t H, int32_t A, int32_t num_iterations, double gamma, double beta,
                             float* mean, const float* lower, const float* upper, const float* s0, int32_t P, uint64_t seed,
                             uint64_t plan_id, void* stream) {
    if (e && !e->comm) return fail("no communicator (call hipets_comm_init)");
    return plan_mppi_impl(e, pop, H, A, num_iterations, gamma, beta, 1, mean, lower, upper, s0, P, seed, plan_id, stream, true);
}

}  // extern "C"

namespace {
int plan_mppi_impl(hipets_engine* e, int32_t pop, int32_t H, int32_t A, int32_t num_iterations, double gamma, double beta, int32_t n_env,
                   float* mean, const float* lower, const float* upper, const float* s0, int32_t P, uint64_t seed, uint64_t plan_id,
                   void* stream, const bool sharded) {
    if (!e || !e->has_model) return fail("engine has no model (call hipets_set_model)");
    if (!mean || !lower || !upper || !s0) return fail("null argument");
    if (pop < 1 || pop > 12000) return fail("population_size %d outside [1, 12000]", pop);
    if (H < 1 || num_iterations < 0) return fail("bad horizon/num_iterations");
    if (A != e->md.act_dim) return fail("act_dim %d != model act_dim %d", A, e->md.act_dim);
    if (n_env < 1 || n_env > 4096) return fail("n_env %d outside [1, 4096]", n_env);
    if (sharded && check_shards(e, pop, P)) return 1;  // identical on every rank, before any collective
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    HCHECK(hipSetDevice(e->device));
    ENTER_STREAM(e, st);
    const size_t nd = (size_t)H * A, npop = (size_t)n_env * pop;
    const int world = sharded ? e->comm_world : 1, rank = sharded ? e->comm_rank : 0;
    const int width = (pop + world - 1) / world;
    if (e->mu.ensure(n_env * nd * 4) || e->past_action.ensure((size_t)n_env * A * 4) || e->population.ensure(npop * nd * 4) ||
        e->values.ensure(npop * 4))
        return 1;
    if (sharded && (e->shard_values.ensure((size_t)width * 4) || e->gathered.ensure((size_t)world * width * 4))) return 1;
    LocalErr le;  // (only a sharded plan carries on after a local failure: its peers wait in the collectives)
    hipets_rollout_opts ro{};
    ro.mode = e->plan_mode;
    ro.seed = seed + (uint64_t)rank * 0x9E3779B97F4A7C15ull;  // ranks draw independent rollout randomness (rank 0: as hipets_plan_mppi)
    ro.n_env = n_env;

    auto prologue = [&]() -> int {
        if (sharded) HCHECK(hipMemsetAsync(e->shard_values.p, 0, (size_t)width * 4, st));  // padding slot of the shorter shards
        HCHECK(hipMemcpyAsync(e->mu.p, mean, n_env * nd * 4, hipMemcpyDeviceToDevice, st));
        hipLaunchKernelGGL(mppi_shift_kernel, dim3((unsigned)((n_env * nd + 255) / 256)), dim3(256), 0, st, n_env, H, A, e->mu.as<float>(), mean,
                           e->past_action.as<float>());
        HCHECK(hipGetLastError());
        return plan_prologue(e, s0, n_env, H, num_iterations, ro.seed, plan_id * (uint64_t)num_iterations, st);
    };
    le.note(prologue());
    if (!sharded && !le.ok()) return le.report();
    for (int k = 0; k < num_iterations; ++k) {
        const uint64_t sid = plan_id * (uint64_t)num_iterations + (uint64_t)k;
        auto sample = [&]() -> int {
            return launch_mppi_sample(n_env, pop, H, A, (float)beta, mean, e->past_action.as<float>(), lower, upper, nullptr, seed, sid,
                                      e->population.as<float>(), st);  // sharded: identical on every rank (same seed, same counters)
        };
        auto update = [&]() -> int {
            if (launch_mppi_update(e, n_env, pop, (int)nd, (float)gamma, e->values.as<float>(), e->population.as<float>(), mean, st)) return 1;
            return trace_iter(e, k, (int)npop, nd, e->population.as<float>(), e->values.as<float>(), mean, nullptr, st, n_env);
        };
        if (le.ok()) le.note(sample());
        ro.stream_id = sid;
        if (sharded) {
            if (sharded_evaluate(e, e->population.as<float>(), pop, H, P, &ro, stream, &le)) return 1;
        } else if (le.ok()) {
            le.note(rollout_impl(e, e->population.as<float>(), nullptr, (int32_t)npop, H, P, &ro, e->values.as<float>(), stream));
        }
        if (le.ok()) le.note(update());
        if (!sharded && !le.ok()) break;
    }
    return le.report();
}
}  // namespace

extern "C" {

int hipets_plan_icem(hipets_engine* e, const hipets_icem_params* p, const float* x0, const float* lower, const float* upper,
                     float* elite, int32_t has_elite, const int32_t* keep_idx, const float* s0, int32_t P, uint64_t seed,
                     uint64_t plan_id, float* out, void* stream) {
    return hipets_plan_icem_batched(e, p, 1, x0, lower, upper, elite, has_elite, keep_idx, s0, P, seed, plan_id, out, stream);
}

int hipets_plan_icem_batched(hipets_engine* e, const hipets_icem_params* p, int32_t n_env, const float* x0, const float* lower,
                             const float* upper, float* elite, int32_t has_elite, const int32_t* keep_idx, const float* s0, int32_t P,
                             uint64_t seed, uint64_t plan_id, float* out, void* stream) {
    return plan_icem_impl(e, p, n_env, x0, lower, upper, elite, has_elite, keep_idx, s0, P, seed, plan_id, out, stream, false);
}

int hipets_plan_icem_sharded(hipets_engine* e, const hipets_icem_params* p, const float* x0, const float* lower, const float* upper,
                             float* elite, int32_t has_elite, const int32_t* keep_idx, const float* s0, int32_t P, uint64_t seed,
                             uint64_t plan_id, float* out, void* stream) {
    if (e && !e->comm) return fail("no communicator (call hipets_comm_init)");
    return plan_icem_impl(e, p, 1, x0, lower, upper, elite, has_elite, keep_idx, s0, P, seed, plan_id, out, stream, true);
}

}  // extern "C"

namespace {
int plan_icem_impl(hipets_engine* e, const hipets_icem_params* p, int32_t n_env, const float* x0, const float* lower, const float* upper,
                   float* elite, int32_t has_elite, const int32_t* keep_idx, const float* s0, int32_t P, uint64_t seed, uint64_t plan_id,
                   float* out, void* stream, const bool sharded) {
    if (!e || !e->has_model) return fail("engine has no model (call hipets_set_model)");
    if (!p || !x0 || !lower || !upper || !elite || !s0 || !out) return fail("null argument");
    if (p->act_dim != e->md.act_dim) return fail("act_dim %d != model act_dim %d", p->act_dim, e->md.act_dim);
    if (p->horizon < 2 || p->horizon > kMaxHorizon) return fail("iCEM horizon %d outside [2, %d]", p->horizon, kMaxHorizon);
    const int K = p->elite_num, keep = p->keep_elite_size, iters = p->num_iterations, H = p->horizon, A = p->act_dim;
    if (K < 1 || keep < 0 || keep > K) return fail("elite_num %d / keep_elite_size %d invalid", K, keep);
    if (p->population_size < 1 || iters < 0 || !(p->population_decay_factor > 0.0)) return fail("bad iCEM parameters");
    if (n_env < 1 || n_env > 4096) return fail("n_env %d outside [1, 4096]", n_env);
    // population sizes (:419-431) and the rows every iteration evaluates are known up front: size the workspace for the largest, and
    // (sharded) refuse on EVERY rank, before the first collective, what one rank's shard of some iteration could not take
    std::vector<int> sizes(iters), rows_of(iters);
    int max_rows = 1;
    for (int i = 0, he = has_elite; i < iters; ++i, he = 1) {
        int n = (int)std::ceil(std::fmax((double)p->population_size * std::pow(p->population_decay_factor, -(double)i), 2.0 * K));
        const int m = p->population_size_module;
        if (m > 0 && n % m) n += m - n % m;
        sizes[i] = n;
        if (n + keep > kMaxPop) return fail("iCEM iteration %d evaluates %d candidates (max %d)", i, n + keep, kMaxPop);
        rows_of[i] = n + (he ? ((i == iters - 1 && i != 0) ? 1 : keep) : 0);
        if (K > rows_of[i]) return fail("elite_num %d invalid", K);
        max_rows = std::max(max_rows, n + keep);
        if (sharded && check_shards(e, rows_of[i], P)) return 1;
    }
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    HCHECK(hipSetDevice(e->device));
    ENTER_STREAM(e, st);
    const size_t nd = (size_t)H * A, ne = (size_t)n_env;
    const int world = sharded ? e->comm_world : 1, rank = sharded ? e->comm_rank : 0;
    const int max_width = (max_rows + world - 1) / world;
    if (e->mu.ensure(ne * nd * 4) || e->disp.ensure(ne * nd * 4) || e->best_solution.ensure(ne * nd * 4) || e->best_value.ensure(ne * 4 + 16) ||
        e->population.ensure(ne * max_rows * nd * 4) || e->values.ensure(ne * max_rows * 4) ||
        e->kept.ensure(ne * std::max(keep, 1) * nd * 4) || e->elite_idx.ensure(ne * K * 4) || e->keep_idx.ensure(ne * std::max(keep, 1) * 4) ||
        e->s0.ensure(ne * e->md.obs_dim * 4))
        return 1;
    if (sharded && (e->shard_values.ensure((size_t)max_width * 4) || e->gathered.ensure((size_t)world * max_width * 4))) return 1;
    hipets_cem_params cp{};
    cp.population_size = std::max(K, 1);
    cp.horizon = H;
    cp.act_dim = A;
    cp.num_iterations = iters;
    cp.elite_num = K;
    cp.alpha = p->alpha;
    cp.return_mean_elites = p->return_mean_elites;
    cp.clipped_normal = 0;  // initial variance ((ub - lb)^2) / 16 (:373) and variance (not std) refit
    cp.unbiased_var = 0;    // :479
    LocalErr le;  // (only a sharded plan carries on after a local failure: its peers wait in the collectives)
    hipets_rollout_opts ro{};
    ro.mode = e->plan_mode;
    ro.seed = seed + (uint64_t)rank * 0x9E3779B97F4A7C15ull;  // ranks draw independent rollout randomness (rank 0: as hipets_plan_icem)
    ro.n_env = n_env;
    float* popbuf = e->population.as<float>();  // [n_env][rows][H][A], rows = this iteration's candidates per environment
    auto prologue = [&]() -> int {
        if (sharded) HCHECK(hipMemsetAsync(e->shard_values.p, 0, (size_t)max_width * 4, st));  // padding slot of the shorter shards
        hipLaunchKernelGGL(cem_init_kernel, dim3((unsigned)((ne * nd + 255) / 256)), dim3(256), 0, st, make_cem(&cp, n_env), x0, lower, upper,
                           e->mu.as<float>(), e->disp.as<float>(), e->best_value.as<float>());
        HCHECK(hipGetLastError());
        HCHECK(hipMemsetAsync(e->best_solution.p, 0, ne * nd * 4, st));
        return stage_h2d(e, e->s0.p, s0, ne * e->md.obs_dim * 4, st);  // the observations are the same for every iteration: staged once
    };
    le.note(prologue());
    if (!sharded && !le.ok()) return le.report();
    for (int i = 0; i < iters; ++i) {
        const int n = sizes[i], rows = rows_of[i], extra = rows - n;
        const uint64_t sid = (plan_id * (uint64_t)iters + (uint64_t)i) * 4;
        auto sample = [&]() -> int {  // identical on every rank of a sharded plan: same seed, same counters
            launch_icem_sample(st, n_env, rows, n, H, A, (float)p->colored_noise_exponent, e->mu.as<float>(), e->disp.as<float>(), lower, upper,
                               (const float*)nullptr, (unsigned long long)seed, (unsigned long long)sid, popbuf);
            HCHECK(hipGetLastError());
            if (!extra) return 0;
            float* tail = popbuf + (size_t)n * nd;  // environment 0's extra rows; the others follow rows * nd floats apart
            if (i == iters - 1 && i != 0) {  // :463-464
                hipLaunchKernelGGL(icem_append_mu_kernel, dim3((unsigned)((ne * nd + 255) / 256)), dim3(256), 0, st, n_env, rows, n, (int)nd,
                                   e->mu.as<float>(), popbuf);
                HCHECK(hipGetLastError());
                return 0;
            }
            const int32_t* kidx = keep_idx ? keep_idx + (size_t)i * n_env * keep : e->keep_idx.as<int32_t>();
            if (!keep_idx) {
                hipLaunchKernelGGL(icem_keep_select_kernel, dim3(n_env), dim3(256), (size_t)K * 8, st, K, keep, (unsigned long long)seed,
                                   (unsigned long long)(sid + 2), e->keep_idx.as<int32_t>());
                HCHECK(hipGetLastError());
            }
            const long long ng = (long long)keep * nd;
            if (i == 0) {  // :450-462: kept elites shifted one step with a fresh tail action
                hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((ng + 255) / 256), n_env), dim3(256), 0, st, keep, (int)nd, elite, kidx,
                                   e->kept.as<float>(), (long long)K * nd, (long long)keep * nd);
                HCHECK(hipGetLastError());
                hipLaunchKernelGGL(icem_shift_kernel, dim3((unsigned)((ne * ng + 255) / 256)), dim3(256), 0, st, n_env, rows, keep, H, A,
                                   e->kept.as<float>(), e->mu.as<float>(), e->disp.as<float>(), (const float*)nullptr,
                                   (unsigned long long)seed, (unsigned long long)(sid + 1), tail);
                HCHECK(hipGetLastError());
            } else {  // :465-466
                hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((ng + 255) / 256), n_env), dim3(256), 0, st, keep, (int)nd, elite, kidx,
                                   tail, (long long)K * nd, (long long)rows * nd);
                HCHECK(hipGetLastError());
            }
            return 0;
        };
        auto refit = [&]() -> int {
            cp.population_size = rows;
            if (check_cem(&cp)) return 1;
            int n2 = 1;
            while (n2 < rows) n2 <<= 1;
            CemDev cr = make_cem(&cp, n_env);
            if (!sharded) {  // the rollouts left their per-row totals: the particle means are formed inside the refit kernel (same sum, same bits)
                cr.totals = e->totals.as<float>();
                cr.P = P;
            }
            hipLaunchKernelGGL(cem_refit_kernel, dim3(refit_blocks((int)nd), n_env), dim3(kRefitThreads), (size_t)n2 * 8 + kRefitScratchBytes, st, cr,
                               e->values.as<float>(), popbuf, e->mu.as<float>(), e->disp.as<float>(), e->best_value.as<float>(),
                               e->best_solution.as<float>(), e->elite_idx.as<int>());
            HCHECK(hipGetLastError());
            const long long nk = (long long)K * nd;  // self.elite = population[elite_idx] (:476), per environment
            hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((nk + 255) / 256), n_env), dim3(256), 0, st, K, (int)nd, popbuf, e->elite_idx.as<int32_t>(),
                               elite, (long long)rows * nd, (long long)K * nd);
            HCHECK(hipGetLastError());
            if (trace_iter(e, i, n_env * rows, nd, popbuf, e->values.as<float>(), e->mu.as<float>(), e->disp.as<float>(), st, n_env)) return 1;
            if (e->has_trace && e->trace.elite_idx)
                HCHECK(hipMemcpyAsync(e->trace.elite_idx + (size_t)i * n_env * K, e->elite_idx.p, ne * K * 4, hipMemcpyDeviceToDevice, st));
            return 0;
        };
        if (le.ok()) le.note(sample());
        ro.stream_id = sid + 3;
        if (sharded) {
            if (sharded_evaluate(e, popbuf, rows, H, P, &ro, stream, &le)) return 1;
        } else if (le.ok()) {
            le.note(rollout_impl(e, popbuf, nullptr, n_env * rows, H, P, &ro, nullptr, stream));  // s0 staged above; returns: refit (CemDev::totals)
        }
        if (le.ok()) le.note(refit());
        if (!sharded && !le.ok()) break;
    }
    if (!le.ok()) return le.report();
    HCHECK(hipMemcpyAsync(out, p->return_mean_elites ? e->mu.p : e->best_solution.p, ne * nd * 4, hipMemcpyDeviceToDevice, st));
    return 0;
}
}  // namespace

extern "C" {

int hipets_planet_set_model(hipets_engine* e, const hipets_planet_desc* d, void* stream) {
    if (!e || !d) return fail("null argument");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    HCHECK(hipSetDevice(e->device));
    ENTER_STREAM(e, st);
    if (d->latent_size < 1 || d->action_size < 1 || d->belief_size < 1 || d->hidden_size < 1) return fail("bad PlaNet dimensions");
    const void* ptrs[] = {d->w_embed, d->b_embed, d->w_ih, d->b_ih, d->w_hh, d->b_hh, d->w_prior1, d->b_prior1,
                          d->w_prior2, d->b_prior2, d->w_rew1, d->b_rew1, d->w_rew2, d->b_rew2, d->w_rew3, d->b_rew3};
    for (const void* p : ptrs)
        if (!p) return fail("null PlaNet tensor");
    const int L = d->latent_size, A = d->action_size, Hb = d->belief_size, F = d->hidden_size;
    auto up16 = [](int x) { return (x + 15) / 16 * 16; };
    PlanetDev pd{};
    pd.latent = L; pd.action = A; pd.belief = Hb; pd.hidden = F; pd.min_std = d->min_std;
    pd.widA = up16(L + A);
    pd.widE = up16(Hb + L);
    const int widB = up16(std::max(Hb, F)), widC = up16(std::max(3 * Hb, F)), widD = up16(std::max(std::max(3 * Hb, 2 * L), 16));
    pd.segA = 0; pd.segB = pd.widA; pd.segC = pd.segB + widB; pd.segD = pd.segC + widC; pd.segE = pd.segD + widD;
    int ld = pd.segE + pd.widE;
    while (ld % 64 != 8) ld += 4;  // conflict-free ds_read_b128 A fragments (see rollout.hpp)
    pd.ld = ld;
    if (planet_smem_bytes(ld) > e->lds_max) return fail("PlaNet model too wide for LDS (row of %d floats)", ld);
    // op table: K, N, source tensors, whether the output feeds another GEMM (chunk-transposed columns)
    struct OpSrc { int K, N; const void* w; const void* b; int permute; };
    const OpSrc ops[kPlanetOps] = {
        {L + A, Hb, d->w_embed, d->b_embed, 1},   {Hb, 3 * Hb, d->w_ih, d->b_ih, 0},     {Hb, 3 * Hb, d->w_hh, d->b_hh, 0},
        {Hb, F, d->w_prior1, d->b_prior1, 1},     {F, 2 * L, d->w_prior2, d->b_prior2, 0}, {Hb + L, F, d->w_rew1, d->b_rew1, 1},
        {F, F, d->w_rew2, d->b_rew2, 1},          {F, 1, d->w_rew3, d->b_rew3, 0}};
    long long woff = 0;
    int boff = 0;
    // execution order: embed, hidden gates, input gates, prior x2, reward head x3 (ops[] above is in tensor order)
    PlanetOp table[kPlanetOps];
    const int order[kPlanetOps] = {PL_EMBED, PL_GH, PL_GI, PL_PRIOR1, PL_PRIOR2, PL_REW1, PL_REW2, PL_REW3};
    const int in_off[kPlanetOps] = {pd.segA, pd.segE, pd.segB, pd.segE, pd.segB, pd.segE, pd.segB, pd.segC};
    const int out_off[kPlanetOps] = {pd.segB, pd.segD, pd.segC, pd.segB, pd.segD, pd.segB, pd.segC, pd.segD};
    const int relu[kPlanetOps] = {1, 0, 0, 1, 0, 1, 1, 0};
    const int post[kPlanetOps] = {PL_POST_NONE, PL_POST_SYNC, PL_POST_GRU, PL_POST_SYNC, PL_POST_SAMPLE, PL_POST_SYNC, PL_POST_SYNC,
                                  PL_POST_REWARD};
    for (int x = 0; x < kPlanetOps; ++x) {
        const int i = order[x];
        table[x].in_off = in_off[x];
        table[x].out_off = out_off[x];
        table[x].relu = relu[x];
        table[x].post = post[x];
        LayerMeta& lm = table[x].lm;
        lm.Kp = up16(ops[i].K);
        lm.Np = up16(ops[i].N);
        lm.woff = woff;
        lm.boff = boff;
        lm.tail_steps = (ops[i].K - (lm.Kp - 16) + 3) / 4;
        woff += (long long)lm.Kp * lm.Np;
        boff += lm.Np;
    }
    if (e->planet_w.ensure((size_t)woff * 4) || e->planet_b.ensure((size_t)boff * 4) || e->planet_member.ensure(16)) return 1;
    int* zero_member = e->planet_member.as<int>();  // the pack kernels index "member 0" of a one-member set
    HCHECK(hipMemsetAsync(zero_member, 0, 4, st));
    if (e->planet_ops.ensure(sizeof(table))) return 1;
    HCHECK(hipMemcpyAsync(e->planet_ops.p, table, sizeof(table), hipMemcpyHostToDevice, st));
    for (int x = 0; x < kPlanetOps; ++x) {
        const int i = order[x];
        const LayerMeta& lm = table[x].lm;
        const long long n = (long long)lm.Kp * lm.Np;
        hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, e->planet_w.as<float>(),
                           reinterpret_cast<const float*>(ops[i].w), zero_member, 1, ops[i].K, ops[i].N, lm.Kp, lm.Np, woff, lm.woff,
                           ops[i].permute, 1);
        HCHECK(hipGetLastError());
        hipLaunchKernelGGL(pack_bias_kernel, dim3((lm.Np + 255) / 256), dim3(256), 0, st, e->planet_b.as<float>(),
                           reinterpret_cast<const float*>(ops[i].b), zero_member, 1, ops[i].N, lm.Np, boff, lm.boff, ops[i].permute);
        HCHECK(hipGetLastError());
    }
    HCHECK(hipStreamSynchronize(st));  // the caller's tensors may go away after return
    pd.w = e->planet_w.as<float>();
    pd.b = e->planet_b.as<float>();
    pd.ops = e->planet_ops.as<PlanetOp>();
    e->pd = pd;
    e->planet_static = planet_static_shape(pd, table);
    e->has_planet = true;
    return 0;
}

}  // extern "C"

namespace {
// hipets_planet_rollout; returns == nullptr: the caller reduces e->totals over the particles itself (hipets_plan_planet_cem: inside the
// refit kernel, CemDev::totals)
int planet_rollout_impl(hipets_engine* e, const float* actions, const float* latent0, const float* belief0, int32_t pop, int32_t H,
                        int32_t P, const hipets_planet_opts* o, float* returns, hipStream_t st) {
    const long long B = (long long)pop * P;
    if (B > 0x7FFFFFFF / std::max(e->pd.belief, 16)) return fail("batch too large");
    if (e->totals.ensure((size_t)B * 4)) return 1;
    PlanetArgs ra{};
    ra.pop = pop; ra.P = P; ra.H = H; ra.B = (int)B;
    ra.actions = actions;
    ra.latent0 = latent0;
    ra.belief0 = belief0;
    ra.totals = e->totals.as<float>();
    ra.eps = o->eps;
    ra.use_philox = (o->eps || o->no_sample) ? 0 : 1;
    ra.seed = o->seed;
    ra.stream_id = o->stream_id;
    ra.trace_latent = o->trace_latent;
    ra.trace_belief = o->trace_belief;
    ra.trace_rewards = o->trace_rewards;
    ra.phase_cycles = reinterpret_cast<long long*>(o->phase_cycles);
    const size_t lds = planet_smem_bytes(e->pd.ld);
    const int nwg = (int)((B + kTile - 1) / kTile);
    // (HIPETS_PLANET_GENERIC=1: the run-time generic instance whatever the shapes -- tests compare the two bit for bit)
    const char* pg = std::getenv("HIPETS_PLANET_GENERIC");
    HCHECK(launch_planet_rollout(nwg, (unsigned)lds, (int)e->lds_max, e->pd, ra, st, e->planet_static && !(pg && pg[0] == '1')));
    if (!returns) return 0;
    hipLaunchKernelGGL(particle_mean_kernel, dim3((pop + 255) / 256), dim3(256), 0, st, e->totals.as<float>(), returns, pop, P);
    HCHECK(hipGetLastError());
    return 0;
}
}  // namespace

extern "C" {

int hipets_planet_rollout(hipets_engine* e, const float* actions, const float* latent0, const float* belief0, int32_t pop, int32_t H,
                          int32_t P, const hipets_planet_opts* o, float* returns, void* stream) {
    if (!e || !e->has_planet) return fail("engine has no PlaNet model (call hipets_planet_set_model)");
    if (!actions || !latent0 || !belief0 || !o || !returns) return fail("null argument");
    if (pop < 1 || H < 1 || P < 1) return fail("bad pop/horizon/particles");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    HCHECK(hipSetDevice(e->device));
    ENTER_STREAM(e, st);
    return planet_rollout_impl(e, actions, latent0, belief0, pop, H, P, o, returns, st);
}

int hipets_plan_planet_cem(hipets_engine* e, const hipets_cem_params* p, const float* x0, const float* lower, const float* upper,
                           const float* latent0, const float* belief0, int32_t P, uint64_t seed, uint64_t plan_id, float* out,
                           void* stream) {
    if (!e || !e->has_planet) return fail("engine has no PlaNet model (call hipets_planet_set_model)");
    if (check_cem(p)) return 1;
    if (!x0 || !lower || !upper || !latent0 || !belief0 || !out) return fail("null argument");
    if (p->act_dim != e->pd.action) return fail("act_dim %d != model action_size %d", p->act_dim, e->pd.action);
    if (P < 1) return fail("bad pop/horizon/particles");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    HCHECK(hipSetDevice(e->device));
    ENTER_STREAM(e, st);
    const CemDev c = make_cem(p, 1);
    const size_t nd = (size_t)c.D;
    if (e->mu.ensure(nd * 4) || e->disp.ensure(nd * 4) || e->best_solution.ensure(nd * 4) || e->best_value.ensure(16) ||
        e->population.ensure((size_t)c.pop * nd * 4) || e->values.ensure((size_t)c.pop * 4))
        return 1;
    hipLaunchKernelGGL(cem_init_kernel, dim3((unsigned)((nd + 255) / 256)), dim3(256), 0, st, c, x0, lower, upper, e->mu.as<float>(),
                       e->disp.as<float>(), e->best_value.as<float>());
    HCHECK(hipGetLastError());
    HCHECK(hipMemsetAsync(e->best_solution.p, 0, nd * 4, st));
    hipets_planet_opts po{};
    po.seed = seed;
    int n2 = 1;
    while (n2 < c.pop) n2 <<= 1;
    for (int i = 0; i < p->num_iterations; ++i) {
        const uint64_t sid = plan_id * (uint64_t)p->num_iterations + (uint64_t)i;
        const long long n = (long long)c.pop * c.D;
        hipLaunchKernelGGL(cem_sample_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, c, e->mu.as<float>(), e->disp.as<float>(),
                           lower, upper, (const float*)nullptr, (unsigned long long)seed, (unsigned long long)sid, e->population.as<float>());
        HCHECK(hipGetLastError());
        po.stream_id = sid;
        // (the particle means of the returns are formed inside the refit kernel -- the same sequential sum and division as
        // particle_mean_kernel, so the same bits -- instead of a launch of its own in between: round 6)
        if (planet_rollout_impl(e, e->population.as<float>(), latent0, belief0, c.pop, c.H, P, &po, nullptr, st)) return 1;
        int* eidx = (e->has_trace && e->trace.elite_idx) ? e->trace.elite_idx + (size_t)i * c.K : nullptr;
        CemDev cr = c;
        cr.totals = e->totals.as<float>();
        cr.P = P;
        hipLaunchKernelGGL(cem_refit_kernel, dim3(refit_blocks(c.D), 1), dim3(kRefitThreads), (size_t)n2 * 8 + kRefitScratchBytes, st, cr, e->values.as<float>(),
                           e->population.as<float>(), e->mu.as<float>(), e->disp.as<float>(), e->best_value.as<float>(),
                           e->best_solution.as<float>(), eidx);
        HCHECK(hipGetLastError());
        if (trace_iter(e, i, c.pop, nd, e->population.as<float>(), e->values.as<float>(), e->mu.as<float>(), e->disp.as<float>(), st)) return 1;
    }
    HCHECK(hipMemcpyAsync(out, p->return_mean_elites ? e->mu.p : e->best_solution.p, nd * 4, hipMemcpyDeviceToDevice, st));
    return 0;
}

int hipets_comm_unique_id(void* id_out) {
    if (!id_out) return fail("null argument");
    if (rccl_load()) return 1;
    NCHECK(g_rccl.GetUniqueId(id_out));
    return 0;
}

int hipets_comm_init(hipets_engine* e, const void* unique_id, int32_t rank, int32_t world) {
    if (!e || !unique_id) return fail("null argument");
    if (world < 1 || rank < 0 || rank >= world) return fail("bad rank %d / world_size %d", rank, world);
    if (rccl_load()) return 1;
    HCHECK(hipSetDevice(e->device));
    if (e->comm) {
        NCHECK(g_rccl.CommDestroy(e->comm));
        e->comm = nullptr;
    }
    RcclId id;
    std::memcpy(id.internal, unique_id, HIPETS_COMM_ID_BYTES);
    NCHECK(g_rccl.CommInitRank(&e->comm, world, id, rank));
    e->comm_rank = rank;
    e->comm_world = world;
    return 0;
}

int hipets_comm_destroy(hipets_engine* e) {
    if (!e) return fail("null engine");
    if (e->comm) {
        HCHECK(hipSetDevice(e->device));
        NCHECK(g_rccl.CommDestroy(e->comm));
        e->comm = nullptr;
    }
    e->comm_rank = 0;
    e->comm_world = 1;
    return 0;
}

int hipets_comm_info(hipets_engine* e, int32_t* rank, int32_t* world_size) {
    if (!e) return fail("null engine");
    int r = e->comm_rank, w = e->comm_world;
    if (e->comm && g_rccl.CommCount && g_rccl.CommUserRank) {  // what the communicator itself says, not what the caller passed in
        NCHECK(g_rccl.CommCount(e->comm, &w));
        NCHECK(g_rccl.CommUserRank(e->comm, &r));
    }
    if (rank) *rank = r;
    if (world_size) *world_size = w;
    return 0;
}

int hipets_plan_cem_sharded(hipets_engine* e, const hipets_cem_params* p, const float* x0, const float* lower, const float* upper,
                            const float* s0, int32_t P, uint64_t seed, uint64_t plan_id, float* out, void* stream) {
    if (!e || !e->has_model) return fail("engine has no model (call hipets_set_model)");
    if (!e->comm) return fail("no communicator (call hipets_comm_init)");
    if (check_cem(p)) return 1;
    if (!x0 || !lower || !upper || !s0 || !out) return fail("null argument");
    if (p->act_dim != e->md.act_dim) return fail("act_dim %d != model act_dim %d", p->act_dim, e->md.act_dim);
    if (check_shards(e, p->population_size, P)) return 1;  // identical on every rank, before any collective
    const int world = e->comm_world, rank = e->comm_rank;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    HCHECK(hipSetDevice(e->device));
    ENTER_STREAM(e, st);
    const CemDev c = make_cem(p, 1);
    const int width = (c.pop + world - 1) / world;
    const size_t nd = (size_t)c.D;
    if (e->mu.ensure(nd * 4) || e->disp.ensure(nd * 4) || e->best_solution.ensure(nd * 4) || e->best_value.ensure(16) ||
        e->population.ensure((size_t)c.pop * nd * 4) || e->values.ensure((size_t)c.pop * 4) || e->shard_values.ensure((size_t)width * 4) ||
        e->gathered.ensure((size_t)world * width * 4))
        return 1;
    LocalErr le;
    hipets_rollout_opts ro{};
    ro.mode = e->plan_mode;
    ro.seed = seed + (uint64_t)rank * 0x9E3779B97F4A7C15ull;  // ranks draw independent rollout randomness (rank 0: as hipets_plan_cem)
    int n2 = 1;
    while (n2 < c.pop) n2 <<= 1;
    auto prologue = [&]() -> int {
        hipLaunchKernelGGL(cem_init_kernel, dim3((unsigned)((nd + 255) / 256)), dim3(256), 0, st, c, x0, lower, upper, e->mu.as<float>(),
                           e->disp.as<float>(), e->best_value.as<float>());
        HCHECK(hipGetLastError());
        HCHECK(hipMemsetAsync(e->best_solution.p, 0, nd * 4, st));
        HCHECK(hipMemsetAsync(e->shard_values.p, 0, (size_t)width * 4, st));  // padding slot of the shorter shards
        return plan_prologue(e, s0, 1, c.H, p->num_iterations, ro.seed, plan_id * (uint64_t)p->num_iterations, st);
    };
    le.note(prologue());
    for (int i = 0; i < p->num_iterations; ++i) {
        const uint64_t sid = plan_id * (uint64_t)p->num_iterations + (uint64_t)i;
        auto sample = [&]() -> int {
            const long long n = (long long)c.pop * c.D;
            hipLaunchKernelGGL(cem_sample_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, c, e->mu.as<float>(), e->disp.as<float>(),
                               lower, upper, (const float*)nullptr, (unsigned long long)seed, (unsigned long long)sid,
                               e->population.as<float>());  // identical on every rank: same seed, same counters
            HCHECK(hipGetLastError());
            return 0;
        };
        auto refit = [&]() -> int {
            int* eidx = (e->has_trace && e->trace.elite_idx) ? e->trace.elite_idx + (size_t)i * c.K : nullptr;
            hipLaunchKernelGGL(cem_refit_kernel, dim3(refit_blocks(c.D), 1), dim3(kRefitThreads), (size_t)n2 * 8 + kRefitScratchBytes, st, c, e->values.as<float>(),
                               e->population.as<float>(), e->mu.as<float>(), e->disp.as<float>(), e->best_value.as<float>(),
                               e->best_solution.as<float>(), eidx);
            HCHECK(hipGetLastError());
            return trace_iter(e, i, c.pop, nd, e->population.as<float>(), e->values.as<float>(), e->mu.as<float>(), e->disp.as<float>(), st);
        };
        if (le.ok()) le.note(sample());
        ro.stream_id = sid;
        if (sharded_evaluate(e, e->population.as<float>(), c.pop, c.H, P, &ro, stream, &le)) return 1;
        if (le.ok()) le.note(refit());
    }
    if (!le.ok()) return le.report();
    HCHECK(hipMemcpyAsync(out, p->return_mean_elites ? e->mu.p : e->best_solution.p, nd * 4, hipMemcpyDeviceToDevice, st));
    return 0;
}

int hipets_timing_enable(hipets_engine* e, int32_t on) {
    if (!e) return fail("null engine");
    e->timing = on != 0;
    e->timing_stride = on > 1 ? on : 1;
    e->launch_counter = 0;
    return 0;
}

int hipets_timing_read(hipets_engine* e, int64_t* launches, double* total_ms, int32_t reset) {
    if (!e) return fail("null engine");
    HCHECK(hipSetDevice(e->device));
    double tot = 0.0;
    for (auto& ev : e->events) {
        HCHECK(hipEventSynchronize(ev.second));
        float ms = 0.f;
        HCHECK(hipEventElapsedTime(&ms, ev.first, ev.second));
        tot += ms;
    }
    if (launches) *launches = (int64_t)e->events.size();
    if (total_ms) *total_ms = tot;
    if (reset) {
        for (auto& ev : e->events) e->event_pool.push_back(ev);
        e->events.clear();
    }
    return 0;
}

}  // extern "C"
