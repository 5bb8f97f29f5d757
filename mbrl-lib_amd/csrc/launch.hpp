// launch.hpp -- host entry points of the kernels that live in their own translation units (rollout_r<R>.hip, planet.hip):
// the rollout kernel is instantiated per row-tile count R and activation, and the instantiations compile in parallel.
#pragma once
#include <hip/hip_runtime.h>

#include "planet_types.hpp"
#include "rollout.hpp"

namespace hipets {

// Launch rollout_kernel<R, ACT> (ACT = md.activation where a specialised instance exists, else the run-time generic one)
// with `grid` workgroups and `lds` bytes of dynamic LDS on `st`.  start / stop (both or neither) ride on the dispatch packet.
hipError_t launch_rollout_r1(int grid, unsigned lds, int lds_max, const ModelDev& md, const RolloutArgs& ra, hipStream_t st, hipEvent_t start, hipEvent_t stop);
hipError_t launch_rollout_r2(int grid, unsigned lds, int lds_max, const ModelDev& md, const RolloutArgs& ra, hipStream_t st, hipEvent_t start, hipEvent_t stop);
hipError_t launch_rollout_r3(int grid, unsigned lds, int lds_max, const ModelDev& md, const RolloutArgs& ra, hipStream_t st, hipEvent_t start, hipEvent_t stop);
hipError_t launch_rollout_r4(int grid, unsigned lds, int lds_max, const ModelDev& md, const RolloutArgs& ra, hipStream_t st, hipEvent_t start, hipEvent_t stop);

// Shapes whose fused fp32 instance is a KSpec::WIDE one (output layer wider than kSplMaxTiles column tiles: no LDS image of the
// outputs, narrow activation buffers): X(hidden column tiles, output column tiles, reward fn, termination fn).  Instantiated for
// R = 1 and 2 (rollout_inst.inc); the host sizes the LDS and chooses R for that layout exactly when the launcher will pick it.
#define HIPETS_WIDE_SHAPES(X) X(13, 47, HIPETS_REW_HALFCHEETAH, HIPETS_TERM_HUMANOID)

// Hidden widths with a KSpec::HID_STATIC instance (hidden layers shape-specialised, everything else generic): X(hidden column tiles).
// 13 tiles = hidden widths 193..208: the reference's default of 200 (conf/dynamics_model/gaussian_mlp_ensemble.yaml:8), which every
// configuration it ships uses.  (Rounds 4-5 also instantiated 8 and 16 tiles -- hid 113..128, 241..256: + 2-10 % over the generic
// instance, profiles/r4_hidden_widths.json -- for widths no shipped configuration or BASELINE config has; round 6 dropped those eight
// instances from the build: other widths run the generic instance, same bits.)
#define HIPETS_HID_STATIC_SHAPES(X) X(13)

// may this model / call run the hidden-static instance for `hc` hidden column tiles?  (SiLU, fp32 arithmetic, the LDS row stride the
// instance was compiled for -- i.e. no layer wider than the hidden ones; RolloutArgs::generic_only == 1 forbids it, 2 allows it)
inline bool hid_static_call(const ModelDev& md, const RolloutArgs& ra, const int hc) {
    return ra.generic_only != 1 && md.precision == HIPETS_PREC_F32 && md.activation == HIPETS_ACT_SILU && md.hidC == hc && md.ld == lean_ld(hc, hc);
}

// Shape-specialised ("lean") fp32 instances per row-tile count R: X(hidden column tiles, output column tiles, reward fn, termination
// fn, obs preprocessing); all of them SiLU, f64 normaliser, stochastic GaussianMLP with in-kernel sampling.  R follows the cost
// model's choice for the configuration (hipets.hip choose_R, which in turn knows this table: lean_shape_exists):
//   BASELINE.json: cfg1 cartpole R = 1, cfg2 / cfg3 R = 3 (a rank's shard of a strong-scaled plan: 1, 2), cfg4 R = 3 at its first
//   iteration and 2 / 4 as the iCEM population decays, cfg4' Humanoid-v4 R = 2 (small batches 1; KSpec::WIDE), cfg5 (2500 row tiles) 2;
//   the workloads the reference ships (round 4): pets_halfcheetah (conf/overrides/pets_halfcheetah.yaml: obs 18 through
//   HalfCheetahEnv.preprocess_fn, pop 400 x 20) R = 2 in DEVICE mode, 1 in FAST mode; pets_cartpole (pop 350 x 20) R = 1, 2;
//   pets_cartpole_paper_version (cartpole_pets reward + CartPoleEnv.preprocess_fn, pop 500 x 20) R = 3.
//   learned rewards + no_termination (pets_pusher 20 / 7, pets_reacher 17 / 7: pop 350 x 20; pets_mppi_halfcheetah: obs 18 through
//   preprocess_fn, MPPI 350 x 20): output layers of 3 column tiles like cfg2's, R = 2 in DEVICE mode, 1 in FAST mode.
#define HIPETS_LEAN_SHAPES_R1(X)                                                                                                               \
    X(13, 1, HIPETS_REW_CARTPOLE, HIPETS_TERM_CARTPOLE, HIPETS_OBS_NONE) X(13, 47, HIPETS_REW_HALFCHEETAH, HIPETS_TERM_HUMANOID, HIPETS_OBS_NONE) \
    X(13, 3, HIPETS_REW_HALFCHEETAH, HIPETS_TERM_NONE, HIPETS_OBS_NONE) X(13, 3, HIPETS_REW_HALFCHEETAH, HIPETS_TERM_NONE, HIPETS_OBS_HALFCHEETAH) \
    X(13, 3, HIPETS_REW_LEARNED, HIPETS_TERM_NONE, HIPETS_OBS_NONE) X(13, 3, HIPETS_REW_LEARNED, HIPETS_TERM_NONE, HIPETS_OBS_HALFCHEETAH) \
    X(13, 2, HIPETS_REW_LEARNED, HIPETS_TERM_HOPPER, HIPETS_OBS_NONE)
#define HIPETS_LEAN_SHAPES_R2(X)                                                                                                               \
    X(13, 3, HIPETS_REW_HALFCHEETAH, HIPETS_TERM_NONE, HIPETS_OBS_NONE) X(13, 47, HIPETS_REW_HALFCHEETAH, HIPETS_TERM_HUMANOID, HIPETS_OBS_NONE) \
    X(13, 6, HIPETS_REW_HALFCHEETAH, HIPETS_TERM_HUMANOID, HIPETS_OBS_NONE) X(13, 3, HIPETS_REW_HALFCHEETAH, HIPETS_TERM_NONE, HIPETS_OBS_HALFCHEETAH) \
    X(13, 1, HIPETS_REW_CARTPOLE, HIPETS_TERM_CARTPOLE, HIPETS_OBS_NONE)                                                                        \
    X(13, 3, HIPETS_REW_LEARNED, HIPETS_TERM_NONE, HIPETS_OBS_NONE) X(13, 3, HIPETS_REW_LEARNED, HIPETS_TERM_NONE, HIPETS_OBS_HALFCHEETAH) \
    X(13, 2, HIPETS_REW_LEARNED, HIPETS_TERM_HOPPER, HIPETS_OBS_NONE)
//   pets_inv_pendulum (learned reward + the inverted_pendulum termination function, obs 4 / act 1, pop 480 x 20): R = 3.
#define HIPETS_LEAN_SHAPES_R3(X)                                                                                                               \
    X(13, 3, HIPETS_REW_HALFCHEETAH, HIPETS_TERM_NONE, HIPETS_OBS_NONE) X(13, 6, HIPETS_REW_HALFCHEETAH, HIPETS_TERM_HUMANOID, HIPETS_OBS_NONE) \
    X(13, 3, HIPETS_REW_HALFCHEETAH, HIPETS_TERM_NONE, HIPETS_OBS_HALFCHEETAH) X(13, 1, HIPETS_REW_CARTPOLE_PETS, HIPETS_TERM_NONE, HIPETS_OBS_CARTPOLE_PETS) \
    X(13, 1, HIPETS_REW_LEARNED, HIPETS_TERM_INVERTED_PENDULUM, HIPETS_OBS_NONE)
#define HIPETS_LEAN_SHAPES_R4(X) X(13, 6, HIPETS_REW_HALFCHEETAH, HIPETS_TERM_HUMANOID, HIPETS_OBS_NONE)
// Shapes with FAST instances ONLY: none since round 5 (pets_hopper -- a termination function over every state dim of a model wider
// than one column tile: obs 11 / act 3, learned reward, pop 350 x 20 -- was one until its DEVICE-mode instances existed: the tables above)
#define HIPETS_LEAN_FAST_SHAPES_R1(X)
#define HIPETS_LEAN_FAST_SHAPES_R2(X)
#define HIPETS_LEAN_FAST_SHAPES_R3(X)
#define HIPETS_LEAN_FAST_SHAPES_R4(X)

// bf16x3 precision instances per R: X(hidden column tiles, output column tiles, reward fn, termination fn); no obs preprocessing
// (round 6: the R = 1 instances -- cfg1, and a rank's shard of a strong-scaled cfg2 plan, in an arithmetic mode that is reported
// separately and parked -- are gone from the build too: four instances; the mode runs R = 3 instances or raises)
#define HIPETS_B3_SHAPES_R1(X)
// (round 5: the R = 2 instances are gone -- two workgroups per CU cap a wave at 256 registers, the three-piece fragments did not fit and
// the two instances spilt 6 / 41 VGPRs to scratch, the only rollout kernels that did; the row-tile rule chooses among R = 1 and 3 for
// this arithmetic mode, hipets.hip choose_R)
#define HIPETS_B3_SHAPES_R2(X)
#define HIPETS_B3_SHAPES_R3(X) X(13, 3, HIPETS_REW_HALFCHEETAH, HIPETS_TERM_NONE) X(13, 6, HIPETS_REW_HALFCHEETAH, HIPETS_TERM_HUMANOID)
#define HIPETS_B3_SHAPES_R4(X)

// what the fused tail's reward / termination lane can see of THIS model: termination functions that test every state dim need all of them
// among the four it holds; a learned reward next to a termination function comes from another lane of the same accumulator (column
// tile 0 holds output columns 0..7)
inline bool fused_term_ok(const ModelDev& md) {
    if (md.term_fn == HIPETS_TERM_INVERTED_PENDULUM && md.obs_dim > 4) return false;
    if (md.term_fn == HIPETS_TERM_HOPPER) return md.reward_fn == HIPETS_REW_LEARNED;  // every lane judges its own dims (FAST instances, KSpec)
    if (md.reward_fn == HIPETS_REW_LEARNED && md.term_fn != HIPETS_TERM_NONE && md.obs_dim >= 8) return false;
    return true;
}

// the model-side facts every lean fp32 instance shares (the call-side ones: lean_call below)
inline bool lean_model(const ModelDev& md) {
    return md.precision == HIPETS_PREC_F32 && md.activation == HIPETS_ACT_SILU && md.normalizer == HIPETS_NORM_F64 && !md.deterministic &&
           md.propagation != HIPETS_PROP_EXPECTATION && md.lv_rows == 1 && (md.reward_fn != HIPETS_REW_LEARNED || md.learned_rewards) && fused_term_ok(md);
}

// is there a lean fp32 instance of this model's shape for R row tiles? (what the launcher of rollout_r<R>.hip will find; the cost
// model prices a (shape, R) pair with an instance lower than one that runs the hidden-static or the generic kernel)
inline bool lean_shape_exists(const ModelDev& md, const int R, const bool fast = false) {
    if (!lean_model(md)) return false;
#define HIPETS_HAS_SHAPE(HC, OC, RW, TM, OB) \
    if (md.hidC == HC && md.outC == OC && md.reward_fn == RW && md.term_fn == TM && md.obs_process == OB && md.ld == lean_ld(HC, OC)) return true;
    switch (R) {
        case 1: HIPETS_LEAN_SHAPES_R1(HIPETS_HAS_SHAPE) break;
        case 2: HIPETS_LEAN_SHAPES_R2(HIPETS_HAS_SHAPE) break;
        case 3: HIPETS_LEAN_SHAPES_R3(HIPETS_HAS_SHAPE) break;
        case 4: HIPETS_LEAN_SHAPES_R4(HIPETS_HAS_SHAPE) break;
        default: break;
    }
    if (fast) switch (R) {
        case 1: HIPETS_LEAN_FAST_SHAPES_R1(HIPETS_HAS_SHAPE) break;
        case 2: HIPETS_LEAN_FAST_SHAPES_R2(HIPETS_HAS_SHAPE) break;
        case 3: HIPETS_LEAN_FAST_SHAPES_R3(HIPETS_HAS_SHAPE) break;
        case 4: HIPETS_LEAN_FAST_SHAPES_R4(HIPETS_HAS_SHAPE) break;
        default: break;
    }
#undef HIPETS_HAS_SHAPE
    return false;
}

// ... and the same question for the bf16x3 instances (rollout_inst.inc's HIPETS_TRY_B3)
inline bool b3_shape_exists(const ModelDev& md, const int R) {
#define HIPETS_HAS_B3(HC, OC, RW, TM) \
    if (md.hidC == HC && md.outC == OC && md.reward_fn == RW && md.term_fn == TM && md.obs_process == HIPETS_OBS_NONE) return true;
    switch (R) {
        case 1: HIPETS_B3_SHAPES_R1(HIPETS_HAS_B3) break;
        case 2: HIPETS_B3_SHAPES_R2(HIPETS_HAS_B3) break;
        case 3: HIPETS_B3_SHAPES_R3(HIPETS_HAS_B3) break;
        default: break;
    }
#undef HIPETS_HAS_B3
    return false;
}

// does the call use nothing a lean instance compiled out? (KSpec in rollout.hpp lists what that is; the obs preprocessing is part
// of an instance's shape since round 4)
inline bool lean_call(const ModelDev& md, const RolloutArgs& ra) {
    return !ra.generic_only && md.activation == HIPETS_ACT_SILU && md.normalizer == HIPETS_NORM_F64 &&
           !md.deterministic && md.propagation != HIPETS_PROP_EXPECTATION && md.lv_rows == 1 && fused_term_ok(md) && !ra.eps && ra.use_philox &&
#if defined(HIPETS_STEP_TRACE) || (defined(HIPETS_LEAN_PROF) && HIPETS_LEAN_PROF)
           !ra.trace_next_obs && !ra.trace_rewards && ra.pop_env == 0 && !ra.init_states && !ra.write_back;  // the stamps go to phase_cycles
#else
           !ra.trace_next_obs && !ra.trace_rewards && !ra.phase_cycles && ra.pop_env == 0 && !ra.init_states && !ra.write_back;
#endif
}

// is the model one of the WIDE shapes (fp32 arithmetic, the row strides the instance was compiled for)?
inline bool wide_model(const ModelDev& md) {
#if HIPETS_WIDE_FUSE
    if (md.precision != HIPETS_PREC_F32 || md.activation != HIPETS_ACT_SILU || md.normalizer != HIPETS_NORM_F64 || md.obs_process != HIPETS_OBS_NONE ||
        md.deterministic || md.propagation == HIPETS_PROP_EXPECTATION || md.lv_rows != 1)
        return false;  // (the model-side conditions of lean_call)
#define HIPETS_IS_WIDE(HC, OC, RW, TM) \
    if (md.hidC == HC && md.outC == OC && md.reward_fn == RW && md.term_fn == TM && md.ld == lean_ld(HC, OC) && md.ld_in > 0) return true;
    HIPETS_WIDE_SHAPES(HIPETS_IS_WIDE)
#undef HIPETS_IS_WIDE
#endif
    return false;
}

hipError_t launch_planet_rollout(int grid, unsigned lds, int lds_max, const PlanetDev& pd, const PlanetArgs& ra, hipStream_t st, bool static_shape);

}  // namespace hipets
