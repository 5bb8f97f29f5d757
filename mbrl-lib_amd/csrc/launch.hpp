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

// does the call use nothing a lean instance compiled out? (KSpec in rollout.hpp lists what that is)
inline bool lean_call(const ModelDev& md, const RolloutArgs& ra) {
    return !ra.generic_only && md.activation == HIPETS_ACT_SILU && md.normalizer == HIPETS_NORM_F64 && md.obs_process == HIPETS_OBS_NONE &&
           !md.deterministic && md.propagation != HIPETS_PROP_EXPECTATION && md.lv_rows == 1 && !ra.eps && ra.use_philox &&
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

hipError_t launch_planet_rollout(int grid, unsigned lds, int lds_max, const PlanetDev& pd, const PlanetArgs& ra, hipStream_t st);

}  // namespace hipets
