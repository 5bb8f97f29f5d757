// launch.hpp -- host entry points of the kernels that live in their own translation units (rollout_r<R>.hip, planet.hip, and
// the 16-wave small-batch variants rollout_r1_w16.hip / planet_w16.hip): the rollout kernel is instantiated per row-tile count R,
// per activation / shape (KSpec) and per workgroup width, and the instantiations compile in parallel.
#pragma once
#include <hip/hip_runtime.h>

#include "planet_types.hpp"
#include "rollout.hpp"

namespace hipets {
inline namespace HIPETS_NS {

// Launch rollout_kernel<R, KSpec> (the instance that matches the model and the call, rollout_inst.inc) with `grid` workgroups
// and `lds` bytes of dynamic LDS on `st`.  start / stop (both or neither) ride on the dispatch packet.
hipError_t launch_rollout_r1(int grid, unsigned lds, int lds_max, const ModelDev& md, const RolloutArgs& ra, hipStream_t st, hipEvent_t start, hipEvent_t stop);
hipError_t launch_rollout_r2(int grid, unsigned lds, int lds_max, const ModelDev& md, const RolloutArgs& ra, hipStream_t st, hipEvent_t start, hipEvent_t stop);
hipError_t launch_rollout_r3(int grid, unsigned lds, int lds_max, const ModelDev& md, const RolloutArgs& ra, hipStream_t st, hipEvent_t start, hipEvent_t stop);
hipError_t launch_rollout_r4(int grid, unsigned lds, int lds_max, const ModelDev& md, const RolloutArgs& ra, hipStream_t st, hipEvent_t start, hipEvent_t stop);

hipError_t launch_planet_rollout(int grid, unsigned lds, int lds_max, const PlanetDev& pd, const PlanetArgs& ra, hipStream_t st);

}  // inline namespace HIPETS_NS

// The 16-wave variants (one column tile per wave: the per-layer MFMA chain of a one-tile workgroup is 4x shorter; used when a
// launch has so few workgroups that CUs idle anyway).  Their code is compiled with HIPETS_WAVES=16 into hipets::w16; the
// descriptors cross the boundary as untyped pointers to the (layout-identical) structs of the caller's namespace.
hipError_t launch_rollout_r1_w16(int grid, unsigned lds, int lds_max, const void* model_dev, const void* rollout_args, hipStream_t st,
                                 hipEvent_t start, hipEvent_t stop);
hipError_t launch_planet_rollout_w16(int grid, unsigned lds, int lds_max, const void* planet_dev, const void* planet_args, hipStream_t st);

}  // namespace hipets
