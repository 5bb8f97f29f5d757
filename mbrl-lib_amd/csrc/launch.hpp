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

hipError_t launch_planet_rollout(int grid, unsigned lds, int lds_max, const PlanetDev& pd, const PlanetArgs& ra, hipStream_t st);

}  // namespace hipets
