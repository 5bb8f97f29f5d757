// planet.hip -- the PlaNet latent rollout kernel (planet.hpp) and its host launcher (launch.hpp).
#include <hip/hip_runtime.h>

#include "launch.hpp"
#include "planet.hpp"

namespace hipets {
#ifdef HIPETS_OPAQUE_ARGS
hipError_t launch_planet_rollout_w16(int grid, unsigned lds, int lds_max, const void* planet_dev, const void* planet_args, hipStream_t st) {
    const PlanetDev& pd = *static_cast<const PlanetDev*>(planet_dev);  // same definitions, other variant namespace: same layout
    const PlanetArgs& ra = *static_cast<const PlanetArgs*>(planet_args);
#else
inline namespace HIPETS_NS {
hipError_t launch_planet_rollout(int grid, unsigned lds, int lds_max, const PlanetDev& pd, const PlanetArgs& ra, hipStream_t st) {
#endif
    static bool attr_set[64] = {};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&planet_rollout_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
    hipLaunchKernelGGL(planet_rollout_kernel, dim3(grid), dim3(kThreads), lds, st, pd, ra);
    return hipGetLastError();
}
#ifndef HIPETS_OPAQUE_ARGS
}  // inline namespace HIPETS_NS
#endif

}  // namespace hipets
