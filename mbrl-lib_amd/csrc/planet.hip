// planet.hip -- the PlaNet latent rollout kernel (planet.hpp) and its host launcher (launch.hpp).
#include <hip/hip_runtime.h>

#include "launch.hpp"
#include "planet.hpp"

namespace hipets {

hipError_t launch_planet_rollout(int grid, unsigned lds, int lds_max, const PlanetDev& pd, const PlanetArgs& ra, hipStream_t st) {
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

}  // namespace hipets
