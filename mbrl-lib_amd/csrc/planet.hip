// planet.hip -- the PlaNet latent rollout kernel (planet.hpp) and its host launcher (launch.hpp).
#include <hip/hip_runtime.h>

#include "launch.hpp"
#include "planet.hpp"

namespace hipets {

namespace {
template <bool STATIC>
hipError_t launch_planet(int grid, unsigned lds, int lds_max, const PlanetDev& pd, const PlanetArgs& ra, hipStream_t st) {
    static bool attr_set[64] = {};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&planet_rollout_kernel<STATIC>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
    hipLaunchKernelGGL(planet_rollout_kernel<STATIC>, dim3(grid), dim3(kThreads), lds, st, pd, ra);
    return hipGetLastError();
}
}  // namespace

// static_shape: the model has conf/dynamics_model/planet.yaml's shapes (planet_static_shape, decided once at hipets_planet_set_model)
hipError_t launch_planet_rollout(int grid, unsigned lds, int lds_max, const PlanetDev& pd, const PlanetArgs& ra, hipStream_t st, bool static_shape) {
    if (static_shape) return launch_planet<true>(grid, lds, lds_max, pd, ra, st);
    return launch_planet<false>(grid, lds, lds_max, pd, ra, st);
}

}  // namespace hipets
