// Translation unit of the SED / image Monte Carlo on cylindrical grids, k_mono (mc_mono.hip.h).  See mc_kernels.h.
#include <hip/hip_runtime.h>

#include "mc_device.hip.h"
#include "mc_mono.hip.h"
#include "mc_kernels.h"

namespace mcgpu {

const void* kpick_mono(bool l3d, bool pola, bool dark, bool scout, bool f32, bool log) {
  if (log)   // the commit pass with default-real records whose deposits go to the log (mc_mono.hip.h "The deposits as a log")
    return bsel(l3d, [&](auto L3D) { return bsel(pola, [&](auto POLA) { return bsel(dark, [&](auto DARK) -> const void* {
      return (const void*)k_mono<MCGPU_BV(L3D), MCGPU_BV(POLA), MCGPU_BV(DARK), false, true, true>;
    }); }); });
  return bsel(l3d, [&](auto L3D) { return bsel(pola, [&](auto POLA) { return bsel(dark, [&](auto DARK) { return bsel(scout, [&](auto SCOUT) {
    return bsel(f32, [&](auto F32) -> const void* {
      if constexpr (MCGPU_BV(SCOUT) && MCGPU_BV(F32)) return nullptr;   // (a scout pass deposits nothing)
      else return (const void*)k_mono<MCGPU_BV(L3D), MCGPU_BV(POLA), MCGPU_BV(DARK), MCGPU_BV(SCOUT), MCGPU_BV(F32)>;
    }); }); }); }); });
}

}  // namespace mcgpu
