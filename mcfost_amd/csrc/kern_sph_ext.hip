// Translation unit of the spherical grid's packet kernel with a dark zone and / or dust classes (k_thermal_sph_ext,
// mc_device.hip.h).  See mc_kernels.h.
#include <hip/hip_runtime.h>

#include "mc_device.hip.h"
#include "mc_kernels.h"

namespace mcgpu {

const void* kpick_thermal_sph_ext(bool l3d, bool pola, bool dark, bool lds, bool var) {
  return bsel(l3d, [&](auto L3D) { return bsel(pola, [&](auto POLA) { return bsel(dark, [&](auto DARK) { return bsel(lds, [&](auto LDSE) {
    return bsel(var, [&](auto VAR) -> const void* {
      if constexpr (MCGPU_BV(DARK) == MCGPU_BV(VAR)) return nullptr;   // (neither: k_thermal_sph; both: refused by the launcher)
      else return (const void*)k_thermal_sph_ext<MCGPU_BV(L3D), MCGPU_BV(POLA), MCGPU_BV(DARK), MCGPU_BV(LDSE), MCGPU_BV(VAR)>;
    }); }); }); }); });
}

}  // namespace mcgpu
