// Translation unit of the single-role packet kernels (mc_device.hip.h): k_thermal_lds / k_thermal on cylindrical grids,
// k_thermal_sph on spherical ones.  See mc_kernels.h.
#include <hip/hip_runtime.h>

#include "mc_device.hip.h"
#include "mc_kernels.h"

namespace mcgpu {

const void* kpick_thermal(bool lds, bool l3d, bool pola, bool dark, bool mrw) {
  return bsel(lds, [&](auto LDSE) { return bsel(l3d, [&](auto L3D) { return bsel(pola, [&](auto POLA) { return bsel(dark, [&](auto DARK) {
    return bsel(mrw, [&](auto MRW) -> const void* {
      if constexpr (MCGPU_BV(LDSE)) return (const void*)k_thermal_lds<MCGPU_BV(L3D), MCGPU_BV(POLA), MCGPU_BV(DARK), MCGPU_BV(MRW)>;
      else return (const void*)k_thermal<MCGPU_BV(L3D), MCGPU_BV(POLA), MCGPU_BV(DARK), MCGPU_BV(MRW)>;
    }); }); }); }); });
}

const void* kpick_thermal_sph(bool l3d, bool pola, bool lds, bool mrw) {
  return bsel(l3d, [&](auto L3D) { return bsel(pola, [&](auto POLA) { return bsel(lds, [&](auto LDSE) {
    return bsel(mrw, [&](auto MRW) -> const void* {
      return (const void*)k_thermal_sph<MCGPU_BV(L3D), MCGPU_BV(POLA), MCGPU_BV(LDSE), MCGPU_BV(MRW)>;
    }); }); }); });
}

}  // namespace mcgpu
