// Translation unit of the role-schedule packet kernel k_thermal_roles (mc_roles.hip.h).  See mc_kernels.h.
#include <hip/hip_runtime.h>

#include "mc_device.hip.h"
#include "mc_voronoi.hip.h"
#include "mc_roles.hip.h"
#include "mc_kernels.h"

namespace mcgpu {

const void* kpick_roles(bool l3d, bool pola, bool dark, bool lds, bool mrw) {
  return bsel(l3d, [&](auto L3D) { return bsel(pola, [&](auto POLA) { return bsel(dark, [&](auto DARK) { return bsel(lds, [&](auto LDSE) {
    return bsel(mrw, [&](auto MRW) -> const void* {
      if constexpr (MCGPU_BV(L3D) && MCGPU_BV(MRW)) return nullptr;   // (the role schedule's walk is 2D)
      else return (const void*)k_thermal_roles<MCGPU_BV(L3D), MCGPU_BV(POLA), MCGPU_BV(DARK), MCGPU_BV(LDSE), MCGPU_BV(MRW)>;
    }); }); }); }); });
}

const void* kpick_roles_param(bool pola, bool tail) {
  return bsel(pola, [&](auto POLA) { return bsel(tail, [&](auto TAIL) -> const void* {
    return (const void*)k_thermal_roles_param<MCGPU_BV(POLA), MCGPU_BV(TAIL)>;
  }); });
}

}  // namespace mcgpu
