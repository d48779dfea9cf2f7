// Translation unit of the role-schedule kernel with dust classes, k_thermal_roles_var (mc_roles.hip.h).  See mc_kernels.h.
#include <hip/hip_runtime.h>

#include "mc_device.hip.h"
#include "mc_voronoi.hip.h"
#include "mc_roles.hip.h"
#include "mc_kernels.h"

namespace mcgpu {

const void* kpick_roles_var(bool l3d, bool pola, bool dark, bool lds) {
  return bsel(l3d, [&](auto L3D) { return bsel(pola, [&](auto POLA) { return bsel(dark, [&](auto DARK) {
    return bsel(lds, [&](auto LDSE) -> const void* {
      return (const void*)k_thermal_roles_var<MCGPU_BV(L3D), MCGPU_BV(POLA), MCGPU_BV(DARK), MCGPU_BV(LDSE)>;
    }); }); }); });
}

}  // namespace mcgpu
