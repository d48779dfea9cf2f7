// Translation unit of the packet kernels on Voronoi grids (mc_voronoi.hip.h, k_thermal_voro_roles of mc_roles.hip.h).
// See mc_kernels.h.
#include <hip/hip_runtime.h>

#include "mc_device.hip.h"
#include "mc_voronoi.hip.h"
#include "mc_roles.hip.h"
#include "mc_kernels.h"

namespace mcgpu {

const void* kpick_voro_cache(bool pola, int block) {
  return bsel(pola, [&](auto POLA) -> const void* {
    if (block > 768) return (const void*)k_thermal_voro_cache<MCGPU_BV(POLA), 1024>;
    if (block > 512) return (const void*)k_thermal_voro_cache<MCGPU_BV(POLA), 768>;
    return (const void*)k_thermal_voro_cache<MCGPU_BV(POLA), 512>;
  });
}
const void* kpick_voro(bool pola) { return pola ? (const void*)k_thermal_voro<true> : (const void*)k_thermal_voro<false>; }
const void* kpick_voro_mrw(bool pola) { return pola ? (const void*)k_thermal_voro_mrw<true> : (const void*)k_thermal_voro_mrw<false>; }
const void* kpick_voro_var(bool pola, bool mrw) {
  return bsel(pola, [&](auto POLA) { return bsel(mrw, [&](auto MRW) -> const void* {
    return (const void*)k_thermal_voro_var<MCGPU_BV(POLA), MCGPU_BV(MRW)>;
  }); });
}
const void* kpick_voro_roles(bool pola, bool mrw) {
  return bsel(pola, [&](auto POLA) { return bsel(mrw, [&](auto MRW) -> const void* {
    return (const void*)k_thermal_voro_roles<MCGPU_BV(POLA), MCGPU_BV(MRW)>;
  }); });
}

}  // namespace mcgpu
