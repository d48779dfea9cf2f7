// Translation unit of the kernels around a launch's end: k_thermal_roles_tail (2D role schedule that hands its last
// packets over), k_thermal_roles_bin (3D: binned deposits, chunks without tails) and k_tail (one packet per wave).
// See mc_kernels.h.
#include <hip/hip_runtime.h>

#include "mc_device.hip.h"
#include "mc_voronoi.hip.h"
#include "mc_roles.hip.h"
#include "mc_binned.hip.h"
#include "mc_tail.hip.h"
#include "mc_kernels.h"

namespace mcgpu {

const void* kpick_roles_tail(bool pola, bool dark, bool lds, bool mrw) {
  return bsel(pola, [&](auto POLA) { return bsel(dark, [&](auto DARK) { return bsel(lds, [&](auto LDSE) {
    return bsel(mrw, [&](auto MRW) -> const void* {
      return (const void*)k_thermal_roles_tail<MCGPU_BV(POLA), MCGPU_BV(DARK), MCGPU_BV(LDSE), MCGPU_BV(MRW)>;
    }); }); }); });
}

const void* kpick_roles_bin(bool pola, bool dark, bool mrw) {
  return bsel(pola, [&](auto POLA) { return bsel(dark, [&](auto DARK) { return bsel(mrw, [&](auto MRW) -> const void* {
    return (const void*)k_thermal_roles_bin<MCGPU_BV(POLA), MCGPU_BV(DARK), MCGPU_BV(MRW)>;
  }); }); });
}

const void* kpick_tail(bool l3d, bool pola, bool dark, bool mrw) {
  return bsel(l3d, [&](auto L3D) { return bsel(pola, [&](auto POLA) { return bsel(dark, [&](auto DARK) {
    return bsel(mrw, [&](auto MRW) -> const void* {
      return (const void*)k_tail<MCGPU_BV(L3D), MCGPU_BV(POLA), MCGPU_BV(DARK), MCGPU_BV(MRW)>;
    }); }); }); });
}

}  // namespace mcgpu
