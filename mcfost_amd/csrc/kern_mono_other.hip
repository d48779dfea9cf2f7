// Translation unit of the SED / image Monte Carlo on spherical and Voronoi grids: k_mono_sph (mc_mono.hip.h),
// k_mono_voro (mc_mono_voronoi.hip.h).  See mc_kernels.h.
#include <hip/hip_runtime.h>

#include "mc_device.hip.h"
#include "mc_voronoi.hip.h"
#include "mc_mono.hip.h"
#include "mc_mono_voronoi.hip.h"
#include "mc_kernels.h"

namespace mcgpu {

const void* kpick_mono_sph(bool l3d, bool pola, bool scout, bool f32) {
  return bsel(l3d, [&](auto L3D) { return bsel(pola, [&](auto POLA) { return bsel(scout, [&](auto SCOUT) {
    return bsel(f32, [&](auto F32) -> const void* {
      if constexpr (MCGPU_BV(SCOUT) && MCGPU_BV(F32)) return nullptr;
      else return (const void*)k_mono_sph<MCGPU_BV(L3D), MCGPU_BV(POLA), MCGPU_BV(SCOUT), MCGPU_BV(F32)>;
    }); }); }); });
}

const void* kpick_mono_voro(bool pola, bool scout, bool f32) {
  return bsel(pola, [&](auto POLA) { return bsel(scout, [&](auto SCOUT) { return bsel(f32, [&](auto F32) -> const void* {
    if constexpr (MCGPU_BV(SCOUT) && MCGPU_BV(F32)) return nullptr;
    else return (const void*)k_mono_voro<MCGPU_BV(POLA), MCGPU_BV(SCOUT), MCGPU_BV(F32)>;
  }); }); });
}

}  // namespace mcgpu
