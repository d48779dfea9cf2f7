// Translation unit of the pool schedule on Voronoi grids (mc_voronoi_pool.hip.h).  See mc_kernels.h.
#include <hip/hip_runtime.h>

#include "mc_device.hip.h"
#include "mc_voronoi.hip.h"
#include "mc_voronoi_pool.hip.h"
#include "mc_kernels.h"

namespace mcgpu {

const void* kpick_voro_pool(bool pola, int block) {
  return bsel(pola, [&](auto POLA) -> const void* {
    if (block > 768) return (const void*)k_thermal_voro_pool<MCGPU_BV(POLA), 1024>;
    if (block > 512) return (const void*)k_thermal_voro_pool<MCGPU_BV(POLA), 768>;
    return (const void*)k_thermal_voro_pool<MCGPU_BV(POLA), 512>;
  });
}

}  // namespace mcgpu
