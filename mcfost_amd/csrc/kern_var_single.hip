// Translation unit of the single-role kernel with dust classes, k_thermal_var (mc_device.hip.h).  See mc_kernels.h.
#include <hip/hip_runtime.h>

#include "mc_device.hip.h"
#include "mc_kernels.h"

namespace mcgpu {

const void* kpick_thermal_var(bool l3d, bool pola, bool dark, bool lds, bool mrw) {
  return bsel(l3d, [&](auto L3D) { return bsel(pola, [&](auto POLA) { return bsel(dark, [&](auto DARK) { return bsel(lds, [&](auto LDSE) {
    return bsel(mrw, [&](auto MRW) -> const void* {
      return (const void*)k_thermal_var<MCGPU_BV(L3D), MCGPU_BV(POLA), MCGPU_BV(DARK), MCGPU_BV(LDSE), MCGPU_BV(MRW)>;
    }); }); }); }); });
}

}  // namespace mcgpu
