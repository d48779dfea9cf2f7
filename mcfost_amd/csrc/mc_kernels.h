// Registry of the packet kernels: the library is built from several translation units (one per kernel family,
// kern_*.hip) so that the ~200 template instantiations compile in parallel and a change to one family rebuilds one
// unit.  Each function returns the host-side handle of one instantiation for hipLaunchKernel /
// hipFuncSetAttribute (nullptr: that combination is not built).  mcgpu.hip holds every launcher; the kernels
// themselves are the templates of mc_device.hip.h, mc_roles.hip.h, mc_tail.hip.h, mc_voronoi.hip.h, mc_mono*.hip.h.
#ifndef MCFOST_AMD_MC_KERNELS_H
#define MCFOST_AMD_MC_KERNELS_H

#include <type_traits>

namespace mcgpu {

// kern_single.hip: one role per wave -- k_thermal_lds (lds) / k_thermal, k_thermal_sph
const void* kpick_thermal(bool lds, bool l3d, bool pola, bool dark, bool mrw);
const void* kpick_thermal_sph(bool l3d, bool pola, bool lds, bool mrw);
// kern_sph_ext.hip: the spherical grid with a dark zone and / or dust classes -- k_thermal_sph_ext (no random walk)
const void* kpick_thermal_sph_ext(bool l3d, bool pola, bool dark, bool lds, bool var);
// kern_roles.hip: waves with roles -- k_thermal_roles (mrw: 2D only)
const void* kpick_roles(bool l3d, bool pola, bool dark, bool lds, bool mrw);
// ... k_thermal_roles_param: 2D, LDS deposits, the flight-parametric crossing in the flying waves (option "crossing" = 1)
const void* kpick_roles_param(bool pola, bool tail);
// kern_tail.hip: the kernels of a launch's end -- k_thermal_roles_tail (2D, hands packets over), k_thermal_roles_bin
// (3D, binned deposits, chunks), k_tail (one packet per wave)
const void* kpick_roles_tail(bool pola, bool dark, bool lds, bool mrw);
const void* kpick_roles_bin(bool pola, bool dark, bool mrw);
const void* kpick_tail(bool l3d, bool pola, bool dark, bool mrw);
// kern_var.hip / kern_var_single.hip: lvariable_dust -- k_thermal_roles_var, k_thermal_var
const void* kpick_roles_var(bool l3d, bool pola, bool dark, bool lds);
const void* kpick_thermal_var(bool l3d, bool pola, bool dark, bool lds, bool mrw);
// kern_voro.hip: Voronoi grids -- k_thermal_voro_cache (block = 512 / 768 / 1024), k_thermal_voro, _mrw, _var, _roles
const void* kpick_voro_cache(bool pola, int block);
const void* kpick_voro(bool pola);
const void* kpick_voro_mrw(bool pola);
const void* kpick_voro_var(bool pola, bool mrw);
const void* kpick_voro_roles(bool pola, bool mrw);
// kern_voro_pool.hip: Voronoi grids, the pool schedule -- k_thermal_voro_pool (block = 512 / 768 / 1024)
const void* kpick_voro_pool(bool pola, int block);
// kern_mono.hip / kern_mono_other.hip: the SED / image Monte Carlo -- k_mono, k_mono_sph, k_mono_voro
const void* kpick_mono(bool l3d, bool pola, bool dark, bool scout, bool f32, bool log = false);
const void* kpick_mono_sph(bool l3d, bool pola, bool scout, bool f32);
const void* kpick_mono_voro(bool pola, bool scout, bool f32);

// a run-time bool as a compile-time one: bsel(b, [&](auto B) { ... MCGPU_BV(B) ... })
template <class F>
inline const void* bsel(bool b, F f) { return b ? f(std::true_type{}) : f(std::false_type{}); }
#define MCGPU_BV(x) (decltype(x)::value)

}  // namespace mcgpu
#endif
