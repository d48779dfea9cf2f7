// The packed default-real device layout of xI_scatt (mcgpu_set_xI_precision(4)): shared by the SED-mode kernels
// (mc_mono.hip.h), the ray tracers, the fetch / set kernels and the deposit log's fold (mc_xilog.hip.h).
#pragma once
#ifndef MCGPU_LANE_EMULATION  // (tests/emu compiles the device headers for one emulated lane on the CPU)
#include <hip/hip_runtime.h>
#endif

namespace mcgpu {

// The default-real device layout of xI_scatt (mcgpu_set_xI_precision(4)): PACKED, and what it packs is chosen so that a
// crossing's deposits touch as few 64-byte lines as possible -- the commit pass is bound by memory-side atomic LINE
// operations (2.0e10 a second for the whole chip, whatever the lanes of an instruction put on a line).
//   * A padded record of 8 values per observer holds only the values a deposit can reach: the n_Stokes Stokes values and,
//     with lsepar_contrib, the TWO origins scattered light has (n_Stokes + 2: star, + 4: thermal,
//     dust_ray_tracing.f90:519-521, 620-623; + 1 and + 3 are direct light, never deposited here).
//   * With lsepar_contrib every deposit adds the SAME flux to I and to exactly one of the two origins (:515-524, :616-627),
//     so I = star + thermal and is not stored: it is their sum when the array is read (fetch, ray tracers).  What is left
//     per observer: Q, U, V (Stokes tracking) + star + thermal, of which one packet reaches four.
//   * A sub-bin holds its nRT observers side by side, padded to whole lines as a whole.  INTERLEAVED: the observer's values
//     together, q * sA + [Stokes..., star, thermal].  SPLIT: [Stokes x nRT | star x nRT] contiguous, [thermal x nRT] from
//     the next line on -- a stellar packet touches the first part only, a thermal one the Stokes lines and the last part:
//     ten observers with Stokes tracking and contributions = 3 lines per crossing either way (interleaved with I: 4;
//     round 4's pairs of padded records: 5; FP64 records: 10).  xi32_layout picks whichever touches fewer lines.
// mcgpu_fetch_xI / mcgpu_set_xI, the ray tracers and the log's fold translate (xi32_value / xi32_offset).
struct Xi32Lay {
  int binf;     // default reals per sub-bin (whole lines)
  int nA, sA;   // the Stokes part: nA values per observer (n_Stokes, or n_Stokes - 1 where I is not stored) at q * sA
  int oS, sS;   // the stellar origin of observer q at oS + q * sS (-1: no contributions)
  int oT, sT;   // the thermal origin
  int split;    // 1: the split arrangement
  int sum_I;    // 1: I is not stored (lsepar_contrib)
};
__host__ __device__ inline int xi32_lines_of(int floats) { return (floats + 15) >> 4; }
__host__ __device__ inline Xi32Lay xi32_layout(int nRT, bool pola, bool contrib) {
  Xi32Lay L;
  const int nS = pola ? 4 : 1;
  if (!contrib) {
    L.nA = nS; L.sA = nS; L.oS = L.oT = -1; L.sS = L.sT = 0; L.split = 0; L.sum_I = 0; L.binf = xi32_lines_of(nRT * nS) << 4;
    return L;
  }
  const int nA = nS - 1, rec = nA + 2;
  const int l_inter = xi32_lines_of(nRT * rec);
  const int l_star = xi32_lines_of(nRT * (nA + 1)), l_thermal = xi32_lines_of(nRT * nA) + xi32_lines_of(nRT);
  L.nA = nA; L.sum_I = 1;
  if (l_star + l_thermal < 2 * l_inter) {
    L.split = 1; L.sA = nA; L.oS = nRT * nA; L.sS = 1; L.oT = l_star << 4; L.sT = 1; L.binf = (l_star + xi32_lines_of(nRT)) << 4;
  } else {
    L.split = 0; L.sA = rec; L.oS = nA; L.sS = rec; L.oT = nA + 1; L.sT = rec; L.binf = l_inter << 4;
  }
  return L;
}
// lines a crossing's deposits touch (the larger of a stellar and a thermal packet's)
__host__ __device__ inline int xi32_lines_touched(const Xi32Lay& L, int nRT) {
  if (!L.split) return L.binf >> 4;
  const int l_star = L.oT >> 4, l_thermal = xi32_lines_of(nRT * L.nA) + xi32_lines_of(nRT);
  return l_star > l_thermal ? l_star : l_thermal;
}
// The flight's deposit weights as a ROW in the order of the sub-bin (mc_mono.hip.h, deposit_rt1_wave_row): the whole
// sub-bin's image where the values are interleaved, the stellar part [Stokes x nRT | origin x nRT] in the split
// arrangement (a thermal packet's origin values are the same numbers, placed on the thermal lines when staged).
// In default reals, a multiple of 4 (the row is kept as 16-byte chunks).
__host__ __device__ inline int xi32_row_floats(const Xi32Lay& L, int nRT) {
  return L.split ? ((nRT * (L.nA + 1) + 3) & ~3) : L.binf;
}
// flux type (0-based index into N_type_flux) of observer q -> its place in the sub-bin; -1: a type no deposit reaches
// (reads as 0), -2: I where it is the sum of the two origins
__host__ __device__ inline int xi32_offset(const Xi32Lay& L, int q, int type, int nS) {
  if (type < nS) {
    if (!L.sum_I) return q * L.sA + type;
    return type == 0 ? -2 : q * L.sA + type - 1;
  }
  if (L.oS < 0) return -1;
  if (type == nS + 1) return L.oS + q * L.sS;
  if (type == nS + 3) return L.oT + q * L.sT;
  return -1;
}
__host__ __device__ inline double xi32_value(const float* bin, const Xi32Lay& L, int q, int type, int nS) {
  const int o = xi32_offset(L, q, type, nS);
  if (o >= 0) return (double)bin[o];
  if (o == -2) return (double)bin[L.oS + q * L.sS] + (double)bin[L.oT + q * L.sT];
  return 0.0;
}

}  // namespace mcgpu
