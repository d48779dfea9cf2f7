// Test shim: the packed default-real xI_scatt layout (mcfost_amd/csrc/mc_xi32.hip.h) compiled for the host, so that the CPU
// suite can check its invariants and the Python mirror (mcfost_amd/engine.py::xi32_layout) without a GPU.
#define MCGPU_LANE_EMULATION 1
#define __host__
#define __device__
#include "../../mcfost_amd/csrc/mc_xi32.hip.h"

extern "C" {
// out[0..8] = binf, nA, sA, oS, sS, oT, sT, split, sum_I; out[9] = lines touched; out[10] = row floats
void xi32_probe_layout(int nRT, int pola, int contrib, int* out) {
  const mcgpu::Xi32Lay L = mcgpu::xi32_layout(nRT, pola != 0, contrib != 0);
  out[0] = L.binf; out[1] = L.nA; out[2] = L.sA; out[3] = L.oS; out[4] = L.sS; out[5] = L.oT; out[6] = L.sT; out[7] = L.split;
  out[8] = L.sum_I; out[9] = mcgpu::xi32_lines_touched(L, nRT); out[10] = mcgpu::xi32_row_floats(L, nRT);
}
// place of flux type `type` (0-based) of observer q: >= 0, -1 (no place), -2 (I = the sum of the two origins)
int xi32_probe_offset(int nRT, int pola, int contrib, int q, int type) {
  const mcgpu::Xi32Lay L = mcgpu::xi32_layout(nRT, pola != 0, contrib != 0);
  return mcgpu::xi32_offset(L, q, type, pola ? 4 : 1);
}
}
