// The fold of the SED commit pass's deposit log (mc_mono.hip.h "The deposits as a log"): sort the records by sub-bin, then
// sum each sub-bin's consecutive records in registers.
//
// A record is (key = sub-bin index | flag_star << 31, flight id, path length l); the flight's row holds its deposit weights
// w[q][0..nv) for the nRT observers (nv = 4 with Stokes tracking, else 1).  xI_scatt[bin][q][slot] += l * w[q][slot] for the
// Stokes slots and, with lsepar_contrib, the copy of l * w[q][0] in the slot of the packet's origin (calc_xI_scatt[_pola],
// dust_ray_tracing.f90:480-632, as save_radiation_field calls it per crossing, radiation_field.f90:63-89).
//
// Sorted by sub-bin (hipCUB's radix sort on the key's low bits: 3e10 records/s), the records of one sub-bin are
// consecutive, so a wave walks a chunk of records with one (observer, slot) per lane: per record one wave-uniform 12-byte
// load, ONE coalesced load of the row (160 bytes for 10 observers) and one multiply-add per lane, summed in registers
// while the sub-bin stays the same; an atomic add per lane only where the sub-bin changes (630 000 sub-bins against 1e9
// records per wavelength).  Measured standalone (tools/xi_fold_bench.hip, config 2's statistics): 1.4e10 records/s, the row
// gather at 2.3 TB/s; with the sort 1.0e10 -- 2.4x what the same deposits cost as global atomics in isolation, 4.7x what
// the commit pass achieved with them inside the transport kernel.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "mc_xi32.hip.h"

namespace mcgpu {

constexpr int XI_SEG_CHUNK = 512;    // records a wave sums (a sub-bin that continues in the next chunk costs one more flush)
constexpr int XI_SEG_UNROLL = 8;     // records whose row loads are in flight together

// temp storage of the sort for n records with keys of end_bit significant bits
size_t xi_sort_temp_bytes(size_t n, int end_bit);

// keys / vals [n] (unsorted; entries a wave reserved and did not use hold a key >= n_bins) -> sorted copies keys2 / vals2 ->
// xI (the packed default-real device layout `xi`, mc_xi32.hip.h: [bin][xi.binf default reals]) += the sums; where the layout
// does not store I (lsepar_contrib) the lanes of I add nothing and the copies go to the two origins' places.  Asynchronous
// on `stream`; returns a hipError_t.
int xi_sort_fold(hipStream_t stream, const unsigned int* keys, const unsigned long long* vals, unsigned int* keys2,
                 unsigned long long* vals2, size_t n, int end_bit, void* temp, size_t temp_bytes, const float* rows, int nRT,
                 int nv, int contrib, unsigned int n_bins, float* xI, Xi32Lay xi);

}  // namespace mcgpu
