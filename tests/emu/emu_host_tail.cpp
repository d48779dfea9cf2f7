// emu_host_tail.cpp -- TEST INFRASTRUCTURE: the library's host side of a launch's tail (mcfost_amd/csrc/host_tail.cpp,
// compiled into THIS test library as it is) driven without a GPU.  Every packet of a frozen-temperature run is handed to
// mcgpu_host::run_tail as a record that was never started (state S_EMIT, as the role kernel hands over a reserved work
// item), so the host threads run whole packets -- emission, flights, interactions, the walk -- with the product's own
// functions, atomics and thread pool; tests/test_host_tail.py compares the sums with the oracle's.
#include "../../mcfost_amd/csrc/host_tail.cpp"
#include "../../oracle/mc_oracle.h"
#include "emu_conv.h"

extern "C" int emu_host_tail_thermal(const oracle_model* m, const oracle_opts* o, const double* E_prior, double* E_abs,
                                     double* sed, double* n_sent, uint64_t* counters, int n_threads, double* ms) {
  Conv cv(m);
  const DevModel& M = cv.M;
  if (cv.voro || M.grid_sph || M.n_classes) return 31;   // (k_tail's grids: cylindrical, one dust class)
  const size_t nsed = (size_t)9 * m->n_lambda * m->N_thet * m->N_phi;
  memset(E_abs, 0, sizeof(double) * m->n_cells);
  memset(sed, 0, sizeof(double) * nsed);
  memset(n_sent, 0, sizeof(double) * m->n_lambda);
  unsigned long long cnt[24];
  memset(cnt, 0, sizeof(cnt));
  int err = 0;
  RunArgs A;
  memset(&A, 0, sizeof(A));
  A.seed = o->seed; A.first_packet = o->first_packet; A.n_packets = o->n_packets;
  A.qscale = o->n_replicas >= 1.0 ? o->n_replicas : 1.0;
  A.frozen = o->frozen; A.E_prior = E_prior; A.E_abs = E_abs; A.sed = sed; A.n_sent = n_sent;
  A.counters = cnt; A.err = &err;
  const bool pola = m->lsepar_pola && m->aniso_method == 1;
  const size_t n = (size_t)o->n_packets;
  std::vector<Rec<true>> rt(pola ? n : 0);
  std::vector<Rec<false>> rf(pola ? 0 : n);
  for (size_t i = 0; i < n; ++i) {
    const unsigned long long pid = o->first_packet + i;
    if (pola) { memset(&rt[i], 0, sizeof(rt[i])); rt[i].p_lo = (uint32_t)pid; rt[i].p_hi = (uint32_t)(pid >> 32); rt[i].flags = S_EMIT; }
    else { memset(&rf[i], 0, sizeof(rf[i])); rf[i].p_lo = (uint32_t)pid; rf[i].p_hi = (uint32_t)(pid >> 32); rf[i].flags = S_EMIT; }
  }
  mcgpu_host::TailJob job;
  memset(&job, 0, sizeof(job));
  job.model = &M; job.args = &A; job.recs = pola ? (const void*)rt.data() : (const void*)rf.data(); job.n = (unsigned int)n;
  job.l3d = m->l3D != 0; job.pola = pola; job.dark = M.dark != nullptr; job.mrw = M.mrw != 0; job.n_threads = n_threads;
  mcgpu_host::run_tail(&job);
  if (ms) *ms = job.ms;
  for (int q = 0; q < ORACLE_N_COUNTERS; ++q) counters[q] = cnt[q];
  return err;
}
