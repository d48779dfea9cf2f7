// host_tail.cpp -- the host side of a launch's tail (host_tail.h; mc_tail.hip.h "The last packets on the host").
//
// This translation unit compiles the product's own device source -- mc_device.hip.h, mc_roles.hip.h, mc_tail.hip.h: the
// very functions k_tail runs -- for the CPU, one lane per packet, and runs the handful of packets k_tail leaves on a
// pool of threads.  Why here and not on the GPU: a packet is one dependent chain of events, a wave alone on its SIMD
// needs 1.0-1.6 us per event whatever is done to the code (DESIGN.md "k_tail"), a host core 50-100 ns; the longest packet
// of a launch has 3e4 (ref4.1) to 5e5 (a thick disk) events.  The GPU keeps what it is good at (thinning 1e4 packets out
// in parallel), the host gets the serial remainder.
//
// Plain C++17 (g++), no HIP: the few builtins the device source uses are defined below for one lane, the atomics as
// real host atomics (the packets of a job run on several threads and share E_abs, the SED, the counters).  Nothing under
// oracle/ is included, linked or called: the CPU oracle is the tests' checker, this is the product.
#define MCGPU_LANE_EMULATION 1   // the device headers' one-lane build (no <hip/hip_runtime.h>, libm for sqrt / log / sincos)
#define MCGPU_HOST_TAIL 1        // ... with bisections where a wave probes a table with 64 lanes
#ifndef _GNU_SOURCE
#define _GNU_SOURCE 1
#endif
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#include "host_tail.h"

#define __device__
#define __host__
#define __global__
#define __shared__
#define __launch_bounds__(...)
#define __forceinline__ inline
#define __HIP_MEMORY_SCOPE_AGENT 0
#define __HIP_MEMORY_SCOPE_WORKGROUP 0
// (relaxed loads and stores of naturally aligned 4- and 8-byte values: single accesses on the hosts this is built for)
template <class Tp> static inline Tp lane_load(const Tp* p) { return *reinterpret_cast<const volatile Tp*>(p); }
template <class Tp, class Vp> static inline void lane_store(Tp* p, Vp v) { *reinterpret_cast<volatile Tp*>(p) = (Tp)v; }
#define __hip_atomic_load(p, order, scope) lane_load(p)
#define __hip_atomic_store(p, v, order, scope) lane_store((p), (v))
static inline void __builtin_amdgcn_s_sleep(int) {}
static inline void __threadfence_block() {}

namespace {
struct lane_dim3 { unsigned x, y, z; };
}
static const lane_dim3 threadIdx{0, 0, 0}, blockIdx{0, 0, 0}, blockDim{1, 1, 1}, gridDim{1, 1, 1};
struct double2 { double x, y; };
static inline double2 make_double2(double a, double b) { return double2{a, b}; }
static inline unsigned long long __ballot(bool p) { return p ? 1ull : 0ull; }
static inline int __ffsll(long long m) { return __builtin_ffsll(m); }
static inline int __popcll(unsigned long long m) { return __builtin_popcountll(m); }
template <class T> static inline T __shfl(T v, int) { return v; }
template <class T> static inline T __shfl_down(T, int) { return T(0); }
template <class T> static inline T __shfl_up(T, int) { return T(0); }
template <class T> static inline T __shfl_xor(T, int) { return T(0); }
static inline unsigned long long wall_clock64() { return 0ull; }
static inline void __syncthreads() {}
static inline int __syncthreads_or(int p) { return p; }
static inline double __longlong_as_double(long long b) { double d; memcpy(&d, &b, 8); return d; }
static inline long long __double_as_longlong(double d) { long long b; memcpy(&b, &d, 8); return b; }

// the device atomics as host atomics (relaxed: the sums commute, nothing is ordered by them)
static inline unsigned int atomicAdd(unsigned int* p, unsigned int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline double atomicAdd(double* p, double v) {
  unsigned long long* q = reinterpret_cast<unsigned long long*>(p);
  unsigned long long o = __atomic_load_n(q, __ATOMIC_RELAXED), n;
  double od;
  do { memcpy(&od, &o, 8); const double nd = od + v; memcpy(&n, &nd, 8); }
  while (!__atomic_compare_exchange_n(q, &o, n, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
  return od;
}
static inline float atomicAdd(float* p, float v) {
  unsigned int* q = reinterpret_cast<unsigned int*>(p);
  unsigned int o = __atomic_load_n(q, __ATOMIC_RELAXED), n;
  float of;
  do { memcpy(&of, &o, 4); const float nf = of + v; memcpy(&n, &nf, 4); }
  while (!__atomic_compare_exchange_n(q, &o, n, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
  return of;
}
static inline void unsafeAtomicAdd(double* p, double v) { (void)atomicAdd(p, v); }
static inline unsigned long long atomicMax(unsigned long long* p, unsigned long long v) {
  unsigned long long o = __atomic_load_n(p, __ATOMIC_RELAXED);
  while (o < v && !__atomic_compare_exchange_n(p, &o, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
  return o;
}
static inline unsigned long long atomicExch(unsigned long long* p, unsigned long long v) { return __atomic_exchange_n(p, v, __ATOMIC_RELAXED); }
static inline int atomicExch(int* p, int v) { return __atomic_exchange_n(p, v, __ATOMIC_RELAXED); }
static inline int atomicCAS(int* p, int cmp, int v) { __atomic_compare_exchange_n(p, &cmp, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED); return cmp; }
static inline unsigned int atomicCAS(unsigned int* p, unsigned int cmp, unsigned int v) {
  __atomic_compare_exchange_n(p, &cmp, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED);
  return cmp;
}

namespace mcgpu {  // the device source's unfused helpers (mc_device.hip.h), for the host compiler
static inline double nd_mul(double a, double b) { volatile double r = a * b; return r; }
static inline double nd_add(double a, double b) { volatile double r = a + b; return r; }
static inline double sqrt_nonneg(double x) { return std::sqrt(x); }   // (the device's Newton sequence is correctly rounded on its domain)
static inline float nf_mul(float a, float b) { volatile float r = a * b; return r; }
static inline float nf_add(float a, float b) { volatile float r = a + b; return r; }
static inline float nf_sub(float a, float b) { volatile float r = a - b; return r; }
double lds_raw[8];   // (what the kernels' `extern __shared__` names in this build; no kernel runs here, only tail_packet)
}  // namespace mcgpu
using std::fabs; using std::floor; using std::sqrt; using std::log; using std::exp; using std::fmax;
using std::fmin; using std::atan2; using std::acos; using std::cos; using std::copysign; using std::pow;

#include "mc_device.hip.h"
#include "mc_voronoi.hip.h"
#include "mc_binned.hip.h"
#include "mc_roles.hip.h"

// A thread's deposits.  The packets of a tail are trapped in a few opaque cells, every thread deposits into those cells'
// lines event after event, and compare-and-swap loops of several threads on one line serialise them (measured: 8 threads
// as fast as one).  So a thread sums its deposits in a small direct-mapped table of its own -- the idea of the Voronoi
// kernels' deposit cache in LDS (mc_voronoi.hip.h) -- and adds an entry to the shared array when another cell takes its
// slot, and everything at the end of the job.  What a thread has not folded yet is part of ITS view of the cell's energy
// (MCGPU_TAIL_UNFOLDED), as a wave's running sum is on the device.
namespace mcgpu_host {
struct DepositCache {
  static constexpr int SLOTS = 1024;   // (x 12 bytes: L1-resident)
  int cell[SLOTS];
  double val[SLOTS];
  double* shared = nullptr;
  void reset(double* E) { shared = E; for (int i = 0; i < SLOTS; ++i) { cell[i] = -1; val[i] = 0.0; } }
  inline void add(int ic, double v) {
    const int s = ic & (SLOTS - 1);
    if (cell[s] != ic) {
      if (cell[s] >= 0 && val[s] != 0.0) (void)atomicAdd(&shared[cell[s]], val[s]);
      cell[s] = ic; val[s] = 0.0;
    }
    val[s] += v;
  }
  inline double unfolded(int ic) const { const int s = ic & (SLOTS - 1); return cell[s] == ic ? val[s] : 0.0; }
  void fold() {
    for (int i = 0; i < SLOTS; ++i)
      if (cell[i] >= 0 && val[i] != 0.0) { (void)atomicAdd(&shared[cell[i]], val[i]); val[i] = 0.0; }
  }
};
static thread_local DepositCache* tl_deposits = nullptr;
}  // namespace mcgpu_host
#define MCGPU_TAIL_DEPOSIT(A, ic, v) mcgpu_host::tl_deposits->add((ic), (v))
#define MCGPU_TAIL_UNFOLDED(ic) mcgpu_host::tl_deposits->unfolded(ic)
// the cell's absorbed energy while other threads fold their deposits into it: a relaxed atomic load (any value the cell
// has held is as good an estimate as the reference's per-thread partial sum; a plain load would be a data race)
static inline double host_load_f64(const double* p) {
  const unsigned long long b = __atomic_load_n(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED);
  double d;
  memcpy(&d, &b, 8);
  return d;
}
#define MCGPU_TAIL_LOAD_E(p) host_load_f64(p)

#include "mc_tail.hip.h"

namespace mcgpu_host {

using namespace mcgpu;

// ---- a small pool of worker threads, started at the first job and kept ----------------------------------------------
namespace {

struct Pool {
  std::mutex job_mu;            // one job at a time (jobs of several contexts queue up here)
  std::mutex mu;
  std::condition_variable cv_work, cv_done;
  std::vector<std::thread> workers;
  void (*fn)(void*, int) = nullptr;   // the job's body: fn(arg, worker index)
  void* arg = nullptr;
  int want = 0;                 // workers the current job uses
  int running = 0;
  unsigned long long gen = 0;
  bool quit = false;

  void worker(int id) {
    unsigned long long seen = 0;
    for (;;) {
      void (*f)(void*, int);
      void* a;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv_work.wait(lk, [&] { return quit || (gen != seen && id < want); });
        if (quit) return;
        seen = gen;
        f = fn; a = arg;
      }
      f(a, id + 1);
      {
        std::lock_guard<std::mutex> lk(mu);
        if (--running == 0) cv_done.notify_all();
      }
    }
  }

  // at least n_threads - 1 workers (started by the CALLER's thread: a thread inherits its creator's CPU affinity, and
  // the HIP runtime's callback thread is not a thread to inherit from)
  void ensure(int n_threads) {
    std::lock_guard<std::mutex> lk(mu);
    while ((int)workers.size() < n_threads - 1) {
      const int id = (int)workers.size();
      workers.emplace_back([this, id] { worker(id); });
    }
  }

  // runs fn(arg, 0 .. n_threads - 1): index 0 on the calling thread
  void run(int n_threads, void (*f)(void*, int), void* a) {
    std::lock_guard<std::mutex> job(job_mu);
    const int extra = n_threads - 1;
    ensure(n_threads);
    {
      std::lock_guard<std::mutex> lk(mu);
      fn = f; arg = a; want = extra; running = extra;
      ++gen;
    }
    if (extra > 0) cv_work.notify_all();
    f(a, 0);
    if (extra > 0) {
      std::unique_lock<std::mutex> lk(mu);
      cv_done.wait(lk, [&] { return running == 0; });
      want = 0;
    }
  }

  ~Pool() {
    {
      std::lock_guard<std::mutex> lk(mu);
      quit = true;
    }
    cv_work.notify_all();
    for (auto& t : workers) t.join();
  }
};

Pool& pool() {
  static Pool* p = new Pool();   // (never destroyed: a worker may outlive main's statics at exit)
  return *p;
}

constexpr int CS_STRIDE = 32;   // a thread's event counts: TAIL_N_COUNTERS + 1 words on two cache lines of their own

struct Work {
  const DevModel* M;
  const RunArgs* A;
  const void* recs;
  unsigned int n;
  Lds T;
  std::atomic<unsigned int> next{0};
  int n_threads;
  std::vector<unsigned int> cs;   // [n_threads][CS_STRIDE]
  void (*one)(Work&, unsigned int, unsigned int*);
};

template <bool L3D, bool POLA, bool DARK, bool MRW>
void one_packet(Work& W, unsigned int i, unsigned int* cs) {
  const Rec<POLA> R = reinterpret_cast<const Rec<POLA>*>(W.recs)[i];
  (void)tail_packet<L3D, POLA, DARK, MRW>(W.T, *W.M, *W.A, R, 0, cs, 0u);
}

void body(void* arg, int tid) {
  Work& W = *static_cast<Work*>(arg);
  unsigned int* cs = &W.cs[(size_t)tid * CS_STRIDE];
  DepositCache cache;
  cache.reset(W.A->E_abs);
  tl_deposits = &cache;
  for (;;) {
    const unsigned int i = W.next.fetch_add(1u, std::memory_order_relaxed);
    if (i >= W.n) break;
    W.one(W, i, cs);
  }
  cache.fold();
  tl_deposits = nullptr;
}

template <bool L3D, bool POLA>
void pick2(Work& W, bool dark, bool mrw) {
  if (dark) { if (mrw) W.one = one_packet<L3D, POLA, true, true>; else W.one = one_packet<L3D, POLA, true, false>; }
  else { if (mrw) W.one = one_packet<L3D, POLA, false, true>; else W.one = one_packet<L3D, POLA, false, false>; }
}

}  // namespace

int default_threads(int n_devices) {
  int hw = (int)std::thread::hardware_concurrency();
  if (hw < 1) hw = 1;
  if (n_devices < 1) n_devices = 1;
  int t = hw / n_devices;
  if (t < 2) t = hw >= 2 ? 2 : 1;
  if (t > 32) t = 32;
  return t;
}

void prepare_threads(int n_threads) { pool().ensure(n_threads > 0 ? n_threads : default_threads(1)); }

void run_tail(TailJob* job) {
  job->ms = 0.0; job->threads_used = 0; job->events = 0ull;
  if (!job->n) return;
  const auto t0 = std::chrono::steady_clock::now();
  const DevModel& M = *static_cast<const DevModel*>(job->model);
  RunArgs A = *static_cast<const RunArgs*>(job->args);
  A.tail_host_max = 0u;   // (whoever runs here runs to the end)
  Work W;
  W.M = &M; W.A = &A; W.recs = job->recs; W.n = job->n;
  // the tables a workgroup stages in LDS, built by the same lds_stage from the host copies
  std::vector<double> lds((lds_bytes(M) + 7) / 8 + 8, 0.0);
  W.T = lds_carve(lds.data(), M);
  lds_stage(W.T, M);
  int nt = job->n_threads > 0 ? job->n_threads : default_threads(1);
  if ((unsigned int)nt > job->n) nt = (int)job->n;
  if (nt < 1) nt = 1;
  W.n_threads = nt;
  W.cs.assign((size_t)nt * CS_STRIDE, 0u);
  if (job->l3d) { if (job->pola) pick2<true, true>(W, job->dark != 0, job->mrw != 0); else pick2<true, false>(W, job->dark != 0, job->mrw != 0); }
  else { if (job->pola) pick2<false, true>(W, job->dark != 0, job->mrw != 0); else pick2<false, false>(W, job->dark != 0, job->mrw != 0); }
  pool().run(nt, body, &W);
  // the threads' event counts -> the counters (as k_tail's waves add theirs)
  unsigned long long longest = 0ull;
  for (int t = 0; t < nt; ++t) {
    const unsigned int* cs = &W.cs[(size_t)t * CS_STRIDE];
    for (int q = 0; q < TAIL_N_COUNTERS; ++q) A.counters[q] += (unsigned long long)cs[q];
    if (cs[TAIL_N_COUNTERS] > longest) longest = cs[TAIL_N_COUNTERS];
    job->events += (unsigned long long)cs[1] + cs[3] + cs[4];
  }
  if (longest > A.counters[10]) A.counters[10] = longest;
  job->threads_used = nt;
  job->ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

}  // namespace mcgpu_host
