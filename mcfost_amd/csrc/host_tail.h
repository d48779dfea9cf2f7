// The library's host side of a launch's tail (host_tail.cpp): the few packets k_tail leaves (mc_tail.hip.h, "The last
// packets on the host") finished on CPU threads by the device source itself, compiled for one lane.  Internal interface
// between mcgpu.hip (hipcc) and host_tail.cpp (plain C++17, g++): no HIP types, no table types -- the two structures
// travel as untyped pointers and both sides include mc_device.hip.h for their layout.
#ifndef MCFOST_AMD_HOST_TAIL_H
#define MCFOST_AMD_HOST_TAIL_H

namespace mcgpu_host {

struct TailJob {
  const void* model;     // mcgpu::DevModel whose table pointers are HOST copies of the device tables
  const void* args;      // mcgpu::RunArgs whose array pointers (E_abs, E_prior, sed, n_sent, counters, err) are HOST copies
  const void* recs;      // n records, Rec<POLA> (mc_roles.hip.h)
  unsigned int n;
  int l3d, pola, dark, mrw;   // the instantiation of tail_packet
  int n_threads;         // <= 0: default_threads()
  // filled in by run_tail
  double ms;             // wall time of the packets
  int threads_used;
  unsigned long long events;   // crossings + interactions the host ran
};

// Runs the job's packets to their end; deposits, SED bins, n_sent and counters are added to the arrays `args` points to
// (atomically: the packets are spread over threads).  Thread-safe (jobs of several contexts queue up).
__attribute__((visibility("hidden"))) void run_tail(TailJob* job);

// Starts the pool's threads for jobs of n_threads, from the calling thread (call it from the thread that enqueues the
// launch: run_tail is called by a thread of the HIP runtime, whose CPU affinity the workers should not inherit).
__attribute__((visibility("hidden"))) void prepare_threads(int n_threads);

// threads a job uses when it does not say: the machine's hardware threads shared among `n_devices` processes or
// contexts, at least 2, at most 32
__attribute__((visibility("hidden"))) int default_threads(int n_devices);

}  // namespace mcgpu_host
#endif
