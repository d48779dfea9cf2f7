import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mcfost_amd.host import model as M
from oracle import Oracle
m = M.build_model(M.ref41())
o = Oracle(m, 1e6)
print("affinity", len(os.sched_getaffinity(0)), "cpu_count", os.cpu_count())
for nt in (1, 16, 64, 128, 256):
    n = 40000*nt
    t=time.perf_counter(); o.run_thermal(n, seed=3, n_threads=nt); dt=time.perf_counter()-t
    print(nt, "threads %.3e packets/s  %.3e per thread" % (n/dt, n/dt/nt), flush=True)
