"""Is the Voronoi live-mode temperature of the GPU (deposit cache, in-flight estimate = global + pending x
workgroups) unbiased against the CPU port?  Compare GPU-vs-GPU (two seeds: the noise floor) with GPU-vs-CPU at
equal packet counts."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from mcfost_amd.engine import Engine
from mcfost_amd.host import model as M
from oracle import Oracle
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
sites = int(sys.argv[2]) if len(sys.argv) > 2 else 30000
m = M.build_voronoi_model(M.ref41(), sites, seed=1)
e = Engine(m, n)
Ta = e.temp_finale(e.run_thermal(n, seed=11)["E_abs"])
Tb = e.temp_finale(e.run_thermal(n, seed=12)["E_abs"])
t = time.perf_counter()
o = Oracle(m, n)
Tc = o.temp_finale(o.run_thermal(n, seed=13, n_threads=16)["E_abs"])
print("cpu run %.1f s" % (time.perf_counter() - t))
sel = (Ta > 1.01) & (Tb > 1.01) & (Tc > 1.01)
def stats(x, y):
    r = (x[sel] - y[sel]) / y[sel]
    return "rms %.4f p75 %.4f mean %.5f" % (np.sqrt(np.mean(r * r)), np.percentile(np.abs(r), 75), r.mean())
print("cells", sel.sum(), "| GPU a vs GPU b:", stats(Ta, Tb), "| GPU a vs CPU:", stats(Ta, Tc), "| GPU b vs CPU:", stats(Tb, Tc))
