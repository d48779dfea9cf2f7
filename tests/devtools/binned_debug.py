"""Developer tool: the binned-deposit kernel against the oracle on the small 3D model, every counter printed."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mcfost_amd.host import model as M
from mcfost_amd.engine import Engine
from oracle import Oracle

m = M.build_model(M.small(n_rad=12, nz=6, n_az=8, l3D=True))
n = 30000
o = Oracle(m, n)
prior = o.run_thermal(2000, seed=1)["E_abs"]
b = o.run_thermal(n, seed=8, frozen=True, E_prior=prior, n_threads=8)
for dep, mb in ((1, 0), (3, 0), (3, 1), (3, 0)):
    e = Engine(m, n)
    e.set_option("deposit", dep)
    e.set_option("deposit_log_mb", mb)
    a = e.run_thermal(n, seed=8, frozen=True, E_prior=prior)
    print(dep, mb, "chunks", e.get_info("bin_chunks"), "overflow", e.get_info("bin_overflow_blocks"))
    print("  gpu", a["counters"])
    print("  cpu", b["counters"])
    print("  E_abs max rel diff", np.abs(a["E_abs"] - b["E_abs"]).max() / b["E_abs"].max(), "sed4 equal", np.array_equal(a["sed"][4], b["sed"][4]))
    e.close()
