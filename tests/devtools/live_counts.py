"""Event totals of the live temperature step, device against oracle, at two packet counts: how much of the difference
is the early in-flight temperature estimate (shrinks with N) and how much is seed-to-seed scatter."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
from mcfost_amd.host import model as M
from mcfost_amd.engine import Engine
from oracle import Oracle

m = M.build_model(M.ref41())
for n in (1_000_000, 8_000_000):
    e, o = Engine(m, n), Oracle(m, n)
    runs = [e.run_thermal(n, seed=s)["counters"] for s in (21, 23)]
    b = o.run_thermal(n, seed=22, n_threads=8)["counters"]
    b1 = o.run_thermal(n, seed=24, n_threads=8)["counters"] if n <= 1_000_000 else None
    for k in ("crossings", "flights", "scatterings", "absorptions"):
        print(n, k, "gpu/oracle %.4f  gpu/gpu' %.4f" % (runs[0][k] / b[k], runs[0][k] / runs[1][k]),
              ("oracle/oracle' %.4f" % (b[k] / b1[k])) if b1 else "", flush=True)
    e.close()
