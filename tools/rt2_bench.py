"""SED-mode packet loop with ray-tracing method 2's deposits (I_spec: one record per crossing) against method 1's
(xI_scatt: one record per crossing and observer) on the ref4.1 2D disk, 10 inclinations.
Usage: python tools/rt2_bench.py [n_photons_lambda=3000]   (x 128 streams per wavelength)"""
import os, sys, time, dataclasses
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mcfost_amd.engine import Engine
from mcfost_amd.host import model as M

n2 = int(float(sys.argv[1])) if len(sys.argv) > 1 else 3000
cfg = dataclasses.replace(M.ref41(), RT_n_incl=10)
m = M.build_model(cfg)
e = Engine(m, 1e7)
T = e.temp_finale(e.run_thermal(10_000_000, seed=1)["E_abs"])
e.set_rt1()
e.set_xI_precision(4)
for lam in (5, 15, 25, 35):
    tb = e.repartition_energie(lam, T, fetch=False)
    row = []
    for mode in ("rt1", "rt2", "none"):
        kw = dict(rt2=(15, 15)) if mode == "rt2" else dict(rt1=(mode == "rt1"))
        e.run_mono(lam, 50, seed=2, n_chunks=128, device_tables=tb, fetch_xI=False, **kw)
        t0 = time.perf_counter()
        a = e.run_mono(lam, n2, seed=3, n_chunks=128, device_tables=tb, fetch_xI=False, **kw)
        dt = time.perf_counter() - t0
        row.append((mode, a["counters"]["packets"] / dt, a["counters"]["crossings"] / a["counters"]["packets"]))
    print(f"lambda {lam} ({m.lam[lam - 1]:.2f} um): " + ", ".join(f"{k} {v:.3g} packets/s" for k, v, c in row) +
          f"  ({row[0][2]:.0f} crossings/packet, scout + commit passes)")
e.close()
