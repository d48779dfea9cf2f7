#!/usr/bin/env python3
"""Timing of the SED-mode packet loop (mcgpu_run_mono) on the ref4.1 grid: packets/s per wavelength,
split of the scout and commit passes.  Not the benchmark (bench.py is)."""
import argparse, json, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mcfost_amd.engine import Engine
from mcfost_amd.host import model as M

ap = argparse.ArgumentParser()
ap.add_argument("--n2", type=int, default=20000)
ap.add_argument("--lams", default="5,15,25,35")
ap.add_argument("--config", default="ref41")
ap.add_argument("--no-rt1", action="store_true")
ap.add_argument("--xi-bytes", type=int, default=8, help="accumulator type of xI_scatt: 8 or 4")
ap.add_argument("--incl", type=int, default=3, help="RT_n_incl (observers of the xI_scatt deposits)")
a = ap.parse_args()
import dataclasses
cfg = dataclasses.replace({"ref41": M.ref41, "ref41_3d": M.ref41_3d, "small": M.small}[a.config](), RT_n_incl=a.incl)
m = M.build_model(cfg)
e = Engine(m, 5e6)
T = e.temp_finale(e.run_thermal(5_000_000, seed=3)["E_abs"])
M.repartition_energie(m, T)
e.close()
e = Engine(m, 5e6)
if not a.no_rt1:
    e.set_rt1()
    e.set_xI_precision(a.xi_bytes)
for lam in [int(x) for x in a.lams.split(",")]:
    e.run_mono(lam, 100, seed=1, rt1=not a.no_rt1, fetch_xI=False)
    t = time.perf_counter()
    r = e.run_mono(lam, a.n2, seed=2, rt1=not a.no_rt1, fetch_xI=False)
    dt = time.perf_counter() - t
    c = r["counters"]
    print(json.dumps(dict(lam=lam, wl=float(m.lam[lam - 1]), frac_E_stars=float(m.frac_E_stars[lam - 1]),
                          packets=c["packets"], wall_s=dt, stream_ms=r["kernel_ms"], packets_per_s=c["packets"] / dt,
                          crossings_pp=c["crossings"] / c["packets"], scatt_pp=c["scatterings"] / c["packets"])))
