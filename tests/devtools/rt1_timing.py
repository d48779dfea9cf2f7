"""Kernel time of the RT1 ray-traced dust SED (mcgpu_rt1_dust_map) on the ref4.1-sized grid, next to the
oracle on the host cores.  Usage: python tests/devtools/rt1_timing.py [n_incl]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from mcfost_amd.host import model as M
from mcfost_amd.engine import Engine
from oracle import Oracle

n_incl = int(sys.argv[1]) if len(sys.argv) > 1 else 10
import dataclasses
cfg = dataclasses.replace(M.ref41(), RT_n_incl=n_incl)
m = M.build_model(cfg)
e = Engine(m, 2e6)
T = e.temp_finale(e.run_thermal(2_000_000, seed=3)["E_abs"])
M.repartition_energie(m, T)
e.close()
e = Engine(m, 2e6)
o = Oracle(m, 2e6)
for lam in (10, 25, 40):
    a = e.run_mono(lam, 500, seed=5, fetch_xI=False)
    ns, Ed = a["n_sent"][lam - 1], m.extra["E_disk"][lam - 1]
    for rep in range(3):
        got, ms = e.dust_map_sed(lam, T, ns, Ed)
    t0 = time.time()
    ref = o.dust_map_sed(lam, e.fetch_xI(), T, ns, Ed, n_threads=16)
    t_cpu = time.time() - t0
    n_rays = n_incl * 128 * 30
    print(f"lam {lam} ({m.lam[lam-1]:.2f} um): MC {a['kernel_ms']:.1f} ms; ray tracing {ms:.3f} ms for {n_rays} rays "
          f"({n_rays / ms * 1e3:.3g} rays/s); oracle 16 threads {t_cpu*1e3:.0f} ms; "
          f"max rel diff {np.abs(got / ref - 1)[ref != 0].max():.2e}; I = {got[:, 0]}", flush=True)

# images: 101 x 101 pixels (ref4.1.para), all inclinations
for lam in (10, 25):
    a = e.run_mono(lam, 10 ** 12, seed=5, n_phot_lim=4000.0, fetch_xI=False)   # image-mode MC: 128 x 4000 packets
    ns, Ed = a["n_sent"][lam - 1], m.extra["E_disk"][lam - 1]
    for rep in range(2):
        img, n_rays, ms = e.dust_map_image(lam, T, ns, Ed, 101, 101, 2.2 * cfg.rout, l_sym_ima=True)
    t0 = time.time()
    ref, nr = o.dust_map_image(lam, e.fetch_xI(), T, ns, Ed, 101, 101, 2.2 * cfg.rout, l_sym_ima=True, n_threads=16)
    t_cpu = time.time() - t0
    print(f"image lam {lam}: MC {a['kernel_ms']:.1f} ms; {n_rays} rays (oracle {nr}) in {ms:.2f} ms ({n_rays / ms * 1e3:.3g} rays/s); "
          f"oracle 16 threads {t_cpu*1e3:.0f} ms; pixels differing by > 1e-6 of the maximum: "
          f"{int((np.abs(img - ref) > 1e-6 * np.abs(ref).max()).sum())} of {ref.size}", flush=True)
