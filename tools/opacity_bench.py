"""opacity + calc_local_scattering_matrices at the size of the reference's lvariable_dust case: ref4.1's
7000 cells with their own dust, 100 grain sizes, 50 wavelengths, 181 angles, polarised (7 tables of 253 MB).
Device: mcgpu_opacity (upload of the grains' tables and of the densities included, results stay in HBM); CPU: the
restatement (oracle_opacity, one thread) on a bounded sample of the classes.
Usage: python tools/opacity_bench.py [n_grains=100] [cpu sample classes=200]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mcfost_amd.engine import Engine
from mcfost_amd.host import model as M

ng = int(sys.argv[1]) if len(sys.argv) > 1 else 100
n_cpu = int(sys.argv[2]) if len(sys.argv) > 2 else 200
m = M.build_model(M.ref41())
g = M.synthetic_grains(m, n_grains=ng)
p_icell, dens = M.settled_grain_density(m, g)
nc, nl, na1 = dens.shape[0], m.n_lambda, 181
e = Engine(m, 1000)
e.opacity(g, p_icell, dens, fetch=False)    # (first call: module load, allocation)
ts = []
for _ in range(3):
    t0 = time.perf_counter()
    e.opacity(g, p_icell, dens, fetch=False)
    ts.append(time.perf_counter() - t0)
out_bytes = 7 * 4 * nc * nl * na1
print(f"device: {nc} classes x {ng} grains x {nl} wavelengths x {na1} angles, 7 tables: {min(ts) * 1e3:.1f} ms per call "
      f"(uploads included; {out_bytes / 1e9:.2f} GB of tables written: {out_bytes / min(ts) / 1e9:.0f} GB/s)")
t0 = time.perf_counter()
e.init_reemission(fetch=False)
print(f"device: init_reemission of the {nc} classes: {(time.perf_counter() - t0) * 1e3:.1f} ms")
d = e.opacity(g, p_icell[:], dens, fetch=True)
try:
    from oracle import Oracle
    o = Oracle(m, 1000)
    t0 = time.perf_counter()
    t = o.opacity(g, dens[:n_cpu])
    dt = time.perf_counter() - t0
    print(f"CPU restatement, 1 thread: {n_cpu} classes in {dt:.2f} s -> {dt / n_cpu * nc:.1f} s for {nc} classes "
          f"({dt / n_cpu * nc / min(ts):.0f} x the device call)")
    same = all(np.array_equal(d[k][:, :n_cpu], t[k]) for k in ("kappa", "kappa_abs_LTE", "tab_albedo_pos", "tab_s11_pos", "tab_s12_o_s11_pos"))
    print("device == restatement on the sample (kappa, kappa_abs_LTE, albedo, s11, s12/s11):", same)
except ImportError:
    pass
e.close()
