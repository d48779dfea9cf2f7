"""BASELINE config 2 end to end on one MI355X: ref4.1 2D disk, temperature step + SED (Monte Carlo SED bins,
xI_scatt, ray-traced SED of the dust for 10 inclinations) through mcfost_amd/host/pipeline.py.
Usage: python tools/run_config2.py [n_thermal=1e8] [n_photons_lambda=10000] [xI bytes = 8 | 4] [RT n_incl = 10]   (x 128 streams per
wavelength; ref4.1.para itself asks for 3 inclinations, the 10 of the default are the harder case the rounds have quoted)"""
import os, sys, time, dataclasses
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mcfost_amd.engine import Engine
from mcfost_amd.host import model as M, pipeline as P

n_th = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
n2 = int(float(sys.argv[2])) if len(sys.argv) > 2 else 10000
n_incl = int(sys.argv[4]) if len(sys.argv) > 4 else 10
cfg = dataclasses.replace(M.ref41(), RT_n_incl=n_incl)
m = M.build_model(cfg)
e = Engine(m, n_th)
xi_bytes = int(sys.argv[3]) if len(sys.argv) > 3 else 8
e.set_rt1()
e.set_xI_precision(xi_bytes)   # 4: xI_scatt in default real, the packed layout of mc_xi32.hip.h (mcgpu_set_xI_precision)
e.run_thermal(1000, seed=1)   # (first launch: module load)
t0 = time.perf_counter()
r = P.temperature_and_sed(P.EngineBackend(e), m, n_th, n2, seed=5)
wall = time.perf_counter() - t0
s = r["seconds"]
n_sed = r["n_sent"].sum()
print(f"thermal step: {n_th:.3g} packets in {s['thermal']:.3f} s ({n_th / s['thermal']:.3g} packets/s incl. Temp_finale)")
print(f"repartition_energie (device, per wavelength): {s['repartition_energie']:.3f} s")
print(f"SED Monte Carlo (xI_scatt in {xi_bytes}-byte sums): {m.n_lambda} wavelengths, {n_sed:.3g} packets in {s['sed_mc']:.3f} s ({n_sed / s['sed_mc']:.3g} packets/s, "
      f"scout + commit passes and the fetch of the SED arrays included)")
if xi_bytes == 4 and r.get("sed_crossings"):
    from mcfost_amd.engine import xi32_layout
    lay = xi32_layout(cfg.RT_n_incl * cfg.RT_n_az, bool(cfg.lsepar_pola and cfg.aniso_method == 1), bool(cfg.lsepar_contrib))
    ops = r["sed_crossings"] * lay["lines_touched"]
    print(f"  {r['sed_crossings']:.4g} crossings ({r['sed_crossings'] / n_sed:.1f} per packet) x {lay['lines_touched']} lines of 64 bytes = {ops:.4g} memory-side "
          f"atomic line operations: {ops / s['sed_mc']:.3g} /s over the whole SED step = {ops / s['sed_mc'] / 2.37e10:.2f} of the 2.37e10 /s the chip does "
          f"(tools/atomic_scope_bench.hip; bench.py's ATOMIC_LINE_PEAK)")
print(f"ray-traced dust SED: {m.n_lambda} x {cfg.RT_n_incl} inclinations in {s['ray_tracing']:.3f} s")
print(f"total {wall:.2f} s;  Tdust {r['Tdust'].min():.1f} .. {r['Tdust'].max():.1f} K")
f = P.sed_flux(m, r["sed_mc"], r["n_sent"])[0].sum(axis=0)   # (N_thet, n_lambda)
k = int(np.argmax(f[m.capt_sup - 1]))
print(f"SED peak (inclination bin {m.capt_sup}): lambda = {m.lam[k]:.2f} um; ray-traced I of the dust there per observer:",
      np.array2string(r["sed_rt"][k, :, 0], precision=3))
e.close()
