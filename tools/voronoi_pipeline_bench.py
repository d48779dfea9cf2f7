"""Round 3's Voronoi paths on BASELINE config 5's stand-in (ref4.1's disk sampled by SPH-like sites): the temperature
step with dust classes (k_thermal_voro_var) against the single-class kernel, the SED step, and the ray-traced SED of
the dust and of the star (k_rt1_dust_map_voro, k_stars_map_sed over optical_length_tot_voro).
Usage: python tools/voronoi_pipeline_bench.py [sites=100000] [packets=2e7]"""
import os, sys, time, dataclasses
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mcfost_amd.engine import Engine
from mcfost_amd.host import model as M

sites = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100000
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 20_000_000
cfg = dataclasses.replace(M.ref41(), RT_n_incl=10)
cache = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cache")
m = M.build_voronoi_model(cfg, sites, seed=1, cache_dir=cache)
print("cells", m.n_cells, flush=True)


def thermal(model, label):
    e = Engine(model, n)
    e.run_thermal(n // 10, seed=1)
    t0 = time.perf_counter()
    r = e.run_thermal(n, seed=2)
    dt = time.perf_counter() - t0
    c = r["counters"]
    print(f"{label}: {n / dt:.3g} packets/s ({dt * 1e3:.0f} ms; {c['crossings'] / n:.0f} crossings, "
          f"{(c['scatterings'] + c['absorptions']) / n:.0f} interactions per packet)", flush=True)
    return e, e.temp_finale(r["E_abs"])


e, T = thermal(m, "temperature step, one dust class (k_thermal_voro_cache)")
e.set_rt1()
for lam in (5, 20, 35):
    tb = e.repartition_energie(lam, T, fetch=False)
    t0 = time.perf_counter()
    a = e.run_mono(lam, 2000, seed=3, n_chunks=128, device_tables=tb, fetch_xI=False)
    dt = time.perf_counter() - t0
    Ed = tb["E_disk"] if isinstance(tb, dict) and "E_disk" in tb else 0.0
    got, ms = e.dust_map_sed(lam, T, a["n_sent"][lam - 1], Ed)
    t1 = time.perf_counter()
    st = e.stars_map_sed(lam, np.array([1.0]), seed=4)
    ms_star = (time.perf_counter() - t1) * 1e3
    print(f"lambda {lam} ({m.lam[lam - 1]:.2f} um): SED step {a['counters']['packets'] / dt:.3g} packets/s; ray-traced SED of the dust "
          f"({got.shape[0]} observers x 3840 rays) {ms:.1f} ms of kernel; stars' map {ms_star:.0f} ms of wall (nearest-site search of "
          f"441 screen points per observer included)", flush=True)
e.close()
m2 = M.build_voronoi_model(cfg, sites, seed=1, cache_dir=cache)
M.init_variable_dust(m2)
e2, _ = thermal(m2, "temperature step, 8 dust classes of |z|/H (k_thermal_voro_var)")
e2.close()
