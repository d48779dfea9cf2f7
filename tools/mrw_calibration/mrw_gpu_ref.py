"""Brute-force references on the GPU for the MRW calibration (thick small disk)."""
import sys, time, numpy as np
sys.path.insert(0, ".")
from mcfost_amd.host import model as M
from mcfost_amd.engine import Engine
out = {}
dm = 1e-2
cfg = M.small(n_rad=30, nz=20, dust_mass=dm)
m = M.build_model(cfg)
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else int(1e8)
out["N"] = N
e = Engine(m, N)
t = time.time(); r = e.run_thermal(N, seed=1); t1 = time.time() - t
print(dm, "live", t1, r["kernel_ms"], r["counters"], flush=True)
out["live1"] = r["E_abs"]
np.savez_compressed("gpurun_out/r2/mrw_ref.npz", **out)
for k, seed in (("frozen1", 3), ("frozen2", 4)):
    t = time.time(); rf = e.run_thermal(N, seed=seed, frozen=True, E_prior=r["E_abs"]); t2 = time.time() - t
    print(dm, k, t2, rf["kernel_ms"], rf["counters"], flush=True)
    out[k] = rf["E_abs"]
    np.savez_compressed("gpurun_out/r2/mrw_ref.npz", **out)
r2 = e.run_thermal(N, seed=2)
out["live2"] = r2["E_abs"]
np.savez_compressed("gpurun_out/r2/mrw_ref.npz", **out)
e.close()
