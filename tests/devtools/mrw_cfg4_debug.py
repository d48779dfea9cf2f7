"""Developer tool: which cells of the config-4 comparison (tests/test_mrw.py) exceed the noise-aware gate, and by how much."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from mcfost_amd.engine import Engine
from mcfost_amd.host import model as M
n = 10_000_000
cfg = M.ref41(); cfg.dust_mass = 1e-2
m0 = M.build_model(cfg); e0 = Engine(m0, n)
r0 = [e0.run_thermal(n, seed=s) for s in (3, 13, 23)]; T0 = np.array([e0.temp_finale(r["E_abs"]) for r in r0]); e0.close()
m1 = M.build_model(cfg); M.init_mrw(m1); e1 = Engine(m1, n)
r1 = [e1.run_thermal(n, seed=s) for s in (4, 14, 24)]; T1 = np.array([e1.temp_finale(r["E_abs"]) for r in r1]); e1.close()
a, b = T0.mean(0), T1.mean(0)
se = np.sqrt(T0.var(0, ddof=1) / 3 + T1.var(0, ddof=1) / 3)
E0 = np.mean([r["E_abs"] for r in r0], axis=0)
order = np.argsort(E0); rel = np.empty_like(se)
for i0 in range(0, order.size, 140):
    idx = order[i0:i0 + 140]; rel[idx] = np.median(se[idx] / np.maximum(a[idx], 1e-30))
se_s = rel * a
sel = (a > 1.2 * cfg.T_min) & (b > 1.2 * cfg.T_min)
dev = np.abs(b - a) / a
bad = np.flatnonzero(sel & (np.abs(b - a) > 0.04 * a + 5.0 * se_s))
print("cells over the gate:", bad.size)
for k in bad[:20]:
    print("cell", k, "ri", k % 100, "zj", k // 100, "T0", T0[:, k], "T1", T1[:, k], "dev %.3f" % dev[k], "se_cell %.3f se_pooled %.3f (rel)" % (se[k] / a[k], rel[k]), "dark-ish? kappa_factor", m0.kappa_factor[k])
print("p99.9 of dev over sel:", np.percentile(dev[sel], 99.9), "max", dev[sel].max())
clear = sel & (se_s < 0.002 * a)
top = np.argsort(-np.where(clear, dev, 0.0))[:30]
print("largest deviations among the clear cells (ri, zj 1-based):")
for k in top:
    print("  ri %3d zj %2d  dev %.4f  T0 %.1f T1 %.1f  rel se %.4f" % (k % 100 + 1, k // 100 + 1, dev[k], a[k], b[k], rel[k]))
