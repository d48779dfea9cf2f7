import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from mcfost_amd.host import model as M
from mcfost_amd.engine import Engine
from oracle import Oracle
m = M.build_model(M.small(n_rad=12, nz=6, n_az=8, l3D=True))
n = 20000
e, o = Engine(m, n), Oracle(m, n)
prior = o.run_thermal(2000, seed=1)["E_abs"]
e.set_E_prior(prior)
bad = []
for p in range(300):
    a = e.run_thermal(1, seed=8, first_packet=p, frozen=True)
    b = o.run_thermal(1, seed=8, first_packet=p, frozen=True, E_prior=prior)
    if a["counters"] != b["counters"]:
        bad.append(p)
        if len(bad) <= 5:
            print(p, a["counters"], b["counters"])
            ia, ib = np.nonzero(a["E_abs"])[0], np.nonzero(b["E_abs"])[0]
            common = np.intersect1d(ia, ib)
            d = np.abs(a["E_abs"][common] - b["E_abs"][common]) / b["E_abs"][common]
            print("  cells dev", len(ia), "orc", len(ib), "common", len(common), "first differing common cell", common[np.argmax(d > 1e-9)] if (d > 1e-9).any() else None)
            g = m.grid
            for ic in list(ib[:0]):
                pass
            # first cell (in oracle order of index) that is only in one
            only_a, only_b = np.setdiff1d(ia, ib), np.setdiff1d(ib, ia)
            def ijk(ic): return (g["cell_map_i"][ic], g["cell_map_j"][ic], g["cell_map_k"][ic])
            print("  only dev", [ijk(c) for c in only_a[:6]], "only orc", [ijk(c) for c in only_b[:6]])
            print("  nsent", np.nonzero(a["n_sent"])[0], np.nonzero(b["n_sent"])[0])
print("bad packets", len(bad), bad[:40])
