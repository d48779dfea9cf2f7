"""Where do a single-context run and a shared-device multi-context run of the same packets differ? (GPU diagnostic)"""
import sys, os, numpy as np
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [root, os.path.join(root, "tests")]
from mcfost_amd.engine import Engine, MultiEngine
from mcfost_amd.host import model as M
from oracle import Oracle
m = M.build_model(M.small()); n = 30001
o = Oracle(m, n)
prior = o.run_thermal(2000, seed=1)["E_abs"]
e = Engine(m, n); a = e.run_thermal(n, seed=11, frozen=True, E_prior=prior); a2 = e.run_thermal(n, seed=11, frozen=True, E_prior=prior); e.close()
me = MultiEngine(m, n, devices=(0, 0), shared_device=True); b = me.run_thermal(n, seed=11, frozen=True, E_prior=prior); me.close()
w = o.run_thermal(n, seed=11, frozen=True, E_prior=prior, n_threads=8)
def cmp(x, y, tag):
    print(tag, "counters equal", x["counters"] == y["counters"])
    for t in range(9):
        d = np.abs(x["sed"][t] - y["sed"][t])
        if d.max() > 0:
            i = np.unravel_index(d.argmax(), d.shape)
            print("  sed[%d]: max abs diff %.3e at %s (values %.17g %.17g), n differing %d" % (t, d.max(), i, x["sed"][t][i], y["sed"][t][i], (d > 0).sum()))
    d = np.abs(x["E_abs"] - y["E_abs"]); i = d.argmax()
    print("  E_abs: max abs diff / max %.3e at cell %d; max rel %.3e" % (d.max() / y["E_abs"].max(), i, (d / np.maximum(y["E_abs"], 1e-300))[y["E_abs"] > 0].max()))
cmp(a, a2, "single vs single again")
cmp(b, a, "multi(0,0) vs single")
cmp(a, w, "single vs oracle")
