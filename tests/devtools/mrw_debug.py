"""Developer tool: the thick test disk of tests/test_mrw.py on the GPU against the oracle, frozen, with and without the walk."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from test_mrw import thick_disk
from mcfost_amd.engine import Engine
from oracle import Oracle

n = 20000
prior = Oracle(thick_disk(mrw=False), n).run_thermal(n, seed=1, n_threads=1)["E_abs"]
for mrw in (False, True):
    m = thick_disk(mrw=mrw) if not mrw else thick_disk()
    want = Oracle(m, n).run_thermal(n, seed=9, frozen=True, E_prior=prior, n_threads=8)
    for sched in (0, 1):
        e = Engine(m, n)
        e.set_option("schedule", sched)
        got = e.run_thermal(n, seed=9, frozen=True, E_prior=prior)
        e.close()
        print("mrw", mrw, "schedule", sched)
        print("  gpu", got["counters"])
        print("  cpu", want["counters"])
        print("  E_abs sum rel diff", got["E_abs"].sum() / want["E_abs"].sum() - 1, " cells differing > 1e-6:", int((np.abs(got["E_abs"] - want["E_abs"]) > 1e-6 * want["E_abs"].max()).sum()))
        rel = np.abs(got["E_abs"] - want["E_abs"]) / np.maximum(np.abs(want["E_abs"]), 1e-300)
        k = int(np.argmax(rel)); print("  max rel diff", rel.max(), "at cell", k, got["E_abs"][k], want["E_abs"][k], "max E", want["E_abs"].max(), " #rel>1e-9:", int((rel > 1e-9).sum()))
