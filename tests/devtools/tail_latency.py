"""Single-lane latency of the packet loop: the longest random walks of a thick disk run alone on the GPU (frozen
temperature, so the CPU oracle tells which packet ids they are); kernel time / events = time per event of a lone packet,
the quantity that sets the tail of a launch."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from mcfost_amd.engine import Engine
from mcfost_amd.host import model as M
from oracle import Oracle

m = M.build_model(M.small(n_rad=30, nz=20, dust_mass=1e-2))
n_tot = 1000000
o = Oracle(m, n_tot)
prior = o.run_thermal(200000, seed=1, n_threads=8)["E_abs"] * (n_tot / 200000)
ev = []
for pid in range(3000):
    c = o.run_thermal(1, seed=5, first_packet=pid, frozen=True, E_prior=prior)["counters"]
    ev.append(c["crossings"] + c["scatterings"] + c["absorptions"])
ev = np.array(ev)
top = np.argsort(ev)[-3:][::-1]
print("events per packet: median %d, max %d (packets %s)" % (np.median(ev), ev.max(), top))
for sched in (0, 1):
    e = Engine(m, n_tot)
    e.set_option("schedule", sched)
    for pid in top:
        e.run_thermal(1, seed=5, first_packet=int(pid), frozen=True, E_prior=prior)
        r = e.run_thermal(1, seed=5, first_packet=int(pid), frozen=True, E_prior=prior)
        c = r["counters"]
        n_ev = c["crossings"] + c["scatterings"] + c["absorptions"]
        print("schedule %d packet %d: %d crossings, %d interactions, kernel %.2f ms -> %.2f us per event" %
              (sched, pid, c["crossings"], c["scatterings"] + c["absorptions"], r["kernel_ms"], 1e3 * r["kernel_ms"] / n_ev))
    e.close()
