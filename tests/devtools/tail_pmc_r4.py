"""Round 4: ONE long packet alone on the GPU, run by k_tail (role schedule, hand-over at once), without and with the
random walk -- for a PMC pass (tools/r4_lone_pmc.sh): instructions and cycles per event of the launch's tail.
Prints one line per case: LONE <case> events <crossings + interactions> walks <steps> kernel_ms <ms>."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from mcfost_amd.engine import Engine
from mcfost_amd.host import model as M
from oracle import Oracle

n_tot = 1000000
for case in ("plain", "mrw"):
    m = M.build_model(M.small(n_rad=30, nz=20, dust_mass=1e-2))
    if case == "mrw":
        M.init_mrw(m)
    o = Oracle(m, n_tot)
    prior = Oracle(M.build_model(M.small(n_rad=30, nz=20, dust_mass=1e-2)), n_tot).run_thermal(200000, seed=1, n_threads=8)["E_abs"] * (n_tot / 200000)
    ev = []
    for pid in range(3000 if case == "plain" else 60000):
        c = o.run_thermal(1, seed=5, first_packet=pid, frozen=True, E_prior=prior)["counters"]
        ev.append(c["crossings"] + c["scatterings"] + c["absorptions"] + c["mrw_steps"])
    pid = int(np.argmax(ev))
    e = Engine(m, n_tot)
    e.set_option("tail", 48)
    r = e.run_thermal(1, seed=5, first_packet=pid, frozen=True, E_prior=prior)
    c = r["counters"]
    n_ev = c["crossings"] + c["scatterings"] + c["absorptions"]
    print("LONE %s packet %d events %d crossings %d scatterings %d absorptions %d walks %d steps %d kernel_ms %.3f tail_ms %.3f us_per_event %.3f" %
          (case, pid, n_ev, c["crossings"], c["scatterings"], c["absorptions"], c["mrw_walks"], c["mrw_steps"], r["kernel_ms"],
           e.get_info("tail_ms"), 1e3 * r["kernel_ms"] / n_ev), flush=True)
    e.close()
