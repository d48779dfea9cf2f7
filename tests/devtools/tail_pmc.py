"""One long packet alone on the GPU (see tail_latency.py), once per schedule, for a PMC pass:
   rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --kernel-trace ... -- python3 tests/devtools/tail_pmc.py"""
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
for pid in range(1500):
    c = o.run_thermal(1, seed=5, first_packet=pid, frozen=True, E_prior=prior)["counters"]
    ev.append(c["crossings"] + c["scatterings"] + c["absorptions"])
pid = int(np.argmax(ev))
for sched, gb, bt in ((1, 1, 64), (0, 1, 0)):
    e = Engine(m, n_tot)
    e.set_option("schedule", sched)
    r = e.run_thermal(1, seed=5, first_packet=pid, frozen=True, E_prior=prior, grid_blocks=gb, block_threads=bt)
    c = r["counters"]
    n_ev = c["crossings"] + c["scatterings"] + c["absorptions"]
    print("schedule %d packet %d: %d crossings, %d scatterings, %d absorptions, kernel %.2f ms -> %.2f us per event" %
          (sched, pid, c["crossings"], c["scatterings"], c["absorptions"], r["kernel_ms"], 1e3 * r["kernel_ms"] / n_ev))
    e.close()
