"""Kernel time of small launches at the right temperature (L_packet scaled to the launch): the constant part
of the launch time is the tail of the longest random walks.  Roles schedule vs the single-role kernel."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from mcfost_amd.engine import Engine
from mcfost_amd.host import model as M
m = M.build_model(M.ref41())
for roles in ("200", "-1"):
    os.environ["MCGPU_ROLES"] = roles
    for n in (10000, 100000, 1000000, 10000000):
        e = Engine(m, n)
        e.run_thermal(n, seed=1)
        ts = []
        for s in (3, 4, 5):
            r = e.run_thermal(n, seed=s)
            ts.append(r["kernel_ms"])
        c = r["counters"]
        print("roles", roles, "n", n, "kernel ms", ["%.2f" % t for t in ts], "interactions/pk %.1f" % ((c["scatterings"] + c["absorptions"]) / n), flush=True)
        e.close()
