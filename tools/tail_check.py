import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from mcfost_amd.engine import Engine
from mcfost_amd.host import model as M
m = M.build_model(M.ref41())
for roles in ("200", "-1"):
    os.environ["MCGPU_ROLES"] = roles
    e = Engine(m, 1e8)
    e.run_thermal(1000, seed=1)
    for n in (1000, 10000, 100000, 1000000):
        ts = []
        for s in (3, 4, 5):
            r = e.run_thermal(n, seed=s)
            ts.append(r["kernel_ms"])
        c = r["counters"]
        print("roles", roles, "n", n, "kernel ms", ["%.2f" % t for t in ts], "interactions/pk %.1f" % ((c["scatterings"] + c["absorptions"]) / n))
    e.close()
