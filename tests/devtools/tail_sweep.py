"""Developer tool: kernel time of one thermal step against the hand-over threshold of the tail kernel (option "tail")."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from mcfost_amd.engine import Engine
from mcfost_amd.host import model as M
cfgname = sys.argv[1] if len(sys.argv) > 1 else "ref41"
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 100_000_000
cfg = {"ref41": M.ref41, "pascucci": M.pascucci, "ref41_3d": M.ref41_3d, "ref41_thick": M.ref41, "ref41_mrw": M.ref41}[cfgname]()
if cfgname in ("ref41_thick", "ref41_mrw"):
    cfg.dust_mass = 1e-2
m = M.build_model(cfg)
if cfgname == "ref41_mrw":
    M.init_mrw(m)
for thr in [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "0,32,96,256,512").split(",")]:
    e = Engine(m, n)
    e.set_option("tail", thr)
    e.run_thermal(n, seed=1)
    ms = [e.run_thermal(n, seed=2 + i)["kernel_ms"] for i in range(3)]
    print(cfgname, "tail", thr, "kernel ms", ["%.1f" % x for x in ms])
    e.close()
