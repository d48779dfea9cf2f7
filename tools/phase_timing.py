import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MCGPU_LIB"] = os.path.abspath("mcfost_amd/csrc/variants/lib_timing.so")
pass
from mcfost_amd.host import model as M
from mcfost_amd.engine import Engine
cfgname = sys.argv[1] if len(sys.argv) > 1 else "ref41"
cfg = {"ref41": M.ref41, "pascucci": M.pascucci, "ref41_3d": M.ref41_3d}[cfgname]()
m = M.build_model(cfg)
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 20_000_000
e = Engine(m, n)
prior = e.run_thermal(2_000_000, seed=5)["E_abs"] * (n / 2e6)
r = e.run_thermal(n, seed=7, frozen=True, E_prior=prior)
c = r["counters"]
tot = c["scatterings"] + c["absorptions"] + c["killed_star"] + c["dark_mirrors"]
print(cfgname, "kernel ms", r["kernel_ms"], "crossings/pk", c["crossings"] / n)
for k, name in (("scatterings", "exit+emit"), ("absorptions", "interact"), ("killed_star", "newflight"), ("dark_mirrors", "flight loop")):
    print("  %-12s %5.1f %%" % (name, 100.0 * c[k] / tot))
