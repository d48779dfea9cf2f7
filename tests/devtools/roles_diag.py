"""Schedule statistics of the role kernel (mc_roles.hip.h, RQ_DIAG): run one launch with the two diagnostic builds
(tools: hipcc -DMCGPU_COUNT_ITERS=1 / =2 into mcfost_amd/csrc/variants/lib_diag{1,2}.so) and print rounds, lanes and
crossing iterations per role and per serving phase.
Usage: python tests/devtools/roles_diag.py [config=ref41] [n=2e7]   (run from the repo root on the GPU box)"""
import os, sys, subprocess, json
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
cfg = sys.argv[1] if len(sys.argv) > 1 else "ref41"
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 20_000_000
child = r'''
import sys, json
sys.path.insert(0, %r)
from mcfost_amd.engine import Engine
from mcfost_amd.host import model as M
m = M.build_model(getattr(M, %r)())
e = Engine(m, %d)
e.set_option("tail", 0)
a = e.run_thermal(%d, seed=3)
print(json.dumps(a["counters"]))
''' % (root, cfg, n, n)
names = {1: ["srv_lanes", "srv_int_lanes", "srv_rounds", "fly_cross_lanes", "srv_iters", "fly_rounds", "emit_lanes", "fly_iters"],
         2: ["emit_rounds", "emit_lanes", "exit_rounds", "exit_lanes", "int_rounds", "int_lanes", "first_rounds", "first_lanes"]}
out = {}
for k in (1, 2):
    env = dict(os.environ, MCGPU_LIB=os.path.join(root, "mcfost_amd/csrc/variants/lib_diag%d.so" % k))
    r = subprocess.run([sys.executable, "-c", child], env=env, capture_output=True, text=True)
    if r.returncode:
        print(r.stderr[-2000:]); sys.exit(1)
    c = json.loads(r.stdout.strip().splitlines()[-1])
    vals = list(c.values())[:8]
    for nm, v in zip(names[k], vals):
        out[nm] = v
print(json.dumps(out))
g = out
print("per packet (n = %d):" % n)
print("  serving rounds %.3f, lanes/round %.1f (interacting %.1f); first-crossing iterations/round %.2f" %
      (g["srv_rounds"] / n, g["srv_lanes"] / g["srv_rounds"], g["srv_int_lanes"] / g["srv_rounds"], g["srv_iters"] / g["srv_rounds"]))
print("  flying rounds %.3f, iterations/round %.1f, lanes/iteration %.1f; crossings/packet by flyers %.1f" %
      (g["fly_rounds"] / n, g["fly_iters"] / max(g["fly_rounds"], 1), g["fly_cross_lanes"] / max(g["fly_iters"], 1), g["fly_cross_lanes"] / n))
for ph in ("emit", "int", "first", "exit"):
    print("  phase %-5s ran in %.3f of the serving rounds with %.1f lanes" %
          (ph, g[ph + "_rounds"] / g["srv_rounds"], g[ph + "_lanes"] / max(g[ph + "_rounds"], 1)))
