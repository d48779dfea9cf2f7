"""Where the Voronoi packet kernel's wave-instructions go (k_thermal_voro_cache, mc_voronoi.hip.h): one launch with each of
the three diagnostic builds (-DMCGPU_VORO_DIAG=1|2|3 into mcfost_amd/csrc/variants/voro_diag{1,2,3}.so), whose event counters
carry wave counts and lane counts of the kernel's phases instead (VD(w, l) in the source).
Usage: python tests/devtools/voro_diag.py [sites=1000000] [n=2e7]   (run from the repo root on the GPU box)"""
import json
import os
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sites = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 20_000_000
child = r'''
import sys, json, pickle, os
sys.path.insert(0, %r)
from mcfost_amd.engine import Engine
from mcfost_amd.host import model as M, voronoi as V
cfg = M.ref41()
pk = "/tmp/voro_diag_model_%d.pkl"
if os.path.exists(pk):
    m = pickle.load(open(pk, "rb"))
else:
    m = M.build_voronoi_model(cfg, %d, seed=1, tessellator=V.device_tessellator(0), platonic=True, density="smoothed", order="file")
    try:
        pickle.dump(m, open(pk, "wb"))
    except Exception:
        os.path.exists(pk) and os.remove(pk)
e = Engine(m, %d)
a = e.run_thermal(%d, seed=3)
print(json.dumps(dict(counters=a["counters"], ms=a["kernel_ms"])))
''' % (root, sites, sites, n, n)
names = {0: None,
         1: ["packets", "outer_rounds", "owner_lanes", "int_rounds", "int_lanes", "cross_rounds", "cross_lanes", "emit_rounds", "emit_lanes", "exit_lanes"],
         2: ["packets", "scan_trips", "scan_lanes", "wall_rounds", "wall_lanes", "cut_rounds", "cut_lanes", "star_rounds", "star_lanes", "recoveries"],
         3: ["packets", "stop_rounds", "stop_lanes", "pass_rounds", "pass_lanes", "c5", "c6", "newflight_rounds", "newflight_lanes", "c9"]}
out = {}
for k in (0, 1, 2, 3):
    env = dict(os.environ)
    if k:
        env["MCGPU_LIB"] = os.path.join(root, "mcfost_amd/csrc/variants/voro_diag%d.so" % k)
    r = subprocess.run([sys.executable, "-c", child], env=env, capture_output=True, text=True)
    if r.returncode:
        print(r.stderr[-3000:])
        sys.exit(1)
    d = json.loads(r.stdout.strip().splitlines()[-1])
    c = d["counters"]
    if k == 0:
        out["events"] = c
        out["ms"] = d["ms"]
    else:
        out["ms_diag%d" % k] = d["ms"]
        for nm, v in zip(names[k], list(c.values())):
            out[nm] = v
print(json.dumps(out))
g, ev = out, out["events"]
W = n / 64.0   # per "wave of packets": numbers below are per packet unless stated
print("per packet (n = %d, %d sites): %.1f crossings, %.1f interactions (%.1f scatterings), %.1f flights; %.1f ms" %
      (n, sites, ev["crossings"] / n, (ev["scatterings"] + ev["absorptions"]) / n, ev["scatterings"] / n, ev["flights"] / n, out["ms"]))


def row(name, rounds, lanes, per="round"):
    print("  %-22s %9.3f wave-rounds per packet, %5.1f of 64 lanes" % (name, g[rounds] / n, g[lanes] / max(g[rounds], 1)))


row("outer rounds", "outer_rounds", "owner_lanes")
row("interaction", "int_rounds", "int_lanes")
row("new flight", "newflight_rounds", "newflight_lanes")
row("crossing", "cross_rounds", "cross_lanes")
row("  scan trips (x4 nb)", "scan_trips", "scan_lanes")
row("  wall loop", "wall_rounds", "wall_lanes")
row("  cut-cell branch", "cut_rounds", "cut_lanes")
row("  star-neighbour", "star_rounds", "star_lanes")
row("  stop branch", "stop_rounds", "stop_lanes")
row("  pass branch", "pass_rounds", "pass_lanes")
row("emission", "emit_rounds", "emit_lanes")
print("  exits (lanes) %.3f per packet; recoveries %d" % (g["exit_lanes"] / n, g["recoveries"]))
