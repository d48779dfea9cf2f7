"""Pass statistics of the Voronoi pool schedule (mc_voronoi_pool.hip.h): the -DMCGPU_VORO_DIAG=4 build
(mcfost_amd/csrc/variants/voro_diag4.so) counts passes and lanes per phase and the rounds a wave found nothing to do.
Usage: python tests/devtools/voro_pool_diag.py [sites=1000000] [n=2e7] [block_threads=0] [log_rec=12] [cache_log=12]"""
import json, os, pickle, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
sites = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 20_000_000
bt = int(sys.argv[3]) if len(sys.argv) > 3 else 0
log_rec = int(sys.argv[4]) if len(sys.argv) > 4 else 12
cache_log = int(sys.argv[5]) if len(sys.argv) > 5 else 12
timing = os.environ.get("VP_TIMING") == "1"   # the -DMCGPU_VORO_DIAG=5 build: clock64 per stage of a pass
os.environ["MCGPU_LIB"] = os.path.join(root, "mcfost_amd/csrc/variants/%s.so" % (os.environ.get("VP_LIB") or ("voro_diag5" if timing else "voro_diag4")))
from mcfost_amd.engine import Engine
from mcfost_amd.host import model as M, voronoi as V
pk = "/tmp/voro_diag_model_%d.pkl" % sites
if os.path.exists(pk):
    m = pickle.load(open(pk, "rb"))
else:
    m = M.build_voronoi_model(M.ref41(), sites, seed=1, tessellator=V.device_tessellator(0), platonic=True, density="smoothed", order="file")
    try:
        pickle.dump(m, open(pk, "wb"))
    except Exception:
        os.path.exists(pk) and os.remove(pk)
e = Engine(m, n)
e.set_option("schedule", 3)
e.set_option("voronoi_pool_log_records", log_rec)
e.set_option("voronoi_cache_log_slots", cache_log)
a = e.run_thermal(n, seed=3, block_threads=bt)
c = list(a["counters"].values())
print(json.dumps(dict(counters=c, ms=a["kernel_ms"], block_threads=bt, log_rec=log_rec, cache_log=cache_log)))
if timing:
    names = ["loop end + fold", "choice", "pop", "touch + publish", "interaction pass", "record", "cell", "crossing (scan)", "stop/pass + stores", "idle"]
    tot = float(sum(c))
    print("%.1f ms; share of the waves' time per stage:" % a["kernel_ms"], ", ".join("%s %.1f %%" % (nm, 100.0 * v / tot) for nm, v in zip(names, c)))
    sys.exit(0)
print("%.1f ms; per packet: crossing passes %.3f (%.1f lanes), interaction passes %.3f (%.1f lanes), emission passes %.4f (%.1f lanes); "
      "idle rounds %.3f, lost pops %.3f; lanes in the two longest-list classes %.2f" %
      (a["kernel_ms"], c[1] / n, c[2] / max(c[1], 1), c[3] / n, c[4] / max(c[3], 1), c[5] / n, c[6] / max(c[5], 1), c[7] / n, c[8] / n, c[9] / n))
