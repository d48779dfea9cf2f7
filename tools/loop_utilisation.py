"""Lane utilisation of the crossing loop: build mcfost_amd/csrc/variants/lib_iters.so with -DMCGPU_COUNT_ITERS
(the dark-mirror counter then counts the loop's wave iterations) and run this on the GPU."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["MCGPU_LIB"] = os.path.join(ROOT, "mcfost_amd/csrc/variants/lib_iters.so")
from mcfost_amd.engine import Engine
from mcfost_amd.host import model as M
cfg = M.ref41()
if "--no-pola" in sys.argv:
    cfg.lsepar_pola = False
m = M.build_model(cfg)
e = Engine(m, 2e7)
r = e.run_thermal(20_000_000, seed=3)
c = r["counters"]
print("TWO =", os.environ.get("MCGPU_TWO", "0"), "kernel ms", r["kernel_ms"], "crossings", c["crossings"],
      "wave iterations", c["dark_mirrors"], "lane utilisation of the crossing loop", c["crossings"] / (64.0 * c["dark_mirrors"]))
