"""Diagnostic (MCGPU_DIAG_FLAGS=2): when do the waves of the packet kernel run out of ids and when do they end?
Prints, per setting, the launch duration seen by the waves, the mean wave duration and the mean time at which a
wave found the id counter exhausted -- the gap between them is time spent draining with ever emptier waves."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
os.environ["MCGPU_DIAG_FLAGS"] = "2"
from mcfost_amd.engine import Engine, _DevArray
from mcfost_amd.host import model as M
import torch
m = M.build_model(M.ref41())
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
# optional: NAME=v1,v2,... sweeps an environment knob of the engine
knob, vals = (sys.argv[2].split("=")[0], sys.argv[2].split("=")[1].split(",")) if len(sys.argv) > 2 else ("MCGPU_NOP", ["0", "0"])
for val in vals:
    os.environ[knob] = val
    e = Engine(m, n)
    e.run_thermal(n // 10, seed=1)
    r = e.run_thermal(n, seed=3)
    p, cnt, nn = C.c_void_p(), C.c_void_p(), C.c_uint64()
    e._chk(e.lib.mcgpu_device_accumulators(e.ctx, C.byref(p), C.byref(nn), C.byref(cnt)), "acc")
    c = torch.as_tensor(_DevArray(cnt.value, 16, "<i8"), device=torch.device("cuda", 0)).cpu().numpy().astype(np.uint64)
    waves = int(c[14]); t0 = (~c[12]) & np.uint64(0xFFFFFFFFFFFFFFFF); span = (int(c[13]) - int(t0)) / 1e5
    print(f"{knob}={val}: kernel {r['kernel_ms']:.1f} ms; waves {waves}; span first start -> last end {span:.1f} ms; "
          f"mean wave duration {int(c[10]) / waves / 1e5:.1f} ms; mean time to 'ids exhausted' {int(c[11]) / waves / 1e5:.1f} ms", flush=True)
    e.close()
