"""How the library's host side of a launch's tail (mcfost_amd/csrc/host_tail.cpp) scales with its threads, without a GPU:
tests/emu/emu_host_tail.cpp hands it whole packets of a frozen run.  python tools/host_tail_scaling.py [n_packets]"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from mcfost_amd.host import model as M
from oracle import Oracle
import test_host_tail as H
import __graft_entry__ as g

if not os.path.exists(H.LIB) or os.path.getmtime(H.LIB) < os.path.getmtime(os.path.join(ROOT, "mcfost_amd", "csrc", "host_tail.cpp")):
    subprocess.check_call(["g++"] + g.HOST_CXX_FLAGS + ["-shared", "-o", H.LIB, H.SRC])
lib = C.CDLL(H.LIB)
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 20000
hw = os.cpu_count()
for name in ("ref41", "ref41 x10 dust + MRW"):
    cfg = M.ref41()
    if "MRW" in name:
        cfg.dust_mass *= 10
    m = M.build_model(cfg)
    if "MRW" in name:
        M.init_mrw(m, gamma=2.0, n_inter=5)
    orc = Oracle(m, n)
    prior = orc.run_thermal(20000, seed=1, n_threads=1)["E_abs"] * (n / 20000)
    for nt in (1, 2, 4, 8, 16, 32, 64, 128):
        if nt > hw:
            break
        a = H.run(lib, orc, n, 7, prior, nt)
        c = a["counters"]
        ev = c[1] + c[3] + c[4]
        print("%-22s %3d threads: %8.1f ms, %d events, longest packet %d events, %.1f ns/event/thread, %.1f events/us"
              % (name, nt, a["ms"], ev, c[10] if len(c) > 10 else 0, a["ms"] * 1e6 * nt / ev, ev / (a["ms"] * 1e3)), flush=True)
