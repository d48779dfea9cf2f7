import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
from mcfost_amd.host import model as M
from mcfost_amd.engine import Engine
m = M.build_model(M.small())
n = 30000
for thr, where in ((0, 0), (40, 1), (40, 2), (100000, 2)):
    e = Engine(m, n, device=0)
    e.set_option("tail", thr); e.set_option("tail_where", where)
    a = e.run_thermal(n, seed=17)
    print(thr, where, {k: e.get_info(k) for k in ("tail_threshold", "tail_ms", "tail_where", "tail_host_packets", "tail_host_ms", "tail_host_threads", "tau_midplane")}, a["counters"]["packets"])
    e.close()
