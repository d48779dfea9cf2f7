#!/usr/bin/env python3
"""Round 6 experiment: BASELINE config 2's SED step through mcgpu_multi_run_sed (the loop over wavelengths inside the
library, sharded by wavelength) with 1, 2 and 3 contexts that SHARE one GPU -- does a second context's kernels fill the
first one's gaps (scout passes, host synchronisations, the ends of its launches)?
Usage: python tools/sed_shared_contexts.py [n_photons_lambda=10000] [contexts=1,2,3]"""
import os, sys, time, dataclasses
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mcfost_amd.engine import Engine, MultiEngine
from mcfost_amd.host import model as M

n2 = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10000
ks = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1,2,3").split(",")]
cfg = dataclasses.replace(M.ref41(), RT_n_incl=10)
m = M.build_model(cfg)
e = Engine(m, 5e6)
T = e.temp_finale(e.run_thermal(20_000_000, seed=3)["E_abs"])
e.close()
lams = list(range(1, m.n_lambda + 1))
ref = None
for k in ks:
    me = MultiEngine(m, 5e6, devices=(0,) * k, shared_device=True)
    for x in me.engines:
        x.set_rt1()
        x.set_xI_precision(4)
    me.run_sed(lams[:2], 50, T)          # (module load, buffers)
    t0 = time.perf_counter()
    r = me.run_sed(lams, n2, T)
    dt = time.perf_counter() - t0
    cr = sum(c["crossings"] for c in r["counters"])
    print(f"{k} context(s) on one GPU: {r['n_sent'].sum():.4g} packets, {cr:.4g} crossings in {dt:.3f} s "
          f"({cr * 3 / dt:.3g} line operations/s); per-wavelength seconds sum {r['seconds'].sum():.3f}", flush=True)
    if ref is None:
        ref = r
    else:   # nothing is summed across contexts: the same numbers whatever the sharding
        assert np.array_equal(r["n_sent"], ref["n_sent"]) and np.array_equal(r["sed"][:, 4], ref["sed"][:, 4])
        print("   SED bins and packets identical to the single context's:", np.allclose(r["sed_rt"], ref["sed_rt"], rtol=1e-12, atol=0))
    me.close()
