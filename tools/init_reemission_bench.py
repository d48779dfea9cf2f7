#!/usr/bin/env python3
"""Times mcgpu_init_reemission (thermal_emission.f90:404-550 on the device) in the reference's lvariable_dust layout:
one dust class per cell (p_icell = icell, mem.f90:213-244), ref4.1 grid -- 7000 classes x 100 temperatures x 39
wavelengths = 218 MB of kdB_dT_CDF that are built in HBM instead of on the host.  Prints one JSON line.
    python tools/init_reemission_bench.py [--reps 5]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    from mcfost_amd.engine import Engine
    from mcfost_amd.host import model as M
    m = M.build_model(M.ref41())
    vd = M.init_variable_dust(m, scattering=False)          # one class per layer ...
    nz, n_rad, nl, nT = m.grid["nz"], m.grid["n_rad"], m.n_lambda, m.tab_Temp.size
    layer = vd["p_icell"] - 1                                # ... expanded to one class per cell
    nc = m.n_cells
    per_cell = {k: np.ascontiguousarray(vd[k].reshape(nl, nz)[:, layer]).reshape(-1) for k in ("kappa", "kappa_abs_LTE", "albedo")}
    t0 = time.perf_counter()
    lq_l, cdf_l = vd["log_Qcool"].reshape(nz, nT), vd["kdB_dT_CDF"].reshape(nz, nT, nl)
    host_tables_s = time.perf_counter() - t0
    m.variable_dust = dict(p_n_cells=nc, p_icell=np.arange(1, nc + 1, dtype=np.int32), log_Qcool=None, kdB_dT_CDF=None,
                           **per_cell)
    e = Engine(m, 1000000)
    e.init_reemission(fetch=False)                           # warm-up (module load)
    ts = []
    for _ in range(args.reps):
        t0 = time.perf_counter()
        e.init_reemission(fetch=False)
        ts.append(time.perf_counter() - t0)
    lq, cdf = e.init_reemission()
    err_cdf = float(np.abs(cdf - cdf_l[layer]).max())        # against the harness's numpy mirror (1.e-6 as a double there)
    ok = lq_l[layer] != -1000.0
    err_lq = float(np.abs(lq[ok] - lq_l[layer][ok]).max())
    r = e.run_thermal(1000000, seed=3)                       # and the step runs on the device-built tables
    table_bytes = nc * nT * (nl + 1) * 8
    print(json.dumps({"what": "mcgpu_init_reemission, one dust class per cell", "classes": nc, "n_T": nT, "n_lambda": nl,
                      "table_MB": table_bytes / 1e6, "ms": 1e3 * float(np.median(ts)), "ms_all": [1e3 * t for t in ts],
                      "write_GBps": table_bytes / float(np.median(ts)) / 1e9,
                      "max_abs_diff_vs_host_mirror": {"kdB_dT_CDF": err_cdf, "log_Qcool": err_lq},
                      "thermal_step_on_them": {"packets": 1000000, "kernel_ms": r["kernel_ms"],
                                               "escaped+killed": r["counters"]["escaped"] + r["counters"]["killed_star"]}}))


if __name__ == "__main__":
    main()
