"""Generates tests/golden/oracle_anchors.npz: outputs of the ORACLE itself (oracle/mc_oracle.c), at fixed seeds on
one thread, for the parts that cannot be pinned against the reference (thermal loop, SED mode, ray tracer).  They are
regression anchors of the checker -- a change of the oracle that moves them must be deliberate -- not reference data.

    python tests/golden/make_oracle_anchors.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]


def compute():
    from mcfost_amd.host import model as M
    from oracle import Oracle
    from helpers import sed_model
    out = {}
    for tag, cfg in (("2d", M.small(RT_n_incl=2)), ("3d", M.small(n_rad=10, nz=5, n_az=6, l3D=True, RT_n_incl=2))):
        m = M.build_model(cfg)
        orc = Oracle(m, 3000)
        prior = orc.run_thermal(2000, seed=1, n_threads=1)["E_abs"]
        th = orc.run_thermal(3000, seed=7, frozen=True, E_prior=prior, n_threads=1)
        out[f"{tag}_counters"] = np.array(list(th["counters"].values()), np.int64)
        out[f"{tag}_E_abs"] = th["E_abs"]
        out[f"{tag}_sed_I"] = th["sed"][0]
        ms = sed_model(cfg, n_thermal=20000)
        o2 = Oracle(ms, 1e5)
        lam = 9
        r = o2.run_mono(lam, 30, seed=5, n_chunks=4, rt1=True, n_threads=1)
        out[f"{tag}_mono_sent"] = r["n_sent_chunk"].astype(np.int64)
        out[f"{tag}_mono_xI_sum"] = r["xI_scatt"].sum(axis=(0, 3, 4))
        args = (lam, r["xI_scatt"], ms.extra["Tdust"], r["n_sent"][lam - 1], ms.extra["E_disk"][lam - 1])
        out[f"{tag}_rt_sed"] = o2.dust_map_sed(*args)
        img, nr = o2.dust_map_image(*args, 9, 9, 2.2 * cfg.rout, ang_disque=17.3)
        out[f"{tag}_rt_image"] = img
        out[f"{tag}_rt_rays"] = np.array([nr], np.int64)
    return out


if __name__ == "__main__":
    np.savez_compressed(os.path.join(HERE, "oracle_anchors.npz"), **compute())
    print("written", os.path.join(HERE, "oracle_anchors.npz"))
