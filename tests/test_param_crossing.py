"""The flight-parametric 2D crossing (option "crossing" = 1, mc_roles.hip.h::fly_step_2d_param; round 5).

Along a straight flight the distances to the walls are functions of constants of the flight (the radial wall of radius
R at s = -b0 -+ sqrt(D0 + R^2 / a), a plane at (zl - z0) / w), so the flying waves of the role kernel need no position
update per crossing.  This is NOT the arithmetic of cross_cylindrical_cell (cylindrical_grid.f90:918-1175: every crossing
re-derived from the current point, nudged by grid_prec, zj through default real), so its golden walks cannot be
reproduced bit for bit and the packet-for-packet tests do not apply.  What must hold instead:
* with the temperature frozen and the same random numbers, a packet visits the same cells but for ties at the rounding
  level: the same flights, scatterings, absorptions and exits as the oracle; the crossings within 1 % (a corner or a
  wall touched at the rounding level is an extra crossing of zero length); the absorbed energy of every well-sampled
  cell to 1e-6;
* live, the temperature agrees with the CPU oracle's like the default kernel's does: the reference's own gate
  p75(|dT| / T) < 5 % (test_suite/test_mcfost.py:46-57,88) and relative RMS <= 3 sigma_MC.
The option is off by default and the library never selects it by itself."""
import os

import numpy as np
import pytest

from helpers import mc_similar, rel_rms
from mcfost_amd.host import model as M
from oracle import Oracle
from test_kernel_emulation import emu, emu_run  # noqa: F401  (the lane emulation's library: a module-scoped fixture)


def _close(a, b, cells=1e-6):
    ca, cb = list(a["counters"].values()) if isinstance(a["counters"], dict) else a["counters"], list(b["counters"].values())
    names = ("packets", "crossings", "flights", "scatterings", "absorptions", "escaped", "killed_star")
    for k, (x, y) in enumerate(zip(ca[:7], cb[:7])):
        if names[k] == "crossings":
            assert abs(x / y - 1.0) < 0.01, (x, y)
        else:
            assert x == y, (names[k], x, y)
    assert np.array_equal(a["n_sent"], b["n_sent"]) and np.array_equal(a["sed"][4], b["sed"][4])
    strong = b["E_abs"] > 1e-3 * b["E_abs"].max()
    assert np.abs(a["E_abs"][strong] / b["E_abs"][strong] - 1.0).max() < cells
    assert abs(a["E_abs"].sum() / b["E_abs"].sum() - 1.0) < 1e-9


@pytest.mark.parametrize("name,n", [("small", 20000), ("small_hg", 10000), ("pascucci", 3000), ("ref41", 3000)])
def test_emulated_parametric_crossing_visits_the_oracles_cells(emu, name, n):
    """One lane on the CPU (tests/emu): the role kernel with the flying waves on fly_step_2d_param, flying-first so that
    nearly every crossing goes through it."""
    cfg = {"small": M.small(), "small_hg": M.small(aniso_method=2, lsepar_pola=False), "pascucci": M.pascucci(), "ref41": M.ref41()}[name]
    m = M.build_model(cfg)
    orc = Oracle(m, n)
    prior = orc.run_thermal(2000, seed=1)["E_abs"]
    b = orc.run_thermal(n, seed=5, frozen=True, E_prior=prior, n_threads=8)
    os.environ["MCGPU_EMU_ROLES"], os.environ["MCGPU_EMU_LDS"], os.environ["MCGPU_EMU_PARAM"] = "0,2,3,128", "1", "1"
    try:
        a = emu_run(emu, orc, n, 5, prior=prior)
    finally:
        for k in ("MCGPU_EMU_ROLES", "MCGPU_EMU_LDS", "MCGPU_EMU_PARAM"):
            os.environ.pop(k, None)
    _close(a, b)


@pytest.mark.gpu
def test_parametric_crossing_on_the_gpu_frozen_and_live():
    from mcfost_amd.engine import Engine
    for cfg, n in ((M.small(), 40000), (M.ref41(), 20000)):
        m = M.build_model(cfg)
        e, o = Engine(m, n), Oracle(m, n)
        prior = o.run_thermal(2000, seed=1)["E_abs"]
        ref = o.run_thermal(n, seed=7, frozen=True, E_prior=prior, n_threads=8)
        e.set_option("crossing", 1)
        _close(e.run_thermal(n, seed=7, frozen=True, E_prior=prior), ref)
        e.set_option("crossing", 0)   # ... and back: the default is the reference's arithmetic, packet for packet
        d = e.run_thermal(n, seed=7, frozen=True, E_prior=prior)
        assert d["counters"] == ref["counters"]
        e.close()
    # live, at a size with statistics: the reference's gate and the 3 sigma tolerance against the CPU oracle
    m = M.build_model(M.ref41())
    n = 4_000_000
    e, o = Engine(m, n), Oracle(m, n)
    e.set_option("crossing", 1)
    a = e.run_thermal(n, seed=21)
    b = o.run_thermal(n, seed=22, n_threads=8)
    assert a["counters"]["packets"] == n and a["counters"]["escaped"] + a["counters"]["killed_star"] == n and a["n_sent"].sum() == n
    Ta, Tb = e.temp_finale(a["E_abs"]), o.temp_finale(b["E_abs"])
    T_floor = 1.01 * m.cfg.T_min
    ok, p75 = mc_similar(Tb, Ta, 0.05, mask_threshold=T_floor)
    assert ok, p75
    sigma = 0.017 * np.sqrt(1.28e5 / n) * np.sqrt(2.0)
    assert rel_rms(Ta, Tb, T_floor) <= 3 * sigma
    sa, sb = a["sed"][0].sum(axis=(0, 1)), b["sed"][0].sum(axis=(0, 1))
    okS, p75S = mc_similar(sb, sa, 0.10, mask_threshold=200.0)
    assert okS, p75S
    # full size, Pascucci: conservation and the emitted-wavelength counts
    mp = M.build_model(M.pascucci())
    ep = Engine(mp, 10_000_000)
    ep.set_option("crossing", 1)
    r = ep.run_thermal(10_000_000, seed=3)
    c = r["counters"]
    assert c["packets"] == 10_000_000 and c["escaped"] + c["killed_star"] == 10_000_000 and r["n_sent"].sum() == 10_000_000
    ep.set_option("crossing", 0)
    r0 = ep.run_thermal(10_000_000, seed=4)
    assert abs(c["crossings"] / r0["counters"]["crossings"] - 1) < 2e-3
    okT, p75T = mc_similar(ep.temp_finale(r0["E_abs"]), ep.temp_finale(r["E_abs"]), 0.05, mask_threshold=1.01 * mp.cfg.T_min)
    assert okT, p75T
    ep.close()
    e.close()
