"""Scattering method 1 (dust_transfer.f90:1288-1316 -- the scattering grain is drawn from the cell's
population, select_scattering_grain dust_prop.f90:1292-1336, then its own phase function, angle_diff_theta
scattering.f90:1387-1429, and Mueller matrix, get_Mueller_matrix_per_grain :1302-1324; the reference's choice when the
per-cell tables of method 2 would not fit, scattering.f90:39-66).

CPU: the oracle's grain selection against a numpy search of the same CDF from both ends; identical grains make method
1 the physics of method 2 (same temperature within the noise).  GPU: the device's temperature step with method 1 equals
the oracle's packet for packet (frozen), polarised / unpolarised / Henyey-Greenstein."""
import ctypes as C
import numpy as np
import pytest

from helpers import mc_similar
from mcfost_amd.host import model as M


def _model(n_grains=10, identical=False, **kw):
    from oracle import Oracle
    m = M.build_model(M.small(n_rad=10, nz=5, **kw))
    g = M.synthetic_grains(m, n_grains=n_grains)
    if identical:     # every size bin gets the optical properties of the middle one
        k0 = n_grains // 2
        for k in ("C_ext", "C_sca", "C_abs", "tab_g"):
            g[k] = np.ascontiguousarray(np.repeat(g[k][:, k0:k0 + 1], n_grains, axis=1))
        for k in ("tab_s11", "tab_s12", "tab_s22", "tab_s33", "tab_s34", "tab_s44"):
            g[k] = np.ascontiguousarray(np.repeat(g[k][:, k0:k0 + 1, :], n_grains, axis=1))
        g["S_grain"] = np.full(n_grains, g["S_grain"][k0], np.float32)
    p_icell, dens = M.settled_grain_density(m, g)
    m.kappa_factor = np.ones_like(m.kappa_factor)
    o0 = Oracle(m, 1000)
    t = o0.opacity(g, dens)
    lq, cdf = o0.init_reemission(kappa_abs_LTE=t["kappa_abs_LTE"].T)
    M.variable_dust_from_opacity(m, p_icell, t, lq, cdf)
    return m, g, p_icell, dens


def test_select_scattering_grain_against_a_numpy_search():
    from oracle import Oracle
    m, g, p_icell, dens = _model()
    M.init_scattering_method1(m, g, dens)
    o = Oracle(m, 1000)
    f = o.lib.oracle_select_scattering_grain
    f.restype = C.c_int
    rng = np.random.default_rng(5)
    nk = np.asarray(g["n_grains_k"])
    for lam in (1, 9, 16):
        for icell in (1, 17, m.n_cells):
            w = np.asarray(g["C_sca"], np.float64)[lam - 1] * dens[icell - 1] * nk
            up, down = np.cumsum(w), np.cumsum(w[::-1])
            # the walk is normalised by kappa * albedo / fact, the default-real albedo makes that the CDF's end to 1e-7
            for r in rng.random(200, dtype=np.float32):
                k = f(C.byref(o.cm), C.c_int(lam), C.c_int(icell), C.c_float(r))
                if r < 0.5:
                    want = int(np.searchsorted(up, float(r) * up[-1], side="right")) + 1
                else:
                    want = len(w) - int(np.searchsorted(down, float(np.float32(1.0) - r) * down[-1], side="right"))
                assert abs(k - want) <= 1 and 1 <= k <= len(w), (lam, icell, r, k, want)
                if k != want:   # only where the draw sits on a step of the CDF (the two normalisations differ by 1e-7)
                    c = up if r < 0.5 else down
                    x = (float(r) if r < 0.5 else float(np.float32(1.0) - r)) * c[-1]
                    assert np.min(np.abs(c - x)) < 1e-5 * c[-1]


def _oracle_ksca_CDF(o, m):
    """ksca_CDF(0:n_grains, p_n_cells, n_lambda) by the oracle's restatement of dust_prop.f90:976-994 -> [n_lambda, p_n_cells, n_grains + 1]."""
    out = np.zeros((m.n_lambda, int(m.variable_dust["p_n_cells"]), int(m.method1["n_grains"]) + 1), np.float64)
    o.lib.oracle_build_ksca_CDF.restype = None
    o.lib.oracle_build_ksca_CDF(C.byref(o.cm), out.ctypes.data_as(C.POINTER(C.c_double)))
    return out


def test_ksca_CDF_and_the_high_memory_grain_selection():
    """`.not. low_mem_scattering` (mem.f90:245-258): ksca_CDF stored (dust_prop.f90:976-994) and the grain selected by
    select_grainsize_high_mem's dichotomy (:1245-1288).  The restated table against numpy (rows start at 0, end at 1, never
    decrease; an empty class is all ones); the dichotomy against searchsorted on the same table; and the two modes of
    select_scattering_grain pick the same grain but where the draw sits on a step of the CDF (the walk of the low-memory
    mode normalises with kappa * albedo, the table with its own last entry)."""
    from oracle import Oracle
    m, g, p_icell, dens = _model()
    dens = np.array(dens)
    dens[3] = 0.0                                             # a class without dust
    M.init_scattering_method1(m, g, dens)
    o = Oracle(m, 1000)
    cdf = _oracle_ksca_CDF(o, m)
    nk = np.asarray(g["n_grains_k"])
    w = np.asarray(g["C_sca"], np.float64)[:, None, :] * dens[None, :, :] * nk[None, None, :]     # [lambda, class, grain]
    want = np.concatenate([np.zeros(w.shape[:2] + (1,)), np.cumsum(w, axis=2)], axis=2)
    tot = want[..., -1:]
    want = np.where(tot > 0, want / np.where(tot > 0, tot, 1.0), 1.0)
    assert np.allclose(cdf, want, rtol=1e-13, atol=0) and np.all(cdf[:, 3, :] == 1.0)
    assert np.all(cdf[:, [0, 1, 2, 4], 0] == 0.0) and np.all(cdf[..., -1] == 1.0) and np.all(np.diff(cdf, axis=2) >= 0)
    f = o.lib.oracle_select_scattering_grain
    f.restype = C.c_int
    lo = [(lam, ic, r, f(C.byref(o.cm), C.c_int(lam), C.c_int(ic), C.c_float(r)))
          for lam in (1, 9, 16) for ic in (1, 17, m.n_cells) for r in np.random.default_rng(7).random(300, dtype=np.float32)]
    m.method1["ksca_CDF"] = cdf
    o2 = Oracle(m, 1000)
    f2 = o2.lib.oracle_select_scattering_grain
    f2.restype = C.c_int
    n_diff = 0
    for lam, ic, r, k_low in lo:
        k = f2(C.byref(o2.cm), C.c_int(lam), C.c_int(ic), C.c_float(r))
        row = cdf[lam - 1, p_icell[ic - 1] - 1]
        if p_icell[ic - 1] - 1 == 3:
            continue                                           # (the empty class never scatters)
        want_k = int(np.searchsorted(row, np.float64(r), side="left"))   # first k with CDF(k) >= r ...
        assert k == max(want_k, 1) or (row[k - 1] == np.float64(r)), (lam, ic, r, k, want_k)   # (... an exact hit ends the search early)
        assert abs(k - k_low) <= 1
        n_diff += k != k_low
    assert n_diff <= 0.02 * len(lo)


def test_identical_grains_give_the_temperature_of_method_2():
    """Every size bin with the same optical properties: drawing the grain changes nothing physical.  Frozen temperature,
    one thread (reproducible): the two methods are two samples of the same transport."""
    from oracle import Oracle
    m, g, p_icell, dens = _model(identical=True, lsepar_pola=False)
    n = 120000
    o2 = Oracle(m, n)
    prior = o2.run_thermal(30000, seed=1, n_threads=1)["E_abs"] * 4.0
    a2 = o2.run_thermal(n, seed=3, n_threads=1, frozen=True, E_prior=prior)
    M.init_scattering_method1(m, g, dens)
    a1 = Oracle(m, n).run_thermal(n, seed=4, n_threads=1, frozen=True, E_prior=prior)
    assert a1["counters"]["scatterings"] > 10000
    # (event totals of 1.2e5 packets scatter by ~1 % from seed to seed: a few deep packets carry much of them)
    assert abs(a1["counters"]["scatterings"] / a2["counters"]["scatterings"] - 1) < 0.05
    assert abs(a1["counters"]["crossings"] / a2["counters"]["crossings"] - 1) < 0.05
    e1, e2 = a1["E_abs"], a2["E_abs"]
    hot = e2 > 0.05 * e2.max()
    assert hot.sum() > 10
    assert abs(e1[hot].sum() / e2[hot].sum() - 1.0) < 0.03
    assert np.median(np.abs(e1[hot] / e2[hot] - 1.0)) < 0.1


@pytest.mark.gpu
@pytest.mark.parametrize("kw", [dict(), dict(lsepar_pola=False), dict(aniso_method=2, lsepar_pola=False), dict(n_az=4, l3D=True)])
def test_device_method1_equals_the_oracle_frozen(kw):
    from oracle import Oracle
    from mcfost_amd.engine import Engine
    m, g, p_icell, dens = _model(**kw)
    M.init_scattering_method1(m, g, dens)
    n = 20000
    e, o = Engine(m, n), Oracle(m, n)
    prior = o.run_thermal(2000, seed=1)["E_abs"]
    b = o.run_thermal(n, seed=6, frozen=True, E_prior=prior, n_threads=8)
    e.set_option("schedule", 1)               # the single-role kernel ...
    a1 = e.run_thermal(n, seed=6, frozen=True, E_prior=prior)
    assert a1["counters"] == b["counters"]
    e.set_option("schedule", 0)               # ... and the role schedule (the default)
    a = e.run_thermal(n, seed=6, frozen=True, E_prior=prior)
    assert a["counters"] == b["counters"]
    assert np.array_equal(a["n_sent"], b["n_sent"]) and np.array_equal(a["sed"][4], b["sed"][4])
    assert np.allclose(a["E_abs"], b["E_abs"], rtol=1e-9, atol=1e-11 * b["E_abs"].max())
    assert np.allclose(a["sed"][0], b["sed"][0], rtol=1e-9, atol=1e-12)
    if m.cfg.lsepar_pola and m.cfg.aniso_method == 1:
        assert np.allclose(a["sed"][1:4], b["sed"][1:4], rtol=1e-5, atol=1e-6 * max(1.0, np.abs(b["sed"][0]).max()))
    assert a["counters"]["scatterings"] > 5000
    # SED mode is method 2's (ray tracing forces it, init_mcfost.f90:1659): a clear error, not a run with other tables
    from mcfost_amd.engine import McgpuError
    with pytest.raises(McgpuError):
        e.run_mono(3, 5, seed=1, n_chunks=4, rt1=False)
    # ksca_CDF on the device: the table equals the restatement bit for bit, and the loop with the dichotomy of
    # select_grainsize_high_mem equals the oracle's with the same table, packet for packet
    cdf = e.build_ksca_CDF()
    want_cdf = _oracle_ksca_CDF(o, m)
    assert np.array_equal(cdf, want_cdf)
    m.method1["ksca_CDF"] = want_cdf
    bh = Oracle(m, n).run_thermal(n, seed=6, frozen=True, E_prior=prior, n_threads=8)
    ah = e.run_thermal(n, seed=6, frozen=True, E_prior=prior)
    assert ah["counters"] == bh["counters"] and np.array_equal(ah["sed"][4], bh["sed"][4])
    assert np.allclose(ah["E_abs"], bh["E_abs"], rtol=1e-9, atol=1e-11 * bh["E_abs"].max())
    assert abs(ah["counters"]["scatterings"] / a["counters"]["scatterings"] - 1) < 0.05   # (the same physics as the walk)
    e.build_ksca_CDF(build=False)
    assert e.run_thermal(n, seed=6, frozen=True, E_prior=prior)["counters"] == b["counters"]
    del m.method1["ksca_CDF"]
    # method 2 on the same context runs other packets: the switch is live
    e.set_scattering_method1(None)
    a2 = e.run_thermal(n, seed=6, frozen=True, E_prior=prior)
    assert a2["counters"] != a["counters"]
    e.close()
