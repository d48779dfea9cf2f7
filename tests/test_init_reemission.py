"""init_reemission (SURVEY 8f rank 4 "host table builders on device"; thermal_emission.f90:404-550): the LTE tables
log_Qcool_minus_extra_heating(T, p_icell) and kdB_dT_CDF(lambda, T, p_icell) built on the device from the context's
kappa_abs_LTE (``mcgpu_init_reemission``).  PARITY: module thermal_emission cannot be compiled here, so the oracle's
restatement (``oracle_init_reemission``) is UNPINNED against the reference; it is pinned by known answers below
(Stefan-Boltzmann law, Wien displacement of the grey re-emission CDF, the harness's independent numpy mirror), and the
device is held to the oracle (exp / log differ from libm in the last place: 1e-12)."""
import ctypes as C

import numpy as np
import pytest

from mcfost_amd.host import model as M
from oracle import Oracle
from oracle.binding import _a, _p
from test_kernel_emulation import emu  # noqa: F401  (the lane emulator's fixture)

SIGMA = 5.670367e-8


def test_oracle_known_answers_grey_dust():
    """a grey dust (kappa_abs = 1) on a fine wavelength grid: the tables have closed forms"""
    m = M.build_model(M.small())
    n = 4000
    edges = np.logspace(-2.0, 5.0, n + 1)
    lam, dlam = np.sqrt(edges[1:] * edges[:-1]), edges[1:] - edges[:-1]
    lq, cdf = Oracle(m, 10).init_reemission(np.ones(n), lam, dlam)
    T = m.tab_Temp.astype(np.float64)
    # Stefan-Boltzmann: 4 pi int B_lambda dlambda = 4 sigma T^4; the table holds log(Q(T) - Q(T_1))
    want = 4.0 * SIGMA * (T ** 4 - T[0] ** 4)
    sel = (T > 3.0) & (T < 3000.0)      # (0.01 .. 1e5 micron holds the whole spectrum for these)
    assert np.allclose(np.exp(lq[0, sel]), want[sel], rtol=2e-4)
    assert lq[0, 0] == -1000.0
    # the CDF is a CDF
    assert np.all(np.diff(cdf[0], axis=1) >= 0.0) and np.allclose(cdf[0, :, -1], 1.0)
    # displacement law: dB_lambda/dT dlambda is x^4 e^x / (e^x - 1)^2 dx in x = hc / (lambda k T), so the median
    # wavelength of the grey re-emission CDF sits at lambda T = (hc/k) / x_median
    x = np.linspace(1e-4, 60.0, 600001)
    f = x ** 4 * np.exp(x) / np.expm1(x) ** 2
    cum = np.cumsum(f)
    x_med = float(np.interp(0.5, cum / cum[-1], x))
    c2 = 299792458.0 * 6.626070040e-34 / 1.38064852e-23 * 1e6   # micron K
    med = np.array([np.interp(0.5, cdf[0, t], edges[1:]) for t in range(T.size)])
    assert np.allclose(med[sel] * T[sel], c2 / x_med, rtol=2e-3)


def test_oracle_against_the_harness_mirror():
    """host/model.py's numpy mirror (written independently in round 1; it takes 1.e-6 as a double, the reference and the
    oracle as a default real: 2.5e-8 on the wavelengths)"""
    for cfg in (M.small(), M.small(n_lambda=39, n_rad=12, nz=6)):
        m = M.build_model(cfg)
        o = Oracle(m, 10)
        lq, cdf = o.init_reemission()
        assert np.allclose(lq[0, 1:], m.log_Qcool[1:], rtol=0, atol=3e-6) and lq[0, 0] == m.log_Qcool[0] == -1000.0
        assert np.allclose(cdf[0], np.asarray(m.kdB_dT_CDF).reshape(cdf[0].shape), rtol=0, atol=1e-6)


def _classes(m, nc=7):
    rng = np.random.default_rng(5)
    ka = np.asarray(m.kappa_abs_LTE)[None, :] * np.exp(rng.normal(0.0, 0.5, (nc, m.n_lambda)))
    ka[1, : m.n_lambda // 2] = 0.0       # a class transparent in the blue half
    ka[2, :] = 0.0                       # a class that does not absorb at all: the table stays 0, log_Qcool -1000
    return ka


def test_emulated_kernel_against_the_oracle(emu):   # noqa: F811
    m = M.build_model(M.small())
    o = Oracle(m, 10)
    ka = _classes(m)
    want_lq, want_cdf = o.init_reemission(ka)
    nc, nl, nT = ka.shape[0], m.n_lambda, m.tab_Temp.size
    lq = np.zeros((nc, nT))
    cdf = np.zeros((nc, nT, nl))
    rc = emu.emu_init_reemission(C.c_int(nc), C.c_int(nT), C.c_int(nl), _p(_a(m.tab_Temp, np.float32), C.c_float),
                                 _p(_a(m.lam, np.float64), C.c_double), _p(_a(m.delta_lam, np.float64), C.c_double),
                                 _p(_a(ka, np.float64), C.c_double), _p(lq, C.c_double), _p(cdf, C.c_double))
    assert rc == 0
    assert np.array_equal(lq == -1000.0, want_lq == -1000.0) and np.all(lq[2] == -1000.0) and np.all(cdf[2] == 0.0)
    ok = want_lq != -1000.0
    assert np.allclose(lq[ok], want_lq[ok], rtol=0, atol=1e-12)
    assert np.allclose(cdf, want_cdf, rtol=0, atol=1e-13)


def _heating(m, ka, frac):
    """dudt and heating_norm of a made-up heating rate: `frac` of each class's cooling rate somewhere in the table"""
    lq0, _ = Oracle(m, 10).init_reemission(ka)
    nT = m.tab_Temp.size
    q_mid = np.exp(np.where(lq0[:, nT // 3] > -999.0, lq0[:, nT // 3], 0.0))
    hn = np.linspace(2.0, 5.0, ka.shape[0])
    return frac * q_mid * hn, hn


def test_extra_heating_known_answers_and_the_emulated_kernel(emu):   # noqa: F811
    """lextra_heating (thermal_emission.f90:486-494): the floor of the cooling rate becomes max(Qcool(T1), dudt / norm)
    -- or, ldudt_implicit, max(Qcool(T1), (ufac T - dudt) / norm).  Known answers for the restatement: no heating = the
    plain tables; exp(log_Qcool) + the floor is the same cooling rate whatever the floor; a heating rate below the
    floor at T1 changes nothing; the re-emission CDF does not depend on it.  Then the device kernel (one lane) = the
    restatement, explicit and implicit."""
    m = M.build_model(M.small())
    o = Oracle(m, 10)
    ka = _classes(m)[:2]
    nc, nl, nT = ka.shape[0], m.n_lambda, m.tab_Temp.size
    lq0, cdf0 = o.init_reemission(ka)
    dudt, hn = _heating(m, ka, 1.0)
    lq1, cdf1 = o.init_reemission(ka, dudt=dudt, heating_norm=hn)
    assert np.array_equal(cdf0, cdf1)
    lqz, _ = o.init_reemission(ka, dudt=np.zeros(nc), heating_norm=hn)
    assert np.array_equal(lqz, lq0)                                  # (a heating rate below the floor: max() keeps the floor)
    # Qcool(T) = exp(lq0) + Qcool(T1) = exp(lq1) + dudt / hn wherever both are defined: the difference of the floors
    both = (lq0 > -999.0) & (lq1 > -999.0)
    d = np.exp(lq0) - np.exp(lq1)
    assert both.sum() > nT
    for c in range(nc):   # the same constant for every T of a class, to the cancellation of the two exponentials
        col = np.flatnonzero(both[c])
        dd = d[c, col]
        assert np.all(np.abs(dd - dd[0]) <= 1e-6 * dd[0] + 1e-11 * np.exp(lq0[c, col]))
        assert abs(dd[0] / (dudt[c] / hn[c]) - 1.0) < 1e-3           # (Qcool(T1) is tiny against the heating)
    assert (lq1 == -1000.0).sum() > (lq0 == -1000.0).sum()          # (temperatures that cool less than they are heated)
    for ufac in (0.0, 1.0e-3 * float(dudt.max() / m.tab_Temp[-1])):
        du = dudt if ufac == 0.0 else -dudt        # (implicit: (ufac T - dudt) / norm with dudt = u^n / dt)
        want_lq, want_cdf = o.init_reemission(ka, dudt=du, heating_norm=hn, ufac_implicit=ufac)
        lq = np.zeros((nc, nT))
        cdf = np.zeros((nc, nT, nl))
        rc = emu.emu_init_reemission_ex(C.c_int(nc), C.c_int(nT), C.c_int(nl), _p(_a(m.tab_Temp, np.float32), C.c_float),
                                        _p(_a(m.lam, np.float64), C.c_double), _p(_a(m.delta_lam, np.float64), C.c_double),
                                        _p(_a(ka, np.float64), C.c_double), _p(_a(du, np.float64), C.c_double),
                                        _p(_a(hn, np.float64), C.c_double), C.c_double(ufac), _p(lq, C.c_double), _p(cdf, C.c_double))
        assert rc == 0
        _same_with_heating(lq, want_lq, (ufac * m.tab_Temp[None, :].astype(np.float64) - du[:, None]) / hn[:, None] if ufac else (du / hn)[:, None])
        assert np.allclose(cdf, want_cdf, rtol=0, atol=1e-13)


def _same_with_heating(lq, want_lq, floor):
    """log(Qcool - floor): where the heating nearly cancels the cooling the difference carries the rounding of Qcool itself
    (exp and the sums differ in the last place between the builds): 1e-12 of Qcool, not of the difference"""
    both = (lq != -1000.0) & (want_lq != -1000.0)
    assert (lq != -1000.0).sum() > 0 and np.abs((lq != -1000.0).astype(int) - (want_lq != -1000.0).astype(int)).sum() <= 1
    qa, qb = np.exp(np.where(both, lq, 0.0)), np.exp(np.where(both, want_lq, 0.0))
    tol = 1e-12 * (qb + np.abs(np.broadcast_to(floor, qb.shape)))
    assert np.all(np.abs(qa - qb)[both] <= tol[both])


@pytest.mark.gpu
def test_extra_heating_on_the_device():
    """mcgpu_init_reemission_ex on a variable-dust context = the restatement; a heating that makes the table decrease with
    T is refused (the reference's "Qrad_minus_dudt is not an increasing function of T")."""
    from mcfost_amd.engine import Engine, McgpuError
    m = M.build_model(M.small())
    vd = M.init_variable_dust(m)
    nc, nl, nT = vd["p_n_cells"], m.n_lambda, m.tab_Temp.size
    ka = vd["kappa_abs_LTE"].reshape(nl, nc).T
    dudt, hn = _heating(M.build_model(M.small()), ka, 0.5)
    want_lq, want_cdf = Oracle(M.build_model(M.small()), 10).init_reemission(ka, dudt=dudt, heating_norm=hn)
    vd["log_Qcool"] = vd["kdB_dT_CDF"] = None
    e = Engine(m, 1000)
    lq, cdf = e.init_reemission(dudt=dudt, heating_norm=hn)
    assert (want_lq == -1000.0).sum() > nc
    _same_with_heating(lq, want_lq, (dudt / hn)[:, None])
    assert np.allclose(cdf, want_cdf, rtol=0, atol=1e-13)
    # implicit, with a u(T) / dt that grows faster than class 0's cooling rate over one step of the temperature grid:
    # Q(T) - (ufac T - dudt) then DEcreases there
    lq0, _ = Oracle(M.build_model(M.small()), 10).init_reemission(ka)
    Q, T, k = np.exp(lq0[0]), m.tab_Temp.astype(np.float64), nT // 3
    a = 2.0 * (Q[k + 1] - Q[k]) / (T[k + 1] - T[k])
    with pytest.raises(McgpuError, match="increase with T"):
        e.init_reemission(dudt=np.full(nc, a * T[k] - 0.5 * Q[k]), heating_norm=np.ones(nc), ufac_implicit=a)
    e.close()


@pytest.mark.gpu
def test_device_tables_against_the_oracle_and_a_run_on_them():
    """single class: the device-built tables equal the oracle's; a frozen step on the device with ITS tables equals the
    oracle's step on the same tables (fetched back) packet for packet"""
    from mcfost_amd.engine import Engine, McgpuError
    m = M.build_model(M.small())
    o = Oracle(m, 20000)
    want_lq, want_cdf = o.init_reemission()
    host_lq, host_cdf = m.log_Qcool, m.kdB_dT_CDF
    m.log_Qcool = m.kdB_dT_CDF = None     # leave them to the device
    e = Engine(m, 20000)
    with pytest.raises(McgpuError, match="mcgpu_init_reemission"):
        e.run_thermal(100, seed=1)
    lq, cdf = e.init_reemission()
    assert np.array_equal(lq == -1000.0, want_lq == -1000.0)
    ok = want_lq != -1000.0
    assert np.allclose(lq[ok], want_lq[ok], rtol=0, atol=1e-12) and np.allclose(cdf, want_cdf, rtol=0, atol=1e-13)
    m.log_Qcool, m.kdB_dT_CDF = lq[0].copy(), cdf[0].copy()
    o2 = Oracle(m, 20000)
    prior = o2.run_thermal(2000, seed=1, n_threads=1)["E_abs"]
    a = o2.run_thermal(20000, seed=7, frozen=True, E_prior=prior, n_threads=4)
    e.set_E_prior(prior)
    b = e.run_thermal(20000, seed=7, frozen=True)
    assert [b["counters"][k] for k in ("escaped", "killed_star", "scatterings", "absorptions")] == \
           [a["counters"][k] for k in ("escaped", "killed_star", "scatterings", "absorptions")]
    assert np.allclose(b["E_abs"], a["E_abs"], rtol=1e-9, atol=1e-9 * a["E_abs"].max())
    assert np.allclose(e.temp_finale(b["E_abs"]), o2.temp_finale(b["E_abs"]), rtol=2e-6)   # default-real logs
    m.log_Qcool, m.kdB_dT_CDF = host_lq, host_cdf


@pytest.mark.gpu
def test_device_tables_per_class():
    """variable dust: every class's tables built on the device from the class opacities (nothing uploaded)"""
    from mcfost_amd.engine import Engine
    m = M.build_model(M.small())
    vd = M.init_variable_dust(m)
    nc, nl, nT = vd["p_n_cells"], m.n_lambda, m.tab_Temp.size
    want_lq, want_cdf = Oracle(M.build_model(M.small()), 10).init_reemission(vd["kappa_abs_LTE"].reshape(nl, nc).T)
    up_lq, up_cdf = vd["log_Qcool"], vd["kdB_dT_CDF"]
    vd["log_Qcool"] = vd["kdB_dT_CDF"] = None
    e = Engine(m, 20000)
    lq, cdf = e.init_reemission()
    ok = want_lq != -1000.0
    assert lq.shape == (nc, nT) and np.array_equal(lq == -1000.0, ~ok)
    assert np.allclose(lq[ok], want_lq[ok], rtol=0, atol=1e-12) and np.allclose(cdf, want_cdf, rtol=0, atol=1e-13)
    # the tables the harness would have uploaded (its numpy mirror) differ only by the literal's precision
    assert np.allclose(cdf.reshape(-1), up_cdf, rtol=0, atol=1e-6)
    # and the step runs on them: the oracle on the fetched tables, packet for packet
    vd["log_Qcool"], vd["kdB_dT_CDF"] = lq.reshape(-1).copy(), cdf.reshape(-1).copy()
    o = Oracle(m, 20000)
    prior = o.run_thermal(2000, seed=1, n_threads=1)["E_abs"]
    a = o.run_thermal(20000, seed=11, frozen=True, E_prior=prior, n_threads=4)
    e.set_E_prior(prior)
    b = e.run_thermal(20000, seed=11, frozen=True)
    assert [b["counters"][k] for k in ("escaped", "killed_star", "scatterings", "absorptions")] == \
           [a["counters"][k] for k in ("escaped", "killed_star", "scatterings", "absorptions")]
    assert np.allclose(b["E_abs"], a["E_abs"], rtol=1e-9, atol=1e-9 * a["E_abs"].max())
    vd["log_Qcool"], vd["kdB_dT_CDF"] = up_lq, up_cdf
