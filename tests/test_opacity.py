"""opacity + calc_local_scattering_matrices (dust_prop.f90:791-1033, 1037-1243): the per-class
opacity and scattering tables from the grains' tables and the local grain densities.

The reference's module cannot be built here (dust_prop -> utils -> SPRNG / generated sources), so the CPU restatement
(oracle_opacity) is PARITY UNPINNED; it is held by known answers (one grain size; identical classes; the normalisations
the reference states: prob_s11_pos ends at 1 with the unresolved forward peak in bin 1, 2 pi sum s11 = k_sca-normalised
phase function) and by an independent numpy evaluation in double precision.  On the GPU: mcgpu_opacity equals the
restatement -- the sums bit for bit (default-real tables are rounded after every term on both sides), entries that pass
through sin / cos / powf within one unit in the last place -- and the temperature step on the device-built tables equals
the oracle's on the oracle-built tables packet for packet."""
import dataclasses
import numpy as np
import pytest

from mcfost_amd.host import model as M

PI = np.pi
FACT = 149597870700.0 * 100.0 * 1.0e-8


def _setup(per_cell=True, n_grains=12, **kw):
    m = M.build_model(M.small(n_rad=10, nz=5, **kw))
    g = M.synthetic_grains(m, n_grains=n_grains)
    p_icell, dens = M.settled_grain_density(m, g, per_cell=per_cell, n_classes=0 if per_cell else 4)
    return m, g, p_icell, dens


def _mirror(m, g, dens):
    """Double-precision numpy evaluation of the same formulas (no default-real rounding): an independent check."""
    f8 = np.float64
    n = dens[None, :, :] * np.asarray(g["n_grains_k"], f8)[None, None, :]            # [1, nc, ng]
    Ce, Cs, Ca = (np.asarray(g[k], f8)[:, None, :] for k in ("C_ext", "C_sca", "C_abs"))
    kap, ksca, kabs = (Ce * n).sum(-1), (Cs * n).sum(-1), (Ca * n).sum(-1)              # [nl, nc]
    gpos = (Cs * n * np.asarray(g["tab_g"], f8)[:, None, :]).sum(-1) / ksca
    S = np.asarray(g["S_grain"], f8)
    w = n * S[None, None, :]                                                            # [1, nc, ng]
    mu = lambda key: np.einsum("lka,lck->lca", np.asarray(g[key], f8), np.broadcast_to(w, (kap.shape[0],) + w.shape[1:]))
    s11 = mu("tab_s11")
    na1 = s11.shape[-1]
    nang = na1 - 1
    th = np.arange(na1) * (PI / nang)
    dth = PI / nang
    inc = s11 * np.sin(th) * dth
    prob = np.zeros_like(s11)
    prob[..., 2:] = np.cumsum(inc[..., 2:], axis=-1)
    prob[..., 1:] += (ksca - prob[..., nang])[..., None]
    prob /= ksca[..., None]
    out = dict(kappa=kap * FACT, kappa_abs_LTE=kabs * FACT, tab_albedo_pos=ksca / kap, tab_g_pos=gpos,
               prob_s11_pos=prob, tab_s11_pos=s11 * dth / (ksca[..., None] * 2 * PI))
    for k in ("12", "22", "33", "34", "44"):
        out[f"tab_s{k}_o_s11_pos"] = mu("tab_s" + k) / s11
    return out


def test_oracle_against_the_double_precision_mirror():
    from oracle import Oracle
    m, g, p_icell, dens = _setup()
    t = Oracle(m, 1000).opacity(g, dens)
    ref = _mirror(m, g, dens)
    assert np.allclose(t["kappa"], ref["kappa"], rtol=1e-12)
    assert np.allclose(t["kappa_abs_LTE"], ref["kappa_abs_LTE"], rtol=1e-12)
    assert np.allclose(t["tab_albedo_pos"], ref["tab_albedo_pos"], rtol=2e-7)
    for k in ("tab_s11_pos", "prob_s11_pos", "tab_s12_o_s11_pos", "tab_s22_o_s11_pos", "tab_s33_o_s11_pos",
              "tab_s34_o_s11_pos", "tab_s44_o_s11_pos"):
        assert np.allclose(t[k], ref[k], rtol=3e-5, atol=3e-6), k   # default-real sums of 12 terms and 180 angles
    # the properties the reference states
    p = t["prob_s11_pos"]
    assert np.all(p[..., 0] == 0.0) and np.allclose(p[..., -1], 1.0, atol=1e-6)
    assert np.all(np.diff(p, axis=-1) >= -1e-7)
    assert np.all(p[..., 1] > 0.0)                                   # the unresolved forward peak sits in bin 1 (:1150)
    assert np.allclose(np.abs(t["tab_s12_o_s11_pos"]).max(), np.abs(ref["tab_s12_o_s11_pos"]).max(), rtol=1e-4)


def test_oracle_known_answers():
    from oracle import Oracle
    m, g, p_icell, dens = _setup(n_grains=1)
    o = Oracle(m, 1000)
    t = o.opacity(g, dens)
    # one grain size: the albedo is C_sca / C_ext whatever the density, kappa is linear in it
    alb = (np.asarray(g["C_sca"], np.float64) / np.asarray(g["C_ext"], np.float64))[:, 0]
    assert np.allclose(t["tab_albedo_pos"], alb[:, None], rtol=2e-7)
    assert np.allclose(t["kappa"], np.asarray(g["C_ext"], np.float64)[:, :1] * dens[:, 0][None, :] * FACT, rtol=1e-14)
    # ... and the Mueller ratios are the grain's own
    r = np.asarray(g["tab_s12"], np.float64)[:, 0] / np.asarray(g["tab_s11"], np.float64)[:, 0]
    assert np.allclose(t["tab_s12_o_s11_pos"], r[:, None, :], rtol=1e-5, atol=1e-7)
    # identical classes give identical rows; an empty class scatters nothing (dust_prop.f90:1222-1236)
    m, g, p_icell, dens = _setup(per_cell=False)
    dens[:] = dens[0]
    dens[-1] = 0.0
    t = o.opacity(g, dens)
    for k in ("kappa", "tab_albedo_pos", "tab_s11_pos", "prob_s11_pos", "tab_s34_o_s11_pos"):
        assert np.array_equal(t[k][:, 0], t[k][:, 1]), k
    assert np.all(t["kappa"][:, -1] == 0.0) and np.all(t["tab_albedo_pos"][:, -1] == 0.0)
    assert np.all(t["prob_s11_pos"][:, -1, 1:] == 1.0) and np.all(t["tab_s11_pos"][:, -1] == 1.0)
    # Henyey-Greenstein (aniso_method 2): the C_sca-weighted asymmetry parameter and the ray tracer's phase function
    m, g, p_icell, dens = _setup(aniso_method=2, lsepar_pola=False)
    t = Oracle(m, 1000).opacity(g, dens)
    n = dens[None] * np.asarray(g["n_grains_k"])[None, None, :]
    cs = np.asarray(g["C_sca"], np.float64)[:, None, :]
    gm = (cs * n * np.asarray(g["tab_g"], np.float64)[:, None, :]).sum(-1) / (cs * n).sum(-1)
    assert np.allclose(t["tab_g_pos"], gm, rtol=3e-6)
    # sum over the sphere: 2 pi sum_l s11(l) sin(theta_l) = 1 up to the 1-degree quadrature
    th = np.arange(181) * (PI / 180)
    tot = 2 * PI * (t["tab_s11_pos"].astype(np.float64) * np.sin(th)).sum(-1)
    ok = t["tab_g_pos"] < 0.6
    assert np.allclose(tot[ok], 1.0, atol=2e-2)


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["mueller_pola", "mueller", "hg", "layers"])
def test_device_opacity_equals_the_restatement(variant):
    from oracle import Oracle
    from mcfost_amd.engine import Engine
    kw = dict(mueller_pola={}, mueller=dict(lsepar_pola=False), hg=dict(aniso_method=2, lsepar_pola=False), layers={})[variant]
    m, g, p_icell, dens = _setup(per_cell=variant != "layers", **kw)
    if variant == "mueller":
        m.p_lambda_fixed = 0                 # every wavelength's own cumulative table (SED mode's layout)
    t = Oracle(m, 1000).opacity(g, dens)
    e = Engine(m, 1000)
    d = e.opacity(g, p_icell, dens)
    e.close()
    for k in ("kappa", "kappa_abs_LTE", "tab_albedo_pos", "tab_g_pos"):
        assert np.array_equal(d[k], t[k]), k          # sums in the reference's order and types: bit for bit
    pcols = d["prob_s11_pos"].shape[0]
    ulp = lambda a, b: np.abs(a.astype(np.float64) - b) <= 1.2e-7 * np.abs(b) + 1e-37
    if variant != "hg":
        assert np.all(ulp(d["prob_s11_pos"], t["prob_s11_pos"][:pcols]))       # (sin: one unit in the last place at most,
        assert np.mean(d["prob_s11_pos"] == t["prob_s11_pos"][:pcols]) > 0.9   #  and rarely)
        assert np.array_equal(d["tab_s11_pos"], t["tab_s11_pos"])
        for k in ("tab_s12_o_s11_pos", "tab_s22_o_s11_pos", "tab_s33_o_s11_pos", "tab_s34_o_s11_pos", "tab_s44_o_s11_pos"):
            if t[k] is not None:
                assert np.array_equal(d[k], t[k]), k
    else:
        assert np.allclose(d["tab_s11_pos"], t["tab_s11_pos"], rtol=1e-6)         # powf


@pytest.mark.gpu
def test_temperature_step_on_device_built_tables():
    """lvariable_dust end to end without a host table builder: grains + densities -> mcgpu_opacity -> mcgpu_init_reemission
    -> the packet loop, against the oracle on the restatement's tables, frozen, packet for packet."""
    from oracle import Oracle
    from mcfost_amd.engine import Engine
    m, g, p_icell, dens = _setup()
    m.kappa_factor = np.ones_like(m.kappa_factor)        # lvariable_dust: the density is in kappa (dust_prop.f90:953)
    o0 = Oracle(m, 1000)
    t = o0.opacity(g, dens)
    lq, cdf = o0.init_reemission(kappa_abs_LTE=t["kappa_abs_LTE"].T)
    n = 20000
    e = Engine(m, n)
    e.opacity(g, p_icell, dens, fetch=False)
    e.init_reemission(fetch=False)
    M.variable_dust_from_opacity(m, p_icell, t, lq, cdf)
    o = Oracle(m, n)
    prior = o.run_thermal(2000, seed=1)["E_abs"]
    a = e.run_thermal(n, seed=5, frozen=True, E_prior=prior)
    b = o.run_thermal(n, seed=5, frozen=True, E_prior=prior, n_threads=8)
    e.close()
    assert a["counters"] == b["counters"]
    assert np.array_equal(a["n_sent"], b["n_sent"])
    assert np.allclose(a["E_abs"], b["E_abs"], rtol=1e-9, atol=1e-11 * b["E_abs"].max())
    assert a["counters"]["scatterings"] > 1000 and a["counters"]["absorptions"] > 1000
