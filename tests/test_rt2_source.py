"""Ray tracing method 2, the source function (init_dust_source_fct2, dust_ray_tracing.f90:717-806 = calc_Isca_rt2_star
:1245-1440 + calc_Isca_rt2 :907-1240 + calc_Jth): from the I_spec / I_spec_star the packet loop deposits to eps_dust2 /
eps_dust2_star of one inclination.  The reference's module is unbuildable here: the restatement is PARITY UNPINNED, held
by known answers -- isotropic scattering makes the scattered field the same in every direction and equal to the
k_sca-weighted mean intensity; no radiation leaves the thermal source J_th / kappa_ext; linearity; the stellar term
follows the phase function of the angle between the star and the observer.  GPU: the device equals the restatement."""
import numpy as np
import pytest

from helpers import sed_model
from mcfost_amd.host import model as M
from oracle import Oracle

AU_TO_CM = 149597870700.0 * 100.0


def _field(m, seed=1, ntf=None):
    cfg = m.cfg
    ns = 4 if (cfg.lsepar_pola and cfg.aniso_method == 1) else 1
    ntf = ns + (4 if cfg.lsepar_contrib else 0)
    rng = np.random.default_rng(seed)
    I = np.zeros((m.n_cells, 15, 15, ntf))
    I[..., 0] = rng.random((m.n_cells, 15, 15)) + 0.5
    if ns == 4:
        I[..., 1:4] = 0.1 * (rng.random((m.n_cells, 15, 15, 3)) - 0.5)
    if cfg.lsepar_contrib:
        I[..., ns + 1] = 0.3 * I[..., 0]
        I[..., ns + 3] = 0.7 * I[..., 0]
    return I, rng.random(m.n_cells) * 5.0


def test_source_function_known_answers():
    # isotropic, unpolarised dust: the Pascucci grain (g = 0)
    m = sed_model(M.small(dust="pascucci", lsepar_pola=False, lsepar_contrib=False, RT_n_incl=3), n_thermal=20000)
    o = Oracle(m, 1000)
    lam = 8
    I, Istar = _field(m)
    T = m.extra["Tdust"]
    n_sent, Ed = 1.0e6, m.extra["E_disk"][lam - 1]
    eps, eps_star = o.init_dust_source_fct2(lam, 2, I, Istar, T, n_sent, Ed)
    assert eps.shape == (m.n_cells, 2, 15, 1) and eps_star.shape == (m.n_cells, 2, 1000, 1)
    # (a) every direction and both hemispheres see the same scattered field ...
    assert np.allclose(eps, eps[:, :1, :1, :], rtol=2e-5)
    # ... and without radiation only the thermal source is left: linear in the field
    zero, zstar = o.init_dust_source_fct2(lam, 2, 0 * I, 0 * Istar, T, n_sent, Ed)
    assert np.all(zstar == 0) and np.all(zero >= 0)
    two, _ = o.init_dust_source_fct2(lam, 2, 2 * I, Istar, T, n_sent, Ed)
    assert np.allclose(two - zero, 2 * (eps - zero), rtol=1e-4, atol=1e-30)
    # (b) the value: (photon_energy / V) * kappa_sca * s11_iso * sum(Inu) / kappa_ext, s11_iso = tab_s11_pos (constant)
    s11 = np.asarray(m.tab_s11_pos, np.float64).reshape(m.n_lambda, -1)[lam - 1]
    assert np.allclose(s11[1:-1], s11[1], rtol=1e-6)
    pe = (m.E_stars[lam - 1] + Ed) * m.lam[lam - 1] * 1e-6 / (n_sent * AU_TO_CM * np.pi)
    want = pe / np.asarray(m.grid["volume"]) * float(m.albedo[lam - 1]) * s11[1] * I[..., 0].sum(axis=(1, 2))
    got = (eps - zero)[:, 0, 0, 0].astype(np.float64)
    ok = want > 1e-3 * want.max()
    assert np.allclose(got[ok], want[ok], rtol=2e-4)
    # (c) the stellar term: isotropic too, proportional to I_spec_star
    assert np.allclose(eps_star, eps_star[:, :1, :1, :], rtol=2e-5)
    ratio = eps_star[:, 0, 0, 0] / (pe / np.asarray(m.grid["volume"]) * float(m.albedo[lam - 1]) * s11[1] * Istar)
    assert np.allclose(ratio[Istar > 0.1], 1.0, rtol=2e-4)


def test_source_function_anisotropic_polarised():
    m = sed_model(M.small(RT_n_incl=3), n_thermal=20000)
    o = Oracle(m, 1000)
    lam = 5
    I, Istar = _field(m)
    T = m.extra["Tdust"]
    eps, eps_star = o.init_dust_source_fct2(lam, 3, I, Istar, T, 1.0e6, m.extra["E_disk"][lam - 1])
    assert eps.shape[-1] == 8 and eps_star.shape[-1] == 4
    assert np.isfinite(eps).all() and np.isfinite(eps_star).all()
    assert np.all(eps[..., 1] >= 0) and np.all(np.abs(eps[..., 2]) <= np.pi + 1e-6)        # (P, angle) form
    assert np.all(eps[..., 1] <= eps[..., 0] * (1 + 1e-5))                                  # polarised <= total
    # I = direct-star-scattered + thermal + dust-scattered contributions (slots n_Stokes+2 .. +4)
    assert np.allclose(eps[..., 0], eps[..., 5] + eps[..., 6] + eps[..., 7], rtol=1e-3, atol=1e-30)   # (default-real sums of 225 terms)
    # forward-throwing dust: the stellar term peaks where the observer looks along the star's light
    es = eps_star[..., 0]
    cell = int(np.argmax(Istar))
    assert es[cell].max() > 3 * np.median(es[cell])
    # the two hemispheres differ (the observer is inclined), the azimuth varies
    assert not np.allclose(eps[:, 0], eps[:, 1], rtol=1e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("kw", [dict(), dict(lsepar_pola=False), dict(lsepar_pola=False, lsepar_contrib=False, dust="pascucci"),
                                dict(aniso_method=2, lsepar_pola=False)])
def test_device_source_function_equals_the_restatement(kw):
    from mcfost_amd.engine import Engine
    m = sed_model(M.small(RT_n_incl=3, **kw), n_thermal=20000)
    o = Oracle(m, 1000)
    e = Engine(m, 1000)
    T = m.extra["Tdust"]
    for lam, ibin in ((5, 1), (12, 3)):
        I, Istar = _field(m, seed=lam)
        Ed = m.extra["E_disk"][lam - 1]
        want, want_s = o.init_dust_source_fct2(lam, ibin, I, Istar, T, 2.0e6, Ed)
        got, got_s = e.init_dust_source_fct2(lam, ibin, I, Istar, T, 2.0e6, Ed)
        # sums in the reference's order and types; sin / cos / atan2f / sqrtf differ in the last place of a default real
        assert np.allclose(got, want, rtol=3e-6, atol=3e-6 * np.abs(want).max(axis=(0, 1, 2))), np.abs(got - want).max()
        assert np.allclose(got_s, want_s, rtol=3e-6, atol=3e-6 * np.abs(want_s).max(axis=(0, 1, 2)))
    e.close()


@pytest.mark.gpu
def test_source_function_from_the_packet_loop_without_leaving_the_device():
    """mcgpu_run_mono(rt1 = 2) leaves I_spec / I_spec_star in HBM, mcgpu_rt2_source reads them there: the same source
    function as the restatement gives on the fetched arrays; and its time at the size of ref4.1."""
    from mcfost_amd.engine import Engine
    m = sed_model(M.small(RT_n_incl=3), n_thermal=20000)
    e = Engine(m, 1e5)
    lam = 9
    a = e.run_mono(lam, 60, seed=4, n_chunks=16, rt2=(15, 15))
    T, Ed = m.extra["Tdust"], m.extra["E_disk"][lam - 1]
    got, got_s = e.init_dust_source_fct2(lam, 2, None, None, T, a["n_sent"][lam - 1], Ed)
    e.close()
    want, want_s = Oracle(m, 1000).init_dust_source_fct2(lam, 2, a["I_spec"], a["I_spec_star"], T, a["n_sent"][lam - 1], Ed)
    assert np.allclose(got, want, rtol=3e-6, atol=3e-6 * np.abs(want).max(axis=(0, 1, 2)))
    assert np.allclose(got_s, want_s, rtol=3e-6, atol=3e-6 * np.abs(want_s).max(axis=(0, 1, 2)))
    assert (got[..., 0] > 0).any() and (got_s[..., 0] > 0).any()
    big = M.build_model(M.ref41())
    e = Engine(big, 1e5)
    I, Istar = _field(big)
    e.init_dust_source_fct2(10, 1, I, Istar, np.full(big.n_cells, 50.0, np.float32), 1e7, 1.0)
    ms = e.last_rt2_ms
    e.init_dust_source_fct2(10, 1, None, None, np.full(big.n_cells, 50.0, np.float32), 1e7, 1.0)
    print("init_dust_source_fct2 at 7000 cells (15 x 15 bins, 15 + 1000 directions): %.2f ms (first call %.2f ms)" % (e.last_rt2_ms, ms))
    assert e.last_rt2_ms < 200.0
    e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("kw", [dict(), dict(lsepar_pola=False, lsepar_contrib=False)])
def test_method2_ray_tracing_equals_the_restatement(kw):
    """Ray tracing method 2 end to end on the device: packet loop with I_spec deposits -> source function -> dust_map's SED
    sampling and an image with the interpolating dust_source_fct, against the oracle's restatement on the same source
    function; and against method 1 of the same Monte Carlo (two estimators of the same scattered light)."""
    from mcfost_amd.engine import Engine
    m = sed_model(M.small(RT_n_incl=3, **kw), n_thermal=50000)
    o = Oracle(m, 1000)
    e = Engine(m, 1e5)
    T = m.extra["Tdust"]
    for lam in (4, 12):
        Ed = m.extra["E_disk"][lam - 1]
        a = e.run_mono(lam, 400, seed=6, n_chunks=32, rt2=(15, 15))
        ns = a["n_sent"][lam - 1]
        for ibin in (1, 3):
            eps, eps_s = e.init_dust_source_fct2(lam, ibin, None, None, T, ns, Ed)
            got, ms = e.rt2_dust_map_sed(lam, T, ns, Ed)
            want = o.rt2_dust_map_sed(lam, ibin, eps, eps_s, T, ns, Ed, n_threads=8)
            # (I and its contributions to 1e-6; Q, U are sums with cancellation of default-real cosf / sinf products)
            assert (want[0] > 0) and np.allclose(got, want, rtol=1e-6, atol=1e-6 * abs(want[0]))
            assert np.allclose(got[[0] + list(range(4 if len(got) > 4 else 1, len(got)))], want[[0] + list(range(4 if len(got) > 4 else 1, len(got)))], rtol=1e-6, atol=1e-12 * abs(want[0]))
            img, n_rays, _ = e.rt2_dust_map_image(lam, T, ns, Ed, 21, 21, 2.2 * m.cfg.rout)
            wimg, wn = o.rt2_dust_map_image(lam, ibin, eps, eps_s, T, ns, Ed, 21, 21, 2.2 * m.cfg.rout, n_threads=8)
            assert n_rays == wn
            # (default-real source function with cosf / sinf in interpolate_Stokes_QU: last-place differences per pixel)
            assert np.allclose(img, wimg, rtol=2e-5, atol=1e-6 * np.abs(wimg).max()), np.abs(img - wimg).max() / np.abs(wimg).max()
        # method 1 on the same model: the two ray tracers see the same disc (Monte Carlo noise and the coarse 15 x 15
        # direction bins of method 2 between them)
        b = e.run_mono(lam, 400, seed=7, n_chunks=32)
        rt1, _ = e.dust_map_sed(lam, T, b["n_sent"][lam - 1], Ed)
        eps, eps_s = None, None
        e.run_mono(lam, 400, seed=6, n_chunks=32, rt2=(15, 15))
        for ibin in (1, 3):
            e.init_dust_source_fct2(lam, ibin, None, None, T, ns, Ed)
            rt2, _ = e.rt2_dust_map_sed(lam, T, ns, Ed)
            assert abs(rt2[0] / rt1[ibin - 1, 0] - 1.0) < 0.25, (lam, ibin, rt2[0], rt1[ibin - 1, 0])
    e.close()
