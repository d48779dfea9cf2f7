"""compute_stars_map for the SED (dust_transfer.f90:1604-1854): the stars' term of the ray-traced SED.
PARITY UNPINNED (dust_transfer.f90 cannot be built here and the reference's ray positions come from SPRNG): the
oracle restates the routine, known answers pin it, the device is held to the oracle."""
import ctypes as C

import numpy as np
import pytest

from mcfost_amd.host import model as M
from oracle import Oracle
from helpers import sed_model
from test_kernel_emulation import emu  # noqa: F401


def test_known_answers():
    m = sed_model(M.small(RT_imax=90.0, RT_n_incl=7), n_thermal=20000)
    o = Oracle(m, 1000)
    lam_thin = m.n_lambda            # longest wavelength: the disk is transparent
    lam_thick = int(np.argmin(np.abs(m.lam - 0.5))) + 1
    flux = np.array([3.0])
    thin, thick = o.stars_map_sed(lam_thin, flux), o.stars_map_sed(lam_thick, flux)
    incl = np.degrees(np.arccos(np.clip(m.rt["tab_w_rt"], -1, 1)))
    assert thin.shape == (m.rt["RT_n_incl"] * m.rt["RT_n_az"],)
    # no dust on the way (the observers above the disk): sum(exp(0) cos) / sum(cos) = 1; less extinction at 3 mm than at 0.5 um
    assert np.all(thin <= flux[0] * (1 + 1e-6)) and np.allclose(thin[incl < 50], flux[0], rtol=1e-6)
    assert np.all(thin >= thick - 1e-9)
    # optically thick wavelength: the pole-on observer sees the star, the edge-on observer does not
    assert thick[np.argmin(incl)] > 0.5 * flux[0] and thick[np.argmax(incl)] < 1e-3 * flux[0]
    assert np.all(np.diff(thick[np.argsort(incl)]) <= 1e-9 * flux[0])          # monotone in inclination
    # linear in the star's flux, independent of the seed to the noise of 1024 rays
    assert np.allclose(o.stars_map_sed(lam_thick, 2 * flux), 2 * thick, rtol=1e-12)
    other = o.stars_map_sed(lam_thick, flux, seed=99)
    sel = thick > 1e-3 * flux[0]
    assert np.allclose(other[sel], thick[sel], rtol=0.05)


@pytest.mark.parametrize("kw", [dict(), dict(n_rad=12, nz=6, n_az=8, l3D=True), dict(grid_type=2), dict(voronoi_sites=1200)])
def test_emulated_kernel_against_the_oracle(emu, kw):   # noqa: F811
    from oracle.binding import _RtOpts
    from oracle.binding import _a, _p
    kw = dict(kw)
    sites = kw.pop("voronoi_sites", 0)   # (Voronoi: index_cell_voronoi of the point of the star's disc, then cross_Voronoi_cell)
    m = sed_model(M.small(RT_imax=90.0, RT_n_incl=5, **kw), n_thermal=20000, voronoi_sites=sites)
    o = Oracle(m, 1000)
    flux = np.array([1.0])
    for lam in (3, int(np.argmin(np.abs(m.lam - 1.0))) + 1, m.n_lambda):
        for ang in (0.0, 30.0):
            want = o.stars_map_sed(lam, flux, seed=5, ang_disque=ang)
            az = _a(m.rt["tab_RT_az"], np.float32)
            opts = _RtOpts(int(lam), float(m.lam[lam - 1]), 1.0, 1.0, float(m.cfg.distance), ang, 0, 100.0,
                           float(m.cfg.rin), float(m.cfg.rout), _p(az, C.c_float), 1)
            got = np.zeros_like(want)
            rc = emu.emu_stars_map_sed(C.byref(o.cm), C.byref(opts), C.c_uint64(5), _p(flux, C.c_double), _p(got, C.c_double))
            assert rc == 0
            assert np.allclose(got, want, rtol=1e-6, atol=1e-12), (lam, ang, got, want)


@pytest.mark.gpu
@pytest.mark.parametrize("kw", [dict(), dict(n_rad=12, nz=6, n_az=8, l3D=True), dict(grid_type=2), dict(voronoi_sites=3000)])
def test_device_against_the_oracle(kw):
    from mcfost_amd.engine import Engine
    kw = dict(kw)
    sites = kw.pop("voronoi_sites", 0)
    m = sed_model(M.small(RT_imax=90.0, RT_n_incl=5, **kw), n_thermal=20000, voronoi_sites=sites)
    o = Oracle(m, 1000)
    e = Engine(m, 1000)
    flux = np.array([2.5])
    for lam in (3, int(np.argmin(np.abs(m.lam - 1.0))) + 1, m.n_lambda):
        want = o.stars_map_sed(lam, flux, seed=7, ang_disque=20.0)
        got = e.stars_map_sed(lam, flux, seed=7, ang_disque=20.0)
        assert np.allclose(got, want, rtol=1e-5, atol=1e-12 * flux[0]), (lam, got, want)
    e.close()


def _ld_table(u=0.6, pmax=0.1, n=21):
    mu = np.linspace(0.0, 1.0, n).astype(np.float32)
    return mu, (1.0 - u * (1.0 - mu)).astype(np.float32), (pmax * (1.0 - mu) ** 2).astype(np.float32)


def test_stars_image_known_answers():
    """compute_stars_map with resolved discs (oracle): the map of a star sums to its flux where the path is thin, the disc
    has the projected radius, limb darkening lowers the limb and keeps the sum, the polarised maps are tangential and
    cancel in the sum, an unresolved star sits in one pixel at the projected position."""
    m = M.build_model(M.small(RT_n_incl=3))
    o = Oracle(m, 1000)
    lam = m.n_lambda                      # longest wavelength: optically thin
    flux = np.array([2.5])
    # unresolved: 5 AU map, 65 pixels (odd: the star in the centre pixel)
    maps, pos = o.stars_map_image(lam, flux, 65, 65, 5.0, seed=3)
    assert maps.shape == (3, 1, 65, 65)
    face_on = int(np.argmin(m.rt["tab_RT_incl"])) if "tab_RT_incl" in m.rt else 0
    for q in range(3):
        assert maps[q, 0, 32, 32] == maps[q].sum() and 0.0 < maps[q].sum() <= flux[0] * (1 + 1e-6)
    assert maps[face_on].sum() > 0.9 * flux[0]          # thin towards the pole at 3 mm: the whole flux arrives
    assert np.allclose(pos, 0.0)
    # resolved: a map so small that the disc spans ~20 pixels
    rs = m.cfg.R_star * 0.00465047                      # Rsun -> AU
    size = 65 * rs / 10.0
    maps, _ = o.stars_map_image(lam, flux, 65, 65, size, seed=3)
    disc = maps[0, 0] > 0
    yy, xx = np.nonzero(disc)
    rad = np.hypot(xx - 32, yy - 32)
    assert 9.0 < rad.max() < 11.0 and disc.sum() > 250            # the projected disc: radius 10 pixels
    assert 0.5 * flux[0] < maps[face_on].sum() <= flux[0] * (1 + 1e-6)
    mu, ld, pld = _ld_table()
    dark, _ = o.stars_map_image(lam, flux, 65, 65, size, seed=3, limb_darkening=(mu, ld))
    assert np.isclose(dark[0].sum(), maps[0].sum(), rtol=1e-3)   # the normalisation carries the limb darkening (:1826)
    centre = lambda a: a[0, 0, 30:35, 30:35].mean()
    ring = lambda a: a[0, 0][(rad.max() - 2 < np.hypot(*np.meshgrid(np.arange(65) - 32, np.arange(65) - 32)))
                             & (np.hypot(*np.meshgrid(np.arange(65) - 32, np.arange(65) - 32)) < rad.max() - 0.5)].mean()
    assert centre(dark) / ring(dark) > 1.15 * centre(maps) / ring(maps)
    pol, _ = o.stars_map_image(lam, flux, 65, 65, size, seed=3, limb_darkening=(mu, ld, pld))
    assert pol.shape[1] == 3 and np.allclose(pol[:, 0], dark[:, 0])
    assert np.abs(pol[0, 1]).max() > 0 and abs(pol[0, 1].sum()) < 0.05 * np.abs(pol[0, 1]).sum()   # Q cancels over the disc


@pytest.mark.gpu
def test_device_stars_image_equals_the_oracle():
    from mcfost_amd.engine import Engine
    mu, ld, pld = _ld_table()
    for cfg, sites in ((M.small(RT_n_incl=3), 0), (M.small(n_rad=10, nz=5, n_az=6, l3D=True, RT_n_incl=2), 0),
                       (M.small(RT_n_incl=2), 2000)):
        m = M.build_voronoi_model(cfg, sites, seed=3) if sites else M.build_model(cfg)
        o = Oracle(m, 1000)
        e = Engine(m, 1000)
        rs = m.cfg.R_star * 0.00465047
        for lam in (3, m.n_lambda):
            for npix, size, limb in ((65, 65 * rs / 10.0, None), (64, 64 * rs / 6.0, (mu, ld, pld)), (33, 5.0, (mu, ld))):
                want, wpos = o.stars_map_image(lam, np.array([1.7]), npix, npix, size, seed=5, limb_darkening=limb)
                got, gpos = e.stars_map_image(lam, np.array([1.7]), npix, npix, size, seed=5, limb_darkening=limb)
                assert got.shape == want.shape and np.allclose(gpos, wpos, atol=1e-12)
                assert np.array_equal(got != 0, want != 0)                     # the same rays in the same pixels
                # (expf / sincos / atan2f, summation order; cos 2 phi of a pixel on the diagonal is 4e-8 or 0: absolute floor)
                assert np.allclose(got, want, rtol=2e-5, atol=1e-6 * np.abs(want).max())
        e.close()
