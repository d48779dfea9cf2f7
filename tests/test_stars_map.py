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


@pytest.mark.parametrize("kw", [dict(), dict(n_rad=12, nz=6, n_az=8, l3D=True)])
def test_emulated_kernel_against_the_oracle(emu, kw):   # noqa: F811
    from oracle.binding import _RtOpts
    from oracle.binding import _a, _p
    m = sed_model(M.small(RT_imax=90.0, RT_n_incl=5, **kw), n_thermal=20000)
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
@pytest.mark.parametrize("kw", [dict(), dict(n_rad=12, nz=6, n_az=8, l3D=True)])
def test_device_against_the_oracle(kw):
    from mcfost_amd.engine import Engine
    m = sed_model(M.small(RT_imax=90.0, RT_n_incl=5, **kw), n_thermal=20000)
    o = Oracle(m, 1000)
    e = Engine(m, 1000)
    flux = np.array([2.5])
    for lam in (3, int(np.argmin(np.abs(m.lam - 1.0))) + 1, m.n_lambda):
        want = o.stars_map_sed(lam, flux, seed=7, ang_disque=20.0)
        got = e.stars_map_sed(lam, flux, seed=7, ang_disque=20.0)
        assert np.allclose(got, want, rtol=1e-5, atol=1e-12 * flux[0]), (lam, got, want)
    e.close()
