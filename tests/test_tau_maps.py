"""The ray tracer's optical-depth maps: compute_tau_map (dust_transfer.f90:2114-2210, option -tau_map) and
compute_tau_surface_map (:2006-2110, option -tau_surface).
PARITY UNPINNED (dust_transfer.f90 cannot be built here): the oracle restates the two routines on top of its
move_to_grid / cross_cell / physical_length, known answers pin it, the device kernel is held to the oracle -- on the CPU
through the lane emulation, on the GPU through the C-ABI."""
import ctypes as C

import numpy as np
import pytest

from mcfost_amd.host import model as M
from oracle import Oracle
from helpers import sed_model
from test_kernel_emulation import emu  # noqa: F401


def _column_tau(m, lam):
    """2 * sum_j kappa * kappa_factor * dz per radius of a 2D cylindrical grid (both halves of the disk)."""
    g = m.grid
    n_rad, nz = m.cfg.n_rad, m.cfg.nz
    kf = np.asarray(m.kappa_factor, np.float64).reshape(nz, n_rad)            # icell = i + n_rad (j - 1)
    zl = np.asarray(g["z_lim"], np.float64).reshape(nz + 2, n_rad)[:nz + 1]   # z_lim(i, j), i fastest
    return 2.0 * m.kappa[lam - 1] * (kf * np.diff(zl, axis=0)).sum(axis=0)


def test_known_answers():
    m = sed_model(M.small(RT_imin=0.0, RT_imax=90.0, RT_n_incl=3), n_thermal=20000)
    o = Oracle(m, 1000)
    nRT = m.rt["RT_n_incl"] * m.rt["RT_n_az"]
    incl = np.degrees(np.arccos(np.clip(m.rt["tab_w_rt"], -1, 1)))
    q0 = int(np.argmin(incl))
    assert incl[q0] < 1e-3                                                    # a pole-on observer
    lam = int(np.argmin(np.abs(m.lam - 1.0))) + 1
    npix, size = 41, 2.2 * m.cfg.rout
    tm, sm = o.tau_maps(lam, npix, npix, size, tau=1.0)
    assert tm.shape == (nRT, npix, npix) and sm.shape == (3, nRT, npix, npix)
    # pole-on: a vertical ray crosses one column of the grid: tau = 2 kappa sum_j kappa_factor dz of its radius; 0 outside
    pix = size / npix
    c = (np.arange(npix) + 0.5) * pix - 0.5 * size
    rr = np.hypot(c[None, :], c[:, None])
    r_lim = np.asarray(m.grid["r_lim"], np.float64)
    col = _column_tau(m, lam)
    ri = np.searchsorted(r_lim, rr) - 1
    inside = (rr > r_lim[0]) & (rr < r_lim[-1])
    want = np.where(inside, col[np.clip(ri, 0, len(col) - 1)], 0.0)
    near_wall = np.min(np.abs(rr[..., None] - r_lim[None, None, :]), axis=-1) < 1e-9 * m.cfg.rout
    assert np.allclose(tm[q0][~near_wall], want[~near_wall], rtol=2e-6, atol=1e-30)
    assert np.all(tm[q0][~inside] == 0.0) and tm[q0].max() > 1.0              # thick at 1 um in the inner disk
    # the tau = 1 surface towards the pole-on observer: above the midplane, over the pixel (x, y of the pixel centre; the
    # image's x axis is the model's x for azimuth 0 ... up to the sign conventions of the plane's basis: radii compared),
    # and the optical depth from that point up to the observer is the tau asked for
    xs, ys, zs = sm[0][q0], sm[1][q0], sm[2][q0]
    reached = (tm[q0] > 1.0) & ~near_wall
    assert reached.sum() > 20
    assert np.all(zs[reached] > 0.0) and np.allclose(np.hypot(xs, ys)[reached], rr[reached], rtol=1e-5)
    none = tm[q0] < 1.0 - 1e-5
    assert np.all(xs[none] == 0.0) and np.all(ys[none] == 0.0) and np.all(zs[none] == 0.0)
    nz, n_rad = m.cfg.nz, m.cfg.n_rad
    kf = np.asarray(m.kappa_factor, np.float64).reshape(nz, n_rad)
    zl = np.asarray(m.grid["z_lim"], np.float64).reshape(nz + 2, n_rad)[:nz + 1]
    jj, ii = np.nonzero(reached)
    for a, b in list(zip(jj, ii))[:: max(1, len(jj) // 12)]:
        i = ri[a, b]
        z = float(zs[a, b])
        above = np.clip(zl[1:, i] - np.maximum(zl[:-1, i], z), 0.0, None)     # the part of each cell above the point
        assert np.isclose(m.kappa[lam - 1] * (kf[:, i] * above).sum(), 1.0, rtol=2e-4), (a, b)
    # a deeper surface lies lower; a surface nobody reaches is empty
    _, sm3 = o.tau_maps(lam, npix, npix, size, tau=3.0)
    both = reached & (tm[q0] > 3.0)
    assert both.sum() > 5 and np.all(sm3[2][q0][both] < zs[both])
    _, smx = o.tau_maps(lam, npix, npix, size, tau=1e9)
    assert not smx.any()
    # symmetric maps for the inclined observers (left-right mirror of the image), thicker towards edge-on through the centre line
    for q in range(nRT):
        assert np.allclose(tm[q], tm[q][:, ::-1], rtol=1e-4, atol=1e-6 * tm[q].max())
    mid = npix // 2
    order = np.argsort(incl)
    centre_tau = np.array([tm[q][mid, mid + 6] for q in order])
    assert np.all(np.diff(centre_tau) > 0)
    # the long-wavelength map is thinner everywhere
    thin, _ = o.tau_maps(m.n_lambda, npix, npix, size, surface=False)
    assert np.all(thin <= tm + 1e-12) and thin.max() < tm.max() * 0.1
    # zoom = 2 is the inner half of the map at twice the resolution: its centre pixels sample the same columns
    z2, _ = o.tau_maps(lam, npix, npix, size, zoom=2.0, surface=False)
    assert np.isclose(z2[q0][mid, mid + 8], tm[q0][mid, mid + 4], rtol=1e-5)


def test_dark_zone_hands_back_the_cell_before():
    """physical_length's mirror (optical_depth.f90:104-112) in the surface map: a ray that meets a flagged cell ends at the
    entry point of the cell before it."""
    m = sed_model(M.small(RT_imin=0.0, RT_imax=60.0, RT_n_incl=2), n_thermal=20000)
    lam = int(np.argmin(np.abs(m.lam - 1.0))) + 1
    n_rad, nz = m.cfg.n_rad, m.cfg.nz
    dark = np.zeros(m.n_cells, np.uint8)
    dark.reshape(nz, n_rad)[0, n_rad // 3: 2 * n_rad // 3] = 1                 # midplane cells of the middle radii
    m.l_dark_zone = dark
    o = Oracle(m, 1000)
    npix, size = 61, 2.2 * m.cfg.rout
    _, sm = o.tau_maps(lam, npix, npix, size, tau=1e6)                          # nobody reaches tau = 1e6: only the mirror answers
    incl = np.degrees(np.arccos(np.clip(m.rt["tab_w_rt"], -1, 1)))
    q0 = int(np.argmin(incl))
    zs = sm[2][q0]
    hit = zs != 0.0
    assert hit.sum() > 4
    zl = np.asarray(m.grid["z_lim"], np.float64).reshape(nz + 2, n_rad)[:nz + 1]
    # pole-on: the cell before the flagged midplane cell is j = 2, entered from above through z_lim(i, 3)
    c = (np.arange(npix) + 0.5) * size / npix - 0.5 * size
    rr = np.hypot(c[None, :], c[:, None])
    ri = np.searchsorted(np.asarray(m.grid["r_lim"], np.float64), rr) - 1
    assert np.all((ri[hit] >= n_rad // 3) & (ri[hit] < 2 * n_rad // 3))
    assert np.allclose(zs[hit], zl[2, ri[hit]], rtol=1e-5)


GRIDS = [dict(), dict(n_rad=12, nz=6, n_az=8, l3D=True), dict(grid_type=2), dict(grid_type=2, n_rad=10, nz=6, n_az=6, l3D=True),
         dict(voronoi_sites=1200)]


def _compare(got, want, what):
    tm, sm = got
    wt, ws = want
    assert np.array_equal(tm != 0, wt != 0), what
    assert np.allclose(tm, wt, rtol=2e-6, atol=0), what
    assert np.array_equal((sm != 0).any(axis=0), (ws != 0).any(axis=0)), what       # the same rays reach the surface
    assert np.allclose(sm, ws, rtol=2e-6, atol=1e-6 * np.abs(ws).max()), what       # (a coordinate of ~0 is rounding)


@pytest.mark.parametrize("kw", GRIDS)
def test_emulated_kernel_against_the_oracle(emu, kw):   # noqa: F811
    from oracle.binding import _RtOpts, _a, _p
    kw = dict(kw)
    sites = kw.pop("voronoi_sites", 0)
    m = sed_model(M.small(RT_imax=90.0, RT_n_incl=3, RT_n_az=2 if kw.get("l3D") else 1, **kw), n_thermal=20000, voronoi_sites=sites)
    o = Oracle(m, 1000)
    nRT = m.rt["RT_n_incl"] * m.rt["RT_n_az"]
    az = _a(m.rt["tab_RT_az"], np.float32)
    for lam, npx, npy, ang, tau in ((3, 24, 17, 0.0, 1.0), (int(np.argmin(np.abs(m.lam - 1.0))) + 1, 19, 19, 30.0, 0.3)):
        size = 2.4 * m.cfg.rout
        want = o.tau_maps(lam, npx, npy, size, zoom=1.3, tau=tau, ang_disque=ang)
        opts = _RtOpts(int(lam), float(m.lam[lam - 1]), 1.0, 1.0, float(m.cfg.distance), ang, 0, 100.0,
                       float(m.cfg.rin), float(m.cfg.rout), _p(az, C.c_float), 1)
        tm = np.zeros((nRT, npy, npx), np.float32)
        sm = np.zeros((3, nRT, npy, npx), np.float32)
        rc = emu.emu_tau_maps(C.byref(o.cm), C.byref(opts), C.c_int(npx), C.c_int(npy), C.c_double(size), C.c_double(1.3),
                              C.c_float(tau), _p(tm, C.c_float), _p(sm, C.c_float))
        assert rc == 0
        assert (want[0] > 0).sum() > 20 and (want[1][2] != 0).sum() > 5
        _compare((tm, sm), want, (kw, lam))


@pytest.mark.gpu
@pytest.mark.parametrize("kw", GRIDS[:4] + [dict(voronoi_sites=3000)])
def test_device_against_the_oracle(kw):
    from mcfost_amd.engine import Engine
    kw = dict(kw)
    sites = kw.pop("voronoi_sites", 0)
    m = sed_model(M.small(RT_imax=90.0, RT_n_incl=3, RT_n_az=2 if kw.get("l3D") else 1, **kw), n_thermal=20000, voronoi_sites=sites)
    o = Oracle(m, 1000)
    e = Engine(m, 1000)
    for lam, npx, npy, ang, tau in ((3, 40, 33, 0.0, 1.0), (int(np.argmin(np.abs(m.lam - 1.0))) + 1, 37, 37, 30.0, 0.3)):
        size = 2.4 * m.cfg.rout
        want = o.tau_maps(lam, npx, npy, size, zoom=1.3, tau=tau, ang_disque=ang)
        tm, sm, ms = e.tau_maps(lam, npx, npy, size, zoom=1.3, tau=tau, ang_disque=ang)
        _compare((tm, sm), want, (kw, lam))
        only, none, _ = e.tau_maps(lam, npx, npy, size, zoom=1.3, tau=tau, ang_disque=ang, surface=False)
        assert none is None and np.array_equal(only, tm)
    # bad arguments are refused
    from mcfost_amd.engine import McgpuError
    with pytest.raises(McgpuError):
        e.tau_maps(3, 0, 5, 1.0)
    with pytest.raises(McgpuError):
        e.tau_maps(3, 5, 5, 1.0, tau=0.0)
    e.close()
