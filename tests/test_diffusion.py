"""The 1+1D diffusion fill of the dark zone (``Temp_approx_diffusion_vertical``, diffusion.f90:292-374) and the
extent of the zone it works on (``define_dark_zone`` steps 1-3, optical_depth.f90:1459-1500, 1621-1628).

PARITY UNPINNED: module ``diffusion`` pulls dust_prop / thermal_emission / the parameter file reader and cannot be
built here (oracle/ref_build/README), and the reference holds no fixture for it.  The oracle restates the routine
line by line; these tests pin it with known answers (the steady state of the scheme, the maximum principle, the cells
it may touch) and the device against the oracle."""
import numpy as np
import pytest

from mcfost_amd.host import model as M
from oracle import Oracle
from test_kernel_emulation import emu  # noqa: F401  (the lane emulator's fixture)


def thick_disk():
    cfg = M.small(n_rad=30, nz=20, dust_mass=1e-2)
    m = M.build_model(cfg)
    lam = int(np.argmin(np.abs(m.lam - 0.81))) + 1
    return m, lam


def test_dark_zone_extent_known_answer():
    m, lam = thick_disk()
    o = Oracle(m, 1000)
    ri_in, ri_out, zj = o.dark_zone_extent(lam, 1500.0)
    g, n_rad, nz = m.grid, m.grid["n_rad"], m.grid["nz"]
    kap = m.kappa[lam - 1] * np.asarray(m.kappa_factor).reshape(nz, n_rad)  # [j, i]
    dr = np.diff(g["r_lim"])
    tau_r = np.cumsum(kap[0] * dr)
    assert ri_in == 1 + int(np.argmax(tau_r > 1500.0)) and ri_in >= 2          # étape 1
    tau_r_out = np.cumsum((kap[0] * dr)[::-1])
    assert ri_out == min(n_rad - int(np.argmax(tau_r_out > 1500.0)), n_rad - 1)  # étape 2
    z_lim = np.asarray(g["z_lim"]).reshape(nz + 2, n_rad)[:nz + 1]
    for i in range(ri_in, ri_out + 1):                                          # étape 3: from the top down
        tau_z = np.cumsum((kap[:, i - 1] * np.diff(z_lim[:, i - 1]))[::-1])
        want = nz - int(np.argmax(tau_z > 1500.0)) if tau_z[-1] > 1500.0 else 0
        assert zj[i - 1] == want, (i, zj[i - 1], want)
    assert np.all(zj[:ri_in - 1] == zj[ri_in - 1]) and np.all(zj[ri_out:] == zj[ri_out - 1])  # :1621-1628
    assert M.dark_zone_extent(m, lam, 1500.0)[:2] == (ri_in, ri_out)
    assert np.array_equal(M.dark_zone_extent(m, lam, 1500.0)[2], zj)


def surface_temperature(m):
    """A smooth stand-in for the Monte Carlo temperature: falls with radius and rises with height."""
    n_rad, nz = m.grid["n_rad"], m.grid["nz"]
    i = np.arange(n_rad)[None, :]
    j = np.arange(nz)[:, None]
    return (120.0 * (1.0 + i) ** -0.4 * (1.0 + 0.08 * j)).astype(np.float32).ravel()


def test_fill_known_answers():
    m, lam = thick_disk()
    o = Oracle(m, 1000)
    ri_in, ri_out, zj = o.dark_zone_extent(lam, 1500.0)
    n_rad, nz = m.grid["n_rad"], m.grid["nz"]
    T0 = surface_temperature(m)
    T1, n_it = o.temp_approx_diffusion_vertical(T0, ri_in, ri_out, zj)
    A, B = T0.reshape(nz, n_rad), T1.reshape(nz, n_rad)
    i_lo, i_hi = max(ri_in - 3, 3), min(ri_out + 3, n_rad - 2)
    assert n_it > 0
    for i in range(1, n_rad + 1):
        top = zj[i - 1] + 3
        col0, col1 = A[:, i - 1], B[:, i - 1]
        if i_lo <= i <= i_hi:
            # above the zone (+ delta_cell_dark_zone) nothing moves; inside, the steady state of a column with a
            # no-flux midplane and the Monte Carlo temperature on top is isothermal at that temperature
            assert np.array_equal(col1[top:], col0[top:])
            assert np.allclose(col1[:top], col0[top], rtol=2e-3), (i, col1[:top + 1])
            assert np.all(np.diff(col1[:top + 1]) >= -1e-6 * col0[top])        # monotone towards the boundary
        else:
            # columns outside [ri_in-3, ri_out+3] (and the first two / last two radii) are only cleaned
            clean = np.zeros(nz, bool)
            if ri_in <= i <= ri_out:
                clean[:zj[i - 1]] = True
            assert np.array_equal(col1[~clean], col0[~clean]) and np.all(col1[clean] == np.float32(m.cfg.T_min))


def test_fill_is_a_fixed_point_of_an_isothermal_column():
    """DensE uniform -> every difference in the stencil is exactly 0 -> one iteration, temperature unchanged to the
    rounding of T -> T^4 -> T."""
    m, lam = thick_disk()
    o = Oracle(m, 1000)
    ri_in, ri_out, zj = o.dark_zone_extent(lam, 1500.0)
    T0 = np.full(m.n_cells, 40.0, np.float32)
    T1, n_it = o.temp_approx_diffusion_vertical(T0, ri_in, ri_out, np.zeros_like(zj))  # nothing cleaned
    assert np.allclose(T1, 40.0, rtol=1e-6)
    assert n_it == min(ri_out + 3, m.grid["n_rad"] - 2) - max(ri_in - 3, 3) + 1        # one step per column


@pytest.mark.gpu
def test_device_fill_against_the_oracle():
    from mcfost_amd.engine import Engine
    m, lam = thick_disk()
    o = Oracle(m, 1000)
    ri_in, ri_out, zj = o.dark_zone_extent(lam, 1500.0)
    T0 = surface_temperature(m)
    want, it_o = o.temp_approx_diffusion_vertical(T0, ri_in, ri_out, zj)
    e = Engine(m, 1000)
    got, it_d = e.temp_approx_diffusion_vertical(T0, ri_in, ri_out, zj)
    assert abs(it_d - it_o) <= 0.01 * it_o + 2, (it_d, it_o)   # exp / pow differ by an ulp: the stopping step may move
    assert np.allclose(got, want, rtol=2e-5), np.abs(got / want - 1).max()
    changed = got != T0
    assert changed.any() and np.array_equal(changed, want != T0)
    e.close()


@pytest.mark.gpu
def test_device_fill_rejects_what_the_reference_does_not_do():
    from mcfost_amd.engine import Engine, McgpuError
    m3 = M.build_model(M.small(n_rad=12, nz=6, n_az=8, l3D=True))
    e = Engine(m3, 1000)
    with pytest.raises(McgpuError):
        e.temp_approx_diffusion_vertical(np.ones(m3.n_cells, np.float32), 3, 5, np.zeros(12, np.int32))
    e.close()


@pytest.mark.gpu
def test_dark_zone_and_diffusion_fill_against_brute_force_at_scale():
    """What the reference does with an optically thick midplane, end to end on the device: define_dark_zone's cells
    (tau = 1500 at the first wavelength beyond 0.81 um) mirror the packets, Temp_finale leaves T_min there, the
    diffusion fill restores the temperature -- against the brute-force run of the same disk (the ref4.1 grid with 10x
    the dust mass: 301 dark cells) through the reference's gate: p75 of |dT|/T below 5 % inside the zone and outside."""
    from mcfost_amd.engine import Engine
    cfg = M.ref41()
    cfg.dust_mass = 1e-2
    m = M.build_model(cfg)
    lam = int(np.searchsorted(m.lam, 0.81)) + 1
    n = 20_000_000
    e0 = Engine(m, n)
    dz, ri_in, ri_out, zj = e0.define_dark_zone(lam, 1500.0)       # the device's own define_dark_zone
    assert dz.sum() > 100 and ri_in >= 2
    assert (ri_in, ri_out) == M.dark_zone_extent(m, lam, 1500.0)[:2]
    r0 = e0.run_thermal(n, seed=3)
    T0 = e0.temp_finale(r0["E_abs"])
    e0.close()
    m1 = M.build_model(cfg)
    m1.l_dark_zone = dz
    e1 = Engine(m1, n)
    r1 = e1.run_thermal(n, seed=4)
    T1 = e1.temp_finale(r1["E_abs"])
    T2, n_it = e1.temp_approx_diffusion_vertical(T1, ri_in, ri_out, zj)
    e1.close()
    d = dz != 0
    assert np.all(T1[d] == np.float32(cfg.T_min)) and r1["counters"]["dark_mirrors"] > 0
    rel = T2[d] / T0[d] - 1.0
    out = (~d) & (T0 > 1.2 * cfg.T_min)
    p75_in, p75_out = np.percentile(np.abs(rel), 75), np.percentile(np.abs(T2[out] / T0[out] - 1.0), 75)
    print("dark zone %d cells: fill vs brute force p50 %.4f, p75 %.4f, max %.3f; outside p75 %.4f; kernel %.0f -> %.0f ms, "
          "%d fill iterations" % (d.sum(), np.median(rel), p75_in, np.abs(rel).max(), p75_out, r0["kernel_ms"], r1["kernel_ms"], n_it))
    assert p75_in < 0.05 and abs(np.median(rel)) < 0.02 and p75_out < 0.01
    assert r1["kernel_ms"] < 0.5 * r0["kernel_ms"]


def _thick_ref_grid():
    cfg = M.small(n_rad=30, nz=20, dust_mass=3e-2)
    m = M.build_model(cfg)
    lam = int(np.searchsorted(m.lam, 0.81)) + 1
    return m, lam


def test_emulated_dark_zone_rays_against_the_oracle(emu):   # noqa: F811
    """Step 4 of define_dark_zone on the device source (one ray per thread) against the oracle's sequential version:
    the same dark cells."""
    import ctypes as C
    from oracle.binding import _a, _p
    m, lam = _thick_ref_grid()
    o = Oracle(m, 1000)
    want = o.define_dark_zone(lam, 1500.0)
    ri_in, ri_out, zj_ext = o.dark_zone_extent(lam, 1500.0)
    assert want.sum() > 20
    # the candidate heights before the extension of :1621-1628
    zj = zj_ext.copy()
    zj[:ri_in - 1] = 0
    zj[ri_out:] = 0
    g = m.grid
    n_rad, nz = g["n_rad"], g["nz"]
    got = np.zeros(m.n_cells, np.uint8)
    # passes with the previous pass's flags until nothing changes, as mcgpu_define_dark_zone runs them: the reference's
    # loop reads the flags of the columns it has already decided (physical_length's mirror)
    for n_pass in range(1, n_rad + 2):
        flag = np.zeros(m.n_cells, np.uint8)
        rc = emu.emu_dark_zone_rays(C.byref(o.cm), C.c_int(lam), C.c_double(1500.0), C.c_int(max(ri_in, 2)), C.c_int(ri_out),
                                    _p(_a(zj, np.int32), C.c_int), _p(_a(g["r_grid"], np.float64), C.c_double),
                                    _p(_a(g["z_grid"], np.float64), C.c_double), _p(got, C.c_ubyte), _p(flag, C.c_ubyte))
        assert rc == 0
        nxt = np.zeros(m.n_cells, np.uint8)
        F = flag.reshape(nz, n_rad)
        for i in range(max(ri_in, 2), ri_out + 1):
            for j in range(zj[i - 1], 0, -1):
                if F[j - 1, i - 1]:
                    nxt.reshape(nz, n_rad)[:j, i - 1] = 1
                    break
        same = np.array_equal(nxt, got)
        got = nxt
        if same:
            break
    assert n_pass <= 3 and np.array_equal(got, want)


@pytest.mark.gpu
def test_device_define_dark_zone_equals_the_oracle():
    from mcfost_amd.engine import Engine
    for m, lam in (_thick_ref_grid(), thick_disk()):
        o = Oracle(m, 1000)
        want = o.define_dark_zone(lam, 1500.0)
        ext = o.dark_zone_extent(lam, 1500.0)
        e = Engine(m, 1000)
        dz, ri_in, ri_out, zj = e.define_dark_zone(lam, 1500.0)
        e.close()
        assert np.array_equal(dz, want) and (ri_in, ri_out) == ext[:2] and np.array_equal(zj, ext[2])
    cfg = M.ref41()
    cfg.dust_mass = 1e-2
    m = M.build_model(cfg)
    lam = int(np.searchsorted(m.lam, 0.81)) + 1
    e = Engine(m, 1000)
    dz, *_ = e.define_dark_zone(lam, 1500.0)
    e.close()
    assert np.array_equal(dz, Oracle(m, 1000).define_dark_zone(lam, 1500.0)) and dz.sum() == 301


@pytest.mark.gpu
def test_dark_zone_and_diffusion_fill_on_variable_dust():
    """lvariable_dust: define_dark_zone (kappa(p_icell, lambda) in the optical depths of steps 1-3 and in the test rays,
    optical_depth.f90:1454-1551) and the diffusion fill (setDiffusion_coeff's Rosseland sum with the cell's own kappa,
    diffusion.f90:34-60) on a settled thick disk: the device against the oracle, and the classes matter."""
    from mcfost_amd.engine import Engine
    m, lam = thick_disk()
    o1 = Oracle(m, 1000)
    dz1 = o1.define_dark_zone(lam, 1500.0)
    M.init_variable_dust(m, n_classes=5, slope=0.3)
    o = Oracle(m, 1000)
    want = o.define_dark_zone(lam, 1500.0)
    ext = o.dark_zone_extent(lam, 1500.0)
    assert want.sum() > 0 and not np.array_equal(want, dz1)            # the classes change the zone
    e = Engine(m, 1000)
    dz, ri_in, ri_out, zj = e.define_dark_zone(lam, 1500.0)
    assert np.array_equal(dz, want) and (ri_in, ri_out) == ext[:2] and np.array_equal(zj, ext[2])
    T0 = surface_temperature(m)
    want_T, it_o = o.temp_approx_diffusion_vertical(T0, ri_in, ri_out, zj)
    got_T, it_d = e.temp_approx_diffusion_vertical(T0, ri_in, ri_out, zj)
    e.close()
    assert abs(it_d - it_o) <= 0.01 * it_o + 2, (it_d, it_o)
    assert np.allclose(got_T, want_T, rtol=2e-5), np.abs(got_T / want_T - 1).max()
    assert (got_T != T0).any()
