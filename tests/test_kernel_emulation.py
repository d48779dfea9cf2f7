"""Kernel control flow on the CPU: tests/emu/emu_kernel.cpp compiles the DEVICE
source (mcfost_amd/csrc/mc_device.hip.h) for the host with a one-lane
emulation of the HIP builtins, WITH FMA contraction enabled like hipcc's
default, and the result is compared with the oracle in frozen-temperature
mode.  This is test infrastructure (a debugger for the state machine), not a
CPU path of the product; the real parity tests are tests/test_gpu_parity.py.
"""
import copy
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from mcfost_amd.host import model as M
from oracle.binding import N_COUNTERS
from oracle import Oracle
from helpers import sed_model, xI_close
from oracle.binding import _Opts, _p

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "emu", "emu_kernel.cpp")
LIB = os.path.join(HERE, "emu", "libemu_kernel.so")
DEV = os.path.join(os.path.dirname(HERE), "mcfost_amd", "csrc", "mc_device.hip.h")
DEV2 = os.path.join(HERE, "emu", "cross_cell_literal.h")
DEV3 = os.path.join(os.path.dirname(HERE), "mcfost_amd", "csrc", "mc_voronoi.hip.h")
DEV4 = os.path.join(os.path.dirname(HERE), "mcfost_amd", "csrc", "mc_mono.hip.h")
DEV5 = os.path.join(os.path.dirname(HERE), "mcfost_amd", "csrc", "mc_roles.hip.h")
DEV6 = os.path.join(os.path.dirname(HERE), "mcfost_amd", "csrc", "mc_raytrace.hip.h")
DEV7 = os.path.join(os.path.dirname(HERE), "mcfost_amd", "csrc", "mc_binned.hip.h")
DEV8 = os.path.join(os.path.dirname(HERE), "mcfost_amd", "csrc", "mc_tail.hip.h")
DEV9 = os.path.join(os.path.dirname(HERE), "mcfost_amd", "csrc", "mc_raytrace_voronoi.hip.h")
DEV10 = os.path.join(os.path.dirname(HERE), "mcfost_amd", "csrc", "mc_voronoi_pool.hip.h")


@pytest.fixture(scope="module")
def emu():
    if (not os.path.exists(LIB)) or os.path.getmtime(LIB) < max(os.path.getmtime(SRC), os.path.getmtime(DEV), os.path.getmtime(DEV2), os.path.getmtime(DEV3), os.path.getmtime(DEV4), os.path.getmtime(DEV5), os.path.getmtime(DEV6), os.path.getmtime(DEV7), os.path.getmtime(DEV8), os.path.getmtime(DEV9), os.path.getmtime(DEV10)):
        fma = ["-mfma"] if "fma" in open("/proc/cpuinfo").read() else []
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=fast"] + fma +
                              ["-o", LIB, SRC])
    return C.CDLL(LIB)


def emu_run(emu, orc, n, seed, first=0, frozen=True, prior=None):
    m = orc.model
    E = np.zeros(m.n_cells)
    sed = np.zeros((9, m.cfg.N_phi, m.cfg.N_thet, m.n_lambda))
    ns = np.zeros(m.n_lambda)
    cnt = np.zeros(N_COUNTERS, np.uint64)
    o = _Opts(seed, first, n, 1, int(frozen), 0, 1.0)
    rc = emu.emu_run_thermal(C.byref(orc.cm), C.byref(o), _p(prior, C.c_double) if prior is not None else None,
                             _p(E, C.c_double), _p(sed, C.c_double), _p(ns, C.c_double), _p(cnt, C.c_uint64))
    assert rc == 0, rc
    return dict(E_abs=E, sed=sed, n_sent=ns, counters=[int(c) for c in cnt])


def check(emu, m, n, seed, rtol=1e-9):
    orc = Oracle(m, n)
    prior = orc.run_thermal(2000, seed=1)["E_abs"]
    a = emu_run(emu, orc, n, seed, prior=prior)
    b = orc.run_thermal(n, seed=seed, frozen=True, E_prior=prior, n_threads=4)
    assert a["counters"] == list(b["counters"].values())
    assert np.array_equal(a["n_sent"], b["n_sent"]) and np.array_equal(a["sed"][4], b["sed"][4])
    assert np.allclose(a["E_abs"], b["E_abs"], rtol=rtol, atol=1e-11 * b["E_abs"].max())
    return a, b


@pytest.mark.parametrize("name", ["small2d", "small3d", "ref41", "pascucci", "ref41_3d", "sph2d", "sph3d"])
def test_emulated_cross_cell_against_reference_golden(emu, name):
    """The device source of the crossing operator on the reference's golden walks, on the CPU: the branch-free
    cross_cell_lean the product runs and the branch-for-branch restatement (tests/emu/cross_cell_literal.h) give the
    reference's next cell exactly and its end points / lengths to 1e-12 (the emulator contracts multiply-adds like
    hipcc), and they agree with each other."""
    from helpers import CONFIGS, load_golden
    m = M.build_model(CONFIGS[name](M))
    m.midplane_snap = 0  # reference-literal arithmetic: what the golden vectors hold
    orc = Oracle(m, 1000)
    wk = load_golden(name)["walk"]
    n = wk.shape[0]
    cols = [np.ascontiguousarray(wk[:, q]) for q in range(6)]
    cell = np.ascontiguousarray(wk[:, 6].astype(np.int32))
    outs = []
    for literal in ((0,) if name.startswith("sph") else (0, 1)):   # (spherical_grid.f90 has the one form)
        x1, y1, z1, l = (np.zeros(n) for _ in range(4))
        nxt = np.zeros(n, np.int32)
        rc = emu.emu_cross_cell(C.byref(orc.cm), literal, n, *[_p(c, C.c_double) for c in cols], _p(cell, C.c_int),
                                _p(x1, C.c_double), _p(y1, C.c_double), _p(z1, C.c_double), _p(nxt, C.c_int),
                                _p(l, C.c_double))
        assert rc == 0
        assert np.array_equal(nxt, wk[:, 10].astype(np.int32)), literal
        scale = np.abs(wk[:, 0]) + np.abs(wk[:, 1]) + np.abs(wk[:, 2]) + wk[:, 11]
        for a, col in ((x1, 7), (y1, 8), (z1, 9), (l, 11)):
            assert np.all(np.abs(a - wk[:, col]) <= 1e-12 * scale), (literal, col)
        outs.append((x1, y1, z1, l))
    if len(outs) == 2:
        for a, b in zip(*outs):
            assert np.all(np.abs(a - b) <= 1e-13 * scale)


def test_azimuthal_sector_in_default_real_is_the_reference_sector_where_it_is_certain(emu):
    """`az_sector_certain` (mc_device.hip.h) replaces the atan2 of a stopping point / an exit from the central hole by a
    default-real arctangent wherever the sector does not depend on its last digits: where it claims certainty the sector
    is the reference's (cylindrical_grid.f90:1121-1126), and it claims it for all but a sliver of the points -- also for
    points generated ON the walls (never certain there) and with coordinates of very different magnitudes."""
    rng = np.random.default_rng(11)
    for n_az in (1, 2, 8, 72, 360, 5000):
        n = 400000
        phi = rng.random(n) * 2 * np.pi
        r = 10.0 ** rng.uniform(-3, 3, n)
        x, y = r * np.cos(phi), r * np.sin(phi)
        # a tenth of the points next to a wall, within 1e-9 .. 1e-3 of a sector width
        w = rng.integers(0, n_az, n // 10) * (2 * np.pi / n_az) + (2 * np.pi / n_az) * rng.choice([-1, 1], n // 10) * 10.0 ** rng.uniform(-9, -3, n // 10)
        x[: n // 10], y[: n // 10] = r[: n // 10] * np.cos(w), r[: n // 10] * np.sin(w)
        x[-4:] = [1.0, 0.0, -1.0, 0.0]; y[-4:] = [0.0, 1.0, 0.0, -1.0]   # on the axes
        kf, kr, ok = (np.zeros(n, np.int32) for _ in range(3))
        assert emu.emu_az_sector(n, n_az, _p(x, C.c_double), _p(y, C.c_double), _p(kf, C.c_int), _p(kr, C.c_int), _p(ok, C.c_int)) == 0
        sure = ok == 1
        assert np.array_equal(kf[sure], kr[sure]), n_az
        assert kr.min() >= 1 and kr.max() <= n_az
        free = np.ones(n, bool); free[: n // 10] = False; free[-4:] = False
        assert (~sure[free]).mean() < 4 * (2e-6 * n_az + 1e-5) + 1e-4, n_az      # (the sliver: 2 delta of every sector)
    # nothing certain without a direction
    z = np.zeros(1)
    kf, kr, ok = (np.zeros(1, np.int32) for _ in range(3))
    emu.emu_az_sector(1, 72, _p(z, C.c_double), _p(z, C.c_double), _p(kf, C.c_int), _p(kr, C.c_int), _p(ok, C.c_int))
    assert ok[0] == 0


def test_emulated_kernel_2d(emu, small_model):
    check(emu, small_model, 5000, 7)


def test_emulated_kernel_3d(emu):
    check(emu, M.build_model(M.small(n_rad=12, nz=6, n_az=8, l3D=True)), 5000, 8)


def _check_spherical(run, orc, m, n, seed, prior):
    """Device result `run` against the oracle on a spherical grid.  The grid's midplane is the cone tan(theta) = 1e-10
    (cylindrical_grid.f90:500) and a packet that crosses it passes two roots ~1e-10 apart, kept or dropped by
    `t <= 1e-15` (spherical_grid.f90:262-275): a last-ulp matter (the device contracts multiply-adds elsewhere), so
    (2D) a zero-length crossing more or less per ~1e5 crossings, same packets and sums otherwise; (3D) the reference
    keeps the label +j until the next polar wall, so which hemisphere's cell collects a path is rounding noise there
    -- the hemispheres are compared summed."""
    a = run
    b = orc.run_thermal(n, seed=seed, frozen=True, E_prior=prior, n_threads=4)
    ca, cb = a["counters"], list(b["counters"].values())
    if not isinstance(ca, list):
        ca = list(ca.values())
    assert ca[0] == cb[0] == n and ca[5] + ca[6] == n
    g = m.grid
    if g["l3D"]:
        assert abs(ca[1] - cb[1]) <= 0.03 * cb[1] and abs(ca[2] - cb[2]) <= 0.05 * cb[2]
        i, j, k = g["cell_map_i"][:m.n_cells], g["cell_map_j"][:m.n_cells], g["cell_map_k"][:m.n_cells]
        key = (i - 1) + g["n_rad"] * (np.abs(j) - 1)

        def prof(E):  # radial-polar profile, hemispheres and azimuths summed
            return np.bincount(key, weights=E, minlength=g["n_rad"] * g["nz"])
        pa, pb = prof(a["E_abs"]), prof(b["E_abs"])
        big = pb > 0.01 * pb.max()
        assert np.all(np.abs(pa[big] - pb[big]) <= 0.35 * pb[big]) and abs(pa.sum() - pb.sum()) <= 0.05 * pb.sum()
    else:
        assert ca[2:] == cb[2:] and abs(ca[1] - cb[1]) <= 3 + 3e-4 * cb[1]
        assert np.array_equal(a["n_sent"], b["n_sent"]) and np.array_equal(a["sed"][4], b["sed"][4])
        assert np.allclose(a["E_abs"], b["E_abs"], rtol=1e-9, atol=1e-11 * b["E_abs"].max())


def test_emulated_kernel_spherical_grid(emu):
    """The packet loop on the spherical grid (spherical_grid.f90 operators in the device source), 2D and 3D, HBM and
    LDS deposits, against the oracle, whose operators are pinned bit for bit to the reference."""
    for kw in (dict(), dict(n_rad=12, nz=6, n_az=8, l3D=True), dict(lsepar_pola=False)):
        m = M.build_model(M.small(grid_type=2, **kw))
        orc = Oracle(m, 4000)
        prior = orc.run_thermal(2000, seed=1)["E_abs"]
        _check_spherical(emu_run(emu, orc, 4000, 17, prior=prior), orc, m, 4000, 17, prior)
        os.environ["MCGPU_EMU_LDS"] = "1"
        try:
            _check_spherical(emu_run(emu, orc, 2000, 18, prior=prior), orc, m, 2000, 18, prior)
        finally:
            del os.environ["MCGPU_EMU_LDS"]


def _check_spherical_loose(a, b, n, tol=0.01):
    """... where a dark zone's mirror meets the midplane cone's double root: a zero-length crossing more or less there changes
    which of two roots the mirrored packet leaves from, so a few packets per thousand mirrors part ways (same physics, the
    other side of a tie): counters within 1 % (3D, where the hemisphere's label is rounding noise too -- _check_spherical --:
    4 %), the absorbed energy likewise."""
    ca, cb = a["counters"], list(b["counters"].values())
    assert ca[0] == cb[0] == n and ca[5] + ca[6] == n
    for x, y in zip(ca[1:8], cb[1:8]):
        assert abs(x - y) <= 5 + tol * y, (ca, cb)
    assert abs(a["E_abs"].sum() / b["E_abs"].sum() - 1.0) < tol


def test_emulated_spherical_grid_with_dark_zone_and_dust_classes(emu):
    """Round 5: the spherical grid's packet loop (k_thermal_sph_ext) honours a dark zone (the mirror of
    optical_depth.f90:104-112 at the wall of a flagged cell) and dust classes (lvariable_dust), 2D and 3D, HBM and LDS
    deposits.  Dust classes: the oracle packet for packet (the cone's zero-length crossings apart, as in
    _check_spherical); a dark zone: the same mirrors within a few ties (see _check_spherical_loose)."""
    for kw in (dict(), dict(n_rad=12, nz=6, n_az=8, l3D=True)):
        m = M.build_model(M.small(grid_type=2, **kw))
        M.init_variable_dust(m)
        orc = Oracle(m, 4000)
        prior = orc.run_thermal(2000, seed=1)["E_abs"]
        _check_spherical(emu_run(emu, orc, 4000, 19, prior=prior), orc, m, 4000, 19, prior)
        md = M.build_model(M.small(grid_type=2, **kw))
        # (flagged cells must not touch the central hole: a packet mirrored at their wall interacts in the cell it came from)
        md.l_dark_zone = ((md.kappa_factor > np.percentile(md.kappa_factor, 85)) & (md.grid["cell_map_i"][:md.n_cells] >= 3)).astype(np.uint8)
        orc = Oracle(md, 4000)
        prior = orc.run_thermal(2000, seed=1)["E_abs"]
        b = orc.run_thermal(4000, seed=17, frozen=True, E_prior=prior, n_threads=4)
        assert b["counters"]["dark_mirrors"] > 500
        tol = 0.04 if kw.get("l3D") else 0.01
        _check_spherical_loose(emu_run(emu, orc, 4000, 17, prior=prior), b, 4000, tol)
        os.environ["MCGPU_EMU_LDS"] = "1"
        try:
            _check_spherical_loose(emu_run(emu, orc, 4000, 17, prior=prior), b, 4000, tol)
        finally:
            del os.environ["MCGPU_EMU_LDS"]


def test_emulated_kernel_hg_isotropic_unpolarised(emu):
    check(emu, M.build_model(M.small(lisotropic=True, lsepar_pola=False)), 3000, 9)
    check(emu, M.build_model(M.small(aniso_method=2, lsepar_pola=False)), 3000, 10)


def test_emulated_kernel_dark_zone(emu, small_model):
    m = copy.copy(small_model)
    dz = np.zeros(m.n_cells, np.uint8)
    dz.reshape(m.cfg.nz, m.cfg.n_rad)[0:2, 4:12] = 1
    m.l_dark_zone = dz
    a, b = check(emu, m, 5000, 12)
    assert a["counters"][7] > 0


def test_emulated_kernel_disk_emission(emu, small_model):
    m = copy.copy(small_model)
    rng = np.random.default_rng(0)
    E_cell = rng.random((m.n_lambda, m.n_cells)) * m.kappa_factor[None, :]
    pe = np.zeros((m.n_lambda, m.n_cells + 1))
    pe[:, 1:] = np.cumsum(E_cell, axis=1)
    pe /= pe[:, -1:]
    m.prob_E_cell = pe.reshape(-1)
    m.frac_E_stars = np.full(m.n_lambda, 0.4)
    # packets born in the thick midplane random-walk for >1e4 flights: FMA-level rounding
    # differences accumulate along such walks (continuous, no discrete divergence)
    check(emu, m, 3000, 13, rtol=1e-6)


def test_axisymmetric_3d_reproduces_2d_packet_for_packet():
    """What the option mcgpu_set_midplane_snap(1) is for (the library defaults to the literal arithmetic): an axisymmetric 3D
    grid must give the same physics as the 2D grid (which has no cell wall at
    the midplane, cylindrical_grid.f90:1032-1040).  With the snap the 3D run
    reproduces the 2D run packet for packet (same seeds, frozen temperature,
    equivalent priors): identical flights / scatterings / absorptions and the
    same vertical energy profile.  The reference-literal arithmetic loses ~6 %
    of the interactions: after a midplane crossing the rounding residue of
    z0 + t*w is on the wrong side about half of the time, and the packet then
    runs through mislabelled cells until the next radial wall."""
    n = 100000
    m2 = M.build_model(M.small(n_rad=12, nz=6))
    m3 = M.build_model(M.small(n_rad=12, nz=6, n_az=8, l3D=True))
    m3l = copy.copy(m3)
    m3l.midplane_snap = 0
    prior2 = Oracle(m2, n).run_thermal(n, seed=1, n_threads=1)["E_abs"]   # one thread: the live run (hence the test) is reproducible
    g3 = m3.grid
    i = g3["cell_map_i"][:m3.n_cells] - 1
    j = np.abs(g3["cell_map_j"][:m3.n_cells]) - 1
    # E_abs is in units of the reference cell's opacity (dust_prop.f90:955): rescale
    prior3 = prior2.reshape(6, 12)[j, i] / 16.0 * (m3.extra["rho0"] / m2.extra["rho0"])
    out = {}
    for name, m, pr in (("2d", m2, prior2), ("snap", m3, prior3), ("literal", m3l, prior3)):
        r = Oracle(m, n).run_thermal(n, seed=3, n_threads=8, frozen=True, E_prior=pr)
        E = r["E_abs"]
        if m.cfg.l3D:
            E2 = np.zeros((6, 12))
            np.add.at(E2, (j, i), E)
        else:
            E2 = E.reshape(6, 12)
        out[name] = (r["counters"], E2 / E2.sum())
    c2, c3, cl = out["2d"][0], out["snap"][0], out["literal"][0]
    for k in ("flights", "scatterings", "absorptions", "escaped", "killed_star"):
        assert c3[k] == c2[k], k
    assert np.allclose(out["snap"][1], out["2d"][1], rtol=1e-9, atol=1e-15)
    assert np.array_equal(Oracle(m2, n).run_thermal(1000, seed=3, frozen=True, E_prior=prior2)["sed"][4],
                          Oracle(m3, n).run_thermal(1000, seed=3, frozen=True, E_prior=prior3)["sed"][4])
    assert cl["flights"] < 0.97 * c2["flights"]          # the literal arithmetic is measurably off


def test_emulated_kernel_voronoi(emu):
    """Voronoi backend (mc_voronoi.hip.h) against the oracle's restatement of Voronoi.f90:
    identical event counts, packet-for-packet, with cut cells and the star site."""
    m = M.build_voronoi_model(M.small(), 1500, seed=3)
    assert m.grid["v_was_cut"].sum() > 0 and m.grid["v_is_star_neighbour"].sum() > 0
    check(emu, m, 4000, 21, rtol=1e-7)
    os.environ["MCGPU_EMU_LDS"] = "1"  # the same through the LDS deposit cache (64 slots here)
    try:
        check(emu, m, 4000, 21, rtol=1e-7)
    finally:
        del os.environ["MCGPU_EMU_LDS"]
    for roles in ("1,2,3,128", "0,2,3,128", "1,0,2,0"):   # and under the role schedule (mc_roles.hip.h, VORO)
        os.environ["MCGPU_EMU_ROLES"] = roles
        try:
            check(emu, m, 4000, 21, rtol=1e-7)
        finally:
            del os.environ["MCGPU_EMU_ROLES"]


def test_emulated_voronoi_pool_schedule(emu):
    """The pool schedule (mc_voronoi_pool.hip.h): packet records in memory, queues by phase and by neighbour-list class,
    one phase per pass -- driven by ONE lane, so every pass pops one record and the scheduler's choices (emission while
    records are free, the fullest queue, partial passes, the end of the launch) all occur.  Packet for packet the oracle,
    whatever the pool's size: 2 records (every queue nearly always empty), 8, 64 (more records than packets in flight)."""
    m = M.build_voronoi_model(M.small(), 1500, seed=3)
    for log_rec in ("1", "3", "6"):
        os.environ["MCGPU_EMU_POOL"] = log_rec
        try:
            check(emu, m, 4000, 21, rtol=1e-7)
        finally:
            del os.environ["MCGPU_EMU_POOL"]
    # star outside the box (move_to_grid_Voronoi at emission), unpolarised; cell-centre disk emission; ISM packets
    cfg = M.small(lsepar_pola=False)
    cfg.star_xyz = (0.0, 0.0, 400.0)
    m2 = M.build_voronoi_model(cfg, 800, seed=5)
    m3 = M.build_voronoi_model(M.small(lsepar_pola=False), 800, seed=6)
    rng = np.random.default_rng(0)
    E_cell = rng.random((m3.n_lambda, m3.n_cells)) * m3.kappa_factor[None, :]
    pe = np.zeros((m3.n_lambda, m3.n_cells + 1))
    pe[:, 1:] = np.cumsum(E_cell, axis=1)
    pe /= pe[:, -1:]
    m3.prob_E_cell = pe.reshape(-1)
    m3.frac_E_stars = np.full(m3.n_lambda, 0.4)
    os.environ["MCGPU_EMU_POOL"] = "4"
    try:
        check(emu, m2, 3000, 22, rtol=1e-7)
        check(emu, m3, 3000, 23, rtol=1e-6)
        check(emu, _with_ism(M.build_voronoi_model(M.small(lsepar_pola=False), 600, seed=4)), 2000, 33, rtol=1e-6)
    finally:
        del os.environ["MCGPU_EMU_POOL"]


def test_emulated_kernel_voronoi_star_outside_and_disk_emission(emu):
    """move_to_grid_Voronoi (star outside the box) and pos_em_cell_voronoi (cell-centre emission)."""
    cfg = M.small(lsepar_pola=False)
    cfg.star_xyz = (0.0, 0.0, 400.0)
    m = M.build_voronoi_model(cfg, 800, seed=5)
    assert m.stars[0, 5] == 1 and m.stars[0, 4] == 0
    check(emu, m, 3000, 22, rtol=1e-7)
    m = M.build_voronoi_model(M.small(lsepar_pola=False), 800, seed=6)
    rng = np.random.default_rng(0)
    E_cell = rng.random((m.n_lambda, m.n_cells)) * m.kappa_factor[None, :]
    pe = np.zeros((m.n_lambda, m.n_cells + 1))
    pe[:, 1:] = np.cumsum(E_cell, axis=1)
    pe /= pe[:, -1:]
    m.prob_E_cell = pe.reshape(-1)
    m.frac_E_stars = np.full(m.n_lambda, 0.4)
    check(emu, m, 3000, 23, rtol=1e-6)
    os.environ["MCGPU_EMU_ROLES"] = "1,2,3,128"
    try:
        check(emu, m, 3000, 23, rtol=1e-6)
    finally:
        del os.environ["MCGPU_EMU_ROLES"]


def emu_mono(emu, orc, lam, n2, seed, rt1=True, n_chunks=8, n_phot_lim=1e9):
    from oracle.binding import _MonoOpts
    m = orc.model
    o = _MonoOpts(seed, lam, lam, n_chunks, 0, float(n2), float(n_phot_lim), int(m.capt_sup), int(rt1), 1)
    xI = np.zeros(orc.xI_shape() if rt1 else (1,))
    sed = np.zeros((9, m.cfg.N_phi, m.cfg.N_thet, m.n_lambda))
    ns = np.zeros(m.n_lambda)
    per = np.zeros(n_chunks, np.uint64)
    cnt = np.zeros(N_COUNTERS, np.uint64)
    rc = emu.emu_run_mono(C.byref(orc.cm), C.byref(o), _p(xI, C.c_double), _p(sed, C.c_double), _p(ns, C.c_double),
                          _p(per, C.c_uint64), _p(cnt, C.c_uint64))
    assert rc == 0, rc
    return dict(xI_scatt=xI, sed=sed, n_sent=ns, n_sent_chunk=per, counters=[int(c) for c in cnt])


def check_mono(emu, m, lam, n2, seed, **kw):
    orc = Oracle(m, 1e5)
    a = emu_mono(emu, orc, lam, n2, seed, **kw)
    b = orc.run_mono(lam, n2, seed=seed, n_chunks=kw.get("n_chunks", 8), n_phot_lim=kw.get("n_phot_lim", 1e9),
                     rt1=kw.get("rt1", True), n_threads=4)
    assert np.array_equal(a["n_sent_chunk"], b["n_sent_chunk"])       # every stream stops at the same packet
    assert a["counters"] == list(b["counters"].values())
    assert np.array_equal(a["n_sent"], b["n_sent"]) and np.array_equal(a["sed"][4], b["sed"][4])
    for t in (0, 5, 6, 7, 8):   # weights: albedo**n_scatt (times the ulp-level renormalisation of update_Stokes)
        assert np.allclose(a["sed"][t], b["sed"][t], rtol=1e-12, atol=1e-12)
    # Q, U, V go through update_Stokes' default-real trigonometry (scattering.f90:1218): FMA-level noise
    assert np.allclose(a["sed"][1:4], b["sed"][1:4], rtol=1e-5, atol=1e-6 * max(1.0, np.abs(b["sed"][0]).max()))
    if kw.get("rt1", True):
        # with Stokes tracking every deposit inherits the default-real trigonometry of update_Stokes (scattering.f90:1206-
        # 1218: acos, cos, sin in default real; the device forms the same cos / sin of the rotation angle algebraically):
        # the tolerances of tests/test_gpu_parity.py::_mono_parity
        pola = m.cfg.lsepar_pola and m.cfg.aniso_method == 1
        xI_close(a["xI_scatt"], b["xI_scatt"], n_midplane_cells=0 if m.cfg.l3D else m.cfg.n_rad,
                 rtol=3e-5 if pola else 1e-6, atol_rel=1e-6 if pola else 1e-8)
    return a, b


def test_emulated_sed_mode_default_real_records(emu):
    """xI_scatt accumulated in default real, the packed layout of mc_xi32.hip.h (mcgpu_set_xI_precision(4)): same packets,
    same SED bins, xI_scatt to FP32 rounding -- odd and even observer counts, polarised or not, 2D and 3D.  (The
    emulation covers the layout, the fetch and the values; the pair staging of the wavefront is a GPU test.)"""
    os.environ["MCGPU_EMU_XI_F32"] = "1"
    try:
        for cfg, lam in ((M.small(RT_n_incl=3), 9), (M.small(RT_n_incl=2, RT_n_az=2, RT_az_max=60.0, lsepar_pola=False), 5),
                         (M.small(n_rad=10, nz=5, n_az=6, l3D=True, RT_n_incl=1), 4),
                         # (the split arrangement of mc_xi32.hip.h: ten observers; eight; ten without Stokes tracking;
                         # and records that keep I: no contributions)
                         (M.small(RT_n_incl=10), 9), (M.small(RT_n_incl=4, RT_n_az=2, RT_az_max=60.0), 3),
                         (M.small(RT_n_incl=5, RT_n_az=2, RT_az_max=60.0, lsepar_pola=False), 5),
                         (M.small(RT_n_incl=10, lsepar_contrib=False), 9)):
            m = sed_model(cfg, n_thermal=20000)
            orc = Oracle(m, 1e5)
            a = emu_mono(emu, orc, lam, 10, 7)
            b = orc.run_mono(lam, 10, seed=7, n_chunks=8, rt1=True, n_threads=4)
            assert np.array_equal(a["n_sent_chunk"], b["n_sent_chunk"]) and a["counters"] == list(b["counters"].values())
            assert np.array_equal(a["sed"][4], b["sed"][4])
            xI_close(a["xI_scatt"], b["xI_scatt"], rtol=1e-4, n_midplane_cells=0 if cfg.l3D else cfg.n_rad, atol_rel=1e-5)
            assert np.abs(b["xI_scatt"]).max() > 0
    finally:
        os.environ.pop("MCGPU_EMU_XI_F32", None)


def test_emulated_sed_mode_2d(emu):
    """SED mode (mc_mono.hip.h): scout + scan + commit against the oracle's sequential streams --
    same stopping packet in every stream, same SED bins, same xI_scatt."""
    m = sed_model(M.small())
    for lam in (3, 9, 14):  # star-dominated, mixed, disk-dominated emission
        a, b = check_mono(emu, m, lam, 6, 40 + lam)
        assert a["sed"][4][0, m.capt_sup - 1, lam - 1] == 8 * 6
    a, b = check_mono(emu, m, 3, 1000, 5, n_phot_lim=150.0)   # n_phot_lim stops the streams
    assert np.all(a["n_sent_chunk"] == 150)


def test_emulated_sed_mode_variants(emu):
    check_mono(emu, sed_model(M.small(lsepar_pola=False)), 4, 5, 7)                    # N_type_flux = 5
    check_mono(emu, sed_model(M.small(lsepar_pola=False, lsepar_contrib=False)), 4, 5, 8)  # = 1
    check_mono(emu, sed_model(M.small(n_rad=10, nz=5, n_az=6, l3D=True)), 4, 5, 9)     # 3D: phik = psup = 1
    check_mono(emu, sed_model(M.small(aniso_method=2, lsepar_pola=False)), 4, 5, 10)   # HG
    m = sed_model(M.small(RT_n_incl=2, RT_n_az=3, RT_az_max=90.0, RT_imin=20.0, RT_imax=70.0))
    check_mono(emu, m, 5, 5, 11)                                                        # several azimuths
    check_mono(emu, sed_model(M.small()), 4, 5, 12, rt1=False)                         # no ray-tracing deposits


def test_emulated_sed_mode_voronoi(emu):
    """SED mode on a Voronoi grid (mc_mono_voronoi.hip.h): cut cells, star site, cell-centre disk emission."""
    m = sed_model(M.small(), voronoi_sites=1200, n_thermal=50000)
    assert m.rt["n_az_rt"] == 1 and m.rt["n_theta_rt"] == 1
    for lam in (3, 14):
        check_mono(emu, m, lam, 6, 70 + lam)
    check_mono(emu, sed_model(M.small(lsepar_pola=False), voronoi_sites=600, n_thermal=30000), 9, 5, 3)


def _with_ism(m, f_star=0.4, f_disk=0.7):
    """Star + disk + interstellar field: 30 % of the packets start on the ISM sphere."""
    m = copy.copy(m)
    g = m.grid
    if g.get("grid_type", 1) == 3:
        lim = g["limits"]
        R = 1.000001 * float(np.sqrt(lim[1] ** 2 + lim[3] ** 2 + lim[5] ** 2))
    else:
        R = 1.000001 * float(np.sqrt(g["Rmax2"] + g["zmax"][-1] ** 2))   # stars.f90:657
    m.ism = dict(R_ISM=R, centre_ISM=(0.0, 0.0, 0.0))
    rng = np.random.default_rng(0)
    E_cell = rng.random((m.n_lambda, m.n_cells)) * m.kappa_factor[None, :]
    pe = np.zeros((m.n_lambda, m.n_cells + 1))
    pe[:, 1:] = np.cumsum(E_cell, axis=1)
    pe /= pe[:, -1:]
    m.prob_E_cell = pe.reshape(-1)
    m.frac_E_stars = np.full(m.n_lambda, f_star)
    m.frac_E_disk = np.full(m.n_lambda, f_disk)
    return m


def test_emulated_ism_emission(emu, small_model):
    """The third branch of emit_packet (emit_packet_ISM): thermal step on 2D, 3D and Voronoi grids, and
    the SED step, where forced scattering never clears flag_ISM so these packets are never binned."""
    a, b = check(emu, _with_ism(small_model), 3000, 31, rtol=1e-6)
    assert a["counters"][5] < 3000                                   # some ISM packets leave unabsorbed
    check(emu, _with_ism(M.build_model(M.small(n_rad=10, nz=5, n_az=6, l3D=True))), 2000, 32, rtol=1e-6)
    check(emu, _with_ism(M.build_voronoi_model(M.small(lsepar_pola=False), 600, seed=4)), 2000, 33, rtol=1e-6)
    a, b = check_mono(emu, _with_ism(small_model), 5, 4, 34)
    assert a["counters"][5] < a["counters"][0] * 0.8


def test_emulated_role_schedule(emu, small_model):
    """mc_roles.hip.h on one lane (n_srv_pref, k_short, fly_iters, emit_qmax): a wave that prefers to serve emits and
    serves and flies the long flights when it has nothing else to do; a wave that prefers to fly serves only while the
    FLY ring is empty; with emit_qmax = 0 every long flight is flown before the next packet starts; k_short = 0 sends
    every flight through the FLY ring.  Packets go through all three rings (records popped, swapped, pushed back for
    their interaction or for binning).  Same packets, same sums as the oracle."""
    for roles in ("1,2,3,128", "0,2,3,128", "1,0,2,0", "0,1,64,4"):
        os.environ["MCGPU_EMU_ROLES"] = roles
        try:
            check(emu, small_model, 4000, 7)
            os.environ["MCGPU_EMU_LDS"] = "1"
            check(emu, small_model, 4000, 7)
            del os.environ["MCGPU_EMU_LDS"]
            check(emu, M.build_model(M.small(n_rad=12, nz=6, n_az=8, l3D=True)), 2000, 8)
            check(emu, M.build_model(M.small(aniso_method=2, lsepar_pola=False)), 2000, 10)
            md = copy.copy(small_model)   # dark zone: mirror at the end of the crossing that leads into it
            dz = np.zeros(md.n_cells, np.uint8)
            kf = md.kappa_factor.copy()
            kf[::md.cfg.n_rad] = 0.0          # never the first radial cell (optical_depth.f90:1524: i >= 2)
            dz[np.argsort(kf)[-40:]] = 1
            md.l_dark_zone = dz
            a, b = check(emu, md, 4000, 12)
            assert a["counters"][7] > 0
            check(emu, _with_ism(small_model), 2000, 31, rtol=1e-6)
        finally:
            os.environ.pop("MCGPU_EMU_ROLES", None)
            os.environ.pop("MCGPU_EMU_LDS", None)


def test_emulated_binned_deposits_and_chunks_without_tails(emu):
    """mc_binned.hip.h + the chunked launch of mcgpu.hip::launch_binned on one lane ("<packets per chunk>,<log blocks>,
    <n_srv_pref>,<k_short>,<fly_iters>"): deposits go through the staging buckets, full blocks to the log (or, when the
    bucket's region is full, straight to the grid), every chunk's log is folded, the regions are re-planned from the
    last chunk's demand, and a chunk hands its unfinished packets -- records in the rings, packets in registers, work
    items reserved but not started -- to the next one.  Same packets, same sums as the oracle, whatever the chunking."""
    m3 = M.build_model(M.small(n_rad=12, nz=6, n_az=8, l3D=True))
    mh = M.build_model(M.small(n_rad=10, nz=5, n_az=6, l3D=True, aniso_method=2, lsepar_pola=False))
    for cfg in ("100000,4096,1,2,3", "300,4096,1,2,3", "97,64,0,2,3", "1000,8,1,0,2", "50,4096,0,1,64"):
        os.environ["MCGPU_EMU_BIN"] = cfg
        try:
            check(emu, m3, 3000, 8)
            check(emu, mh, 2000, 10)
        finally:
            os.environ.pop("MCGPU_EMU_BIN", None)


def test_emulated_tail_kernel(emu, small_model):
    """mc_tail.hip.h on one lane: the role kernel hands its last packets over (MCGPU_EMU_TAIL = packets left per
    workgroup at which it does; 3D: sixth field of MCGPU_EMU_BIN, on the last chunk) and k_tail finishes them -- the
    draws of an interaction taken from the batch drawn ahead, the table searches as wave-parallel probes, emission of
    work items that were reserved but never started.  With a threshold larger than the packet count EVERY packet runs
    through the tail kernel from its emission on.  Same packets, same sums as the oracle."""
    md = copy.copy(small_model)   # dark zone
    dz = np.zeros(md.n_cells, np.uint8)
    kf = md.kappa_factor.copy()
    kf[::md.cfg.n_rad] = 0.0
    dz[np.argsort(kf)[-40:]] = 1
    md.l_dark_zone = dz
    try:
        for thr, roles in (("40", "1,2,3,128"), ("100000", "1,2,3,128"), ("25", "0,1,64,4")):
            os.environ["MCGPU_EMU_TAIL"], os.environ["MCGPU_EMU_ROLES"] = thr, roles
            check(emu, small_model, 3000, 7)
            os.environ["MCGPU_EMU_LDS"] = "1"
            check(emu, small_model, 3000, 7)
            del os.environ["MCGPU_EMU_LDS"]
            check(emu, M.build_model(M.small(aniso_method=2, lsepar_pola=False)), 2000, 10)
            check(emu, M.build_model(M.small(lisotropic=True, lsepar_pola=False)), 2000, 11)
            a, b = check(emu, md, 3000, 12)
            assert a["counters"][7] > 0
            check(emu, _with_ism(small_model), 2000, 31, rtol=1e-6)
        os.environ.pop("MCGPU_EMU_TAIL"); os.environ.pop("MCGPU_EMU_ROLES")
        m3 = M.build_model(M.small(n_rad=12, nz=6, n_az=8, l3D=True))
        for cfg in ("100000,4096,1,2,3,50", "300,4096,1,2,3,100000", "97,64,0,2,3,30"):
            os.environ["MCGPU_EMU_BIN"] = cfg
            check(emu, m3, 3000, 8)
    finally:
        for k in ("MCGPU_EMU_TAIL", "MCGPU_EMU_ROLES", "MCGPU_EMU_LDS", "MCGPU_EMU_BIN"):
            os.environ.pop(k, None)


def test_emulated_tail_hand_over_records_carry_the_whole_packet(emu, small_model):
    """k_tail's hand-over to the host (mc_tail.hip.h "The last packets on the host"): every packet of the tail is written
    back as a record after 1, 3 or 40 of its events and taken up again from that record, round after round, until none is
    left (MCGPU_EMU_TAIL_HOST) -- at any point of its life the record holds everything: lazy Stokes state flushed, pending
    deposits made, the walk's interaction count, the flags.  Same packets, same sums as the oracle."""
    md = copy.copy(small_model)   # dark zone
    dz = np.zeros(md.n_cells, np.uint8)
    kf = md.kappa_factor.copy()
    kf[::md.cfg.n_rad] = 0.0
    dz[np.argsort(kf)[-40:]] = 1
    md.l_dark_zone = dz
    try:
        for ev in ("1", "3", "40"):
            os.environ["MCGPU_EMU_TAIL_HOST"] = ev
            os.environ["MCGPU_EMU_TAIL"], os.environ["MCGPU_EMU_ROLES"] = "100000", "1,2,3,128"
            check(emu, small_model, 2000, 7)
            os.environ["MCGPU_EMU_LDS"] = "1"
            os.environ["MCGPU_EMU_TAIL"] = "40"
            check(emu, small_model, 2000, 7)
            del os.environ["MCGPU_EMU_LDS"]
            check(emu, M.build_model(M.small(aniso_method=2, lsepar_pola=False)), 1500, 10)
            a, b = check(emu, md, 2000, 12)
            assert a["counters"][7] > 0
            check(emu, _with_ism(small_model), 1500, 31, rtol=1e-6)
            os.environ.pop("MCGPU_EMU_TAIL"); os.environ.pop("MCGPU_EMU_ROLES")
            m3 = M.build_model(M.small(n_rad=12, nz=6, n_az=8, l3D=True))
            os.environ["MCGPU_EMU_BIN"] = "300,4096,1,2,3,100000"
            check(emu, m3, 2000, 8)
            os.environ.pop("MCGPU_EMU_BIN")
    finally:
        for k in ("MCGPU_EMU_TAIL", "MCGPU_EMU_ROLES", "MCGPU_EMU_LDS", "MCGPU_EMU_BIN", "MCGPU_EMU_TAIL_HOST"):
            os.environ.pop(k, None)


def emu_dust_map(emu, orc, lam, xI, Tdust, n_sent, E_disk, ang_disque=0.0, l_sym_ima=True, tau_obs=100.0):
    from oracle.binding import _RtOpts
    m = orc.model
    az = np.ascontiguousarray(m.rt["tab_RT_az"], np.float32)
    o = _RtOpts(int(lam), float(m.lam[lam - 1]), float(m.E_stars[lam - 1] + E_disk), float(n_sent),
                float(m.cfg.distance), float(ang_disque), int(l_sym_ima), float(tau_obs), float(m.cfg.rin),
                float(m.cfg.rout), _p(az, C.c_float), 1)
    out = np.zeros((m.rt["RT_n_incl"] * m.rt["RT_n_az"], m.rt["N_type_flux"]))
    x = np.ascontiguousarray(xI, np.float64)
    T = np.ascontiguousarray(Tdust, np.float32)
    rc = emu.emu_rt1_dust_map(C.byref(orc.cm), C.byref(o), _p(x, C.c_double), _p(T, C.c_float), _p(out, C.c_double))
    assert rc == 0, rc
    return out


@pytest.mark.parametrize("kw", [dict(), dict(l3D=True, n_az=4), dict(lsepar_pola=False, lsepar_contrib=False),
                                dict(RT_n_az=2, RT_az_max=90.0)])
def test_emulated_rt1_dust_map(emu, kw):
    """mc_raytrace.hip.h against the oracle's dust_map restatement on the same xI_scatt and Tdust: the same
    rays through the same cells, so the fluxes agree to summation order."""
    cfg = M.small(**{**dict(n_rad=10, nz=6, RT_n_incl=2), **kw})
    m = sed_model(cfg, n_thermal=20000)
    orc = Oracle(m, 1e5)
    for lam, ang, sym in ((3, 0.0, True), (m.n_lambda - 6, 17.0, False)):  # 17 deg: no ray on a sub-bin edge
        b = orc.run_mono(lam, 40, seed=5, n_chunks=4, rt1=True, n_threads=1)
        xI = b["xI_scatt"].copy()
        if not cfg.l3D:  # a ray through a midplane cell has its midpoint at z = +-rounding (see helpers.xI_close):
            xI[:cfg.n_rad] = xI[:cfg.n_rad].mean(axis=3, keepdims=True)  # make psup irrelevant there
        args = (lam, xI, m.extra["Tdust"], b["n_sent"][lam - 1], m.extra["E_disk"][lam - 1])
        ref = orc.dust_map_sed(*args, ang_disque=ang, l_sym_ima=sym)
        got = emu_dust_map(emu, orc, *args, ang_disque=ang, l_sym_ima=sym)
        assert np.abs(ref[:, 0]).max() > 0
        assert np.allclose(got, ref, rtol=1e-10, atol=1e-14 * np.abs(ref).max())


def _blur_midplane_layer(m, xI, Tdust):
    """Spherical grids: a ray through the midplane cone is one crossing or two (the double root ~1e-10 apart, kept or dropped
    at the last ulp), so in the layer next to the cone the path's midpoint -- its sub-bin, and in 3D the hemisphere's
    label -- is rounding noise in any build: make both irrelevant there."""
    g, cfg = m.grid, m.cfg
    xI, T = xI.copy(), np.array(Tdust, np.float32).copy()
    j = np.asarray(g["cell_map_j"])[:m.n_cells]
    lay = np.abs(j) == 1
    xI[lay] = xI[lay].mean(axis=(3, 4), keepdims=True)
    if cfg.l3D:
        i, k = np.asarray(g["cell_map_i"])[:m.n_cells], np.asarray(g["cell_map_k"])[:m.n_cells]
        up, dn = np.flatnonzero(j == 1), np.flatnonzero(j == -1)
        assert np.array_equal(i[up], i[dn]) and np.array_equal(k[up], k[dn])
        xI[up] = xI[dn] = 0.5 * (xI[up] + xI[dn])
        T[up] = T[dn] = 0.5 * (T[up] + T[dn])
    return xI, T


@pytest.mark.parametrize("kw", [dict(), dict(lsepar_pola=False), dict(l3D=True, n_az=4)])
def test_emulated_rt1_dust_map_spherical(emu, kw):
    """The ray tracer with spherical_grid.f90's operators (picked at run time in rt1_integ_ray / optical_length_tot)."""
    cfg = M.small(**{**dict(grid_type=2, n_rad=10, nz=6, RT_n_incl=2), **kw})
    m = sed_model(cfg, n_thermal=20000)
    orc = Oracle(m, 1e5)
    for lam, ang, sym in ((3, 0.0, True), (m.n_lambda - 6, 17.0, False)):
        b = orc.run_mono(lam, 40, seed=5, n_chunks=4, rt1=True, n_threads=1)
        xI, T = _blur_midplane_layer(m, b["xI_scatt"], m.extra["Tdust"])
        args = (lam, xI, T, b["n_sent"][lam - 1], m.extra["E_disk"][lam - 1])
        ref = orc.dust_map_sed(*args, ang_disque=ang, l_sym_ima=sym)
        got = emu_dust_map(emu, orc, *args, ang_disque=ang, l_sym_ima=sym)
        assert np.abs(ref[:, 0]).max() > 0
        assert np.allclose(got, ref, rtol=1e-9, atol=1e-13 * np.abs(ref).max()), np.abs(got / ref - 1).max()


@pytest.mark.parametrize("kw", [dict(), dict(lsepar_pola=False)])
def test_emulated_rt1_on_a_voronoi_grid(emu, kw):
    """The ray tracer on a Voronoi grid (mc_raytrace_voronoi.hip.h: move_to_grid_Voronoi, cross_Voronoi_cell with
    previous_cell = 0 as integ_ray_dust has it, cut cells, the star's site) against the oracle: SED sampling and an image."""
    from oracle.binding import _RtOpts
    cfg = M.small(RT_n_incl=2, **kw)
    m = sed_model(cfg, voronoi_sites=1200, n_thermal=30000)
    orc = Oracle(m, 1e5)
    for lam, ang, sym in ((3, 0.0, True), (m.n_lambda - 6, 17.0, False)):
        b = orc.run_mono(lam, 40, seed=5, n_chunks=4, rt1=True, n_threads=1)
        args = (lam, b["xI_scatt"], m.extra["Tdust"], b["n_sent"][lam - 1], m.extra["E_disk"][lam - 1])
        ref = orc.dust_map_sed(*args, ang_disque=ang, l_sym_ima=sym)
        got = emu_dust_map(emu, orc, *args, ang_disque=ang, l_sym_ima=sym)
        assert np.abs(ref[:, 0]).max() > 0
        assert np.allclose(got, ref, rtol=1e-9, atol=1e-13 * np.abs(ref).max()), np.abs(got / ref - 1).max()
    lam = 9
    b = orc.run_mono(lam, 10 ** 9, seed=5, n_chunks=4, n_phot_lim=300.0, rt1=True, n_threads=1)
    xI, T = b["xI_scatt"], np.ascontiguousarray(m.extra["Tdust"], np.float32)
    az = np.ascontiguousarray(m.rt["tab_RT_az"], np.float32)
    ns, Ed = b["n_sent"][lam - 1], m.extra["E_disk"][lam - 1]
    npx, npy = 9, 9
    ref, nr = orc.dust_map_image(lam, xI, T, ns, Ed, npx, npy, 2.2 * cfg.rout, zoom=1.5, ang_disque=17.3, l_sym_ima=False)
    o = _RtOpts(int(lam), float(m.lam[lam - 1]), float(m.E_stars[lam - 1] + Ed), float(ns), float(cfg.distance),
                17.3, 0, 100.0, float(cfg.rin), float(cfg.rout), _p(az, C.c_float), 1)
    got = np.zeros_like(ref)
    n_rays = C.c_int(0)
    rc = emu.emu_rt1_image(C.byref(orc.cm), C.byref(o), C.c_int(npx), C.c_int(npy), C.c_double(2.2 * cfg.rout),
                           C.c_double(1.5), _p(np.ascontiguousarray(xI, np.float64), C.c_double), _p(T, C.c_float),
                           _p(got, C.c_double), C.byref(n_rays))
    assert rc == 0 and n_rays.value == nr and ref[0].max() > 0
    assert np.allclose(got, ref, rtol=1e-9, atol=1e-13 * np.abs(ref).max())


def test_emulated_rt1_image(emu):
    """k_rt1_image (one pixel per wavefront, sub-pixel refinement) against the oracle's dust_map method 2: the same
    refinement decisions (same number of rays) and the same pixels."""
    from oracle.binding import _RtOpts
    cfg = M.small(n_rad=10, nz=6, RT_n_incl=2, RT_n_az=2, RT_az_max=60.0)
    m = sed_model(cfg, n_thermal=20000)
    orc = Oracle(m, 1e5)
    lam = 9
    b = orc.run_mono(lam, 10 ** 9, seed=5, n_chunks=4, n_phot_lim=300.0, rt1=True, n_threads=1)  # image-mode MC: fixed count
    assert np.all(b["n_sent_chunk"] == 300)
    xI = b["xI_scatt"].copy()
    xI[:cfg.n_rad] = xI[:cfg.n_rad].mean(axis=3, keepdims=True)
    az = np.ascontiguousarray(m.rt["tab_RT_az"], np.float32)
    T = np.ascontiguousarray(m.extra["Tdust"], np.float32)
    for npx, npy, sym, ang in ((9, 9, False, 17.3), (12, 7, True, 0.0)):  # (17 deg + the 45 deg diagonal = a sub-bin edge)
        ns, Ed = b["n_sent"][lam - 1], m.extra["E_disk"][lam - 1]
        ref, nr = orc.dust_map_image(lam, xI, T, ns, Ed, npx, npy, 2.2 * cfg.rout, zoom=1.5, ang_disque=ang, l_sym_ima=sym)
        o = _RtOpts(int(lam), float(m.lam[lam - 1]), float(m.E_stars[lam - 1] + Ed), float(ns), float(cfg.distance),
                    float(ang), int(sym), 100.0, float(cfg.rin), float(cfg.rout), _p(az, C.c_float), 1)
        got = np.zeros_like(ref)
        n_rays = C.c_int(0)
        rc = emu.emu_rt1_image(C.byref(orc.cm), C.byref(o), C.c_int(npx), C.c_int(npy), C.c_double(2.2 * cfg.rout),
                               C.c_double(1.5), _p(xI, C.c_double), _p(T, C.c_float), _p(got, C.c_double), C.byref(n_rays))
        assert rc == 0
        assert n_rays.value == nr and nr >= 5 * ref[0].size * (0.5 if sym else 1.0)
        assert ref[0].max() > 0
        assert np.allclose(got, ref, rtol=1e-10, atol=1e-14 * np.abs(ref).max())
        if sym:
            assert not ref[..., npx // 2 + npx % 2:].any() and not got[..., npx // 2 + npx % 2:].any()


def test_emulated_sed_mode_spherical(emu):
    """SED mode on a spherical grid (k_mono_sph: mono_body with the operators of spherical_grid.f90), 2D and 3D, with and
    without the rt1 deposits.  Tolerances as in _check_spherical: the midplane cone's double root is a last-ulp matter
    (a zero-length crossing more or less), in 3D the hemisphere's label too (compared summed)."""
    for kw, lam in ((dict(), 5), (dict(lsepar_pola=False), 9), (dict(n_rad=10, nz=5, n_az=6, l3D=True), 4)):
        cfg = M.small(grid_type=2, **kw)
        m = sed_model(cfg, n_thermal=20000)
        orc = Oracle(m, 1e5)
        for rt1 in (False, True):
            a = emu_mono(emu, orc, lam, 6, 41, rt1=rt1)
            b = orc.run_mono(lam, 6, seed=41, n_chunks=8, rt1=rt1, n_threads=4)
            ca, cb = a["counters"], list(b["counters"].values())
            assert np.array_equal(a["n_sent_chunk"], b["n_sent_chunk"])
            assert ca[0] == cb[0] and ca[2:] == cb[2:] and abs(ca[1] - cb[1]) <= 3 + (3e-2 if cfg.l3D else 3e-4) * cb[1]
            assert np.array_equal(a["sed"][4], b["sed"][4])
            assert np.allclose(a["sed"][0], b["sed"][0], rtol=1e-12, atol=1e-12)
            if rt1:
                xa, xb = a["xI_scatt"], b["xI_scatt"]
                assert np.abs(xb).max() > 0
                if cfg.l3D:   # hemispheres (and the sub-bins that go with them) summed: cells (j, -j) of one (i, k)
                    g = m.grid
                    i, j, k = g["cell_map_i"][:m.n_cells], g["cell_map_j"][:m.n_cells], g["cell_map_k"][:m.n_cells]
                    key = (i - 1) + g["n_rad"] * ((np.abs(j) - 1) + g["nz"] * (k - 1))
                    fold = lambda x: np.stack([np.bincount(key, weights=x.reshape(m.n_cells, -1)[:, q]) for q in range(x[0].size)], 1)
                    sa, sb = fold(xa.sum(axis=(3, 4), keepdims=True)), fold(xb.sum(axis=(3, 4), keepdims=True))
                    assert np.allclose(sa, sb, rtol=1e-6, atol=1e-8 * np.abs(sb).max())
                else:
                    # the layer next to the midplane cone: a path through the cone is one crossing or two (the double
                    # root), so its midpoint -- hence its sub-bin -- is a last-ulp matter there: the cell's total
                    scale = np.abs(xb).max()
                    assert np.allclose(xa.sum(axis=(3, 4)), xb.sum(axis=(3, 4)), rtol=1e-6, atol=1e-8 * scale)
                    xI_close(xa[cfg.n_rad:], xb[cfg.n_rad:])
