"""The device tessellator (mcfost_amd/csrc/mc_tessellate.hip.h, C-ABI mcgpu_voronoi_tesselation): what the reference gets
from voro_C (voro++_wrapper.cpp:43-277) -- neighbours, walls, volumes, the cut of elongated cells, the cut of a star's
neighbours at the stellar surface.

voro++ is absent from this image and from /root/reference (un-vendored, lib/install.sh:135), so the tessellation is
"parity unpinned" against the reference's library; it is pinned against an independent construction of the same
mathematical object: scipy.spatial.Voronoi (qhull) of the sites mirrored across the walls (mcfost_amd/host/voronoi.py:
_tessellate) -- the same CSR bit for bit, volumes to 1e-12.  On the CPU the kernel source runs through the one-lane
emulation of tests/emu/emu_tessellate.cpp; `-m gpu` runs the kernel itself through the C-ABI.
"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from mcfost_amd.host import model as M
from mcfost_amd.host import voronoi as V

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "emu", "emu_tessellate.cpp")
LIB = os.path.join(HERE, "emu", "libemu_tessellate.so")
DEV = os.path.join(os.path.dirname(HERE), "mcfost_amd", "csrc", "mc_tessellate.hip.h")


@pytest.fixture(scope="module")
def emu_kernel():
    if (not os.path.exists(LIB)) or os.path.getmtime(LIB) < max(os.path.getmtime(SRC), os.path.getmtime(DEV)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", LIB, SRC])
    lib = C.CDLL(LIB)

    def run(n, xyz, h, limits, threshold, vectors, cd_o_h, cells, knn, extra, max_neighbours, knn_first=None):
        n_run, k = (knn.shape if knn_first is None else (knn_first.size - 1, 0))
        nn = np.zeros(n_run, np.int32)
        ng = np.zeros((n_run, max_neighbours), np.int32)
        vol, edge, vol0 = np.zeros(n_run), np.zeros(n_run), np.zeros(n_run)
        cut = np.zeros(n_run, np.uint8)
        dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
        lim = (C.c_double * 6)(*[float(x) for x in limits])
        lib.emu_voronoi_tesselation(
            C.c_int(n), xyz.ctypes.data_as(dp), h.ctypes.data_as(dp), lim, C.c_double(threshold),
            C.c_int(0 if vectors is None else vectors.shape[0]), None if vectors is None else vectors.ctypes.data_as(dp),
            C.c_double(cd_o_h), C.c_int(n_run), None if cells is None else cells.ctypes.data_as(ip), C.c_int(k),
            knn.ctypes.data_as(ip), None if knn_first is None else knn_first.ctypes.data_as(ip),
            None if extra is None else extra.ctypes.data_as(dp), C.c_int(max_neighbours), nn.ctypes.data_as(ip),
            ng.ctypes.data_as(ip), vol.ctypes.data_as(dp), edge.ctypes.data_as(dp), cut.ctypes.data_as(C.POINTER(C.c_ubyte)),
            vol0.ctypes.data_as(dp))
        run.volume_uncut = vol0
        return nn, ng, vol, edge, cut
    return run


def _disk(n_sites, seed=3):
    cfg = M.small()
    sites = V.sample_disk_sites(cfg, n_sites, seed)
    zl = 1.2 * np.abs(sites[:, 2]).max()
    L = 1.001 * cfg.rout
    return sites, (-L, L, -L, L, -zl, zl)


def _same_grid(a, b, vol_rtol=1e-12):
    for k in ("v_first", "v_last", "v_neigh", "v_was_cut", "v_is_star_neighbour", "v_wall_first", "v_wall_cells", "star_icell"):
        assert np.array_equal(a[k], b[k]), k
    assert np.allclose(a["volume"], b["volume"], rtol=vol_rtol, atol=0)
    assert np.array_equal(a["v_xyz"], b["v_xyz"]) and np.array_equal(a["v_h"], b["v_h"])


@pytest.mark.parametrize("mode", ["delaunay", "knn"])
def test_emulated_tessellator_equals_scipy(emu_kernel, mode):
    """Disk-like sites with a void above and below (cells that reach the box), the star's site inside: the CSR of
    neighbours and walls, was_cut, the star's neighbours and the per-wall lists equal scipy's bit for bit, the volumes
    to 1e-12, and they tile the box.  Both candidate sources: Delaunay neighbours (no security radius) and k nearest
    sites with the security radius (several rounds: the cells at the disk's surface want hundreds of candidates)."""
    for n_sites in (400, 3000):
        sites, limits = _disk(n_sites)
        star = [(0.0, 0.0, 0.0, 0.01)]
        a = V.build_voronoi_grid(sites, limits, stars_xyz_r=star, cut=True)
        kern = emu_kernel if mode == "delaunay" else (lambda *args, **kw: emu_kernel(*args, **kw))
        if mode == "knn":
            orig = V.tessellate_knn
            try:
                V.tessellate_knn = lambda *args, **kw: orig(*args, **dict(kw, candidates="knn"))
                b = V.build_voronoi_grid(sites, limits, stars_xyz_r=star, h=a["v_h"][:n_sites], cut=True, tessellator=kern)
            finally:
                V.tessellate_knn = orig
            assert b["tess_rounds"].shape[0] >= 2
        else:
            b = V.build_voronoi_grid(sites, limits, stars_xyz_r=star, h=a["v_h"][:n_sites], cut=True, tessellator=kern)
        _same_grid(a, b)
        box = (limits[1] - limits[0]) * (limits[3] - limits[2]) * (limits[5] - limits[4])
        assert abs(b["volume"].sum() / box - 1.0) < 1e-12
        assert np.allclose(b["tess_rmax"][:n_sites][a["v_was_cut"][:n_sites] > 0] > 3.0 * a["v_h"][:n_sites][a["v_was_cut"][:n_sites] > 0], True)


def test_emulated_tessellator_default_h_and_star_outside(emu_kernel):
    """h from the volumes (the harness's default), no cut, a star outside the box: same grid as scipy's."""
    sites, limits = _disk(800, seed=5)
    star = [(0.0, 0.0, 10.0 * limits[5], 0.01)]
    a = V.build_voronoi_grid(sites, limits, stars_xyz_r=star)
    b = V.build_voronoi_grid(sites, limits, stars_xyz_r=star, tessellator=emu_kernel)
    assert np.array_equal(a["v_neigh"], b["v_neigh"]) and np.allclose(a["v_h"], b["v_h"], rtol=1e-12)
    assert a["star_icell"][0] == 0 and b["n_cells"] == 800


def test_platonic_cut_and_stellar_surface(emu_kernel):
    """The reference's cuts (platonic=True): the threshold is 3 h, the radius of the sphere the packet loop uses is
    PS%cutting_distance_o_h (the dodecahedron has the volume of the sphere of radius 3 h: 2.7314 h), the neighbour list is
    the UNCUT cell's (voro++_wrapper.cpp:195-207 stores it before the cut), a cut cell's volume is the intersection with
    the dodecahedron -- at most its volume and at most the uncut cell's, equal to the solid's when the solid lies inside
    the cell --, and a site closer than 2 R* to the star loses the cap beyond the stellar surface."""
    vec, cd = V.platonic_solid(12, 3.0)
    assert vec.shape == (12, 3) and np.allclose(np.linalg.norm(vec, axis=1), 1.0) and abs(cd - 2.7314) < 1e-4
    solid = lambda hh: (15 + 7 * np.sqrt(5)) / 4 * (2 * cd * hh / (((1 + np.sqrt(5)) / 2) ** 3 / np.sqrt(((1 + np.sqrt(5)) / 2) ** 2 + 1))) ** 3
    assert abs(solid(1.0) / (4 * np.pi / 3 * 27.0) - 1.0) < 1e-12       # the solid's volume = the sphere's of radius 3 h
    sites, limits = _disk(3000)
    r_star = 0.05
    star = [(0.0, 0.0, 0.0, r_star)]
    b = V.build_voronoi_grid(sites, limits, stars_xyz_r=star, cut=True, tessellator=emu_kernel)
    h = b["v_h"][:3000] * 0.5                                            # (smaller h: more cells are elongated)
    u = V.build_voronoi_grid(sites, limits, stars_xyz_r=star, h=h, cut=True, tessellator=emu_kernel)
    c = V.build_voronoi_grid(sites, limits, stars_xyz_r=star, h=h, cut=True, tessellator=emu_kernel, platonic=True)
    assert np.array_equal(u["v_neigh"], c["v_neigh"]) and np.array_equal(u["v_first"], c["v_first"])
    assert abs(c["v_cut_o_h"] - cd) < 1e-12 and u["v_cut_o_h"] == 3.0
    cut = c["v_was_cut"][:3000] > 0
    assert cut.sum() > 100 and np.array_equal(cut, c["tess_rmax"][:3000] > 3.0 * h)
    vu, vc = u["volume"][:3000], c["volume"][:3000]
    near = np.linalg.norm(sites, axis=1) < 2 * r_star
    assert np.allclose(vc[~cut & ~near], vu[~cut & ~near], rtol=1e-12)
    assert np.allclose(c["volume_uncut"][:3000], vu, rtol=1e-12)        # the kernel's second output: the volume before the cuts
    assert np.all(vc[cut] <= vu[cut] * (1 + 1e-12)) and np.all(vc[cut] <= solid(h[cut]) * (1 + 1e-12))
    full = cut & (vu > 50 * solid(h))   # (cells much larger than the solid mostly contain it)
    if full.any():
        assert np.allclose(vc[full], solid(h[full]), rtol=0.05)
    # the stellar surface: put a site at 1.5 R* from the star
    s2 = np.concatenate([sites, [[1.5 * r_star, 0.0, 0.0]]])
    h2 = np.concatenate([h, [h.mean()]])
    p0 = V.build_voronoi_grid(s2, limits, stars_xyz_r=star, h=h2, cut=True, tessellator=emu_kernel)
    p1 = V.build_voronoi_grid(s2, limits, stars_xyz_r=star, h=h2, cut=True, tessellator=emu_kernel, platonic=True)
    assert p1["v_is_star_neighbour"][3000] == 1 and p1["volume"][3000] < p0["volume"][3000]


def test_tessellator_refuses_bad_input(emu_kernel):
    sites, limits = _disk(300)
    dup = np.concatenate([sites, sites[:1]])
    with pytest.raises(RuntimeError):
        V.build_voronoi_grid(dup, limits, tessellator=emu_kernel)


@pytest.mark.gpu
def test_device_tessellator_equals_scipy_and_the_emulation(emu_kernel):
    """The kernel itself through the C-ABI: the grid of scipy at 3000 sites; the emulated lane's at 60 000 (CSR bit for
    bit, volumes 1e-12), with the reference's cuts; and 1e6 sites tile their box to 1e-9 in a fraction of a second of
    kernel time."""
    import time
    kern = V.device_tessellator()
    sites, limits = _disk(3000)
    star = [(0.0, 0.0, 0.0, 0.01)]
    a = V.build_voronoi_grid(sites, limits, stars_xyz_r=star, cut=True)
    b = V.build_voronoi_grid(sites, limits, stars_xyz_r=star, h=a["v_h"][:3000], cut=True, tessellator=kern)
    _same_grid(a, b)
    sites, limits = _disk(60000, seed=9)
    e = V.build_voronoi_grid(sites, limits, stars_xyz_r=star, cut=True, tessellator=emu_kernel, platonic=True)
    d = V.build_voronoi_grid(sites, limits, stars_xyz_r=star, h=e["v_h"][:60000], cut=True, tessellator=kern, platonic=True)
    _same_grid(e, d)
    cfg = M.ref41()
    sites = V.sample_disk_sites(cfg, 1_000_000, 1)
    zl = 1.2 * np.abs(sites[:, 2]).max()
    L = 1.001 * cfg.rout
    limits = (-L, L, -L, L, -zl, zl)
    kern.kernel_ms = 0.0
    t = time.time()
    g = V.build_voronoi_grid(sites, limits, stars_xyz_r=[(0, 0, 0, 0.0093)], h=np.full(1_000_000, 1e30), tessellator=kern)
    dt = time.time() - t
    box = (2 * L) ** 2 * 2 * zl
    assert abs(g["volume"].sum() / box - 1.0) < 1e-9 and g["v_neigh"].size > 14e6
    print("1e6 sites: %.1f s on the host (qhull + assembly), %.0f ms in the kernel, %.1f faces per cell" %
          (dt, kern.kernel_ms, g["v_neigh"].size / 1e6))
    assert kern.kernel_ms < 5000.0
