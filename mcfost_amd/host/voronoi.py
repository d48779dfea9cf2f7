"""Harness-side Voronoi grid: what ``Voronoi_tesselation`` (Voronoi.f90:183-640)
hands to the packet loop, built with ``scipy.spatial`` instead of voro++.

The reference gets its tessellation from voro++ (``voro++_wrapper.cpp:43-277``,
github.com/cpinte/voro, un-vendored) which cannot be built here, so the
*tessellation* is this harness's own; the packet loop only sees its products:

* ``Voronoi_xyz(3,n)`` default real and ``Voronoi(:)%xyz`` dp (:23-30, :398-404)
* ``Voronoi(:)%h`` (smoothing length; the cut radius is ``h * cutting_distance_o_h``)
* ``first_neighbour/last_neighbour`` + ``neighbours_list`` (CSR, 1-based; a negative
  entry ``-iwall`` is one of the 6 box walls, :560-600)
* ``was_cut``, ``is_star_neighbour`` flags, the wall planes (:1275-1280) and the
  per-wall lists of wall-adjacent cells (:577-610)
* stars are appended as extra sites with ``h = huge_real`` (:361-376)

Cells are bounded by the box exactly as voro++ bounds them: every site whose cell
reaches a wall is mirrored across that wall, which makes the wall plane the
bisector between the site and its own mirror image.
"""
from __future__ import annotations

import numpy as np
from scipy.spatial import Voronoi, ConvexHull

f32 = np.float32
f64 = np.float64

# wall normals in the order of Voronoi.f90:1275-1280: (-x, +x, -y, +y, -z, +z)
_WALL_AXIS = (0, 0, 1, 1, 2, 2)
_WALL_SIGN = (-1.0, 1.0, -1.0, 1.0, -1.0, 1.0)


def sample_disk_sites(cfg, n_sites, seed):
    """Sites distributed like SPH particles of the cfg's disk: surface density
    ``r**surf`` between rin and rout, Gaussian in z with the flared scale height."""
    rng = np.random.default_rng(seed)
    p = cfg.surf + 2.0  # pdf(r) ~ r * Sigma(r) = r**(surf+1)
    a, b = cfg.rin ** p, cfg.rout ** p
    r = (a + (b - a) * rng.random(n_sites)) ** (1.0 / p)
    phi = 2 * np.pi * rng.random(n_sites)
    H = cfg.sclht * (r / cfg.rref) ** cfg.exp_beta
    z = H * rng.standard_normal(n_sites)
    return np.stack([r * np.cos(phi), r * np.sin(phi), z], axis=1)


def _tessellate(pts, limits):
    """Bounded Voronoi diagram of ``pts`` inside the box ``limits``.

    Returns (neighbour sets, wall sets, volumes, farthest-vertex distances).  Iterates until no cell of an
    original site has a vertex outside the box."""
    n = pts.shape[0]
    lo = np.array([limits[0], limits[2], limits[4]])
    hi = np.array([limits[1], limits[3], limits[5]])
    span = float(np.max(hi - lo))
    tol = 1e-9 * span
    mirror_sets = [np.zeros(n, bool) for _ in range(6)]
    # start: mirror the sites closest to every wall (cheap first guess)
    for iw in range(6):
        ax, sg = _WALL_AXIS[iw], _WALL_SIGN[iw]
        d = (hi[ax] - pts[:, ax]) if sg > 0 else (pts[:, ax] - lo[ax])
        k = max(16, n // 50)
        mirror_sets[iw][np.argsort(d)[:k]] = True
    for _ in range(12):
        allp, owner, wall = [pts], [np.arange(n)], [np.zeros(n, np.int32)]
        for iw in range(6):
            idx = np.nonzero(mirror_sets[iw])[0]
            q = pts[idx].copy()
            ax = _WALL_AXIS[iw]
            plane = hi[ax] if _WALL_SIGN[iw] > 0 else lo[ax]
            q[:, ax] = 2 * plane - q[:, ax]
            allp.append(q)
            owner.append(idx)
            wall.append(np.full(idx.size, iw + 1, np.int32))
        allp = np.concatenate(allp)
        owner = np.concatenate(owner)
        wall = np.concatenate(wall)
        vor = Voronoi(allp)
        grew = False
        for i in range(n):
            reg = vor.regions[vor.point_region[i]]
            if len(reg) == 0 or -1 in reg:
                for iw in range(6):
                    if not mirror_sets[iw][i]:
                        mirror_sets[iw][i] = True
                        grew = True
                continue
            vv = vor.vertices[reg]
            for iw in range(6):
                ax = _WALL_AXIS[iw]
                out = (vv[:, ax] > hi[ax] + tol).any() if _WALL_SIGN[iw] > 0 else (vv[:, ax] < lo[ax] - tol).any()
                if out and not mirror_sets[iw][i]:
                    mirror_sets[iw][i] = True
                    grew = True
        if not grew:
            break
    else:
        raise RuntimeError("bounded Voronoi tessellation did not converge")

    neigh = [set() for _ in range(n)]
    walls = [set() for _ in range(n)]
    for a, b in vor.ridge_points:
        if a >= n and b >= n:
            continue
        if a < n and b < n:
            neigh[a].add(int(b))
            neigh[b].add(int(a))
            continue
        i, m = (a, b) if a < n else (b, a)
        if owner[m] == i:  # the face lies on the wall plane
            walls[i].add(int(wall[m]))
        # a ridge with the image of another site is a zero-area contact on the wall
    vol = np.zeros(n)
    rmax = np.zeros(n)
    for i in range(n):
        vv = vor.vertices[vor.regions[vor.point_region[i]]]
        vol[i] = ConvexHull(vv).volume
        rmax[i] = np.sqrt(((vv - pts[i]) ** 2).sum(axis=1).max())
    return neigh, walls, vol, rmax


def build_voronoi_grid(sites, limits, stars_xyz_r=(), h=None, cutting_distance_o_h=3.0, cut=False):
    """CSR neighbour lists etc. for ``sites`` (n,3) + star sites appended last.

    ``h``: per-site smoothing length; default ``1.2 * volume**(1/3)``.  With ``cut``, cells
    whose farthest vertex is beyond ``cutting_distance_o_h * h`` are flagged ``was_cut`` (the
    reference cuts such elongated cells with planes, voro++_wrapper.cpp:180-228; only the
    flag, ``h`` and ``cutting_distance_o_h`` enter the packet loop, Voronoi.f90:939-975)."""
    sites = np.asarray(sites, f64)
    n_before = sites.shape[0]
    stars_xyz_r = np.asarray(stars_xyz_r, f64).reshape(-1, 4)
    star_icell = []
    pts = [sites]
    for s in stars_xyz_r:
        inside = (limits[0] < s[0] < limits[1]) and (limits[2] < s[1] < limits[3]) and (limits[4] < s[2] < limits[5])
        if inside:
            pts.append(s[None, :3])
            star_icell.append(sum(p.shape[0] for p in pts))
        else:
            star_icell.append(0)
    pts = np.concatenate(pts)
    n = pts.shape[0]
    neigh, walls, vol, rmax = _tessellate(pts, limits)

    if h is None:
        h_sites = 1.2 * np.cbrt(vol[:n_before])
    else:
        h_sites = np.asarray(h, f64)
    hh = np.concatenate([h_sites, np.full(n - n_before, np.finfo(f32).max, f64)])

    first = np.zeros(n, np.int32)
    last = np.zeros(n, np.int32)
    lst = []
    for i in range(n):
        first[i] = len(lst) + 1
        lst.extend(sorted(j + 1 for j in neigh[i]))
        lst.extend(-w for w in sorted(walls[i]))
        last[i] = len(lst)
    neighbours_list = np.array(lst, np.int32)

    is_star_neighbour = np.zeros(n, np.uint8)
    for ic in star_icell:
        if ic > 0:
            for j in neigh[ic - 1]:
                is_star_neighbour[j] = 1

    was_cut = np.zeros(n, np.uint8)
    if cut:
        was_cut[:n_before] = rmax[:n_before] > cutting_distance_o_h * h_sites

    wall_first = np.zeros(7, np.int32)
    wall_cells = []
    for iw in range(1, 7):
        wall_first[iw - 1] = len(wall_cells)
        wall_cells.extend(i + 1 for i in range(n) if iw in walls[i])
    wall_first[6] = len(wall_cells)

    wl = np.zeros((6, 4), f32)
    for iw in range(6):
        wl[iw, _WALL_AXIS[iw]] = _WALL_SIGN[iw]
        wl[iw, 3] = limits[iw]
    xyz32 = pts.astype(f32)
    return dict(
        grid_type=3, n_cells=n, n_cells_before_stars=n_before, l3D=1,
        v_xyz=xyz32, v_xyz_dp=pts.copy(),  # Voronoi_xyz is real; %xyz keeps dp (:398-403)
        v_h=hh, v_first=first, v_last=last, v_neigh=neighbours_list,
        v_was_cut=was_cut, v_is_star_neighbour=is_star_neighbour, v_walls=wl,
        v_cut_o_h=float(cutting_distance_o_h), v_wall_first=wall_first,
        v_wall_cells=np.array(wall_cells, np.int32), volume=vol,
        limits=np.asarray(limits, f64), star_icell=np.array(star_icell, np.int32),
        # placeholders so code that reads the cylindrical keys keeps working
        n_rad=0, nz=0, n_az=1,
    )
