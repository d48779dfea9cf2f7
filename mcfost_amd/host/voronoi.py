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


def spatial_order(sites, bits=10):
    """A permutation that lists the sites along a Morton (Z-order) curve over their per-axis RANKS (equal-count bins, so
    that a thin disk uses all bits of its z axis): cells that are neighbours in space get neighbouring indices -- and
    neighbouring addresses in the device's records.  The reference's cells keep the particles' ids the same way
    (``Voronoi(icell)%id``, Voronoi.f90:20-30: the cells are the particles it kept, not the file's order)."""
    sites = np.asarray(sites, np.float64)
    n = sites.shape[0]
    key = np.zeros(n, np.uint64)
    for ax in range(3):
        rank = np.empty(n, np.int64)
        rank[np.argsort(sites[:, ax], kind="stable")] = np.arange(n)
        q = (rank * (1 << bits) // max(n, 1)).astype(np.uint64)
        for b in range(bits):
            key |= ((q >> np.uint64(b)) & np.uint64(1)) << np.uint64(3 * b + ax)
    return np.argsort(key, kind="stable")


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


def platonic_solid(n_faces=12, radius_o_h=3.0):
    """``init_Platonic_Solid`` (Voronoi.f90:108-192): the unit normals of the solid that cuts elongated cells and the
    distance of its faces in units of h -- the solid has the volume of the sphere of radius ``radius_o_h`` h."""
    Phi = (1.0 + np.sqrt(5.0)) / 2.0
    if n_faces != 12:
        raise ValueError("platonic_solid: the dodecahedron (the reference's choice, Voronoi.f90:243) is the one built here")
    radius = np.sqrt(3.0) / 2.0 * Phi
    dist_to_face = Phi ** 3 / (2.0 * np.sqrt(Phi ** 2 + 1.0))
    volume = (15.0 + 7.0 * np.sqrt(5.0)) / 4.0
    f = 1.0 / np.sqrt(1.0 + Phi * Phi)
    fP = f * Phi
    v = np.array([[0, fP, f], [0, -fP, f], [0, fP, -f], [0, -fP, -f], [f, 0, fP], [-f, 0, fP], [f, 0, -fP], [-f, 0, -fP],
                  [fP, f, 0], [-fP, f, 0], [fP, -f, 0], [-fP, -f, 0]], f64)
    face_o_edge = dist_to_face / radius
    vol_rel = volume / (4.0 * np.pi / 3.0 * radius ** 3)
    return v, float(radius_o_h * face_o_edge / vol_rel ** (1.0 / 3.0))


def device_tessellator(device=0):
    """The product's tessellation kernel (``mcgpu_voronoi_tesselation``, include/mcgpu.h) as the callable
    ``tessellate_knn`` drives; fails loudly without a HIP device."""
    import ctypes as C
    from ..engine import load_library, McgpuError
    lib = load_library()

    def run(n, xyz, h, limits, threshold, vectors, cd_o_h, cells, knn, extra, max_neighbours, knn_first=None):
        if knn_first is None:
            n_run, k = knn.shape
        else:
            n_run, k = knn_first.size - 1, 0
        nn = np.zeros(n_run, np.int32)
        ng = np.zeros((n_run, max_neighbours), np.int32)
        vol, edge, vol0 = np.zeros(n_run, f64), np.zeros(n_run, f64), np.zeros(n_run, f64)
        cut = np.zeros(n_run, np.uint8)
        ms = C.c_double()
        dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
        lim = (C.c_double * 6)(*[float(x) for x in limits])
        rc = lib.mcgpu_voronoi_tesselation(
            C.c_int(device), C.c_int(n), xyz.ctypes.data_as(dp), h.ctypes.data_as(dp), lim, C.c_double(threshold),
            C.c_int(0 if vectors is None else vectors.shape[0]), None if vectors is None else vectors.ctypes.data_as(dp),
            C.c_double(cd_o_h), C.c_int(n_run), None if cells is None else cells.ctypes.data_as(ip), C.c_int(k),
            knn.ctypes.data_as(ip), None if knn_first is None else knn_first.ctypes.data_as(ip),
            None if extra is None else extra.ctypes.data_as(dp), C.c_int(max_neighbours),
            nn.ctypes.data_as(ip), ng.ctypes.data_as(ip), vol.ctypes.data_as(dp), edge.ctypes.data_as(dp),
            cut.ctypes.data_as(C.POINTER(C.c_ubyte)), C.byref(ms), vol0.ctypes.data_as(dp))
        if rc:
            raise McgpuError("mcgpu_voronoi_tesselation failed with code %d" % rc)
        run.kernel_ms += ms.value
        run.volume_uncut = vol0   # (of the last call: _grid_from_knn reads it after the full pass)
        return nn, ng, vol, edge, cut
    run.kernel_ms = 0.0
    return run


def delaunay_candidates(pts):
    """Per site the sites that share a Delaunay edge with it (qhull), ragged rows sorted by distance: every site whose
    bisector can bound the cell, and no other."""
    from scipy.spatial import Delaunay
    tri = Delaunay(pts)
    if tri.coplanar.size:
        raise RuntimeError("tessellation: %d sites are not in the triangulation (duplicates?)" % tri.coplanar.shape[0])
    indptr, indices = tri.vertex_neighbor_vertices
    indptr = np.asarray(indptr, np.int64)
    indices = np.asarray(indices, np.int64)
    owner = np.repeat(np.arange(pts.shape[0]), np.diff(indptr))
    d2 = ((pts[indices] - pts[owner]) ** 2).sum(axis=1)
    order = np.lexsort((d2, owner))
    return np.ascontiguousarray(indptr.astype(np.int32)), np.ascontiguousarray(indices[order].astype(np.int32))


def tessellate_knn(pts, limits, h, kernel, threshold=3.0, vectors=None, cd_o_h=3.0, extra=None, k0=40, max_neighbours=96,
                   workers=-1, candidates="delaunay"):
    """Every cell clipped out of the box on the device (``kernel``: ``device_tessellator()``, or the emulated lane of
    tests/emu).  ``candidates``: "delaunay" -- the host hands every cell its Delaunay neighbours (one qhull run: 33 s per
    million sites) and the kernel needs no security radius; "knn" -- the k nearest sites from a kd-tree, and again with
    four times as many for the cells whose security radius the list did not reach: fine for point sets without voids,
    hopeless at a disk's surface, whose cells reach far into the void and would want every site."""
    from scipy.spatial import cKDTree
    pts = np.ascontiguousarray(pts, f64)
    h = np.ascontiguousarray(h, f64)
    n = pts.shape[0]
    if candidates == "delaunay":
        first, cand = delaunay_candidates(pts)
        nn, ng, v, e, c = kernel(n, pts, h, limits, threshold, vectors, cd_o_h, None, cand,
                                 None if extra is None else np.ascontiguousarray(extra, f64), max_neighbours, knn_first=first)
        if (nn < 0).any():
            raise RuntimeError("tessellation: %d cells have more faces or vertices than the kernel holds" % int((nn < 0).sum()))
        return nn, ng, v, e, c, [(int(np.diff(first).max()), n, n)]
    tree = cKDTree(pts)
    n_neigh = np.full(n, -1, np.int32)
    neigh = np.zeros((n, max_neighbours), np.int32)
    vol, edge = np.zeros(n, f64), np.zeros(n, f64)
    cut = np.zeros(n, np.uint8)
    todo = np.arange(n, dtype=np.int32)
    k = int(k0)
    rounds = []
    while todo.size:
        kk = min(k, n - 1)
        _, idx = tree.query(pts[todo], k=kk + 1, workers=workers)
        idx = idx.reshape(todo.size, kk + 1)
        # the site itself is its own nearest neighbour (distance 0): drop it wherever it stands in the row
        self_col = idx == todo[:, None]
        first_self = np.where(self_col.any(axis=1), self_col.argmax(axis=1), kk)
        keep = np.ones(idx.shape, bool)
        keep[np.arange(todo.size), first_self] = False
        cand = np.ascontiguousarray(idx[keep].reshape(todo.size, kk).astype(np.int32))
        if kk == n - 1 and kk < k:   # every site is a candidate: pad so that the kernel sees the end of the list
            cand = np.ascontiguousarray(np.concatenate([cand, np.full((todo.size, 1), -1, np.int32)], axis=1))
        ex = None if extra is None else np.ascontiguousarray(extra[todo], f64)
        nn, ng, v, e, c = kernel(n, pts, h, limits, threshold, vectors, cd_o_h,
                                 None if todo.size == n else np.ascontiguousarray(todo), cand, ex, max_neighbours)
        if (nn == -2).any():
            raise RuntimeError("tessellation: %d cells have more faces or vertices than the kernel holds" % int((nn == -2).sum()))
        ok = nn >= 0
        sel = todo[ok]
        n_neigh[sel] = nn[ok]; neigh[sel] = ng[ok]; vol[sel] = v[ok]; edge[sel] = e[ok]; cut[sel] = c[ok]
        rounds.append((kk, int(todo.size), int(ok.sum())))
        todo = todo[~ok]
        if todo.size and kk >= n - 1:
            raise RuntimeError("tessellation: cells incomplete with every site as a candidate")
        k *= 4
    return n_neigh, neigh, vol, edge, cut, rounds


def build_voronoi_grid(sites, limits, stars_xyz_r=(), h=None, cutting_distance_o_h=3.0, cut=False, tessellator=None,
                       platonic=False):
    """CSR neighbour lists etc. for ``sites`` (n,3) + star sites appended last.

    ``h``: per-site smoothing length; default ``1.2 * volume**(1/3)``.  With ``cut``, cells
    whose farthest vertex is beyond ``cutting_distance_o_h * h`` are flagged ``was_cut`` (the
    reference cuts such elongated cells with planes, voro++_wrapper.cpp:180-228; only the
    flag, ``h`` and ``cutting_distance_o_h`` enter the packet loop, Voronoi.f90:939-975).

    ``tessellator``: None = scipy.spatial (the harness's own, minutes per million sites), or the callable of
    ``device_tessellator()`` = the product's kernel (``tessellate_knn``).  ``platonic`` (with a tessellator and ``cut``):
    the reference's cut in full -- threshold 3 h (Voronoi.f90:233), the dodecahedron of ``platonic_solid`` cutting the
    VOLUME of such cells (voro++_wrapper.cpp:209-227; the neighbour list stays the uncut cell's, as the reference stores
    it before the cut, :195-207), ``PS%cutting_distance_o_h`` as the radius of the sphere the packet loop puts in their
    place, and the cut of a star's close neighbours at the stellar surface (:229-262)."""
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
    if tessellator is not None:
        return _grid_from_knn(pts, n_before, limits, stars_xyz_r, star_icell, h, cutting_distance_o_h, cut, tessellator, platonic)
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


def _grid_from_knn(pts, n_before, limits, stars_xyz_r, star_icell, h, cutting_distance_o_h, cut, kernel, platonic):
    """``build_voronoi_grid`` on the device tessellator: the same dictionary, assembled without a Python loop over cells."""
    n = pts.shape[0]
    huge = float(np.finfo(f32).max)
    vectors, threshold, cd = None, float(cutting_distance_o_h), float(cutting_distance_o_h)
    if platonic and cut:
        vectors, cd = platonic_solid(12, 3.0)
        threshold = 3.0
    if h is None:
        # (the default needs the volumes: one pass without cuts)
        _, _, v0, _, _, _ = tessellate_knn(pts, limits, np.full(n, huge), kernel, threshold=threshold)
        h_sites = 1.2 * np.cbrt(v0[:n_before])
    else:
        h_sites = np.asarray(h, f64)
    hh = np.concatenate([h_sites, np.full(n - n_before, huge, f64)])
    h_run = hh if cut else np.full(n, huge)
    n_neigh, neigh, vol, rmax, was_cut, rounds = tessellate_knn(pts, limits, h_run, kernel, threshold=threshold, vectors=vectors,
                                                                cd_o_h=cd)
    vol_uncut = getattr(kernel, "volume_uncut", None)   # (the kernel's second output, of this one full pass)
    vol_uncut = vol.copy() if (vol_uncut is None or np.shape(vol_uncut) != vol.shape) else np.array(vol_uncut, f64)
    # rows in the harness's order: the sites by increasing id, then the walls -1, -2, ...
    big = np.int64(1) << 40
    key = np.where(neigh >= 0, neigh.astype(np.int64), big - neigh.astype(np.int64))
    key[np.arange(neigh.shape[1])[None, :] >= n_neigh[:, None]] = np.int64(1) << 50
    order = np.argsort(key, axis=1, kind="stable")
    srt = np.take_along_axis(neigh, order, axis=1)
    valid = np.arange(neigh.shape[1])[None, :] < n_neigh[:, None]
    flat = srt[valid]
    neighbours_list = np.where(flat >= 0, flat + 1, flat).astype(np.int32)      # 1-based sites, -iwall for the walls
    last = np.cumsum(n_neigh).astype(np.int32)
    first = (last - n_neigh + 1).astype(np.int32)
    owner = np.repeat(np.arange(n), n_neigh)

    is_star_neighbour = np.zeros(n, np.uint8)
    star_of = np.full(n, -1, np.int64)
    for i_star, ic in enumerate(star_icell):
        if ic > 0:
            nb = neighbours_list[first[ic - 1] - 1:last[ic - 1]]
            nb = nb[nb > 0] - 1
            is_star_neighbour[nb] = 1
            star_of[nb] = i_star
    if platonic and cut and (star_of >= 0).any():
        # a star's neighbour closer than 2 R*: cut at the stellar surface and the volume recomputed (voro++_wrapper.cpp:241-262)
        cells = np.flatnonzero(star_of >= 0)
        sx = stars_xyz_r[star_of[cells]]
        dvec = sx[:, :3] - pts[cells]
        dist = np.linalg.norm(dvec, axis=1)
        close = dist < 2.0 * sx[:, 3]
        if close.any():
            cells, dvec, dist, sx = cells[close], dvec[close], dist[close], sx[close]
            if (dist - sx[:, 3] < 0).any():
                raise RuntimeError("tessellation: a site lies inside a star")
            extra = np.zeros((n, 4), f64)
            extra[cells, :3] = dvec / dist[:, None]
            extra[cells, 3] = dist - sx[:, 3]
            sub = np.zeros(n, bool)
            sub[cells] = True
            # (only those cells: the kernel takes the list of cells to build).  Candidates: the cell's own faces of the full
            # pass -- every site that bounds the uncut cell, so none is missing for the cell cut once more and no security
            # radius is involved (a kd-tree list of 256 left the surface cells next to a star, which reach far into the
            # void, incomplete, and they silently kept the volume without the stellar cut)
            rows = [neigh[c, :n_neigh[c]] for c in cells]
            rows = [r[r >= 0] for r in rows]
            rows = [r[np.argsort(((pts[r] - pts[c]) ** 2).sum(axis=1), kind="stable")] for r, c in zip(rows, cells)]
            cfirst = np.zeros(cells.size + 1, np.int32)
            cfirst[1:] = np.cumsum([r.size for r in rows])
            cand = np.ascontiguousarray(np.concatenate(rows).astype(np.int32))
            nn2, _, v2, _, _ = kernel(n, np.ascontiguousarray(pts), np.ascontiguousarray(h_run), limits, threshold, vectors, cd,
                                      np.ascontiguousarray(cells.astype(np.int32)), cand, np.ascontiguousarray(extra[cells]),
                                      neigh.shape[1], knn_first=cfirst)
            if (nn2 < 0).any():
                raise RuntimeError("tessellation: %d star neighbours could not be cut at the stellar surface (code %d)"
                                   % (int((nn2 < 0).sum()), int(nn2[nn2 < 0][0])))
            vol[cells] = v2

    wl_sets = []
    wall_first = np.zeros(7, np.int32)
    for iw in range(1, 7):
        wall_first[iw - 1] = sum(x.size for x in wl_sets)
        wl_sets.append(np.unique(owner[neighbours_list == -iw]) + 1)
    wall_first[6] = sum(x.size for x in wl_sets)
    wall_cells = np.concatenate(wl_sets).astype(np.int32) if wl_sets else np.zeros(0, np.int32)

    wl = np.zeros((6, 4), f32)
    for iw in range(6):
        wl[iw, _WALL_AXIS[iw]] = _WALL_SIGN[iw]
        wl[iw, 3] = limits[iw]
    wc = np.zeros(n, np.uint8)
    if cut:
        wc[:n_before] = was_cut[:n_before]
    return dict(
        grid_type=3, n_cells=n, n_cells_before_stars=n_before, l3D=1,
        v_xyz=pts.astype(f32), v_xyz_dp=pts.copy(),
        v_h=hh, v_first=first, v_last=last, v_neigh=neighbours_list,
        v_was_cut=wc, v_is_star_neighbour=is_star_neighbour, v_walls=wl,
        v_cut_o_h=float(cd), v_wall_first=wall_first, v_wall_cells=wall_cells, volume=vol,
        limits=np.asarray(limits, f64), star_icell=np.array(star_icell, np.int32),
        n_rad=0, nz=0, n_az=1, tess_rounds=np.array(rounds, np.int64).reshape(-1, 3), tess_rmax=rmax,
        volume_uncut=vol_uncut,
    )
