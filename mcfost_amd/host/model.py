"""Host-side model tables for the continuum Monte Carlo packet loop.

In MCFOST these tables are produced by the Fortran host (parameter reader,
grid, density, Mie, stellar spectra, ``init_reemission``) *before* the packet
loop runs; the HIP engine only consumes them through the C-ABI
(``include/mcgpu.h``).  This module is the harness-side mirror that lets the
tests and ``bench.py`` stand up the BASELINE configs without the Fortran host
(which cannot be built in this image).  Everything here is setup code outside
the hot path; each builder cites the reference routine it follows
(file:line into ``/root/reference/src``).

Precision notes: the reference mixes default ``real`` and ``real(dp)``; where a
value ends up in a table the packet loop reads, the same rounding is applied
here (``np.float32`` intermediates) so that grids agree bit-for-bit with the
reference's own ``define_cylindrical_grid`` (checked in
``tests/test_ref_geometry.py`` against ``oracle/_ref``).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

f32 = np.float32
f64 = np.float64

# ---- constants.f90:8-115 -------------------------------------------------
PI = 3.141592653589793238462643383279502884197
HP = 6.626070040e-34
KB = 1.38064852e-23
C_LIGHT = 299792458.0
THERMAL_CONST = float(f32(C_LIGHT * HP / KB))  # real, parameter (constants.f90:24)
AU_TO_M = 149597870700.0
AU_TO_CM = AU_TO_M * 100.0
RSUN = 6.957e8
RSUN_TO_AU = RSUN / AU_TO_M
MUM_TO_CM = 1.0e-4
GXMSUN = 1.3271244e20
GGRAV = float(f32(6.67428e-11))
MSUN_TO_G = GXMSUN / GGRAV * 1.0e3
TINY_REAL = float(np.finfo(np.float32).tiny)
TINY_DP = float(np.finfo(np.float64).tiny)
CUTOFF = float(f32(7.0))  # parameters.f90:111
NANG_SCATT = 180  # parameters.f90:29


@dataclass
class DiskConfig:
    """The subset of a MCFOST parameter file the thermal packet loop depends on
    (single zone, single dust population; ``read_param.f90:16-555``)."""

    name: str = "ref4.1"
    # grid (ref4.1.para:16)
    n_rad: int = 100
    nz: int = 70
    n_az: int = 1
    n_rad_in: int = 20
    l3D: bool = False
    grid_type: int = 1   # 1 = cylindrical, 2 = spherical (parameter file line "grid geometry", read_param.f90)
    # wavelengths (ref4.1.para:9)
    n_lambda: int = 50
    lambda_min: float = 0.1
    lambda_max: float = 3000.0
    # disk zone (ref4.1.para:44-50)
    dust_mass: float = 1.0e-3  # Msun
    sclht: float = 10.0
    rref: float = 100.0
    rin: float = 1.0
    edge: float = 0.0
    rout: float = 300.0
    exp_beta: float = 1.125
    surf: float = -0.5
    # star (ref4.1.para:78): blackbody (-star_bb)
    T_star: float = 5000.0
    R_star: float = 2.0  # Rsun
    star_xyz: tuple = (0.0, 0.0, 0.0)
    # temperature grid (read_param.f90:237)
    n_T: int = 100
    T_min: float = 1.0
    T_max: float = 3000.0
    # scattering
    aniso_method: int = 1
    lisotropic: bool = False
    lsepar_pola: bool = True
    # SED bins (read_param.f90:180)
    N_thet: int = 10
    N_phi: int = 1
    l_sym_centrale: bool = True
    l_sym_axiale: bool = True
    # synthetic dust family: "silicate" (ref4.1-like mix) or "pascucci"
    dust: str = "silicate"
    # SED Monte Carlo + ray-tracing directions (ref4.1.para:5,19-20; read_param.f90:145-149,180-184)
    n_photons_loop: int = 128
    RT_imin: float = 0.0
    RT_imax: float = 45.0
    RT_n_incl: int = 3
    lRT_i_centered: bool = False
    RT_az_min: float = 0.0
    RT_az_max: float = 0.0
    RT_n_az: int = 1
    lsepar_contrib: bool = True
    capt_interet: int = 1
    delta_capt: int = 1
    distance: float = 140.0


def ref41() -> DiskConfig:
    """ref4.1.para (BASELINE config 2)."""
    return DiskConfig()


def ref41_3d(n_az: int = 72, nz: int = 50) -> DiskConfig:
    """ref4.1_3D.para (BASELINE config 3): 100 x 50 x 72, 720 000 cells."""
    return DiskConfig(name="ref4.1_3D", nz=nz, n_az=n_az, l3D=True)


def pascucci() -> DiskConfig:
    """Pascucci_3.0.para (BASELINE config 1; Pascucci et al. 2004 benchmark):
    isotropic scattering forced by ``init_Pascucci_benchmark``
    (benchmarks.f90:15-35)."""
    return DiskConfig(
        name="Pascucci",
        n_lambda=61,
        lambda_min=0.110662,
        lambda_max=2168.76,
        dust_mass=1.1e-6,
        sclht=99.73557010035817,   # Pascucci_3.0.para:47, every digit
        rref=500.0,
        rin=1.0,
        rout=1000.0,
        exp_beta=1.125,
        surf=0.125,
        T_star=5800.0,
        R_star=1.0,
        lisotropic=True,
        lsepar_pola=False,
        dust="pascucci",
    )


def small(n_rad: int = 20, nz: int = 10, n_az: int = 1, l3D: bool = False,
          n_lambda: int = 24, **kw) -> DiskConfig:
    """A reduced grid for fast tests (same physics as ref4.1)."""
    return DiskConfig(name="small", n_rad=n_rad, nz=nz, n_az=n_az, l3D=l3D,
                      n_rad_in=5, n_lambda=n_lambda, **kw)


# --------------------------------------------------------------------------
# Cell mapping (cylindrical_grid.f90:45-179)
# --------------------------------------------------------------------------
def cell_mapping_sizes(n_rad, nz, n_az, l3D):
    j_start = -nz if l3D else 1
    n_cells = 2 * n_rad * nz * n_az if l3D else n_rad * nz
    jstart2 = min(1, j_start) - 1
    jend2 = nz + 1
    if jstart2 < 0:
        ntot2 = (n_rad + 2) * (jend2 - jstart2) * n_az
    else:
        ntot2 = (n_rad + 2) * (jend2 - jstart2 + 1) * n_az
    return n_cells, ntot2, jstart2, jend2 - jstart2 + 1


def build_cell_mapping(n_rad, nz, n_az, l3D):
    """Returns cell_map (Fortran-ordered flat, dims (0:n_rad+1, jlo:nz+1, n_az)),
    cell_map_i/j/k and lexit_cell."""
    n_cells, ntot2, jlo, jn = cell_mapping_sizes(n_rad, nz, n_az, l3D)
    j_start = -nz if l3D else 1
    cm = np.zeros((n_az, jn, n_rad + 2), dtype=np.int32)  # C order == Fortran (i,j,k)
    cmi = np.zeros(ntot2, np.int32)
    cmj = np.zeros(ntot2, np.int32)
    cmk = np.zeros(ntot2, np.int32)
    lexit = np.zeros(ntot2, np.int32)
    icell = 0
    for k in range(1, n_az + 1):
        for j in range(j_start, nz + 1):
            if j == 0:
                continue
            for i in range(1, n_rad + 1):
                icell += 1
                cmi[icell - 1], cmj[icell - 1], cmk[icell - 1] = i, j, k
                cm[k - 1, j - jlo, i] = icell
    assert icell == n_cells
    jend2 = nz + 1
    for k in range(1, n_az + 1):
        for j in (jlo, jend2):
            for i in range(0, n_rad + 2):
                icell += 1
                if abs(j) == jend2:
                    lexit[icell - 1] = 2
                if i == n_rad + 1:
                    lexit[icell - 1] = 1
                cmi[icell - 1], cmj[icell - 1], cmk[icell - 1] = i, j, k
                cm[k - 1, j - jlo, i] = icell
    for k in range(1, n_az + 1):
        for j in range(j_start, nz + 1):
            if j == 0:
                continue
            for i in (0, n_rad + 1):
                icell += 1
                if i == n_rad + 1:
                    lexit[icell - 1] = 1
                cmi[icell - 1], cmj[icell - 1], cmk[icell - 1] = i, j, k
                cm[k - 1, j - jlo, i] = icell
    assert icell == ntot2
    return cm.reshape(-1), cmi, cmj, cmk, lexit


# --------------------------------------------------------------------------
# Grid (cylindrical_grid.f90:183-676), single region, log radial grid
# --------------------------------------------------------------------------
def _radial_grid(cfg: DiskConfig):
    """tab_r(1:n_rad+1), r_lim, r_lim_2 (cylindrical_grid.f90:256-388): the radial grid both geometries share."""
    n_rad = cfg.n_rad
    rmin_zone = cfg.rin - 5 * cfg.edge  # read_param.f90:279
    rmax_zone = cfg.rout
    Rmin, Rmax = rmin_zone, rmax_zone
    n_rad_in = max(cfg.n_rad_in, 1)

    tab_r = np.zeros(n_rad + 2, f64)  # 1-based
    R0 = Rmin
    tab_r[1] = R0
    ln_delta_r = (1.0 / float(n_rad - n_rad_in + 1)) * math.log(Rmax / R0)  # :309
    delta_r = math.exp(ln_delta_r)
    puiss = 0.0
    p = 1 + cfg.surf - cfg.exp_beta  # :319
    if p > puiss:
        puiss = p
    if puiss == 0.0:
        for i in range(2, 2 + n_rad_in):
            fr = float(f32(2.0) ** f32(i - 1) - f32(1.0)) / float(f32(2.0) ** f32(n_rad_in) - f32(1.0))
            tab_r[i] = math.exp(math.log(R0) - (math.log(R0) - math.log(R0 * delta_r)) * fr)
    else:
        den = float(f32(2.0) ** f32(n_rad_in + 1) - f32(1.0))
        for i in range(2, 2 + n_rad_in):
            num = float(f32(2.0) ** f32(i) - f32(1.0))
            a = R0 ** puiss
            tab_r[i] = (a - (a - (R0 * delta_r) ** puiss) * num / den) ** (1.0 / puiss)  # :336
    for i in range(2 + n_rad_in, n_rad + 2):
        tab_r[i] = tab_r[i - 1] * delta_r  # :349

    tab_r2 = tab_r * tab_r
    r_lim = np.zeros(n_rad + 1, f64)
    r_lim_2 = np.zeros(n_rad + 1, f64)
    r_lim[0] = Rmin
    r_lim_2[0] = Rmin ** 2
    r_lim[1:] = tab_r[2:n_rad + 2]
    r_lim_2[1:] = tab_r2[2:n_rad + 2]
    return tab_r, tab_r2, r_lim, r_lim_2, Rmin, Rmax


_libm = None


def _tanf(x):
    """Default-real tan as the Fortran runtime evaluates it (libm's tanf; numpy's vectorised float32 tan differs in the
    last place for some arguments)."""
    global _libm
    if _libm is None:
        import ctypes
        import ctypes.util
        _libm = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
        _libm.tanf.restype = ctypes.c_float
        _libm.tanf.argtypes = [ctypes.c_float]
    return float(_libm.tanf(float(x)))


def _phi_grid(cfg: DiskConfig, pi_sp):
    """tan_phi_lim, phi_grid (cylindrical_grid.f90:583-600), default-real arithmetic."""
    n_az = cfg.n_az
    tan_phi_lim = np.zeros(n_az, f64)
    phi_c = np.zeros(n_az, f64)
    if cfg.l3D:
        delta_phi = f32(f32(2.0) * f32(pi_sp) / f32(n_az))  # :587 default real
        for k in range(1, n_az + 1):
            phi_c[k - 1] = float(f32(delta_phi * f32(f32(k) - f32(0.5))))
            phi = f32(delta_phi * f32(k))
            mod = float(np.mod(f32(phi - f32(0.5) * f32(pi_sp)), f32(pi_sp)))
            if abs(mod) < 1.0e-6:
                tan_phi_lim[k - 1] = 1.0e300
            else:
                tan_phi_lim[k - 1] = _tanf(phi)  # default-real tan
    return tan_phi_lim, phi_c


def phi_wall_sin_cos(cfg: DiskConfig):
    """sin_phi_lim, cos_phi_lim of the azimuthal walls (cylindrical_grid.f90:586-599; default-real phi, libm's sinf /
    cosf) for distance_to_closest_wall_cyl's 3D branch.  Where the reference stores the sentinel pair (cos, sin) =
    (0, 1e300) for a wall at phi = pi/2 (mod pi) the true pair (0, 1) is returned (the sentinel would make that wall
    infinitely far for the random walk); ``reference=True`` keeps the sentinel (for the comparison with the module)."""
    import ctypes
    import ctypes.util
    libm = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
    for fn in (libm.sinf, libm.cosf):
        fn.restype, fn.argtypes = ctypes.c_float, [ctypes.c_float]
    n_az = cfg.n_az
    pi_sp = f32(3.1415926535)
    sp, cp = np.zeros(n_az, f64), np.ones(n_az, f64)
    if cfg.l3D:
        delta_phi = f32(f32(2.0) * pi_sp / f32(n_az))
        for k in range(1, n_az + 1):
            phi = f32(delta_phi * f32(k))
            if abs(float(np.mod(f32(phi - f32(0.5) * pi_sp), pi_sp))) < 1.0e-6:
                cp[k - 1], sp[k - 1] = 0.0, 1.0
            else:
                cp[k - 1], sp[k - 1] = float(libm.cosf(float(phi))), float(libm.sinf(float(phi)))
    return sp, cp


def define_spherical_grid(cfg: DiskConfig):
    """The spherical branch of define_cylindrical_grid (cylindrical_grid.f90:496-580): the cylindrical grid's radial
    bins taken as spherical radii, nz polar bins uniform in cos(theta) per hemisphere (theta measured from the
    midplane), the cell mapping of build_cylindrical_cell_mapping.  Pinned bit for bit to the reference compiled in
    oracle/_ref (tests/test_ref_geometry.py)."""
    n_rad, nz, n_az = cfg.n_rad, cfg.nz, cfg.n_az
    tab_r, tab_r2, r_lim, r_lim_2, Rmin, Rmax = _radial_grid(cfg)
    tab_r3 = tab_r2 * tab_r
    r_lim_3 = np.zeros(n_rad + 1, f64)
    r_lim_3[0] = Rmin ** 3
    r_lim_3[1:] = tab_r3[2:n_rad + 2]
    w_lim = np.zeros(nz + 1, f64)
    theta_lim = np.zeros(nz + 1, f64)
    tan_theta_lim = np.zeros(nz + 1, f64)
    tan_theta_lim[0] = 1.0e-10
    w_lim[nz] = 1.0
    theta_lim[nz] = float(f32(3.1415926535)) / 2.0   # `pi/2.` with the routine's local default-real pi (:191)
    tan_theta_lim[nz] = 1.0e30
    for j in range(1, nz):   # uniform in cosine (:531-539)
        w = float(j) / float(nz)
        c = math.sqrt(1.0 - w * w)
        w_lim[j] = w
        tan_theta_lim[j] = w / c
        theta_lim[j] = math.atan(tan_theta_lim[j])
    dcos = 1.0 / float(f32(nz))   # 1.0_dp/real(nz)
    pi_sp = f32(3.1415926535)
    V = np.zeros((nz, n_rad), f64)
    rg = np.zeros((nz, n_rad), f64)
    zg = np.zeros((nz, n_rad), f64)
    for i in range(1, n_rad + 1):
        rsph = math.sqrt(r_lim[i] * r_lim[i - 1])
        for j in range(1, nz + 1):
            w = 0.5 * (w_lim[j] + w_lim[j - 1])
            uv = math.sqrt(1.0 - w * w)
            rg[j - 1, i - 1] = rsph * uv
            zg[j - 1, i - 1] = rsph * w
        if (tab_r3[i + 1] - tab_r3[i]) > float(f32(1.0e-6)) * tab_r3[i]:
            Vi = float(f32(f32(4.0) / f32(3.0)) * pi_sp) * (tab_r3[i + 1] - tab_r3[i])   # 4.0/3.0*pi in default real
        else:
            Vi = float(f32(4.0) * pi_sp) * rsph ** 2 * (tab_r[i + 1] - tab_r[i])
        V[:, i - 1] = Vi * dcos
    tan_phi_lim, phi_c = _phi_grid(cfg, float(pi_sp))
    if cfg.l3D:
        V = V * 0.5 / float(f32(n_az))
    n_cells, ntot2, jlo, jn = cell_mapping_sizes(n_rad, nz, n_az, cfg.l3D)
    cm, cmi, cmj, cmk, lexit = build_cell_mapping(n_rad, nz, n_az, cfg.l3D)
    ii = cmi[:n_cells] - 1
    jj = np.abs(cmj[:n_cells]) - 1
    return dict(
        grid_type=2, n_rad=n_rad, nz=nz, n_az=n_az, l3D=int(cfg.l3D), n_cells=n_cells, ntot2=ntot2,
        jdim_lo=jlo, jdim_n=jn, r_lim=r_lim, r_lim_2=r_lim_2, r_lim_3=r_lim_3, w_lim=w_lim, theta_lim=theta_lim,
        tan_theta_lim=tan_theta_lim, tan_phi_lim=tan_phi_lim,
        # (the cylindrical arrays the structures carry along; unused by the spherical operators)
        zmax=np.ones(n_rad, f64), z_lim=np.zeros(n_rad * (nz + 2), f64), zmaxmax=0.0,
        Rmax2=Rmax * Rmax, cell_map=cm, cell_map_i=cmi, cell_map_j=cmj, cell_map_k=cmk, lexit_cell=lexit,
        volume=V[jj, ii].copy(), r_grid=rg[jj, ii].copy(), z_grid=zg[jj, ii] * np.sign(cmj[:n_cells]),
        phi_grid=phi_c[cmk[:n_cells] - 1].copy(), Rmin=Rmin, Rmax=Rmax,
    )


def define_cylindrical_grid(cfg: DiskConfig):
    if int(getattr(cfg, "grid_type", 1)) == 2:
        return define_spherical_grid(cfg)
    n_rad, nz, n_az = cfg.n_rad, cfg.nz, cfg.n_az
    tab_r, tab_r2, r_lim, r_lim_2, Rmin, Rmax = _radial_grid(cfg)
    rmin_zone, rmax_zone = Rmin, Rmax

    zmax = np.zeros(n_rad, f64)
    rc = np.zeros(n_rad, f64)
    for i in range(1, n_rad + 1):
        rcyl = 0.5 * (r_lim[i] + r_lim[i - 1])
        rc[i - 1] = rcyl
        H = 0.0
        if rmin_zone < rcyl < rmax_zone:
            H = cfg.sclht * (rcyl / cfg.rref) ** cfg.exp_beta
        zmax[i - 1] = CUTOFF * H  # :430
    assert np.all(zmax > TINY_REAL)
    cell_height = zmax / float(f32(nz))  # :459
    z_lim = np.zeros((nz + 2, n_rad), f64)  # [j-1, i-1]; flat Fortran order (i,j)
    for j in range(1, nz + 1):
        z_lim[j - 1, :] = (float(j) - 1.0) * cell_height
    z_lim[nz, :] = zmax
    z_lim[nz + 1, :] = float(f32(1.0e30))  # :493
    zmaxmax = float(zmax.max())

    pi_sp = float(f32(3.1415926535))  # local single-precision pi (:191)
    V = np.zeros((nz, n_rad), f64)
    for i in range(1, n_rad + 1):
        if (tab_r2[i + 1] - tab_r2[i]) > float(f32(1.0e-6)) * tab_r2[i]:
            dr2 = 2.0 * pi_sp * (tab_r2[i + 1] - tab_r2[i])
        else:
            dr2 = 4.0 * pi_sp * rc[i - 1] * (tab_r[i + 1] - tab_r[i])
        V[:, i - 1] = dr2 * cell_height[i - 1]
    z_c = z_lim[:nz, :] + 0.5 * cell_height[None, :]

    tan_phi_lim, phi_c = _phi_grid(cfg, pi_sp)
    if cfg.l3D:
        V = V * 0.5 / float(f32(n_az))

    n_cells, ntot2, jlo, jn = cell_mapping_sizes(n_rad, nz, n_az, cfg.l3D)
    cm, cmi, cmj, cmk, lexit = build_cell_mapping(n_rad, nz, n_az, cfg.l3D)
    ii = cmi[:n_cells] - 1
    jj = np.abs(cmj[:n_cells]) - 1
    volume = V[jj, ii].copy()
    r_grid = rc[ii].copy()
    z_grid = z_c[jj, ii] * np.sign(cmj[:n_cells])
    phi_grid = phi_c[cmk[:n_cells] - 1].copy()
    return dict(
        n_rad=n_rad, nz=nz, n_az=n_az, l3D=int(cfg.l3D), n_cells=n_cells, ntot2=ntot2,
        jdim_lo=jlo, jdim_n=jn, r_lim=r_lim, r_lim_2=r_lim_2, zmax=zmax,
        z_lim=np.ascontiguousarray(z_lim.reshape(-1)), tan_phi_lim=tan_phi_lim,
        zmaxmax=zmaxmax, Rmax2=Rmax * Rmax, cell_map=cm, cell_map_i=cmi, cell_map_j=cmj,
        cell_map_k=cmk, lexit_cell=lexit, volume=volume, r_grid=r_grid, z_grid=z_grid,
        phi_grid=phi_grid, Rmin=Rmin, Rmax=Rmax,
    )


# --------------------------------------------------------------------------
# Wavelength and temperature grids
# --------------------------------------------------------------------------
def init_lambda(n_lambda, lambda_min, lambda_max):
    """wavelengths.f90:49-62.  lambda_min/max are default real."""
    lmin, lmax = f32(lambda_min), f32(lambda_max)
    delta = math.exp((1.0 / float(n_lambda)) * float(np.log(f32(lmax / lmin))))
    lam = np.zeros(n_lambda, f64)
    inf = np.zeros(n_lambda, f64)
    sup = np.zeros(n_lambda, f64)
    inf[0] = float(lmin)
    lam[0] = float(lmin) * math.sqrt(delta)
    sup[0] = float(lmin) * delta
    for i in range(1, n_lambda):
        lam[i] = lam[i - 1] * delta
        sup[i] = sup[i - 1] * delta
        inf[i] = sup[i - 1]
    return lam, inf, sup, sup - inf


def init_tab_temp(n_T, T_min, T_max):
    """Temperature.f90:23-39 (tab_Temp is default real)."""
    delta_T = math.exp((1.0 / float(n_T)) * float(np.log(f32(f32(T_max) / f32(T_min)))))
    tab = np.zeros(n_T, f32)
    tab[0] = f32(float(f32(T_min)) * math.sqrt(delta_T))
    for t in range(1, n_T):
        tab[t] = f32(delta_T * float(tab[t - 1]))
    return tab


# --------------------------------------------------------------------------
# Density -> kappa_factor (density.f90:84-187,1906; dust_prop.f90:945-956)
# --------------------------------------------------------------------------
def dust_density(cfg: DiskConfig, grid):
    """Dust mass density per cell in g/cm^3, normalised to ``cfg.dust_mass``."""
    r, z = grid["r_grid"], grid["z_grid"]
    fact_exp = (r / cfg.rref) ** (cfg.surf - cfg.exp_beta)
    coeff_exp = 2 * (r / cfg.rref) ** (2 * cfg.exp_beta)
    rho = fact_exp * np.exp(-((z / cfg.sclht) ** 2) / coeff_exp)
    rmin_zone = cfg.rin - 5 * cfg.edge
    rho = np.where((r > cfg.rout) | (r < rmin_zone), 0.0, rho)
    if cfg.edge > 0:
        inner = r < cfg.rin
        rho = np.where(inner, rho * np.exp(-((r - cfg.rin) ** 2) / (2.0 * cfg.edge ** 2)), rho)
    mass = float(np.sum(rho * grid["volume"])) * AU_TO_CM ** 3  # g if rho in g/cm3
    rho *= cfg.dust_mass * MSUN_TO_G / mass
    return rho


# --------------------------------------------------------------------------
# Synthetic dust optical properties.
# The reference gets these from Mie theory on optical-constant files that are
# not available offline (MCFOST_UTILS); the harness uses smooth analytic
# stand-in *data* of similar magnitude.  They are inputs, not algorithm.
# --------------------------------------------------------------------------
def synthetic_dust(cfg: DiskConfig, lam):
    """Returns kappa_ext [cm^2/g dust], albedo, g, and Mueller-matrix ratios on
    the 0..180 deg grid, per wavelength."""
    n_lambda = lam.size
    th = np.arange(NANG_SCATT + 1) * (PI / NANG_SCATT)
    mu = np.cos(th)
    if cfg.dust == "pascucci":
        # single 0.12 um grain (Pascucci et al. 2004): Rayleigh-like fall-off
        x = 2 * PI * 0.12 / lam
        qabs = np.minimum(1.0, 0.9 * x) * (1 + 1.5 * np.exp(-0.5 * ((np.log(lam / 9.7)) / 0.2) ** 2))
        qsca = np.minimum(1.2, 1.1 * x ** 4)
        kabs = 1.7e4 * qabs
        ksca = 1.7e4 * qsca
        g = np.zeros(n_lambda)
    else:
        # a^-3.5, 0.03 um - 1 mm silicate-like mix
        kabs = 3.0e3 * (1 + lam / 0.3) ** -1.0 * (1 + (lam / 300.0) ** 2) ** -0.25
        kabs *= 1 + 2.0 * np.exp(-0.5 * (np.log(lam / 9.7) / 0.15) ** 2) \
                  + 1.0 * np.exp(-0.5 * (np.log(lam / 18.0) / 0.2) ** 2)
        ksca = 4.0e3 * (1 + (lam / 1.0) ** 1.6) ** -1.0 + 2.0 * (1 + (lam / 1000.0) ** 2) ** -1
        g = 0.75 / (1 + (lam / 8.0) ** 1.2) + 0.05
    kext = kabs + ksca
    albedo = ksca / kext
    s11 = np.zeros((n_lambda, NANG_SCATT + 1))
    for l in range(n_lambda):
        gg = g[l]
        s11[l] = (1 - gg * gg) / (1 + gg * gg - 2 * gg * mu) ** 1.5
    pol = (1 - mu ** 2) / (1 + mu ** 2)  # Rayleigh-like polarisability shape
    pmax = 0.6 / (1 + (0.5 / lam) ** 2) if cfg.dust != "pascucci" else np.ones(n_lambda)
    s12 = -pmax[:, None] * pol[None, :]
    s33 = np.tile(2 * mu / (1 + mu ** 2), (n_lambda, 1))
    s22 = np.ones_like(s11)
    s34 = 0.2 * pol[None, :] * np.sin(th)[None, :] * np.ones((n_lambda, 1))
    s44 = s33.copy()
    return dict(kext=kext, albedo=albedo, g=g, s11=s11, s12_o_s11=s12, s22_o_s11=s22,
                s33_o_s11=s33, s34_o_s11=s34, s44_o_s11=s44)


def scattering_cdf(s11_row, k_sca_tot):
    """prob_s11_pos for one wavelength (dust_prop.f90:1139-1154).  ``s11_row``
    is normalised like tab_s11_pos, i.e. sum_l s11(l) sin(theta_l) dtheta ~
    k_sca_tot."""
    dtheta = PI / NANG_SCATT
    prob = np.zeros(NANG_SCATT + 1, f64)
    for l in range(2, NANG_SCATT + 1):
        theta = float(f32(l)) * dtheta
        prob[l] = prob[l - 1] + s11_row[l] * math.sin(theta) * dtheta
    prob[1:] = prob[1:] + k_sca_tot - prob[NANG_SCATT]
    prob = prob / k_sca_tot
    return prob.astype(f32)


# --------------------------------------------------------------------------
# Stellar spectrum, blackbody case (stars.f90:305-328, 505-617)
# --------------------------------------------------------------------------
def star_energy(cfg: DiskConfig, lam, lam_inf, lam_sup):
    r_au = cfg.R_star * RSUN_TO_AU
    surface = 4 * PI * r_au ** 2
    n_hr = 1000
    # spanl: log-spaced samples between lambda_min and lambda_max (utils.f90)
    hr = np.exp(np.linspace(math.log(float(f32(cfg.lambda_min))), math.log(float(f32(cfg.lambda_max))), n_hr))
    wl = hr * 1.0e-6
    cst_wl = THERMAL_CONST / (cfg.T_star * wl)
    spec = np.maximum(1.0 / (((np.exp(np.minimum(cst_wl, 700.0)) - 1.0) + 1.0e-30) * wl ** 5), 1e-200)
    E = np.zeros(lam.size, f64)
    loghr, logspec = np.log(hr), np.log(spec + 1e-30)
    for l in range(lam.size):
        sel = (hr > lam_inf[l]) & (hr < lam_sup[l])
        N = int(sel.sum())
        terme = float(spec[sel].sum())
        dev = (hr[sel].mean() / lam[l]) if N > 1 else 0.0
        if terme > TINY_DP and N > 3 and abs(dev - 1.0) < 0.1:
            E[l] = terme / N * surface
        else:
            E[l] = surface * math.exp(float(np.interp(math.log(lam[l]), loghr, logspec)))
    return E, r_au


# --------------------------------------------------------------------------
# init_reemission (thermal_emission.f90:404-550)
# --------------------------------------------------------------------------
def init_reemission(lam, delta_lam, tab_Temp, kappa_abs_LTE):
    n_lambda, n_T = lam.size, tab_Temp.size
    cst_E = 2.0 * HP * C_LIGHT ** 2 * (4.0 * PI)
    B = np.zeros((n_T, n_lambda), f64)
    dB = np.zeros((n_T, n_lambda), f64)
    wl = lam * 1.0e-6
    dwl = delta_lam * 1.0e-6
    for t in range(n_T):
        cst = THERMAL_CONST / float(tab_Temp[t])
        cst_wl = cst / wl
        ok = cst_wl < 500.0
        ce = np.exp(np.where(ok, cst_wl, 1.0))
        b = np.where(ok, 1.0 / ((wl ** 5) * (ce - 1.0)) * dwl, 0.0)
        B[t] = b
        dB[t] = np.where(ok, b * cst_wl * ce / (ce - 1.0), 0.0)
    log_Qcool = np.zeros(n_T, f64)
    Qcool0 = 0.0
    for t in range(n_T):
        integ = 0.0
        for l in range(n_lambda):
            integ = integ + kappa_abs_LTE[l] * B[t, l]
        Qcool = integ * cst_E
        if t == 0:
            Qcool0 = Qcool
        q = Qcool - Qcool0
        log_Qcool[t] = math.log(q) if q > TINY_DP else -1000.0
    cdf = np.zeros((n_T, n_lambda), f64)
    for t in range(n_T):
        integ3 = np.zeros(n_lambda + 1, f64)
        for l in range(1, n_lambda + 1):
            integ3[l] = integ3[l - 1] + kappa_abs_LTE[l - 1] * dB[t, l - 1]
        if integ3[n_lambda] > TINY_DP:
            cdf[t] = integ3[1:] / integ3[n_lambda]
    return log_Qcool, cdf


# --------------------------------------------------------------------------
# The assembled model
# --------------------------------------------------------------------------
@dataclass
class Model:
    cfg: DiskConfig
    grid: dict
    lam: np.ndarray
    delta_lam: np.ndarray
    kappa: np.ndarray
    kappa_abs_LTE: np.ndarray
    albedo: np.ndarray
    kappa_factor: np.ndarray
    prob_s11_pos: np.ndarray
    s12_o_s11: np.ndarray
    s22_o_s11: np.ndarray
    s33_o_s11: np.ndarray
    s34_o_s11: np.ndarray
    s44_o_s11: np.ndarray
    tab_g_pos: np.ndarray
    tab_Temp: np.ndarray
    log_Qcool: np.ndarray
    kdB_dT_CDF: np.ndarray
    spectre_emission_cumul: np.ndarray
    frac_E_stars: np.ndarray
    frac_E_disk: np.ndarray
    CDF_E_star: np.ndarray
    E_stars: np.ndarray
    L_tot: float
    stars: np.ndarray  # rows: x, y, z, r, icell, out_model
    rho_dust: np.ndarray
    l_dark_zone: Optional[np.ndarray] = None
    p_lambda_fixed: int = 1
    midplane_snap: int = 1  # the HARNESS's choice (the parity tests compare two builds packet for packet); the library's
                            # default is 0, the reference's literal arithmetic (include/mcgpu.h: mcgpu_set_midplane_snap)
    extra: dict = field(default_factory=dict)
    # SED mode (ray-tracing method 1): tab_s11_pos(0:nang, n_lambda) and the observer directions
    tab_s11_pos: Optional[np.ndarray] = None
    rt: Optional[dict] = None
    prob_E_cell: Optional[np.ndarray] = None
    ism: Optional[dict] = None   # {"R_ISM": float, "centre_ISM": (3,)} (stars.f90:27-28); None: no ISM field
    mrw: Optional[dict] = None   # tables of the modified random walk (init_mrw); None: off
    variable_dust: Optional[dict] = None   # per-class tables of lvariable_dust (init_variable_dust); None: one class
    method1: Optional[dict] = None         # scattering method 1 (init_scattering_method1); None: method 2

    @property
    def capt_sup(self):
        """read_param.f90:184"""
        return min(self.cfg.N_thet, self.cfg.capt_interet + self.cfg.delta_capt)

    @property
    def n_cells(self):
        return self.grid["n_cells"]

    @property
    def n_lambda(self):
        return self.lam.size

    def L_packet_th(self, n_packets_total: float) -> float:
        """thermal_emission.f90:355-356."""
        return self.L_tot / float(n_packets_total)


def init_directions_ray_tracing(cfg: DiskConfig, l3D: bool):
    """Observer directions of the ray-tracing (dust_ray_tracing.f90:234-300), the layout of
    xI_scatt (:91-98, :152) and N_type_flux (init_mcfost.f90:1603-1616)."""
    deg = PI / 180.0
    ni, na = cfg.RT_n_incl, cfg.RT_n_az
    incl = np.zeros(ni, f32)
    if ni == 1:
        incl[0] = cfg.RT_imin
    else:
        cmin, cmax = math.cos(float(f32(cfg.RT_imin)) * deg), math.cos(float(f32(cfg.RT_imax)) * deg)
        for i in range(1, ni + 1):
            if cfg.lRT_i_centered:
                fr = float(f32(f32(i) - f32(0.5)) / f32(ni))
            else:
                fr = float(f32(f32(i) - f32(1)) / f32(f32(ni) - f32(1)))
            incl[i - 1] = math.acos(cmin + fr * (cmax - cmin)) / deg
    az = np.zeros(na, f32)
    if na == 1:
        az[0] = cfg.RT_az_min
    else:
        for i in range(1, na + 1):
            az[i - 1] = f32(cfg.RT_az_min) + f32(f32(i) - f32(1)) / f32(f32(na) - f32(1)) * f32(cfg.RT_az_max - cfg.RT_az_min)
    uv = np.zeros(ni, f64)
    w = np.zeros(ni, f64)
    for i in range(ni):
        if abs(float(incl[i])) > 1e-20:
            uv[i] = math.sin(float(incl[i]) * deg)
            w[i] = math.cos(float(incl[i]) * deg)
        else:
            uv[i] = math.copysign(float(f32(1e-20)), float(incl[i]))
            w[i] = 1.0
    u = np.zeros((na, ni), f64)  # Fortran (RT_n_incl, RT_n_az): ibin fastest
    v = np.zeros((na, ni), f64)
    for i in range(ni):
        for a in range(na):
            u[a, i] = uv[i] * math.sin(float(az[a]) * deg)
            v[a, i] = -uv[i] * math.cos(float(az[a]) * deg)
    if cfg.lsepar_pola and cfg.aniso_method == 1:
        ntf = 8 if cfg.lsepar_contrib else 4
    else:
        ntf = 5 if cfg.lsepar_contrib else 1
    return dict(RT_n_incl=ni, RT_n_az=na, tab_RT_incl=incl, tab_RT_az=az, tab_u_rt=u.reshape(-1),
                tab_v_rt=v.reshape(-1), tab_w_rt=w, n_az_rt=1 if l3D else 45, n_theta_rt=1 if l3D else 2,
                N_type_flux=ntf, lsepar_contrib=int(cfg.lsepar_contrib))


def init_variable_dust(m: "Model", n_classes: int = 0, slope: float = 0.15, identical: bool = False,
                       scattering: bool = True):
    """``lvariable_dust`` tables (mem.f90:213-244) for the thermal step, in the reference's layouts: every cell gets
    a class ``p_icell`` and the opacity / re-emission tables gain that axis.  The reference builds them from the
    local grain-size distribution of a settled disk (dust_prop.f90:791-1243, a host table builder); here the classes
    are the vertical layers (``n_classes`` <= nz of them, folded over |j|) and the per-class dust is a smooth
    stand-in for settling: towards the midplane the extinction flattens in wavelength (bigger grains) and the albedo
    rises -- inputs, not algorithm.  ``identical``: every class gets the model's own tables (the known-answer case:
    the run must equal the single-class run)."""
    g = m.grid
    n_cells = m.n_cells
    nl, nT = m.n_lambda, m.tab_Temp.size
    if g.get("grid_type", 1) == 3:
        # Voronoi grid (what a multi-grain SPH dump gives; the reference then has one class per cell, p_n_cells =
        # n_cells): the classes are bins of |z| / H(r), 0.5 scale heights each -- the same stand-in for settling
        nc = int(n_classes) if n_classes else 8
        xyz = np.asarray(g["v_xyz_dp"], f64).reshape(-1, 3)[:n_cells]
        rc = np.maximum(np.hypot(xyz[:, 0], xyz[:, 1]), 1e-30)
        H = m.cfg.sclht * (rc / m.cfg.rref) ** m.cfg.exp_beta
        p_icell = (np.minimum((np.abs(xyz[:, 2]) / (0.5 * H)).astype(np.int64), nc - 1) + 1).astype(np.int32)
    else:
        n_rad, nz = g["n_rad"], g["nz"]
        nc = int(n_classes) if n_classes else nz
        j = np.abs(np.asarray(g["cell_map_j"], np.int64)[:n_cells])              # 1..nz
        p_icell = (np.minimum((j - 1) * nc // nz, nc - 1) + 1).astype(np.int32)
    lam = np.asarray(m.lam, f64)
    kappa = np.zeros((nl, nc), f64)          # Fortran (p_n_cells, n_lambda): class fastest
    kabs = np.zeros((nl, nc), f64)
    alb = np.zeros((nl, nc), f32)
    lq = np.zeros((nc, nT), f64)             # Fortran (n_T, p_n_cells)
    cdf = np.zeros((nc, nT, nl), f64)        # Fortran (n_lambda, n_T, p_n_cells)
    for c in range(nc):
        if identical:
            k_c, a_c = np.asarray(m.kappa, f64), np.asarray(m.albedo, f32)
        else:
            depth = 1.0 - (c + 0.5) / nc       # 1 at the midplane class, 0 at the surface class
            k_c = np.asarray(m.kappa, f64) * (lam / 10.0) ** (slope * depth)
            a_c = np.clip(np.asarray(m.albedo, f64) * (1.0 + 0.3 * depth), 0.0, 0.95).astype(f32)
        ka_c = np.asarray(m.kappa_abs_LTE, f64) if identical else k_c * (1.0 - a_c.astype(f64))
        kappa[:, c], kabs[:, c], alb[:, c] = k_c, ka_c, a_c
        if identical:
            lq[c], cdf[c] = m.log_Qcool, np.asarray(m.kdB_dT_CDF, f64).reshape(nT, nl)
        else:
            lq[c], cdf[c] = init_reemission(lam, np.asarray(m.delta_lam, f64), m.tab_Temp, ka_c)
    m.variable_dust = dict(p_n_cells=nc, p_icell=p_icell, kappa=kappa.reshape(-1), kappa_abs_LTE=kabs.reshape(-1),
                           albedo=alb.reshape(-1), log_Qcool=lq.reshape(-1), kdB_dT_CDF=cdf.reshape(-1))
    if scattering:
        # scattering tables per class, Fortran (0:nang, p_n_cells, n_lambda): bigger grains towards the midplane
        # scatter more forward (the HG shape with the class's g) and polarise less
        na1 = NANG_SCATT + 1
        th = np.arange(na1) * (PI / NANG_SCATT)
        mu = np.cos(th)
        dtheta = PI / NANG_SCATT
        base = {k: np.asarray(getattr(m, k), f32).reshape(nl, na1) for k in
                ("prob_s11_pos", "s12_o_s11", "s22_o_s11", "s33_o_s11", "s34_o_s11", "s44_o_s11")}
        out = {k: np.zeros((nl, nc, na1), f32) for k in base}
        gtab = np.zeros((nl, nc), f32)
        s11tab = np.zeros((nl, nc, na1), f32)     # tab_s11_pos per class: the phase function of the rt1 deposits
        base_s11 = np.asarray(m.tab_s11_pos, f32).reshape(nl, na1) if m.tab_s11_pos is not None else None
        for c in range(nc):
            depth = 0.0 if identical else 1.0 - (c + 0.5) / nc
            g_c = np.clip(np.asarray(m.tab_g_pos, f64) + 0.15 * depth, -0.9, 0.9)
            gtab[:, c] = np.asarray(m.tab_g_pos, f32) if identical else g_c.astype(f32)
            for l in range(nl):
                if identical:
                    out["prob_s11_pos"][l, c] = base["prob_s11_pos"][l]
                    if base_s11 is not None:
                        s11tab[l, c] = base_s11[l]
                else:
                    s11 = (1 - g_c[l] ** 2) / (1 + g_c[l] ** 2 - 2 * g_c[l] * mu) ** 1.5
                    k_sca = kappa[l, c] * float(alb[l, c])
                    norm = float(np.sum(s11[1:NANG_SCATT] * np.sin(th[1:NANG_SCATT]) * dtheta))
                    out["prob_s11_pos"][l, c] = scattering_cdf(s11 * (k_sca / norm) * 0.97, k_sca)
                    s11tab[l, c] = (s11 * (k_sca / norm) * 0.97).astype(f32)
            for k in ("s12_o_s11", "s22_o_s11", "s33_o_s11", "s34_o_s11", "s44_o_s11"):
                out[k][:, c, :] = base[k] * (f32(1.0 - 0.3 * depth) if k in ("s12_o_s11", "s34_o_s11") else f32(1.0))
        m.variable_dust.update({k: v.reshape(-1) for k, v in out.items()})
        m.variable_dust["tab_g_pos"] = gtab.reshape(-1)
        if base_s11 is not None:
            m.variable_dust["tab_s11_pos"] = s11tab.reshape(-1)
    return m.variable_dust


def synthetic_grains(m: "Model", n_grains: int = 16, a_min: float = 0.03, a_max: float = 1000.0, aexp: float = 3.5):
    """Per-grain tables for ``opacity`` / ``calc_local_scattering_matrices`` (dust_prop.f90:791-1243) in the
    reference's layouts and types (grains.f90:38-54): a power-law size distribution of ``n_grains`` log-spaced sizes
    with smooth stand-ins for the Mie results (size parameter x = 2 pi a / lambda: Q_abs ~ min(1, x), Q_sca ~ min(1.2, x^4),
    g rising with x, Henyey-Greenstein S11 normalised to Q_sca with 3 % of it left to the unresolved forward peak,
    Rayleigh-like polarisation that fades with x) -- inputs, not algorithm.  Arrays are C-ordered with the Fortran
    first index last: ``C_ext[n_lambda, n_grains]``, ``tab_s11[n_lambda, n_grains, nang+1]``."""
    lam = np.asarray(m.lam, f64)
    nl, na1 = lam.size, NANG_SCATT + 1
    a = np.exp(np.linspace(math.log(a_min), math.log(a_max), n_grains))
    nk = a ** (1.0 - aexp)                      # n(a) da with da ~ a on a log grid
    nk = nk / nk.sum()
    S = (PI * a * a).astype(f32)                # micron^2
    x = 2 * PI * a[None, :] / lam[:, None]      # [nl, ng]
    feat = 1 + 1.5 * np.exp(-0.5 * (np.log(lam / 9.7) / 0.2) ** 2)[:, None] / (1 + (a[None, :] / 2.0) ** 2)
    qabs = np.minimum(1.0, 0.9 * x) * feat
    qsca = np.minimum(1.2, 1.1 * x ** 4)
    g = 0.85 * x ** 2 / (1.0 + x ** 2)
    th = np.arange(na1) * (PI / NANG_SCATT)
    mu, dtheta = np.cos(th), PI / NANG_SCATT
    s11 = (1 - g[..., None] ** 2) / (1 + g[..., None] ** 2 - 2 * g[..., None] * mu) ** 1.5      # [nl, ng, na1]
    norm = np.sum(s11[..., 1:NANG_SCATT] * np.sin(th[1:NANG_SCATT]) * dtheta, axis=-1)
    s11 = s11 * (0.97 * qsca / norm)[..., None]                                              # "normalised to Q_sca"
    pol = (1 - mu ** 2) / (1 + mu ** 2)
    pmax = 1.0 / (1.0 + x ** 2)
    s12 = -pmax[..., None] * pol * s11
    s22 = s11.copy()
    s33 = (2 * mu / (1 + mu ** 2)) * s11
    s34 = 0.2 * pol * np.sin(th) * s11 * (x / (1 + x))[..., None]
    s44 = s33.copy()
    c32 = lambda v: np.ascontiguousarray(v, f32)
    return dict(n_grains=n_grains, a=a, grain_RE_LTE_start=1, grain_RE_LTE_end=n_grains,
                C_ext=c32((qabs + qsca) * S[None, :]), C_sca=c32(qsca * S[None, :]), C_abs=c32(qabs * S[None, :]),
                tab_g=c32(g), tab_s11=c32(s11), tab_s12=c32(s12), tab_s22=c32(s22), tab_s33=c32(s33), tab_s34=c32(s34),
                tab_s44=c32(s44), S_grain=S, n_grains_k=np.ascontiguousarray(nk, f64))


def settled_grain_density(m: "Model", grains: dict, xi: float = 0.25, per_cell: bool = True, n_classes: int = 0):
    """``dust_density_o_n_grains(n_grains, p_n_cells)`` (density.f90:32, here ``[p_n_cells, n_grains]`` C-ordered) of a
    settled disk: grain k has the scale height H (a_k / a_min)^-xi, its column density that of the model.
    ``per_cell``: every cell its own class (the reference's lvariable_dust: p_icell = identity, kappa_factor = 1, the
    density carries the cell's dust); otherwise ``n_classes`` vertical layers with relative abundances (kappa_factor stays
    the model's).  Scaled so that the extinction at the reference wavelength matches the model's where the dust is
    unsettled.  Returns ``(p_icell, dens)``."""
    g, cfg = m.grid, m.cfg
    n_cells, nz = m.n_cells, g["nz"]
    a = np.asarray(grains["a"], f64)
    r, z = np.asarray(g["r_grid"], f64), np.abs(np.asarray(g["z_grid"], f64))
    H = cfg.sclht * (r / cfg.rref) ** cfg.exp_beta
    shrink = (a / a[0]) ** (-xi)                                       # H_k / H
    if per_cell:
        p_icell = np.arange(1, n_cells + 1, dtype=np.int32)
        zz = (z / H)[:, None]
        w = np.exp(-0.5 * zz ** 2 * (1.0 / shrink[None, :] ** 2 - 1.0)) / shrink[None, :]
        dens = np.asarray(m.kappa_factor, f64)[:, None] * w
    else:
        nc = int(n_classes) if n_classes else nz
        j = np.abs(np.asarray(g["cell_map_j"], np.int64)[:n_cells])
        p_icell = (np.minimum((j - 1) * nc // nz, nc - 1) + 1).astype(np.int32)
        zz = ((np.arange(nc) + 0.5) / nc * 7.0)[:, None]   # layer centres in scale heights (the grid is cut at 7 H)
        dens = np.exp(-0.5 * zz ** 2 * (1.0 / shrink[None, :] ** 2 - 1.0)) / shrink[None, :]
    # calibration: sum_k C_ext(k, l0) n_k dens0 * fact = the model's kappa(l0) for unsettled dust of unit kappa_factor
    l0 = int(np.argmin(np.abs(np.asarray(m.lam) - 1.0)))
    fact = AU_TO_CM * 1.0e-8
    k0 = float(np.sum(np.asarray(grains["C_ext"], f64)[l0] * np.asarray(grains["n_grains_k"], f64))) * fact
    dens = dens * (float(np.asarray(m.kappa, f64)[l0]) / k0)
    return p_icell, np.ascontiguousarray(dens, f64)


def variable_dust_from_opacity(m: "Model", p_icell, tabs: dict, lq=None, cdf=None):
    """The ``variable_dust`` description of a model (see ``init_variable_dust``) from the tables ``opacity`` built
    (``Oracle.opacity`` / ``Engine.opacity``, the reference's layouts); ``lq`` / ``cdf``: the classes' re-emission
    tables (None: left to ``mcgpu_init_reemission`` on the device)."""
    nc = int(np.max(p_icell))
    vd = dict(p_n_cells=nc, p_icell=np.asarray(p_icell, np.int32), kappa=np.asarray(tabs["kappa"], f64).reshape(-1),
              kappa_abs_LTE=np.asarray(tabs["kappa_abs_LTE"], f64).reshape(-1), albedo=np.asarray(tabs["tab_albedo_pos"], f32).reshape(-1),
              log_Qcool=None if lq is None else np.asarray(lq, f64).reshape(-1),
              kdB_dT_CDF=None if cdf is None else np.asarray(cdf, f64).reshape(-1))
    vd["prob_s11_pos"] = np.asarray(tabs["prob_s11_pos"], f32).reshape(-1)
    na1 = NANG_SCATT + 1
    zeros = np.zeros((m.n_lambda, nc, na1), f32)
    for k_out, k_in in (("s12_o_s11", "tab_s12_o_s11_pos"), ("s22_o_s11", "tab_s22_o_s11_pos"), ("s33_o_s11", "tab_s33_o_s11_pos"),
                        ("s34_o_s11", "tab_s34_o_s11_pos"), ("s44_o_s11", "tab_s44_o_s11_pos")):
        vd[k_out] = np.asarray(tabs[k_in] if tabs.get(k_in) is not None else zeros, f32).reshape(-1)
    vd["tab_g_pos"] = np.asarray(tabs["tab_g_pos"], f32).reshape(-1)
    vd["tab_s11_pos"] = np.asarray(tabs["tab_s11_pos"], f32).reshape(-1)
    m.variable_dust = vd
    return vd


def init_scattering_method1(m: "Model", grains: dict, dens):
    """Scattering method 1 (dust_transfer.f90:1288-1316): the per-grain tables in the normalisation
    ``normalise_Mueller_matrix`` gives them for this method (scattering.f90:515-555) -- ``prob_s11(n_lambda, n_grains,
    0:nang)``, the cumulative of s11 sin(theta) dtheta with the unresolved forward peak (here: what the quadrature misses of
    Q_sca) added from bin 1 on, normalised to 1; ``tab_s1x = s1x / s11`` with ``tab_s11 = 1`` -- from ``synthetic_grains``'s
    tables (which are normalised to Q_sca, method 2's convention).  Host-side Mie set-up: inputs, not algorithm."""
    na1 = NANG_SCATT + 1
    th = np.arange(na1) * (PI / NANG_SCATT)
    dtheta = PI / NANG_SCATT
    s11 = np.asarray(grains["tab_s11"], f64)                           # [nl, ng, na1]
    inc = s11 * np.sin(th) * dtheta
    prob = np.zeros_like(s11)
    prob[..., 2:] = np.cumsum(inc[..., 2:], axis=-1)
    qsca = (np.asarray(grains["C_sca"], f64) / np.asarray(grains["S_grain"], f64)[None, :])
    norm = np.maximum(qsca, prob[..., NANG_SCATT])                      # "normalization" >= the numerical integral
    prob[..., 1:] += (norm - prob[..., NANG_SCATT])[..., None]
    prob /= prob[..., NANG_SCATT][..., None]
    with np.errstate(divide="ignore", invalid="ignore"):
        ratio = lambda k: np.where(s11 > 0, np.asarray(grains[k], f64) / s11, 0.0)
        tabs = {k: np.ascontiguousarray(ratio(k), f32) for k in ("tab_s12", "tab_s22", "tab_s33", "tab_s34", "tab_s44")}
    m.method1 = dict(n_grains=int(grains["n_grains"]), n_grains_k=np.asarray(grains["n_grains_k"], f64),
                     dens=np.ascontiguousarray(dens, f64), C_sca=np.asarray(grains["C_sca"], f32),
                     tab_g=np.asarray(grains["tab_g"], f32),
                     prob_s11=np.ascontiguousarray(np.transpose(prob, (2, 1, 0)), f32),   # Fortran (n_lambda, n_grains, 0:nang)
                     tab_s11=np.ones_like(s11, dtype=f32), **tabs)
    return m.method1


def cumulative_zeta(n: int = 10000):
    """``initialize_cumulative_zeta`` (MRW.f90:16-53): zeta(y) = 2 sum_{j>=1} (-1)^(j+1) y^(j^2) (Min et al. 2009,
    eq. 7) on y_i = (i-1)/(n-1); the last point is the limit 1."""
    y = np.arange(n, dtype=f64) / f64(n - 1)
    zeta = np.zeros(n, f64)
    j = 0
    while True:
        j += 1
        term = y[:-1] ** float(j * j)
        if not term.any():
            break
        zeta[:-1] += term if j % 2 else -term
    zeta[-1] = 0.5
    return zeta * 2.0


def init_mrw(m: "Model", gamma: float = 2.0, n_inter: int = 5, n_zeta: int = 10000, weights: str = "dB_dT",
             ext_factor: float = 0.4, exit_spectrum: str = "in_flight"):
    """Tables of the modified random walk (``mcgpu_set_mrw``), one value per temperature of ``tab_Temp``, for the
    reference cell (the engine scales by ``kappa_factor``) -- the working form of ``compute_Planck_opacities``
    (diffusion.f90:631-696), which the reference evaluates at a fixed T = 20 K only:

      chi        mean of the transport extinction kappa (1 - albedo g), harmonic (the diffusion coefficient is
                 D = 1/(3 chi)): the reference's ``rec_Planck_opacity``
      kappa_dep  mean of kappa_abs_LTE, arithmetic: the opacity the walk's path deposits energy with

    ``weights``: "dB_dT" (default) weighs wavelengths with dB/dT -- the spectrum a packet carries between two
    immediate re-emissions (``kdB_dT_CDF``), i.e. the Rosseland mean of Min et al. (2009); "B" weighs with the Planck
    function (Robitaille 2010; the reference's comments)."""
    wl = np.asarray(m.lam, f64) * 1.0e-6
    dwl = np.asarray(m.delta_lam, f64) * 1.0e-6
    # the asymmetry parameter the loop's scattering events actually have: with the tabulated phase function the
    # thermal step samples every wavelength's angle from the table of p_lambda = 1 (dust_transfer.f90:491-502)
    g = np.asarray(m.tab_g_pos, f64)
    if getattr(m.cfg, "lisotropic", False):
        g = np.zeros_like(g)
    elif int(getattr(m.cfg, "aniso_method", 1)) == 1 and m.p_lambda_fixed:
        g = np.full_like(g, g[0])
    n_T = m.tab_Temp.size
    nl = wl.size
    # lvariable_dust: one set of tables per class ([n_classes, n_T]; the walk of a cell reads its class's)
    vd = getattr(m, "variable_dust", None)
    if vd is not None:
        nc = int(vd["p_n_cells"])
        kap_c = np.asarray(vd["kappa"], f64).reshape(nl, nc).T
        alb_c = np.asarray(vd["albedo"], f64).reshape(nl, nc).T
        kab_c = np.asarray(vd["kappa_abs_LTE"], f64).reshape(nl, nc).T
        if vd.get("tab_g_pos") is not None:
            g_c = np.asarray(vd["tab_g_pos"], f64).reshape(nl, nc).T
            if getattr(m.cfg, "lisotropic", False):
                g_c = np.zeros_like(g_c)
            elif int(getattr(m.cfg, "aniso_method", 1)) == 1 and m.p_lambda_fixed:
                g_c = np.repeat(g_c[:, :1], nl, axis=1)
        else:
            g_c = np.tile(g, (nc, 1))
    else:
        nc = 1
        kap_c, alb_c, kab_c, g_c = (np.asarray(v, f64)[None, :] for v in (m.kappa, m.albedo, m.kappa_abs_LTE, g))
    chi, kdep, ext = np.zeros((nc, n_T), f64), np.zeros((nc, n_T), f64), np.zeros((nc, n_T), f64)
    exit_cdf = np.zeros((nc, n_T, nl), f64) if exit_spectrum == "in_flight" else None
    for c in range(nc):
        k_tr = kap_c[c] * (1.0 - alb_c[c] * g_c[c])
        k_abs = kab_c[c]
        for t in range(n_T):
            cst_wl = THERMAL_CONST / float(m.tab_Temp[t]) / wl
            ok = cst_wl < 500.0
            ce = np.exp(np.where(ok, cst_wl, 1.0))
            wgt = np.where(ok, 1.0 / ((wl ** 5) * (ce - 1.0)) * dwl, 0.0)
            if weights == "dB_dT":
                wgt = np.where(ok, wgt * cst_wl * ce / (ce - 1.0), 0.0)
            elif weights != "B":
                raise ValueError("init_mrw: weights is 'dB_dT' or 'B'")
            norm = wgt.sum()
            if norm > 0.0:
                chi[c, t] = norm / (wgt / k_tr).sum()
                kdep[c, t] = (wgt * k_abs).sum() / norm
                ext[c, t] = ext_factor * 0.7104 * (wgt / k_tr ** 2).sum() / (wgt / k_tr).sum()
                if exit_cdf is not None:
                    exit_cdf[c, t] = np.cumsum(wgt) / norm
            if exit_cdf is not None:
                exit_cdf[c, t, -1] = 1.0   # (every row ends in exactly 1, also one without weight: chi = 0 there, no walk)
    if vd is None:
        chi, kdep, ext = chi[0], kdep[0], ext[0]
    else:
        chi, kdep, ext = chi.reshape(-1), kdep.reshape(-1), ext.reshape(-1)
    # (the series saturates at 1 to within rounding for y > 0.9: the sampler wants a non-decreasing table)
    zeta = np.minimum(np.maximum.accumulate(cumulative_zeta(n_zeta)), 1.0)
    m.mrw = dict(zeta=zeta, chi=chi, kappa_dep=kdep, ext=ext, gamma=float(gamma),
                 n_inter=int(n_inter), weights=weights,
                 exit_cdf=None if exit_cdf is None else exit_cdf.reshape(-1))
    return m.mrw


def dark_zone_extent(m: "Model", lam: int, tau_max: float = 1500.0):
    """Steps 1-3 of ``define_dark_zone`` (optical_depth.f90:1459-1500, 1621-1628) on a 2D cylindrical grid: the radii
    ``ri_in_dark_zone`` / ``ri_out_dark_zone`` where the midplane optical depth at wavelength ``lam`` (1-based) exceeds
    ``tau_max`` from either side, and per radius the topmost layer ``zj_sup_dark_zone`` below which the vertical
    optical depth from the surface does.  The sums run in default real like the reference's ``total_sum``."""
    g = m.grid
    n_rad, nz = g["n_rad"], g["nz"]
    kap = float(m.kappa[lam - 1]) * np.asarray(m.kappa_factor, np.float64).reshape(-1)[:n_rad * nz].reshape(nz, n_rad)
    r_lim = np.asarray(g["r_lim"], np.float64)
    z_lim = np.asarray(g["z_lim"], np.float64).reshape(nz + 2, n_rad)[:nz + 1]
    tmax = np.float32(tau_max)

    def first_over(terms):
        tot = np.float32(0.0)
        for k, t in enumerate(terms):
            tot = np.float32(np.float64(tot) + t)
            if tot > tmax:
                return k
        return None

    k = first_over(kap[0] * np.diff(r_lim))
    ri_in = n_rad if k is None else k + 1
    k = first_over((kap[0] * np.diff(r_lim))[::-1])
    ri_out = 1 if k is None else n_rad - k
    if ri_out == n_rad:
        ri_out = n_rad - 1
    zj = np.zeros(n_rad, np.int32)
    for i in range(ri_in, ri_out + 1):
        k = first_over((kap[:, i - 1] * np.diff(z_lim[:, i - 1]))[::-1])
        zj[i - 1] = 0 if k is None else nz - k
    if ri_in <= ri_out:
        zj[:ri_in - 1] = zj[ri_in - 1]
        zj[ri_out:] = zj[ri_out - 1]
    return ri_in, ri_out, zj


def repartition_energie(m: "Model", Tdust):
    """thermal_emission.f90:1771-1949 for every wavelength (LTE grains, no ISM field, no
    emission weights): fills ``frac_E_stars``, ``frac_E_disk`` and ``prob_E_cell`` of the model
    for the SED step from the dust temperature of the thermal step."""
    cst_wl_max = math.log(float(np.finfo(f32).max)) - 1.0e-4
    nl, nc = m.n_lambda, m.n_cells
    Td = np.asarray(Tdust, f64)
    dark = np.zeros(nc, bool) if m.l_dark_zone is None else (np.asarray(m.l_dark_zone) != 0)
    pe = np.zeros((nl, nc + 1), f64)
    fs = np.zeros(nl, f64)
    fd = np.zeros(nl, f64)
    E_disk_all = []
    vol = np.asarray(m.grid["volume"], f64)
    vd = getattr(m, "variable_dust", None)
    for l in range(nl):
        wl = m.lam[l] * 1.0e-6
        E_cell = np.zeros(nc, f64)
        ok = (~dark) & (Td >= TINY_REAL)
        cst = np.full(nc, np.inf)
        cst[ok] = THERMAL_CONST / (Td[ok] * wl)
        ok &= cst < cst_wl_max
        if vd is not None:   # lvariable_dust: kappa_abs_LTE(p_icell, lambda) (thermal_emission.f90:1822)
            kabs = np.asarray(vd["kappa_abs_LTE"], f64).reshape(nl, -1)[l][np.asarray(vd["p_icell"]) - 1][ok]
        else:
            kabs = m.kappa_abs_LTE[l]
        E_cell[ok] = 4.0 * kabs * m.kappa_factor[ok] * vol[ok] / ((wl ** 5) * (np.exp(cst[ok]) - 1.0))
        E_disk = float(np.sum(E_cell))
        E_disk_all.append(E_disk)
        E_star = float(m.E_stars[l])
        fs[l] = E_star / (E_star + E_disk)
        fd[l] = 1.0  # (E_star + E_disk) / (E_star + E_disk + E_ISM), E_ISM = 0
        pe[l, 1:] = np.cumsum(E_cell)
        if pe[l, nc] > TINY_DP:
            pe[l] /= pe[l, nc]
        else:
            pe[l] = 0.0
    m.frac_E_stars, m.frac_E_disk, m.prob_E_cell = fs, fd, pe.reshape(-1)
    m.extra["E_disk"] = np.array(E_disk_all)   # E_disk(lambda) (thermal_emission.f90:1902)
    return m


def star_cell(grid, x, y, z):
    """index_cell_cyl for a point inside the inner hole or in the grid
    (stars.f90:789-808); only the inner-hole case is needed by the configs."""
    r2 = x * x + y * y + (z * z if grid.get("grid_type", 1) == 2 else 0.0)
    if r2 < grid["r_lim_2"][0]:
        i, j, k = 0, 1, 1
        idx = i + (grid["n_rad"] + 2) * ((j - grid["jdim_lo"]) + grid["jdim_n"] * (k - 1))
        return int(grid["cell_map"][idx])
    raise NotImplementedError("star inside the gridded region")


def build_model(cfg: DiskConfig, grid=None, rho=None) -> Model:
    """``grid``/``rho`` given: a Voronoi grid from ``host/voronoi.py`` with its own
    density (``build_voronoi_model``); otherwise the cylindrical grid of ``cfg``."""
    if grid is None:
        grid = define_cylindrical_grid(cfg)
    lam, lam_inf, lam_sup, dlam = init_lambda(cfg.n_lambda, cfg.lambda_min, cfg.lambda_max)
    tab_Temp = init_tab_temp(cfg.n_T, cfg.T_min, cfg.T_max)
    if rho is None:
        rho = dust_density(cfg, grid)
    nz_idx = np.nonzero(rho * grid["volume"] > TINY_REAL)[0]
    icell_ref = int(nz_idx[0])  # find_non_empty_cell (density.f90:2030)
    rho0 = rho[icell_ref]
    kappa_factor = rho / rho0  # dust_prop.f90:955

    d = synthetic_dust(cfg, lam)
    # kappa in AU^-1 at the reference cell: inverse of dust_prop.f90:1372
    kappa = d["kext"] * rho0 * AU_TO_CM
    albedo = d["albedo"].astype(f32)
    kappa_abs = kappa * (1.0 - d["albedo"])
    na1 = NANG_SCATT + 1
    prob = np.zeros((cfg.n_lambda, na1), f32)
    tab_s11 = np.zeros((cfg.n_lambda, na1), f32)
    dtheta = PI / NANG_SCATT
    th = np.arange(na1) * dtheta
    for l in range(cfg.n_lambda):
        k_sca = kappa[l] * float(albedo[l])
        # normalise s11 like tab_s11_pos: integral over sin(theta) dtheta = k_sca
        norm = float(np.sum(d["s11"][l][1:NANG_SCATT] * np.sin(th[1:NANG_SCATT]) * dtheta))
        s11n = d["s11"][l] * (k_sca / norm) * 0.97  # 3 % unresolved forward peak -> bin 1
        prob[l] = scattering_cdf(s11n, k_sca)
        tab_s11[l] = (s11n * dtheta / (k_sca * 2.0 * PI)).astype(f32)  # dust_prop.f90:1172

    log_Qcool, cdf = init_reemission(lam, dlam, tab_Temp, kappa_abs)

    E_stars, r_au = star_energy(cfg, lam, lam_inf, lam_sup)
    cum = np.zeros(cfg.n_lambda + 1, f64)
    for l in range(1, cfg.n_lambda + 1):  # thermal_emission.f90:329-341
        cum[l] = cum[l - 1] + E_stars[l - 1] * (dlam[l - 1] * 1.0e-6)
    cum = cum / cum[cfg.n_lambda]
    E_star_tot = float(np.sum(E_stars * dlam * 1.0e-6))
    L_tot = 2.0 * PI * HP * C_LIGHT ** 2 * E_star_tot  # :355

    sx, sy, sz = cfg.star_xyz
    if grid.get("grid_type", 1) == 3:  # Voronoi.f90:357-376: the star is a site of its own
        ic = int(grid["star_icell"][0])
        stars = np.array([[sx, sy, sz, r_au, ic, 0 if ic > 0 else 1]], f64)
    else:
        stars = np.array([[sx, sy, sz, r_au, star_cell(grid, sx, sy, sz), 0]], f64)
    CDF_E_star = np.zeros((2, cfg.n_lambda), f64)  # (lambda, 0:n_stars) Fortran order
    CDF_E_star[1, :] = 1.0

    return Model(
        cfg=cfg, grid=grid, lam=lam, delta_lam=dlam, kappa=kappa, kappa_abs_LTE=kappa_abs,
        albedo=albedo, kappa_factor=kappa_factor, prob_s11_pos=prob,
        s12_o_s11=d["s12_o_s11"].astype(f32), s22_o_s11=d["s22_o_s11"].astype(f32),
        s33_o_s11=d["s33_o_s11"].astype(f32), s34_o_s11=d["s34_o_s11"].astype(f32),
        s44_o_s11=d["s44_o_s11"].astype(f32), tab_g_pos=d["g"].astype(f32), tab_Temp=tab_Temp,
        log_Qcool=log_Qcool, kdB_dT_CDF=cdf, spectre_emission_cumul=cum,
        frac_E_stars=np.ones(cfg.n_lambda, f64), frac_E_disk=np.ones(cfg.n_lambda, f64),
        CDF_E_star=CDF_E_star.reshape(-1), E_stars=E_stars, L_tot=L_tot, stars=stars,
        rho_dust=rho, extra=dict(icell_ref=icell_ref + 1, rho0=rho0),
        tab_s11_pos=tab_s11, rt=init_directions_ray_tracing(cfg, bool(grid["l3D"])),
    )


VORONOI_CACHE_VERSION = 1


def build_voronoi_model(cfg: DiskConfig, n_sites: int, seed: int = 1, box_z_over_h: float = 6.0,
                        cut: bool = True, cache_dir: Optional[str] = None, tessellator=None, platonic: bool = False,
                        density: str = "smoothed", order: str = "file") -> Model:
    """BASELINE config 5 stand-in: ``n_sites`` SPH-like sites drawn from the cfg's disk,
    tessellated in a box (``Voronoi.f90:183-640`` hands the same arrays to the loop), the
    star added as its own site, densities from the analytic disk evaluated at the sites."""
    from . import voronoi as V

    sites = V.sample_disk_sites(cfg, n_sites, seed)
    site_id = np.arange(n_sites)
    if order == "morton":     # (cells along a space-filling curve; site_id maps a cell back to its particle)
        site_id = V.spatial_order(sites)
        sites = np.ascontiguousarray(sites[site_id])
    zlim = box_z_over_h * cfg.sclht * (cfg.rout / cfg.rref) ** cfg.exp_beta
    zlim = max(zlim, 1.001 * float(np.abs(sites[:, 2]).max()))
    L = 1.001 * cfg.rout
    limits = (-L, L, -L, L, -zlim, zlim)
    lam = init_lambda(cfg.n_lambda, cfg.lambda_min, cfg.lambda_max)[0]
    _, r_au = star_energy(cfg, lam, *init_lambda(cfg.n_lambda, cfg.lambda_min, cfg.lambda_max)[1:3])
    sx, sy, sz = cfg.star_xyz
    # SPH smoothing length h = 1.2 (m/rho)^(1/3) from the analytic number density of the sites
    rs = np.hypot(sites[:, 0], sites[:, 1])
    Hs = cfg.sclht * (rs / cfg.rref) ** cfg.exp_beta
    p = cfg.surf + 2.0
    pdf_r = p * rs ** (p - 1.0) / (cfg.rout ** p - cfg.rin ** p)  # per unit r
    dens = n_sites * pdf_r / (2 * PI * rs) * np.exp(-0.5 * (sites[:, 2] / Hs) ** 2) / (math.sqrt(2 * PI) * Hs)
    h = 1.2 * np.cbrt(1.0 / dens)
    # (the tessellation of 1e6 sites takes minutes on the host: ``cache_dir`` keeps it between runs of one model)
    grid, cache = None, None
    if cache_dir:
        import hashlib
        import os
        # the key names everything the tessellation depends on: the disk that the sites sample, the box, the star's
        # site, and the version of the builder (bump VORONOI_CACHE_VERSION when host/voronoi.py changes its output)
        geo = repr((VORONOI_CACHE_VERSION, n_sites, seed, box_z_over_h, int(cut), cfg.rin, cfg.rout, cfg.sclht, cfg.rref,
                    cfg.exp_beta, cfg.surf, tuple(cfg.star_xyz), float(r_au), limits) + ((int(platonic),) if platonic else ()) + ((order,) if order != "file" else ()))
        cache = os.path.join(cache_dir, "voronoi_%d_%s.npz" % (n_sites, hashlib.sha1(geo.encode()).hexdigest()[:12]))
        if os.path.exists(cache):
            z = np.load(cache)
            grid = {k: (z[k] if z[k].ndim else z[k].item()) for k in z.files}
    if grid is None:
        # (tessellator: None = scipy on the host; V.device_tessellator() = the product's kernel, mcgpu_voronoi_tesselation --
        # the same grid bit for bit (tests/test_tessellation.py); platonic: the reference's cuts in full, host/voronoi.py)
        grid = V.build_voronoi_grid(sites, limits, stars_xyz_r=[(sx, sy, sz, r_au)], h=h, cut=cut, tessellator=tessellator,
                                    platonic=platonic)
        if cache:
            # written under a private name and moved into place: concurrent builders (one per rank) never leave a torn file
            os.makedirs(cache_dir, exist_ok=True)
            tmp = "%s.%d.tmp.npz" % (cache, os.getpid())
            np.savez(tmp, **grid)
            os.replace(tmp, cache)
    # Density: what an SPH dump hands over is the kernel-SMOOTHED density of equal-mass particles.  The raw m / V_i of
    # randomly drawn sites is not that: Poisson-Voronoi volumes scatter by ~40 %, the small cells become clumps many
    # times denser than the disk, and the packets trapped in them -- 119 interactions per packet against 11 in the same
    # disk on the cylindrical grid -- make it another problem.  So rho_i is the geometric mean of m / V over the cell
    # and its Voronoi neighbours (~16: the kernel of a tessellation; the mean of the logarithms, so that a large
    # neighbour does not drain a small cell and an exponential profile stays what it is), normalised to the dust mass.
    # Star sites carry no dust.
    nb = grid["n_cells_before_stars"]
    vol = np.asarray(grid["volume"], f64)
    first, last, neigh = np.asarray(grid["v_first"]), np.asarray(grid["v_last"]), np.asarray(grid["v_neigh"])
    owner = np.repeat(np.arange(grid["n_cells"]), last - first + 1)       # CSR rows (1-based first / last)
    gas = (neigh >= 1) & (neigh <= nb) & (owner < nb)                     # dusty neighbours of dusty cells
    # (the density estimate takes the cell as voro++ first builds it; the cut only reduces the volume the density fills,
    # voro++_wrapper.cpp:209-227 -- so that a cut changes a cell's mass, not the disk around it)
    lv = np.log(np.asarray(grid.get("volume_uncut", vol), f64)[:nb])
    l_sum = lv + np.bincount(owner[gas], weights=lv[neigh[gas] - 1], minlength=nb)[:nb]
    n_sum = 1.0 + np.bincount(owner[gas], minlength=nb)[:nb]
    rho = np.zeros(grid["n_cells"], f64)
    if density == "sph":
        # ... or the density the dump itself carries: an SPH code knows rho_i = m (hfact / h_i)^3 from its own kernel sum
        # (hfact = 1.2), the reference reads it with the particles and gives every cell ITS particle's density -- the
        # tessellation only supplies the volumes (cut for elongated cells, so that the mass rho V of a particle at the
        # disk's ragged surface is not that of its huge Voronoi cell; voro++_wrapper.cpp:209-227).  With h from the
        # analytic number density of the sites this is the cfg's disk sampled at the sites, without the Poisson noise
        # of m / V: the Voronoi run then is the ref4.1 disk of the cylindrical runs (same optical depths: ~11
        # interactions per packet instead of the 130-210 of the smoothed m / V at 1e6 sites).
        rho[:nb] = (1.2 / np.asarray(grid["v_h"], f64)[:nb]) ** 3
    elif density == "smoothed":
        rho[:nb] = np.exp(-l_sum / n_sum)
    else:
        raise ValueError("build_voronoi_model: density is 'smoothed' or 'sph'")
    rho[:nb] *= cfg.dust_mass * MSUN_TO_G / (float(np.sum(rho[:nb] * vol[:nb])) * AU_TO_CM ** 3)
    m = build_model(cfg, grid=grid, rho=rho)
    m.extra["site_id"] = site_id       # cell i (0-based, before the stars' cells) is particle site_id[i] of the sample
    return m
