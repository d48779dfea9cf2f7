"""ctypes bindings for the CPU oracle (``libmc_oracle.so``) and for the
compiled reference geometry (``oracle/_ref/libmcfost_ref_geom.so``).

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
N_SED_TYPES = 9
N_COUNTERS = 10
COUNTER_NAMES = ("packets", "crossings", "flights", "scatterings", "absorptions",
                 "escaped", "killed_star", "dark_mirrors", "mrw_walks", "mrw_steps")


def oracle_lib_path():
    return os.path.join(_HERE, "libmc_oracle.so")


def ref_lib_path():
    return os.path.join(_HERE, "_ref", "libmcfost_ref_geom.so")


def build_oracle(force=False):
    """Compile the C restatement with gcc (plain IEEE evaluation: no FMA
    contraction, so it can be compared bit-for-bit with ``oracle/_ref``)."""
    out = oracle_lib_path()
    src = os.path.join(_HERE, "mc_oracle.c")
    if force or not os.path.exists(out) or os.path.getmtime(out) < max(
            os.path.getmtime(src), os.path.getmtime(os.path.join(_HERE, "mc_oracle.h"))):
        subprocess.check_call(
            ["gcc", "-O2", "-fPIC", "-shared", "-fopenmp", "-ffp-contract=off", "-Wall",
             "-o", out, src, "-lm"])
    return out


def build_ref(force=False):
    """Build ``oracle/_ref`` from the reference sources where they lie.  Only
    possible where ``/root/reference`` exists (not on the GPU box, which uses
    the prebuilt file).  Returns the path or ``None``."""
    out = ref_lib_path()
    if os.path.exists(out) and not force:
        return out
    if not os.path.isdir("/root/reference/src"):
        return out if os.path.exists(out) else None
    subprocess.check_call(["make", "-C", os.path.join(_HERE, "ref_build")],
                          stdout=subprocess.DEVNULL)
    return out if os.path.exists(out) else None


class _Star(C.Structure):
    _fields_ = [("x", C.c_double), ("y", C.c_double), ("z", C.c_double), ("r", C.c_double),
                ("icell", C.c_int), ("out_model", C.c_int)]


_dp = C.POINTER(C.c_double)
_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int)
_up = C.POINTER(C.c_ubyte)


class _Model(C.Structure):
    _fields_ = [
        ("n_rad", C.c_int), ("nz", C.c_int), ("n_az", C.c_int), ("l3D", C.c_int),
        ("n_cells", C.c_int), ("ntot2", C.c_int), ("jdim_lo", C.c_int), ("jdim_n", C.c_int),
        ("r_lim_2", _dp), ("zmax", _dp), ("z_lim", _dp), ("tan_phi_lim", _dp),
        ("zmaxmax", C.c_double), ("Rmax2", C.c_double),
        ("cell_map", _ip), ("cell_map_i", _ip), ("cell_map_j", _ip), ("cell_map_k", _ip),
        ("lexit_cell", _ip), ("volume", _dp),
        ("n_stars", C.c_int), ("stars", C.POINTER(_Star)),
        ("n_lambda", C.c_int), ("kappa", _dp), ("kappa_abs_LTE", _dp), ("albedo", _fp),
        ("kappa_factor", _dp), ("l_dark_zone", _up),
        ("nang_scatt", C.c_int), ("aniso_method", C.c_int), ("lisotropic", C.c_int),
        ("lsepar_pola", C.c_int), ("p_lambda_fixed", C.c_int),
        ("prob_s11_pos", _fp), ("s12_o_s11", _fp), ("s22_o_s11", _fp), ("s33_o_s11", _fp),
        ("s34_o_s11", _fp), ("s44_o_s11", _fp), ("tab_g_pos", _fp),
        ("n_T", C.c_int), ("tab_Temp", _fp), ("log_Qcool", _dp), ("kdB_dT_CDF", _dp),
        ("spectre_emission_cumul", _dp), ("frac_E_stars", _dp), ("frac_E_disk", _dp),
        ("CDF_E_star", _dp), ("prob_E_cell", _dp), ("L_packet_th", C.c_double),
        ("T_min", C.c_float),
        ("N_thet", C.c_int), ("N_phi", C.c_int), ("l_sym_centrale", C.c_int),
        ("l_sym_axiale", C.c_int), ("midplane_snap", C.c_int),
        ("grid_type", C.c_int), ("v_xyz", _fp), ("v_xyz_dp", _dp), ("v_h", _dp),
        ("v_first", _ip), ("v_last", _ip), ("v_neigh", _ip), ("v_was_cut", _up),
        ("v_is_star_neighbour", _up), ("v_walls", _fp), ("v_cut_o_h", C.c_double),
        ("v_wall_first", _ip), ("v_wall_cells", _ip),
        ("tan_theta_lim", _dp), ("theta_lim", _dp), ("r_lim_3", _dp),
        ("R_ISM", C.c_double), ("centre_ISM", C.c_double * 3),
        ("RT_n_incl", C.c_int), ("RT_n_az", C.c_int), ("tab_u_rt", _dp), ("tab_v_rt", _dp),
        ("tab_w_rt", _dp), ("n_az_rt", C.c_int), ("n_theta_rt", C.c_int), ("N_type_flux", C.c_int),
        ("lsepar_contrib", C.c_int), ("tab_s11_pos", _fp),
        ("mrw", C.c_int), ("mrw_n_zeta", C.c_int), ("mrw_zeta", _dp), ("mrw_chi", _dp), ("mrw_kappa_dep", _dp),
        ("mrw_ext", _dp), ("mrw_exit_cdf", _dp), ("mrw_gamma", C.c_float), ("mrw_n_inter", C.c_int),
        ("p_n_cells", C.c_int), ("p_icell", _ip), ("v_kappa", _dp), ("v_kappa_abs_LTE", _dp), ("v_albedo", _fp),
        ("v_log_Qcool", _dp), ("v_kdB_dT_CDF", _dp), ("v_prob_s11_pos", _fp), ("v_s12_o_s11", _fp), ("v_s22_o_s11", _fp),
        ("v_s33_o_s11", _fp), ("v_s34_o_s11", _fp), ("v_s44_o_s11", _fp), ("v_tab_g_pos", _fp), ("r_lim", _dp),
        ("v_tab_s11_pos", _fp),
        ("scattering_method1", C.c_int), ("m1_n_grains", C.c_int), ("m1_C_sca", _fp), ("m1_nk", _dp), ("m1_dens", _dp), ("m1_ksca_CDF", _dp),
        ("m1_prob_s11", _fp), ("m1_tab_g", _fp), ("m1_s11", _fp), ("m1_s12", _fp), ("m1_s22", _fp), ("m1_s33", _fp),
        ("m1_s34", _fp), ("m1_s44", _fp), ("sin_phi_lim", _dp), ("cos_phi_lim", _dp),
    ]


class _MonoOpts(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("lambda_", C.c_int), ("p_lambda", C.c_int), ("n_chunks", C.c_int),
                ("first_chunk", C.c_int), ("n_photons2", C.c_double), ("n_phot_lim", C.c_double), ("capt_sup", C.c_int),
                ("rt1", C.c_int), ("n_threads", C.c_int), ("n_theta_I", C.c_int), ("n_phi_I", C.c_int),
                ("I_spec", C.POINTER(C.c_double)), ("I_spec_star", C.POINTER(C.c_double))]


class _RtOpts(C.Structure):
    _fields_ = [("lambda_", C.c_int), ("wl_um", C.c_double), ("E_src", C.c_double), ("n_sent_photons", C.c_double),
                ("distance", C.c_double), ("ang_disque", C.c_double), ("l_sym_ima", C.c_int),
                ("tau_dark_zone_obs", C.c_double), ("Rmin", C.c_double), ("Rmax", C.c_double), ("tab_RT_az", _fp),
                ("n_threads", C.c_int)]


class _Opts(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("first_packet", C.c_uint64), ("n_packets", C.c_uint64),
                ("n_threads", C.c_int), ("frozen", C.c_int), ("tau_fp32", C.c_int),
                ("n_replicas", C.c_double)]


def _a(x, dt):
    return np.ascontiguousarray(x, dtype=dt)


def _p(arr, ct):
    return arr.ctypes.data_as(C.POINTER(ct))


class Oracle:
    """The CPU oracle bound to one model (``mcfost_amd.host.model.Model``-like
    object: duck-typed, nothing from the product is imported here)."""

    def __init__(self, model, n_packets_total):
        self.lib = C.CDLL(build_oracle())
        self.model = model
        self._keep = []
        self.cm = self._make_struct(model, float(n_packets_total))
        L = self.lib
        L.oracle_run_thermal.restype = C.c_int
        L.oracle_packet_rand.restype = C.c_float
        L.oracle_packet_rand.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32]

    def _hold(self, arr, ct):
        self._keep.append(arr)
        return _p(arr, ct)

    def _make_struct(self, m, n_tot):
        g, cfg = m.grid, m.cfg
        s = _Model()
        s.grid_type = int(g.get("grid_type", 1))
        if s.grid_type == 3:
            s.n_cells, s.l3D, s.n_az = int(g["n_cells"]), 1, 1
            s.v_xyz = self._hold(_a(g["v_xyz"], np.float32), C.c_float)
            s.v_xyz_dp = self._hold(_a(g["v_xyz_dp"], np.float64), C.c_double)
            s.v_h = self._hold(_a(g["v_h"], np.float64), C.c_double)
            for k in ("v_first", "v_last", "v_neigh", "v_wall_first", "v_wall_cells"):
                setattr(s, k, self._hold(_a(g[k], np.int32), C.c_int))
            s.v_was_cut = self._hold(_a(g["v_was_cut"], np.uint8), C.c_ubyte)
            s.v_is_star_neighbour = self._hold(_a(g["v_is_star_neighbour"], np.uint8), C.c_ubyte)
            s.v_walls = self._hold(_a(g["v_walls"], np.float32), C.c_float)
            s.v_cut_o_h = float(g["v_cut_o_h"])
        else:
            for k in ("n_rad", "nz", "n_az", "l3D", "n_cells", "ntot2", "jdim_lo", "jdim_n"):
                setattr(s, k, int(g[k]))
            s.r_lim_2 = self._hold(_a(g["r_lim_2"], np.float64), C.c_double)
            s.zmax = self._hold(_a(g["zmax"], np.float64), C.c_double)
            s.z_lim = self._hold(_a(g["z_lim"], np.float64), C.c_double)
            s.tan_phi_lim = self._hold(_a(g["tan_phi_lim"], np.float64), C.c_double)
            s.zmaxmax = float(g["zmaxmax"])
            s.Rmax2 = float(g["Rmax2"])
            for k in ("cell_map", "cell_map_i", "cell_map_j", "cell_map_k", "lexit_cell"):
                setattr(s, k, self._hold(_a(g[k], np.int32), C.c_int))
            if s.grid_type == 2:  # spherical grid: cylindrical_grid.f90:28-31
                for k in ("tan_theta_lim", "theta_lim", "r_lim_3"):
                    setattr(s, k, self._hold(_a(g[k], np.float64), C.c_double))
        s.volume = self._hold(_a(g["volume"], np.float64), C.c_double)
        ns = m.stars.shape[0]
        stars = (_Star * ns)()
        for i in range(ns):
            x, y, z, r, ic, om = m.stars[i]
            stars[i] = _Star(x, y, z, r, int(ic), int(om))
        self._keep.append(stars)
        s.n_stars = ns
        s.stars = stars
        s.n_lambda = m.n_lambda
        s.kappa = self._hold(_a(m.kappa, np.float64), C.c_double)
        s.kappa_abs_LTE = self._hold(_a(m.kappa_abs_LTE, np.float64), C.c_double)
        s.albedo = self._hold(_a(m.albedo, np.float32), C.c_float)
        s.kappa_factor = self._hold(_a(m.kappa_factor, np.float64), C.c_double)
        if m.l_dark_zone is not None:
            s.l_dark_zone = self._hold(_a(m.l_dark_zone, np.uint8), C.c_ubyte)
        s.nang_scatt = 180
        s.aniso_method = cfg.aniso_method
        s.lisotropic = int(cfg.lisotropic)
        s.lsepar_pola = int(cfg.lsepar_pola)
        s.p_lambda_fixed = int(m.p_lambda_fixed)
        for k in ("prob_s11_pos", "s12_o_s11", "s22_o_s11", "s33_o_s11", "s34_o_s11",
                  "s44_o_s11", "tab_g_pos"):
            setattr(s, k, self._hold(_a(getattr(m, k), np.float32), C.c_float))
        s.n_T = m.tab_Temp.size
        s.tab_Temp = self._hold(_a(m.tab_Temp, np.float32), C.c_float)
        s.log_Qcool = self._hold(_a(m.log_Qcool, np.float64), C.c_double)
        s.kdB_dT_CDF = self._hold(_a(m.kdB_dT_CDF, np.float64), C.c_double)
        s.spectre_emission_cumul = self._hold(_a(m.spectre_emission_cumul, np.float64), C.c_double)
        s.frac_E_stars = self._hold(_a(m.frac_E_stars, np.float64), C.c_double)
        s.frac_E_disk = self._hold(_a(m.frac_E_disk, np.float64), C.c_double)
        s.CDF_E_star = self._hold(_a(m.CDF_E_star, np.float64), C.c_double)
        pe = getattr(m, "prob_E_cell", None)
        if pe is not None:
            s.prob_E_cell = self._hold(_a(pe, np.float64), C.c_double)
        s.L_packet_th = m.L_packet_th(n_tot)
        s.T_min = float(cfg.T_min)
        s.N_thet, s.N_phi = cfg.N_thet, cfg.N_phi
        s.l_sym_centrale, s.l_sym_axiale = int(cfg.l_sym_centrale), int(cfg.l_sym_axiale)
        s.midplane_snap = int(getattr(m, "midplane_snap", 0))
        ism = getattr(m, "ism", None)
        if ism is not None:
            s.R_ISM = float(ism["R_ISM"])
            for q in range(3):
                s.centre_ISM[q] = float(ism["centre_ISM"][q])
        rt = getattr(m, "rt", None)
        if rt is not None:
            s.RT_n_incl, s.RT_n_az = int(rt["RT_n_incl"]), int(rt["RT_n_az"])
            for k in ("tab_u_rt", "tab_v_rt", "tab_w_rt"):
                setattr(s, k, self._hold(_a(rt[k], np.float64), C.c_double))
            s.n_az_rt, s.n_theta_rt = int(rt["n_az_rt"]), int(rt["n_theta_rt"])
            s.N_type_flux, s.lsepar_contrib = int(rt["N_type_flux"]), int(rt["lsepar_contrib"])
            s.tab_s11_pos = self._hold(_a(m.tab_s11_pos, np.float32), C.c_float)
        if "r_lim" in g:
            s.r_lim = self._hold(_a(g["r_lim"], np.float64), C.c_double)
        vd = getattr(m, "variable_dust", None)
        if vd is not None:   # reference layouts (mcfost_amd.host.model.init_variable_dust)
            s.p_n_cells = int(vd["p_n_cells"])
            s.p_icell = self._hold(_a(vd["p_icell"], np.int32), C.c_int)
            s.v_kappa = self._hold(_a(vd["kappa"], np.float64), C.c_double)
            s.v_kappa_abs_LTE = self._hold(_a(vd["kappa_abs_LTE"], np.float64), C.c_double)
            s.v_albedo = self._hold(_a(vd["albedo"], np.float32), C.c_float)
            s.v_log_Qcool = self._hold(_a(vd["log_Qcool"], np.float64), C.c_double)
            s.v_kdB_dT_CDF = self._hold(_a(vd["kdB_dT_CDF"], np.float64), C.c_double)
            if vd.get("prob_s11_pos") is not None:
                for k, f in (("prob_s11_pos", "v_prob_s11_pos"), ("s12_o_s11", "v_s12_o_s11"), ("s22_o_s11", "v_s22_o_s11"),
                             ("s33_o_s11", "v_s33_o_s11"), ("s34_o_s11", "v_s34_o_s11"), ("s44_o_s11", "v_s44_o_s11"),
                             ("tab_g_pos", "v_tab_g_pos")):
                    setattr(s, f, self._hold(_a(vd[k], np.float32), C.c_float))
            if vd.get("tab_s11_pos") is not None:
                s.v_tab_s11_pos = self._hold(_a(vd["tab_s11_pos"], np.float32), C.c_float)
        if g.get("l3D") and g.get("grid_type", 1) in (1, 2) and "tan_phi_lim" in g:   # the walk's azimuthal walls
            from mcfost_amd.host.model import phi_wall_sin_cos
            sp, cp = g.get("sin_phi_lim"), g.get("cos_phi_lim")
            if sp is None:
                sp, cp = phi_wall_sin_cos(m.cfg)
            s.sin_phi_lim = self._hold(_a(sp, np.float64), C.c_double)
            s.cos_phi_lim = self._hold(_a(cp, np.float64), C.c_double)
        m1 = getattr(m, "method1", None)
        if m1 is not None:   # scattering method 1 (mcfost_amd.host.model.init_scattering_method1)
            s.scattering_method1, s.m1_n_grains = 1, int(m1["n_grains"])
            s.m1_nk = self._hold(_a(m1["n_grains_k"], np.float64), C.c_double)
            s.m1_dens = self._hold(_a(m1["dens"], np.float64), C.c_double)
            for k, f in (("C_sca", "m1_C_sca"), ("prob_s11", "m1_prob_s11"), ("tab_g", "m1_tab_g"), ("tab_s11", "m1_s11"),
                         ("tab_s12", "m1_s12"), ("tab_s22", "m1_s22"), ("tab_s33", "m1_s33"), ("tab_s34", "m1_s34"),
                         ("tab_s44", "m1_s44")):
                setattr(s, f, self._hold(_a(m1[k], np.float32), C.c_float))
            if m1.get("ksca_CDF") is not None:   # the high-memory grain selection (oracle_build_ksca_CDF / Oracle.build_ksca_CDF)
                s.m1_ksca_CDF = self._hold(_a(m1["ksca_CDF"], np.float64), C.c_double)
        mrw = getattr(m, "mrw", None)
        if mrw is not None:
            s.mrw, s.mrw_n_zeta = 1, int(mrw["zeta"].size)
            for k in ("zeta", "chi", "kappa_dep", "ext"):
                setattr(s, "mrw_" + k, self._hold(_a(mrw[k], np.float64), C.c_double))
            if mrw.get("exit_cdf") is not None:
                s.mrw_exit_cdf = self._hold(_a(mrw["exit_cdf"], np.float64), C.c_double)
            s.mrw_gamma, s.mrw_n_inter = float(mrw["gamma"]), int(mrw["n_inter"])
        return s

    def find_voronoi_cell(self, iwall, x, y, z):
        return np.array([self.lib.oracle_find_voronoi_cell(C.byref(self.cm), C.c_int(int(iwall)), C.c_double(a), C.c_double(b),
                                                           C.c_double(c)) for a, b, c in zip(x, y, z)], np.int32)

    def distance_to_closest_wall(self, icell, x, y, z):
        self.lib.oracle_distance_to_closest_wall_cyl.restype = C.c_double
        return np.array([self.lib.oracle_distance_to_closest_wall_cyl(C.byref(self.cm), C.c_int(int(c)), C.c_double(a),
                                                                        C.c_double(b), C.c_double(d))
                         for c, a, b, d in zip(icell, x, y, z)])

    def mrw_zeta_table(self, n=10000):
        out = np.zeros(n)
        self.lib.oracle_mrw_zeta_table(C.c_int(n), _p(out, C.c_double))
        return out

    def mrw_sample_y(self, xi):
        self.lib.oracle_mrw_sample_y.restype = C.c_double
        return np.array([self.lib.oracle_mrw_sample_y(C.byref(self.cm), C.c_float(float(v))) for v in xi])

    def xI_shape(self):
        """xI_scatt(n_az_rt, n_theta_rt, N_type_flux, RT_n_incl*RT_n_az, n_cells), C order reversed"""
        rt = self.model.rt
        return (self.model.n_cells, rt["RT_n_incl"] * rt["RT_n_az"], rt["N_type_flux"], rt["n_theta_rt"],
                rt["n_az_rt"])

    def run_mono(self, lam, n_photons2, n_phot_lim=None, p_lambda=None, seed=1, n_chunks=None,
                 rt1=True, n_threads=1, first_chunk=0, rt2=None):
        """One wavelength (1-based ``lam``) of the SED Monte Carlo.  ``rt2 = (n_theta_I, n_phi_I)``: ray tracing method
        2's deposits instead of method 1's (``I_spec [n_cells, n_phi_I, n_theta_I, N_type_flux]``, ``I_spec_star``)."""
        m = self.model
        nl, nt, nphi = m.n_lambda, m.cfg.N_thet, m.cfg.N_phi
        n_chunks = int(n_chunks or m.cfg.n_photons_loop)
        if n_phot_lim is None:  # read_param.f90:551
            n_phot_lim = float(np.float32(1.0e4) * np.float32(nt) * np.float32(nphi) * np.float32(n_photons2))
        I_spec = I_star = None
        if rt2 is not None:
            rt1 = False
            ntf = (4 if (m.cfg.lsepar_pola and m.cfg.aniso_method == 1) else 1) + (4 if m.cfg.lsepar_contrib else 0)
            I_spec = np.zeros((m.n_cells, int(rt2[1]), int(rt2[0]), ntf), np.float64)
            I_star = np.zeros(m.n_cells, np.float64)
        o = _MonoOpts(seed, int(lam), int(p_lambda or lam), n_chunks, int(first_chunk), float(n_photons2), float(n_phot_lim),
                      int(m.capt_sup), 2 if rt2 is not None else int(rt1), n_threads,
                      int(rt2[0]) if rt2 is not None else 0, int(rt2[1]) if rt2 is not None else 0,
                      _p(I_spec, C.c_double) if rt2 is not None else None, _p(I_star, C.c_double) if rt2 is not None else None)
        xI = np.zeros(self.xI_shape() if rt1 else (1,), np.float64)
        sed = np.zeros((N_SED_TYPES, nphi, nt, nl), np.float64)
        n_sent = np.zeros(nl, np.float64)
        per_chunk = np.zeros(n_chunks, np.uint64)
        cnt = np.zeros(N_COUNTERS, np.uint64)
        self.lib.oracle_run_mono.restype = C.c_int
        rc = self.lib.oracle_run_mono(C.byref(self.cm), C.byref(o), _p(xI, C.c_double), _p(sed, C.c_double),
                                      _p(n_sent, C.c_double), _p(per_chunk, C.c_uint64), _p(cnt, C.c_uint64))
        if rc:
            raise RuntimeError(f"oracle_run_mono failed: {rc}")
        return dict(xI_scatt=xI, sed=sed, n_sent=n_sent, n_sent_chunk=per_chunk, I_spec=I_spec, I_spec_star=I_star,
                    counters=dict(zip(COUNTER_NAMES, (int(c) for c in cnt))))

    # -- packet loop -------------------------------------------------------
    def run_thermal(self, n_packets, seed=1, first_packet=0, n_threads=1, frozen=False,
                    E_prior=None, tau_fp32=False, n_replicas=1.0):
        m = self.model
        nl, nt, nphi = m.n_lambda, m.cfg.N_thet, m.cfg.N_phi
        E = np.zeros(m.n_cells, np.float64)
        sed = np.zeros((N_SED_TYPES, nphi, nt, nl), np.float64)
        n_sent = np.zeros(nl, np.float64)
        cnt = np.zeros(N_COUNTERS, np.uint64)
        o = _Opts(seed, first_packet, n_packets, n_threads, int(frozen), int(tau_fp32),
                  float(n_replicas))
        ep = None
        if E_prior is not None:
            ep = _a(E_prior, np.float64)
        rc = self.lib.oracle_run_thermal(
            C.byref(self.cm), C.byref(o), _p(ep, C.c_double) if ep is not None else None,
            _p(E, C.c_double), _p(sed, C.c_double), _p(n_sent, C.c_double), _p(cnt, C.c_uint64))
        if rc:
            raise RuntimeError(f"oracle_run_thermal failed: {rc}")
        return dict(E_abs=E, sed=sed, n_sent=n_sent,
                    counters=dict(zip(COUNTER_NAMES, (int(c) for c in cnt))))

    def run_thermal_with_radiation_field(self, n_packets, **kw):
        """run_thermal + xN_abs[n_cells], xJ_abs[n_lambda, n_cells] (radiation_field.f90:54-55)."""
        m = self.model
        xN, xJ = np.zeros(m.n_cells), np.zeros((m.n_lambda, m.n_cells))
        self.lib.oracle_set_radiation_field_outputs(_p(xN, C.c_double), _p(xJ, C.c_double))
        try:
            res = self.run_thermal(n_packets, **kw)
        finally:
            self.lib.oracle_set_radiation_field_outputs(None, None)
        res["xN_abs"], res["xJ_abs"] = xN, xJ
        return res

    def temp_finale(self, E_abs):
        T = np.zeros(self.model.n_cells, np.float32)
        E = _a(E_abs, np.float64)
        self.lib.oracle_temp_finale(C.byref(self.cm), _p(E, C.c_double), _p(T, C.c_float))
        return T

    def dust_map_sed(self, lam, xI_scatt, Tdust, n_sent_photons, E_disk, ang_disque=0.0, l_sym_ima=True,
                     tau_dark_zone_obs=100.0, n_threads=1):
        """Ray-traced SED of the dust at wavelength ``lam`` (1-based), RT method 1: (nRT, N_type_flux)."""
        m = self.model
        rt = m.rt
        az = _a(rt["tab_RT_az"], np.float32)
        o = _RtOpts(int(lam), float(m.lam[lam - 1]), float(m.E_stars[lam - 1] + E_disk), float(n_sent_photons),
                    float(m.cfg.distance), float(ang_disque), int(l_sym_ima), float(tau_dark_zone_obs),
                    float(m.cfg.rin), float(m.cfg.rout), _p(az, C.c_float), int(n_threads))
        nRT = rt["RT_n_incl"] * rt["RT_n_az"]
        out = np.zeros((nRT, rt["N_type_flux"]), np.float64)
        x = _a(xI_scatt, np.float64)
        T = _a(Tdust, np.float32)
        self.lib.oracle_dust_map_sed.restype = C.c_int
        rc = self.lib.oracle_dust_map_sed(C.byref(self.cm), C.byref(o), _p(x, C.c_double), _p(T, C.c_float),
                                          _p(out, C.c_double))
        if rc:
            raise RuntimeError(f"oracle_dust_map_sed failed: {rc}")
        return out

    def rt2_dust_map_sed(self, lam, ibin, eps_dust2, eps_dust2_star, Tdust, n_sent_photons, E_disk, l_sym_ima=True,
                         tau_dark_zone_obs=100.0, n_threads=1):
        """Ray-traced SED of the dust of inclination ``ibin`` with method 2's source function (``init_dust_source_fct2``'s
        ``eps_dust2 [n_cells, 2, nang_rt, N_type_flux]`` / ``eps_dust2_star``): (N_type_flux,)."""
        m = self.model
        e2, es = _a(eps_dust2, np.float32), _a(eps_dust2_star, np.float32)
        zg = _a(np.abs(m.grid["z_grid"]), np.float64)
        self.lib.oracle_set_rt2_source(_p(e2, C.c_float), _p(es, C.c_float), C.c_int(e2.shape[2]), C.c_int(es.shape[2]),
                                       C.c_int(int(ibin)), _p(zg, C.c_double))
        try:
            out = self.dust_map_sed(lam, np.zeros(1), Tdust, n_sent_photons, E_disk, 0.0, l_sym_ima, tau_dark_zone_obs, n_threads)
        finally:
            self.lib.oracle_set_rt2_source(None, None, C.c_int(0), C.c_int(0), C.c_int(0), None)
        return out[int(ibin) - 1]

    def rt2_dust_map_image(self, lam, ibin, eps_dust2, eps_dust2_star, Tdust, n_sent_photons, E_disk, npix_x, npix_y, map_size,
                           zoom=1.0, l_sym_ima=False, tau_dark_zone_obs=100.0, n_threads=1):
        """The image of inclination ``ibin`` with method 2's source function: (N_type_flux, npix_y, npix_x), rays traced."""
        m = self.model
        e2, es = _a(eps_dust2, np.float32), _a(eps_dust2_star, np.float32)
        zg = _a(np.abs(m.grid["z_grid"]), np.float64)
        self.lib.oracle_set_rt2_source(_p(e2, C.c_float), _p(es, C.c_float), C.c_int(e2.shape[2]), C.c_int(es.shape[2]),
                                       C.c_int(int(ibin)), _p(zg, C.c_double))
        try:
            img, n = self.dust_map_image(lam, np.zeros(1), Tdust, n_sent_photons, E_disk, npix_x, npix_y, map_size, zoom, 0.0,
                                         l_sym_ima, tau_dark_zone_obs, n_threads)
        finally:
            self.lib.oracle_set_rt2_source(None, None, C.c_int(0), C.c_int(0), C.c_int(0), None)
        return img[:, 0, int(ibin) - 1], n

    def stars_map_sed(self, lam, star_flux, seed=1, ang_disque=0.0):
        """compute_stars_map for the SED: the stars' flux towards every observer, (nRT,)."""
        m = self.model
        rt = m.rt
        az = _a(rt["tab_RT_az"], np.float32)
        o = _RtOpts(int(lam), float(m.lam[lam - 1]), 1.0, 1.0, float(m.cfg.distance), float(ang_disque), 0, 100.0,
                    float(m.cfg.rin), float(m.cfg.rout), _p(az, C.c_float), 1)
        out = np.zeros(rt["RT_n_incl"] * rt["RT_n_az"])
        rc = self.lib.oracle_stars_map_sed(C.byref(self.cm), C.byref(o), C.c_uint64(int(seed)),
                                           _p(_a(star_flux, np.float64), C.c_double), _p(out, C.c_double))
        if rc:
            raise RuntimeError(f"oracle_stars_map_sed failed: {rc}")
        return out

    def tau_maps(self, lam, npix_x, npix_y, map_size, zoom=1.0, tau=1.0, ang_disque=0.0, surface=True):
        """compute_tau_map / compute_tau_surface_map for every observer: ``(tau_map [nRT, npix_y, npix_x],
        tau_surface_map [3, nRT, npix_y, npix_x] or None)``, default reals."""
        m = self.model
        rt = m.rt
        az = _a(rt["tab_RT_az"], np.float32)
        o = _RtOpts(int(lam), float(m.lam[lam - 1]), 1.0, 1.0, float(m.cfg.distance), float(ang_disque), 0, 100.0,
                    float(m.cfg.rin), float(m.cfg.rout), _p(az, C.c_float), 1)
        nRT = rt["RT_n_incl"] * rt["RT_n_az"]
        tm = np.zeros((nRT, npix_y, npix_x), np.float32)
        sm = np.zeros((3, nRT, npix_y, npix_x), np.float32) if surface else None
        rc = self.lib.oracle_tau_maps(C.byref(self.cm), C.byref(o), C.c_int(int(npix_x)), C.c_int(int(npix_y)),
                                      C.c_double(float(map_size)), C.c_double(float(zoom)), C.c_float(float(tau)),
                                      _p(tm, C.c_float), _p(sm, C.c_float) if surface else None)
        if rc:
            raise RuntimeError(f"oracle_tau_maps failed: {rc}")
        return tm, sm

    def init_dust_source_fct2(self, lam, ibin, I_spec, I_spec_star, Tdust, n_sent_photons, E_disk, nang_rt=15, nang_star=1000,
                              p_lambda=None):
        """Ray tracing method 2's source function of inclination ``ibin`` (1-based) from ``I_spec [n_cells, n_phi_I,
        n_theta_I, N_type_flux]`` / ``I_spec_star [n_cells]``: ``(eps_dust2 [n_cells, 2, nang_rt, N_type_flux],
        eps_dust2_star [n_cells, 2, nang_star, n_Stokes])`` (default real)."""
        m = self.model
        rt = m.rt
        az = _a(rt["tab_RT_az"], np.float32)
        o = _RtOpts(int(lam), float(m.lam[lam - 1]), float(m.E_stars[lam - 1] + E_disk), float(n_sent_photons),
                    float(m.cfg.distance), 0.0, 0, 100.0, float(m.cfg.rin), float(m.cfg.rout), _p(az, C.c_float), 1)
        I_spec = _a(I_spec, np.float64)
        nc, nphi, nth, ntf = I_spec.shape
        ns = 4 if (m.cfg.lsepar_pola and m.cfg.aniso_method == 1) else 1
        eps = np.zeros((nc, 2, nang_rt, ntf), np.float32)
        eps_star = np.zeros((nc, 2, nang_star, ns), np.float32)
        rc = self.lib.oracle_init_dust_source_fct2(
            C.byref(self.cm), C.byref(o), C.c_int(int(p_lambda or lam)), C.c_int(int(ibin)), C.c_int(nth), C.c_int(nphi),
            C.c_int(nang_rt), C.c_int(nang_star), _p(I_spec, C.c_double), _p(_a(I_spec_star, np.float64), C.c_double),
            _p(_a(Tdust, np.float32), C.c_float), _p(_a(m.grid["r_grid"], np.float64), C.c_double),
            _p(_a(np.abs(m.grid["z_grid"]), np.float64), C.c_double), _p(eps, C.c_float), _p(eps_star, C.c_float))
        if rc:
            raise RuntimeError(f"oracle_init_dust_source_fct2 failed: {rc}")
        return eps, eps_star

    def stars_map_image(self, lam, star_flux, npix_x, npix_y, map_size, zoom=1.0, seed=1, ang_disque=0.0,
                        limb_darkening=None):
        """compute_stars_map for images (resolved discs, limb darkening): ``(maps [nRT, n_maps, npix_y, npix_x],
        star_position [2, nRT, n_stars])``."""
        m = self.model
        rt = m.rt
        az = _a(rt["tab_RT_az"], np.float32)
        o = _RtOpts(int(lam), float(m.lam[lam - 1]), 1.0, 1.0, float(m.cfg.distance), float(ang_disque), 0, 100.0,
                    float(m.cfg.rin), float(m.cfg.rout), _p(az, C.c_float), 1)
        nRT, ns = rt["RT_n_incl"] * rt["RT_n_az"], int(np.asarray(star_flux).size)
        mu = ld = pld = None
        n_mu = 0
        if limb_darkening is not None:
            mu, ld = _a(limb_darkening[0], np.float32), _a(limb_darkening[1], np.float32)
            pld = _a(limb_darkening[2], np.float32) if len(limb_darkening) > 2 else None
            n_mu = mu.size
        n_maps = 3 if pld is not None else 1
        maps = np.zeros((nRT, n_maps, npix_y, npix_x), np.float64)
        pos = np.zeros((2, nRT, ns), np.float64)
        rc = self.lib.oracle_stars_map_image(
            C.byref(self.cm), C.byref(o), C.c_uint64(int(seed)), _p(_a(star_flux, np.float64), C.c_double), C.c_int(npix_x),
            C.c_int(npix_y), C.c_double(map_size), C.c_double(zoom), C.c_int(n_mu), _p(mu, C.c_float) if n_mu else None,
            _p(ld, C.c_float) if n_mu else None, _p(pld, C.c_float) if pld is not None else None, _p(maps, C.c_double),
            _p(pos, C.c_double))
        if rc:
            raise RuntimeError(f"oracle_stars_map_image failed: {rc}")
        return maps, pos

    def dust_map_image(self, lam, xI_scatt, Tdust, n_sent_photons, E_disk, npix_x, npix_y, map_size, zoom=1.0,
                       ang_disque=0.0, l_sym_ima=False, tau_dark_zone_obs=100.0, n_threads=1):
        """Ray-traced image of the dust at wavelength ``lam`` (dust_map method 2):
        (N_type_flux, RT_n_az, RT_n_incl, npix_y, npix_x) and the number of rays traced."""
        m = self.model
        rt = m.rt
        az = _a(rt["tab_RT_az"], np.float32)
        o = _RtOpts(int(lam), float(m.lam[lam - 1]), float(m.E_stars[lam - 1] + E_disk), float(n_sent_photons),
                    float(m.cfg.distance), float(ang_disque), int(l_sym_ima), float(tau_dark_zone_obs),
                    float(m.cfg.rin), float(m.cfg.rout), _p(az, C.c_float), int(n_threads))
        out = np.zeros((rt["N_type_flux"], rt["RT_n_az"], rt["RT_n_incl"], npix_y, npix_x), np.float64)
        x = _a(xI_scatt, np.float64)
        T = _a(Tdust, np.float32)
        n_rays = C.c_int(0)
        self.lib.oracle_dust_map_image.restype = C.c_int
        rc = self.lib.oracle_dust_map_image(C.byref(self.cm), C.byref(o), C.c_int(npix_x), C.c_int(npix_y),
                                            C.c_double(map_size), C.c_double(zoom), _p(x, C.c_double), _p(T, C.c_float),
                                            _p(out, C.c_double), C.byref(n_rays))
        if rc:
            raise RuntimeError(f"oracle_dust_map_image failed: {rc}")
        return out, n_rays.value

    def define_dark_zone(self, lam, tau_max):
        """optical_depth.f90:1425-1651 (2D): the flags the reference's thermal step computes with
        tau_max = tau_dark_zone_eq_th = 1500 at the first wavelength beyond wl_seuil = 0.81 um."""
        m = self.model
        g = m.grid
        out = np.zeros(m.n_cells, np.uint8)
        rl, rg, zg = _a(g["r_lim"], np.float64), _a(g["r_grid"], np.float64), _a(g["z_grid"], np.float64)
        self.lib.oracle_define_dark_zone.restype = C.c_int
        rc = self.lib.oracle_define_dark_zone(C.byref(self.cm), C.c_int(int(lam)), C.c_double(float(tau_max)),
                                              _p(rl, C.c_double), _p(rg, C.c_double), _p(zg, C.c_double),
                                              _p(out, C.c_ubyte))
        if rc:
            raise RuntimeError(f"oracle_define_dark_zone failed: {rc}")
        return out

    def dark_zone_extent(self, lam, tau_max):
        """(ri_in, ri_out, zj_sup[n_rad]) of define_dark_zone's steps 1-3 (optical_depth.f90:1459-1500, 1579-1586)."""
        m = self.model
        rl = _a(m.grid["r_lim"], np.float64)
        a, b = C.c_int(), C.c_int()
        zs = np.zeros(m.grid["n_rad"], np.int32)
        self.lib.oracle_dark_zone_extent.restype = C.c_int
        rc = self.lib.oracle_dark_zone_extent(C.byref(self.cm), C.c_int(int(lam)), C.c_double(float(tau_max)),
                                              _p(rl, C.c_double), C.byref(a), C.byref(b), _p(zs, C.c_int))
        if rc:
            raise RuntimeError(f"oracle_dark_zone_extent failed: {rc}")
        return a.value, b.value, zs

    def temp_approx_diffusion_vertical(self, Tdust, ri_in, ri_out, zj_sup):
        """Temp_approx_diffusion_vertical (diffusion.f90:292-374): returns (Tdust after the fill, iterations)."""
        m = self.model
        T = np.array(Tdust, np.float32)
        lam, dl = _a(m.lam, np.float64), _a(m.delta_lam, np.float64)
        zs = _a(zj_sup, np.int32)
        n_it = C.c_int()
        self.lib.oracle_temp_approx_diffusion_vertical.restype = C.c_int
        rc = self.lib.oracle_temp_approx_diffusion_vertical(C.byref(self.cm), _p(lam, C.c_double), _p(dl, C.c_double),
                                                            C.c_int(int(ri_in)), C.c_int(int(ri_out)), _p(zs, C.c_int),
                                                            _p(T, C.c_float), C.byref(n_it))
        if rc:
            raise RuntimeError(f"oracle_temp_approx_diffusion_vertical failed: {rc}")
        return T, n_it.value

    def init_reemission(self, kappa_abs_LTE=None, lam=None, delta_lam=None, dudt=None, heating_norm=None, ufac_implicit=0.0):
        """init_reemission (thermal_emission.f90:404-550): ``kappa_abs_LTE [classes, n_lambda]`` (default: the
        model's single class) -> ``(log_Qcool [classes, n_T], kdB_dT_CDF [classes, n_T, n_lambda])``; ``lam`` /
        ``delta_lam`` [micron] default to the model's wavelength grid.  ``dudt``, ``heating_norm`` [classes]:
        lextra_heating (:486-494), ``ufac_implicit`` > 0: ldudt_implicit."""
        m = self.model
        lam = m.lam if lam is None else lam
        delta_lam = m.delta_lam if delta_lam is None else delta_lam
        ka = np.atleast_2d(np.asarray(m.kappa_abs_LTE if kappa_abs_LTE is None else kappa_abs_LTE, np.float64))
        nc, nl = ka.shape
        nT = m.tab_Temp.size
        ka_ref = _a(ka.T, np.float64)  # (p_n_cells, n_lambda) column-major = [lambda][class] in memory
        lq = np.zeros((nc, nT), np.float64)
        cdf = np.zeros((nc, nT, nl), np.float64)
        du = None if dudt is None else _a(np.broadcast_to(np.asarray(dudt, np.float64), (nc,)), np.float64)
        hn = None if dudt is None else _a(np.broadcast_to(np.asarray(heating_norm, np.float64), (nc,)), np.float64)
        self.lib.oracle_init_reemission_ex.restype = C.c_int
        rc = self.lib.oracle_init_reemission_ex(
            C.c_int(nc), C.c_int(nT), C.c_int(nl), _p(_a(m.tab_Temp, np.float32), C.c_float),
            _p(_a(lam, np.float64), C.c_double), _p(_a(delta_lam, np.float64), C.c_double),
            _p(ka_ref, C.c_double), _p(du, C.c_double) if du is not None else None,
            _p(hn, C.c_double) if hn is not None else None, C.c_double(float(ufac_implicit)),
            _p(lq, C.c_double), _p(cdf, C.c_double))
        if rc:
            raise RuntimeError(f"oracle_init_reemission failed: {rc}")
        return lq, cdf

    def opacity(self, grains, dens, aniso_method=None, lsepar_pola=None):
        """opacity + calc_local_scattering_matrices (dust_prop.f90:791-1243) for every wavelength: ``grains`` as
        ``mcfost_amd.host.model.synthetic_grains`` returns them, ``dens [p_n_cells, n_grains]``.  Returns the tables in
        the reference's layouts (C-ordered: ``kappa[n_lambda, p_n_cells]``, ``tab_s11_pos[n_lambda, p_n_cells, nang+1]``)."""
        m = self.model
        am = int(m.cfg.aniso_method if aniso_method is None else aniso_method)
        pola = int(bool(m.cfg.lsepar_pola) if lsepar_pola is None else bool(lsepar_pola)) if am == 1 else 0
        ng, nl = int(grains["n_grains"]), m.n_lambda
        dens = _a(dens, np.float64)
        nc = dens.shape[0]
        nang = int(np.asarray(grains["tab_s11"]).shape[-1]) - 1
        f32, f64 = np.float32, np.float64
        out = dict(kappa=np.zeros((nl, nc), f64), kappa_abs_LTE=np.zeros((nl, nc), f64), tab_albedo_pos=np.zeros((nl, nc), f32),
                   tab_g_pos=np.zeros((nl, nc), f32), tab_s11_pos=np.zeros((nl, nc, nang + 1), f32),
                   prob_s11_pos=np.zeros((nl, nc, nang + 1), f32))
        for k in ("tab_s12_o_s11_pos", "tab_s22_o_s11_pos", "tab_s33_o_s11_pos", "tab_s34_o_s11_pos", "tab_s44_o_s11_pos"):
            out[k] = np.zeros((nl, nc, nang + 1), f32) if pola else None
        pf = lambda k: _p(_a(grains[k], f32), C.c_float)
        po = lambda k: _p(out[k], C.c_float) if out[k] is not None else None
        self.lib.oracle_opacity.restype = C.c_int
        rc = self.lib.oracle_opacity(
            C.c_int(ng), C.c_int(nl), C.c_int(nc), C.c_int(nang), C.c_int(am), C.c_int(pola),
            C.c_int(int(grains["grain_RE_LTE_start"])), C.c_int(int(grains["grain_RE_LTE_end"])),
            pf("C_ext"), pf("C_sca"), pf("C_abs"), pf("tab_g"), pf("tab_s11"), pf("tab_s12"), pf("tab_s22"), pf("tab_s33"),
            pf("tab_s34"), pf("tab_s44"), pf("S_grain"), _p(_a(grains["n_grains_k"], f64), C.c_double), _p(dens, C.c_double),
            _p(out["kappa"], C.c_double), _p(out["kappa_abs_LTE"], C.c_double), po("tab_albedo_pos"), po("tab_g_pos"),
            po("tab_s11_pos"), po("prob_s11_pos"), po("tab_s12_o_s11_pos"), po("tab_s22_o_s11_pos"), po("tab_s33_o_s11_pos"),
            po("tab_s34_o_s11_pos"), po("tab_s44_o_s11_pos"))
        if rc:
            raise RuntimeError(f"oracle_opacity failed: {rc}")
        return out

    def repartition_energie(self, lam, Tdust, E_ISM=0.0, weight=None):
        """repartition_energie(lam) (thermal_emission.f90:1771-1949, LTE): frac_E_stars, frac_E_disk, E_disk and
        prob_E_cell(0:n_cells) of the 1-based wavelength ``lam`` for the dust temperature ``Tdust``."""
        m = self.model
        f = self.lib.oracle_repartition_energie
        f.restype = C.c_int
        T = np.ascontiguousarray(Tdust, np.float32)
        wgt = np.ascontiguousarray(weight, np.float32) if weight is not None else None
        pe = np.zeros(m.n_cells + 1)
        fs, fd, ed = C.c_double(), C.c_double(), C.c_double()
        rc = f(C.byref(self.cm), C.c_int(int(lam)), C.c_double(float(m.lam[lam - 1])), C.c_double(float(m.E_stars[lam - 1])),
               C.c_double(float(E_ISM)), _p(T, C.c_float), _p(wgt, C.c_float) if wgt is not None else None,
               C.byref(fs), C.byref(fd), C.byref(ed), _p(pe, C.c_double))
        if rc:
            raise RuntimeError(f"oracle_repartition_energie: no energy at wavelength {lam}")
        return dict(frac_E_stars=fs.value, frac_E_disk=fd.value, E_disk=ed.value, prob_E_cell=pe)

    # -- Voronoi operators ---------------------------------------------------
    def cross_voronoi(self, x0, y0, z0, u, v, w, cell, prev):
        n = len(cell)
        out = {k: np.zeros(n) for k in ("x1", "y1", "z1", "l", "l_contrib", "l_void_before")}
        nxt = np.zeros(n, np.int32)
        f = self.lib.oracle_cross_voronoi_cell
        f.restype = None
        f.argtypes = [C.c_void_p] + [C.c_double] * 6 + [C.c_int, C.c_int] + [_dp] * 3 + [_ip] + [_dp] * 3
        d = [C.c_double() for _ in range(6)]
        nc = C.c_int()
        mp = C.addressof(self.cm)
        for i in range(n):
            f(mp, x0[i], y0[i], z0[i], u[i], v[i], w[i], int(cell[i]), int(prev[i]), C.byref(d[0]),
              C.byref(d[1]), C.byref(d[2]), C.byref(nc), C.byref(d[3]), C.byref(d[4]), C.byref(d[5]))
            for k, q in zip(("x1", "y1", "z1", "l", "l_contrib", "l_void_before"), d):
                out[k][i] = q.value
            nxt[i] = nc.value
        out["next_cell"] = nxt
        return out

    def index_cell_voronoi(self, x, y, z):
        f = self.lib.oracle_index_cell_voronoi
        f.restype = None
        f.argtypes = [C.c_void_p] + [C.c_double] * 3 + [_ip]
        out = np.zeros(len(x), np.int32)
        ic = C.c_int()
        mp = C.addressof(self.cm)
        for i in range(len(x)):
            f(mp, x[i], y[i], z[i], C.byref(ic))
            out[i] = ic.value
        return out

    def move_to_grid_voronoi(self, x, y, z, u, v, w):
        f = self.lib.oracle_move_to_grid_voronoi
        f.restype = None
        f.argtypes = [C.c_void_p] + [_dp] * 3 + [C.c_double] * 3 + [_ip, _ip]
        n = len(x)
        xo, yo, zo = np.array(x, float), np.array(y, float), np.array(z, float)
        ic = np.zeros(n, np.int32); ok = np.zeros(n, np.int32)
        mp = C.addressof(self.cm)
        for i in range(n):
            a, b, c_ = C.c_double(xo[i]), C.c_double(yo[i]), C.c_double(zo[i])
            i1, i2 = C.c_int(), C.c_int()
            f(mp, C.byref(a), C.byref(b), C.byref(c_), u[i], v[i], w[i], C.byref(i1), C.byref(i2))
            xo[i], yo[i], zo[i], ic[i], ok[i] = a.value, b.value, c_.value, i1.value, i2.value
        return xo, yo, zo, ic, ok

    # -- unit operators (batched in Python; small n only) -------------------
    def cross_cell(self, x0, y0, z0, u, v, w, cell):
        n = len(cell)
        x1 = np.zeros(n); y1 = np.zeros(n); z1 = np.zeros(n); l = np.zeros(n)
        nxt = np.zeros(n, np.int32)
        f = self.lib.oracle_cross_spherical_cell if self.cm.grid_type == 2 else self.lib.oracle_cross_cylindrical_cell
        f.argtypes = [C.c_void_p] + [C.c_double] * 6 + [C.c_int, C.c_int] + [_dp] * 3 + [_ip] + [_dp] * 3
        a, b, c_, d = C.c_double(), C.c_double(), C.c_double(), C.c_double()
        lc, lv = C.c_double(), C.c_double()
        nc = C.c_int()
        mp = C.addressof(self.cm)
        for i in range(n):
            f(mp, x0[i], y0[i], z0[i], u[i], v[i], w[i], int(cell[i]), 0, C.byref(a), C.byref(b),
              C.byref(c_), C.byref(nc), C.byref(d), C.byref(lc), C.byref(lv))
            x1[i], y1[i], z1[i], l[i], nxt[i] = a.value, b.value, c_.value, d.value, nc.value
        return x1, y1, z1, nxt, l

    def index_cell(self, x, y, z):
        n = len(x)
        out = np.zeros(n, np.int32)
        f = self.lib.oracle_index_cell_sph if self.cm.grid_type == 2 else self.lib.oracle_index_cell_cyl
        f.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double, _ip]
        ic = C.c_int()
        mp = C.addressof(self.cm)
        for i in range(n):
            f(mp, x[i], y[i], z[i], C.byref(ic))
            out[i] = ic.value
        return out

    def test_exit_grid(self, icell, x, y, z):
        mp = C.addressof(self.cm)
        if self.cm.grid_type == 2:
            g = self.lib.oracle_test_exit_grid_sph
            g.argtypes = [C.c_void_p, C.c_int]
            g.restype = C.c_int
            return np.array([g(mp, int(icell[i])) for i in range(len(x))], np.int32)
        f = self.lib.oracle_test_exit_grid_cyl
        f.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_double]
        f.restype = C.c_int
        return np.array([f(mp, int(icell[i]), x[i], y[i], z[i]) for i in range(len(x))], np.int32)

    def move_to_grid(self, x, y, z, u, v, w):
        n = len(x)
        xo, yo, zo = np.array(x, float), np.array(y, float), np.array(z, float)
        ic = np.zeros(n, np.int32); li = np.zeros(n, np.int32)
        f = self.lib.oracle_move_to_grid_sph if self.cm.grid_type == 2 else self.lib.oracle_move_to_grid_cyl
        f.argtypes = [C.c_void_p, _dp, _dp, _dp, C.c_double, C.c_double, C.c_double, _ip, _ip]
        mp = C.addressof(self.cm)
        for i in range(n):
            a, b, c_ = C.c_double(xo[i]), C.c_double(yo[i]), C.c_double(zo[i])
            cc, ll = C.c_int(0), C.c_int(0)
            f(mp, C.byref(a), C.byref(b), C.byref(c_), u[i], v[i], w[i], C.byref(cc), C.byref(ll))
            xo[i], yo[i], zo[i], ic[i], li[i] = a.value, b.value, c_.value, cc.value, ll.value
        return xo, yo, zo, ic, li

    def pos_em_cell(self, icell, r1, r2, r3):
        n = len(icell)
        x = np.zeros(n); y = np.zeros(n); z = np.zeros(n)
        f = self.lib.oracle_pos_em_cell_sph if self.cm.grid_type == 2 else self.lib.oracle_pos_em_cell_cyl
        f.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float, _dp, _dp, _dp]
        mp = C.addressof(self.cm)
        a, b, c_ = C.c_double(), C.c_double(), C.c_double()
        for i in range(n):
            f(mp, int(icell[i]), float(r1[i]), float(r2[i]), float(r3[i]), C.byref(a), C.byref(b),
              C.byref(c_))
            x[i], y[i], z[i] = a.value, b.value, c_.value
        return x, y, z

    def cdapres(self, cospsi, phi, u0, v0, w0):
        f = self.lib.oracle_cdapres
        f.argtypes = [C.c_double] * 5 + [_dp] * 3
        a, b, c_ = C.c_double(), C.c_double(), C.c_double()
        f(cospsi, phi, u0, v0, w0, C.byref(a), C.byref(b), C.byref(c_))
        return a.value, b.value, c_.value

    def hg(self, g, rand):
        f = self.lib.oracle_hg
        f.argtypes = [C.c_float, C.c_float, C.c_int, _ip, _dp]
        it, cp = C.c_int(), C.c_double()
        f(g, rand, 180, C.byref(it), C.byref(cp))
        return it.value, cp.value

    def angle_diff_theta_pos(self, p_lambda, rand, rand2):
        f = self.lib.oracle_angle_diff_theta_pos
        f.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float, _ip, _dp]
        it, cp = C.c_int(), C.c_double()
        f(C.addressof(self.cm), p_lambda, rand, rand2, C.byref(it), C.byref(cp))
        return it.value, cp.value

    def select_wl_em(self, rand):
        f = self.lib.oracle_select_wl_em
        f.argtypes = [C.c_void_p, C.c_float, _ip]
        lam = C.c_int()
        f(C.addressof(self.cm), rand, C.byref(lam))
        return lam.value

    def update_stokes(self, S, d0, d1, M):
        f = self.lib.oracle_update_stokes
        f.argtypes = [_dp] + [C.c_double] * 6 + [_dp]
        Sa = _a(S, np.float64).copy()
        Ma = _a(np.asarray(M).T, np.float64).copy()  # column-major M(i,j)
        f(_p(Sa, C.c_double), *d0, *d1, _p(Ma, C.c_double))
        return Sa

    def philox(self, ctr, key):
        c = (C.c_uint32 * 4)(*ctr)
        k = (C.c_uint32 * 2)(*key)
        o = (C.c_uint32 * 4)()
        self.lib.oracle_philox4x32_10(c, k, o)
        return [int(v) for v in o]

    def packet_rand(self, seed, packet, n):
        return float(self.lib.oracle_packet_rand(seed, packet, n))


class RefGeom:
    """The reference's own geometry routines (``oracle/_ref``).  One grid per
    process (the reference allocates its module arrays once)."""

    def __init__(self):
        path = build_ref()
        if path is None or not os.path.exists(path):
            raise FileNotFoundError("oracle/_ref/libmcfost_ref_geom.so not built")
        self.lib = C.CDLL(path)
        self.ready = False

    def setup_grid(self, cfg):
        ierr = C.c_int()
        self.sph = int(getattr(cfg, "grid_type", 1)) == 2
        setup = self.lib.ref_setup_grid_sph if self.sph else self.lib.ref_setup_grid
        self._sfx = "_sph" if self.sph else ""
        setup.argtypes = [C.c_int] * 5 + [C.c_double] * 7 + [_ip]
        setup(cfg.n_rad, cfg.nz, cfg.n_az, int(cfg.l3D), cfg.n_rad_in, cfg.rin,
                                cfg.edge, cfg.rout, cfg.rref, cfg.sclht, cfg.exp_beta, cfg.surf,
                                C.byref(ierr))
        if ierr.value:
            raise RuntimeError("reference grid already set up in this process")
        self.cfg = cfg
        self.ready = True
        a, b, c_, d = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        self.lib.ref_grid_sizes(C.byref(a), C.byref(b), C.byref(c_), C.byref(d))
        self.n_cells, self.ntot2, self.jlo, self.jn = a.value, b.value, c_.value, d.value

    def get_grid(self):
        cfg = self.cfg
        n_rad, nz, n_az = cfg.n_rad, cfg.nz, cfg.n_az
        o = dict(
            r_lim=np.zeros(n_rad + 1), r_lim_2=np.zeros(n_rad + 1), zmax=np.zeros(n_rad),
            z_lim=np.zeros(n_rad * (nz + 2)), tan_phi_lim=np.zeros(n_az),
            volume=np.zeros(self.n_cells), r_grid=np.zeros(self.n_cells),
            z_grid=np.zeros(self.n_cells),
            cell_map=np.zeros((n_rad + 2) * self.jn * n_az, np.int32),
            cell_map_i=np.zeros(self.ntot2, np.int32), cell_map_j=np.zeros(self.ntot2, np.int32),
            cell_map_k=np.zeros(self.ntot2, np.int32), lexit_cell=np.zeros(self.ntot2, np.int32))
        rm = C.c_double()
        self.lib.ref_get_grid(*[_p(o[k], C.c_double) for k in
                                ("r_lim", "r_lim_2", "zmax", "z_lim", "tan_phi_lim", "volume",
                                 "r_grid", "z_grid")],
                              *[_p(o[k], C.c_int) for k in
                                ("cell_map", "cell_map_i", "cell_map_j", "cell_map_k", "lexit_cell")],
                              C.byref(rm))
        o["Rmax2"] = rm.value
        if self.sph:
            for k, n in (("tan_theta_lim", nz + 1), ("theta_lim", nz + 1), ("w_lim", nz + 1), ("r_lim_3", n_rad + 1)):
                o[k] = np.zeros(n)
            self.lib.ref_get_grid_sph(*[_p(o[k], C.c_double) for k in ("tan_theta_lim", "theta_lim", "w_lim", "r_lim_3")])
        return o

    def cross_cell(self, x0, y0, z0, u, v, w, cell):
        n = len(cell)
        arrs = [_a(q, np.float64) for q in (x0, y0, z0, u, v, w)]
        cell = _a(cell, np.int32)
        x1 = np.zeros(n); y1 = np.zeros(n); z1 = np.zeros(n); l = np.zeros(n)
        nxt = np.zeros(n, np.int32)
        getattr(self.lib, "ref_cross_cell" + self._sfx)(C.c_int(n), *[_p(q, C.c_double) for q in arrs], _p(cell, C.c_int),
                                _p(x1, C.c_double), _p(y1, C.c_double), _p(z1, C.c_double),
                                _p(nxt, C.c_int), _p(l, C.c_double))
        return x1, y1, z1, nxt, l

    def index_cell(self, x, y, z):
        n = len(x)
        arrs = [_a(q, np.float64) for q in (x, y, z)]
        ic = np.zeros(n, np.int32)
        getattr(self.lib, "ref_index_cell" + self._sfx)(C.c_int(n), *[_p(q, C.c_double) for q in arrs], _p(ic, C.c_int))
        return ic

    def test_exit_grid(self, icell, x, y, z):
        n = len(x)
        ic = _a(icell, np.int32)
        arrs = [_a(q, np.float64) for q in (x, y, z)]
        out = np.zeros(n, np.int32)
        getattr(self.lib, "ref_test_exit_grid" + self._sfx)(C.c_int(n), _p(ic, C.c_int), *[_p(q, C.c_double) for q in arrs],
                                    _p(out, C.c_int))
        return out

    def move_to_grid(self, x, y, z, u, v, w):
        n = len(x)
        xo, yo, zo = (_a(q, np.float64).copy() for q in (x, y, z))
        d = [_a(q, np.float64) for q in (u, v, w)]
        ic = np.zeros(n, np.int32); li = np.zeros(n, np.int32)
        getattr(self.lib, "ref_move_to_grid" + self._sfx)(C.c_int(n), _p(xo, C.c_double), _p(yo, C.c_double),
                                  _p(zo, C.c_double), *[_p(q, C.c_double) for q in d],
                                  _p(ic, C.c_int), _p(li, C.c_int))
        return xo, yo, zo, ic, li

    def pos_em_cell(self, icell, r1, r2, r3):
        n = len(icell)
        ic = _a(icell, np.int32)
        r = [_a(q, np.float32) for q in (r1, r2, r3)]
        x = np.zeros(n); y = np.zeros(n); z = np.zeros(n)
        getattr(self.lib, "ref_pos_em_cell" + self._sfx)(C.c_int(n), _p(ic, C.c_int), *[_p(q, C.c_float) for q in r],
                                 _p(x, C.c_double), _p(y, C.c_double), _p(z, C.c_double))
        return x, y, z

    def kdtree_nearest(self, sites, queries):
        """1-based index of the site nearest to every query, by the reference's kdtree2 (find_Voronoi_cell)."""
        s, q = _a(sites, np.float64), _a(queries, np.float64)
        idx = np.zeros(q.shape[0], np.int32)
        self.lib.ref_kdtree_nearest(C.c_int(s.shape[0]), _p(s, C.c_double), C.c_int(q.shape[0]), _p(q, C.c_double),
                                    _p(idx, C.c_int))
        return idx

    def distance_to_closest_wall(self, icell, x, y, z):
        n = len(icell)
        ic = _a(icell, np.int32)
        d = np.zeros(n)
        self.lib.ref_distance_to_closest_wall(C.c_int(n), _p(ic, C.c_int), _p(_a(x, np.float64), C.c_double),
                                              _p(_a(y, np.float64), C.c_double), _p(_a(z, np.float64), C.c_double),
                                              _p(d, C.c_double))
        return d

    def init_tab_temp(self, n_T, T_min, T_max):
        out = np.zeros(n_T, np.float32)
        self.lib.ref_init_tab_temp.argtypes = [C.c_int, C.c_float, C.c_float, _fp]
        self.lib.ref_init_tab_temp(n_T, T_min, T_max, _p(out, C.c_float))
        return out

    def init_lambda(self, n, lmin, lmax):
        o = [np.zeros(n) for _ in range(4)]
        self.lib.ref_init_lambda.argtypes = [C.c_int, C.c_float, C.c_float] + [_dp] * 4
        self.lib.ref_init_lambda(n, lmin, lmax, *[_p(q, C.c_double) for q in o])
        return o

    def constants(self):
        o = np.zeros(16)
        self.lib.ref_constants(_p(o, C.c_double))
        return o
