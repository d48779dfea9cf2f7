"""Host-side binding of the MI355X packet engine (``libmcfost_hip.so``).

This is the Python mirror of the Fortran shim (``mcfost_amd/fortran/mcgpu_f.f90``):
it hands the model tables to the C-ABI of ``include/mcgpu.h`` and calls the
replacement of ``mc_photon_loop`` (``dust_transfer.f90:439-572``) and
``Temp_finale`` (``thermal_emission.f90:870-906``).  There is no CPU path
here: if the HIP library is missing or no device is usable, construction
fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MCGPU_LIB") or os.path.join(_HERE, "csrc", "libmcfost_hip.so")

N_SED_TYPES = 9
N_COUNTERS = 10
COUNTER_NAMES = ("packets", "crossings", "flights", "scatterings", "absorptions",
                 "escaped", "killed_star", "dark_mirrors", "mrw_walks", "mrw_steps")
SED_NAMES = ("sed", "sed_q", "sed_u", "sed_v", "n_phot_sed", "sed_star", "sed_star_scat",
             "sed_disk", "sed_disk_scat")

# every symbol include/mcgpu.h declares
ABI_SYMBOLS = (
    "mcgpu_create", "mcgpu_destroy", "mcgpu_last_error", "mcgpu_set_grid_cyl", "mcgpu_set_stars",
    "mcgpu_set_opacity", "mcgpu_set_scattering", "mcgpu_set_thermal", "mcgpu_set_sed_bins",
    "mcgpu_set_E_prior", "mcgpu_run_thermal", "mcgpu_launch_thermal", "mcgpu_sync",
    "mcgpu_device_accumulators", "mcgpu_fetch", "mcgpu_set_stream", "mcgpu_temp_finale",
    "mcgpu_probe_cross_cell", "mcgpu_probe_index_cell", "mcgpu_probe_philox",
    "mcgpu_probe_packet_rand", "mcgpu_set_midplane_snap", "mcgpu_set_grid_voronoi",
    "mcgpu_probe_cross_voronoi", "mcgpu_set_rt1", "mcgpu_run_mono", "mcgpu_fetch_xI",
    "mcgpu_device_xI", "mcgpu_set_ism", "mcgpu_rt1_dust_map", "mcgpu_set_xI", "mcgpu_rt1_image", "mcgpu_set_xI_precision",
    "mcgpu_get_xI_precision", "mcgpu_set_option", "mcgpu_get_info", "mcgpu_counters_to_accum", "mcgpu_counters_from_accum",
    "mcgpu_set_grid_sph", "mcgpu_temp_approx_diffusion_vertical", "mcgpu_set_mrw", "mcgpu_fetch_radiation_field", "mcgpu_set_variable_dust", "mcgpu_rt1_stars_map_sed", "mcgpu_define_dark_zone", "mcgpu_init_reemission", "mcgpu_multi_create", "mcgpu_multi_destroy", "mcgpu_multi_size", "mcgpu_multi_ctx", "mcgpu_multi_last_error",
    "mcgpu_repartition_energie", "mcgpu_opacity", "mcgpu_set_variable_dust_s11", "mcgpu_set_scattering_method1", "mcgpu_set_rt2", "mcgpu_fetch_I_spec", "mcgpu_rt1_stars_map_image", "mcgpu_set_I_spec", "mcgpu_rt2_source", "mcgpu_rt2_dust_map", "mcgpu_rt2_image", "mcgpu_shard_packets", "mcgpu_multi_run_thermal", "mcgpu_multi_run_mono", "mcgpu_multi_run_sed", "mcgpu_multi_rccl_ranks", "mcgpu_multi_create_ex", "mcgpu_multi_reductions", "mcgpu_set_mrw_exit_spectrum", "mcgpu_voronoi_tesselation", "mcgpu_build_ksca_CDF", "mcgpu_init_reemission_ex", "mcgpu_tau_maps",
)


class McgpuError(RuntimeError):
    pass


class RunOpts(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("first_packet", C.c_uint64), ("n_packets", C.c_uint64),
                ("n_replicas", C.c_double), ("frozen", C.c_int), ("accumulate", C.c_int),
                ("grid_blocks", C.c_int), ("block_threads", C.c_int)]


class MonoOpts(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("lambda_", C.c_int), ("p_lambda", C.c_int), ("n_chunks", C.c_int),
                ("first_chunk", C.c_int), ("n_photons2", C.c_uint64), ("n_phot_lim", C.c_double),
                ("capt_sup", C.c_int), ("rt1", C.c_int), ("accumulate", C.c_int), ("grid_blocks", C.c_int), ("block_threads", C.c_int)]


class GrainTables(C.Structure):       # mcgpu_grain_tables
    _fields_ = [("n_grains", C.c_int), ("grain_RE_LTE_start", C.c_int), ("grain_RE_LTE_end", C.c_int)] + \
               [(k, C.POINTER(C.c_float)) for k in ("C_ext", "C_sca", "C_abs", "tab_g", "tab_s11", "tab_s12", "tab_s22",
                                                    "tab_s33", "tab_s34", "tab_s44", "S_grain")] + \
               [("n_grains_k", C.POINTER(C.c_double))]


class OpacityTables(C.Structure):     # mcgpu_opacity_tables
    _fields_ = [("kappa", C.POINTER(C.c_double)), ("kappa_abs_LTE", C.POINTER(C.c_double))] + \
               [(k, C.POINTER(C.c_float)) for k in ("tab_albedo_pos", "tab_g_pos", "tab_s11_pos", "prob_s11_pos",
                                                    "tab_s12_o_s11_pos", "tab_s22_o_s11_pos", "tab_s33_o_s11_pos",
                                                    "tab_s34_o_s11_pos", "tab_s44_o_s11_pos")]


class RtOpts(C.Structure):
    _fields_ = [("lambda_", C.c_int), ("wl_um", C.c_double), ("E_src", C.c_double), ("n_sent_photons", C.c_double),
                ("distance", C.c_double), ("ang_disque", C.c_double), ("l_sym_ima", C.c_int),
                ("tau_dark_zone_obs", C.c_double), ("Rmin", C.c_double), ("Rmax", C.c_double)]


class SedWavelength(C.Structure):
    _fields_ = [("lambda_", C.c_int), ("p_lambda", C.c_int), ("wl_um", C.c_double), ("E_star", C.c_double),
                ("E_ISM", C.c_double), ("seed", C.c_uint64), ("cost", C.c_double)]


_lib = None


def xi32_layout(nRT, pola, contrib):
    """The packed default-real device layout of xI_scatt the library chooses for ``nRT`` observers
    (``mcfost_amd/csrc/mc_xi32.hip.h::xi32_layout``, mirrored for the tests and the bench's accounting): default reals per
    sub-bin, the arrangement, and the 64-byte lines one crossing's deposits touch."""
    lines = lambda n: (n + 15) // 16
    nS = 4 if pola else 1
    if not contrib:
        return dict(binf=16 * lines(nRT * nS), split=False, lines_touched=lines(nRT * nS), values_per_deposit=nS)
    nA = nS - 1      # I is not stored: it is the sum of the two origins
    l_inter, l_star, l_th = lines(nRT * (nA + 2)), lines(nRT * (nA + 1)), lines(nRT * nA) + lines(nRT)
    if l_star + l_th < 2 * l_inter:
        return dict(binf=16 * (l_star + lines(nRT)), split=True, lines_touched=max(l_star, l_th), values_per_deposit=nA + 1)
    return dict(binf=16 * l_inter, split=False, lines_touched=l_inter, values_per_deposit=nA + 1)


def load_library(path: str = LIB_PATH):
    """Load the C-ABI library; fail loudly if it was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(path):
        raise McgpuError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    lib = C.CDLL(path)
    lib.mcgpu_last_error.restype = C.c_char_p
    lib.mcgpu_last_error.argtypes = [C.c_void_p]
    for s in ABI_SYMBOLS:
        getattr(lib, s)  # AttributeError if the symbol is not exported
    _lib = lib
    return lib


def _a(x, dt):
    return np.ascontiguousarray(x, dtype=dt)


def _p(arr, ct):
    return arr.ctypes.data_as(C.POINTER(ct))


class _DevArray:
    """Exposes a raw device pointer through ``__cuda_array_interface__`` so
    that torch can wrap the accumulator without a copy."""

    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": typestr,
                                         "data": (int(ptr), False), "version": 2}


class Engine:
    """One device context holding one model (the tables ``init_dust_transfer``
    prepares, ``dust_transfer.f90:41-340``)."""

    def __init__(self, model, n_packets_total, device: int = 0, _borrowed_ctx=None):
        self.lib = load_library()
        self.model = model
        self.ctx = C.c_void_p()
        self._owns_ctx = _borrowed_ctx is None
        if _borrowed_ctx is not None:   # a context of a MultiEngine: it creates and destroys it
            self.ctx = C.c_void_p(_borrowed_ctx)
            self.device = device
            self._upload(model, float(n_packets_total))
            return
        # If this process also uses torch (device_accumulators / device_xI hand the engine's buffers to it), torch
        # must initialise its HIP context first: torch wheels carry their own HIP runtime, and brought up second it
        # has been seen to report "No HIP GPUs are available".
        import sys
        t = sys.modules.get("torch")
        if t is not None:
            try:
                if t.cuda.is_available():
                    t.cuda.init()
            except Exception:
                pass
        rc = self.lib.mcgpu_create(C.c_int(device), C.byref(self.ctx))
        if rc:
            self.ctx = C.c_void_p()
            raise McgpuError(f"mcgpu_create(device={device}) failed with code {rc} "
                             "(1 = no usable HIP device; the engine has no CPU path)")
        self.device = device
        self._upload(model, float(n_packets_total))

    # -- plumbing ----------------------------------------------------------
    def _chk(self, rc, what):
        if rc:
            msg = self.lib.mcgpu_last_error(self.ctx)
            raise McgpuError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")

    def close(self):
        if getattr(self, "ctx", None) and self.ctx.value:
            if getattr(self, "_owns_ctx", True):
                self.lib.mcgpu_destroy(self.ctx)
            self.ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _upload(self, m, n_tot):
        g, cfg, L = m.grid, m.cfg, self.lib
        d, i32 = np.float64, np.int32
        if g.get("grid_type", 1) == 3:
            u8 = np.uint8
            self._chk(L.mcgpu_set_grid_voronoi(
                self.ctx, C.c_int(g["n_cells"]), _p(_a(g["v_xyz"], np.float32), C.c_float),
                _p(_a(g["v_xyz_dp"], d), C.c_double), _p(_a(g["v_h"], d), C.c_double),
                _p(_a(g["v_first"], i32), C.c_int), _p(_a(g["v_last"], i32), C.c_int),
                _p(_a(g["v_neigh"], i32), C.c_int), C.c_longlong(int(np.asarray(g["v_neigh"]).size)),
                _p(_a(g["v_was_cut"], u8), C.c_ubyte), _p(_a(g["v_is_star_neighbour"], u8), C.c_ubyte),
                _p(_a(g["v_walls"], np.float32), C.c_float), C.c_double(g["v_cut_o_h"]),
                _p(_a(g["v_wall_first"], i32), C.c_int), _p(_a(g["v_wall_cells"], i32), C.c_int),
                _p(_a(g["volume"], d), C.c_double)), "mcgpu_set_grid_voronoi")
        else:
            self._upload_grid_cyl(m)
        self._upload_tables(m, n_tot)

    def _upload_grid_cyl(self, m):
        g, L = m.grid, self.lib
        d, i32 = np.float64, np.int32
        if g.get("grid_type", 1) == 2:
            self._chk(L.mcgpu_set_grid_sph(
                self.ctx, C.c_int(g["n_rad"]), C.c_int(g["nz"]), C.c_int(g["n_az"]), C.c_int(g["l3D"]),
                _p(_a(g["r_lim_2"], d), C.c_double), _p(_a(g["r_lim_3"], d), C.c_double),
                _p(_a(g["tan_theta_lim"], d), C.c_double), _p(_a(g["theta_lim"], d), C.c_double),
                _p(_a(g["tan_phi_lim"], d), C.c_double), C.c_double(g["Rmax2"]), _p(_a(g["volume"], d), C.c_double),
                _p(_a(g["cell_map"], i32), C.c_int), _p(_a(g["cell_map_i"], i32), C.c_int),
                _p(_a(g["cell_map_j"], i32), C.c_int), _p(_a(g["cell_map_k"], i32), C.c_int),
                _p(_a(g["lexit_cell"], i32), C.c_int)), "mcgpu_set_grid_sph")
            return
        self._chk(L.mcgpu_set_grid_cyl(
            self.ctx, C.c_int(g["n_rad"]), C.c_int(g["nz"]), C.c_int(g["n_az"]), C.c_int(g["l3D"]),
            _p(_a(g["r_lim_2"], d), C.c_double), _p(_a(g["zmax"], d), C.c_double),
            _p(_a(g["z_lim"], d), C.c_double), _p(_a(g["tan_phi_lim"], d), C.c_double),
            C.c_double(g["zmaxmax"]), C.c_double(g["Rmax2"]), _p(_a(g["volume"], d), C.c_double),
            _p(_a(g["cell_map"], i32), C.c_int), _p(_a(g["cell_map_i"], i32), C.c_int),
            _p(_a(g["cell_map_j"], i32), C.c_int), _p(_a(g["cell_map_k"], i32), C.c_int),
            _p(_a(g["lexit_cell"], i32), C.c_int)), "mcgpu_set_grid_cyl")
        self._chk(L.mcgpu_set_midplane_snap(self.ctx, C.c_int(int(getattr(m, "midplane_snap", 1)))),
                  "mcgpu_set_midplane_snap")

    def _upload_tables(self, m, n_tot):
        cfg, L = m.cfg, self.lib
        d, i32 = np.float64, np.int32
        ism = getattr(m, "ism", None)
        if ism is not None:
            self._chk(L.mcgpu_set_ism(self.ctx, C.c_double(float(ism["R_ISM"])),
                                      _p(_a(ism["centre_ISM"], d), C.c_double)), "mcgpu_set_ism")
        st = np.asarray(m.stars, d)
        cols = [_a(st[:, q], d) for q in range(4)]
        self._chk(L.mcgpu_set_stars(
            self.ctx, C.c_int(st.shape[0]), *[_p(c, C.c_double) for c in cols],
            _p(_a(st[:, 4], i32), C.c_int), _p(_a(st[:, 5], i32), C.c_int)), "mcgpu_set_stars")
        dark = None if m.l_dark_zone is None else _p(_a(m.l_dark_zone, np.uint8), C.c_ubyte)
        self._chk(L.mcgpu_set_opacity(
            self.ctx, C.c_int(m.n_lambda), _p(_a(m.kappa, d), C.c_double),
            _p(_a(m.kappa_abs_LTE, d), C.c_double), _p(_a(m.albedo, np.float32), C.c_float),
            _p(_a(m.kappa_factor, d), C.c_double), dark), "mcgpu_set_opacity")
        f = np.float32
        self._chk(L.mcgpu_set_scattering(
            self.ctx, C.c_int(180), C.c_int(cfg.aniso_method), C.c_int(int(cfg.lisotropic)),
            C.c_int(int(cfg.lsepar_pola)), C.c_int(int(m.p_lambda_fixed)),
            _p(_a(m.prob_s11_pos, f), C.c_float), _p(_a(m.s12_o_s11, f), C.c_float),
            _p(_a(m.s22_o_s11, f), C.c_float), _p(_a(m.s33_o_s11, f), C.c_float),
            _p(_a(m.s34_o_s11, f), C.c_float), _p(_a(m.s44_o_s11, f), C.c_float),
            _p(_a(m.tab_g_pos, f), C.c_float)), "mcgpu_set_scattering")
        pe = getattr(m, "prob_E_cell", None)
        self._chk(L.mcgpu_set_thermal(
            self.ctx, C.c_int(m.tab_Temp.size), _p(_a(m.tab_Temp, f), C.c_float),
            # (both None: the tables are left to init_reemission(), i.e. built on the device)
            None if m.log_Qcool is None else _p(_a(m.log_Qcool, d), C.c_double),
            None if m.kdB_dT_CDF is None else _p(_a(m.kdB_dT_CDF, d), C.c_double),
            _p(_a(m.spectre_emission_cumul, d), C.c_double), _p(_a(m.frac_E_stars, d), C.c_double),
            _p(_a(m.frac_E_disk, d), C.c_double), _p(_a(m.CDF_E_star, d), C.c_double),
            None if pe is None else _p(_a(pe, d), C.c_double),
            C.c_double(m.L_packet_th(n_tot)), C.c_float(cfg.T_min)), "mcgpu_set_thermal")
        self._chk(L.mcgpu_set_sed_bins(
            self.ctx, C.c_int(cfg.N_thet), C.c_int(cfg.N_phi), C.c_int(int(cfg.l_sym_centrale)),
            C.c_int(int(cfg.l_sym_axiale))), "mcgpu_set_sed_bins")
        if getattr(m, "variable_dust", None) is not None:
            self.set_variable_dust(m.variable_dust)
        if getattr(m, "mrw", None) is not None:      # (after the classes: its tables have one row per class)
            self.set_mrw(m.mrw)

    def set_variable_dust(self, vd):
        """Per-class tables of ``lvariable_dust`` (``mcfost_amd.host.model.init_variable_dust``); ``None``: one class."""
        if vd is None:
            self._chk(self.lib.mcgpu_set_variable_dust(self.ctx, C.c_int(0), *([None] * 13)),
                      "mcgpu_set_variable_dust")
            return
        d = np.float64
        self._chk(self.lib.mcgpu_set_variable_dust(
            self.ctx, C.c_int(int(vd["p_n_cells"])), _p(_a(vd["p_icell"], np.int32), C.c_int),
            _p(_a(vd["kappa"], d), C.c_double), _p(_a(vd["kappa_abs_LTE"], d), C.c_double),
            _p(_a(vd["albedo"], np.float32), C.c_float),
            None if vd.get("log_Qcool") is None else _p(_a(vd["log_Qcool"], d), C.c_double),
            None if vd.get("kdB_dT_CDF") is None else _p(_a(vd["kdB_dT_CDF"], d), C.c_double),
            *[(_p(_a(vd[k], np.float32), C.c_float) if vd.get("prob_s11_pos") is not None else None)
              for k in ("prob_s11_pos", "s12_o_s11", "s22_o_s11", "s33_o_s11", "s34_o_s11", "s44_o_s11", "tab_g_pos")]),
            "mcgpu_set_variable_dust")
        if getattr(self.model, "method1", None) is not None:
            self.set_scattering_method1(self.model.method1)
        if vd.get("tab_s11_pos") is not None:   # the phase function of the rt1 deposits, per class
            self._chk(self.lib.mcgpu_set_variable_dust_s11(self.ctx, _p(_a(vd["tab_s11_pos"], np.float32), C.c_float)),
                      "mcgpu_set_variable_dust_s11")

    def build_ksca_CDF(self, build=True, fetch=True):
        """``ksca_CDF(0:n_grains, p_n_cells, n_lambda)`` on the device (``mcgpu_build_ksca_CDF``): scattering method 1 then selects
        the grain by ``select_grainsize_high_mem``'s dichotomy.  Returns the table ``[n_lambda, p_n_cells, n_grains + 1]``
        (C order of the reference's layout) when ``fetch``."""
        if not build:
            self._chk(self.lib.mcgpu_build_ksca_CDF(self.ctx, C.c_int(0), None), "mcgpu_build_ksca_CDF")
            return None
        m1 = self.model.method1
        out = np.zeros((self.model.n_lambda, int(self.model.variable_dust["p_n_cells"]), int(m1["n_grains"]) + 1), np.float64) if fetch else None
        self._chk(self.lib.mcgpu_build_ksca_CDF(self.ctx, C.c_int(1), _p(out, C.c_double) if fetch else None), "mcgpu_build_ksca_CDF")
        return out

    def opacity(self, grains, p_icell, dens, fetch=True):
        """``opacity`` + ``calc_local_scattering_matrices`` (dust_prop.f90:791-1243) on the device: builds the per-class
        opacity and scattering tables of the context from the grains' tables (``mcfost_amd.host.model.synthetic_grains``)
        and ``dens [p_n_cells, n_grains]``; they replace what ``mcgpu_set_variable_dust`` would have set (call
        ``init_reemission`` next).  ``fetch``: returns them in the reference's layouts (C-ordered, wavelength first)."""
        m = self.model
        f32, f64 = np.float32, np.float64
        dens = _a(dens, f64)
        nc, nl, na1 = dens.shape[0], m.n_lambda, int(np.asarray(grains["tab_s11"]).shape[-1])
        pola = bool(m.cfg.lsepar_pola) and int(m.cfg.aniso_method) == 1
        keep = [_a(grains[k], f32) for k in ("C_ext", "C_sca", "C_abs", "tab_g", "tab_s11", "tab_s12", "tab_s22", "tab_s33",
                                            "tab_s34", "tab_s44", "S_grain")]
        nk = _a(grains["n_grains_k"], f64)
        G = GrainTables(int(grains["n_grains"]), int(grains["grain_RE_LTE_start"]), int(grains["grain_RE_LTE_end"]),
                        *[_p(v, C.c_float) for v in keep[:10]], _p(keep[10], C.c_float), _p(nk, C.c_double))
        out, O = None, None
        if fetch:
            pcols = 1 if int(m.p_lambda_fixed) else nl
            out = dict(kappa=np.zeros((nl, nc), f64), kappa_abs_LTE=np.zeros((nl, nc), f64), tab_albedo_pos=np.zeros((nl, nc), f32),
                       tab_g_pos=np.zeros((nl, nc), f32), tab_s11_pos=np.zeros((nl, nc, na1), f32),
                       prob_s11_pos=np.zeros((pcols, nc, na1), f32))
            for k in ("tab_s12_o_s11_pos", "tab_s22_o_s11_pos", "tab_s33_o_s11_pos", "tab_s34_o_s11_pos", "tab_s44_o_s11_pos"):
                out[k] = np.zeros((nl, nc, na1), f32) if pola else None
            pf = lambda k: _p(out[k], C.c_float) if out[k] is not None else None
            O = OpacityTables(_p(out["kappa"], C.c_double), _p(out["kappa_abs_LTE"], C.c_double), pf("tab_albedo_pos"), pf("tab_g_pos"),
                              pf("tab_s11_pos"), pf("prob_s11_pos"), pf("tab_s12_o_s11_pos"), pf("tab_s22_o_s11_pos"),
                              pf("tab_s33_o_s11_pos"), pf("tab_s34_o_s11_pos"), pf("tab_s44_o_s11_pos"))
        self._chk(self.lib.mcgpu_opacity(self.ctx, C.byref(G), C.c_int(nc), _p(_a(p_icell, np.int32), C.c_int),
                                         _p(dens, C.c_double), C.byref(O) if O is not None else None), "mcgpu_opacity")
        return out

    def set_scattering_method1(self, m1):
        """Scattering method 1 (``mcfost_amd.host.model.init_scattering_method1``): the scattering grain is drawn from
        the cell's population (dust_transfer.f90:1288-1316); ``None``: back to method 2."""
        if m1 is None:
            self._chk(self.lib.mcgpu_set_scattering_method1(self.ctx, None, None, C.c_int(0), None), "mcgpu_set_scattering_method1")
            return
        f32 = np.float32
        keep = [_a(m1[k], f32) for k in ("C_sca", "C_sca", "C_sca", "tab_g", "tab_s11", "tab_s12", "tab_s22", "tab_s33", "tab_s34",
                                        "tab_s44", "C_sca")]
        nk, dens, prob = _a(m1["n_grains_k"], np.float64), _a(m1["dens"], np.float64), _a(m1["prob_s11"], f32)
        G = GrainTables(int(m1["n_grains"]), 1, int(m1["n_grains"]), *[_p(v, C.c_float) for v in keep[:10]], _p(keep[10], C.c_float),
                        _p(nk, C.c_double))
        self._chk(self.lib.mcgpu_set_scattering_method1(self.ctx, C.byref(G), _p(prob, C.c_float), C.c_int(dens.shape[0]),
                                                        _p(dens, C.c_double)), "mcgpu_set_scattering_method1")

    def init_reemission(self, fetch=True, dudt=None, heating_norm=None, ufac_implicit=0.0):
        """``init_reemission`` (thermal_emission.f90:404-550) on the device: rebuilds ``log_Qcool_minus_extra_heating``
        and ``kdB_dT_CDF`` of the context (of every class with variable dust) from its ``kappa_abs_LTE``.  Returns
        ``(log_Qcool [classes, n_T], kdB_dT_CDF [classes, n_T, n_lambda])`` when ``fetch``.  ``dudt``, ``heating_norm``
        [classes] (and ``ufac_implicit`` > 0 for ldudt_implicit): lextra_heating (:486-494, ``mcgpu_init_reemission_ex``)."""
        m = self.model
        vd = getattr(m, "variable_dust", None)
        nc = int(vd["p_n_cells"]) if vd is not None else 1
        nT, nl = m.tab_Temp.size, m.n_lambda
        lq = np.zeros((nc, nT), np.float64) if fetch else None
        cdf = np.zeros((nc, nT, nl), np.float64) if fetch else None
        du = None if dudt is None else _a(np.broadcast_to(np.asarray(dudt, np.float64), (nc,)), np.float64)
        hn = None if dudt is None else _a(np.broadcast_to(np.asarray(heating_norm, np.float64), (nc,)), np.float64)
        self._chk(self.lib.mcgpu_init_reemission_ex(
            self.ctx, _p(_a(m.lam, np.float64), C.c_double), _p(_a(m.delta_lam, np.float64), C.c_double),
            _p(du, C.c_double) if du is not None else None, _p(hn, C.c_double) if hn is not None else None,
            C.c_double(float(ufac_implicit)),
            _p(lq, C.c_double) if fetch else None, _p(cdf, C.c_double) if fetch else None), "mcgpu_init_reemission")
        return lq, cdf

    def set_mrw(self, mrw):
        """Tables of the modified random walk (``mcfost_amd.host.model.init_mrw``); ``None`` switches it off."""
        if mrw is None:
            self._chk(self.lib.mcgpu_set_mrw(self.ctx, C.c_int(0), None, None, None, None, C.c_double(2.0), C.c_int(5),
                                             None), "mcgpu_set_mrw")
            return
        d = np.float64
        self._chk(self.lib.mcgpu_set_mrw(
            self.ctx, C.c_int(mrw["zeta"].size), _p(_a(mrw["zeta"], d), C.c_double), _p(_a(mrw["chi"], d), C.c_double),
            _p(_a(mrw["kappa_dep"], d), C.c_double), _p(_a(mrw["ext"], d), C.c_double), C.c_double(mrw["gamma"]),
            C.c_int(mrw["n_inter"]),
            _p(_a(self.model.grid["r_lim"], d), C.c_double) if "r_lim" in self.model.grid else None), "mcgpu_set_mrw")
        xc = mrw.get("exit_cdf")   # the spectrum a walk leaves its sphere with (None: the emission spectrum)
        self._chk(self.lib.mcgpu_set_mrw_exit_spectrum(self.ctx, _p(_a(xc, d), C.c_double) if xc is not None else None),
                  "mcgpu_set_mrw_exit_spectrum")

    # -- the packet loop ---------------------------------------------------
    def _opts(self, n_packets, seed, first_packet, frozen, n_replicas, accumulate, grid_blocks,
              block_threads):
        return RunOpts(int(seed), int(first_packet), int(n_packets), float(n_replicas), int(frozen),
                       int(accumulate), int(grid_blocks), int(block_threads))

    def set_E_prior(self, E_prior):
        self._chk(self.lib.mcgpu_set_E_prior(self.ctx, _p(_a(E_prior, np.float64), C.c_double)),
                  "mcgpu_set_E_prior")

    def launch_thermal(self, n_packets, seed=1, first_packet=0, frozen=False, n_replicas=1.0,
                       accumulate=False, grid_blocks=0, block_threads=0):
        if accumulate and getattr(self, "_holds_global_sums", False):
            raise McgpuError("accumulate after allreduce_device: the accumulators hold the all-reduced totals, "
                             "accumulating onto them and reducing again would count them world_size times")
        if not accumulate:
            self._holds_global_sums = False
        o = self._opts(n_packets, seed, first_packet, frozen, n_replicas, accumulate, grid_blocks,
                       block_threads)
        self._chk(self.lib.mcgpu_launch_thermal(self.ctx, C.byref(o)), "mcgpu_launch_thermal")

    def sync(self):
        ms = C.c_double()
        self._chk(self.lib.mcgpu_sync(self.ctx, C.byref(ms)), "mcgpu_sync")
        return ms.value

    def fetch(self):
        m = self.model
        nl, nt, nphi = m.n_lambda, m.cfg.N_thet, m.cfg.N_phi
        E = np.zeros(m.n_cells, np.float64)
        sed = np.zeros((N_SED_TYPES, nphi, nt, nl), np.float64)
        n_sent = np.zeros(nl, np.float64)
        cnt = np.zeros(N_COUNTERS, np.uint64)
        self._chk(self.lib.mcgpu_fetch(self.ctx, _p(E, C.c_double), _p(sed, C.c_double),
                                       _p(n_sent, C.c_double), _p(cnt, C.c_uint64)), "mcgpu_fetch")
        return dict(E_abs=E, sed=sed, n_sent=n_sent,
                    counters=dict(zip(COUNTER_NAMES, (int(c) for c in cnt))))

    def run_thermal(self, n_packets, seed=1, first_packet=0, frozen=False, E_prior=None,
                    n_replicas=1.0, accumulate=False, grid_blocks=0, block_threads=0):
        """``mc_photon_loop`` for the thermal step; returns E_abs, sed, n_sent,
        counters and the kernel time in ms."""
        if E_prior is not None:
            self.set_E_prior(E_prior)
        self.launch_thermal(n_packets, seed, first_packet, frozen, n_replicas, accumulate,
                            grid_blocks, block_threads)
        ms = self.sync()
        out = self.fetch()
        out["kernel_ms"] = ms
        return out

    def temp_finale(self, E_abs=None):
        T = np.zeros(self.model.n_cells, np.float32)
        ep = None if E_abs is None else _p(_a(E_abs, np.float64), C.c_double)
        self._chk(self.lib.mcgpu_temp_finale(self.ctx, ep, _p(T, C.c_float)), "mcgpu_temp_finale")
        return T

    def set_stream(self, stream_ptr):
        self._chk(self.lib.mcgpu_set_stream(self.ctx, C.c_void_p(stream_ptr)), "mcgpu_set_stream")

    def device_accumulators(self):
        """(fused f64 accumulator, u64 counters) as zero-copy torch tensors on
        this context's device: what the RCCL all-reduce operates on."""
        import torch

        acc, cnt, n = C.c_void_p(), C.c_void_p(), C.c_uint64()
        self._chk(self.lib.mcgpu_device_accumulators(self.ctx, C.byref(acc), C.byref(n), C.byref(cnt)),
                  "mcgpu_device_accumulators")
        dev = torch.device("cuda", self.device)
        t_acc = torch.as_tensor(_DevArray(acc.value, n.value, "<f8"), device=dev)
        t_cnt = torch.as_tensor(_DevArray(cnt.value, N_COUNTERS, "<i8"), device=dev)
        return t_acc, t_cnt

    def define_dark_zone(self, lam, tau_max=1500.0):
        """``define_dark_zone`` (optical_depth.f90:1425-1651, 2D): returns (l_dark_zone[n_cells] u8, ri_in, ri_out, zj_sup)."""
        m, g = self.model, self.model.grid
        dz = np.zeros(m.n_cells, np.uint8)
        zj = np.zeros(g["n_rad"], np.int32)
        a, b = C.c_int(), C.c_int()
        d = np.float64
        self._chk(self.lib.mcgpu_define_dark_zone(
            self.ctx, C.c_int(int(lam)), C.c_double(float(tau_max)), _p(_a(g["r_lim"], d), C.c_double),
            _p(_a(g["r_grid"], d), C.c_double), _p(_a(g["z_grid"], d), C.c_double), _p(_a(g["z_lim"], d), C.c_double),
            _p(dz, C.c_ubyte), C.byref(a), C.byref(b), _p(zj, C.c_int)), "mcgpu_define_dark_zone")
        return dz, a.value, b.value, zj

    def temp_approx_diffusion_vertical(self, Tdust, ri_in, ri_out, zj_sup):
        """``Temp_approx_diffusion_vertical`` (diffusion.f90:292-374) on the device: returns (Tdust, iterations)."""
        m = self.model
        T = np.array(Tdust, np.float32)
        n_it = C.c_int()
        self._chk(self.lib.mcgpu_temp_approx_diffusion_vertical(
            self.ctx, _p(_a(m.lam, np.float64), C.c_double), _p(_a(m.delta_lam, np.float64), C.c_double),
            C.c_int(int(ri_in)), C.c_int(int(ri_out)), _p(_a(zj_sup, np.int32), C.c_int), _p(T, C.c_float),
            C.byref(n_it)), "mcgpu_temp_approx_diffusion_vertical")
        return T, n_it.value

    def fetch_radiation_field(self, xN=True, xJ=True):
        """xN_abs [n_cells] and xJ_abs [n_lambda, n_cells] of the last thermal launch (needs
        ``set_option("radiation_field", bits)`` before it; radiation_field.f90:54-55)."""
        m = self.model
        a = np.zeros(m.n_cells) if xN else None
        b = np.zeros((m.n_lambda, m.n_cells)) if xJ else None
        self._chk(self.lib.mcgpu_fetch_radiation_field(self.ctx, _p(a, C.c_double) if xN else None,
                                                       _p(b, C.c_double) if xJ else None), "mcgpu_fetch_radiation_field")
        return a, b

    def allreduce_device(self, all_reduce):
        """ONE collective per temperature iteration: the counters join the fused accumulator as doubles
        (``mcgpu_counters_to_accum``), ``all_reduce(tensor)`` sums it over the ranks (RCCL with the ``nccl``
        backend), the summed counters go back (``mcgpu_counters_from_accum``).  Afterwards this rank holds the GLOBAL
        sums: an accumulating launch on top of them is refused (every rank would add its new part to a copy of the
        old total: world_size-fold overcount) -- use ``mcgpu_multi_run_thermal``, which rescales, or reduce once."""
        import torch
        self._holds_global_sums = True

        self._chk(self.lib.mcgpu_counters_to_accum(self.ctx), "mcgpu_counters_to_accum")
        torch.cuda.synchronize(self.device)        # the engine's stream -> the collective's stream
        acc, _ = self.device_accumulators()
        all_reduce(acc)
        torch.cuda.synchronize(self.device)
        self._chk(self.lib.mcgpu_counters_from_accum(self.ctx), "mcgpu_counters_from_accum")

    def set_option(self, name, value):
        """Per-context run option (``mcgpu_set_option``): "deposit" 0/1/2 = auto / HBM atomics / LDS,
        "schedule" 0/1 = auto / single-role kernel, "speculation" 1/0 (SED mode)."""
        self._chk(self.lib.mcgpu_set_option(self.ctx, name.encode(), C.c_int(int(value))), "mcgpu_set_option")

    def repartition_energie(self, lam, Tdust, E_ISM=0.0, weight=None, fetch=True):
        """``repartition_energie(lam)`` on the device (``mcgpu_repartition_energie``): returns frac_E_stars, frac_E_disk,
        E_disk and (``fetch``) prob_E_cell(0:n_cells) of the 1-based wavelength ``lam``; the cumulative distribution stays
        on the device for a following ``run_mono(lam, ..., device_tables=True)``."""
        m = self.model
        T = _a(Tdust, np.float32)
        wgt = _a(weight, np.float32) if weight is not None else None
        pe = np.zeros(m.n_cells + 1, np.float64) if fetch else None
        fs, fd, ed = C.c_double(), C.c_double(), C.c_double()
        self._chk(self.lib.mcgpu_repartition_energie(
            self.ctx, C.c_int(int(lam)), C.c_double(float(m.lam[lam - 1])), C.c_double(float(m.E_stars[lam - 1])),
            C.c_double(float(E_ISM)), _p(T, C.c_float), _p(wgt, C.c_float) if wgt is not None else None,
            C.byref(fs), C.byref(fd), C.byref(ed), _p(pe, C.c_double) if pe is not None else None),
            "mcgpu_repartition_energie")
        return dict(frac_E_stars=fs.value, frac_E_disk=fd.value, E_disk=ed.value, prob_E_cell=pe)

    def get_info(self, name):
        """Diagnostics of the last launches (``mcgpu_get_info``), e.g. "bin_chunks", "bin_overflow_blocks"."""
        v = C.c_double(0.0)
        self._chk(self.lib.mcgpu_get_info(self.ctx, name.encode(), C.byref(v)), "mcgpu_get_info")
        return v.value

    def set_xI_precision(self, bytes_per_value):
        """8 (default): FP64 xI_scatt sums; 4: default real like the reference's array, half the atomic lines -- for
        runs with many observers (``mcgpu_set_xI_precision``)."""
        self._chk(self.lib.mcgpu_set_xI_precision(self.ctx, C.c_int(int(bytes_per_value))), "mcgpu_set_xI_precision")

    def device_xI(self):
        """The xI_scatt accumulator (engine layout, FP64 or FP32 as set) as a zero-copy torch tensor."""
        import torch

        p, n = C.c_void_p(), C.c_uint64()
        self._chk(self.lib.mcgpu_device_xI(self.ctx, C.byref(p), C.byref(n)), "mcgpu_device_xI")
        self.lib.mcgpu_get_xI_precision.argtypes = [C.c_void_p]
        dt = "<f4" if self.lib.mcgpu_get_xI_precision(self.ctx) == 4 else "<f8"
        return torch.as_tensor(_DevArray(p.value, n.value, dt), device=torch.device("cuda", self.device))

    # -- probes (parity tests) ---------------------------------------------
    def probe_cross_cell(self, x0, y0, z0, u, v, w, cell):
        n = len(cell)
        ins = [_a(q, np.float64) for q in (x0, y0, z0, u, v, w)]
        cell = _a(cell, np.int32)
        x1, y1, z1, l = (np.zeros(n) for _ in range(4))
        nxt = np.zeros(n, np.int32)
        self._chk(self.lib.mcgpu_probe_cross_cell(
            self.ctx, C.c_int(n), *[_p(q, C.c_double) for q in ins], _p(cell, C.c_int),
            _p(x1, C.c_double), _p(y1, C.c_double), _p(z1, C.c_double), _p(nxt, C.c_int),
            _p(l, C.c_double)), "mcgpu_probe_cross_cell")
        return x1, y1, z1, nxt, l

    # -- SED mode ------------------------------------------------------------
    def set_rt1(self):
        m, rt = self.model, self.model.rt
        d = np.float64
        self._chk(self.lib.mcgpu_set_rt1(
            self.ctx, C.c_int(rt["RT_n_incl"]), C.c_int(rt["RT_n_az"]), _p(_a(rt["tab_u_rt"], d), C.c_double),
            _p(_a(rt["tab_v_rt"], d), C.c_double), _p(_a(rt["tab_w_rt"], d), C.c_double), C.c_int(rt["n_az_rt"]),
            C.c_int(rt["n_theta_rt"]), C.c_int(rt["N_type_flux"]), C.c_int(rt["lsepar_contrib"]),
            _p(_a(m.tab_s11_pos, np.float32), C.c_float), C.c_int(m.n_lambda)), "mcgpu_set_rt1")
        self._rt1 = True

    def xI_shape(self):
        rt = self.model.rt
        return (self.model.n_cells, rt["RT_n_incl"] * rt["RT_n_az"], rt["N_type_flux"], rt["n_theta_rt"],
                rt["n_az_rt"])

    def fetch_xI(self):
        """xI_scatt in the reference's layout (see ``xI_shape``), FP64 sums: ``mcgpu_fetch_xI``."""
        x64 = np.zeros(self.xI_shape(), np.float64)
        self._chk(self.lib.mcgpu_fetch_xI(self.ctx, None, _p(x64, C.c_double)), "mcgpu_fetch_xI")
        return x64

    def set_xI(self, xI_scatt):
        """Replace the device xI_scatt by a host array of ``xI_shape()`` (``mcgpu_set_xI``)."""
        if not getattr(self, "_rt1", False):
            self.set_rt1()
        x = _a(xI_scatt, np.float64)
        if x.shape != self.xI_shape():
            raise McgpuError(f"xI_scatt has shape {x.shape}, expected {self.xI_shape()}")
        self._chk(self.lib.mcgpu_set_xI(self.ctx, _p(x, C.c_double)), "mcgpu_set_xI")

    def set_rt2(self, n_theta_I=15, n_phi_I=15):
        """Ray tracing method 2 (``lscatt_ray_tracing2``, 2D): allocates ``I_spec`` / ``I_spec_star`` on the device
        (dust_ray_tracing.f90:102-105: 15 x 15 direction bins by default)."""
        cfg = self.model.cfg
        ntf = (4 if (cfg.lsepar_pola and cfg.aniso_method == 1) else 1) + (4 if cfg.lsepar_contrib else 0)
        self._chk(self.lib.mcgpu_set_rt2(self.ctx, C.c_int(int(n_theta_I)), C.c_int(int(n_phi_I)), C.c_int(ntf),
                                         C.c_int(int(cfg.lsepar_contrib))), "mcgpu_set_rt2")
        self._rt2 = (int(n_theta_I), int(n_phi_I), ntf)

    def fetch_I_spec(self):
        """``(I_spec [n_cells, n_phi_I, n_theta_I, N_type_flux], I_spec_star [n_cells])`` in FP64 (the device's sums)."""
        nt, nphi, ntf = self._rt2
        a = np.zeros((self.model.n_cells, nphi, nt, ntf), np.float64)
        b = np.zeros(self.model.n_cells, np.float64)
        self._chk(self.lib.mcgpu_fetch_I_spec(self.ctx, None, _p(a, C.c_double), None, _p(b, C.c_double)), "mcgpu_fetch_I_spec")
        return a, b

    def set_I_spec(self, I_spec, I_spec_star):
        """Hands ``I_spec [n_cells, n_phi_I, n_theta_I, N_type_flux]`` / ``I_spec_star [n_cells]`` to the device."""
        self._chk(self.lib.mcgpu_set_I_spec(self.ctx, _p(_a(I_spec, np.float64), C.c_double),
                                            _p(_a(I_spec_star, np.float64), C.c_double)), "mcgpu_set_I_spec")

    def init_dust_source_fct2(self, lam, ibin, I_spec, I_spec_star, Tdust, n_sent_photons, E_disk, nang_rt=15, nang_star=1000,
                              p_lambda=None):
        """``init_dust_source_fct2`` of inclination ``ibin`` on the device (``mcgpu_rt2_source``); ``I_spec`` = None: the
        field the last ``run_mono(rt2=...)`` left in HBM.  Returns ``(eps_dust2 [n_cells, 2, nang_rt, N_type_flux],
        eps_dust2_star [n_cells, 2, nang_star, n_Stokes])``."""
        m = self.model
        if not getattr(self, "_rt1", False):
            self.set_rt1()
        if I_spec is not None:
            shp = np.asarray(I_spec).shape
            if getattr(self, "_rt2", (0, 0, 0))[:2] != (shp[2], shp[1]):
                self.set_rt2(shp[2], shp[1])
            self.set_I_spec(I_spec, I_spec_star)
        nt, nphi, ntf = self._rt2
        ns = 4 if (m.cfg.lsepar_pola and m.cfg.aniso_method == 1) else 1
        o = RtOpts(int(lam), float(m.lam[lam - 1]), float(m.E_stars[lam - 1] + E_disk), float(n_sent_photons),
                   float(m.cfg.distance), 0.0, 0, 100.0, float(m.cfg.rin), float(m.cfg.rout))
        eps = np.zeros((m.n_cells, 2, nang_rt, ntf), np.float32)
        eps_star = np.zeros((m.n_cells, 2, nang_star, ns), np.float32)
        ms = C.c_double()
        self._chk(self.lib.mcgpu_rt2_source(
            self.ctx, C.byref(o), C.c_int(int(p_lambda or lam)), C.c_int(int(ibin)), _p(_a(Tdust, np.float32), C.c_float),
            _p(_a(m.grid["r_grid"], np.float64), C.c_double), _p(_a(np.abs(m.grid["z_grid"]), np.float64), C.c_double),
            C.c_int(nang_rt), C.c_int(nang_star), _p(eps, C.c_float), _p(eps_star, C.c_float), C.byref(ms)), "mcgpu_rt2_source")
        self.last_rt2_ms = ms.value
        return eps, eps_star

    def rt2_dust_map_sed(self, lam, Tdust, n_sent_photons, E_disk, l_sym_ima=True, tau_dark_zone_obs=100.0):
        """Ray-traced SED of the dust with method 2's source function (the inclination of the last
        ``init_dust_source_fct2``): (N_type_flux,) and the kernel time."""
        m = self.model
        rt = m.rt
        o = RtOpts(int(lam), float(m.lam[lam - 1]), float(m.E_stars[lam - 1] + E_disk), float(n_sent_photons),
                   float(m.cfg.distance), 0.0, int(l_sym_ima), float(tau_dark_zone_obs), float(m.cfg.rin), float(m.cfg.rout))
        out = np.zeros(self._rt2[2], np.float64)
        ms = C.c_double()
        self._chk(self.lib.mcgpu_rt2_dust_map(self.ctx, C.byref(o), _p(_a(rt["tab_RT_az"], np.float32), C.c_float),
                                              _p(_a(Tdust, np.float32), C.c_float), _p(out, C.c_double), C.byref(ms)),
                  "mcgpu_rt2_dust_map")
        return out, ms.value

    def rt2_dust_map_image(self, lam, Tdust, n_sent_photons, E_disk, npix_x, npix_y, map_size, zoom=1.0, l_sym_ima=False,
                           tau_dark_zone_obs=100.0):
        """The image of that inclination with method 2's source function: (N_type_flux, npix_y, npix_x), rays, kernel ms."""
        m = self.model
        rt = m.rt
        o = RtOpts(int(lam), float(m.lam[lam - 1]), float(m.E_stars[lam - 1] + E_disk), float(n_sent_photons),
                   float(m.cfg.distance), 0.0, int(l_sym_ima), float(tau_dark_zone_obs), float(m.cfg.rin), float(m.cfg.rout))
        img = np.zeros((self._rt2[2], npix_y, npix_x), np.float64)
        n = C.c_uint64(0)
        ms = C.c_double()
        self._chk(self.lib.mcgpu_rt2_image(self.ctx, C.byref(o), _p(_a(rt["tab_RT_az"], np.float32), C.c_float),
                                           _p(_a(Tdust, np.float32), C.c_float), C.c_int(npix_x), C.c_int(npix_y),
                                           C.c_double(map_size), C.c_double(zoom), _p(img, C.c_double), C.byref(n), C.byref(ms)),
                  "mcgpu_rt2_image")
        return img, int(n.value), ms.value

    def run_mono(self, lam, n_photons2, n_phot_lim=None, p_lambda=None, seed=1, n_chunks=None, rt1=True,
                 accumulate=False, grid_blocks=0, block_threads=0, fetch_xI=True, first_chunk=0, device_tables=None,
                 rt2=None):
        """One wavelength (1-based ``lam``) of the SED Monte Carlo: ``mcgpu_run_mono`` with the
        model's frac_E_stars / prob_E_cell of that wavelength -- or, with ``device_tables`` = the dict
        ``repartition_energie(lam, ...)`` returned, with the fractions of that call and the cumulative distribution it
        left on the device.  ``rt2 = (n_theta_I, n_phi_I)``: method 2's deposits (``I_spec``, ``I_spec_star``) instead."""
        m = self.model
        if rt2 is not None:
            if getattr(self, "_rt2", (0, 0, 0))[:2] != (int(rt2[0]), int(rt2[1])):
                self.set_rt2(*rt2)
            rt1 = 2
        if rt1 == 1 and not getattr(self, "_rt1", False):
            self.set_rt1()
        nt, nphi = m.cfg.N_thet, m.cfg.N_phi
        n_chunks = int(n_chunks or m.cfg.n_photons_loop)
        if n_phot_lim is None:  # read_param.f90:551
            n_phot_lim = float(np.float32(1.0e4) * np.float32(nt) * np.float32(nphi) * np.float32(n_photons2))
        o = MonoOpts(seed, int(lam), int(p_lambda or lam), n_chunks, int(first_chunk), int(n_photons2), float(n_phot_lim),
                     int(m.capt_sup), int(rt1), int(accumulate), grid_blocks, block_threads)
        pe = getattr(m, "prob_E_cell", None)
        pe_l = None
        fs, fd = float(m.frac_E_stars[lam - 1]), float(m.frac_E_disk[lam - 1])
        if device_tables is not None:
            fs, fd = float(device_tables["frac_E_stars"]), float(device_tables["frac_E_disk"])
        elif pe is not None:
            pe_l = _a(np.asarray(pe).reshape(m.n_lambda, m.n_cells + 1)[lam - 1], np.float64)
        per_chunk = np.zeros(n_chunks, np.uint64)
        ms = C.c_double()
        self._chk(self.lib.mcgpu_run_mono(
            self.ctx, C.byref(o), C.c_double(fs), C.c_double(fd),
            _p(pe_l, C.c_double) if pe_l is not None else None, _p(per_chunk, C.c_uint64), C.byref(ms)),
            "mcgpu_run_mono")
        out = self.fetch()
        out["n_sent_chunk"] = per_chunk
        out["kernel_ms"] = ms.value
        if rt1 == 1 and fetch_xI:
            x64 = np.zeros(self.xI_shape(), np.float64)
            x32 = np.zeros(self.xI_shape(), np.float32)
            self._chk(self.lib.mcgpu_fetch_xI(self.ctx, _p(x32, C.c_float), _p(x64, C.c_double)), "mcgpu_fetch_xI")
            out["xI_scatt"], out["xI_scatt_f32"] = x64, x32
        if rt1 == 2:
            out["I_spec"], out["I_spec_star"] = self.fetch_I_spec()
        return out

    def dust_map_sed(self, lam, Tdust, n_sent_photons, E_disk, ang_disque=0.0, l_sym_ima=True,
                     tau_dark_zone_obs=100.0):
        """Ray-traced SED of the dust at wavelength ``lam`` (``mcgpu_rt1_dust_map``) from the xI_scatt the
        last ``run_mono(lam, rt1=True)`` left on the device: (nRT, N_type_flux) and the kernel time."""
        m = self.model
        rt = m.rt
        o = RtOpts(int(lam), float(m.lam[lam - 1]), float(m.E_stars[lam - 1] + E_disk), float(n_sent_photons),
                   float(m.cfg.distance), float(ang_disque), int(l_sym_ima), float(tau_dark_zone_obs),
                   float(m.cfg.rin), float(m.cfg.rout))
        out = np.zeros((rt["RT_n_incl"] * rt["RT_n_az"], rt["N_type_flux"]), np.float64)
        ms = C.c_double()
        self._chk(self.lib.mcgpu_rt1_dust_map(
            self.ctx, C.byref(o), _p(_a(rt["tab_RT_az"], np.float32), C.c_float), _p(_a(Tdust, np.float32), C.c_float),
            _p(out, C.c_double), C.byref(ms)), "mcgpu_rt1_dust_map")
        return out, ms.value

    def stars_map_sed(self, lam, star_flux, seed=1, ang_disque=0.0):
        """The stars' term of the ray-traced SED (``mcgpu_rt1_stars_map_sed`` = ``compute_stars_map`` with an
        unresolved star): (nRT,) fluxes for ``star_flux[n_stars]`` = factor * prob_E_star(lambda, :)."""
        m = self.model
        rt = m.rt
        if not getattr(self, "_rt1", False):
            self.set_rt1()
        o = RtOpts(int(lam), float(m.lam[lam - 1]), 1.0, 1.0, float(m.cfg.distance), float(ang_disque), 0, 100.0,
                   float(m.cfg.rin), float(m.cfg.rout))
        out = np.zeros(rt["RT_n_incl"] * rt["RT_n_az"], np.float64)
        self._chk(self.lib.mcgpu_rt1_stars_map_sed(
            self.ctx, C.byref(o), _p(_a(rt["tab_RT_az"], np.float32), C.c_float), C.c_uint64(int(seed)),
            _p(_a(star_flux, np.float64), C.c_double), _p(out, C.c_double)), "mcgpu_rt1_stars_map_sed")
        return out

    def tau_maps(self, lam, npix_x, npix_y, map_size, zoom=1.0, tau=1.0, ang_disque=0.0, surface=True):
        """The ray tracer's optical-depth maps (``mcgpu_tau_maps`` = ``compute_tau_map`` + ``compute_tau_surface_map``):
        ``(tau_map [nRT, npix_y, npix_x], tau_surface_map [3, nRT, npix_y, npix_x] or None, kernel ms)``."""
        m = self.model
        rt = m.rt
        if not getattr(self, "_rt1", False):
            self.set_rt1()
        o = RtOpts(int(lam), float(m.lam[lam - 1]), 1.0, 1.0, float(m.cfg.distance), float(ang_disque), 0, 100.0,
                   float(m.cfg.rin), float(m.cfg.rout))
        nRT = rt["RT_n_incl"] * rt["RT_n_az"]
        tm = np.zeros((nRT, npix_y, npix_x), np.float32)
        sm = np.zeros((3, nRT, npix_y, npix_x), np.float32) if surface else None
        ms = C.c_double()
        self._chk(self.lib.mcgpu_tau_maps(
            self.ctx, C.byref(o), _p(_a(rt["tab_RT_az"], np.float32), C.c_float), C.c_int(int(npix_x)), C.c_int(int(npix_y)),
            C.c_double(float(map_size)), C.c_double(float(zoom)), C.c_double(float(tau)), _p(tm, C.c_float),
            _p(sm, C.c_float) if surface else None, C.byref(ms)), "mcgpu_tau_maps")
        return tm, sm, ms.value

    def stars_map_image(self, lam, star_flux, npix_x, npix_y, map_size, zoom=1.0, seed=1, ang_disque=0.0,
                        limb_darkening=None):
        """The stars in an image (``mcgpu_rt1_stars_map_image`` = ``compute_stars_map`` with resolved discs):
        ``(maps [nRT, n_maps, npix_y, npix_x], star_position [2, nRT, n_stars] arcsec)``; ``limb_darkening = (mu, I(mu)
        [, P(mu)])`` switches limb darkening (and, with P, the polarised maps) on."""
        m = self.model
        rt = m.rt
        if not getattr(self, "_rt1", False):
            self.set_rt1()
        o = RtOpts(int(lam), float(m.lam[lam - 1]), 1.0, 1.0, float(m.cfg.distance), float(ang_disque), 0, 100.0,
                   float(m.cfg.rin), float(m.cfg.rout))
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
        self._chk(self.lib.mcgpu_rt1_stars_map_image(
            self.ctx, C.byref(o), _p(_a(rt["tab_RT_az"], np.float32), C.c_float), C.c_uint64(int(seed)),
            _p(_a(star_flux, np.float64), C.c_double), C.c_int(npix_x), C.c_int(npix_y), C.c_double(map_size), C.c_double(zoom),
            C.c_int(n_mu), _p(mu, C.c_float) if n_mu else None, _p(ld, C.c_float) if n_mu else None,
            _p(pld, C.c_float) if pld is not None else None, _p(maps, C.c_double), _p(pos, C.c_double)),
            "mcgpu_rt1_stars_map_image")
        return maps, pos

    def dust_map_image(self, lam, Tdust, n_sent_photons, E_disk, npix_x, npix_y, map_size, zoom=1.0, ang_disque=0.0,
                       l_sym_ima=False, tau_dark_zone_obs=100.0):
        """Ray-traced image of the dust at wavelength ``lam`` (``mcgpu_rt1_image``):
        (N_type_flux, RT_n_az, RT_n_incl, npix_y, npix_x), the number of rays traced and the kernel time."""
        m = self.model
        rt = m.rt
        o = RtOpts(int(lam), float(m.lam[lam - 1]), float(m.E_stars[lam - 1] + E_disk), float(n_sent_photons),
                   float(m.cfg.distance), float(ang_disque), int(l_sym_ima), float(tau_dark_zone_obs),
                   float(m.cfg.rin), float(m.cfg.rout))
        out = np.zeros((rt["N_type_flux"], rt["RT_n_az"], rt["RT_n_incl"], int(npix_y), int(npix_x)), np.float64)
        ms, nr = C.c_double(), C.c_uint64()
        self._chk(self.lib.mcgpu_rt1_image(
            self.ctx, C.byref(o), _p(_a(rt["tab_RT_az"], np.float32), C.c_float), _p(_a(Tdust, np.float32), C.c_float),
            C.c_int(int(npix_x)), C.c_int(int(npix_y)), C.c_double(float(map_size)), C.c_double(float(zoom)),
            _p(out, C.c_double), C.byref(nr), C.byref(ms)), "mcgpu_rt1_image")
        return out, int(nr.value), ms.value

    def probe_cross_voronoi(self, x0, y0, z0, u, v, w, cell, previous_cell):
        n = len(cell)
        d = np.float64
        ins = [_a(q, d) for q in (x0, y0, z0, u, v, w)]
        outs = [np.zeros(n, d) for _ in range(6)]
        nxt = np.zeros(n, np.int32)
        self._chk(self.lib.mcgpu_probe_cross_voronoi(
            self.ctx, C.c_int(n), *[_p(q, C.c_double) for q in ins], _p(_a(cell, np.int32), C.c_int),
            _p(_a(previous_cell, np.int32), C.c_int), *[_p(q, C.c_double) for q in outs[:3]],
            _p(nxt, C.c_int), *[_p(q, C.c_double) for q in outs[3:]]), "mcgpu_probe_cross_voronoi")
        return dict(x1=outs[0], y1=outs[1], z1=outs[2], next_cell=nxt, l=outs[3], l_contrib=outs[4],
                    l_void_before=outs[5])

    def probe_index_cell(self, x, y, z):
        n = len(x)
        ins = [_a(q, np.float64) for q in (x, y, z)]
        ic = np.zeros(n, np.int32)
        self._chk(self.lib.mcgpu_probe_index_cell(self.ctx, C.c_int(n), *[_p(q, C.c_double) for q in ins],
                                                  _p(ic, C.c_int)), "mcgpu_probe_index_cell")
        return ic

    def probe_philox(self, ctr, key):
        c = (C.c_uint32 * 4)(*ctr)
        k = (C.c_uint32 * 2)(*key)
        o = (C.c_uint32 * 4)()
        self._chk(self.lib.mcgpu_probe_philox(self.ctx, c, k, o), "mcgpu_probe_philox")
        return [int(v) for v in o]

    def probe_packet_rand(self, seed, packet, n):
        out = np.zeros(n, np.float32)
        self._chk(self.lib.mcgpu_probe_packet_rand(self.ctx, C.c_uint64(seed), C.c_uint64(packet),
                                                   C.c_int(n), _p(out, C.c_float)),
                  "mcgpu_probe_packet_rand")
        return out


class MultiEngine:
    """Several GPUs of one node behind one host thread (``mcgpu_multi_*``): one context per device holding a
    replica of the model, packets sharded by id range, ONE RCCL all-reduce of the fused accumulator per
    temperature iteration inside the library."""

    def __init__(self, model, n_packets_total, devices=(0,), shared_device=False, force_rccl=False):
        """``shared_device``: every context on ONE device (``devices`` all equal) with the library's own sum kernel in
        place of the RCCL all-reduce (``MCGPU_MULTI_SHARED_DEVICE``): how a box with one GPU executes the n_dev > 1 code.
        ``force_rccl`` (``MCGPU_MULTI_FORCE_RCCL``): the communicator and the grouped ``ncclAllReduce`` also with one
        device -- how a box with one GPU executes the RCCL calls."""
        self.lib = load_library()
        self.model = model
        self.h = C.c_void_p()
        devs = (C.c_int * len(devices))(*[int(d) for d in devices])
        self.lib.mcgpu_multi_ctx.restype = C.c_void_p
        self.lib.mcgpu_multi_last_error.restype = C.c_char_p
        self.lib.mcgpu_multi_reductions.restype = C.c_uint64
        rc = self.lib.mcgpu_multi_create_ex(C.c_int(len(devices)), devs, C.c_uint((1 if shared_device else 0) | (2 if force_rccl else 0)), C.byref(self.h))
        if rc:
            self.h = C.c_void_p()
            raise McgpuError(f"mcgpu_multi_create({list(devices)}) failed with code {rc}")
        self.engines = [Engine(model, n_packets_total, device=int(d),
                               _borrowed_ctx=self.lib.mcgpu_multi_ctx(self.h, C.c_int(i)))
                        for i, d in enumerate(devices)]

    def set_E_prior(self, E_prior):
        for e in self.engines:
            e.set_E_prior(E_prior)

    def run_thermal(self, n_packets, seed=1, first_packet=0, frozen=False, E_prior=None, n_replicas=1.0,
                    accumulate=False, grid_blocks=0, block_threads=0):
        if E_prior is not None:
            self.set_E_prior(E_prior)
        e0 = self.engines[0]
        o = e0._opts(n_packets, seed, first_packet, frozen, n_replicas, accumulate, grid_blocks, block_threads)
        m = self.model
        E = np.zeros(m.n_cells, np.float64)
        sed = np.zeros((N_SED_TYPES, m.cfg.N_phi, m.cfg.N_thet, m.n_lambda), np.float64)
        n_sent = np.zeros(m.n_lambda, np.float64)
        cnt = np.zeros(N_COUNTERS, np.uint64)
        ms = C.c_double()
        rc = self.lib.mcgpu_multi_run_thermal(self.h, C.byref(o), _p(E, C.c_double), _p(sed, C.c_double),
                                              _p(n_sent, C.c_double), _p(cnt, C.c_uint64), C.byref(ms))
        if rc:
            msg = self.lib.mcgpu_multi_last_error(self.h)
            raise McgpuError(f"mcgpu_multi_run_thermal failed ({rc}): {msg.decode() if msg else ''}")
        return dict(E_abs=E, sed=sed, n_sent=n_sent, kernel_ms=ms.value,
                    counters=dict(zip(COUNTER_NAMES, (int(c) for c in cnt))))

    def run_mono(self, lam, n_photons2, n_phot_lim=None, p_lambda=None, seed=1, n_chunks=None, rt1=True,
                 accumulate=False, fetch_xI=True, first_chunk=0, Tdust=None, rt2=None):
        """One wavelength of the SED Monte Carlo on every device (``mcgpu_multi_run_mono``): the streams are split
        among the devices, ONE all-reduce of [sed | n_sent | counters] and one of xI_scatt; same outputs as
        ``Engine.run_mono``, read from device 0.  ``Tdust`` given: every device first builds the wavelength's emission
        tables itself (``mcgpu_repartition_energie``) instead of taking the model's."""
        m = self.model
        e0 = self.engines[0]
        tables = None
        if Tdust is not None:
            tables = [e.repartition_energie(lam, Tdust, fetch=False) for e in self.engines][0]
        if rt2 is not None:   # method 2's deposits (I_spec, I_spec_star) instead of xI_scatt
            for e in self.engines:
                if getattr(e, "_rt2", (0, 0, 0))[:2] != (int(rt2[0]), int(rt2[1])):
                    e.set_rt2(*rt2)
            rt1 = 2
        for e in self.engines:
            if rt1 == 1 and not getattr(e, "_rt1", False):
                e.set_rt1()
        nt, nphi = m.cfg.N_thet, m.cfg.N_phi
        n_chunks = int(n_chunks or m.cfg.n_photons_loop)
        if n_phot_lim is None:  # read_param.f90:551
            n_phot_lim = float(np.float32(1.0e4) * np.float32(nt) * np.float32(nphi) * np.float32(n_photons2))
        o = MonoOpts(seed, int(lam), int(p_lambda or lam), n_chunks, int(first_chunk), int(n_photons2), float(n_phot_lim),
                     int(m.capt_sup), int(rt1), int(accumulate), 0, 0)
        pe = getattr(m, "prob_E_cell", None)
        pe_l = None
        fs, fd = float(m.frac_E_stars[lam - 1]), float(m.frac_E_disk[lam - 1])
        if tables is not None:
            fs, fd = tables["frac_E_stars"], tables["frac_E_disk"]
        elif pe is not None:
            pe_l = _a(np.asarray(pe).reshape(m.n_lambda, m.n_cells + 1)[lam - 1], np.float64)
        per_chunk = np.zeros(n_chunks, np.uint64)
        ms = C.c_double()
        rc = self.lib.mcgpu_multi_run_mono(
            self.h, C.byref(o), C.c_double(fs), C.c_double(fd),
            _p(pe_l, C.c_double) if pe_l is not None else None, _p(per_chunk, C.c_uint64), C.byref(ms))
        if rc:
            msg = self.lib.mcgpu_multi_last_error(self.h)
            raise McgpuError(f"mcgpu_multi_run_mono failed ({rc}): {msg.decode() if msg else ''}")
        out = e0.fetch()
        out["n_sent_chunk"] = per_chunk
        out["kernel_ms"] = ms.value
        if rt1 == 2:
            out["I_spec"], out["I_spec_star"] = e0.fetch_I_spec()
        if rt1 == 1 and fetch_xI:
            x64 = np.zeros(e0.xI_shape(), np.float64)
            x32 = np.zeros(e0.xI_shape(), np.float32)
            e0._chk(self.lib.mcgpu_fetch_xI(e0.ctx, _p(x32, C.c_float), _p(x64, C.c_double)), "mcgpu_fetch_xI")
            out["xI_scatt"], out["xI_scatt_f32"] = x64, x32
        return out

    def run_sed(self, lambdas, n_photons2, Tdust, seeds=None, costs=None, n_chunks=None, n_phot_lim=None, ray_tracing=True,
                ang_disque=0.0, l_sym_ima=True, tau_dark_zone_obs=100.0, E_ISM=0.0):
        """The SED step sharded BY WAVELENGTH (``mcgpu_multi_run_sed``): every device takes whole wavelengths of ``lambdas``
        (1-based; longest first by ``costs``) -- emission tables from ``Tdust``, scout / commit passes, ray-traced SED of the
        dust -- and only the results travel: ``sed`` (n, 9, N_phi, N_thet), ``n_sent``, ``E_disk``, ``sed_rt`` (n, nRT,
        N_type_flux), ``counters``, ``device_of``, ``seconds`` per wavelength.  No xI_scatt is reduced."""
        m = self.model
        rt = m.rt
        lambdas = [int(l) for l in lambdas]
        n = len(lambdas)
        for e in self.engines:
            if not getattr(e, "_rt1", False):
                e.set_rt1()
        nt, nphi = m.cfg.N_thet, m.cfg.N_phi
        n_chunks = int(n_chunks or m.cfg.n_photons_loop)
        if n_phot_lim is None:  # read_param.f90:551
            n_phot_lim = float(np.float32(1.0e4) * np.float32(nt) * np.float32(nphi) * np.float32(n_photons2))
        o = MonoOpts(0, 0, 0, n_chunks, 0, int(n_photons2), float(n_phot_lim), int(m.capt_sup), 1, 0, 0, 0)
        wl = (SedWavelength * n)()
        for i, lam in enumerate(lambdas):
            wl[i] = SedWavelength(lam, lam, float(m.lam[lam - 1]), float(m.E_stars[lam - 1]), float(E_ISM),
                                  int(seeds[i]) if seeds is not None else 1 + lam, float(costs[i]) if costs is not None else 0.0)
        ro = RtOpts(0, 0.0, 0.0, 0.0, float(m.cfg.distance), float(ang_disque), int(l_sym_ima), float(tau_dark_zone_obs),
                    float(m.cfg.rin), float(m.cfg.rout))
        nRT = rt["RT_n_incl"] * rt["RT_n_az"]
        sed = np.zeros((n, N_SED_TYPES, nphi, nt), np.float64)
        n_sent, E_disk, seconds = np.zeros(n), np.zeros(n), np.zeros(n)
        sed_rt = np.zeros((n, nRT, rt["N_type_flux"]), np.float64)
        cnt = np.zeros((n, N_COUNTERS), np.uint64)
        dev = np.zeros(n, np.int32)
        rc = self.lib.mcgpu_multi_run_sed(
            self.h, C.byref(o), C.c_int(n), wl, _p(_a(Tdust, np.float32), C.c_float), C.byref(ro) if ray_tracing else None,
            _p(_a(rt["tab_RT_az"], np.float32), C.c_float) if ray_tracing else None, _p(sed, C.c_double), _p(n_sent, C.c_double),
            _p(E_disk, C.c_double), _p(sed_rt, C.c_double) if ray_tracing else None, _p(cnt, C.c_uint64), _p(dev, C.c_int),
            _p(seconds, C.c_double))
        if rc:
            msg = self.lib.mcgpu_multi_last_error(self.h)
            raise McgpuError(f"mcgpu_multi_run_sed failed ({rc}): {msg.decode() if msg else ''}")
        return dict(sed=sed, n_sent=n_sent, E_disk=E_disk, sed_rt=sed_rt, seconds=seconds, device_of=dev,
                    counters=[dict(zip(COUNTER_NAMES, (int(c) for c in row))) for row in cnt])

    def reductions(self):
        """Collectives the handle has executed (RCCL all-reduces or the shared-device sums standing in for them)."""
        return int(self.lib.mcgpu_multi_reductions(self.h))

    def rccl_ranks(self):
        """Ranks of the handle's RCCL communicator as ``ncclCommCount`` reports them (0 before the first collective
        and on one device, which never opens one)."""
        return int(self.lib.mcgpu_multi_rccl_ranks(self.h))

    def close(self):
        for e in getattr(self, "engines", []):
            e.close()
        self.engines = []
        if getattr(self, "h", None) and self.h.value:
            self.lib.mcgpu_multi_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
