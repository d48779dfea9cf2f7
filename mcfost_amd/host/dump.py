"""Stream-file exchange with the Fortran host example
(``mcfost_amd/fortran/thermal_host_example.f90``): the model tables in the order
the Fortran program reads them, and its result file."""
from __future__ import annotations

import numpy as np


def write_model(model, n_packets_total, path):
    m, g, cfg = model, model.grid, model.cfg
    i32, f64, f32 = np.int32, np.float64, np.float32
    st = np.asarray(m.stars, f64)
    with open(path, "wb") as f:
        np.array([g["n_rad"], g["nz"], g["n_az"], g["l3D"], g["n_cells"], g["ntot2"], g["cell_map"].size,
                  st.shape[0], m.n_lambda, 180, cfg.aniso_method, int(cfg.lisotropic), int(cfg.lsepar_pola),
                  m.tab_Temp.size, cfg.N_thet, cfg.N_phi, int(cfg.l_sym_centrale), int(cfg.l_sym_axiale)],
                 i32).tofile(f)
        np.array([g["Rmax2"], m.L_packet_th(n_packets_total)], f64).tofile(f)
        np.array([cfg.T_min], f32).tofile(f)
        for k in ("r_lim_2", "zmax", "z_lim", "tan_phi_lim", "volume"):
            np.ascontiguousarray(g[k], f64).tofile(f)
        for k in ("cell_map", "cell_map_i", "cell_map_j", "cell_map_k", "lexit_cell"):
            np.ascontiguousarray(g[k], i32).tofile(f)
        for q in range(4):
            np.ascontiguousarray(st[:, q], f64).tofile(f)
        np.ascontiguousarray(st[:, 4], i32).tofile(f)
        np.ascontiguousarray(st[:, 5], i32).tofile(f)
        np.ascontiguousarray(m.kappa, f64).tofile(f)
        np.ascontiguousarray(m.kappa_abs_LTE, f64).tofile(f)
        np.ascontiguousarray(m.albedo, f32).tofile(f)
        np.ascontiguousarray(m.kappa_factor, f64).tofile(f)
        for k in ("prob_s11_pos", "s12_o_s11", "s22_o_s11", "s33_o_s11", "s34_o_s11", "s44_o_s11", "tab_g_pos"):
            np.ascontiguousarray(getattr(m, k), f32).tofile(f)
        np.ascontiguousarray(m.tab_Temp, f32).tofile(f)
        for k in ("log_Qcool", "kdB_dT_CDF", "spectre_emission_cumul", "frac_E_stars", "frac_E_disk", "CDF_E_star"):
            np.ascontiguousarray(getattr(m, k), f64).tofile(f)


def read_result(model, path):
    m, cfg = model, model.cfg
    a = np.fromfile(path, np.float64)
    n_c = m.n_cells
    n_s = 9 * m.n_lambda * cfg.N_thet * cfg.N_phi
    assert a.size == n_c + n_s + m.n_lambda
    return dict(E_abs=a[:n_c], sed=a[n_c:n_c + n_s].reshape(9, cfg.N_phi, cfg.N_thet, m.n_lambda),
                n_sent=a[n_c + n_s:])
