// emu_conv.h -- TEST INFRASTRUCTURE shared by tests/emu/emu_kernel.cpp and tests/emu/emu_host_tail.cpp: the oracle's model
// structure converted to the device's (DevModel + VoroGrid), the tables laid out as the setters of mcgpu.hip lay them out.
// Include after the device headers and oracle/mc_oracle.h.
#pragma once
#include <vector>

using namespace mcgpu;

// oracle_model -> DevModel (+ VoroGrid), as the setters of mcgpu.hip do
struct Conv {
  DevModel M;
  VoroGrid G;
  bool voro;
  std::vector<double> ch, sx, ct, vk, vka, kfpad;
  std::vector<float> val, vsc[8];
  std::vector<int> sc, vcls;
  std::vector<VoroCell> vcell;
  std::vector<VoroNb> vnb;
  std::vector<unsigned char> vnbcls;   // VoroGrid::nb_cls as mcgpu_set_grid_voronoi builds it
  double dummy = 0.0;
  explicit Conv(const oracle_model* m) {
    memset(&M, 0, sizeof(M));
    memset(&G, 0, sizeof(G));
    voro = m->grid_type == 3;
    M.n_rad = m->n_rad; M.nz = m->nz; M.n_az = m->n_az; M.l3D = m->l3D; M.n_cells = m->n_cells;
    M.r_lim_2 = m->r_lim_2; M.zmax = m->zmax; M.tan_phi_lim = m->tan_phi_lim;
    ch.assign(m->n_rad > 0 ? m->n_rad : 1, 0.0);
    if (voro) { M.n_rad = 0; M.nz = 0; M.n_az = 0; M.l3D = 1; M.r_lim_2 = &dummy; }
    else if (m->grid_type == 2) for (int i = 0; i < m->n_rad; ++i) ch[i] = 1.0;
    else for (int i = 0; i < m->n_rad; ++i) ch[i] = m->nz >= 2 ? m->z_lim[i + m->n_rad] : m->zmax[i];
    M.ch = ch.data();
    M.zmaxmax = m->zmaxmax; M.Rmax2 = m->Rmax2; M.volume = m->volume;
    if (voro) {  // the records mcgpu_set_grid_voronoi + mcgpu_set_opacity build
      vcell.resize(m->n_cells);
      vnb.resize(m->v_last[m->n_cells - 1]);
      for (int i = 0; i < m->n_cells; ++i) {
        VoroCell& Cc = vcell[i];
        Cc.x = m->v_xyz[3 * i]; Cc.y = m->v_xyz[3 * i + 1]; Cc.z = m->v_xyz[3 * i + 2];
        Cc.first = m->v_first[i] - 1; Cc.count = m->v_last[i] - m->v_first[i] + 1;
        Cc.flags = (m->v_was_cut && m->v_was_cut[i] ? 1 : 0) | (m->v_is_star_neighbour && m->v_is_star_neighbour[i] ? 2 : 0);
        Cc.kf = m->kappa_factor[i];
        for (int q = m->v_first[i] - 1; q < m->v_last[i]; ++q) {
          const int id = m->v_neigh[q];
          vnb[q].id = id;
          if (id > 0) { vnb[q].x = m->v_xyz[3 * (id - 1)]; vnb[q].y = m->v_xyz[3 * (id - 1) + 1]; vnb[q].z = m->v_xyz[3 * (id - 1) + 2]; }
          else { vnb[q].x = vnb[q].y = vnb[q].z = 0.0f; }
        }
      }
#ifdef EMU_CONV_POOL   // (the pool schedule's list-length classes: mc_voronoi_pool.hip.h)
      vnbcls.assign(vnb.size(), (unsigned char)0);
      for (size_t q = 0; q < vnb.size(); ++q)
        if (vnb[q].id > 0) vnbcls[q] = (unsigned char)vp_class_of(vcell[vnb[q].id - 1].count);
      G.nb_cls = vnbcls.data();
#endif
      G.n_cells = m->n_cells; G.cell = vcell.data(); G.nb = vnb.data(); G.h = m->v_h; G.xyz_dp = m->v_xyz_dp;
      G.wall_first = m->v_wall_first; G.wall_cells = m->v_wall_cells; G.cut_o_h = m->v_cut_o_h;
      memcpy(G.walls, m->v_walls, 24 * sizeof(float));
    }
    M.n_stars = m->n_stars;
    sx.resize(4 * m->n_stars);
    sc.resize(4 * m->n_stars);
    for (int s = 0; s < m->n_stars; ++s) {
      sx[4 * s] = m->stars[s].x; sx[4 * s + 1] = m->stars[s].y; sx[4 * s + 2] = m->stars[s].z; sx[4 * s + 3] = m->stars[s].r;
      int ic = m->stars[s].icell;
      if (voro) { sc[4 * s] = ic; sc[4 * s + 1] = 0; sc[4 * s + 2] = 0; }
      else { sc[4 * s] = m->cell_map_i[ic - 1]; sc[4 * s + 1] = m->cell_map_j[ic - 1]; sc[4 * s + 2] = m->cell_map_k[ic - 1]; }
      sc[4 * s + 3] = m->stars[s].out_model;
    }
    M.star_xyzr = sx.data(); M.star_cell = sc.data();
    M.n_lambda = m->n_lambda; M.kappa = m->kappa; M.kappa_abs = m->kappa_abs_LTE; M.albedo = m->albedo;
    kfpad.assign(m->kappa_factor, m->kappa_factor + m->n_cells); kfpad.push_back(0.0);  // (+ the entry of "no cell")
    M.kappa_factor = kfpad.data();
    bool any_dark = false;
    if (m->l_dark_zone) for (int i = 0; i < m->n_cells; ++i) any_dark |= m->l_dark_zone[i] != 0;
    M.dark = any_dark ? m->l_dark_zone : nullptr;
    M.nang = m->nang_scatt; M.aniso_method = m->aniso_method; M.lisotropic = m->lisotropic;
    M.p_lambda_fixed = m->p_lambda_fixed;
    M.prob_s11 = m->prob_s11_pos; M.s12 = m->s12_o_s11; M.s22 = m->s22_o_s11; M.s33 = m->s33_o_s11;
    M.s34 = m->s34_o_s11; M.s44 = m->s44_o_s11; M.tab_g = m->tab_g_pos;
    ct.resize(m->nang_scatt + 1);
    for (int k = 0; k <= m->nang_scatt; ++k) ct[k] = std::cos(((double)k) * PI / (double)m->nang_scatt);
    M.cos_tab = ct.data();
    M.n_T = m->n_T; M.log_Qcool = m->log_Qcool; M.cdf = m->kdB_dT_CDF; M.spec_cum = m->spectre_emission_cumul;
    M.frac_E_stars = m->frac_E_stars; M.frac_E_disk = m->frac_E_disk; M.CDF_E_star = m->CDF_E_star;
    M.prob_E_cell = m->prob_E_cell; M.L_packet_th = m->L_packet_th;
    M.N_thet = m->N_thet; M.N_phi = m->N_phi; M.sym_c = m->l_sym_centrale; M.sym_a = m->l_sym_axiale;
    M.midplane_snap = m->midplane_snap;
    M.grid_sph = m->grid_type == 2;
    M.tan_theta_lim = m->tan_theta_lim; M.theta_lim = m->theta_lim; M.r_lim_3 = m->r_lim_3;
    M.R_ISM = m->R_ISM;
    for (int q = 0; q < 3; ++q) M.centre_ISM[q] = m->centre_ISM[q];
    M.mrw = m->mrw; M.mrw_n_zeta = m->mrw_n_zeta; M.mrw_n_inter = m->mrw_n_inter; M.mrw_gamma = m->mrw_gamma;
    M.mrw_zeta = m->mrw_zeta; M.mrw_chi = m->mrw_chi; M.mrw_kdep = m->mrw_kappa_dep; M.mrw_ext = m->mrw_ext; M.mrw_exit_cdf = m->mrw_exit_cdf;
    M.r_lim = m->r_lim;
    M.sin_phi = m->sin_phi_lim; M.cos_phi = m->cos_phi_lim;   // (the walk's azimuthal walls, 3D)
    M.n_classes = m->p_n_cells;
    if (m->p_n_cells) {  // class-major copies, as mcgpu_set_variable_dust lays them out
      const int nc = m->p_n_cells, nl = m->n_lambda;
      vcls.resize(m->n_cells); vk.resize((size_t)nc * nl); vka.resize((size_t)nc * nl); val.resize((size_t)nc * nl);
      for (int i = 0; i < m->n_cells; ++i) vcls[i] = m->p_icell[i] - 1;
      for (int c = 0; c < nc; ++c)
        for (int l = 0; l < nl; ++l) {
          vk[(size_t)c * nl + l] = m->v_kappa[c + (size_t)nc * l];
          vka[(size_t)c * nl + l] = m->v_kappa_abs_LTE[c + (size_t)nc * l];
          val[(size_t)c * nl + l] = m->v_albedo[c + (size_t)nc * l];
        }
      M.cell_class = vcls.data(); M.v_kappa = vk.data(); M.v_kabs = vka.data(); M.v_albedo = val.data();
      M.v_lq = m->v_log_Qcool; M.v_cdf = m->v_kdB_dT_CDF;
      if (m->v_prob_s11_pos) {
        const int na1 = m->nang_scatt + 1, ncol = m->p_lambda_fixed ? 1 : nl;
        auto relay = [&](const float* src, int cols, std::vector<float>& t) {
          t.resize((size_t)nc * cols * na1);
          for (int c = 0; c < nc; ++c)
            for (int l = 0; l < cols; ++l)
              memcpy(&t[((size_t)c * cols + l) * na1], &src[((size_t)l * nc + c) * na1], na1 * sizeof(float));
          return t.data();
        };
        M.v_prob = relay(m->v_prob_s11_pos, ncol, vsc[0]); M.v_s12 = relay(m->v_s12_o_s11, nl, vsc[1]);
        M.v_s22 = relay(m->v_s22_o_s11, nl, vsc[2]); M.v_s33 = relay(m->v_s33_o_s11, nl, vsc[3]);
        M.v_s34 = relay(m->v_s34_o_s11, nl, vsc[4]); M.v_s44 = relay(m->v_s44_o_s11, nl, vsc[5]);
        vsc[6].resize((size_t)nc * nl);
        for (int c = 0; c < nc; ++c)
          for (int l = 0; l < nl; ++l) vsc[6][(size_t)c * nl + l] = m->v_tab_g_pos[c + (size_t)nc * l];
        M.v_g = vsc[6].data();
        M.v_scatt = 1;
        if (m->v_tab_s11_pos) M.v_s11 = relay(m->v_tab_s11_pos, nl, vsc[7]);   // (mcgpu_set_variable_dust_s11)
      }
    }
  }
};

