"""Shared helpers for the parity tests."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

CONFIGS = {
    "ref41": lambda M: M.ref41(),
    "pascucci": lambda M: M.pascucci(),
    "small2d": lambda M: M.small(),
    "small3d": lambda M: M.small(n_rad=12, nz=6, n_az=8, l3D=True),
    "ref41_3d": lambda M: M.ref41_3d(),
    "sph2d": lambda M: M.small(grid_type=2),
    "sph3d": lambda M: M.small(n_rad=12, nz=6, n_az=8, l3D=True, grid_type=2),
}


def load_golden(name):
    return np.load(os.path.join(GOLDEN, f"geom_{name}.npz"))


def mc_similar(x, y, threshold, mask_threshold=0.0):
    """The reference's own Monte-Carlo-aware comparator
    (test_suite/test_mcfost.py:46-57): 75th percentile of |x-y|/x over pixels
    with |x| > mask must be below the threshold."""
    x = np.asarray(x, float).ravel()
    y = np.asarray(y, float).ravel()
    mask = np.abs(x) > mask_threshold
    p75 = np.percentile(np.abs(x[mask] - y[mask]) / x[mask], 75)
    return p75 < threshold, p75


def rel_rms(T, Tref, T_floor):
    sel = Tref > T_floor
    return float(np.sqrt(np.mean(((T[sel] - Tref[sel]) / Tref[sel]) ** 2)))


def sed_model(cfg, n_thermal=100000, voronoi_sites=0, seed=3):
    """A model whose SED-step emission tables (frac_E_stars, prob_E_cell) come from a thermal step of
    the CPU oracle, like run_sed_mc's come from the temperature step."""
    from mcfost_amd.host import model as M
    from oracle import Oracle
    m = M.build_voronoi_model(cfg, voronoi_sites, seed=seed) if voronoi_sites else M.build_model(cfg)
    orc = Oracle(m, n_thermal)
    T = orc.temp_finale(orc.run_thermal(n_thermal, seed=3, n_threads=1)["E_abs"])  # 1 thread: reproducible
    M.repartition_energie(m, T)
    m.extra["Tdust"] = T
    return m


def xI_close(xa, xb, rtol=1e-6, n_midplane_cells=0, atol_rel=1e-8):
    """xI_scatt of two runs of the same packets.  A path that crosses a midplane cell of a 2D grid from
    its upper wall to the mirrored lower wall has its midpoint at z = +-rounding, so whether its deposit
    counts as "above" or "below" (psup, radiation_field.f90:78-82) is rounding noise in any build of the
    algorithm: the comparison is made on the sum of the two.  Likewise a deposit may (rarely) fall into the
    neighbouring azimuth sub-bin when FMA-level drift moves a midpoint across a sub-bin edge."""
    scale = np.abs(xb).max()
    assert np.allclose(xa.sum(axis=(3, 4)), xb.sum(axis=(3, 4)), rtol=rtol, atol=atol_rel * scale)
    sa, sb = xa.sum(axis=3), xb.sum(axis=3)
    bad = np.abs(sa - sb) > rtol * np.abs(sb) + atol_rel * scale
    assert bad.sum() <= max(4, 2e-4 * np.count_nonzero(sb)), (bad.sum(), np.count_nonzero(sb))
    if n_midplane_cells:  # away from the midplane layer (cells 1..n_rad of a 2D grid) psup itself must agree
        ua, ub = xa[n_midplane_cells:], xb[n_midplane_cells:]
        bad2 = np.abs(ua - ub) > rtol * np.abs(ub) + atol_rel * scale
        assert bad2.sum() <= max(4, 2e-4 * np.count_nonzero(ub)), (bad2.sum(), np.count_nonzero(ub))
    return bad.sum()




class OracleBackend:
    """The CPU oracle behind mcfost_amd.host.pipeline.temperature_and_sed (checker side of the end-to-end test)."""

    def __init__(self, orc, n_threads=8):
        self.o, self.nt, self.sed_tables = orc, n_threads, False

    def run_thermal(self, n, seed):
        return self.o.run_thermal(n, seed=seed, n_threads=self.nt)

    def temp_finale(self, E_abs):
        return self.o.temp_finale(E_abs)

    def run_mono(self, lam, n2, seed, n_chunks):
        if not self.sed_tables:   # the oracle copies the emission tables when it is built: take the SED step's
            from oracle import Oracle
            self.o = Oracle(self.o.model, 1e5)   # (the packet count only scales L_packet_th: not used by the SED step)
            self.sed_tables = True
        return self.o.run_mono(lam, n2, seed=seed, n_chunks=n_chunks, rt1=True, n_threads=self.nt)

    def dust_map(self, lam, Tdust, res, E_disk):
        return self.o.dust_map_sed(lam, res["xI_scatt"], Tdust, res["n_sent"][lam - 1], E_disk, n_threads=self.nt)

    def stars_map(self, lam, star_flux, seed):
        return self.o.stars_map_sed(lam, star_flux, seed=seed)

    # --- run_image_mc (mcfost_amd.host.pipeline.image) ---
    def run_image_mc(self, lam, n_photons_image, seed, n_chunks, method):
        if not self.sed_tables:
            from oracle import Oracle
            self.o = Oracle(self.o.model, 1e5)
            self.sed_tables = True
        return self.o.run_mono(lam, 10 ** 12, seed=seed, n_chunks=n_chunks, n_phot_lim=float(n_photons_image), rt1=True,
                               rt2=(15, 15) if method == 2 else None, n_threads=self.nt)

    def dust_image(self, lam, Tdust, res, E_disk, npix_x, npix_y, map_size, zoom, ang_disque):
        return self.o.dust_map_image(lam, res["xI_scatt"], Tdust, res["n_sent"][lam - 1], E_disk, npix_x, npix_y, map_size,
                                     zoom=zoom, ang_disque=ang_disque, l_sym_ima=False, n_threads=self.nt)[0]

    def dust_image_method2(self, lam, ibin, Tdust, res, E_disk, npix_x, npix_y, map_size, zoom):
        ns = res["n_sent"][lam - 1]
        eps, eps_s = self.o.init_dust_source_fct2(lam, ibin, res["I_spec"], res["I_spec_star"], Tdust, ns, E_disk)
        return self.o.rt2_dust_map_image(lam, ibin, eps, eps_s, Tdust, ns, E_disk, npix_x, npix_y, map_size, zoom=zoom,
                                         n_threads=self.nt)[0]

    def stars_image(self, lam, star_flux, npix_x, npix_y, map_size, zoom, seed, ang_disque):
        return self.o.stars_map_image(lam, star_flux, npix_x, npix_y, map_size, zoom=zoom, seed=seed, ang_disque=ang_disque)[0]

    def tau_maps(self, lam, npix_x, npix_y, map_size, zoom, tau, ang_disque):
        return self.o.tau_maps(lam, npix_x, npix_y, map_size, zoom=zoom, tau=tau, ang_disque=ang_disque)
