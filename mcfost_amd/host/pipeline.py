"""Host-side sequence of a "temperature + SED" run (BASELINE config 2) over the engine: what
``transfert_poussiere`` does around the packet loops (dust_transfer.f90:188-433) --

    run_thermal_mc      (:575-688)   one temperature iteration + Temp_finale
    repartition_energie (thermal_emission.f90:1771)   emission tables of the SED step from Tdust
    run_sed_mc          (:828-1042)  per wavelength: the monochromatic packet loop (SED bins + xI_scatt),
                                     then the ray-traced SED of the dust (dust_map, RT method 1)
    run_image_mc        (:692-824)   one wavelength: the image-mode packet loop (a fixed number of packets per stream),
                                     then per observer the source function, dust_map's pixels, the stars' discs and,
                                     if asked, the optical-depth maps (``image`` below)

The Fortran host keeps doing this itself; this mirror drives the same C-ABI calls from Python
for the tests, ``tools/run_config2.py`` and as an executable description of the call order.  ``backend`` is
anything with the Engine's methods (``EngineBackend`` below; the tests pass the CPU oracle through an adapter)."""
from __future__ import annotations

import time

import numpy as np

from . import model as M


class EngineBackend:
    """The MI355X engine (mcfost_amd.engine.Engine) behind the pipeline's four calls."""

    def __init__(self, engine):
        self.e = engine

    def run_thermal(self, n, seed):
        return self.e.run_thermal(n, seed=seed)

    def temp_finale(self, E_abs):
        return self.e.temp_finale(E_abs)

    def temp_approx_diffusion_vertical(self, Tdust, ri_in, ri_out, zj_sup):
        return self.e.temp_approx_diffusion_vertical(Tdust, ri_in, ri_out, zj_sup)[0]

    def repartition_energie(self, lam, Tdust):
        """repartition_energie(lam) on the device (mcgpu_repartition_energie): the wavelength's emission tables stay in
        HBM for the run_mono that follows; returns E_disk(lam)."""
        self._tables = (lam, self.e.repartition_energie(lam, Tdust, fetch=False))
        return self._tables[1]["E_disk"]

    def run_mono(self, lam, n2, seed, n_chunks):
        t = getattr(self, "_tables", None)
        return self.e.run_mono(lam, n2, seed=seed, n_chunks=n_chunks, fetch_xI=False,
                               device_tables=t[1] if t is not None and t[0] == lam else None)

    def dust_map(self, lam, Tdust, res, E_disk):
        return self.e.dust_map_sed(lam, Tdust, res["n_sent"][lam - 1], E_disk)[0]

    def stars_map(self, lam, star_flux, seed):
        return self.e.stars_map_sed(lam, star_flux, seed=seed)


    # --- run_image_mc ---
    def run_image_mc(self, lam, n_photons_image, seed, n_chunks, method):
        """mc_photon_loop with lmono0 (dust_transfer.f90:711-713): every stream sends exactly n_photons_image packets; the
        deposits of ray tracing method 1 (xI_scatt) or 2 (I_spec) stay on the device."""
        t = getattr(self, "_tables", None)
        return self.e.run_mono(lam, 10 ** 12, n_phot_lim=float(n_photons_image), seed=seed, n_chunks=n_chunks, fetch_xI=False,
                               device_tables=t[1] if t is not None and t[0] == lam else None,
                               rt2=(15, 15) if method == 2 else None)

    def dust_image(self, lam, Tdust, res, E_disk, npix_x, npix_y, map_size, zoom, ang_disque):
        return self.e.dust_map_image(lam, Tdust, res["n_sent"][lam - 1], E_disk, npix_x, npix_y, map_size, zoom=zoom,
                                     ang_disque=ang_disque)[0]

    def dust_image_method2(self, lam, ibin, Tdust, res, E_disk, npix_x, npix_y, map_size, zoom):
        ns = res["n_sent"][lam - 1]
        self.e.init_dust_source_fct2(lam, ibin, None, None, Tdust, ns, E_disk)
        return self.e.rt2_dust_map_image(lam, Tdust, ns, E_disk, npix_x, npix_y, map_size, zoom=zoom)[0]

    def stars_image(self, lam, star_flux, npix_x, npix_y, map_size, zoom, seed, ang_disque):
        return self.e.stars_map_image(lam, star_flux, npix_x, npix_y, map_size, zoom=zoom, seed=seed, ang_disque=ang_disque)[0]

    def tau_maps(self, lam, npix_x, npix_y, map_size, zoom, tau, ang_disque):
        return self.e.tau_maps(lam, npix_x, npix_y, map_size, zoom=zoom, tau=tau, ang_disque=ang_disque)[:2]


def temperature_and_sed(backend, m, n_thermal, n_photons_lambda, lambdas=None, seed=1, n_chunks=None,
                        ray_tracing=True, diff_approx=None):
    """Returns Tdust, the nine Monte Carlo SED arrays ``sed_mc`` (9, N_phi, N_thet, n_lambda), ``n_sent``
    (packets sent per wavelength in the SED step), ``sed_rt`` (n_lambda, nRT, N_type_flux; dust only) and the
    wall time of each stage."""
    t = {}
    t0 = time.perf_counter()
    th = backend.run_thermal(int(n_thermal), seed)
    Tdust = backend.temp_finale(th["E_abs"])
    t["thermal"] = time.perf_counter() - t0
    if diff_approx is not None:  # ldiff_approx with a dark zone: dust_transfer.f90:316 / :659
        t0 = time.perf_counter()
        Tdust = backend.temp_approx_diffusion_vertical(Tdust, *diff_approx)
        t["diffusion"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    on_device = hasattr(backend, "repartition_energie")   # the engine builds each wavelength's tables itself (:924)
    if not on_device:
        M.repartition_energie(m, Tdust)
    E_disk = {} if on_device else dict(enumerate(m.extra["E_disk"], start=1))
    t["repartition_energie"] = time.perf_counter() - t0
    nl = m.n_lambda
    lambdas = list(lambdas) if lambdas is not None else list(range(1, nl + 1))
    sed = np.zeros((9, m.cfg.N_phi, m.cfg.N_thet, nl))
    n_sent = np.zeros(nl)
    rt = m.rt
    sed_rt = np.zeros((nl, rt["RT_n_incl"] * rt["RT_n_az"], rt["N_type_flux"]))
    sed_rt_stars = np.zeros((nl, rt["RT_n_incl"] * rt["RT_n_az"]))
    t["sed_mc"] = t["ray_tracing"] = 0.0
    sed_crossings = 0.0   # cell crossings of the SED step's committed packets, all wavelengths (each one a set of deposits)
    for lam in lambdas:
        if on_device:
            t0 = time.perf_counter()
            E_disk[lam] = backend.repartition_energie(lam, Tdust)
            t["repartition_energie"] += time.perf_counter() - t0
        t0 = time.perf_counter()
        r = backend.run_mono(lam, int(n_photons_lambda), seed + lam, n_chunks)
        t["sed_mc"] += time.perf_counter() - t0
        sed[..., lam - 1] = r["sed"][..., lam - 1]
        n_sent[lam - 1] = r["n_sent"][lam - 1]
        sed_crossings += float(r["counters"]["crossings"]) if "counters" in r else 0.0
        if ray_tracing:
            t0 = time.perf_counter()
            sed_rt[lam - 1] = backend.dust_map(lam, Tdust, r, E_disk[lam])
            if hasattr(backend, "stars_map"):   # compute_stars_map (dust_transfer.f90:1583-1586): added to type 1 by the caller
                sed_rt_stars[lam - 1] = backend.stars_map(lam, stars_flux_factor(m, lam), seed + 7919 * lam)
            t["ray_tracing"] += time.perf_counter() - t0
    if on_device:   # what sed_flux() and the callers' ray-tracing calls read
        Ed = np.zeros(nl)
        for lam, v in E_disk.items():
            Ed[lam - 1] = v
        m.extra["E_disk"] = Ed
    return dict(Tdust=Tdust, sed_mc=sed, n_sent=n_sent, sed_rt=sed_rt, sed_rt_stars=sed_rt_stars, seconds=t,
                thermal_counters=th["counters"], E_disk=E_disk, sed_crossings=sed_crossings)


def stars_flux_factor(m, lam):
    """``factor * prob_E_star(lambda, :)`` of compute_stars_map (dust_transfer.f90:1657-1659, :1819): the flux an
    unobscured star sends to the observer, in the units of Stokes_ray_tracing."""
    pc_to_AU, AU_to_Rsun = 648000.0 / np.pi, 149597870700.0 / 6.957e8
    factor = float(m.E_stars[lam - 1]) * float(m.lam[lam - 1]) * 1.0e-6 / (m.cfg.distance * pc_to_AU * AU_to_Rsun) ** 2 * 1.35e-12
    cdf = np.asarray(m.CDF_E_star, float).reshape(-1, m.n_lambda)[:, lam - 1]     # (0:n_stars) of this wavelength
    return factor * np.diff(cdf)


def sed_flux(m, sed_mc, n_sent):
    """Monte Carlo SED per packet energy: sed(lambda) * E_totale(lambda) / n_phot_envoyes(lambda), the factor the
    reference applies when it writes sed2 (output.f90: ecriture_sed), without the unit constants."""
    E_tot = np.asarray(m.E_stars) + np.asarray(m.extra["E_disk"])
    with np.errstate(divide="ignore", invalid="ignore"):
        f = np.where(n_sent > 0, E_tot / n_sent, 0.0)
    return sed_mc * f


def image(backend, m, Tdust, lam, n_photons_image, npix_x, npix_y, map_size, zoom=1.0, seed=1, n_chunks=None, method=1,
          ang_disque=0.0, tau_surface=None):
    """One wavelength of ``run_image_mc`` (dust_transfer.f90:692-824): ``repartition_energie(lambda)`` (:737), the image-mode
    packet loop (:748; ``n_photons_image`` packets per stream), then per observer (:775-801) the source function and
    ``dust_map``'s pixels -- ``init_dust_source_fct1`` + RT method 1 for every (ibin, iaz), or ``init_dust_source_fct2`` +
    method 2 per inclination (iaz = 1, 2D) -- with ``compute_stars_map``'s discs added to Stokes I (``dust_map`` :1582-1590)
    and, with ``tau_surface``, the maps of ``compute_tau_map`` / ``compute_tau_surface_map``.
    Returns ``image`` (N_type_flux, nRT, npix_y, npix_x; q = ibin - 1 + RT_n_incl (iaz - 1)), ``stars`` (nRT, npix_y,
    npix_x), ``n_sent`` and, if asked, ``tau_map`` / ``tau_surface_map``."""
    rt = m.rt
    nRT, n_incl, ntf = rt["RT_n_incl"] * rt["RT_n_az"], rt["RT_n_incl"], rt["N_type_flux"]
    if hasattr(backend, "repartition_energie"):
        E_disk = backend.repartition_energie(lam, Tdust)
    else:
        M.repartition_energie(m, Tdust)
        E_disk = float(m.extra["E_disk"][lam - 1])
    r = backend.run_image_mc(lam, int(n_photons_image), seed, n_chunks, method)
    img = np.zeros((ntf, nRT, npix_y, npix_x))
    if method == 1:
        d = np.asarray(backend.dust_image(lam, Tdust, r, E_disk, npix_x, npix_y, map_size, zoom, ang_disque))
        img[:] = d.reshape(ntf, nRT, npix_y, npix_x)           # (N_type_flux, RT_n_az, RT_n_incl, ...): q = ibin-1 + n_incl (iaz-1)
    else:
        if rt["RT_n_az"] != 1 or m.cfg.l3D:
            raise ValueError("ray tracing method 2: 2D grids, one observer azimuth (dust_transfer.f90:789-791)")
        for ibin in range(1, n_incl + 1):
            img[:, ibin - 1] = np.asarray(backend.dust_image_method2(lam, ibin, Tdust, r, E_disk, npix_x, npix_y, map_size, zoom)).reshape(ntf, npix_y, npix_x)
    stars = np.asarray(backend.stars_image(lam, stars_flux_factor(m, lam), npix_x, npix_y, map_size, zoom, seed + 7919 * lam,
                                           ang_disque))[:, 0]
    img[0] += stars
    out = dict(image=img, stars=stars, n_sent=r["n_sent"][lam - 1], E_disk=E_disk)
    if tau_surface is not None:
        out["tau_map"], out["tau_surface_map"] = backend.tau_maps(lam, npix_x, npix_y, map_size, zoom, float(tau_surface), ang_disque)
    return out

