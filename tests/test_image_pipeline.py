"""run_image_mc end to end (dust_transfer.f90:692-824) through mcfost_amd.host.pipeline.image: emission tables of the
wavelength, the image-mode packet loop, the dust's image with ray tracing method 1 (every observer) or 2 (per
inclination), the stars' discs, the optical-depth maps -- the host's call order as an executable description.
CPU: the sequence on the oracle alone (shapes, symmetry, where the star lands, the maps).  GPU: the same sequence on the
device against the oracle -- the same packets (counter-based streams, fixed packet counts), so the images agree pixel for
pixel to the source functions' rounding, but for the pixels whose refinement test sits on its 1 % threshold."""
import numpy as np
import pytest

from helpers import OracleBackend, sed_model
from mcfost_amd.host import model as M, pipeline as P
from oracle import Oracle


def _close(img, ref):
    img, ref = np.asarray(img, float), np.asarray(ref, float)
    top = np.abs(ref).max()
    off = ~np.isclose(img, ref, rtol=3e-5, atol=1e-6 * top)
    pixels = off.reshape(-1, *off.shape[-2:]).any(axis=0)
    assert pixels.sum() <= max(4, 0.01 * pixels.size), pixels.sum()
    assert np.allclose(img[off], ref[off], rtol=0.03, atol=1e-4 * top)


class _HostTables:
    """A backend without its own repartition_energie: pipeline.image then takes the host's tables (model.repartition_energie)."""

    def __init__(self, backend):
        self._b = backend

    def __getattr__(self, name):
        if name == "repartition_energie":
            raise AttributeError(name)
        return getattr(self._b, name)


def test_image_sequence_on_the_oracle():
    m = sed_model(M.small(RT_imin=0.0, RT_imax=60.0, RT_n_incl=2), n_thermal=20000)
    T = m.extra["Tdust"]
    lam = int(np.argmin(np.abs(m.lam - 1.0))) + 1
    npix, size = 15, 2.2 * m.cfg.rout
    r = P.image(OracleBackend(Oracle(m, 1000), n_threads=4), m, T, lam, 150, npix, npix, size, seed=3, n_chunks=4, tau_surface=1.0)
    nRT, ntf = 2, m.rt["N_type_flux"]
    assert r["image"].shape == (ntf, nRT, npix, npix) and r["stars"].shape == (nRT, npix, npix)
    assert r["n_sent"] == 4 * 150                                     # every stream sends exactly n_photons_image packets
    assert (r["image"][0] >= 0).all() and r["image"][0].sum() > 0
    # the star (at the origin, unresolved at this scale) sits in the centre pixel of every observer's map, thin path pole-on
    for q in range(nRT):
        assert r["stars"][q].argmax() == (npix // 2) * npix + npix // 2
    assert r["stars"][0].sum() > 0.5 * P.stars_flux_factor(m, lam).sum()
    # the dust's image of the pole-on observer is symmetric left-right and up-down to the Monte Carlo noise of xI_scatt's
    # azimuthal sub-bins; the maps belong to the same pixels: where the centre ray meets dust, the pixel shows dust
    dust = r["image"][0, 0] - r["stars"][0]
    assert abs(dust[:, : npix // 2].sum() / dust[:, npix // 2 + 1:].sum() - 1.0) < 0.5
    assert (dust[r["tau_map"][0] > 0.1] > 0).all() and (r["tau_map"][0] > 0.1).sum() > 20
    assert r["tau_surface_map"].shape == (3, nRT, npix, npix)
    # method 2 of the same Monte Carlo settings: the same star, a dust image of the same order
    r2 = P.image(OracleBackend(Oracle(m, 1000), n_threads=4), m, T, lam, 150, npix, npix, size, seed=3, n_chunks=4, method=2)
    assert np.allclose(r2["stars"], r["stars"]) and r2["n_sent"] == r["n_sent"]
    d2 = r2["image"][0, 0] - r2["stars"][0]
    assert 0.3 < d2.sum() / dust.sum() < 3.0


@pytest.mark.gpu
@pytest.mark.parametrize("kw,sites,method", [(dict(), 0, 1), (dict(), 0, 2), (dict(lsepar_pola=False, RT_n_az=2, RT_az_max=60.0), 0, 1),
                                             (dict(grid_type=2), 0, 1), (dict(grid_type=2), 0, 2), (dict(), 2500, 1)])
def test_image_sequence_on_the_device_equals_the_oracle(kw, sites, method):
    from mcfost_amd.engine import Engine
    m = sed_model(M.small(RT_imin=0.0, RT_imax=70.0, RT_n_incl=2, **kw), n_thermal=30000, voronoi_sites=sites)
    T = m.extra["Tdust"]
    lam = int(np.argmin(np.abs(m.lam - 1.0))) + 1
    npix, size = 20, 2.2 * m.cfg.rout
    e = Engine(m, 1e5)
    # (the emission tables of the wavelength: the device builds its own in the product's sequence, equal to the host's to
    # 2e-7 -- checked here -- but a cumulative distribution that differs in the seventh digit picks another cell for a packet
    # in 1e7; for a pixel-for-pixel comparison both sides take the host's tables)
    assert np.isclose(P.EngineBackend(e).repartition_energie(lam, T), m.extra["E_disk"][lam - 1], rtol=1e-6)
    got = P.image(_HostTables(P.EngineBackend(e)), m, T, lam, 400, npix, npix, size, zoom=1.1, seed=5, n_chunks=16, method=method,
                  ang_disque=0.0 if method == 2 else 20.0, tau_surface=1.0)
    e.close()
    want = P.image(OracleBackend(Oracle(m, 1000), n_threads=8), m, T, lam, 400, npix, npix, size, zoom=1.1, seed=5, n_chunks=16,
                   method=method, ang_disque=0.0 if method == 2 else 20.0, tau_surface=1.0)
    assert got["n_sent"] == want["n_sent"] == 16 * 400
    assert got["E_disk"] == want["E_disk"]
    assert np.array_equal(got["stars"] != 0, want["stars"] != 0)
    assert np.allclose(got["stars"], want["stars"], rtol=2e-5, atol=1e-6 * np.abs(want["stars"]).max())
    assert want["image"][0].max() > 0
    _close(got["image"], want["image"])   # (stars included)
    assert np.allclose(got["tau_map"], want["tau_map"], rtol=2e-6)
    assert np.array_equal((got["tau_surface_map"] != 0).any(axis=0), (want["tau_surface_map"] != 0).any(axis=0))
    assert np.allclose(got["tau_surface_map"], want["tau_surface_map"], rtol=2e-6, atol=1e-6 * np.abs(want["tau_surface_map"]).max())
