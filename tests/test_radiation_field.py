"""xN_abs / xJ_abs: the optional accumulators of save_radiation_field's thermal branch (radiation_field.f90:54-55;
xN_abs is what run_mcfost_phantom hands back, mcfost2phantom.f90:361)."""
import numpy as np
import pytest

from mcfost_amd.host import model as M
from oracle import Oracle


def test_oracle_accumulators_are_consistent_with_the_absorbed_energy():
    m = M.build_model(M.small())
    o = Oracle(m, 5000)
    r = o.run_thermal_with_radiation_field(5000, seed=3, n_threads=1)
    # one segment per crossing of a real cell; E_abs = sum_lambda kappa_abs(lambda) * xJ_abs(:, lambda)
    assert 0 < r["xN_abs"].sum() <= r["counters"]["crossings"]
    assert np.allclose((m.kappa_abs_LTE[:, None] * r["xJ_abs"]).sum(axis=0), r["E_abs"], rtol=1e-10)
    assert np.array_equal(r["xN_abs"] > 0, r["xJ_abs"].sum(axis=0) > 0)
    # off again afterwards
    r2 = o.run_thermal(5000, seed=3, n_threads=1)
    assert np.allclose(r2["E_abs"], r["E_abs"], rtol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("grid", ["cyl2d", "cyl3d", "voronoi"])
def test_device_accumulators_against_the_oracle(grid):
    from mcfost_amd.engine import Engine
    if grid == "voronoi":
        m = M.build_voronoi_model(M.small(), 3000, seed=2)
    elif grid == "cyl3d":
        m = M.build_model(M.small(n_rad=12, nz=6, n_az=8, l3D=True))
    else:
        m = M.build_model(M.small())
    n = 20000
    o = Oracle(m, n)
    prior = o.run_thermal(2000, seed=1)["E_abs"]
    want = o.run_thermal_with_radiation_field(n, seed=5, frozen=True, E_prior=prior, n_threads=8)
    e = Engine(m, n)
    e.set_option("radiation_field", 3)
    got = e.run_thermal(n, seed=5, frozen=True, E_prior=prior)
    xN, xJ = e.fetch_radiation_field()
    assert got["counters"] == want["counters"]
    assert np.array_equal(xN, want["xN_abs"])                      # integer counts: exact
    # (a bin of one wavelength of one cell may hold a single grazing segment, whose length carries the crossing's
    # FMA-level rounding relative to the cell size, not to itself)
    assert np.allclose(xJ, want["xJ_abs"], rtol=1e-7, atol=1e-9 * want["xJ_abs"].max())
    assert np.allclose(xJ.sum(axis=0), want["xJ_abs"].sum(axis=0), rtol=1e-9, atol=1e-12 * want["xJ_abs"].sum(axis=0).max())
    assert np.allclose(got["E_abs"], want["E_abs"], rtol=1e-9, atol=1e-11 * want["E_abs"].max())
    # accumulate: a second launch adds to them
    e.run_thermal(n, seed=5, frozen=True, E_prior=prior, accumulate=True)
    xN2, _ = e.fetch_radiation_field(xJ=False)
    assert np.array_equal(xN2, 2 * xN)
    # only what was asked for is kept
    e2 = Engine(m, n)
    e2.set_option("radiation_field", 1)
    e2.run_thermal(1000, seed=5, frozen=True, E_prior=prior)
    from mcfost_amd.engine import McgpuError
    with pytest.raises(McgpuError):
        e2.fetch_radiation_field(xN=False, xJ=True)
    e.close(); e2.close()
