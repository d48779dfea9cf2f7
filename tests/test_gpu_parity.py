"""Parity tests proper: the HIP engine, called through the C-ABI
(include/mcgpu.h), against the CPU oracle on the same seeded inputs, against
the committed golden vectors of the reference's own geometry routines, and --
at BASELINE sizes -- through size-independent properties.

Tolerances (stated here, used below):
  * integer / index outputs (cell ids, SED packet counts, n_sent, event
    counters): bit-exact;
  * FP64 positions and path lengths of ONE operator call: |diff| <= 1e-12 *
    (|x|+|y|+|z|+l)  (the device contracts a*b+c into FMA, the reference
    build does not);
  * absorbed energy per cell, frozen-temperature mode (same packets, same
    random numbers, different summation order and FMA): rtol 1e-9 plus 1e-11 of
    the largest cell -- a packet that ends a radial crossing within default-real
    rounding of a layer's wall (cylindrical_grid.f90:1116 computes zj through
    default real) is booked in one layer or the other by the last place of z1,
    and the sliver it then crosses (1e-4 of a deposit) moves between the two
    vertically adjacent cells: 1e-12 ... 4e-12 of the largest cell, about once
    per 1e7 crossings, every counter equal (round 4 measured it with the lane
    emulation; the old bound of 1e-12 held or not with the random sample);
  * live Bjorkman & Wood mode is not bit-reproducible by construction (the
    reference's own threads race the same way): statistical gate = relative
    RMS of Tdust over cells with T > 1.01 T_min  <= 3 * sigma_MC(N) and the
    reference's own gate p75(|dT|/T) < 5 % (test_suite/test_mcfost.py:88).
"""
import numpy as np
import pytest

from helpers import CONFIGS, load_golden, mc_similar, rel_rms, sed_model
from mcfost_amd.host import model as M

pytestmark = pytest.mark.gpu


def _engine(model, n_tot):
    from mcfost_amd.engine import Engine
    return Engine(model, n_tot)


def _oracle(model, n_tot):
    from oracle import Oracle
    return Oracle(model, n_tot)


@pytest.fixture(scope="module")
def small_pair(small_model):
    return small_model, _engine(small_model, 2e4), _oracle(small_model, 2e4)


def test_native_library_is_loaded(small_pair):
    import mcfost_amd.engine as eng
    assert eng._lib is not None and "libmcfost_hip.so" in eng.LIB_PATH
    maps = open("/proc/self/maps").read()
    assert "libmcfost_hip.so" in maps


def test_philox_on_device(small_pair):
    m, e, o = small_pair
    assert e.probe_philox([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert e.probe_philox([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert e.probe_philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]
    for seed, pk in ((1, 0), (269753, 12345678901), (2 ** 40 + 3, 2 ** 33 + 5)):
        dev = e.probe_packet_rand(seed, pk, 23)
        ref = np.array([o.packet_rand(seed, pk, n) for n in range(23)], np.float32)
        assert np.array_equal(dev, ref)


@pytest.mark.parametrize("name", list(CONFIGS))
def test_device_geometry_against_reference_golden(name):
    """cross_cylindrical_cell / index_cell_cyl evaluated by the device code on
    the reference's golden walks."""
    cfg = CONFIGS[name](M)
    m = M.build_model(cfg)
    m.midplane_snap = 0  # reference-literal arithmetic: what the golden vectors hold
    e = _engine(m, 1e5)
    g = load_golden(name)
    wk = g["walk"]
    x1, y1, z1, nxt, l = e.probe_cross_cell(wk[:, 0], wk[:, 1], wk[:, 2], wk[:, 3], wk[:, 4], wk[:, 5],
                                            wk[:, 6].astype(np.int32))
    assert np.array_equal(nxt, wk[:, 10].astype(np.int32))
    scale = np.abs(wk[:, 0]) + np.abs(wk[:, 1]) + np.abs(wk[:, 2]) + wk[:, 11]
    for a, col in ((x1, 7), (y1, 8), (z1, 9), (l, 11)):
        assert np.all(np.abs(a - wk[:, col]) <= 1e-12 * scale), col
    assert np.array_equal(e.probe_index_cell(g["pos_x"], g["pos_y"], g["pos_z"]), g["index_icell"])
    assert np.array_equal(e.probe_index_cell(g["idx2_x"], g["idx2_y"], g["idx2_z"]), g["idx2_icell"])
    e.close()


def test_spherical_grid_on_the_gpu():
    """spherical_grid.f90 on the device: the probes against the reference's golden walks (next cell exact, end points
    and lengths to 1e-12), then the packet loop against the oracle (2D: same packets; 3D: hemispheres summed, see
    tests/test_kernel_emulation.py::_check_spherical), HBM and LDS deposits."""
    from test_kernel_emulation import _check_spherical
    for name in ("sph2d", "sph3d"):
        m = M.build_model(CONFIGS[name](M))
        e = _engine(m, 20000)
        g = load_golden(name)
        wk = g["walk"]
        x1, y1, z1, nxt, l = e.probe_cross_cell(wk[:, 0], wk[:, 1], wk[:, 2], wk[:, 3], wk[:, 4], wk[:, 5],
                                                wk[:, 6].astype(np.int32))
        assert np.array_equal(nxt, wk[:, 10].astype(np.int32))
        scale = np.abs(wk[:, 0]) + np.abs(wk[:, 1]) + np.abs(wk[:, 2]) + wk[:, 11]
        for a, col in ((x1, 7), (y1, 8), (z1, 9), (l, 11)):
            assert np.all(np.abs(a - wk[:, col]) <= 1e-12 * scale), col
        assert np.array_equal(e.probe_index_cell(g["pos_x"], g["pos_y"], g["pos_z"]), g["index_icell"])
        assert np.array_equal(e.probe_index_cell(g["idx2_x"], g["idx2_y"], g["idx2_z"]), g["idx2_icell"])
        o = _oracle(m, 20000)
        prior = o.run_thermal(2000, seed=1)["E_abs"]
        _check_spherical(e.run_thermal(20000, seed=31, frozen=True, E_prior=prior), o, m, 20000, 31, prior)
        e.set_option("deposit", 1)
        _check_spherical(e.run_thermal(20000, seed=32, frozen=True, E_prior=prior), o, m, 20000, 32, prior)
        live = e.run_thermal(50000, seed=5)
        assert live["counters"]["escaped"] + live["counters"]["killed_star"] == 50000
        assert e.temp_finale(live["E_abs"]).max() > 50.0
        e.close()


def test_spherical_grid_dark_zone_and_dust_classes_on_the_gpu():
    """Round 5: the spherical grid's temperature step takes a dark zone (l_dark_zone flags: the mirror of
    optical_depth.f90:104-112) or dust classes (lvariable_dust) -- k_thermal_sph_ext --, HBM and LDS deposits; bars as in
    tests/test_kernel_emulation.py::test_emulated_spherical_grid_with_dark_zone_and_dust_classes.  Both at once, the
    random walk with either, and the SED mode / ray tracer on such a context are refused with a message."""
    from test_kernel_emulation import _check_spherical, _check_spherical_loose
    from mcfost_amd.engine import McgpuError
    for kw in (dict(), dict(n_rad=12, nz=6, n_az=8, l3D=True)):
        m = M.build_model(M.small(grid_type=2, **kw))
        M.init_variable_dust(m)
        n = 20000
        e, o = _engine(m, n), _oracle(m, n)
        prior = o.run_thermal(2000, seed=1)["E_abs"]
        _check_spherical(e.run_thermal(n, seed=19, frozen=True, E_prior=prior), o, m, n, 19, prior)
        e.set_option("deposit", 1)
        _check_spherical(e.run_thermal(n, seed=20, frozen=True, E_prior=prior), o, m, n, 20, prior)
        live = e.run_thermal(50000, seed=5)
        assert live["counters"]["escaped"] + live["counters"]["killed_star"] == 50000
        e.close()
        md = M.build_model(M.small(grid_type=2, **kw))
        # (flagged cells must not touch the central hole: a packet mirrored at their wall interacts in the cell it came from)
        md.l_dark_zone = ((md.kappa_factor > np.percentile(md.kappa_factor, 85)) & (md.grid["cell_map_i"][:md.n_cells] >= 3)).astype(np.uint8)
        e, o = _engine(md, n), _oracle(md, n)
        prior = o.run_thermal(2000, seed=1)["E_abs"]
        b = o.run_thermal(n, seed=17, frozen=True, E_prior=prior, n_threads=8)
        assert b["counters"]["dark_mirrors"] > 1000
        tol = 0.04 if kw.get("l3D") else 0.01
        a = e.run_thermal(n, seed=17, frozen=True, E_prior=prior)
        _check_spherical_loose(dict(a, counters=list(a["counters"].values())), b, n, tol)
        e.set_option("deposit", 1)
        a = e.run_thermal(n, seed=17, frozen=True, E_prior=prior)
        _check_spherical_loose(dict(a, counters=list(a["counters"].values())), b, n, tol)
        e.close()
    m = M.build_model(M.small(grid_type=2))
    m.l_dark_zone = ((m.kappa_factor > np.percentile(m.kappa_factor, 85)) & (m.grid["cell_map_i"][:m.n_cells] >= 3)).astype(np.uint8)
    M.init_variable_dust(m)
    e = _engine(m, 1000)
    with pytest.raises(McgpuError, match="not both"):
        e.run_thermal(1000, seed=1)
    e.close()


def _frozen_parity(m, n, seed, n_prior=2000, rtol=1e-9, **kw):
    e, o = _engine(m, n), _oracle(m, n)
    prior = o.run_thermal(n_prior, seed=1)["E_abs"]
    a = e.run_thermal(n, seed=seed, frozen=True, E_prior=prior, **kw)
    b = o.run_thermal(n, seed=seed, frozen=True, E_prior=prior, n_threads=8)
    assert a["counters"] == b["counters"]
    assert np.array_equal(a["n_sent"], b["n_sent"])
    assert np.array_equal(a["sed"][4], b["sed"][4])   # packets per (lambda, inclination) bin: exact
    for t in (0, 5, 6, 7, 8):
        # Stokes I: exactly the packet count when Stokes are not tracked; with lsepar_pola
        # update_Stokes renormalises I by M11*S1_0/S(1) (scattering.f90:1294): 1 +- ulp per packet
        if m.cfg.lsepar_pola and m.cfg.aniso_method == 1:
            assert np.allclose(a["sed"][t], b["sed"][t], rtol=1e-12, atol=1e-9), t
        else:
            assert np.array_equal(a["sed"][t], b["sed"][t]), t
    assert np.allclose(a["sed"][1:4], b["sed"][1:4], rtol=1e-5, atol=1e-5 * max(1.0, np.abs(b["sed"][0]).max()))
    assert np.allclose(a["E_abs"], b["E_abs"], rtol=rtol, atol=1e-11 * b["E_abs"].max())
    Ta, Tb = e.temp_finale(a["E_abs"]), o.temp_finale(b["E_abs"])
    assert np.allclose(Ta, Tb, rtol=2e-6)
    e.close()
    return a, b


def test_frozen_parity_small_2d(small_model):
    _frozen_parity(small_model, 20000, seed=7)


def test_frozen_parity_small_3d():
    m = M.build_model(M.small(n_rad=12, nz=6, n_az=8, l3D=True))
    _frozen_parity(m, 20000, seed=8)


def test_frozen_parity_unpolarised_isotropic_hg():
    """Pascucci-style isotropic scattering, HG branch and no Stokes update."""
    m = M.build_model(M.small(lisotropic=True, lsepar_pola=False))
    _frozen_parity(m, 10000, seed=9)
    m = M.build_model(M.small(aniso_method=2, lsepar_pola=False))
    _frozen_parity(m, 10000, seed=10)


def test_frozen_parity_per_wavelength_phase_function(small_model):
    import copy
    m = copy.copy(small_model)
    m.p_lambda_fixed = 0
    _frozen_parity(m, 10000, seed=11)


def test_frozen_parity_dark_zone(small_model):
    """Dark-zone mirror (optical_depth.f90:104-112) with synthetic flags on the
    densest midplane cells."""
    import copy
    m = copy.copy(small_model)
    dz = np.zeros(m.n_cells, np.uint8)
    kf = m.kappa_factor.reshape(m.cfg.nz, m.cfg.n_rad)
    flags = dz.reshape(m.cfg.nz, m.cfg.n_rad)
    flags[0:2, 4:12] = 1
    assert kf[0, 4] > 0
    m.l_dark_zone = dz
    a, b = _frozen_parity(m, 20000, seed=12)
    assert a["counters"]["dark_mirrors"] > 0
    assert np.all(a["E_abs"][dz == 1] == 0)


def test_frozen_parity_disk_emission(small_model):
    """emit_packet's disk branch: select_cellule + pos_em_cell + isotropic
    direction (dust_transfer.f90:1121-1142)."""
    import copy
    m = copy.copy(small_model)
    rng = np.random.default_rng(0)
    E_cell = rng.random((m.n_lambda, m.n_cells)) * m.kappa_factor[None, :]
    pe = np.zeros((m.n_lambda, m.n_cells + 1))
    pe[:, 1:] = np.cumsum(E_cell, axis=1)
    pe /= pe[:, -1:]
    m.prob_E_cell = pe.reshape(-1)
    m.frac_E_stars = np.full(m.n_lambda, 0.4)
    # packets born in the thick midplane random-walk for >1e4 flights: FMA/libm-level rounding
    # differences accumulate along such walks (continuous; the event counters stay identical)
    _frozen_parity(m, 10000, seed=13, rtol=1e-5)


def test_frozen_parity_ref41_full_grid(ref41_model):
    """BASELINE config 2 grid (100 x 70, 50 wavelengths) at an oracle-sized
    packet count."""
    _frozen_parity(ref41_model, 100000, seed=14, n_prior=20000)


def test_independent_of_launch_geometry(small_model):
    """Counter-based per-packet streams: the result must not depend on how
    many workgroups/lanes share the work (frozen mode)."""
    e, o = _engine(small_model, 1e4), _oracle(small_model, 1e4)
    prior = o.run_thermal(2000, seed=1)["E_abs"]
    a = e.run_thermal(10000, seed=3, frozen=True, E_prior=prior, grid_blocks=1, block_threads=64)
    b = e.run_thermal(10000, seed=3, frozen=True, E_prior=prior, grid_blocks=64, block_threads=256)
    assert a["counters"] == b["counters"] and np.array_equal(a["sed"][4], b["sed"][4])
    assert np.allclose(a["E_abs"], b["E_abs"], rtol=1e-10, atol=0)
    e.close()


def test_accumulate_is_linear(small_model):
    """Two launches over adjacent packet ranges, accumulated on the device,
    equal one launch over the union (linearity of the estimator)."""
    e, o = _engine(small_model, 1e4), _oracle(small_model, 1e4)
    prior = o.run_thermal(2000, seed=1)["E_abs"]
    whole = e.run_thermal(9000, seed=5, frozen=True, E_prior=prior)
    e.run_thermal(4000, seed=5, frozen=True, first_packet=0)
    parts = e.run_thermal(5000, seed=5, frozen=True, first_packet=4000, accumulate=True)
    assert parts["counters"] == whole["counters"] and np.array_equal(parts["sed"][4], whole["sed"][4])
    assert np.allclose(parts["E_abs"], whole["E_abs"], rtol=1e-10, atol=0)
    e.close()


@pytest.mark.parametrize("n", [1_000_000, 8_000_000])
def test_live_mode_statistical_parity_ref41(ref41_model, n):
    """The reference algorithm proper (live immediate re-emission), at two packet counts: the event counts of device and
    oracle differ by an early-estimate effect of the in-flight temperature (the oracle scales a per-thread partial sum by
    nb_proc, thermal_emission.f90:670; the device folds a global sum) that must SHRINK with N: the gate does."""
    m = ref41_model
    e, o = _engine(m, n), _oracle(m, n)
    a = e.run_thermal(n, seed=21)
    b = o.run_thermal(n, seed=22, n_threads=8)         # independent noise realisation
    ca, cb = a["counters"], b["counters"]
    assert ca["packets"] == n and ca["escaped"] + ca["killed_star"] == n
    # measured (tests/devtools/live_counts.py): device / oracle = 1.067 at 1e6 packets, 1.028 at 8e6; device seed to seed
    # 0.3 %, the 8-thread oracle seed to seed 3 % (its per-thread sums are the noisier estimate): the bound is the
    # effect's size at N, max(3 %, 7 % sqrt(1e6 / N)), plus that scatter of the oracle
    tol = max(0.03, 0.07 * np.sqrt(1.0e6 / n)) + 0.03
    for k in ("crossings", "flights", "scatterings", "absorptions"):
        assert abs(ca[k] / cb[k] - 1) < tol, (k, ca[k] / cb[k], tol)
    Ta, Tb = e.temp_finale(a["E_abs"]), o.temp_finale(b["E_abs"])
    T_floor = 1.01 * m.cfg.T_min
    # sigma_MC(N) ~ 1.7 % sqrt(1.28e5/N) for one run; two independent runs -> sqrt(2)
    sigma = 0.017 * np.sqrt(1.28e5 / n) * np.sqrt(2.0)
    rms = rel_rms(Ta, Tb, T_floor)
    ok, p75 = mc_similar(Tb, Ta, 0.05, mask_threshold=T_floor)
    assert ok, p75
    assert rms <= 3 * sigma, (rms, sigma)
    # thermal SED: the reference's SED gate is p75 < 10 % (test_mcfost.py:104-109)
    sa, sb = a["sed"][0].sum(axis=(0, 1)), b["sed"][0].sum(axis=(0, 1))
    okS, p75S = mc_similar(sb, sa, 0.10, mask_threshold=200.0)
    assert okS, p75S
    e.close()


def test_full_size_properties_ref41(ref41_model):
    """BASELINE config 2 at a GPU-sized packet count (1e7): properties that do
    not need the oracle."""
    m = ref41_model
    n = 10_000_000
    e = _engine(m, n)
    a = e.run_thermal(n, seed=31)
    c = a["counters"]
    assert c["packets"] == n and c["escaped"] + c["killed_star"] == n          # energy conservation
    assert a["n_sent"].sum() == n and a["sed"][4].sum() == c["escaped"]
    assert c["flights"] == c["scatterings"] + c["absorptions"] + c["escaped"] + c["killed_star"]
    assert np.allclose(a["sed"][0], a["sed"][5:9].sum(axis=0))
    # emitted wavelengths follow the stellar CDF
    cdf = np.cumsum(a["n_sent"]) / n
    assert np.max(np.abs(cdf - m.spectre_emission_cumul[1:])) < 1e-3
    # two different seeds agree within Monte Carlo noise
    b = e.run_thermal(n, seed=32)
    Ta, Tb = e.temp_finale(a["E_abs"]), e.temp_finale(b["E_abs"])
    sigma = 0.017 * np.sqrt(1.28e5 / n) * np.sqrt(2.0)
    assert rel_rms(Ta, Tb, 1.01 * m.cfg.T_min) <= 3 * sigma
    # device Temp_finale on the device accumulator == on the fetched array
    assert np.array_equal(e.temp_finale(), Tb)
    e.close()


@pytest.mark.parametrize("name", ["pascucci", "ref41", "ref41_3d"])
def test_properties_at_the_benchmark_size(name):
    """The size bench.py runs (1e8 packets per step: BASELINE's headline, configs[1] and the 3D grid): the properties that do
    not need the oracle -- every packet accounted for, the emitted wavelengths on the stellar distribution to 3 sigma of 1e8
    draws, Stokes I the sum of its origins, two seeds inside the Monte Carlo noise, a frozen-temperature rerun of the
    same seed bit-equal in every integer and (2D: the LDS-private grid folds in launch order; 3D: the log folds in block
    order) to rounding in the energies."""
    cfg = {"pascucci": M.pascucci, "ref41": M.ref41, "ref41_3d": M.ref41_3d}[name]()
    m = M.build_model(cfg)
    n = 100_000_000
    e = _engine(m, n)
    a = e.run_thermal(n, seed=41)
    c = a["counters"]
    assert c["packets"] == n and c["escaped"] + c["killed_star"] == n
    assert a["n_sent"].sum() == n and a["sed"][4].sum() == c["escaped"]
    assert c["flights"] == c["scatterings"] + c["absorptions"] + c["escaped"] + c["killed_star"]
    assert np.allclose(a["sed"][0], a["sed"][5:9].sum(axis=0))
    cdf = np.cumsum(a["n_sent"]) / n
    assert np.max(np.abs(cdf - m.spectre_emission_cumul[1:])) < 3.0 * 0.5 / np.sqrt(n) + 1e-12
    b = e.run_thermal(n, seed=42)
    Ta, Tb = e.temp_finale(a["E_abs"]), e.temp_finale(b["E_abs"])
    sigma = 0.017 * np.sqrt(1.28e5 / n) * np.sqrt(2.0) * np.sqrt(max(m.n_cells, 7000) / 7000.0)
    assert rel_rms(Ta, Tb, 1.01 * m.cfg.T_min) <= 3 * sigma
    f1 = e.run_thermal(n, seed=43, frozen=True, E_prior=a["E_abs"])
    f2 = e.run_thermal(n, seed=43, frozen=True, E_prior=a["E_abs"])
    assert f1["counters"] == f2["counters"] and np.array_equal(f1["n_sent"], f2["n_sent"]) and np.array_equal(f1["sed"][4], f2["sed"][4])
    assert np.allclose(f1["E_abs"], f2["E_abs"], rtol=1e-9, atol=1e-12 * f1["E_abs"].max())
    e.close()


def test_device_accumulator_is_visible_to_torch(small_model):
    import torch
    e = _engine(small_model, 1e4)
    res = e.run_thermal(5000, seed=2)
    acc, cnt = e.device_accumulators()
    assert acc.is_cuda and acc.dtype == torch.float64
    n_c = small_model.n_cells
    assert np.array_equal(acc[:n_c].cpu().numpy(), res["E_abs"])
    assert int(cnt[0].item()) == 5000
    e.close()


def test_fortran_host_through_iso_c_binding(small_model, tmp_path):
    """The drop-in boundary as the reference's host would use it: a Fortran program
    (mcfost_amd/fortran/thermal_host_example.f90) hands the module arrays to the
    ISO_C_BINDING shim (mcgpu_f.f90) which calls the C-ABI."""
    import os
    import subprocess
    from mcfost_amd.host import dump
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "mcfost_amd", "fortran", "build", "thermal_host_example")
    if not os.path.exists(exe):
        pytest.skip("Fortran host example not built (amdflang missing at build time)")
    n, seed = 200000, 4242
    fin, fout, fpr = str(tmp_path / "model.bin"), str(tmp_path / "result.bin"), str(tmp_path / "prior.bin")
    dump.write_model(small_model, n, fin)
    o = _oracle(small_model, n)
    prior = o.run_thermal(2000, seed=1)["E_abs"]
    np.ascontiguousarray(prior, np.float64).tofile(fpr)
    # reproducible (frozen-temperature) mode through the multi-device entry with one device: the Fortran-driven run
    # equals the oracle packet for packet
    out = subprocess.run([exe, fin, fout, str(n), str(seed), "1", fpr], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "packets/s" in out.stdout
    f = dump.read_result(small_model, fout)
    b = o.run_thermal(n, seed=seed, frozen=True, E_prior=prior, n_threads=8)
    assert np.array_equal(f["n_sent"], b["n_sent"]) and np.array_equal(f["sed"][4], b["sed"][4])
    assert np.allclose(f["E_abs"], b["E_abs"], rtol=1e-9, atol=1e-11 * b["E_abs"].max())
    # live mode (the reference algorithm) still runs through the same binding
    out = subprocess.run([exe, fin, fout, str(n), str(seed)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    f = dump.read_result(small_model, fout)
    assert f["n_sent"].sum() == n and f["E_abs"].sum() > 0


def test_multi_device_entry_with_one_device_equals_the_single_context_call(small_model):
    """mcgpu_multi_run_thermal (one host thread, RCCL all-reduce of the fused [E_abs | sed | n_sent | counters] buffer)
    with n_dev = 1 returns what mcgpu_run_thermal returns."""
    from mcfost_amd.engine import MultiEngine
    n = 30000
    o = _oracle(small_model, n)
    prior = o.run_thermal(2000, seed=1)["E_abs"]
    e = _engine(small_model, n)
    a = e.run_thermal(n, seed=11, frozen=True, E_prior=prior)
    e.close()
    me = MultiEngine(small_model, n, devices=(0,))
    b = me.run_thermal(n, seed=11, frozen=True, E_prior=prior)
    me.close()
    assert a["counters"] == b["counters"] and b["counters"]["packets"] == n
    assert np.array_equal(a["n_sent"], b["n_sent"]) and np.array_equal(a["sed"][4], b["sed"][4])
    assert np.allclose(a["E_abs"], b["E_abs"], rtol=1e-12, atol=0)


def test_multi_device_entry_accumulates_and_runs_the_sed_step(small_model):
    """mcgpu_multi_*: (a) two accumulating calls equal one call on the union of their packets (on several devices each
    call first scales the all-reduced totals every device holds by 1 / n_dev, see include/mcgpu.h; on one device
    nothing is scaled and no communicator is opened); (b) mcgpu_multi_run_mono with one device returns what
    mcgpu_run_mono returns."""
    from mcfost_amd.engine import MultiEngine
    n = 20000
    o = _oracle(small_model, 2 * n)
    prior = o.run_thermal(2000, seed=1)["E_abs"]
    me = MultiEngine(small_model, 2 * n, devices=(0,))
    me.run_thermal(n, seed=11, first_packet=0, frozen=True, E_prior=prior)
    ab = me.run_thermal(n, seed=11, first_packet=n, frozen=True, accumulate=True)
    assert me.rccl_ranks() == 0          # one device: no RCCL communicator was ever needed
    me.close()
    e = _engine(small_model, 2 * n)
    u = e.run_thermal(2 * n, seed=11, frozen=True, E_prior=prior)
    e.close()
    assert ab["counters"] == u["counters"] and u["counters"]["packets"] == 2 * n
    assert np.array_equal(ab["n_sent"], u["n_sent"]) and np.array_equal(ab["sed"][4], u["sed"][4])
    # (two launches against one: which packets the tail kernel finishes depends on the launch, and the two kernels'
    # copies of the crossing differ in their contraction of multiply-adds -- the layer slivers of the module docstring)
    assert np.allclose(ab["E_abs"], u["E_abs"], rtol=1e-9, atol=1e-11 * u["E_abs"].max())

    m = sed_model(M.small(RT_n_incl=3))
    e = _engine(m, 1e5)
    a = e.run_mono(5, 40, seed=3, n_chunks=8)
    e.close()
    me = MultiEngine(m, 1e5, devices=(0,))
    b = me.run_mono(5, 40, seed=3, n_chunks=8)
    me.close()
    assert np.array_equal(a["n_sent_chunk"], b["n_sent_chunk"]) and a["counters"] == b["counters"]
    assert np.array_equal(a["sed"][4], b["sed"][4])
    assert np.allclose(a["xI_scatt"], b["xI_scatt"], rtol=1e-12, atol=0)


def test_single_role_schedule_option(small_model):
    """option "schedule" = 1 (the single-role kernel) and "deposit" = 1 (HBM atomics): same packets, same sums."""
    m = small_model
    o = _oracle(m, 20000)
    prior = o.run_thermal(2000, seed=1)["E_abs"]
    ref = o.run_thermal(20000, seed=21, frozen=True, E_prior=prior, n_threads=8)
    for opts in ({"schedule": 1}, {"deposit": 1}, {"schedule": 1, "deposit": 1}):
        e = _engine(m, 20000)
        for k, v in opts.items():
            e.set_option(k, v)
        a = e.run_thermal(20000, seed=21, frozen=True, E_prior=prior)
        assert a["counters"] == ref["counters"], opts
        assert np.array_equal(a["n_sent"], ref["n_sent"])
        assert np.allclose(a["E_abs"], ref["E_abs"], rtol=1e-9, atol=1e-11 * ref["E_abs"].max())
        e.close()


def test_edge_cases_empty_and_tiny_runs(small_model):
    """n_packets = 0, 1 and a non-multiple of the wave / batch sizes."""
    e, o = _engine(small_model, 1000), _oracle(small_model, 1000)
    prior = o.run_thermal(500, seed=1)["E_abs"]
    z = e.run_thermal(0, seed=1)
    assert z["counters"]["packets"] == 0 and z["E_abs"].sum() == 0 and z["sed"].sum() == 0
    for n in (1, 63, 65, 129, 1000):
        a = e.run_thermal(n, seed=5, frozen=True, E_prior=prior, first_packet=17)
        b = o.run_thermal(n, seed=5, frozen=True, E_prior=prior, first_packet=17)
        assert a["counters"] == b["counters"], n
        assert np.allclose(a["E_abs"], b["E_abs"], rtol=1e-9, atol=1e-12 * max(b["E_abs"].max(), 1e-300))
    e.close()


def test_packet_ids_beyond_32_bits(small_model):
    """Config 3 runs 1e9 packets over 8 GPUs: ids are 64-bit end to end."""
    e, o = _engine(small_model, 1000), _oracle(small_model, 1000)
    prior = o.run_thermal(500, seed=1)["E_abs"]
    first = (1 << 33) + 12345
    a = e.run_thermal(500, seed=9, frozen=True, E_prior=prior, first_packet=first)
    b = o.run_thermal(500, seed=9, frozen=True, E_prior=prior, first_packet=first)
    c = o.run_thermal(500, seed=9, frozen=True, E_prior=prior, first_packet=12345)
    assert a["counters"] == b["counters"] and np.array_equal(a["n_sent"], b["n_sent"])
    assert not np.array_equal(b["n_sent"], c["n_sent"])      # the high word matters
    e.close()


def test_single_wavelength_and_coarse_grids():
    """Ragged / minimal table shapes: 2 wavelengths, 3 x 2 cells, 3D with 2 azimuths."""
    for cfg in (M.small(n_rad=3, nz=2, n_lambda=2), M.small(n_rad=4, nz=1, n_lambda=3),
                M.small(n_rad=5, nz=3, n_az=2, l3D=True, n_lambda=4)):
        cfg.n_rad_in = 1
        m = M.build_model(cfg)
        _frozen_parity(m, 3000, seed=3, n_prior=500)


def test_abi_rejects_what_it_cannot_reproduce(small_model):
    """Error behaviour of the C-ABI: explicit codes, never a silent wrong answer."""
    import copy
    from mcfost_amd.engine import Engine, McgpuError
    m = copy.copy(small_model)
    g = dict(m.grid)
    zl = g["z_lim"].copy()
    zl[m.cfg.n_rad * 3 + 2] *= 1.01                  # non-uniform vertical grid (e.g. lidefix)
    g["z_lim"] = zl
    m.grid = g
    with pytest.raises(McgpuError, match="z_lim"):
        Engine(m, 100)
    m = copy.copy(small_model)
    g = dict(m.grid)
    cm = g["cell_map_i"].copy()
    cm[5], cm[6] = cm[6], cm[5]                      # a permuted cell numbering
    g["cell_map_i"] = cm
    m.grid = g
    with pytest.raises(McgpuError, match="cell_map"):
        Engine(m, 100)
    m = copy.copy(small_model)
    m.frac_E_stars = np.full(m.n_lambda, 0.5)        # disk emission without prob_E_cell
    with pytest.raises(McgpuError, match="prob_E_cell"):
        Engine(m, 100)
    e = _engine(small_model, 100)
    with pytest.raises(McgpuError, match="frozen"):
        e.run_thermal(10, frozen=True)               # frozen mode without a prior
    e.close()


# ---------------------------------------------------------------------------
# Voronoi grid backend (mc_voronoi.hip.h)
# ---------------------------------------------------------------------------
@pytest.fixture(scope="module")
def voro_model():
    return M.build_voronoi_model(M.small(), 3000, seed=3)


def test_voronoi_cross_cell_probe_bit_exact(voro_model):
    """cross_Voronoi_cell on the device vs the oracle: same next cell, and -- because the
    plane tests are kept in unfused default real and the position update unfused FP64 --
    the same lengths and end points to the last bit."""
    m = voro_model
    g = m.grid
    e, o = _engine(m, 1e4), _oracle(m, 1e4)
    rng = np.random.default_rng(5)
    n = 4000
    cells = rng.integers(1, g["n_cells"] + 1, n).astype(np.int32)
    x = g["v_xyz_dp"][cells - 1] * (1 + 1e-3 * rng.standard_normal((n, 3)))
    d = rng.standard_normal((n, 3))
    d /= np.linalg.norm(d, axis=1)[:, None]
    prev = np.zeros(n, np.int32)
    # half of the probes come with a previous cell: one of the cell's own neighbours
    for i in range(0, n, 2):
        nb = g["v_neigh"][g["v_first"][cells[i] - 1] - 1:g["v_last"][cells[i] - 1]]
        prev[i] = nb[rng.integers(0, nb.size)]
    a = e.probe_cross_voronoi(x[:, 0], x[:, 1], x[:, 2], d[:, 0], d[:, 1], d[:, 2], cells, prev)
    b = o.cross_voronoi(x[:, 0], x[:, 1], x[:, 2], d[:, 0], d[:, 1], d[:, 2], cells, prev)
    assert np.array_equal(a["next_cell"], b["next_cell"])
    for k in ("l", "l_contrib", "l_void_before", "x1", "y1", "z1"):
        assert np.array_equal(a[k], b[k]), k
    e.close()


def test_frozen_parity_voronoi(voro_model):
    a, b = _frozen_parity(voro_model, 20000, seed=31, rtol=1e-7)
    assert a["counters"]["killed_star"] == b["counters"]["killed_star"]
    assert voro_model.grid["v_was_cut"].sum() > 0


def test_frozen_parity_voronoi_unpolarised_star_outside_box():
    cfg = M.small(lsepar_pola=False)
    cfg.star_xyz = (0.0, 0.0, 400.0)
    m = M.build_voronoi_model(cfg, 1500, seed=5)
    assert m.stars[0, 5] == 1
    a, b = _frozen_parity(m, 20000, seed=32, rtol=1e-7)
    assert a["counters"]["escaped"] == 20000  # packets that miss the box are binned directly


def test_voronoi_launch_geometry_independence(voro_model):
    m = voro_model
    e, o = _engine(m, 1e4), _oracle(m, 1e4)
    prior = o.run_thermal(2000, seed=1)["E_abs"]
    ref = e.run_thermal(10000, seed=9, frozen=True, E_prior=prior)
    for gb, bt in ((1, 64), (7, 128), (300, 256), (3, 768), (2, 1024)):   # (the 512-, 768- and 1024-thread builds)
        r = e.run_thermal(10000, seed=9, frozen=True, E_prior=prior, grid_blocks=gb, block_threads=bt)
        assert r["counters"] == ref["counters"]
        assert np.array_equal(r["sed"][4], ref["sed"][4])
        assert np.allclose(r["E_abs"], ref["E_abs"], rtol=1e-10, atol=1e-11 * ref["E_abs"].max())
    e.set_option("schedule", 2)   # the role schedule on this grid (opt-in: slower here, same packets)
    for gb, bt in ((0, 0), (3, 256)):
        r = e.run_thermal(10000, seed=9, frozen=True, E_prior=prior, grid_blocks=gb, block_threads=bt)
        assert r["counters"] == ref["counters"]
        assert np.array_equal(r["sed"][4], ref["sed"][4])
        assert np.allclose(r["E_abs"], ref["E_abs"], rtol=1e-10, atol=1e-11 * ref["E_abs"].max())
    e.set_option("schedule", 3)   # the pool schedule (mc_voronoi_pool.hip.h; opt-in: test_voronoi_pool_schedule)
    for gb, bt in ((0, 0), (7, 128)):
        r = e.run_thermal(10000, seed=9, frozen=True, E_prior=prior, grid_blocks=gb, block_threads=bt)
        assert r["counters"] == ref["counters"]
        assert np.array_equal(r["sed"][4], ref["sed"][4])
        assert np.allclose(r["E_abs"], ref["E_abs"], rtol=1e-10, atol=1e-11 * ref["E_abs"].max())
    e.close()


def test_voronoi_pool_schedule(voro_model):
    """The pool schedule (mc_voronoi_pool.hip.h; option "schedule" = 3, round 5): packet records in HBM, queues by phase
    and by neighbour-list class in LDS, one phase per wave pass.  Packet for packet the
    oracle whatever the pool holds -- 64 records per workgroup (fewer than a workgroup's lanes: most passes are thin),
    256, 4096 (the default) -- and whatever the launch geometry; a deposit cache of 64 slots must miss; a launch of fewer
    packets than one wave; live mode conserves packets and agrees with the one-packet-per-lane kernel."""
    m = voro_model
    e, o = _engine(m, 1e5), _oracle(m, 1e5)
    prior = o.run_thermal(2000, seed=1)["E_abs"]
    n = 30000
    ref = o.run_thermal(n, seed=17, frozen=True, E_prior=prior, n_threads=8)
    e.set_option("schedule", 3)

    def same(r, b=ref):
        assert r["counters"] == b["counters"], (r["counters"], b["counters"])
        assert np.array_equal(r["n_sent"], b["n_sent"]) and np.array_equal(r["sed"][4], b["sed"][4])
        assert np.allclose(r["E_abs"], b["E_abs"], rtol=1e-7, atol=1e-11 * b["E_abs"].max())
        for t in (0, 5, 6, 7, 8):
            assert np.allclose(r["sed"][t], b["sed"][t], rtol=1e-12, atol=1e-9), t

    for log_rec in (6, 8, 12):
        e.set_option("voronoi_pool_log_records", log_rec)
        for gb, bt in ((0, 0), (5, 256), (2, 768), (300, 64)):
            same(e.run_thermal(n, seed=17, frozen=True, E_prior=prior, grid_blocks=gb, block_threads=bt))
    e.set_option("voronoi_cache_log_slots", 6)
    same(e.run_thermal(n, seed=17, frozen=True, E_prior=prior))
    e.set_option("voronoi_cache_log_slots", 13)
    small = o.run_thermal(37, seed=18, frozen=True, E_prior=prior)
    same(e.run_thermal(37, seed=18, frozen=True, E_prior=prior), small)
    # live mode: conservation, and the temperature of the single-role kernel within the reference's gate
    nl = 1_000_000
    a = e.run_thermal(nl, seed=41)
    assert a["counters"]["packets"] == nl and a["counters"]["escaped"] + a["counters"]["killed_star"] == nl and a["n_sent"].sum() == nl
    e.set_option("schedule", 1)
    b = e.run_thermal(nl, seed=42)
    for k in ("crossings", "flights", "scatterings", "absorptions"):   # (the early in-flight estimates differ with the schedule)
        assert abs(a["counters"][k] / b["counters"][k] - 1) < 0.05, k
    Ta, Tb = e.temp_finale(a["E_abs"]), e.temp_finale(b["E_abs"])
    well = (Tb > 1.01 * m.cfg.T_min) & (b["E_abs"] > np.median(b["E_abs"]))
    ok, p75 = mc_similar(Tb[well], Ta[well], 0.05)
    assert ok, p75
    e.close()


def test_voronoi_live_statistical_parity(voro_model):
    """Live immediate re-emission on the Voronoi grid: device vs oracle, independent noise."""
    m = voro_model
    n = 1_000_000
    e, o = _engine(m, n), _oracle(m, n)
    a = e.run_thermal(n, seed=41)
    b = o.run_thermal(n, seed=42, n_threads=8)
    ca, cb = a["counters"], b["counters"]
    assert ca["packets"] == n and ca["escaped"] + ca["killed_star"] == n
    for k in ("crossings", "flights", "scatterings", "absorptions"):
        assert abs(ca[k] / cb[k] - 1) < 0.05, k
    Ta, Tb = e.temp_finale(a["E_abs"]), o.temp_finale(b["E_abs"])
    # cells that absorbed enough packets to have a defined temperature: the top half by energy
    well = (Tb > 1.01 * m.cfg.T_min) & (b["E_abs"] > np.median(b["E_abs"]))
    assert well.sum() > 1000
    ok, p75 = mc_similar(Tb[well], Ta[well], 0.05)
    assert ok, p75
    sa, sb = a["sed"][0].sum(axis=(0, 1)), b["sed"][0].sum(axis=(0, 1))
    okS, p75S = mc_similar(sb, sa, 0.10, mask_threshold=200.0)
    assert okS, p75S
    e.close()


def test_voronoi_abi_errors(voro_model):
    import ctypes as C
    from mcfost_amd.engine import McgpuError
    e = _engine(voro_model, 1e4)
    with pytest.raises(McgpuError):  # cylindrical probes refuse a Voronoi context
        e.probe_index_cell(np.zeros(1), np.zeros(1), np.zeros(1))
    e.close()
    m = M.build_voronoi_model(M.small(lsepar_pola=False), 300, seed=8)
    m.l_dark_zone = np.ones(m.n_cells, np.uint8)
    with pytest.raises(McgpuError):  # no dark zone on Voronoi grids (dust_transfer.f90:290-293)
        _engine(m, 1e4)


def test_voronoi_deposit_paths_agree(voro_model):
    """LDS deposit cache (default; 1024- and 512-thread workgroups, tiny cache that must miss)
    vs plain HBM atomics: same packets, same sums."""
    m = voro_model
    e, o = _engine(m, 1e4), _oracle(m, 1e4)
    prior = o.run_thermal(2000, seed=1)["E_abs"]
    ref = o.run_thermal(30000, seed=13, frozen=True, E_prior=prior, n_threads=8)
    runs = []
    runs.append(e.run_thermal(30000, seed=13, frozen=True, E_prior=prior))
    runs.append(e.run_thermal(30000, seed=13, frozen=True, E_prior=prior, block_threads=512))
    e.set_option("voronoi_cache_log_slots", 6)
    runs.append(e.run_thermal(30000, seed=13, frozen=True, E_prior=prior))
    e.set_option("voronoi_cache_log_slots", 13)
    e.set_option("deposit", 1)
    runs.append(e.run_thermal(30000, seed=13, frozen=True, E_prior=prior))
    for r in runs:
        assert r["counters"] == ref["counters"]
        assert np.allclose(r["E_abs"], ref["E_abs"], rtol=1e-7, atol=1e-11 * ref["E_abs"].max())
    e.close()


# ---------------------------------------------------------------------------
# SED mode (mc_mono.hip.h)
# ---------------------------------------------------------------------------
def _mono_parity(m, lam, n2, seed, n_chunks=64, **kw):
    from helpers import xI_close
    e, o = _engine(m, 1e5), _oracle(m, 1e5)
    for name, value in kw.pop("options", {}).items():
        e.set_option(name, value)
    rt1 = kw.get("rt1", True)
    a = e.run_mono(lam, n2, seed=seed, n_chunks=n_chunks, **kw)
    b = o.run_mono(lam, n2, seed=seed, n_chunks=n_chunks, n_threads=8,
                   **{k: v for k, v in kw.items() if k in ("n_phot_lim", "p_lambda", "rt1")})
    assert np.array_equal(a["n_sent_chunk"], b["n_sent_chunk"])   # every stream stops at the same packet
    assert a["counters"] == b["counters"]
    assert np.array_equal(a["n_sent"], b["n_sent"]) and np.array_equal(a["sed"][4], b["sed"][4])
    for t in (0, 5, 6, 7, 8):
        assert np.allclose(a["sed"][t], b["sed"][t], rtol=1e-11, atol=1e-11), t
    assert np.allclose(a["sed"][1:4], b["sed"][1:4], rtol=1e-5, atol=1e-6 * max(1.0, np.abs(b["sed"][0]).max()))
    if rt1:
        # with Stokes tracking every deposit inherits the default-real trigonometry of update_Stokes
        # (scattering.f90:1218; device sincosf vs glibc sinf/cosf: 1 ulp of default real per scattering)
        pola = m.cfg.lsepar_pola and m.cfg.aniso_method == 1
        xI_close(a["xI_scatt"], b["xI_scatt"], n_midplane_cells=0 if m.cfg.l3D else m.cfg.n_rad,
                 rtol=3e-5 if pola else 1e-6, atol_rel=1e-6 if pola else 1e-8)
        assert np.array_equal(a["xI_scatt_f32"], a["xI_scatt"].astype(np.float32))
    e.close()
    return a, b


@pytest.fixture(scope="module")
def sed_small():
    from helpers import sed_model
    return sed_model(M.small())


def test_sed_mode_parity_2d(sed_small):
    """mcgpu_run_mono vs the oracle's sequential streams: star-dominated, mixed and disk-dominated
    wavelengths; exact stopping packet per stream, SED bins, xI_scatt."""
    m = sed_small
    for lam in (3, 9, 14):
        a, b = _mono_parity(m, lam, 12, 60 + lam)
        assert a["sed"][4][0, m.capt_sup - 1, lam - 1] == 64 * 12
        assert a["n_sent"][lam - 1] == a["n_sent_chunk"].sum() == a["counters"]["packets"]


def test_sed_mode_variants():
    from helpers import sed_model
    _mono_parity(sed_model(M.small(lsepar_pola=False)), 4, 8, 7)                         # N_type_flux = 5
    _mono_parity(sed_model(M.small(lsepar_pola=False, lsepar_contrib=False)), 4, 8, 8)   # N_type_flux = 1
    _mono_parity(sed_model(M.small(n_rad=10, nz=5, n_az=6, l3D=True)), 4, 8, 9)          # 3D
    _mono_parity(sed_model(M.small(aniso_method=2, lsepar_pola=False)), 4, 8, 10)        # HG
    _mono_parity(sed_model(M.small(RT_n_incl=2, RT_n_az=3, RT_az_max=90.0, RT_imin=20.0, RT_imax=70.0)), 5, 8, 11)
    _mono_parity(sed_model(M.small()), 4, 8, 12, rt1=False)                              # no RT deposits


def test_sed_mode_speculative_commit(sed_small):
    """Counts large enough for the speculative commit (most of every stream is deposited right after a short probe,
    only the rest is scouted): every stream still stops at the oracle's packet, same SED bins and xI_scatt; and the
    plain two-pass path (option "speculation" = 0) gives the same."""
    m = sed_small
    for lam in (3, 9):
        a, b = _mono_parity(m, lam, 700, 40 + lam, n_chunks=16)
        assert a["sed"][4][0, m.capt_sup - 1, lam - 1] == 16 * 700
    _mono_parity(m, 9, 700, 49, n_chunks=16, options={"speculation": 0})
    # accumulate = 1 (the call does not own the accumulators): no speculation, still exact
    e, o = _engine(m, 1e5), _oracle(m, 1e5)
    e.run_mono(3, 5, seed=1, n_chunks=16)
    a = e.run_mono(9, 700, seed=49, n_chunks=16, accumulate=True)
    b = o.run_mono(9, 700, seed=49, n_chunks=16, n_threads=8)
    assert np.array_equal(a["n_sent_chunk"], b["n_sent_chunk"])
    e.close()


def test_sed_mode_default_real_records():
    """mcgpu_set_xI_precision(4): xI_scatt accumulated in default real (the reference's own type), the observers' records
    packed side by side.  Same packets and SED bins; xI_scatt to FP32 rounding; fetch / set / ray tracing / the zero-copy
    tensor all follow the type; back to FP64 the 1e-6 parity returns."""
    import torch
    from helpers import sed_model, xI_close
    # (observer counts on both sides of the layout's choice: 3 and 4 -> interleaved, 10 and 8 -> split, with and without
    # Stokes tracking, with and without contributions)
    for cfg, lam, n2 in ((M.small(RT_n_incl=3), 9, 40), (M.small(RT_n_incl=2, RT_n_az=2, RT_az_max=60.0, lsepar_pola=False), 5, 40),
                         (M.small(n_rad=10, nz=5, n_az=6, l3D=True, RT_n_incl=4), 4, 700),
                         (M.small(RT_n_incl=10), 9, 40), (M.small(RT_n_incl=5, RT_n_az=2, RT_az_max=60.0, lsepar_pola=False), 5, 40),
                         (M.small(RT_n_incl=4, RT_n_az=2, RT_az_max=60.0), 3, 40), (M.small(RT_n_incl=10, lsepar_contrib=False), 9, 40)):
        m = sed_model(cfg, n_thermal=50000)
        e, o = _engine(m, 1e5), _oracle(m, 1e5)
        e.set_rt1()
        e.set_xI_precision(4)
        a = e.run_mono(lam, n2, seed=3, n_chunks=16)
        b = o.run_mono(lam, n2, seed=3, n_chunks=16, n_threads=8)
        assert np.array_equal(a["n_sent_chunk"], b["n_sent_chunk"]) and a["counters"] == b["counters"]
        assert np.array_equal(a["sed"][4], b["sed"][4])
        xI_close(a["xI_scatt"], b["xI_scatt"], rtol=1e-4, n_midplane_cells=0 if cfg.l3D else cfg.n_rad, atol_rel=1e-5)
        t = e.device_xI()
        nRT = m.rt["RT_n_incl"] * m.rt["RT_n_az"]
        # (the packed default-real layout, mc_xi32.hip.h: per sub-bin the observers side by side -- the Stokes values and
        # the two origins a deposit can have; with contributions I is not stored, it is the sum of the origins -- padded
        # to whole 64-byte lines per sub-bin, interleaved or split, whichever a crossing touches in fewer lines)
        ntf = m.rt["N_type_flux"]
        nS, contrib = (4 if ntf in (4, 8) else 1), ntf in (5, 8)
        from mcfost_amd.engine import xi32_layout
        lay = xi32_layout(nRT, nS == 4, contrib)
        binf = lay["binf"]
        assert lay["lines_touched"] == {3: 1, 4: 2 if (nS == 4 and contrib) else 1, 8: 3, 10: 3 if nS == 4 else 1}[nRT]
        # (the library's own choice -- the mirror in engine.py serves the bench's accounting)
        assert (e.get_info("xi_bin_floats"), e.get_info("xi_lines_per_crossing"), e.get_info("xi_split")) == \
            (lay["binf"], lay["lines_touched"], float(lay["split"]))
        assert t.dtype == torch.float32 and t.numel() == m.n_cells * m.rt["n_theta_rt"] * m.rt["n_az_rt"] * binf
        x_f = a["xI_scatt"]
        stored = x_f.sum() - (x_f[..., 0, :, :].sum() if contrib else 0.0)    # (axes: icell, iRT, type, psup, phik)
        assert abs(float(t.double().sum()) / stored - 1) < 1e-6
        if contrib:   # I is the sum of its two origins
            assert np.allclose(x_f[..., 0, :, :], x_f[..., nS + 1, :, :] + x_f[..., nS + 3, :, :], rtol=3e-7, atol=0)
        # ray tracing from the default-real records == the oracle's on the same values (psup-symmetrised, see above;
        # handed to the device and read back: what the device holds)
        x = a["xI_scatt"].copy()
        if not cfg.l3D:
            x[:cfg.n_rad] = x[:cfg.n_rad].mean(axis=3, keepdims=True)
        x = x.astype(np.float32).astype(np.float64)
        e.set_xI(x)
        x_dev = e.fetch_xI()
        keep = [k for k in range(ntf) if not (contrib and k == 0)]
        assert np.array_equal(x_dev[:, :, keep], x[:, :, keep])
        assert np.allclose(x_dev[:, :, 0], x[:, :, 0], rtol=3e-7, atol=0)
        x = x_dev.astype(np.float64)
        ns, Ed = a["n_sent"][lam - 1], m.extra["E_disk"][lam - 1]
        got, _ = e.dust_map_sed(lam, m.extra["Tdust"], ns, Ed)
        ref = o.dust_map_sed(lam, x, m.extra["Tdust"], ns, Ed, n_threads=8)
        assert np.allclose(got, ref, rtol=1e-9, atol=1e-13 * np.abs(ref).max())
        e.set_xI_precision(8)          # back to FP64 sums: the accumulator is dropped and rebuilt
        a8 = e.run_mono(lam, n2, seed=3, n_chunks=16)
        pola = cfg.lsepar_pola and cfg.aniso_method == 1          # (tolerances of _mono_parity)
        xI_close(a8["xI_scatt"], b["xI_scatt"], n_midplane_cells=0 if cfg.l3D else cfg.n_rad,
                 rtol=3e-5 if pola else 1e-6, atol_rel=1e-6 if pola else 1e-8)
        e.close()


def test_sed_mode_packet_cap_and_launch_geometry(sed_small):
    m = sed_small
    a, b = _mono_parity(m, 3, 100000, 5, n_chunks=16, n_phot_lim=700.0)   # n_phot_lim ends every stream
    assert np.all(a["n_sent_chunk"] == 700)
    e = _engine(m, 1e5)
    ref = e.run_mono(9, 10, seed=3, n_chunks=32)
    for gb, bt in ((1, 64), (5, 128), (40, 256)):
        r = e.run_mono(9, 10, seed=3, n_chunks=32, grid_blocks=gb, block_threads=bt)
        assert np.array_equal(r["n_sent_chunk"], ref["n_sent_chunk"]) and r["counters"] == ref["counters"]
        assert np.allclose(r["sed"], ref["sed"], rtol=1e-10, atol=1e-10)
        assert np.allclose(r["xI_scatt"].sum(axis=(3, 4)), ref["xI_scatt"].sum(axis=(3, 4)), rtol=1e-9,
                           atol=1e-12 * np.abs(ref["xI_scatt"]).max())
    # accumulate: two wavelengths into the same SED arrays, like the lambda loop of run_sed_mc
    r1 = e.run_mono(3, 10, seed=3, n_chunks=32)
    r2 = e.run_mono(9, 10, seed=3, n_chunks=32, accumulate=True)
    assert np.array_equal(r2["sed"][4][..., 2], r1["sed"][4][..., 2]) and r2["n_sent"][2] == r1["n_sent"][2]
    assert np.array_equal(r2["sed"][4][..., 8], ref["sed"][4][..., 8])
    e.close()


@pytest.mark.parametrize("n2", [2000, 10000])
def test_sed_mode_full_size_properties(ref41_model, n2):
    """BASELINE config 2's SED half: 128 streams x 2000 packets in capt_sup, and the configuration's own 10 000."""
    from mcfost_amd.host import model as MM
    import copy
    m = copy.copy(ref41_model)
    e = _engine(m, 2e6)
    T = e.temp_finale(e.run_thermal(2_000_000, seed=3)["E_abs"])
    MM.repartition_energie(m, T)
    e.close()
    e = _engine(m, 2e6)
    lam = 20
    a = e.run_mono(lam, n2, seed=11)
    assert a["sed"][4][0, m.capt_sup - 1, lam - 1] == 128 * n2
    assert a["n_sent"][lam - 1] == a["n_sent_chunk"].sum() == a["counters"]["packets"]
    c = a["counters"]
    assert c["escaped"] + c["killed_star"] + c["absorptions"] == c["packets"]
    # forced scattering only removes energy: what escapes is below what was sent
    assert 0 < a["sed"][0][..., lam - 1].sum() <= c["escaped"]
    # the streams are statistically identical: packets sent per stream scatter like a negative binomial
    k = a["n_sent_chunk"].astype(float)
    assert abs(k.std() / k.mean() - np.sqrt((1 - n2 / k.mean()) / n2)) < 0.02
    x = a["xI_scatt"]
    assert np.all(x[:, :, 0] >= 0) and np.all(x[:, :, 4] == 0) and np.all(x[:, :, 6] == 0)
    assert np.allclose(x[:, :, 0], x[:, :, 5] + x[:, :, 7], rtol=1e-9, atol=1e-12 * x.max())   # star + disk = total
    e.close()


def test_sed_mode_abi_errors(sed_small):
    from mcfost_amd.engine import McgpuError
    e = _engine(sed_small, 1e4)
    with pytest.raises(McgpuError):
        e.run_mono(0, 5)                      # wavelength out of range
    with pytest.raises(McgpuError):
        e.run_mono(3, 5, n_chunks=2 ** 23)    # more streams than the engine takes
    e.close()


def test_sed_mode_voronoi():
    """SED mode on a Voronoi grid: same checks as on the cylindrical grids."""
    from helpers import sed_model
    m = sed_model(M.small(), voronoi_sites=3000, n_thermal=100000)
    for lam in (3, 9, 14):
        a, b = _mono_parity(m, lam, 10, 80 + lam)
        assert a["sed"][4][0, m.capt_sup - 1, lam - 1] == 64 * 10
    _mono_parity(sed_model(M.small(lsepar_pola=False), voronoi_sites=1500, n_thermal=50000), 9, 10, 4)
    # default-real records (the commit pass with the flight's deposit weights, mc_mono.hip.h) on this grid and on a
    # spherical one: the same packets, xI_scatt to default-real rounding
    from helpers import xI_close
    for mm, lam in ((m, 9), (sed_model(M.small(grid_type=2), n_thermal=20000), 5)):
        e, o = _engine(mm, 1e5), _oracle(mm, 1e5)
        e.set_rt1()
        e.set_xI_precision(4)
        a = e.run_mono(lam, 10, seed=77, n_chunks=16)
        b = o.run_mono(lam, 10, seed=77, n_chunks=16, n_threads=8)
        assert np.array_equal(a["n_sent_chunk"], b["n_sent_chunk"])
        assert np.array_equal(a["sed"][4], b["sed"][4])
        if mm is m:
            assert a["counters"] == b["counters"]
            xI_close(a["xI_scatt"], b["xI_scatt"], rtol=1e-4, atol_rel=1e-5)
        else:   # (the spherical grid's midplane cone: sub-bins compared summed, test_sed_mode_on_spherical_grids)
            xa, xb = a["xI_scatt"], b["xI_scatt"]
            assert np.allclose(xa.sum(axis=(2, 3, 4)), xb.sum(axis=(2, 3, 4)), rtol=2e-3, atol=1e-5 * np.abs(xb).max())
        e.close()


# ---------------------------------------------------------------------------
# RT1 ray-traced dust SED (mc_raytrace.hip.h)
# ---------------------------------------------------------------------------
def _dust_map_parity(cfg, lam, n2, seed, ang=0.0, sym=True, tau_obs=100.0):
    from helpers import sed_model
    m = sed_model(cfg, n_thermal=50000)
    e, o = _engine(m, 1e5), _oracle(m, 1e5)
    a = e.run_mono(lam, n2, seed=seed, n_chunks=32)
    if not cfg.l3D:
        # a ray that crosses a midplane cell from its upper to its lower wall has its midpoint at z = +-rounding
        # (see helpers.xI_close): make psup irrelevant in that layer and hand the array back (mcgpu_set_xI)
        x = a["xI_scatt"].copy()
        x[:cfg.n_rad] = x[:cfg.n_rad].mean(axis=3, keepdims=True)
        e.set_xI(x)
        assert np.array_equal(e.fetch_xI(), x)
    ns, Ed = a["n_sent"][lam - 1], m.extra["E_disk"][lam - 1]
    got, ms = e.dust_map_sed(lam, m.extra["Tdust"], ns, Ed, ang_disque=ang, l_sym_ima=sym, tau_dark_zone_obs=tau_obs)
    ref = o.dust_map_sed(lam, e.fetch_xI(), m.extra["Tdust"], ns, Ed, ang_disque=ang, l_sym_ima=sym,
                         tau_dark_zone_obs=tau_obs, n_threads=8)
    e.close()
    assert (ref[:, 0] > 0).all() and ms > 0
    # same rays through the same cells; exp() and the summation order differ
    assert np.allclose(got, ref, rtol=1e-9, atol=1e-13 * np.abs(ref).max()), np.abs(got / ref - 1).max()
    return got, ref


def test_rt1_dust_map_parity_2d():
    """mcgpu_rt1_dust_map on the xI_scatt the SED Monte Carlo left in HBM vs the oracle's dust_map restatement
    on the same (fetched) xI_scatt: scattered starlight, mixed and thermal wavelengths."""
    cfg = M.small(RT_n_incl=3)
    for lam in (3, 9, 14):
        got, ref = _dust_map_parity(cfg, lam, 50, 20 + lam)
        assert np.allclose(got[:, 0], got[:, 5] + got[:, 6] + got[:, 7], rtol=1e-9, atol=0)


def test_rt1_dust_map_variants():
    _dust_map_parity(M.small(lsepar_pola=False), 5, 30, 3)                                      # N_type_flux = 5
    _dust_map_parity(M.small(lsepar_pola=False, lsepar_contrib=False), 5, 30, 4)                # N_type_flux = 1
    _dust_map_parity(M.small(n_rad=10, nz=5, n_az=6, l3D=True), 5, 30, 5)                       # 3D
    _dust_map_parity(M.small(RT_n_incl=2, RT_n_az=3, RT_az_max=90.0, RT_imin=20.0, RT_imax=70.0), 12, 30, 6,
                     ang=17.0, sym=False)                                                       # rotated, full plane
    _dust_map_parity(M.small(), 12, 30, 7, tau_obs=0.5)                                         # early cut-off


def test_rt1_image_parity():
    """mcgpu_rt1_image (one wavefront per pixel, sub-pixel refinement) after an image-mode Monte Carlo (every
    stream sends exactly n_photons_image packets) vs the oracle's dust_map method 2: same number of rays (same
    refinement decisions), same pixels."""
    from helpers import sed_model
    for cfg, ang, sym, npx, npy in ((M.small(RT_n_incl=3), 0.0, True, 33, 33),
                                    (M.small(RT_n_incl=2, RT_n_az=2, RT_az_max=60.0), 17.3, False, 24, 15),
                                    (M.small(n_rad=10, nz=5, n_az=6, l3D=True, lsepar_pola=False), 0.0, False, 16, 16)):
        m = sed_model(cfg, n_thermal=50000)
        e, o = _engine(m, 1e5), _oracle(m, 1e5)
        lam = 9
        a = e.run_mono(lam, 10 ** 12, seed=4, n_chunks=16, n_phot_lim=500.0)
        assert np.all(a["n_sent_chunk"] == 500)
        x = a["xI_scatt"].copy()
        if not cfg.l3D:
            x[:cfg.n_rad] = x[:cfg.n_rad].mean(axis=3, keepdims=True)   # see _dust_map_parity
            e.set_xI(x)
        ns, Ed = a["n_sent"][lam - 1], m.extra["E_disk"][lam - 1]
        got, n_rays, ms = e.dust_map_image(lam, m.extra["Tdust"], ns, Ed, npx, npy, 2.2 * cfg.rout, zoom=1.2,
                                           ang_disque=ang, l_sym_ima=sym)
        ref, nr = o.dust_map_image(lam, x, m.extra["Tdust"], ns, Ed, npx, npy, 2.2 * cfg.rout, zoom=1.2, ang_disque=ang,
                                   l_sym_ima=sym, n_threads=8)
        e.close()
        assert n_rays == nr and ms > 0
        assert ref[0].max() > 0
        assert np.allclose(got, ref, rtol=1e-9, atol=1e-13 * np.abs(ref).max()), np.abs(got - ref).max() / np.abs(ref).max()
        if sym:   # only the left half is computed (the reference mirrors it when it writes the image)
            assert not got[..., npx // 2 + npx % 2:].any()


def test_rt1_dust_map_abi_errors(sed_small):
    from mcfost_amd.engine import McgpuError
    e = _engine(sed_small, 1e5)
    with pytest.raises(McgpuError):   # no xI_scatt yet
        e.dust_map_sed(3, sed_small.extra["Tdust"], 100.0, 0.0)
    e.run_mono(3, 5, seed=1, n_chunks=8, fetch_xI=False)
    with pytest.raises(McgpuError):
        e.dust_map_sed(3, sed_small.extra["Tdust"], 0.0, 0.0)        # n_sent_photons must be positive
    with pytest.raises(McgpuError):
        e.dust_map_sed(0, sed_small.extra["Tdust"], 100.0, 0.0)      # lambda out of range
    with pytest.raises(McgpuError):
        e.dust_map_image(3, sed_small.extra["Tdust"], 100.0, 0.0, 0, 8, 100.0)     # no pixels
    with pytest.raises(McgpuError):
        e.dust_map_image(3, sed_small.extra["Tdust"], 100.0, 0.0, 8, 8, -1.0)      # map size
    e.close()


def test_temperature_and_sed_end_to_end():
    """BASELINE config 2 in small: temperature step (live), emission tables of the SED step from the GPU's own
    Tdust, SED Monte Carlo of every wavelength, ray-traced dust SED -- engine vs CPU oracle through the same host
    sequence (mcfost_amd/host/pipeline.py), each with its own noise, against the reference's gates
    (test_suite/test_mcfost.py:88,104-109: p75 of the relative difference < 5 % on T, < 10 % on the SEDs)."""
    import copy
    from helpers import OracleBackend
    from mcfost_amd.host import pipeline as P
    cfg = M.small(RT_n_incl=3)
    n_th, n2, nch = 400000, 800, 32
    mg, mc = M.build_model(cfg), M.build_model(cfg)
    e = _engine(mg, n_th)
    g = P.temperature_and_sed(P.EngineBackend(e), mg, n_th, n2, seed=11, n_chunks=nch)
    e.close()
    c = P.temperature_and_sed(OracleBackend(_oracle(mc, n_th)), mc, n_th, n2, seed=23, n_chunks=nch)
    sel = c["Tdust"] > 1.01 * cfg.T_min
    assert np.percentile(np.abs(g["Tdust"][sel] / c["Tdust"][sel] - 1), 75) < 0.05
    # Monte Carlo SED: flux per inclination bin and wavelength (azimuth summed), where the oracle has signal
    fg, fc = P.sed_flux(mg, g["sed_mc"], g["n_sent"])[0].sum(axis=0), P.sed_flux(mc, c["sed_mc"], c["n_sent"])[0].sum(axis=0)
    ok = c["sed_mc"][4].sum(axis=0) >= 200          # bins with at least 200 packets
    assert ok.sum() > 0.5 * ok.size
    assert np.percentile(np.abs(fg[ok] / fc[ok] - 1), 75) < 0.10
    # ray-traced SED of the dust, Stokes I of every observer and wavelength
    ig, ic_ = g["sed_rt"][:, :, 0], c["sed_rt"][:, :, 0]
    assert (ic_ > 0).all()
    assert np.percentile(np.abs(ig / ic_ - 1), 75) < 0.10
    # the stars' term (compute_stars_map): same screens and rays apart from the two runs' seeds
    sg, sc = g["sed_rt_stars"], c["sed_rt_stars"]
    assert (sc > 0).all() and np.allclose(sg, sc, rtol=0.05)
    # and the stages ran on the device: every wavelength sent packets, every stream stopped by its count
    assert (g["n_sent"] >= nch * n2).all()


def test_ism_emission_on_the_gpu(small_model):
    """emit_packet's third branch (emit_packet_ISM, stars.f90:728-785) through mcgpu_set_ism: thermal step on
    2D / 3D / Voronoi grids and the SED step; a draw beyond frac_E_disk without the sphere is an error."""
    import copy
    from mcfost_amd.engine import McgpuError

    def with_ism(m, f_star=0.4, f_disk=0.7):
        m = copy.copy(m)
        g = m.grid
        if g.get("grid_type", 1) == 3:
            lim = g["limits"]
            R = 1.000001 * float(np.sqrt(lim[1] ** 2 + lim[3] ** 2 + lim[5] ** 2))
        else:
            R = 1.000001 * float(np.sqrt(g["Rmax2"] + g["zmax"][-1] ** 2))
        m.ism = dict(R_ISM=R, centre_ISM=(0.0, 0.0, 0.0))
        rng = np.random.default_rng(0)
        E_cell = rng.random((m.n_lambda, m.n_cells)) * m.kappa_factor[None, :]
        pe = np.zeros((m.n_lambda, m.n_cells + 1))
        pe[:, 1:] = np.cumsum(E_cell, axis=1)
        pe /= pe[:, -1:]
        m.prob_E_cell = pe.reshape(-1)
        m.frac_E_stars = np.full(m.n_lambda, f_star)
        m.frac_E_disk = np.full(m.n_lambda, f_disk)
        return m

    a, b = _frozen_parity(with_ism(small_model), 20000, seed=51, rtol=1e-5)
    assert a["counters"]["escaped"] < 20000        # ISM packets that were never absorbed are not binned
    _frozen_parity(with_ism(M.build_model(M.small(n_rad=10, nz=5, n_az=6, l3D=True))), 10000, seed=52, rtol=1e-5)
    _frozen_parity(with_ism(M.build_voronoi_model(M.small(lsepar_pola=False), 1000, seed=4)), 10000, seed=53, rtol=1e-5)
    _mono_parity(with_ism(small_model), 5, 6, 54)
    m = with_ism(small_model)
    m.ism = None
    e = _engine(m, 1e4)
    with pytest.raises(McgpuError):
        e.run_thermal(2000, seed=1)
    e.close()


def test_frozen_parity_pascucci_and_3d_baseline_grids():
    """BASELINE configs 1 and 3 on their own grids (Pascucci 100x70, 61 wavelengths, isotropic; ref4.1_3D at
    100 x 50 x 12 azimuths -- the full 72 azimuths in the property test below)."""
    _frozen_parity(M.build_model(M.pascucci()), 60000, seed=61, n_prior=20000)
    # thick-midplane random walks of 1e4+ flights: FMA-level drift accumulates in the path lengths (counts stay exact)
    _frozen_parity(M.build_model(M.ref41_3d(n_az=12)), 60000, seed=62, n_prior=20000, rtol=1e-6)


def test_full_size_properties_ref41_3d():
    """BASELINE config 3's grid (100 x 50 x 72 = 720 000 cells) at a GPU-sized packet count: conservation,
    azimuthal symmetry of the result, agreement of its azimuthal mean with the 2D run of the same disk."""
    m3 = M.build_model(M.ref41_3d())
    n = 20_000_000
    e = _engine(m3, n)
    a = e.run_thermal(n, seed=71)
    c = a["counters"]
    assert c["packets"] == n and c["escaped"] + c["killed_star"] == n and a["n_sent"].sum() == n
    T3 = e.temp_finale(a["E_abs"]).reshape(72, 100, 100)        # (k, j: -50..-1,1..50, i)
    e.close()
    Tm = T3.mean(axis=0)
    hot = Tm > 1.2 * m3.cfg.T_min
    # every azimuth is a noisy copy of the mean: no systematic azimuthal structure
    dev = (T3[:, hot] / Tm[hot] - 1.0)
    assert abs(dev.mean(axis=1)).max() < 0.01
    # north / south symmetry of the azimuthal mean
    north, south = Tm[50:, :], Tm[49::-1, :]
    sel = (north > 1.2 * m3.cfg.T_min) & (south > 1.2 * m3.cfg.T_min)
    assert np.median(np.abs(north[sel] / south[sel] - 1.0)) < 0.01
    # and the 2D grid of the same disk (nz = 50) gives the same temperature
    cfg2 = M.ref41()
    cfg2.nz = 50
    m2 = M.build_model(cfg2)
    e2 = _engine(m2, n)
    T2 = e2.temp_finale(e2.run_thermal(n, seed=72)["E_abs"]).reshape(50, 100)
    e2.close()
    sel = (T2 > 1.2 * cfg2.T_min) & (north > 1.2 * cfg2.T_min)
    ok, p75 = mc_similar(T2[sel], north[sel], 0.05)
    assert ok, p75


@pytest.mark.parametrize("snap", [0, 1])
def test_3d_midplane_conventions_frozen_parity(snap):
    """3D grids, both treatments of a crossing that lands on z = 0 (include/mcgpu.h: mcgpu_set_midplane_snap).
    snap = 1 (engine default): the landing point is put at sign(grid_prec, w), the reference's own correction for
    z1 == 0 (cylindrical_grid.f90:1158-1165) applied to every rounding residue -- device and oracle then agree packet
    for packet.  snap = 0 is the reference's literal arithmetic (what the golden walks pin): the SIGN of the residue
    of z0 + t*w decides the hemisphere of the next cell, and it is decided by the last ulp (FMA or not), so a device
    build and a host build of the same formula part ways on a few crossings in a million; the packets concerned cross
    one cell in the mirror hemisphere and their histories part from there: the run is the same statistically (the
    reference's own gate, p75 < 5 % on T, holds either way -- next test) but no longer packet for packet."""
    for cfg, n, seed in ((M.small(n_rad=12, nz=6, n_az=8, l3D=True), 20000, 8), (M.ref41_3d(n_az=12), 40000, 62)):
        m = M.build_model(cfg)
        m.midplane_snap = snap
        if snap == 1:
            _frozen_parity(m, n, seed=seed, n_prior=20000, rtol=1e-6)
            continue
        e, o = _engine(m, n), _oracle(m, n)
        prior = o.run_thermal(20000, seed=1)["E_abs"]
        a = e.run_thermal(n, seed=seed, frozen=True, E_prior=prior)
        b = o.run_thermal(n, seed=seed, frozen=True, E_prior=prior, n_threads=8)
        e.close()
        ca, cb = a["counters"], b["counters"]
        for k in ("packets", "escaped", "killed_star"):
            assert ca[k] == cb[k]
        assert np.array_equal(a["n_sent"], b["n_sent"])
        # (a packet that lands on the other side of the midplane for one crossing meets other random numbers'
        # outcomes from there on: its history parts from the oracle's, so the totals agree statistically only)
        for k in ("crossings", "flights", "scatterings", "absorptions"):
            assert abs(ca[k] - cb[k]) <= 3 + 5e-2 * cb[k], (k, ca, cb)   # (measured 1.1 ... 3.003 % over rounds 3 and 4)
        # hemispheres summed: the deposits are the same to the packets that parted
        n_az, nz2, n_rad = cfg.n_az, 2 * cfg.nz, cfg.n_rad
        Ea, Eb = a["E_abs"].reshape(n_az, nz2, n_rad), b["E_abs"].reshape(n_az, nz2, n_rad)
        Ea, Eb = Ea[:, :cfg.nz][:, ::-1] + Ea[:, cfg.nz:], Eb[:, :cfg.nz][:, ::-1] + Eb[:, cfg.nz:]
        assert np.isclose(Ea.sum(), Eb.sum(), rtol=3e-2)


def test_3d_midplane_conventions_give_the_same_temperature():
    """Full-size 3D grid (720 000 cells), live mode, 2e7 packets with either convention: same conservation laws and
    statistically the same temperature -- the choice is a matter of reproducibility, not of physics."""
    n = 20_000_000
    T = {}
    for snap in (0, 1):
        m3 = M.build_model(M.ref41_3d())
        m3.midplane_snap = snap
        e = _engine(m3, n)
        a = e.run_thermal(n, seed=71)
        c = a["counters"]
        assert c["packets"] == n and c["escaped"] + c["killed_star"] == n and a["n_sent"].sum() == n
        T[snap] = e.temp_finale(a["E_abs"]).reshape(72, 100, 100).mean(axis=0)
        e.close()
    sel = (T[0] > 1.2 * m3.cfg.T_min) & (T[1] > 1.2 * m3.cfg.T_min)
    ok, p75 = mc_similar(T[0][sel], T[1][sel], 0.01)
    assert ok, p75
    north, south = T[0][50:, :], T[0][49::-1, :]
    s2 = (north > 1.2 * m3.cfg.T_min) & (south > 1.2 * m3.cfg.T_min)
    assert np.median(np.abs(north[s2] / south[s2] - 1.0)) < 0.01


def test_frozen_parity_with_the_reference_style_dark_zone():
    """A disk massive enough to have a dark zone by define_dark_zone's rule (optical_depth.f90:1425-1651, restated
    in the oracle): mirror at the zone's edge in the thermal step, packets dropped inside it in the SED step."""
    from helpers import sed_model
    cfg = M.small(n_rad=30, nz=20, dust_mass=3e-2)
    m = M.build_model(cfg)
    o = _oracle(m, 1e5)
    lam = int(np.argmax(m.lam > 0.81)) + 1
    dz = o.define_dark_zone(lam, 1500.0)
    assert 0 < dz.sum() < dz.size // 2
    m.l_dark_zone = dz
    a, b = _frozen_parity(m, 30000, seed=81, rtol=1e-6)
    assert a["counters"]["dark_mirrors"] > 0
    ms = sed_model(cfg, n_thermal=30000)
    ms.l_dark_zone = dz
    _mono_parity(ms, 6, 8, 82)


def test_voronoi_at_scale_properties():
    """BASELINE config 5's stand-in at scale: 100 000 SPH-like sites sampled from the ref4.1 disk (the tessellation is
    cached under tools/cache; about a minute of scipy otherwise).  (a) conservation over 1e7 live packets; (b) the
    deposit cache and plain HBM atomics run the same packets to the same sums; (c) live mode against the CPU oracle on
    the same tessellation (independent noise): event rates per packet within 3 %, the reference's own gate
    p75(|dT| / T) < 5 % (test_suite/test_mcfost.py:88) over the cells that absorbed enough packets.
    (Not tested, because it is not true at this resolution: agreement with the 2D cylindrical run of the same disk.  The
    vertical optical depths through the tessellation match the cylindrical grid's to ~20 %, but 1e5 sites put 3-5 cells
    across the whole disk inside 10 AU, so the cell that absorbs the starlight reaches down to the midplane: the
    midplane comes out 2-3x hotter and the mid-infrared SED 10x brighter than on the 100 x 70 grid, with the CPU oracle
    exactly as with the device -- a property of the stand-in, slowly converging with the number of sites.)"""
    import os
    cache = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "cache")
    cfg = M.ref41()
    mv = M.build_voronoi_model(cfg, 100000, seed=1, cache_dir=cache)
    n = 10_000_000
    e = _engine(mv, n)
    a = e.run_thermal(n, seed=81)
    c = a["counters"]
    assert c["packets"] == n and c["escaped"] + c["killed_star"] == n and a["n_sent"].sum() == n
    Tv = e.temp_finale(a["E_abs"])
    # (b) same packets through both deposit paths (frozen on the live run's energies)
    nf = 2_000_000
    r1 = e.run_thermal(nf, seed=82, frozen=True, E_prior=a["E_abs"])
    e.set_option("deposit", 1)
    r2 = e.run_thermal(nf, seed=82, frozen=True, E_prior=a["E_abs"])
    e.close()
    assert r1["counters"] == r2["counters"] and np.array_equal(r1["sed"][4], r2["sed"][4])
    assert np.allclose(r1["E_abs"], r2["E_abs"], rtol=1e-9, atol=1e-11 * r2["E_abs"].max())
    # (c) the CPU oracle on the same tessellation
    no = 2_000_000
    o = _oracle(mv, no)
    b = o.run_thermal(no, seed=84, n_threads=16)
    cb = b["counters"]
    for k in ("crossings", "flights", "scatterings", "absorptions"):
        assert abs((c[k] / n) / (cb[k] / no) - 1) < 0.03, k
    To = o.temp_finale(b["E_abs"])
    well = (To > 1.5 * cfg.T_min) & (b["E_abs"] > np.percentile(b["E_abs"], 75))
    assert well.sum() > 10000
    okT, p75 = mc_similar(To[well], Tv[well], 0.05)
    assert okT, p75


def test_rt2_deposits_parity():
    """Ray tracing method 2's deposits (save_radiation_field, radiation_field.f90:91-129): I_spec per cell and direction
    bin and I_spec_star for unscattered starlight, mcgpu_run_mono with rt1 = 2 against the oracle -- same stopping packets
    and counters, the arrays to the rounding of the summation order; polarised with contributions, unpolarised, no
    contributions; conservation: every crossing deposits l * I exactly once (sum of both arrays = sum of xJ-like total)."""
    from helpers import sed_model
    for cfg, lam in ((M.small(), 3), (M.small(), 12), (M.small(lsepar_pola=False), 5),
                     (M.small(lsepar_pola=False, lsepar_contrib=False), 9)):
        m = sed_model(cfg, n_thermal=50000)
        e, o = _engine(m, 1e5), _oracle(m, 1e5)
        a = e.run_mono(lam, 40, seed=11, n_chunks=16, rt2=(15, 15))
        b = o.run_mono(lam, 40, seed=11, n_chunks=16, n_threads=8, rt2=(15, 15))
        assert np.array_equal(a["n_sent_chunk"], b["n_sent_chunk"]) and a["counters"] == b["counters"]
        assert np.array_equal(a["sed"][4], b["sed"][4])
        scale = np.abs(b["I_spec"]).max()
        # a path whose midpoint sits on the midplane or on a bin edge to rounding may land in the neighbouring bin
        # (see helpers.xI_close): compare the sums over the direction bins tightly, the bins themselves almost everywhere
        # (Q, U, V inherit the default-real trigonometry of update_Stokes, scattering.f90:1218: 1 ulp of default real per
        # scattering between sincosf and glibc -- the tolerance of the rt1 test)
        ns = 4 if (cfg.lsepar_pola and cfg.aniso_method == 1) else 1
        rt = np.full(a["I_spec"].shape[-1], 1e-9)
        at = np.full(a["I_spec"].shape[-1], 1e-12 * scale)
        if ns == 4:
            rt[1:4], at[1:4] = 3e-5, 1e-6 * scale
        sa, sb = a["I_spec"].sum(axis=(1, 2)), b["I_spec"].sum(axis=(1, 2))
        assert np.all(np.abs(sa - sb) <= rt * np.abs(sb) + at)
        # the midplane layer (cells 1..n_rad): a path from its upper to its mirrored lower wall has its midpoint at
        # z = +-rounding, and the sign picks cos(theta) or its mirror image (radiation_field.f90:114-118): compare the sum of
        # a bin and its mirror there
        nr = cfg.n_rad
        ua, ub = a["I_spec"].copy(), b["I_spec"].copy()
        ua[:nr] = ua[:nr] + ua[:nr, :, ::-1]
        ub[:nr] = ub[:nr] + ub[:nr, :, ::-1]
        bad = np.abs(ua - ub) > rt * np.abs(ub) + at
        assert bad.sum() <= max(4, 2e-4 * np.count_nonzero(ub)), (bad.sum(), np.count_nonzero(ub))
        assert np.allclose(a["I_spec_star"], b["I_spec_star"], rtol=1e-9, atol=1e-12 * max(scale, b["I_spec_star"].max()))
        if lam == 3:
            assert a["I_spec_star"].sum() > 0 and a["I_spec"][..., 0].sum() > 0
        if cfg.lsepar_contrib:   # I = its star + dust parts (slots n_Stokes + 2 and n_Stokes + 4)
            assert np.allclose(a["I_spec"][..., 0], a["I_spec"][..., ns + 1] + a["I_spec"][..., ns + 3], rtol=1e-9, atol=1e-12 * scale)
        # a second call accumulates; the default-real fetch is the rounding of the sums
        a2 = e.run_mono(lam, 40, seed=12, n_chunks=16, rt2=(15, 15), accumulate=True)
        assert a2["I_spec"].sum() > 1.5 * a["I_spec"].sum()
        e.close()
    # 3D grids refuse it (the reference: "only 2D")
    from mcfost_amd.engine import McgpuError
    e = _engine(M.build_model(M.small(n_rad=10, nz=5, n_az=4, l3D=True)), 1e4)
    with pytest.raises(McgpuError):
        e.set_rt2()
    e.close()


def test_sed_mode_on_spherical_grids():
    """k_mono_sph (mono_body with spherical_grid.f90's operators), 2D and 3D: every stream stops at the oracle's packet, the
    same SED bins, xI_scatt per cell; tolerances of the midplane cone's double root as in the thermal test
    (test_kernel_emulation._check_spherical): a zero-length crossing more or less, sub-bins of the layer next to the cone
    and (3D) the hemisphere's label compared summed."""
    from helpers import sed_model, xI_close
    for kw, lam in ((dict(), 5), (dict(lsepar_pola=False), 9), (dict(n_rad=10, nz=5, n_az=6, l3D=True), 4)):
        cfg = M.small(grid_type=2, **kw)
        m = sed_model(cfg, n_thermal=20000)
        e, o = _engine(m, 1e5), _oracle(m, 1e5)
        for rt1 in (False, True):
            a = e.run_mono(lam, 12, seed=41, n_chunks=32, rt1=rt1)
            b = o.run_mono(lam, 12, seed=41, n_chunks=32, rt1=rt1, n_threads=8)
            ca, cb = a["counters"], b["counters"]
            assert np.array_equal(a["n_sent_chunk"], b["n_sent_chunk"])
            for k in cb:
                if k == "crossings":
                    assert abs(ca[k] - cb[k]) <= 3 + (3e-2 if cfg.l3D else 3e-4) * cb[k]
                else:
                    assert ca[k] == cb[k], k
            assert np.array_equal(a["sed"][4], b["sed"][4])
            assert np.allclose(a["sed"][0], b["sed"][0], rtol=1e-11, atol=1e-11)
            if rt1:
                pola = cfg.lsepar_pola and cfg.aniso_method == 1
                rtol, atol_rel = (3e-5, 1e-6) if pola else (1e-6, 1e-8)
                xa, xb = a["xI_scatt"], b["xI_scatt"]
                scale = np.abs(xb).max()
                assert scale > 0
                if cfg.l3D:
                    g = m.grid
                    i, j, k = g["cell_map_i"][:m.n_cells], g["cell_map_j"][:m.n_cells], g["cell_map_k"][:m.n_cells]
                    key = (i - 1) + g["n_rad"] * ((np.abs(j) - 1) + g["nz"] * (k - 1))
                    ta, tb = xa.sum(axis=(3, 4)).reshape(m.n_cells, -1), xb.sum(axis=(3, 4)).reshape(m.n_cells, -1)
                    fa = np.stack([np.bincount(key, weights=ta[:, q]) for q in range(ta.shape[1])], 1)
                    fb = np.stack([np.bincount(key, weights=tb[:, q]) for q in range(tb.shape[1])], 1)
                    assert np.allclose(fa, fb, rtol=rtol, atol=atol_rel * np.abs(fb).max())
                else:
                    assert np.allclose(xa.sum(axis=(3, 4)), xb.sum(axis=(3, 4)), rtol=rtol, atol=atol_rel * scale)
                    xI_close(xa[cfg.n_rad:], xb[cfg.n_rad:], rtol=rtol, atol_rel=atol_rel)
        e.close()


def test_ray_tracer_on_spherical_grids():
    """rt1_integ_ray / optical_length_tot with spherical_grid.f90's operators: the dust's SED (2D, 3D), an image, the stars'
    SED and image -- the device against the oracle on the same xI_scatt (the layer next to the midplane cone blurred:
    test_kernel_emulation._blur_midplane_layer says why)."""
    from helpers import sed_model
    from test_kernel_emulation import _blur_midplane_layer
    for kw in (dict(RT_n_incl=3), dict(lsepar_pola=False), dict(n_rad=10, nz=5, n_az=6, l3D=True)):
        cfg = M.small(grid_type=2, **kw)
        m = sed_model(cfg, n_thermal=50000)
        e, o = _engine(m, 1e5), _oracle(m, 1e5)
        for lam in (3, 12):
            a = e.run_mono(lam, 30, seed=20 + lam, n_chunks=32)
            x, T = _blur_midplane_layer(m, a["xI_scatt"], m.extra["Tdust"])
            e.set_xI(x)
            ns, Ed = a["n_sent"][lam - 1], m.extra["E_disk"][lam - 1]
            got, ms = e.dust_map_sed(lam, T, ns, Ed)
            ref = o.dust_map_sed(lam, e.fetch_xI(), T, ns, Ed, n_threads=8)
            assert (ref[:, 0] > 0).all() and ms > 0
            assert np.allclose(got, ref, rtol=1e-9, atol=1e-13 * np.abs(ref).max()), np.abs(got / ref - 1).max()
            flux = np.array([1.3])
            want = o.stars_map_sed(lam, flux, seed=4)
            have = e.stars_map_sed(lam, flux, seed=4)
            assert np.allclose(have, want, rtol=2e-6) and (want > 0).all()
        if not cfg.l3D:
            got, n_rays, ms = e.dust_map_image(12, T, ns, Ed, 24, 24, 2.2 * cfg.rout, zoom=1.2, l_sym_ima=False, ang_disque=17.3)
            ref, nr = o.dust_map_image(12, x, T, ns, Ed, 24, 24, 2.2 * cfg.rout, zoom=1.2, l_sym_ima=False, ang_disque=17.3, n_threads=8)
            assert n_rays == nr and ref[0].max() > 0
            assert np.allclose(got, ref, rtol=1e-9, atol=1e-13 * np.abs(ref).max())
            rs = cfg.R_star * 0.00465047
            want, wpos = o.stars_map_image(12, flux, 33, 33, 33 * rs / 6.0, seed=5)
            have, hpos = e.stars_map_image(12, flux, 33, 33, 33 * rs / 6.0, seed=5)
            assert np.array_equal(have != 0, want != 0) and np.allclose(have, want, rtol=2e-5, atol=1e-6 * np.abs(want).max())
        e.close()


def test_ray_tracer_on_a_voronoi_grid():
    """mcgpu_rt1_dust_map / mcgpu_rt1_image on a Voronoi grid (k_rt1_dust_map_voro, k_rt1_image_voro) on the xI_scatt the
    SED Monte Carlo left in HBM, against the oracle's integ_ray_dust with the grid's operators; and the face-on ray-traced
    SED against that of the same disk on the cylindrical grid (the tessellation is a sample of it: tens of per cent)."""
    from helpers import sed_model
    for kw in (dict(RT_n_incl=3), dict(lsepar_pola=False)):
        cfg = M.small(**kw)
        m = sed_model(cfg, voronoi_sites=3000, n_thermal=100000)
        e, o = _engine(m, 1e5), _oracle(m, 1e5)
        for lam in (3, 12):
            a = e.run_mono(lam, 30, seed=20 + lam, n_chunks=32)
            ns, Ed, T = a["n_sent"][lam - 1], m.extra["E_disk"][lam - 1], m.extra["Tdust"]
            got, ms = e.dust_map_sed(lam, T, ns, Ed)
            ref = o.dust_map_sed(lam, e.fetch_xI(), T, ns, Ed, n_threads=8)
            assert (ref[:, 0] > 0).all() and ms > 0
            assert np.allclose(got, ref, rtol=1e-9, atol=1e-13 * np.abs(ref).max()), np.abs(got / ref - 1).max()
        img, n_rays, ms = e.dust_map_image(12, T, ns, Ed, 24, 24, 2.2 * cfg.rout, zoom=1.2, l_sym_ima=False, ang_disque=17.3)
        want, nr = o.dust_map_image(12, e.fetch_xI(), T, ns, Ed, 24, 24, 2.2 * cfg.rout, zoom=1.2, l_sym_ima=False,
                                    ang_disque=17.3, n_threads=8)
        assert n_rays == nr and want[0].max() > 0
        assert np.allclose(img, want, rtol=1e-9, atol=1e-13 * np.abs(want).max())
        e.close()
        if cfg.lsepar_pola:   # the same disk on the cylindrical grid
            mc = sed_model(cfg, n_thermal=100000)
            ec = _engine(mc, 1e5)
            ac = ec.run_mono(12, 30, seed=32, n_chunks=32)
            cyl, _ = ec.dust_map_sed(12, mc.extra["Tdust"], ac["n_sent"][11], mc.extra["E_disk"][11])
            ec.close()
            # face-on only: seen inclined, this 3000-site disk hides its inner rim behind its own coarse surface cells -- the
            # Monte Carlo SED of the same grid falls with the inclination in the same way (bins 0..2: 1.1, 0.37, 0.09 of the
            # cylindrical grid's)
            assert abs(got[0, 0] / cyl[0, 0] - 1.0) < 0.5, got[:, 0] / cyl[:, 0]
            assert got[2, 0] < got[1, 0] < got[0, 0]


@pytest.mark.parametrize("grid", ["spherical", "voronoi"])
def test_temperature_and_sed_end_to_end_on_other_grids(grid):
    """test_temperature_and_sed_end_to_end on a spherical and on a Voronoi grid: temperature step, emission tables, SED
    Monte Carlo, ray-traced SED of the dust and of the star -- the whole host sequence on the device against the oracle,
    two independent runs, the reference's gates.  (Voronoi: every third wavelength -- the oracle's side of it is the cost.)"""
    from helpers import OracleBackend
    from mcfost_amd.host import pipeline as P
    n_th, n2, nch = 400000, 800, 32
    if grid == "spherical":
        cfg = M.small(grid_type=2, RT_n_incl=3)
        mg, mc = M.build_model(cfg), M.build_model(cfg)
        lams = list(range(1, mg.n_lambda + 1))
    else:
        cfg = M.small(RT_n_incl=3)
        mg, mc = M.build_voronoi_model(cfg, 3000, seed=3), M.build_voronoi_model(cfg, 3000, seed=3)
        lams = list(range(1, mg.n_lambda + 1, 3))
    li = np.array(lams) - 1
    e = _engine(mg, n_th)
    g = P.temperature_and_sed(P.EngineBackend(e), mg, n_th, n2, lambdas=lams, seed=11, n_chunks=nch)
    e.close()
    c = P.temperature_and_sed(OracleBackend(_oracle(mc, n_th)), mc, n_th, n2, lambdas=lams, seed=23, n_chunks=nch)
    sel = c["Tdust"] > 1.01 * cfg.T_min
    assert np.percentile(np.abs(g["Tdust"][sel] / c["Tdust"][sel] - 1), 75) < 0.05
    fg = P.sed_flux(mg, g["sed_mc"], g["n_sent"])[0].sum(axis=0)[:, li]
    fc = P.sed_flux(mc, c["sed_mc"], c["n_sent"])[0].sum(axis=0)[:, li]
    # two independent runs: a bin of N packets on each side scatters by at least sqrt(2 / N) in the ratio (more: the packets
    # carry weights under forced scattering), and the 75th percentile of |N(0, s)| is 1.15 s -- the reference's 10 % gate
    # is a statement about bins with at least ~500 packets (6.3 % per bin, expected p75 about 7 %; round 3's floor of 200
    # packets put the pure counting noise AT the gate: 0.1013 on one box of round 4)
    ng, nc = g["sed_mc"][4].sum(axis=0)[:, li], c["sed_mc"][4].sum(axis=0)[:, li]
    ok = (nc >= 500) & (ng >= 500)
    assert ok.sum() >= 20 and ((nc >= 50) & (ng >= 50)).sum() > 0.3 * nc.size, ok.sum()
    assert np.percentile(np.abs(fg[ok] / fc[ok] - 1), 75) < 0.10
    ig, ic_ = g["sed_rt"][li, :, 0], c["sed_rt"][li, :, 0]
    assert (ic_ > 0).all()
    assert np.percentile(np.abs(ig / ic_ - 1), 75) < 0.10
    sg, sc = g["sed_rt_stars"][li], c["sed_rt_stars"][li]
    # (the star's flux spans decades over the wavelengths and, on the Voronoi grid, over the inclinations: absolute floor)
    assert (sc > 0).any() and np.allclose(sg, sc, rtol=0.05, atol=1e-4 * sc.max())
    assert (g["n_sent"][li] >= nch * n2).all()
