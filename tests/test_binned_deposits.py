"""Binned deposits (mcfost_amd/csrc/mc_binned.hip.h) and chunks without tails (mc_roles.hip.h): the thermal step of 3D
cylindrical grids writes (cell, value) records to a log in HBM that is folded into E_abs between the chunks of a run,
and a chunk hands its unfinished packets to the next one.  save_radiation_field (radiation_field.f90:53) only changes
its summation order, and which launch runs a packet cannot matter (counter-based random numbers), so:
  * frozen mode: counters, n_sent, SED packet counts bit-exact against the oracle, E_abs rtol 1e-9 -- also with a log so
    small that the run takes many chunks and overflows its regions (the graceful path: plain atomics);
  * binned == HBM atomics on the same packets;
  * live mode (in-flight temperature from the folded grid, scaled by the packets started since): statistically the
    temperature of the atomic path.
"""
import numpy as np
import pytest

from helpers import mc_similar
from mcfost_amd.host import model as M

pytestmark = pytest.mark.gpu


def _engine(model, n_tot):
    from mcfost_amd.engine import Engine
    return Engine(model, n_tot)


def _oracle(model, n_tot):
    from oracle import Oracle
    return Oracle(model, n_tot)


def _same_packets(a, b, rtol=1e-9):
    assert a["counters"] == b["counters"]
    assert np.array_equal(a["n_sent"], b["n_sent"])
    assert np.array_equal(a["sed"][4], b["sed"][4])
    assert np.allclose(a["E_abs"], b["E_abs"], rtol=rtol, atol=1e-11 * b["E_abs"].max())


@pytest.mark.parametrize("log_mb", [0, 1])
def test_binned_frozen_parity_against_the_oracle(log_mb):
    """log_mb = 1: 1365 blocks for ~40 buckets x 256 workgroups -> chunks of 1024 packets, most blocks overflow."""
    m = M.build_model(M.small(n_rad=12, nz=6, n_az=8, l3D=True))
    n = 30000
    e, o = _engine(m, n), _oracle(m, n)
    prior = o.run_thermal(2000, seed=1)["E_abs"]
    e.set_option("deposit", 3)
    e.set_option("deposit_log_mb", log_mb)
    a = e.run_thermal(n, seed=8, frozen=True, E_prior=prior)
    b = o.run_thermal(n, seed=8, frozen=True, E_prior=prior, n_threads=8)
    _same_packets(a, b)
    assert e.get_info("bin_buckets") >= 16
    if log_mb:
        assert e.get_info("bin_chunks") >= 10
    # a second launch on the same context (the log, its plan and the carry buffers are reused), accumulating
    a2 = e.run_thermal(n, seed=9, first_packet=n, frozen=True, E_prior=prior, accumulate=True)
    b2 = o.run_thermal(n, seed=9, first_packet=n, frozen=True, E_prior=prior, n_threads=8)
    assert np.allclose(a2["E_abs"], a["E_abs"] + b2["E_abs"], rtol=1e-9, atol=1e-11 * a2["E_abs"].max())
    assert a2["counters"]["packets"] == 2 * n
    e.close()


def test_binned_equals_atomics_on_the_baseline_grid():
    """ref4.1_3D with 12 azimuths (the true radial / vertical grid), polarised, frozen: the two deposit paths run the
    same packets; a mid-sized log gives several chunks with carried packets."""
    m = M.build_model(M.ref41_3d(n_az=12))
    n = 300000
    o = _oracle(m, n)
    prior = o.run_thermal(20000, seed=1)["E_abs"]
    res = {}
    for dep, mb in ((1, 0), (3, 0), (3, 64)):
        e = _engine(m, n)
        e.set_option("deposit", dep)
        e.set_option("deposit_log_mb", mb)
        res[(dep, mb)] = e.run_thermal(n, seed=63, frozen=True, E_prior=prior)
        if mb:
            assert e.get_info("bin_chunks") >= 4
        e.close()
    _same_packets(res[(3, 0)], res[(1, 0)], rtol=1e-10)
    _same_packets(res[(3, 64)], res[(1, 0)], rtol=1e-10)


def test_binned_live_mode_gives_the_temperature_of_the_atomic_path():
    m = M.build_model(M.ref41_3d(n_az=12))
    n = 4_000_000
    T = {}
    for dep in (1, 3):
        e = _engine(m, n)
        e.set_option("deposit", dep)
        if dep == 3:
            e.set_option("deposit_log_mb", 512)   # several chunks: the in-flight temperature lags by one of them
        a = e.run_thermal(n, seed=11 + dep)
        c = a["counters"]
        assert c["packets"] == n and c["escaped"] + c["killed_star"] == n and a["n_sent"].sum() == n
        T[dep] = e.temp_finale(a["E_abs"])
        if dep == 3:
            assert e.get_info("bin_chunks") >= 3
        e.close()
    sel = (T[1] > 1.5 * m.cfg.T_min) & (T[3] > 1.5 * m.cfg.T_min)
    ok, p75 = mc_similar(T[1][sel], T[3][sel], 0.05)
    assert ok, p75
    # no systematic offset beyond the noise of the mean
    assert abs(np.mean(T[3][sel] / T[1][sel]) - 1.0) < 0.01


def test_tail_kernel_frozen_parity_and_the_automatic_threshold():
    """mc_tail.hip.h through the C-ABI: a role kernel that hands its last packets -- or, with a huge threshold, every
    packet the moment the work counter runs out -- to the tail kernel returns the oracle's counters and sums, on 2D
    (LDS grid; dark zone; HG), 3D (binned deposits) and with the random walk; option "tail" = -1 picks the threshold
    from the model's optical thickness (first launch) and from the last launch's interactions per packet."""
    import copy
    small = M.build_model(M.small())
    md = copy.copy(small)
    dz = np.zeros(md.n_cells, np.uint8)
    kf = md.kappa_factor.copy()
    kf[::md.cfg.n_rad] = 0.0
    dz[np.argsort(kf)[-40:]] = 1
    md.l_dark_zone = dz
    models = [small, md, M.build_model(M.small(aniso_method=2, lsepar_pola=False)),
              M.build_model(M.small(n_rad=12, nz=6, n_az=8, l3D=True))]
    n = 30000
    host_ran = 0
    for m in models:
        o = _oracle(m, n)
        prior = o.run_thermal(2000, seed=1)["E_abs"]
        b = o.run_thermal(n, seed=17, frozen=True, E_prior=prior, n_threads=8)
        # (tail_where: 1 = k_tail finishes every packet; 2 = k_tail thins the tail out and the library's host threads --
        # host_tail.cpp, the device source compiled for the CPU -- finish the last packets; with 4 packets left to the host
        # most are handed over in the middle of their lives, with 100000 every packet of the tail runs on the host)
        for thr, where, host_pk in ((0, 0, 0), (40, 1, 0), (100000, 1, 0), (40, 2, 0), (40, 2, 4), (100000, 2, 60000 if m is small else 64), (100000, 0, 0)):
            e = _engine(m, n)
            e.set_option("tail", thr)
            e.set_option("tail_where", where)
            e.set_option("tail_host_packets", host_pk)
            e.set_option("host_threads", 3 if host_pk == 4 else 0)
            if m.cfg.l3D:
                e.set_option("deposit", 3)   # (this grid would fit into LDS: binned deposits, whose last chunk hands over to k_tail<3D>)
            a = e.run_thermal(n, seed=17, frozen=True, E_prior=prior)
            w = e.get_info("tail_where")   # (0: the launch had no tail kernel)
            assert w == (0 if thr == 0 else (1 if where == 1 else 2))
            if w:
                if where != 1:
                    # (how many packets reach k_tail, and so the host, is the schedule's business: a workgroup hands over what it
                    # has LEFT when its work runs out -- on a small launch that can be nothing at all)
                    hp = e.get_info("tail_host_packets")
                    assert hp <= (host_pk if host_pk else 8 * max(e.get_info("tail_host_threads"), 32))
                    assert (e.get_info("tail_host_events") > 0) == (hp > 0)
                    host_ran += hp
            e.close()
            _same_packets(a, b)
    assert host_ran > 100   # ... but over the four models and their settings the host threads did run packets
    # the automatic choice
    e = _engine(M.build_model(M.pascucci()), 1e6)
    assert e.get_info("tail_threshold") == 0            # thin: no hand-over (its launch has no tail)
    e.close()
    e = _engine(M.build_model(M.ref41()), 1e6)
    assert e.get_info("tau_midplane") > 1000 and e.get_info("tail_threshold") == 48
    e.run_thermal(200000, seed=1)
    assert e.get_info("tail_threshold") == 48           # ... confirmed by the launch: ~11 interactions per packet
    e.close()
