"""The library's host side of a launch's tail (mcfost_amd/csrc/host_tail.cpp; mc_tail.hip.h "The last packets on the
host") without a GPU: tests/emu/emu_host_tail.cpp compiles that very file into a test library and hands it EVERY
packet of a frozen-temperature run as a never-started record, so whole packets -- emission, flights, interactions, the
random walk -- run on the host threads with the product's functions, atomics and thread pool.  The oracle is the
checker, here as everywhere; the product does not link it (tests/test_abi_and_host.py)."""
import copy
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from mcfost_amd.host import model as M
from oracle import Oracle
from oracle.binding import N_COUNTERS, _Opts, _p

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), "mcfost_amd", "csrc")
SRC = os.path.join(HERE, "emu", "emu_host_tail.cpp")
LIB = os.path.join(HERE, "emu", "libemu_host_tail.so")


@pytest.fixture(scope="module")
def lib():
    deps = [SRC, os.path.join(HERE, "emu", "emu_conv.h")] + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".cpp"))]
    if (not os.path.exists(LIB)) or os.path.getmtime(LIB) < max(os.path.getmtime(d) for d in deps):
        import __graft_entry__ as g
        subprocess.check_call(["g++"] + g.HOST_CXX_FLAGS + ["-shared", "-o", LIB, SRC])   # (the product's own flags)
    return C.CDLL(LIB)


def run(lib, orc, n, seed, prior, n_threads):
    m = orc.model
    E = np.zeros(m.n_cells)
    sed = np.zeros((9, m.cfg.N_phi, m.cfg.N_thet, m.n_lambda))
    ns = np.zeros(m.n_lambda)
    cnt = np.zeros(N_COUNTERS, np.uint64)
    ms = C.c_double(0.0)
    o = _Opts(seed, 0, n, 1, 1, 0, 1.0)
    rc = lib.emu_host_tail_thermal(C.byref(orc.cm), C.byref(o), _p(prior, C.c_double), _p(E, C.c_double), _p(sed, C.c_double),
                                   _p(ns, C.c_double), _p(cnt, C.c_uint64), n_threads, C.byref(ms))
    assert rc == 0, rc
    return dict(E_abs=E, sed=sed, n_sent=ns, counters=[int(c) for c in cnt], ms=ms.value)


def check(lib, m, n, seed, n_threads=4, rtol=1e-9):
    orc = Oracle(m, n)
    prior = orc.run_thermal(2000, seed=1)["E_abs"]
    a = run(lib, orc, n, seed, prior, n_threads)
    b = orc.run_thermal(n, seed=seed, frozen=True, E_prior=prior, n_threads=4)
    assert a["counters"] == list(b["counters"].values())
    assert np.array_equal(a["n_sent"], b["n_sent"]) and np.array_equal(a["sed"][4], b["sed"][4])
    assert np.allclose(a["E_abs"], b["E_abs"], rtol=rtol, atol=1e-11 * b["E_abs"].max())
    for t in (0, 5, 6, 7, 8):   # Stokes I (the gates of tests/test_gpu_parity.py::_frozen_parity)
        if m.cfg.lsepar_pola and m.cfg.aniso_method == 1:
            assert np.allclose(a["sed"][t], b["sed"][t], rtol=1e-12, atol=1e-9), t
        else:
            assert np.array_equal(a["sed"][t], b["sed"][t]), t
    assert np.allclose(a["sed"][1:4], b["sed"][1:4], rtol=1e-5, atol=1e-5 * max(1.0, np.abs(b["sed"][0]).max()))
    return a, b


@pytest.mark.parametrize("threads", [1, 4, 7])
def test_host_tail_2d(lib, threads):
    m = M.build_model(M.small())
    check(lib, m, 3000, 7, threads)
    check(lib, M.build_model(M.small(aniso_method=2, lsepar_pola=False)), 2000, 10, threads)
    check(lib, M.build_model(M.small(lisotropic=True, lsepar_pola=False)), 2000, 11, threads)


def test_host_tail_dark_zone_and_3d(lib):
    m = M.build_model(M.small())
    md = copy.copy(m)
    dz = np.zeros(md.n_cells, np.uint8)
    kf = md.kappa_factor.copy()
    kf[::md.cfg.n_rad] = 0.0
    dz[np.argsort(kf)[-40:]] = 1
    md.l_dark_zone = dz
    a, _ = check(lib, md, 3000, 12)
    assert a["counters"][7] > 0
    check(lib, M.build_model(M.small(n_rad=12, nz=6, n_az=8, l3D=True)), 3000, 8)


def test_host_tail_walks_like_the_oracle(lib):
    """The modified random walk on the host threads (2D and 3D thick disks): the bisection forms of the walk's searches."""
    from test_mrw import thick_disk, thick_disk_3d, _counters_equal_but_for_parted_packets
    n, seed = 3000, 10
    m = thick_disk()
    orc = Oracle(m, n)
    prior = Oracle(thick_disk(mrw=False), n).run_thermal(20000, seed=1, n_threads=1)["E_abs"] * (n / 20000)
    want = orc.run_thermal(n, seed=seed, frozen=True, E_prior=prior, n_threads=4)
    assert want["counters"]["mrw_walks"] > 100
    got = run(lib, orc, n, seed, prior, 4)
    assert got["counters"] == list(want["counters"].values())
    err = np.abs(got["E_abs"] - want["E_abs"]) / (1e-8 * np.abs(want["E_abs"]) + 1e-10 * want["E_abs"].max())
    assert err.max() < 1.0
    m3 = thick_disk_3d()
    orc3 = Oracle(m3, n)
    prior3 = Oracle(thick_disk_3d(mrw=False), n).run_thermal(20000, seed=1, n_threads=1)["E_abs"] * (n / 20000)
    want3 = orc3.run_thermal(n, seed=seed, frozen=True, E_prior=prior3, n_threads=4)
    got3 = run(lib, orc3, n, seed, prior3, 4)
    gc, wc = dict(zip(want3["counters"].keys(), got3["counters"])), want3["counters"]
    _counters_equal_but_for_parted_packets(gc, wc)
    assert abs(got3["E_abs"].sum() / want3["E_abs"].sum() - 1.0) < 1e-3


def test_host_tail_speed_per_event(lib):
    """What the hand-over is for: a host thread runs an event (crossing or interaction) of ref4.1 in well under the
    1.0-1.6 us a wave needs for a lone packet's (DESIGN.md "k_tail")."""
    m = M.build_model(M.ref41())
    n = 4000
    orc = Oracle(m, n)
    prior = orc.run_thermal(20000, seed=1, n_threads=8)["E_abs"] * (n / 20000)
    a = run(lib, orc, n, 7, prior, 1)
    ev = a["counters"][1] + a["counters"][3] + a["counters"][4]
    ns = a["ms"] * 1e6 / ev
    print("host tail: %.1f ns per event on one thread (%d events)" % (ns, ev))
    assert ns < 400.0
