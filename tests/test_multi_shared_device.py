"""The n_dev > 1 code of the multi-device entry (include/mcgpu.h: mcgpu_multi_*), executed on ONE GPU.

The pool hands out one GPU per box, and RCCL refuses a communicator that names a device twice, so
`mcgpu_multi_create_ex(..., MCGPU_MULTI_SHARED_DEVICE, ...)` opens the n_dev contexts on one device and sums with the
library's own kernel where distinct devices call ncclAllReduce.  Everything else is the code a node with 8 GPUs runs:
the packet shards (dust_transfer.f90:480-489: packets are independent, any thread may run any of them), `n_replicas =
n_dev` for the in-flight temperature (the `* nb_proc` of thermal_emission.f90:668-670), the 1 / n_dev rescale of an
accumulating call, the counters riding in the accumulator's tail, the chunked 3D launch on several contexts, the host
threads of mcgpu_multi_run_mono, and the error path.

Bars: frozen mode = the single-context run packet for packet (counters, n_sent, SED packet counts exact; E_abs rtol
1e-9: the shards sum in another order); accumulate twice = the union; SED streams split = the single run (stopping
packets exact, xI_scatt / I_spec to summation order); an error on context 1 leaves nothing running and the handle usable.
"""
import numpy as np
import pytest

from helpers import mc_similar, sed_model
from mcfost_amd.host import model as M

pytestmark = pytest.mark.gpu


def _engine(model, n_tot):
    from mcfost_amd.engine import Engine
    return Engine(model, n_tot)


def _multi(model, n_tot, n_dev):
    from mcfost_amd.engine import MultiEngine
    return MultiEngine(model, n_tot, devices=(0,) * n_dev, shared_device=True)


def _oracle(model, n_tot):
    from oracle import Oracle
    return Oracle(model, n_tot)


def _same_packets(a, b, rtol=1e-9, oracle=False):
    """Same packets on both sides.  Two device runs: every SED array to the summation order.  Against the oracle: Q, U, V
    carry the default-real trigonometry of update_Stokes (scattering.f90:1218; sincosf against glibc), the tolerance of
    tests/test_gpu_parity.py::_frozen_parity."""
    assert a["counters"] == b["counters"], (a["counters"], b["counters"])
    assert np.array_equal(a["n_sent"], b["n_sent"])
    assert np.array_equal(a["sed"][4], b["sed"][4])
    if oracle:
        for t in (0, 5, 6, 7, 8):
            assert np.allclose(a["sed"][t], b["sed"][t], rtol=1e-12, atol=1e-9), t
        assert np.allclose(a["sed"][1:4], b["sed"][1:4], rtol=1e-5, atol=1e-5 * max(1.0, np.abs(b["sed"][0]).max()))
    else:
        assert np.allclose(a["sed"], b["sed"], rtol=1e-9, atol=1e-12 * np.abs(b["sed"]).max())
    assert np.allclose(a["E_abs"], b["E_abs"], rtol=rtol, atol=1e-11 * b["E_abs"].max())


def test_duplicate_devices_need_the_flag():
    """A production handle (flags = 0) refuses a device named twice; the shared-device handle refuses distinct ones."""
    import ctypes as C
    from mcfost_amd.engine import load_library
    MCGPU_ERR_ARG = 3   # include/mcgpu.h
    lib = load_library()
    h = C.c_void_p()
    devs = (C.c_int * 2)(0, 0)
    assert lib.mcgpu_multi_create(C.c_int(2), devs, C.byref(h)) == MCGPU_ERR_ARG and not h.value
    assert lib.mcgpu_multi_create_ex(C.c_int(2), devs, C.c_uint(2), C.byref(h)) == MCGPU_ERR_ARG   # unknown flag
    assert lib.mcgpu_multi_create_ex(C.c_int(2), devs, C.c_uint(1), C.byref(h)) == 0 and h.value
    assert lib.mcgpu_multi_size(h) == 2
    lib.mcgpu_multi_reductions.restype = C.c_uint64
    assert lib.mcgpu_multi_reductions(h) == 0 and lib.mcgpu_multi_rccl_ranks(h) == 0
    lib.mcgpu_multi_destroy(h)


@pytest.mark.parametrize("n_dev", [2, 4, 3])
def test_thermal_2d_shards_equal_the_single_context_run(small_model, n_dev):
    """2D, frozen: n_dev shards + the reduction = one context running every packet -- and the oracle's run.  n = 30001
    is not a multiple of any n_dev (ragged shards); n_dev = 3: the 1 / n_dev rescale is not a power of two."""
    n = 30001
    o = _oracle(small_model, n)
    prior = o.run_thermal(2000, seed=1)["E_abs"]
    e = _engine(small_model, n)
    a = e.run_thermal(n, seed=11, frozen=True, E_prior=prior)
    e.close()
    me = _multi(small_model, n, n_dev)
    b = me.run_thermal(n, seed=11, frozen=True, E_prior=prior)
    assert me.reductions() == 1 and me.rccl_ranks() == 0
    # every context holds the totals afterwards, counters included (what the next temperature iteration starts from)
    for x in me.engines:
        f = x.fetch()
        assert f["counters"] == b["counters"] and np.array_equal(f["E_abs"], b["E_abs"])
    me.close()
    _same_packets(b, a)
    assert b["counters"]["packets"] == n
    _same_packets(b, o.run_thermal(n, seed=11, frozen=True, E_prior=prior, n_threads=8), oracle=True)


@pytest.mark.parametrize("n_dev", [2, 4])
def test_accumulate_twice_is_the_union(small_model, n_dev):
    """Two accumulating calls on n_dev contexts (the second first scales every context's totals by 1 / n_dev and clears
    the counters of contexts > 0) = one call on the union of their packets; a third, non-accumulating call starts over."""
    n = 20000
    o = _oracle(small_model, 2 * n)
    prior = o.run_thermal(2000, seed=1)["E_abs"]
    me = _multi(small_model, 2 * n, n_dev)
    first = me.run_thermal(n, seed=11, first_packet=0, frozen=True, E_prior=prior)
    ab = me.run_thermal(n, seed=11, first_packet=n, frozen=True, accumulate=True)
    assert me.reductions() == 2
    again = me.run_thermal(n, seed=11, first_packet=0, frozen=True)
    me.close()
    e = _engine(small_model, 2 * n)
    u = e.run_thermal(2 * n, seed=11, frozen=True, E_prior=prior)
    e.close()
    _same_packets(ab, u)
    assert u["counters"]["packets"] == 2 * n
    _same_packets(again, first, rtol=1e-12)


def test_chunked_3d_launch_on_several_contexts():
    """3D grids: every context runs its shard through the binned-deposit chunks (plan -> packets -> fold, carried
    packets, tail kernel), all asynchronous on its own stream, and the reduction waits for all of them: a small log
    forces many chunks per context.  Frozen: = the single context = the oracle, packet for packet."""
    m = M.build_model(M.small(n_rad=12, nz=6, n_az=8, l3D=True))
    n = 30000
    o = _oracle(m, n)
    prior = o.run_thermal(2000, seed=1)["E_abs"]
    ref = o.run_thermal(n, seed=8, frozen=True, E_prior=prior, n_threads=8)
    for log_mb, n_dev in ((0, 2), (1, 2), (1, 4)):
        me = _multi(m, n, n_dev)
        for x in me.engines:
            x.set_option("deposit", 3)
            x.set_option("deposit_log_mb", log_mb)
        a = me.run_thermal(n, seed=8, frozen=True, E_prior=prior)
        if log_mb:
            assert all(x.get_info("bin_chunks") >= 4 for x in me.engines)
        a2 = me.run_thermal(n, seed=9, first_packet=n, frozen=True, accumulate=True)
        me.close()
        _same_packets(a, ref, oracle=True)
        assert a2["counters"]["packets"] == 2 * n
        b2 = o.run_thermal(n, seed=9, first_packet=n, frozen=True, E_prior=prior, n_threads=8)
        assert np.allclose(a2["E_abs"], ref["E_abs"] + b2["E_abs"], rtol=1e-9, atol=1e-11 * a2["E_abs"].max())


def test_live_mode_on_several_contexts_gives_the_single_context_temperature(small_model):
    """Live mode (the reference algorithm): each context scales its local partial sum by n_replicas = n_dev
    (thermal_emission.f90:670); the temperature agrees with the single context's within the reference's gate."""
    n = 400000
    e = _engine(small_model, n)
    Ta = e.temp_finale(e.run_thermal(n, seed=5)["E_abs"])
    me = _multi(small_model, n, 4)
    r = me.run_thermal(n, seed=6)
    Tb = me.engines[0].temp_finale(r["E_abs"])
    me.close()
    e.close()
    assert r["counters"]["packets"] == n and r["n_sent"].sum() == n
    ok, p75 = mc_similar(Ta, Tb, 0.05, mask_threshold=1.01 * small_model.cfg.T_min)
    assert ok, p75


def test_sed_streams_split_over_contexts():
    """mcgpu_multi_run_mono: the streams of one wavelength split into contiguous ranges, one host thread per context,
    one reduction of [sed | n_sent | counters] and one of xI_scatt (method 1) or I_spec + I_spec_star (method 2, which
    round 3 left per-device): = the single context's run.  Then an accumulating second call = the sum of two runs, and
    a method-1 call after a method-2 call does not reduce stale I_spec again."""
    m = sed_model(M.small(RT_n_incl=3))
    e = _engine(m, 1e5)
    a = e.run_mono(5, 40, seed=3, n_chunks=8)
    a2 = e.run_mono(5, 40, seed=4, n_chunks=8, accumulate=True)
    r = e.run_mono(7, 40, seed=3, n_chunks=8, rt2=(15, 15))
    e.close()
    for n_dev in (2, 4):
        me = _multi(m, 1e5, n_dev)
        b = me.run_mono(5, 40, seed=3, n_chunks=8)
        assert np.array_equal(a["n_sent_chunk"], b["n_sent_chunk"]) and a["counters"] == b["counters"]
        assert np.array_equal(a["sed"][4], b["sed"][4]) and np.array_equal(a["n_sent"], b["n_sent"])
        assert np.allclose(a["xI_scatt"], b["xI_scatt"], rtol=1e-9, atol=1e-12 * np.abs(a["xI_scatt"]).max())
        b2 = me.run_mono(5, 40, seed=4, n_chunks=8, accumulate=True)
        assert a2["counters"] == b2["counters"] and np.array_equal(a2["sed"][4], b2["sed"][4])
        assert np.allclose(a2["xI_scatt"], b2["xI_scatt"], rtol=1e-9, atol=1e-12 * np.abs(a2["xI_scatt"]).max())
        # method 2 on the same handle: I_spec summed over the contexts (every context then holds the global field)
        s = me.run_mono(7, 40, seed=3, n_chunks=8, rt2=(15, 15))
        assert np.array_equal(r["n_sent_chunk"], s["n_sent_chunk"]) and r["counters"] == s["counters"]
        scale = np.abs(r["I_spec"]).max()
        assert np.allclose(r["I_spec"], s["I_spec"], rtol=1e-9, atol=1e-12 * scale)
        assert np.allclose(r["I_spec_star"], s["I_spec_star"], rtol=1e-9, atol=1e-12 * max(scale, r["I_spec_star"].max()))
        for x in me.engines[1:]:
            Ix, Isx = x.fetch_I_spec()
            assert np.array_equal(Ix, s["I_spec"]) and np.array_equal(Isx, s["I_spec_star"])
        # ... and xI_scatt is untouched by that call: it still holds the accumulated sums of the two method-1 calls
        assert np.array_equal(me.engines[0].fetch_xI(), b2["xI_scatt"])
        me.close()
    # more contexts than streams: refused before anything runs
    from mcfost_amd.engine import McgpuError
    me = _multi(m, 1e5, 4)
    with pytest.raises(McgpuError):
        me.run_mono(5, 40, seed=3, n_chunks=3)
    me.close()


def test_an_error_on_one_context_drains_the_others(small_model):
    """Context 1 cannot launch (frozen mode without its prior) after context 0 already has: the call returns that
    context's error, nothing is left running (the handle is immediately usable), and the next good call is exact."""
    from mcfost_amd.engine import McgpuError
    n = 20000
    o = _oracle(small_model, n)
    prior = o.run_thermal(2000, seed=1)["E_abs"]
    me = _multi(small_model, n, 2)
    me.engines[0].set_E_prior(prior)          # ... but not on context 1
    with pytest.raises(McgpuError) as ex:
        me.run_thermal(n, seed=11, frozen=True)
    assert "device 1" in str(ex.value) and "mcgpu_set_E_prior" in str(ex.value)
    good = me.run_thermal(n, seed=11, frozen=True, E_prior=prior)
    me.close()
    _same_packets(good, o.run_thermal(n, seed=11, frozen=True, E_prior=prior, n_threads=8), oracle=True)


def test_bench_dry_run_in_library_mode_with_two_contexts():
    """`python bench.py --gpus 2 --shared-device`: the bench's library mode (ONE process, mcgpu_multi_*) on two contexts of
    this box's one GPU -- the line an 8-GPU node's single-process run would print, dry."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--shared-device", "--packets", "2e6",
                          "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extra"],
                         capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["value"] > 0 and "shared-device" in line["launcher"]
    assert line["config"]["packets_per_gpu"] == 2000000 and abs(line["config"]["crossings_per_packet"] - 252) < 5


def test_sed_step_sharded_by_wavelength_equals_the_single_context():
    """`mcgpu_multi_run_sed`: run_sed_mc's loop over the wavelengths (dust_transfer.f90:899-1027) is the partition -- every
    context takes whole wavelengths (emission tables from Tdust, scout / commit passes, the ray-traced SED of the dust) and
    only their results travel; no xI_scatt is reduced, nothing is summed across contexts.  With 2 and 3 contexts a
    wavelength's results are the single context's: stopping packets, counters and SED packet counts exactly, the sums to
    the order of one context's own atomic additions; the longest wavelengths (by the cost hint) are dealt out first."""
    m = sed_model(M.small(RT_n_incl=2))
    T = m.extra["Tdust"]
    lams = [3, 8, 12, 17, 20, 22]
    ref = None
    for n_dev in (1, 2, 3):
        me = _multi(m, 1e5, n_dev)
        costs = [1.0, 5.0, 2.0, 4.0, 3.0, 6.0]
        r = me.run_sed(lams, 30, T, seeds=[100 + l for l in lams], costs=costs, n_chunks=8, l_sym_ima=True)
        me.close()
        assert sorted(set(r["device_of"].tolist())) == list(range(n_dev))
        if n_dev == 3:   # longest first, each to the least loaded: 6 -> 0, 5 -> 1, 4 -> 2, then 3 -> 2, 2 -> 1, 1 -> 0
            assert r["device_of"].tolist() == [0, 1, 1, 2, 2, 0]
        assert all(c["packets"] > 0 for c in r["counters"]) and np.all(r["n_sent"] > 0) and np.all(r["seconds"] > 0)
        if ref is None:
            ref = r
            # ... and the single context's results are those of the per-wavelength calls a host makes today
            e = _engine(m, 1e5)
            e.set_rt1()
            for i, lam in enumerate(lams):
                td = e.repartition_energie(lam, T, fetch=False)
                a = e.run_mono(lam, 30, seed=100 + lam, n_chunks=8, fetch_xI=False, device_tables=td)
                assert a["counters"] == r["counters"][i] and a["n_sent"][lam - 1] == r["n_sent"][i]
                assert np.array_equal(a["sed"][4][..., lam - 1], r["sed"][i, 4])
                assert abs(td["E_disk"] - r["E_disk"][i]) <= 1e-12 * abs(td["E_disk"])
                s, _ = e.dust_map_sed(lam, T, a["n_sent"][lam - 1], td["E_disk"], l_sym_ima=True)
                assert np.allclose(s, r["sed_rt"][i], rtol=1e-9, atol=1e-12 * np.abs(s).max())
            e.close()
            continue
        assert r["counters"] == ref["counters"]
        assert np.array_equal(r["n_sent"], ref["n_sent"]) and np.array_equal(r["sed"][:, 4], ref["sed"][:, 4])
        assert np.allclose(r["sed"], ref["sed"], rtol=1e-9, atol=1e-12 * np.abs(ref["sed"]).max())
        assert np.allclose(r["E_disk"], ref["E_disk"], rtol=1e-12)
        assert np.allclose(r["sed_rt"], ref["sed_rt"], rtol=1e-9, atol=1e-12 * np.abs(ref["sed_rt"]).max())
