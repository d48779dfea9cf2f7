"""RCCL executed on a box with ONE GPU.

The reference sums its threads' private accumulators after the loop (thermal_emission.f90:668-670,
dust_transfer.f90:480-489); on several GPUs that sum is ONE all-reduce of the fused accumulator over RCCL
(DESIGN.md section 4).  RCCL refuses a communicator that names a device twice and the pool hands out one GPU, so the
shared-device handle of tests/test_multi_shared_device.py replaces the collective by a kernel.  Here the collective itself
runs, over a world of one rank, through both hosts:

* the library's one-host-thread entry with MCGPU_MULTI_FORCE_RCCL: `ncclCommInitAll(1)`, `ncclGroupStart`, one
  `ncclAllReduce` per buffer a call deposits into (FP64 accumulator with the counters in its tail, xI_scatt as default
  real pairs and as FP64, I_spec / I_spec_star), `ncclGroupEnd` -- the sum over one rank must leave every buffer bit for bit;
* the one-process-per-GPU host: `python -m torch.distributed.run --nproc-per-node 1 bench.py --gpus 1 --force-dist`
  opens the `nccl` process group and all-reduces the fused device buffer (`Engine.allreduce_device`).
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from helpers import sed_model
from mcfost_amd.host import model as M

pytestmark = pytest.mark.gpu


def _forced(model, n_tot):
    from mcfost_amd.engine import MultiEngine
    return MultiEngine(model, n_tot, devices=(0,), force_rccl=True)


def test_flag_combinations():
    """FORCE_RCCL goes with distinct devices only (the shared-device handle has no communicator); unknown bits are refused."""
    import ctypes as C
    from mcfost_amd.engine import load_library
    lib = load_library()
    h = C.c_void_p()
    devs = (C.c_int * 2)(0, 0)
    assert lib.mcgpu_multi_create_ex(C.c_int(1), devs, C.c_uint(3), C.byref(h)) == 3     # MCGPU_ERR_ARG
    assert lib.mcgpu_multi_create_ex(C.c_int(1), devs, C.c_uint(4), C.byref(h)) == 3
    assert lib.mcgpu_multi_create_ex(C.c_int(2), devs, C.c_uint(2), C.byref(h)) == 3     # a device named twice
    assert lib.mcgpu_multi_create_ex(C.c_int(1), devs, C.c_uint(2), C.byref(h)) == 0
    assert lib.mcgpu_multi_rccl_ranks(h) == 0                                            # (opened by the first collective)
    assert lib.mcgpu_multi_destroy(h) == 0


def test_forced_single_rank_all_reduce_of_the_thermal_accumulator(small_model):
    """mcgpu_multi_run_thermal on one device with the forced communicator = mcgpu_run_thermal packet for packet (E_abs, every
    SED array, n_sent, the ten counters through the accumulator's tail and back), also accumulating; RCCL reports one rank,
    one collective per call."""
    from mcfost_amd.engine import Engine
    n = 30000
    e = Engine(small_model, 2 * n)
    prior = e.run_thermal(2000, seed=1)["E_abs"]
    a = e.run_thermal(n, seed=11, frozen=True, E_prior=prior)
    a2 = e.run_thermal(n, seed=12, first_packet=n, frozen=True, accumulate=True)
    e.close()
    me = _forced(small_model, 2 * n)
    assert me.rccl_ranks() == 0
    b = me.run_thermal(n, seed=11, frozen=True, E_prior=prior)
    assert me.rccl_ranks() == 1 and me.reductions() == 1
    b2 = me.run_thermal(n, seed=12, first_packet=n, frozen=True, accumulate=True)
    assert me.reductions() == 2
    me.close()
    for x, y in ((a, b), (a2, b2)):
        # the same packets: integers exact (the counters went through the accumulator's tail as doubles and back); the FP64
        # sums to the order in which a launch's workgroups fold their private grids (two launches of ONE context differ alike)
        assert x["counters"] == y["counters"]
        assert np.array_equal(x["n_sent"], y["n_sent"]) and np.array_equal(x["sed"][4], y["sed"][4])
        assert np.allclose(x["sed"], y["sed"], rtol=1e-9, atol=1e-12 * np.abs(y["sed"]).max())
        assert np.allclose(x["E_abs"], y["E_abs"], rtol=1e-9, atol=1e-11 * y["E_abs"].max())
    assert b2["counters"]["packets"] == 2 * n


def test_forced_single_rank_all_reduce_on_a_3d_grid_with_chunks():
    """The chunked 3D launch (binned deposits) followed by the collective on the same stream."""
    from mcfost_amd.engine import Engine
    m = M.build_model(M.small(n_rad=12, nz=6, n_az=8, l3D=True))
    n = 30000
    e = Engine(m, n)
    e.set_option("deposit", 3)
    e.set_option("deposit_log_mb", 1)
    prior = e.run_thermal(2000, seed=1)["E_abs"]
    a = e.run_thermal(n, seed=8, frozen=True, E_prior=prior)
    e.close()
    me = _forced(m, n)
    me.engines[0].set_option("deposit", 3)
    me.engines[0].set_option("deposit_log_mb", 1)
    b = me.run_thermal(n, seed=8, frozen=True, E_prior=prior)
    assert me.engines[0].get_info("bin_chunks") >= 4 and me.rccl_ranks() == 1
    me.close()
    assert a["counters"] == b["counters"] and np.array_equal(a["sed"][4], b["sed"][4])
    # (two launches of the binned kernel sum a cell's deposits in the order its waves flushed them)
    assert np.allclose(a["E_abs"], b["E_abs"], rtol=1e-9, atol=1e-11 * a["E_abs"].max())


@pytest.mark.parametrize("xi_bytes", [4, 8])
def test_forced_single_rank_all_reduce_of_the_sed_buffers(xi_bytes):
    """mcgpu_multi_run_mono: the grouped all-reduce of [sed | n_sent | counters] and xI_scatt (ncclFloat pairs or
    ncclDouble), then of I_spec + I_spec_star for ray tracing method 2: = the single context, bit for bit."""
    from mcfost_amd.engine import Engine
    m = sed_model(M.small(RT_n_incl=3))
    e = Engine(m, 1e5)
    e.set_xI_precision(xi_bytes)
    a = e.run_mono(5, 40, seed=3, n_chunks=8)
    r = e.run_mono(7, 40, seed=3, n_chunks=8, rt2=(15, 15))
    e.close()
    me = _forced(m, 1e5)
    me.engines[0].set_xI_precision(xi_bytes)
    b = me.run_mono(5, 40, seed=3, n_chunks=8)
    assert me.rccl_ranks() == 1 and me.reductions() == 1
    s = me.run_mono(7, 40, seed=3, n_chunks=8, rt2=(15, 15))
    assert me.reductions() == 2
    me.close()
    assert np.array_equal(a["n_sent_chunk"], b["n_sent_chunk"]) and a["counters"] == b["counters"]
    assert np.array_equal(a["sed"][4], b["sed"][4]) and np.array_equal(a["n_sent"], b["n_sent"])
    assert np.allclose(a["sed"], b["sed"], rtol=1e-9, atol=1e-12 * np.abs(a["sed"]).max())
    # (xI_scatt is summed with atomics in an order that differs from launch to launch)
    assert np.allclose(a["xI_scatt"], b["xI_scatt"], rtol=1e-9 if xi_bytes == 8 else 2e-5,
                       atol=(1e-12 if xi_bytes == 8 else 1e-6) * np.abs(a["xI_scatt"]).max())
    assert r["counters"] == s["counters"] and np.array_equal(r["n_sent_chunk"], s["n_sent_chunk"])
    scale = np.abs(r["I_spec"]).max()
    assert np.allclose(r["I_spec"], s["I_spec"], rtol=1e-9, atol=1e-12 * scale)
    assert np.allclose(r["I_spec_star"], s["I_spec_star"], rtol=1e-9, atol=1e-12 * max(scale, r["I_spec_star"].max()))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_torchrun_world_of_one_all_reduces_the_device_buffer():
    """The one-process-per-GPU host of `bench.py --gpus N` with N = 1 and --force-dist: the launcher starts before any GPU
    call, the rank opens the nccl (= RCCL) process group and every step all-reduces the fused [E_abs | sed | n_sent |
    counters] buffer in HBM."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "1", "--force-dist", "--packets", "2e6",
           "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extra"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=1200, env=env, cwd=root)
    assert out.returncode == 0, (out.stdout[-1000:], out.stderr[-3000:])
    line = json.loads([l for l in out.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["value"] > 0 and "torch.distributed" in line["launcher"]
    assert line["dist"] == {"backend": "nccl", "world_size": 1, "device_allreduces": 3}
    assert line["config"]["packets_per_gpu"] == 2000000 and abs(line["config"]["crossings_per_packet"] - 252) < 5
