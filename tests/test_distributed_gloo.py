"""N > 1 path on CPU: world_size 2 over gloo.  The rank-local packet loop is
stood in for by the oracle (this is a test, so it may); what is under test is
the sharding + fused all-reduce logic of mcfost_amd.distributed."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    """A TCP port nobody listens on right now (a fixed one can still be in TIME_WAIT from the previous test)."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _worker(rank, world, port, n_packets, out_dir):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    import torch.distributed as dist
    from mcfost_amd import distributed as D
    from mcfost_amd.host import model as M
    from oracle import Oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m = M.build_model(M.small())
    orc = Oracle(m, n_packets)
    prior = orc.run_thermal(1000, seed=1)["E_abs"]

    def run_local(n, seed, first_packet, n_replicas, **kw):
        return orc.run_thermal(n, seed=seed, first_packet=first_packet, n_replicas=n_replicas,
                               frozen=True, E_prior=prior)

    res = D.run_thermal_sharded(run_local, n_packets, 42, rank, world)
    np.savez(os.path.join(out_dir, f"r{rank}.npz"), E=res["E_abs"], sed=res["sed"], ns=res["n_sent"],
             cnt=np.array(list(res["counters"].values())))
    dist.destroy_process_group()


def test_two_ranks_equal_one(tmp_path):
    sys.path[:0] = [ROOT]
    from mcfost_amd.host import model as M
    from oracle import Oracle
    n = 3001  # odd: uneven shards
    port = _free_port()
    mp.spawn(_worker, args=(2, port, n, str(tmp_path)), nprocs=2, join=True)
    m = M.build_model(M.small())
    orc = Oracle(m, n)
    prior = orc.run_thermal(1000, seed=1)["E_abs"]
    one = orc.run_thermal(n, seed=42, frozen=True, E_prior=prior)
    r0 = np.load(tmp_path / "r0.npz")
    r1 = np.load(tmp_path / "r1.npz")
    for k in ("E", "sed", "ns", "cnt"):
        assert np.array_equal(r0[k], r1[k])          # every rank holds the global sum
    assert np.array_equal(r0["sed"][4], one["sed"][4]) and np.array_equal(r0["ns"], one["n_sent"])
    assert np.array_equal(r0["cnt"], np.array(list(one["counters"].values())))
    assert np.allclose(r0["E"], one["E_abs"], rtol=1e-12, atol=0)


def _worker_mono(rank, world, port, out_dir):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    import torch.distributed as dist
    from helpers import sed_model
    from mcfost_amd import distributed as D
    from mcfost_amd.host import model as M
    from oracle import Oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m = sed_model(M.small(), n_thermal=20000)
    orc = Oracle(m, 1e5)
    try:   # more ranks than streams: refused on EVERY rank before any collective (nobody is left waiting in one)
        D.run_mono_sharded(orc.run_mono, 9, 5, 1, 77, rank, world)
        raise AssertionError("expected ValueError")
    except ValueError:
        pass
    res = D.run_mono_sharded(orc.run_mono, 9, 5, 13, 77, rank, world)   # 13 streams: uneven split
    # the ray-traced SED of the dust from the all-reduced xI_scatt: every rank holds the same input, so any rank
    # (or each rank for its share of the observers) can compute it
    rt = orc.dust_map_sed(9, res["xI_scatt"], m.extra["Tdust"], res["n_sent"][8], m.extra["E_disk"][8])
    np.savez(os.path.join(out_dir, f"m{rank}.npz"), sed=res["sed"], ns=res["n_sent"], per=res["n_sent_chunk"],
             xI=res["xI_scatt"], cnt=np.array(list(res["counters"].values())), rt=rt)
    dist.destroy_process_group()


def test_sed_mode_streams_shard_over_ranks(tmp_path):
    """SED mode on 2 ranks: the streams are split, every stream still stops at its own packet, and the
    all-reduced result equals the single-rank run stream for stream."""
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    from helpers import sed_model
    from mcfost_amd.host import model as M
    from oracle import Oracle
    port = _free_port()
    mp.spawn(_worker_mono, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    m = sed_model(M.small(), n_thermal=20000)
    one = Oracle(m, 1e5).run_mono(9, 5, seed=77, n_chunks=13)
    r0, r1 = np.load(tmp_path / "m0.npz"), np.load(tmp_path / "m1.npz")
    for k in ("sed", "ns", "per", "xI", "cnt", "rt"):
        assert np.array_equal(r0[k], r1[k])
    rt_one = Oracle(m, 1e5).dust_map_sed(9, one["xI_scatt"], m.extra["Tdust"], one["n_sent"][8], m.extra["E_disk"][8])
    assert (rt_one[:, 0] > 0).all() and np.allclose(r0["rt"], rt_one, rtol=1e-9, atol=0)
    assert np.array_equal(r0["per"], one["n_sent_chunk"])
    assert np.array_equal(r0["sed"][4], one["sed"][4]) and np.array_equal(r0["ns"], one["n_sent"])
    assert np.array_equal(r0["cnt"], np.array(list(one["counters"].values())))
    assert np.allclose(r0["sed"], one["sed"], rtol=1e-12, atol=1e-14)
    assert np.allclose(r0["xI"], one["xI_scatt"], rtol=1e-10, atol=1e-14 * np.abs(one["xI_scatt"]).max())
