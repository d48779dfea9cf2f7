"""N > 1 path on CPU: world_size 2 over gloo.  The rank-local packet loop is
stood in for by the oracle (this is a test, so it may); what is under test is
the sharding + fused all-reduce logic of mcfost_amd.distributed."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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
    port = 29500 + (os.getpid() % 1000)
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
