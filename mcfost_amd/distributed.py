"""Multi-GPU sharding of the packet loop: one process per GPU.

The reference runs packets on independent OpenMP threads with private
accumulators that are summed after the loop (``dust_transfer.f90:480-489``,
``thermal_emission.f90:668``).  The same structure over ranks: the packet id
range is split into disjoint contiguous shards (the per-packet Philox streams
make the result independent of the split), every rank holds a replica of the
tables, and ONE all-reduce (RCCL over xGMI when the backend is ``nccl``) sums
the fused accumulator ``[E_abs | sed | n_sent | counters]`` (the ten event
counters ride along as doubles, exact below 2^53) per temperature iteration.  The in-flight temperature uses the local partial sum
times ``world_size`` -- the reference's ``* nb_proc``
(``thermal_emission.f90:670``).
"""
from __future__ import annotations

import numpy as np


def shard_packets(n_packets: int, rank: int, world_size: int):
    """Contiguous, disjoint, exhaustive split: returns (first_packet, count)."""
    base, rem = divmod(int(n_packets), int(world_size))
    count = base + (1 if rank < rem else 0)
    first = rank * base + min(rank, rem)
    return first, count


def pack_results(res):
    """[E_abs | sed | n_sent | counters] as ONE float64 vector (the layout of the device's fused accumulator)."""
    cnt = np.array(list(res["counters"].values()), dtype=np.float64)
    assert np.all(cnt < 2.0 ** 53)
    return np.concatenate([res["E_abs"].ravel(), res["sed"].ravel(), res["n_sent"].ravel(), cnt])


def unpack_results(acc, like):
    n_c, n_s, n_n = like["E_abs"].size, like["sed"].size, like["n_sent"].size
    cnt = acc[n_c + n_s + n_n:]
    return dict(E_abs=acc[:n_c].copy(), sed=acc[n_c:n_c + n_s].reshape(like["sed"].shape).copy(),
                n_sent=acc[n_c + n_s:n_c + n_s + n_n].copy(),
                counters=dict(zip(like["counters"].keys(), (int(round(c)) for c in cnt))))


def allreduce_host(res, group=None):
    """Sum host-side results over ranks with ONE all-reduce (any torch.distributed backend)."""
    import torch
    import torch.distributed as dist

    t_acc = torch.from_numpy(pack_results(res))
    dist.all_reduce(t_acc, op=dist.ReduceOp.SUM, group=group)
    return unpack_results(t_acc.numpy(), res)


def run_thermal_sharded(run_local, n_packets: int, seed: int, rank: int, world_size: int,
                        reduce=allreduce_host, **kw):
    """One temperature iteration on ``world_size`` ranks.

    ``run_local(n_packets, seed=, first_packet=, n_replicas=, **kw)`` is this
    rank's packet loop (``Engine.run_thermal`` on a GPU rank).  Returns the
    globally summed result on every rank.
    """
    first, count = shard_packets(n_packets, rank, world_size)
    res = run_local(count, seed=seed, first_packet=first, n_replicas=float(world_size), **kw)
    if world_size == 1:
        return res
    out = reduce(res)
    out["kernel_ms"] = res.get("kernel_ms")
    return out


def run_thermal_device(engine, n_packets: int, seed: int, rank: int, world_size: int, **kw):
    """GPU path: the accumulators stay in HBM; the all-reduce runs on the fused
    device buffer (RCCL).  Returns (result dict after reduction, kernel ms)."""
    import torch
    import torch.distributed as dist

    first, count = shard_packets(n_packets, rank, world_size)
    engine.launch_thermal(count, seed=seed, first_packet=first, n_replicas=float(world_size), **kw)
    ms = engine.sync()
    if world_size > 1:
        engine.allreduce_device(lambda t: dist.all_reduce(t, op=dist.ReduceOp.SUM))
    out = engine.fetch()
    out["kernel_ms"] = ms
    return out


# ---------------------------------------------------------------------------
# SED mode: the n_photons_loop streams of a wavelength are independent and carry
# their own stopping rule (dust_transfer.f90:525-553), so they shard as they are.
# ---------------------------------------------------------------------------
def shard_streams(n_chunks: int, rank: int, world_size: int):
    """Contiguous range of stream ids of this rank: (first_chunk, count)."""
    return shard_packets(n_chunks, rank, world_size)


def run_mono_sharded(run_local, lam: int, n_photons2: int, n_chunks: int, seed: int, rank: int, world_size: int,
                     group=None, **kw):
    """One wavelength of the SED Monte Carlo on ``world_size`` ranks.

    ``run_local(lam, n_photons2, seed=, n_chunks=, first_chunk=, **kw)`` is this rank's loop
    (``Engine.run_mono``).  Every rank runs its own streams to their own stopping packets; ONE
    all-reduce sums ``[sed | n_sent | xI_scatt | counters | packets sent per stream]`` (the integer parts as
    doubles, exact below 2^53; the per-stream counts occupy disjoint ranges, so their sum is a gather).  The
    result equals the single-rank run stream for stream.  Needs ``world_size <= n_chunks`` (checked on every
    rank before any collective, so that no rank is left waiting in one)."""
    import torch
    import torch.distributed as dist

    if world_size > n_chunks:
        raise ValueError("run_mono_sharded: %d ranks for %d streams (world_size must be <= n_chunks)"
                         % (world_size, n_chunks))
    first, count = shard_streams(n_chunks, rank, world_size)
    res = run_local(lam, n_photons2, seed=seed, n_chunks=count, first_chunk=first, **kw)
    if world_size == 1:
        return res
    parts = [res["sed"].ravel(), res["n_sent"].ravel()]
    has_xI = "xI_scatt" in res
    if has_xI:
        parts.append(res["xI_scatt"].ravel())
    n_f = sum(p.size for p in parts)
    cnt = np.array(list(res["counters"].values()), dtype=np.float64)
    per = np.zeros(n_chunks)
    per[first:first + count] = res["n_sent_chunk"].astype(np.float64)
    assert np.all(cnt < 2.0 ** 53) and np.all(per < 2.0 ** 53)
    acc = torch.from_numpy(np.concatenate(parts + [cnt, per]))
    dist.all_reduce(acc, op=dist.ReduceOp.SUM, group=group)
    a = acc.numpy()
    n_s, n_n, n_c = res["sed"].size, res["n_sent"].size, cnt.size
    out = dict(sed=a[:n_s].reshape(res["sed"].shape).copy(), n_sent=a[n_s:n_s + n_n].copy(),
               n_sent_chunk=np.rint(a[n_f + n_c:]).astype(np.uint64),
               counters=dict(zip(res["counters"].keys(), (int(round(c)) for c in a[n_f:n_f + n_c]))))
    if has_xI:
        out["xI_scatt"] = a[n_s + n_n:n_f].reshape(res["xI_scatt"].shape).copy()
    return out
