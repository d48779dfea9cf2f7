#!/usr/bin/env python3
"""bench.py -- photon packets/s of the thermal Monte Carlo packet loop on the
BASELINE workload (ref4.1.para 2D cylindrical disk, 1e8 packets per GPU).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of mc_photon_loop (dust_transfer.f90:439-572) for the
temperature step: `--packets` packets per GPU through the persistent HIP
kernel with all tables resident in HBM, followed (N > 1) by the single RCCL
all-reduce of the fused accumulator [E_abs | sed | n_sent].  Weak scaling:
per-GPU work is fixed.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

BYTES_PER_CROSSING = 28.0    # kappa_factor 8 B + next-cell id 4 B + E_abs RMW 16 B
BYTES_PER_INTERACTION = 8.0  # E_abs read for Temp_LTE
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s


def _quota_cores():
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:  # container CPU quota (cgroup v2): more threads than the quota only get throttled
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, int(round(int(quota) / int(period)))))
    except Exception:
        pass
    return cores


def pmc_traffic(config, n_local, world):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes of THIS command
    (`tools/collect_profiles.sh`: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate runs of
    `python bench.py --steps 1 --warmup 0`, summarised in profiles/r01_pmc_{fetch,write}.json).  Counters are in
    KB; FETCH_SIZE is doubled for gfx950 (MI355X guide, HBM section).  None when the command differs."""
    if config != "ref41" or world != 1 or int(n_local) != 100000000:
        return None
    try:
        f = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_fetch.json")))
        w = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_write.json")))
        fk = f["pmc_k_thermal_sum_over_launches"]["FETCH_SIZE"] / f["pmc_k_thermal_launches"]["FETCH_SIZE"]
        wk = w["pmc_k_thermal_sum_over_launches"]["WRITE_SIZE"] / w["pmc_k_thermal_launches"]["WRITE_SIZE"]
        return (2.0 * fk + wk) * 1024.0
    except Exception:
        return None


def cpu_baseline(model, n_total, target_s=15.0):
    """The CPU restatement (oracle, kind "port") timed on this box's host cores
    on a bounded sample of the same workload."""
    from oracle import Oracle
    cores = _quota_cores()
    orc = Oracle(model, n_total)
    t = time.perf_counter()
    orc.run_thermal(20000 * cores, seed=99, n_threads=cores)
    rate = 20000 * cores / (time.perf_counter() - t)
    n = int(max(20000 * cores, min(rate * target_s, 5e7)))
    orc = Oracle(model, n)   # a self-consistent run of n packets: its own packet luminosity, its own temperature
    t = time.perf_counter()
    res = orc.run_thermal(n, seed=100, n_threads=cores)
    dt = time.perf_counter() - t
    base = dict(value=n / dt, unit="packets/s", cores=cores, kind="port",
                sample="%d packets of the same thermal workload, %d OpenMP threads, %.1f s; "
                       "%.1f crossings/packet" % (n, cores, dt, res["counters"]["crossings"] / n))
    return base, orc.temp_finale(res["E_abs"]), n


def tdust_parity(T_gpu, n_gpu, T_cpu, n_cpu, T_min):
    """Temperature of the GPU step against the CPU port's (independent noise): relative RMS over the cells with
    T > 1.01 T_min, the reference's own gate p75(|dT|/T) (test_suite/test_mcfost.py:46-57,88: < 5 %), and the
    tolerance 3 sigma_MC with sigma_MC(N) = 1.7 % sqrt(1.28e5 / N) per run."""
    import numpy as np
    sel = (T_cpu > 1.01 * T_min) & (T_gpu > 1.01 * T_min)
    rel = (T_gpu[sel] - T_cpu[sel]) / T_cpu[sel]
    # sigma_MC was measured on the 7000-cell ref4.1 grid; the noise per cell scales with sqrt(cells / packets)
    sigma = 0.017 * float(np.sqrt(1.28e5 / n_gpu + 1.28e5 / n_cpu)) * float(np.sqrt(max(T_cpu.size, 7000) / 7000.0))
    rms, p75 = float(np.sqrt(np.mean(rel ** 2))), float(np.percentile(np.abs(rel), 75))
    return dict(rel_rms=rms, p75=p75, tolerance_rel_rms=3.0 * sigma, cells=int(sel.sum()), ok=bool(rms <= 3.0 * sigma),
                reference_gate_p75_below_5pct=bool(p75 < 0.05))


def bench_sed(args, world, rank, local_rank):
    """SED mode on the ref4.1 grid (not the headline metric): one step = the SED Monte Carlo
    (mcgpu_run_mono: scout + commit passes, ray-tracing deposits) of the listed wavelengths, every stream
    asked for packets/128/len(wavelengths) packets in the stop bin; streams sharded over the ranks."""
    import torch
    import torch.distributed as dist
    from mcfost_amd import distributed as D
    from mcfost_amd.engine import Engine
    from mcfost_amd.host import model as M

    m = M.build_model(M.ref41())
    e = Engine(m, 5e6, device=local_rank)
    T = e.temp_finale(e.run_thermal(5_000_000, seed=3)["E_abs"])
    M.repartition_energie(m, T)
    e.close()
    eng = Engine(m, 5e6, device=local_rank)
    lams = [int(x) for x in args.sed_lambdas.split(",")]
    n_streams = m.cfg.n_photons_loop * world                 # weak scaling: 128 streams per GPU
    first, count = D.shard_streams(n_streams, rank, world)
    # packets in the stop bin per stream so that a step sends about --packets packets per GPU (1 in ~11 lands there)
    n2 = max(10, int(args.packets / 11.0 / m.cfg.n_photons_loop / len(lams)))

    def step(i):
        sent = 0
        for lam in lams:
            r = eng.run_mono(lam, n2, seed=100 + i, n_chunks=count, first_chunk=first, fetch_xI=False)
            sent += int(r["n_sent_chunk"].sum())
            if world > 1:   # one all-reduce of [sed | n_sent] + xI_scatt per wavelength
                acc, cnt = eng.device_accumulators()
                dist.all_reduce(acc)
                dist.all_reduce(eng.device_xI())
                torch.cuda.current_stream().synchronize()
        return sent

    for i in range(args.warmup):
        step(-1 - i)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sent = 0
    for i in range(args.steps):
        sent += step(i)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tot = torch.tensor([float(sent), dt], dtype=torch.float64, device="cuda")
    if world > 1:
        s2 = tot.clone()
        dist.all_reduce(s2, op=dist.ReduceOp.SUM)
        dist.all_reduce(tot, op=dist.ReduceOp.MAX)
        sent_all, dt = float(s2[0].item()), float(tot[1].item())
    else:
        sent_all = float(sent)
    if rank == 0:
        cnt = eng.fetch()["counters"]
        cross_pp = cnt["crossings"] / max(cnt["packets"], 1)
        nRT = m.rt["RT_n_incl"] * m.rt["RT_n_az"]
        # per crossing: kappa_factor 8 B + one 64-byte xI_scatt record RMW per observer
        bytes_step = sent_all / world / args.steps * cross_pp * (8.0 + 2 * 64.0 * nRT)
        line = {
            "metric": "photon packets/sec (whole node), SED-mode MC packet loop with ray-tracing deposits",
            "value": sent_all / dt, "unit": "packets/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "ref4.1 2D disk 100x70, SED Monte Carlo of wavelengths %s, 128 streams/GPU x %d packets "
                                   "in the stop bin each, RT1 deposits for %d observers (both passes timed)"
                                   % (args.sed_lambdas, n2, nRT),
                       "packets_per_gpu_per_step": sent_all / world / args.steps, "crossings_per_packet": cross_pp},
            "roofline": {"bound": "hbm", "achieved": bytes_step / (dt / args.steps) / 1e9, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": bytes_step / (dt / args.steps) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                         "kernel": "k_mono (scout + commit)", "algorithmic_bytes_per_launch": bytes_step},
        }
        if world == 1 and not args.no_cpu_baseline:
            from oracle import Oracle
            cores = _quota_cores()
            orc = Oracle(m, 5e6)
            t = time.perf_counter()
            r = orc.run_mono(lams[0], max(2, int(2e5 / 11 / m.cfg.n_photons_loop)), seed=5, n_threads=cores)
            dtc = time.perf_counter() - t
            line["cpu_baseline"] = dict(value=r["counters"]["packets"] / dtc, unit="packets/s", cores=cores, kind="port",
                                        sample="%d packets of wavelength %d of the same SED workload, %d OpenMP threads, "
                                               "%.1f s" % (r["counters"]["packets"], lams[0], cores, dtc))
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--packets", type=float, default=1e8, help="packets per GPU per step")
    ap.add_argument("--config", default="ref41", choices=["ref41", "ref41_3d", "pascucci", "voronoi", "sed"])
    ap.add_argument("--sed-lambdas", default="5,15,25,35",
                    help="--config sed: wavelengths (1-based) whose SED Monte Carlo one step runs")
    ap.add_argument("--sites", type=int, default=100000,
                    help="--config voronoi: number of SPH-like sites of the tessellation (BASELINE config 5 stand-in)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pascucci", action="store_true", help="skip the extra Pascucci-disk timing of the default run")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--grid-blocks", type=int, default=0)
    ap.add_argument("--block-threads", type=int, default=0)
    ap.add_argument("--no-pola", action="store_true", help="tuning aid: do not track Stokes Q,U,V")
    ap.add_argument("--frozen", action="store_true",
                    help="tuning aid: time the loop with the temperature feedback frozen to a prior "
                         "(keeps the physics identical across diagnostic builds); not the benchmark")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from mcfost_amd import distributed as D
    from mcfost_amd.engine import Engine
    from mcfost_amd.host import model as M

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    if args.config == "sed":
        return bench_sed(args, world, rank, local_rank)
    cfg = {"ref41": M.ref41, "ref41_3d": M.ref41_3d, "pascucci": M.pascucci, "voronoi": M.ref41}[args.config]()
    if args.no_pola:
        cfg.lsepar_pola = False
    if args.config == "voronoi":
        cfg.name = "ref4.1 disk as %d Voronoi sites" % args.sites
        model = M.build_voronoi_model(cfg, args.sites, seed=1)
    else:
        model = M.build_model(cfg)
    n_local = int(args.packets)
    n_total = n_local * world
    eng = Engine(model, n_total, device=local_rank)
    first = rank * n_local

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.frozen:
        env_flags = os.environ.pop("MCGPU_DIAG_FLAGS", None)
        eng.set_E_prior(eng.run_thermal(min(n_local, 2_000_000), seed=5)["E_abs"] * (n_local / min(n_local, 2_000_000)))
        if env_flags is not None:
            os.environ["MCGPU_DIAG_FLAGS"] = env_flags

    def step(i):
        eng.launch_thermal(n_local, seed=1000 + i, first_packet=first, n_replicas=float(world),
                           frozen=args.frozen, grid_blocks=args.grid_blocks, block_threads=args.block_threads)
        ms = eng.sync()
        if world > 1:   # ONE all-reduce of the fused [E_abs | sed | n_sent | counters] buffer
            eng.allreduce_device(dist.all_reduce)
        return ms

    for i in range(args.warmup):
        step(-1 - i)
    barrier()
    t0 = time.perf_counter()
    kernel_ms = []
    for i in range(args.steps):
        kernel_ms.append(step(i))
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    out = eng.fetch()      # after the all-reduce: global sums of the last step
    cnt = out["counters"]

    # BASELINE.json quotes its metric on the Pascucci 2D disk and lists ref4.1 as the single-GPU configuration
    # (configs[1], the headline line above): the same loop on the Pascucci disk is timed next to it, same packet
    # count, same barriers and all-reduce, and reported under "pascucci_2d".
    extra = None
    if args.config == "ref41" and not args.no_pascucci and not args.frozen:
        pm = M.build_model(M.pascucci())
        pe = Engine(pm, n_total, device=local_rank)

        def pstep(i):
            pe.launch_thermal(n_local, seed=2000 + i, first_packet=first, n_replicas=float(world))
            ms = pe.sync()
            if world > 1:
                pe.allreduce_device(dist.all_reduce)
            return ms

        pstep(-1)
        barrier()
        tp0 = time.perf_counter()
        pk = [pstep(i) for i in range(2)]
        barrier()
        dtp = time.perf_counter() - tp0
        if world > 1:
            tt = torch.tensor([dtp], dtype=torch.float64, device="cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dtp = float(tt.item())
        pc = pe.fetch()["counters"]
        extra = {"value": n_total * 2 / dtp, "unit": "packets/s", "steps": 2, "warmup": 1, "ms_per_step": dtp / 2 * 1e3,
                 "kernel_ms": sum(pk) / 2, "workload": "Pascucci 2D disk 100x70, 61 wavelengths, isotropic scattering, "
                 "%.3g packets/GPU/step" % n_local,
                 "crossings_per_packet": pc["crossings"] / max(pc["packets"], 1),
                 "interactions_per_packet": (pc["scatterings"] + pc["absorptions"]) / max(pc["packets"], 1)}
        pe.close()
    if rank == 0:
        ms_step = dt / args.steps * 1e3
        value = n_total * args.steps / dt
        n_units = cnt["packets"] if world == 1 else cnt["packets"] / world
        cross_pp = cnt["crossings"] / max(cnt["packets"], 1)
        inter_pp = (cnt["scatterings"] + cnt["absorptions"]) / max(cnt["packets"], 1)
        per_crossing = BYTES_PER_CROSSING
        if args.config == "voronoi":
            # a crossing reads the cell record (32 B) and its inlined
            # neighbour list (16 B per neighbour), then the E_abs RMW (16 B)
            g = model.grid
            per_crossing = 32.0 + 16.0 * (g["v_neigh"].size / g["n_cells"]) + 16.0
        bytes_launch = n_local * (cross_pp * per_crossing + inter_pp * BYTES_PER_INTERACTION)
        k_ms = sum(kernel_ms) / len(kernel_ms)
        achieved = bytes_launch / (k_ms * 1e-3) / 1e9
        line = {
            "metric": "photon packets/sec (whole node), thermal MC packet loop, %s" % cfg.name, "value": value,
            "unit": "packets/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s 2D cylindrical disk %dx%dx%d, %d wavelengths, %.3g packets/GPU/step, "
                                   "temperature step (live Bjorkman&Wood re-emission), synthetic dust tables, "
                                   "blackbody star" % (cfg.name, cfg.n_rad, cfg.nz, cfg.n_az, cfg.n_lambda, n_local)
                       if not (cfg.l3D or args.config == "voronoi") else
                       ("%s, %d cells, %.3g packets/GPU/step" % (cfg.name, model.n_cells, n_local)
                        if args.config == "voronoi" else
                        "%s 3D cylindrical disk %dx%dx%d, %.3g packets/GPU/step"
                        % (cfg.name, cfg.n_rad, cfg.nz, cfg.n_az, n_local)),
                       "packets_per_gpu": n_local, "parallelism": "packets sharded x%d, tables replicated" % world,
                       "crossings_per_packet": cross_pp, "interactions_per_packet": inter_pp},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic(args.config, n_local, world),
                         "kernel": "k_thermal_voro" if args.config == "voronoi" else "k_thermal_roles", "kernel_ms": k_ms,
                         "algorithmic_bytes_per_launch": bytes_launch},
        }
        if extra is not None:
            line["pascucci_2d"] = extra
        if world == 1 and not args.no_cpu_baseline:
            base, T_cpu, n_cpu = cpu_baseline(model, n_total, args.cpu_seconds)
            line["cpu_baseline"] = base
            if not args.frozen:  # Tdust of the last timed step against the CPU port's
                line["tdust_vs_cpu"] = tdust_parity(eng.temp_finale(out["E_abs"]), n_total, T_cpu, n_cpu, cfg.T_min)
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
