#!/usr/bin/env python3
"""bench.py -- photon packets/s of the thermal Monte Carlo packet loop on the
BASELINE workloads: the Pascucci 2D disk BASELINE.json's metric is quoted on (the
headline line) and ref4.1.para 2D (configs[1], carried as a second full block),
1e8 packets per GPU and step each.

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


def source_hash():
    """Hash of the kernel sources: a committed PMC summary only counts for the code it was measured on."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "mcfost_amd", "csrc", "*.h")) +
                    glob.glob(os.path.join(ROOT, "mcfost_amd", "csrc", "*.hip"))):
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def pmc_summary(config, n_local, world):
    """Counter evidence for the dominant kernel from the committed PMC passes of THIS command and THIS source
    (`tools/collect_profiles.sh`: rocprofv3 --pmc in separate passes of `python bench.py --config <config> --steps 1
    --warmup 0`, summarised in profiles/r02_pmc_<config>.json with the hash of the kernel sources).  HBM traffic per
    launch = 2 x FETCH_SIZE (gfx950 correction, MI355X guide, HBM section) + WRITE_SIZE, counters in KB.  Returns {} when
    there is no summary for this command or the sources changed since (so nothing stale is ever reported)."""
    if world != 1:
        return {}
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "r02_pmc_%s.json" % config)))
        if d.get("source_hash") != source_hash() or int(d.get("packets", 0)) != int(n_local):
            return {}
        return d.get("per_launch", {})
    except Exception:
        return {}


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

    cfg = M.ref41()
    if args.sed_observers:   # ref4.1.para asks for RT n_incl = 3; BASELINE config 2 quotes 10 inclinations
        cfg.RT_n_incl = args.sed_observers
    m = M.build_model(cfg)
    e = Engine(m, 5e6, device=local_rank)
    T = e.temp_finale(e.run_thermal(5_000_000, seed=3)["E_abs"])
    M.repartition_energie(m, T)
    e.close()
    eng = Engine(m, 5e6, device=local_rank)
    if args.xI_precision == 4:
        eng.set_xI_precision(4)
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
                       "packets_per_gpu_per_step": sent_all / world / args.steps, "crossings_per_packet": cross_pp,
                       "observers": nRT, "xI_record": "f32 pairs" if args.xI_precision == 4 else "f64",
                       "records_per_s": sent_all / dt * cross_pp * nRT},
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


def thermal_block(M, D, Engine, dist, torch, args, config, world, rank, local_rank, steps, warmup, with_cpu):
    """Times `steps` passes of the thermal packet loop on one configuration; returns (block dict or None on ranks > 0)."""
    cfg = {"ref41": M.ref41, "ref41_mrw": M.ref41, "ref41_var": M.ref41, "ref41_3d": M.ref41_3d, "pascucci": M.pascucci, "voronoi": M.ref41}[config]()
    if args.no_pola:
        cfg.lsepar_pola = False
    if args.dust_mass:   # a heavier (optically thicker) disk than the configuration's: where the random walk matters
        cfg.dust_mass = args.dust_mass
        cfg.name += " with M_dust = %g Msun" % args.dust_mass
    if config == "voronoi":
        cfg.name = "ref4.1 disk as %d Voronoi sites" % args.sites
        model = M.build_voronoi_model(cfg, args.sites, seed=1, cache_dir=os.path.join(ROOT, "tools", "cache"))
    else:
        model = M.build_model(cfg)
    if config == "ref41_var":   # every layer of the disk its own dust (lvariable_dust), HBM-gather kernel
        cfg.name += " with variable dust (%d classes%s)" % (cfg.nz, ", all equal to the model's dust" if args.var_identical else "")
        M.init_variable_dust(model, identical=args.var_identical)
    if config == "ref41_mrw":   # BASELINE config 4: ref4.1 with the modified random walk (gamma_MRW = 2, MRW.f90:11)
        cfg.name += " + MRW (gamma %g)" % args.mrw_gamma
        M.init_mrw(model, gamma=args.mrw_gamma)
    n_local = int(args.packets)
    n_total = n_local * world
    eng = Engine(model, n_total, device=local_rank)
    first = rank * n_local

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.frozen:
        eng.set_E_prior(eng.run_thermal(min(n_local, 2_000_000), seed=5)["E_abs"] * (n_local / min(n_local, 2_000_000)))

    def step(i):
        eng.launch_thermal(n_local, seed=1000 + i, first_packet=first, n_replicas=float(world),
                           frozen=args.frozen, grid_blocks=args.grid_blocks, block_threads=args.block_threads)
        ms = eng.sync()   # HIP events on the engine's own stream around the launch
        if world > 1:     # ONE all-reduce of the fused [E_abs | sed | n_sent | counters] buffer
            eng.allreduce_device(dist.all_reduce)
        return ms

    for i in range(warmup):
        step(-1 - i)
    barrier()
    t0 = time.perf_counter()
    kernel_ms = []
    for i in range(steps):
        kernel_ms.append(step(i))
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    out = eng.fetch()      # after the all-reduce: global sums of the last step
    cnt = out["counters"]
    block = None
    if rank == 0:
        cross_pp = cnt["crossings"] / max(cnt["packets"], 1)
        inter_pp = (cnt["scatterings"] + cnt["absorptions"]) / max(cnt["packets"], 1)
        per_crossing = BYTES_PER_CROSSING
        if config == "voronoi":
            # a crossing reads the cell record (32 B) and its inlined
            # neighbour list (16 B per neighbour), then the E_abs RMW (16 B)
            g = model.grid
            per_crossing = 32.0 + 16.0 * (g["v_neigh"].size / g["n_cells"]) + 16.0
        bytes_launch = n_local * (cross_pp * per_crossing + inter_pp * BYTES_PER_INTERACTION)
        k_ms = sum(kernel_ms) / len(kernel_ms)
        achieved = bytes_launch / (k_ms * 1e-3) / 1e9
        pmc = pmc_summary(config, n_local, world)
        roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": pmc.get("hbm_bytes"),
                "kernel": {"voronoi": "k_thermal_voro_cache", "ref41_var": "k_thermal_var"}.get(config, "k_thermal_roles"), "kernel_ms": k_ms,
                "algorithmic_bytes_per_launch": bytes_launch,
                # what `achieved` is: SURVEY 8(d)'s per-crossing byte model x the kernel's own event counts / kernel time.
                # The 2D working set (absorbed-energy grid, tables, packet records) lives in LDS and L2, so the model
                # bytes are NOT HBM traffic: `traffic` (PMC) is, and the binding resource is instruction issue
                # (valu_busy, waves_per_simd)
                "note": "achieved = algorithmic bytes (SURVEY 8d model) / kernel time; HBM utilisation = traffic / kernel_ms",
                "hbm_frac_measured": (pmc["hbm_bytes"] / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if pmc.get("hbm_bytes") else None,
                "valu_busy": pmc.get("valu_busy"), "wait_frac": pmc.get("wait_frac"),
                "waves_per_simd": pmc.get("waves_per_simd"), "fp64_tflops": pmc.get("fp64_tflops"),
                "lane_utilisation": pmc.get("lane_utilisation")}
        block = {"value": n_total * steps / dt, "unit": "packets/s", "steps": steps, "warmup": warmup,
                 "ms_per_step": dt / steps * 1e3,
                 "config": {"workload": ("%s 2D cylindrical disk %dx%dx%d, %d wavelengths, %.3g packets/GPU/step, temperature "
                                         "step (live Bjorkman&Wood re-emission), synthetic dust tables, blackbody star"
                                         % (cfg.name, cfg.n_rad, cfg.nz, cfg.n_az, cfg.n_lambda, n_local))
                            if not (cfg.l3D or config == "voronoi") else
                            ("%s, %d cells, %.3g packets/GPU/step" % (cfg.name, model.n_cells, n_local) if config == "voronoi"
                             else "%s 3D cylindrical disk %dx%dx%d, %.3g packets/GPU/step"
                             % (cfg.name, cfg.n_rad, cfg.nz, cfg.n_az, n_local)),
                            "packets_per_gpu": n_local, "parallelism": "packets sharded x%d, tables replicated" % world,
                            "crossings_per_packet": cross_pp, "interactions_per_packet": inter_pp,
                            **({"mrw_walks_per_packet": cnt["mrw_walks"] / max(cnt["packets"], 1),
                                "mrw_steps_per_packet": cnt["mrw_steps"] / max(cnt["packets"], 1)}
                               if config == "ref41_mrw" else {})},
                 "roofline": roof}
        if cfg.l3D and config != "voronoi":   # binned deposits (mc_binned.hip.h): how the step was chunked, what overflowed
            block["binned_deposits"] = {k: eng.get_info("bin_" + k) for k in
                                        ("buckets", "log_blocks", "chunks", "deposits_per_packet", "overflow_blocks", "drained_records")}
        if with_cpu:
            base, T_cpu, n_cpu = cpu_baseline(model, n_total, args.cpu_seconds)
            block["cpu_baseline"] = base
            if not args.frozen:  # Tdust of the last timed step against the CPU port's
                block["tdust_vs_cpu"] = tdust_parity(eng.temp_finale(out["E_abs"]), n_total, T_cpu, n_cpu, cfg.T_min)
    eng.close()
    return block, cfg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--packets", type=float, default=1e8, help="packets per GPU per step")
    ap.add_argument("--config", default="pascucci", choices=["pascucci", "ref41", "ref41_mrw", "ref41_var", "ref41_3d", "voronoi", "sed"],
                    help="pascucci (default): the disk BASELINE.json's metric is quoted on, with ref4.1 (configs[1]) as a "
                         "second full block of the same line")
    ap.add_argument("--sed-lambdas", default="5,15,25,35",
                    help="--config sed: wavelengths (1-based) whose SED Monte Carlo one step runs")
    ap.add_argument("--sites", type=int, default=100000,
                    help="--config voronoi: number of SPH-like sites of the tessellation (BASELINE config 5 stand-in)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sed-observers", type=int, default=0, help="--config sed: RT_n_incl (default: the configuration's)")
    ap.add_argument("--xI-precision", type=int, default=8, choices=[4, 8], help="--config sed: mcgpu_set_xI_precision")
    ap.add_argument("--var-identical", action="store_true", help="--config ref41_var: every class gets the model's own tables "
                    "(the physics of --config ref41 through the HBM-gather kernel)")
    ap.add_argument("--mrw-gamma", type=float, default=2.0, help="--config ref41_mrw: gamma_MRW")
    ap.add_argument("--dust-mass", type=float, default=0.0, help="override the disk's dust mass [Msun] (thermal configs)")
    ap.add_argument("--no-ref41", action="store_true", help="--config pascucci: skip the ref4.1 block")
    ap.add_argument("--no-pascucci", action="store_true", help="(kept for older command lines: same as --no-ref41)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
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
    with_cpu = world == 1 and not args.no_cpu_baseline
    block, cfg = thermal_block(M, D, Engine, dist, torch, args, args.config, world, rank, local_rank, args.steps,
                               args.warmup, with_cpu)
    extra = None
    if args.config == "pascucci" and not (args.no_ref41 or args.no_pascucci) and not args.frozen:
        # BASELINE.json lists ref4.1 as the single-GPU configuration (configs[1]): the same loop, same packet count,
        # same steps, barriers and all-reduce, with its own roofline, CPU baseline and temperature parity
        extra, _ = thermal_block(M, D, Engine, dist, torch, args, "ref41", world, rank, local_rank, args.steps,
                                 args.warmup, with_cpu)
    if rank == 0:
        line = {"metric": "photon packets/sec (whole node), thermal MC packet loop, %s" % cfg.name,
                "value": block["value"], "unit": "packets/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": block["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": "f64", "data": "synthetic", "config": block["config"], "roofline": block["roofline"]}
        for k in ("cpu_baseline", "tdust_vs_cpu"):
            if k in block:
                line[k] = block[k]
        if extra is not None:
            line["ref41_2d"] = extra
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
