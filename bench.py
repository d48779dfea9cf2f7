#!/usr/bin/env python3
"""bench.py -- photon packets/s of the Monte Carlo packet loop on the BASELINE workloads.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

The headline line is the Pascucci 2D disk BASELINE.json's metric is quoted on, 1e8 packets per GPU and step.
One "step" = one pass of mc_photon_loop (dust_transfer.f90:439-572) for the temperature step with all tables resident
in HBM, followed (N > 1) by the single RCCL all-reduce of the fused accumulator [E_abs | sed | n_sent | counters].
Weak scaling: per-GPU work is fixed.  Rank 0 prints ONE JSON line.

N > 1 runs either as one process per GPU under torch.distributed.run (backend nccl = RCCL), or -- started plainly --
as ONE process that drives all N devices through the library's own multi-device entry (mcgpu_multi_run_thermal:
RCCL inside the library, the entry a single-process Fortran host binds); the line says which ("launcher").

On one GPU the same line carries a block per BASELINE configuration that fits one GPU -- ref4.1 2D (configs[1]),
ref4.1_3D, the Voronoi stand-in, ref4.1 with 10x the dust + MRW, the SED-mode Monte Carlo with 10 observers -- each
with its own roofline, CPU baseline and temperature parity (`--no-extra` leaves them out, `--config X` runs one alone).
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
VALU_ISSUE_PEAK = 1024 * 2.4e9 / 4.0   # wave-instructions/s: 256 CUs x 4 SIMDs, one VALU instruction per 4 cycles at 2.4 GHz
ATOMIC_LINE_PEAK = 2.37e10   # memory-side atomic operations/s, any type or footprint (profiles/r02_atomic_scope_bench.log)
PROFILE_ROUND = "r06"


def np_sum(a):
    import numpy as np
    return np.asarray(a).sum()


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
                    glob.glob(os.path.join(ROOT, "mcfost_amd", "csrc", "*.hip")) +
                    glob.glob(os.path.join(ROOT, "mcfost_amd", "csrc", "*.cpp"))):
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def pmc_summary(config, n_local, world):
    """Counter evidence for the dominant kernel from the committed PMC passes of THIS command and THIS source
    (`tools/collect_profiles.sh`: rocprofv3 --pmc in separate passes of `python bench.py --config <config> --steps 1
    --warmup 0`, summarised in profiles/<round>_pmc_<config>.json with the hash of the kernel sources).  HBM traffic per
    launch = 2 x FETCH_SIZE (gfx950 correction, MI355X guide, HBM section) + WRITE_SIZE, counters in KB.  Returns {} when
    there is no summary for this command or the sources changed since (so nothing stale is ever reported)."""
    if world != 1:
        return {}
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "%s_pmc_%s.json" % (PROFILE_ROUND, config))))
        if d.get("source_hash") != source_hash() or int(d.get("packets", 0)) != int(n_local):
            return {}
        out = dict(d.get("per_launch", {}))
        # (these fields are NOT measured by this run: they are copied from the committed counter passes of the same command on
        # the same kernel sources, and the line says so)
        out["counters_from"] = "profiles/%s_pmc_%s.json @ %s" % (PROFILE_ROUND, config, d.get("source_hash"))
        return out
    except Exception:
        return {}


def cpu_baseline(model, n_total, target_s=15.0):
    """The CPU restatement (oracle, kind "port") timed on this box's host cores
    on a bounded sample of the same workload."""
    from oracle import Oracle
    cores = _quota_cores()
    orc = Oracle(model, n_total)
    t = time.perf_counter()
    orc.run_thermal(5000 * cores, seed=99, n_threads=cores)
    rate = 5000 * cores / (time.perf_counter() - t)
    n = int(max(5000 * cores, min(rate * target_s, 5e7)))
    orc = Oracle(model, n)   # a self-consistent run of n packets: its own packet luminosity, its own temperature
    t = time.perf_counter()
    res = orc.run_thermal(n, seed=100, n_threads=cores)
    dt = time.perf_counter() - t
    base = dict(value=n / dt, unit="packets/s", cores=cores, kind="port",
                sample="%d packets of the same thermal workload, %d OpenMP threads, %.1f s; "
                       "%.1f crossings/packet" % (n, cores, dt, res["counters"]["crossings"] / n))
    return base, orc.temp_finale(res["E_abs"]), n


def tdust_parity(T_gpu, n_gpu, T_cpu, n_cpu, T_min, noise_pair=None, bins=None, xN_cpu=None):
    """Temperature of the GPU step against the CPU port's (independent noise): relative RMS over the cells with
    T > 1.01 T_min, the reference's own gate p75(|dT|/T) (test_suite/test_mcfost.py:46-57,88: < 5 %), and the
    tolerance 3 sigma_MC.  2D grids: sigma_MC(N) = 1.7 % sqrt(1.28e5 / N) per run (measured on
    the 7000-cell ref4.1 grid).  Other grids (noise_pair = the temperatures of two runs of n_cpu packets with
    different seeds): the seed-to-seed scatter measured on this very grid, sigma(n) = rms(T_a / T_b - 1) / sqrt(2)."""
    import numpy as np
    sel = (T_cpu > 1.01 * T_min) & (T_gpu > 1.01 * T_min)
    rel = (T_gpu[sel] - T_cpu[sel]) / T_cpu[sel]
    if noise_pair is not None:
        Ta, Tb = noise_pair
        s2 = sel & (Ta > 1.01 * T_min) & (Tb > 1.01 * T_min)
        sig_n = float(np.sqrt(np.mean((Ta[s2] / Tb[s2] - 1.0) ** 2))) / np.sqrt(2.0)   # one run of n_cpu packets
        sigma = sig_n * float(np.sqrt(1.0 + n_cpu / n_gpu))
        model = "seed-to-seed scatter of two runs of %d packets on this grid" % n_cpu
    else:
        sigma = 0.017 * float(np.sqrt(1.28e5 / n_gpu + 1.28e5 / n_cpu)) * float(np.sqrt(max(T_cpu.size, 7000) / 7000.0))
        model = "1.7 % sqrt(1.28e5 / N) per run (BASELINE.md section 2)"
    rms, p75 = float(np.sqrt(np.mean(rel ** 2))), float(np.percentile(np.abs(rel), 75))
    out = dict(rel_rms=rms, p75=p75, tolerance_rel_rms=3.0 * sigma, noise_model=model, cells=int(sel.sum()),
               ok=bool(rms <= 3.0 * sigma), reference_gate_p75_below_5pct=bool(p75 < 0.05))
    if xN_cpu is not None:
        # grids of 1e5 ... 1e6 cells: the CPU sample puts a handful of packets into most cells, and the reference's gate
        # (which its suite applies to converged maps) then measures that sample's noise: the all-cells flag above is kept
        # for the record, the cell-level statement is the same percentile over the cells a sample of n_cpu packets
        # RESOLVES -- by expected statistics, not by a noise realisation: xN_abs (radiation_field.f90:55, the packets that
        # crossed the cell) of an independent GPU run of n_cpu packets >= 200, i.e. sigma_T / T of about 2 % in the CPU map
        res = sel & (xN_cpu >= 200.0)
        relq = (T_gpu[res] - T_cpu[res]) / T_cpu[res]
        out["resolved_cells"] = int(res.sum())
        out["resolved_cells_rule"] = "xN_abs >= 200 in a run of the CPU sample's packet count"
        if res.sum() > 100:
            out["p75_resolved_cells"] = float(np.percentile(np.abs(relq), 75))
            out["reference_gate_p75_below_5pct_resolved_cells"] = bool(out["p75_resolved_cells"] < 0.05)
            out["all_cells_gate_note"] = ("reference_gate_p75_below_5pct is over every cell, most of which the CPU sample of %d "
                                          "packets does not resolve; the cell-level gate is the resolved-cells one" % n_cpu)
    if bins is not None:
        # ... and the gate on the temperature MAP: the cells averaged over what the axisymmetric disk cannot tell apart
        # (3D: the azimuths of a ring; Voronoi: cells of one (log r, |z| / r) bin), where the CPU sample's per-cell noise
        # averages out and a difference of the two codes would not
        nb = int(bins.max()) + 1
        cnt = np.bincount(bins[sel], minlength=nb)
        mg = np.bincount(bins[sel], weights=T_gpu[sel], minlength=nb)
        mc = np.bincount(bins[sel], weights=T_cpu[sel], minlength=nb)
        okb = cnt >= 8
        relb = np.abs(mg[okb] / mc[okb] - 1.0)
        if relb.size:
            out["map_bins"] = int(okb.sum())
            out["p75_map"] = float(np.percentile(relb, 75))
            out["reference_gate_p75_below_5pct_map"] = bool(out["p75_map"] < 0.05)
    return out


class Par:
    """How the N GPUs are driven: "single" (N = 1), "torchrun" (one process per GPU, torch.distributed nccl = RCCL) or
    "library" (ONE process, mcgpu_multi_*: RCCL inside the library)."""

    def __init__(self, gpus, shared_device=False, force_dist=False):
        self.shared_device = bool(shared_device)   # library mode on ONE GPU: every context on device 0 (a dry run of the
        self.world = int(os.environ.get("WORLD_SIZE", "1"))   # n_dev > 1 code; mcgpu_multi_create_ex, include/mcgpu.h)
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.dist = self.torch = None
        self.rccl_ranks = None
        self.n_allreduce = 0
        if force_dist and "WORLD_SIZE" not in os.environ:
            raise SystemExit("--force-dist needs the launcher's environment (RANK / WORLD_SIZE / MASTER_*): start it as "
                             "`python -m torch.distributed.run --nproc-per-node 1 --master-addr 127.0.0.1 bench.py --gpus 1 --force-dist`")
        if self.world > 1 or force_dist:   # (--force-dist: the process group and its all-reduce also over a world of ONE)
            self.mode = "torchrun"
            import torch
            import torch.distributed as dist
            self.torch, self.dist = torch, dist
            torch.cuda.set_device(self.local_rank)
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("nccl", device_id=torch.device("cuda", self.local_rank))
            self.n_gpus = self.world
        elif gpus > 1:
            self.mode = "library"
            self.n_gpus = gpus
            if not self.shared_device:
                import torch
                n_have = torch.cuda.device_count()   # (counts without initialising the GPU)
                if n_have < gpus:
                    raise SystemExit("bench.py --gpus %d: this node shows %d GPU(s). Nothing was measured. (One process per GPU: "
                                     "`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`; a dry run of the "
                                     "n_dev > 1 code on one GPU: --shared-device.)" % (gpus, n_have))
        else:
            self.mode = "single"
            self.n_gpus = 1

    def barrier(self):
        if self.mode == "torchrun":
            self.dist.barrier()
            self.torch.cuda.synchronize()

    def max_over_ranks(self, x):
        if self.mode != "torchrun":
            return x
        t = self.torch.tensor([x], dtype=self.torch.float64, device="cuda")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, x):
        if self.mode != "torchrun":
            return x
        t = self.torch.tensor([x], dtype=self.torch.float64, device="cuda")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    def finish(self):
        if self.mode == "torchrun":
            self.dist.destroy_process_group()


def build_workload(M, args, config):
    cfg = {"ref41": M.ref41, "ref41_mrw": M.ref41, "ref41_var": M.ref41, "ref41_3d": M.ref41_3d, "pascucci": M.pascucci,
           "voronoi": M.ref41}[config]()
    if args.no_pola:
        cfg.lsepar_pola = False
    dust_mass = args.dust_mass
    if config == "ref41_mrw" and not dust_mass:
        dust_mass = 10.0 * cfg.dust_mass   # the stock disk never walks (its cells are thin where packets are re-emitted)
    if dust_mass:   # a heavier (optically thicker) disk than the configuration's: where the random walk matters
        cfg.dust_mass = dust_mass
        cfg.name += " with M_dust = %g Msun" % dust_mass
    if config == "voronoi":
        cfg.name = "ref4.1 disk as %d Voronoi sites" % args.sites
        # the tessellation is built here, on the box, by the product's kernel (mcgpu_voronoi_tesselation: one thread per
        # cell; the candidates of a cell are its Delaunay neighbours from one qhull run on the host), with the
        # reference's cuts of elongated cells and at the stellar surface (voro++_wrapper.cpp:209-262)
        from mcfost_amd.host import voronoi as V
        t_tess = time.perf_counter()
        kern = V.device_tessellator(int(os.environ.get("LOCAL_RANK", "0")))
        model = M.build_voronoi_model(cfg, args.sites, seed=1, tessellator=kern, platonic=True, density=args.voronoi_density,
                                      order=args.site_order)
        model.extra["tessellation_s"] = time.perf_counter() - t_tess
        model.extra["tessellation_kernel_ms"] = kern.kernel_ms
    else:
        model = M.build_model(cfg)
    model.midplane_snap = 0   # 3D grids: the library's default, the reference's literal arithmetic (include/mcgpu.h)
    if config == "ref41_var":   # every layer of the disk its own dust (lvariable_dust), HBM-gather kernel
        cfg.name += " with variable dust (%d classes%s)" % (cfg.nz, ", all equal to the model's dust" if args.var_identical else "")
        M.init_variable_dust(model, identical=args.var_identical)
    if config == "ref41_mrw":   # BASELINE config 4: ref4.1 with the modified random walk (gamma_MRW = 2, MRW.f90:11)
        cfg.name += " + MRW (gamma %g)" % args.mrw_gamma
        M.init_mrw(model, gamma=args.mrw_gamma, n_inter=args.mrw_n_inter)
    return cfg, model


def thermal_block(par, args, config, steps, warmup, with_cpu, n_local, crossing=None):
    """Times `steps` passes of the thermal packet loop on one configuration; returns (block or None on ranks > 0, cfg)."""
    from mcfost_amd.engine import Engine, MultiEngine
    from mcfost_amd.host import model as M
    cfg, model = build_workload(M, args, config)
    world = par.n_gpus
    n_local = int(n_local)
    n_total = n_local * world
    if par.mode == "library":
        me = MultiEngine(model, n_total, devices=((0,) * world if par.shared_device else tuple(range(world))),
                         shared_device=par.shared_device)
        eng = me.engines[0]
    else:
        me = None
        eng = Engine(model, n_total, device=par.local_rank)
    first = par.rank * n_local
    if args.tail >= 0:
        for x in (me.engines if me is not None else [eng]):
            x.set_option("tail", args.tail)
    for name, value in (("schedule", args.schedule), ("crossing", args.crossing if crossing is None else crossing), ("voronoi_pool_log_records", args.pool_log_records),
                        ("tail_where", args.tail_where), ("host_threads", args.host_threads), ("tail_host_packets", args.tail_host_packets),
                        ("voronoi_cache_log_slots", args.cache_log_slots), ("deposit_log_mb", args.deposit_log_mb)):
        if value >= 0:
            for x in (me.engines if me is not None else [eng]):
                x.set_option(name, value)
    if args.frozen:
        eng.set_E_prior(eng.run_thermal(min(n_local, 2_000_000), seed=5)["E_abs"] * (n_local / min(n_local, 2_000_000)))
    last = {}

    def step(i):
        if me is not None:   # ONE call: shards, launches on every device, the all-reduce, the fetch from device 0
            last["out"] = me.run_thermal(n_total, seed=1000 + i, frozen=args.frozen, grid_blocks=args.grid_blocks,
                                         block_threads=args.block_threads)
            return last["out"]["kernel_ms"]
        eng.launch_thermal(n_local, seed=1000 + i, first_packet=first, n_replicas=float(world),
                           frozen=args.frozen, grid_blocks=args.grid_blocks, block_threads=args.block_threads)
        ms = eng.sync()   # HIP events on the engine's own stream around the launch
        if par.mode == "torchrun":     # ONE all-reduce of the fused [E_abs | sed | n_sent | counters] buffer
            eng.allreduce_device(par.dist.all_reduce)
            par.n_allreduce += 1
        return ms

    for i in range(warmup):
        step(-1 - i)
    par.barrier()
    t0 = time.perf_counter()
    kernel_ms = []
    for i in range(steps):
        kernel_ms.append(step(i))
    par.barrier()
    dt = par.max_over_ranks(time.perf_counter() - t0)
    out = last["out"] if me is not None else eng.fetch()      # after the all-reduce: global sums of the last step
    if me is not None:
        par.rccl_ranks = me.rccl_ranks()
    cnt = out["counters"]
    block = None
    if par.rank == 0:
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
        if config == "voronoi" and args.sites != 1000000:   # (the committed counter passes are those of the default tessellation)
            pmc = {}
        if (args.crossing if crossing is None else crossing) == 1:   # (... and those of the default crossing, not of option "crossing" = 1)
            pmc = {}
        rate_gpu = n_local / (k_ms * 1e-3)
        valu_pp = (pmc.get("insts_per_packet") or {}).get("valu")
        binned = cfg.l3D and config != "voronoi" and eng.get_info("bin_buckets") > 0
        # What binds the kernel: the 2D working set lives in LDS and L2, so the kernels issue
        # vector-ALU instructions, not HBM requests; `frac` stays SURVEY 8(d)'s algorithmic-bytes figure.
        roof = {"bound": "valu-issue", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": pmc.get("hbm_bytes"),
                "kernel": {"voronoi": "k_thermal_voro_cache", "ref41_var": "k_thermal_roles_var"}.get(
                    config, "k_thermal_roles_bin (+ k_fold_bins)" if binned else "k_thermal_roles"),
                "kernel_ms": k_ms, "algorithmic_bytes_per_launch": bytes_launch,
                "note": "achieved / frac = algorithmic bytes (SURVEY 8d model) / kernel time against the HBM peak; the bound "
                        "names what the counters show binding the kernel; hbm_frac_measured = PMC traffic / kernel time",
                "valu_issue_frac": (valu_pp * rate_gpu / VALU_ISSUE_PEAK) if valu_pp else None,
                "hbm_frac_measured": (pmc["hbm_bytes"] / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if pmc.get("hbm_bytes") else None,
                "valu_busy": min(1.0, pmc["valu_busy"]) if pmc.get("valu_busy") else None, "wait_frac": pmc.get("wait_frac"),
                "waves_per_simd": pmc.get("waves_per_simd"), "fp64_tflops": pmc.get("fp64_tflops"),
                "lane_utilisation": pmc.get("lane_utilisation"),
                # where traffic, valu_*, wait_frac, waves_per_simd, fp64_tflops, lane_utilisation come from (null: no committed
                # counter passes for these kernel sources -- the fields above are then null too)
                "counters_from": pmc.get("counters_from")}
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
        # the end of a launch is the serial latency of its longest packets ("k_tail"): the tail kernel's
        # share of the last step and the longest packet's events (crossings + interactions), so that
        # tail_ms ~ events x us_per_event can be read off the record
        tail_ms, ev_max = eng.get_info("tail_ms"), eng.get_info("longest_packet_events")
        longest = {k: eng.get_info("longest_packet_" + k) for k in ("crossings", "scatterings", "absorptions", "walks", "steps")} if ev_max > 10000 else None
        block["tail"] = {"tail_ms": tail_ms, "longest_packet_events": ev_max,
                         # the longest packet's own counts: crossings over its whole life, the rest as far as k_tail ran it
                         "longest_packet": longest,
                         "us_per_event_if_one_packet": (tail_ms * 1e3 / ev_max) if (tail_ms > 0 and ev_max > 0) else None,
                         "tail_threshold": eng.get_info("tail_threshold"),
                         # where the launch's last packets ended (mc_tail.hip.h "The last packets on the host"): k_tail thins
                         # the tail out, the library's host threads finish the last ones
                         "where": {0: None, 1: "device (k_tail)", 2: "device (k_tail), the last packets on host threads"}[int(eng.get_info("tail_where"))],
                         "host_ms": eng.get_info("tail_host_ms"), "host_packets": eng.get_info("tail_host_packets"),
                         "host_threads": eng.get_info("tail_host_threads"), "host_events": eng.get_info("tail_host_events")}
        if block["tail"]["host_events"]:
            block["tail"]["host_ns_per_event_per_thread"] = (block["tail"]["host_ms"] * 1e6 * block["tail"]["host_threads"] /
                                                            block["tail"]["host_events"])
        if config == "voronoi" and me is None:
            # The crossing engine alone: the same grid, the same kernel, the same packets, but a transparent disk
            # (kappa_factor = 0: no interaction ever) -- crossings per second without the 119 interactions per packet of the
            # stand-in disk that share the step with them; and the step's split that rate implies.
            import copy
            import numpy as np
            mt = copy.copy(model)
            mt.kappa_factor = np.zeros_like(np.asarray(model.kappa_factor))
            et = Engine(mt, n_total, device=par.local_rank)
            n_t = int(min(n_local, 2e7))
            et.run_thermal(n_t, seed=77)
            rt_ = et.run_thermal(n_t, seed=78)
            et.close()
            c_t = rt_["counters"]["crossings"]
            rate_t = c_t / (rt_["kernel_ms"] * 1e-3)
            block["crossing_engine"] = {
                "transparent_disk": {"packets": n_t, "crossings_per_packet": c_t / n_t, "kernel_ms": rt_["kernel_ms"],
                                     "crossings_per_s": rate_t, "packets_per_s": n_t / (rt_["kernel_ms"] * 1e-3)},
                "step": {"crossings_per_s": cross_pp * n_local / (k_ms * 1e-3), "interactions_per_s": inter_pp * n_local / (k_ms * 1e-3)},
                "note": "crossings alone, straight through the whole grid (every cell's neighbour list cold), against the step's "
                        "rate with its interactions included: the step's crossings are short hops between the few thousand cells "
                        "of the inner disk, whose records stay in L2 -- the interactions (119 per packet on this stand-in, 10.8 on "
                        "the cylindrical grid of the same disk) do not hide a faster crossing engine"}
        if config == "voronoi":
            block["tessellation"] = {"sites": args.sites, "host_s": model.extra.get("tessellation_s"),
                                     "kernel_ms": model.extra.get("tessellation_kernel_ms"),
                                     "faces_per_cell": float(model.grid["v_neigh"].size / model.grid["n_cells"]),
                                     "cut_cells": int(np_sum(model.grid["v_was_cut"])),
                                     "builder": "mcgpu_voronoi_tesselation (Delaunay candidates from qhull, clipping on the device)"}
        if binned:   # binned deposits (mc_binned.hip.h): how the step was chunked, what overflowed
            block["binned_deposits"] = {k: eng.get_info("bin_" + k) for k in
                                        ("buckets", "log_blocks", "chunks", "deposits_per_packet", "overflow_blocks", "drained_records")}
        if with_cpu:
            base, T_cpu, n_cpu = cpu_baseline(model, n_total, args.cpu_seconds)
            block["cpu_baseline"] = base
            if not args.frozen:  # Tdust of the last timed step against the CPU port's
                pair = None
                # this model's own noise at the CPU sample's packet count: the stock law was measured on the 7000-cell
                # ref4.1 disk; 3D and Voronoi grids have other cell counts, and the thick disk of config 4 is noisier in its
                # midplane (a few trapped packets carry its deep cells' energy: round 3's line failed the stock law there,
                # 0.00751 against 0.00747)
                xN = None
                if (cfg.l3D or config in ("voronoi", "ref41_mrw")) and me is None:
                    pair = [eng.temp_finale(eng.run_thermal(n_cpu, seed=s)["E_abs"]) for s in (7001, 7002)]
                    if cfg.l3D or config == "voronoi":   # which cells a sample of n_cpu packets resolves: xN_abs of a third run
                        eng.set_option("radiation_field", 1)
                        eng.run_thermal(n_cpu, seed=7003)
                        xN, _ = eng.fetch_radiation_field(xN=True, xJ=False)
                        eng.set_option("radiation_field", 0)
                T_gpu = eng.temp_finale(out["E_abs"])
                bins = None
                if config == "voronoi":
                    import numpy as np
                    xyz = np.asarray(model.grid["v_xyz_dp"], float).reshape(-1, 3)
                    rc = np.maximum(np.hypot(xyz[:, 0], xyz[:, 1]), 1e-6)
                    ir = np.clip((np.log(rc / cfg.rin) / np.log(cfg.rout / cfg.rin) * 60).astype(int), 0, 59)
                    iz = np.clip((np.abs(xyz[:, 2]) / rc / 0.5 * 30).astype(int), 0, 29)
                    bins = ir * 30 + iz
                elif cfg.l3D:
                    import numpy as np
                    per_az = model.n_cells // cfg.n_az       # (cells are ordered azimuth outermost: mcfost_amd/host/model.py)
                    bins = np.arange(model.n_cells) % per_az
                    bins = (bins % cfg.n_rad) + cfg.n_rad * (np.abs((bins // cfg.n_rad) - cfg.nz + 0.5).astype(int))   # (both hemispheres)
                block["tdust_vs_cpu"] = tdust_parity(T_gpu, n_total, T_cpu, n_cpu, cfg.T_min, pair, bins, xN)
                if config == "ref41_mrw" and me is None and n_local <= 2e7:   # (two more brute-force runs: the 1e7-packet block's)
                    # the walk against the brute-force loop on the GPU at the same packet count (the walk is "parity
                    # unpinned": the reference's MRW is a stub): the reference's p75 gate, the
                    # worst cells, and the columns of the illuminated inner rim that round 3 had to bound at 8 %
                    import numpy as np
                    eng.set_mrw(None)
                    rb = eng.run_thermal(n_local, seed=4321)
                    T_b = eng.temp_finale(rb["E_abs"])
                    pb = [eng.temp_finale(eng.run_thermal(n_local, seed=s)["E_abs"]) for s in (4322,)]
                    sel = (T_b > 1.2 * cfg.T_min) & (T_gpu > 1.2 * cfg.T_min)
                    rel = np.abs(T_gpu[sel] / T_b[sel] - 1.0)
                    noise = np.abs(pb[0][sel] / T_b[sel] - 1.0)          # brute force against brute force: the noise alone
                    ri = np.arange(model.n_cells) % cfg.n_rad + 1
                    zj = np.arange(model.n_cells) // cfg.n_rad + 1
                    rim = ((ri >= 17) & (ri <= 23) & (zj <= 12))[sel]
                    block["mrw_vs_brute_force_gpu"] = {
                        "packets": n_local, "brute_force_ms": rb["kernel_ms"], "cells": int(sel.sum()),
                        "p75": float(np.percentile(rel, 75)), "p99": float(np.percentile(rel, 99)), "max": float(rel.max()),
                        "rim_cells": int(rim.sum()), "rim_mean_signed": float((T_gpu[sel] / T_b[sel] - 1.0)[rim].mean()),
                        "rim_p99": float(np.percentile(rel[rim], 99)), "rim_max": float(rel[rim].max()),
                        "noise_brute_vs_brute": {"p75": float(np.percentile(noise, 75)), "p99": float(np.percentile(noise, 99)),
                                                 "max": float(noise.max()), "rim_max": float(noise[rim].max())},
                        "reference_gate_p75_below_5pct": bool(np.percentile(rel, 75) < 0.05)}
    (me or eng).close()
    return block, cfg


def sed_block(par, args, steps, warmup, with_cpu, packets, observers, all_lambdas=False, n2=None):
    """SED mode on the ref4.1 grid (the second half of BASELINE config 2): one step = the SED Monte Carlo
    (mcgpu_run_mono: scout + commit passes, ray-tracing deposits) of the listed wavelengths, every stream asked for
    packets/128/len(wavelengths) packets in the stop bin; streams sharded over the GPUs."""
    from mcfost_amd import distributed as D
    from mcfost_amd.engine import Engine, MultiEngine
    from mcfost_amd.host import model as M

    world = par.n_gpus
    cfg = M.ref41()
    if observers:   # ref4.1.para asks for RT n_incl = 3; BASELINE config 2 quotes 10 inclinations
        cfg.RT_n_incl = observers
    m = M.build_model(cfg)
    e = Engine(m, 5e6, device=par.local_rank)
    T = e.temp_finale(e.run_thermal(5_000_000, seed=3)["E_abs"])   # the dust temperature the SED step emits with
    e.close()
    if par.mode == "library":
        me = MultiEngine(m, 5e6, devices=((0,) * world if par.shared_device else tuple(range(world))),
                         shared_device=par.shared_device)
        engs = me.engines
    else:
        me = None
        engs = [Engine(m, 5e6, device=par.local_rank)]
    eng = engs[0]
    if args.xI_precision == 4:
        for x in engs:
            x.set_xI_precision(4)
    if args.xi_log >= 0:
        for x in engs:
            x.set_option("xi_log", args.xi_log)
    # (the default line runs EVERY wavelength of config 2's SED at reduced packets; --config sed the listed ones)
    lams = list(range(1, m.n_lambda + 1)) if all_lambdas else [int(x) for x in args.sed_lambdas.split(",")]
    lam_text = ("all %d wavelengths" % m.n_lambda) if all_lambdas else ("wavelengths %s" % args.sed_lambdas)
    n_streams = m.cfg.n_photons_loop * world                 # weak scaling: 128 streams per GPU
    first, count = D.shard_streams(n_streams, par.rank, par.world)
    # packets in the stop bin per stream so that a step sends about `packets` packets per GPU (1 in ~11 lands there)
    if n2 is None:
        n2 = max(10, int(packets / 11.0 / m.cfg.n_photons_loop / len(lams)))

    xlog_tot = {"launches": 0.0, "records": 0.0, "flights": 0.0}
    ev_tot = {"packets": 0.0, "crossings": 0.0}     # the step's commit passes, every wavelength (rank 0's streams)

    def step(i):
        sent = 0
        for k in xlog_tot:
            xlog_tot[k] = 0.0
        for k in ev_tot:
            ev_tot[k] = 0.0
        for lam in lams:
            # repartition_energie(lambda) on the device (dust_transfer.f90:924), then the wavelength's packet loop (:939)
            if me is not None:   # ONE call per wavelength: streams split, both all-reduces inside the library
                r = me.run_mono(lam, n2, seed=100 + i, n_chunks=n_streams, fetch_xI=False, Tdust=T)
            else:
                td = eng.repartition_energie(lam, T, fetch=False)
                r = eng.run_mono(lam, n2, seed=100 + i, n_chunks=count, first_chunk=first, fetch_xI=False, device_tables=td,
                                 block_threads=args.block_threads)
            sent += int(r["n_sent_chunk"].sum())
            for k in ev_tot:        # (a wavelength's run starts its counters at zero)
                ev_tot[k] += float(r["counters"][k])
            for k, name in (("launches", "xi_log_chunks"), ("records", "xi_log_records"), ("flights", "xi_log_flights")):
                xlog_tot[k] += eng.get_info(name)
            if par.mode == "torchrun":   # one all-reduce of [sed | n_sent] + xI_scatt per wavelength
                acc, cnt = eng.device_accumulators()
                par.dist.all_reduce(acc)
                par.dist.all_reduce(eng.device_xI())
                par.torch.cuda.current_stream().synchronize()
        return sent

    for i in range(warmup):
        step(-1 - i)
    par.barrier()
    t0 = time.perf_counter()
    sent = 0
    for i in range(steps):
        sent += step(i)
    par.barrier()
    dt = par.max_over_ranks(time.perf_counter() - t0)
    sent_all = par.sum_over_ranks(float(sent))
    block = None
    if par.rank == 0:
        if me is not None:
            par.rccl_ranks = me.rccl_ranks()
        # crossings per packet over ALL the step's wavelengths (until round 6 this read the context's counters after the
        # loop, i.e. the LAST wavelength's -- 77 crossings per packet at 3 mm where the packet-weighted mean over the 50
        # wavelengths is twice that: the block's bytes and line operations were under-reported by that factor)
        cross_pp = ev_tot["crossings"] / max(ev_tot["packets"], 1.0)
        nRT = m.rt["RT_n_incl"] * m.rt["RT_n_az"]
        # per crossing: kappa_factor 8 B + the read and the write of the values one deposit reaches per observer -- in default
        # real the packed layout's (mc_xi32.hip.h: the Stokes values, or with contributions Q, U, V + the packet's origin:
        # I is not stored there), 4 bytes each (16 B with Stokes tracking); one 64-byte record in FP64
        from mcfost_amd.engine import xi32_layout
        lay = xi32_layout(nRT, bool(cfg.lsepar_pola and cfg.aniso_method == 1), bool(cfg.lsepar_contrib))
        rec_bytes = 4.0 * lay["values_per_deposit"] if args.xI_precision == 4 else 64.0
        bytes_step = sent_all / world / steps * cross_pp * (8.0 + 2 * rec_bytes * nRT)
        # memory-side line operations per crossing: FP64 records one 64-byte line per observer; default real the lines of
        # the packed sub-bin a packet's deposits touch
        lines_per_crossing = float(lay["lines_touched"]) if args.xI_precision == 4 else float(nRT)
        line_ops_s = sent_all / world / dt * cross_pp * lines_per_crossing
        block = {
            "metric": "photon packets/sec (whole node), SED-mode MC packet loop with ray-tracing deposits",
            "value": sent_all / dt, "unit": "packets/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "ref4.1 2D disk 100x70, SED Monte Carlo of %s, 128 streams/GPU x %d packets "
                                   "in the stop bin each, RT1 deposits for %d observers (both passes timed)"
                                   % (lam_text, n2, nRT),
                       "packets_per_gpu_per_step": sent_all / world / steps, "crossings_per_packet": cross_pp,
                       "observers": nRT, "xI_record": ("f32 packed%s, %d lines per crossing" % (" (split)" if lay["split"] else "", int(lines_per_crossing))) if args.xI_precision == 4 else "f64",
                       "records_per_s": sent_all / dt * cross_pp * nRT},
            # one 64-byte record per crossing and observer is one memory-side atomic line operation: that rate, not
            # bytes, binds this mode
            "roofline": {"bound": "atomic-rate", "achieved": bytes_step / (dt / steps) / 1e9, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": bytes_step / (dt / steps) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                         "atomic_rate_frac": line_ops_s / ATOMIC_LINE_PEAK, "atomic_line_ops_per_s": line_ops_s,
                         "kernel": "k_mono (scout + commit)", "algorithmic_bytes_per_launch": bytes_step},
        }
        # the deposits of the last wavelength's commit passes: logged and folded (mc_xilog.hip.h), or global atomics
        block["xi_log"] = dict(xlog_tot, of="the last step's commit passes, rank 0 (0: atomics)")
        if with_cpu:
            from oracle import Oracle
            cores = _quota_cores()
            orc = Oracle(m, 5e6)
            t = time.perf_counter()
            r = orc.run_mono(lams[0], max(2, int(2e5 / 11 / m.cfg.n_photons_loop)), seed=5, n_threads=cores)
            dtc = time.perf_counter() - t
            block["cpu_baseline"] = dict(value=r["counters"]["packets"] / dtc, unit="packets/s", cores=cores, kind="port",
                                         sample="%d packets of wavelength %d of the same SED workload, %d OpenMP threads, "
                                                "%.1f s" % (r["counters"]["packets"], lams[0], cores, dtc))
    (me or eng).close()
    return block


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--packets", type=float, default=1e8, help="packets per GPU per step")
    ap.add_argument("--config", default="pascucci", choices=["pascucci", "ref41", "ref41_mrw", "ref41_var", "ref41_3d", "voronoi", "sed"],
                    help="pascucci (default): the disk BASELINE.json's metric is quoted on, with the other configurations as "
                         "blocks of the same line")
    ap.add_argument("--sed-lambdas", default="5,15,25,35",
                    help="--config sed: wavelengths (1-based) whose SED Monte Carlo one step runs")
    ap.add_argument("--sites", type=int, default=1000000,
                    help="--config voronoi: number of SPH-like sites of the tessellation (BASELINE config 5 stand-in)")
    ap.add_argument("--site-order", default="file", choices=["file", "morton"],
                    help="--config voronoi: the order of the cells -- file: as the sample lists its particles; morton: along a "
                         "space-filling curve (host/voronoi.py::spatial_order), neighbours in space at neighbouring addresses")
    ap.add_argument("--voronoi-density", default="smoothed", choices=["sph", "smoothed"],
                    help="--config voronoi: the cells' dust density -- smoothed (default): round 3's neighbour-averaged m / V; "
                         "sph: the analytic density at the site, m (1.2 / h)^3 (what a dump's own SPH density would be): at 1e6 "
                         "sites the few cells of the inner rim then carry its midplane density through their whole depth, "
                         "1300 crossings + 1250 interactions per packet (profiles/r04_voronoi_sph_density.json)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sed-observers", type=int, default=0, help="--config sed: RT_n_incl (default: the configuration's)")
    ap.add_argument("--xI-precision", type=int, default=4, choices=[4, 8],
                    help="--config sed: mcgpu_set_xI_precision -- 4 (default here): xI_scatt accumulated in default real, the type of "
                         "the reference's own array (dust_ray_tracing.f90:33), in the packed layout of mc_xi32.hip.h; 8: FP64 sums "
                         "(the library's default)")
    ap.add_argument("--var-identical", action="store_true", help="--config ref41_var: every class gets the model's own tables "
                    "(the physics of --config ref41 through the HBM-gather kernel)")
    ap.add_argument("--mrw-gamma", type=float, default=2.0, help="--config ref41_mrw: gamma_MRW")
    ap.add_argument("--mrw-n-inter", type=int, default=5, help="--config ref41_mrw: a walk may start after more than this many "
                    "interactions in a row in one cell (dust_transfer.f90:1223: 5; 0..6)")
    ap.add_argument("--dust-mass", type=float, default=0.0, help="override the disk's dust mass [Msun] (thermal configs; "
                    "ref41_mrw defaults to 10x the stock mass)")
    ap.add_argument("--no-ref41", action="store_true", help="--config pascucci: the headline block alone (same as --no-extra)")
    ap.add_argument("--no-extra", action="store_true", help="--config pascucci: the headline block alone")
    ap.add_argument("--no-pascucci", action="store_true", help="(kept for older command lines: same as --no-extra)")
    ap.add_argument("--cpu-seconds", type=float, default=8.0)
    ap.add_argument("--shared-device", action="store_true",
                    help="--gpus N started plainly (library mode) on a box with ONE GPU: the N contexts share device 0 and the "
                         "library's own sum stands in for the RCCL all-reduce -- a dry run of the multi-device code, not a "
                         "scaling measurement (the line says so in \"launcher\")")
    ap.add_argument("--force-dist", action="store_true",
                    help="under torch.distributed.run with ONE process: still open the nccl (RCCL) process group and all-reduce the "
                         "fused device buffer over that world of one -- how a box with one GPU executes the one-process-per-GPU path")
    ap.add_argument("--tail", type=int, default=-1, help="tuning aid: option \"tail\" (packets left per workgroup at the hand-over to "
                    "k_tail; -1 = the library's choice)")
    ap.add_argument("--xi-log", type=int, default=-1, help="--config sed: option \"xi_log\": 1 (default) = the commit pass logs its xI_scatt deposits "
                    "and a fold sums them (default-real records), 0 = global atomics")
    ap.add_argument("--tail-where", type=int, default=-1, help="option \"tail_where\": 1 = k_tail finishes every packet, 2 = the library's "
                    "host threads finish the last ones (default)")
    ap.add_argument("--host-threads", type=int, default=-1, help="option \"host_threads\": host threads of a launch's tail (0 = automatic)")
    ap.add_argument("--tail-host-packets", type=int, default=-1, help="option \"tail_host_packets\": packets k_tail leaves to the host (0 = 8 per thread)")
    ap.add_argument("--schedule", type=int, default=-1, help="tuning aid: option \"schedule\" (include/mcgpu.h)")
    ap.add_argument("--crossing", type=int, default=-1, help="option \"crossing\": 1 = the flight-parametric 2D crossing in the flying "
                    "waves (statistical parity only; include/mcgpu.h)")
    ap.add_argument("--pool-log-records", type=int, default=-1, help="tuning aid: option \"voronoi_pool_log_records\"")
    ap.add_argument("--cache-log-slots", type=int, default=-1, help="tuning aid: option \"voronoi_cache_log_slots\"")
    ap.add_argument("--deposit-log-mb", type=int, default=-1, help="tuning aid: option \"deposit_log_mb\" (3D grids: the deposit log's size)")
    ap.add_argument("--grid-blocks", type=int, default=0)
    ap.add_argument("--block-threads", type=int, default=0)
    ap.add_argument("--no-pola", action="store_true", help="tuning aid: do not track Stokes Q,U,V")
    ap.add_argument("--frozen", action="store_true",
                    help="tuning aid: time the loop with the temperature feedback frozen to a prior "
                         "(keeps the physics identical across diagnostic builds); not the benchmark")
    args = ap.parse_args()

    par = Par(args.gpus, args.shared_device, args.force_dist)
    if par.mode == "torchrun" and par.world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE = %d" % (args.gpus, par.world))
    world = par.n_gpus
    with_cpu = world == 1 and not args.no_cpu_baseline
    launcher = {"single": "one process, one GPU", "torchrun": "one process per GPU, torch.distributed nccl (RCCL)",
                "library": "one process, mcgpu_multi_* (RCCL inside the library)"}[par.mode]
    if par.mode == "library" and par.shared_device:
        launcher = "one process, mcgpu_multi_* with every context on device 0 (shared-device dry run: NOT a scaling measurement)"

    if args.config == "sed":
        line = sed_block(par, args, args.steps, args.warmup, with_cpu, args.packets, args.sed_observers)
        if par.rank == 0:
            line["launcher"] = launcher
            if par.rccl_ranks is not None:
                line["rccl_ranks"] = par.rccl_ranks
            print(json.dumps(line))
        par.finish()
        return

    def say(name, blk):
        # One line per block as it completes, in front of the final combined line: the driver's record keeps the END of
        # stdout, and the combined line is long.  (Prefixed: the one line that starts with `{` is the contract's.)
        if par.rank == 0 and blk is not None:
            print("block %s: %s" % (name, json.dumps(blk)), flush=True)

    block, cfg = thermal_block(par, args, args.config, args.steps, args.warmup, with_cpu, args.packets)
    say("headline_" + args.config, block)
    extras = {}
    if args.config == "pascucci" and not (args.no_ref41 or args.no_pascucci or args.no_extra) and not args.frozen:
        # BASELINE.json lists ref4.1 as the single-GPU configuration (configs[1]): the same loop, same packet count,
        # same steps, barriers and all-reduce, with its own roofline, CPU baseline and temperature parity
        extras["ref41_2d"], _ = thermal_block(par, args, "ref41", args.steps, args.warmup, with_cpu, args.packets)
        say("ref41_2d", extras["ref41_2d"])
        if world == 1:   # ... and, on one GPU, every other BASELINE configuration under the same clock
            extras["ref41_3d"], _ = thermal_block(par, args, "ref41_3d", args.steps, args.warmup, with_cpu, args.packets)
            say("ref41_3d", extras["ref41_3d"])
            try:
                extras["voronoi"], _ = thermal_block(par, args, "voronoi", max(1, args.steps - 1), 1, with_cpu, args.packets)
            except Exception as ex:   # (the tessellation of the stand-in needs scipy's Qhull and a minute of host time)
                extras["voronoi"] = {"skipped": repr(ex)}
            say("voronoi", extras["voronoi"])
            # BASELINE config 4: an optically thick midplane (10x the dust) with the modified random walk, at the production
            # size (1e8 packets, one step) and at 1e7, where the launch's tail -- the serial latency of its longest packets,
            # whatever the packet count (DESIGN.md "k_tail") -- weighs ten times more
            extras["ref41_mrw"], _ = thermal_block(par, args, "ref41_mrw", 1, 1, with_cpu, min(args.packets, 1e7))
            if args.packets > 1e7:
                big, _ = thermal_block(par, args, "ref41_mrw", 1, 1, False, args.packets)
                extras["ref41_mrw"]["at_%.0e_packets" % args.packets] = {k: big[k] for k in ("value", "unit", "ms_per_step", "tail")}
            say("ref41_mrw", extras["ref41_mrw"])
            # the SED half of BASELINE config 2 at its own size: 10 observers, every wavelength, 10 000 packets in the stop bin
            # per stream (n_photons_lambda of the configuration; rounds 4 / 5 ran 710 / 3 551, where a wavelength's launches are
            # small and the rate sits below the full run's).  --packets below the default scales it down.
            extras["sed"] = sed_block(par, args, 1, 0, with_cpu, 0, args.sed_observers or 10, all_lambdas=True,
                                      n2=max(10, int(round(10000 * min(1.0, args.packets / 1e8)))))
            say("sed", extras["sed"])
            # the headline workload with option "crossing" = 1: the flight-parametric crossing in the flying waves -- NOT the
            # reference's arithmetic, so not the headline; statistical parity only (its tdust_vs_cpu against the same CPU port)
            pb, _ = thermal_block(par, args, "pascucci", args.steps, args.warmup, with_cpu, args.packets, crossing=1)
            pb["note"] = ("option \"crossing\" = 1 (off by default): wall distances as functions of the path parameter along a flight; "
                          "same cells but for ties at the rounding level, gated statistically (tests/test_param_crossing.py)")
            pb["roofline"]["kernel"] = "k_thermal_roles_param"
            extras["pascucci_parametric_crossing"] = pb
            say("pascucci_parametric_crossing", pb)
    if par.rank == 0:
        line = {"metric": "photon packets/sec (whole node), thermal MC packet loop, %s" % cfg.name,
                "value": block["value"], "unit": "packets/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": block["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": "f64", "data": "synthetic", "config": block["config"], "roofline": block["roofline"],
                "launcher": launcher}
        if par.rccl_ranks is not None:
            line["rccl_ranks"] = par.rccl_ranks
        if par.mode == "torchrun":
            line["dist"] = {"backend": par.dist.get_backend(), "world_size": par.dist.get_world_size(),
                            "device_allreduces": par.n_allreduce}
        for k in ("cpu_baseline", "tdust_vs_cpu", "binned_deposits", "tail", "tessellation", "crossing_engine", "mrw_vs_brute_force_gpu"):
            if k in block:
                line[k] = block[k]
        line.update(extras)
        print(json.dumps(line))
    par.finish()


if __name__ == "__main__":
    main()
