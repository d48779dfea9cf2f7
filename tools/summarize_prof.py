"""Summarise the rocprofv3 CSV output of tools/collect_profiles.sh for one configuration into
<dst>/r06_kt_<config>.json (kernel trace statistics) and <dst>/r06_pmc_<config>.json (counters of the dominant
kernel per launch, derived figures, and the hash of the kernel sources they were measured on: bench.py only
reports them while that hash matches).

    python3 tools/summarize_prof.py <dir with kt/ f/ w/ a/ b/> <config> <dst>
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
src, config, dst = sys.argv[1], sys.argv[2], sys.argv[3]
KERNELS = ("k_thermal", "k_mono", "k_fold_bins", "k_tail")
ROUND = "r06"


def dominant(name):
    return any(k in name for k in KERNELS)


# ---- kernel trace -----------------------------------------------------------------------------------------
kt = {"config": config, "command": os.environ.get("PROF_CMD", "python3 bench.py --config %s --no-cpu-baseline --no-extra" % config).strip()}
for f in glob.glob(os.path.join(src, "kt", "**", "*kernel_stats.csv"), recursive=True):
    kt["kernel_stats"] = list(csv.DictReader(open(f)))
dur = collections.defaultdict(list)
for f in glob.glob(os.path.join(src, "kt", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"][:120]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        if dominant(r["Kernel_Name"]):
            kt["launch"] = {k: r.get(k) for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size",
                                                   "Scratch_Size", "Workgroup_Size", "Grid_Size") if k in r}
kt["kernel_durations_ns"] = {k: {"calls": len(v), "avg": sum(v) / len(v), "min": min(v), "max": max(v)} for k, v in dur.items()}
json.dump(kt, open(os.path.join(dst, "%s_kt_%s.json" % (ROUND, config)), "w"), indent=1)

# ---- counters ---------------------------------------------------------------------------------------------
# Every pass runs ONE step (--steps 1 --warmup 0); a step may be several launches (the chunks of a run with binned
# deposits and the folds between them): the counters and the kernel time are summed over the launches of the step.
cnt, launches, dur_pass = collections.defaultdict(float), collections.defaultdict(int), []
for sub in ("f", "w", "a", "b"):
    for f in glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if dominant(r["Kernel_Name"]):
                cnt[r["Counter_Name"]] += float(r["Counter_Value"])
                launches[r["Counter_Name"]] += 1
    for f in glob.glob(os.path.join(src, sub, "**", "*kernel_trace.csv"), recursive=True):
        tot = 0
        for r in csv.DictReader(open(f)):
            if dominant(r["Kernel_Name"]):
                tot += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        if tot:
            dur_pass.append(tot)
per = dict(cnt)
dur_pmc = dur_pass
n_packets = int(float(sys.argv[4])) if len(sys.argv) > 4 else 100000000
# (the SED step's packets are what its streams needed, not --packets: read them from the bench line of the trace run)
try:
    for line in open(os.path.join(src, "kt.log")):
        if line.startswith("{") and '"packets_per_gpu_per_step"' in line:
            n_packets = int(json.loads(line)["config"]["packets_per_gpu_per_step"])
except (OSError, ValueError, KeyError):
    pass
out = {"config": config, "command": kt["command"] + " --steps 1 --warmup 0", "packets": n_packets,
       "counters_per_step": per, "launches_per_step": dict(launches)}
try:
    from bench import source_hash
    out["source_hash"] = source_hash()
except Exception as e:  # pragma: no cover
    out["source_hash"] = None
d = {}
t_ns = sum(dur_pmc) / len(dur_pmc) if dur_pmc else None   # kernel time under the profiler (slower clock than unprofiled)
if t_ns:
    d["kernel_ms_profiled"] = t_ns * 1e-6
if "FETCH_SIZE" in per and "WRITE_SIZE" in per:   # KB; FETCH_SIZE reads half the bytes on gfx950 (guide, HBM section)
    d["hbm_bytes"] = (2.0 * per["FETCH_SIZE"] + per["WRITE_SIZE"]) * 1024.0
    d["fetch_bytes_corrected"] = 2.0 * per["FETCH_SIZE"] * 1024.0
    d["write_bytes"] = per["WRITE_SIZE"] * 1024.0
if per.get("SQ_WAVE_CYCLES"):
    d["wait_frac"] = per.get("SQ_WAIT_ANY", 0.0) / per["SQ_WAVE_CYCLES"]                  # waves parked (s_waitcnt, sleep)
    d["issue_stall_frac"] = per.get("SQ_WAIT_INST_ANY", 0.0) / per["SQ_WAVE_CYCLES"]
    d["valu_active_per_wave"] = per.get("SQ_ACTIVE_INST_VALU", 0.0) / per["SQ_WAVE_CYCLES"]
if per.get("SQ_ACTIVE_INST_VALU"):
    d["lane_utilisation"] = per.get("SQ_THREAD_CYCLES_VALU", 0.0) / (64.0 * per["SQ_ACTIVE_INST_VALU"])
if per.get("GRBM_GUI_ACTIVE"):
    cyc = per["GRBM_GUI_ACTIVE"] / 8.0     # rocprofv3 sums the 8 XCDs
    n_simd = 256 * 4
    if t_ns:
        d["clock_ghz_profiled"] = cyc / t_ns
    # SQ_* cycle counters are in quad-cycles (guide, cycle constants)
    if per.get("SQ_ACTIVE_INST_VALU"):
        d["valu_busy"] = 4.0 * per["SQ_ACTIVE_INST_VALU"] / (n_simd * cyc)
    if per.get("SQ_WAVE_CYCLES"):
        d["waves_per_simd"] = 4.0 * per["SQ_WAVE_CYCLES"] / (n_simd * cyc)
f64 = 2.0 * per.get("SQ_INSTS_VALU_FMA_F64", 0.0) + per.get("SQ_INSTS_VALU_MUL_F64", 0.0) + \
    per.get("SQ_INSTS_VALU_ADD_F64", 0.0) + per.get("SQ_INSTS_VALU_TRANS_F64", 0.0)
if f64 and t_ns:
    d["fp64_tflops"] = f64 * 64.0 * d.get("lane_utilisation", 1.0) / t_ns * 1e-3
if per.get("SQ_INSTS_VALU"):
    d["insts_per_packet"] = {"valu": per["SQ_INSTS_VALU"] / n_packets, "salu": per.get("SQ_INSTS_SALU", 0.0) / n_packets,
                             "lds": per.get("SQ_INSTS_LDS", 0.0) / n_packets}
out["per_launch"] = d
json.dump(out, open(os.path.join(dst, "%s_pmc_%s.json" % (ROUND, config)), "w"), indent=1)
print(json.dumps(out, indent=1)[:2500])
