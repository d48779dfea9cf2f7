"""Summarise rocprofv3 CSV output (kernel stats and PMC counters) into small text files
under profiles/ (tuning aid)."""
import collections, csv, glob, json, os, sys
src, dst, tag = sys.argv[1], sys.argv[2], sys.argv[3]
os.makedirs(dst, exist_ok=True)
out = {}
for f in glob.glob(os.path.join(src, "**", "*kernel_stats.csv"), recursive=True):
    rows = list(csv.DictReader(open(f)))
    out.setdefault("kernel_stats", []).extend(rows)
for f in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True):
    acc = collections.defaultdict(float); n = collections.defaultdict(int)
    for r in csv.DictReader(open(f)):
        if "k_thermal" in r["Kernel_Name"] or "k_mono" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    for k in acc:
        out.setdefault("pmc_k_thermal_sum_over_launches", {})[k] = acc[k]
        out.setdefault("pmc_k_thermal_launches", {})[k] = n[k]
json.dump(out, open(os.path.join(dst, tag + ".json"), "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
