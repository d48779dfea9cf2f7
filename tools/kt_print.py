"""Prints a rocprofv3 kernel trace (CSV) in launch order: name, start, duration.  python3 tools/kt_print.py <dir> [max_rows]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
lim = int(sys.argv[2]) if len(sys.argv) > 2 else 10**9
for r in rows[:lim]:
    if "rocclr" in r["Kernel_Name"]:
        continue
    print("%-44s start %9.3f ms dur %9.3f ms" % (r["Kernel_Name"][:44], (int(r["Start_Timestamp"]) - t0) / 1e6,
                                                   (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
