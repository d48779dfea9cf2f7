#!/bin/bash
# round 4: PMC pass of tests/devtools/tail_pmc_r4.py (one long packet alone in k_tail, without / with the walk)
#   tools/r4_lone_pmc.sh <out.log>
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/$1
P=$R/gpurun_out/prof/lonepmc; rm -rf $P; mkdir -p $P
cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $P/a -o a -- python3 $R/tests/devtools/tail_pmc_r4.py > $P/a.log 2>&1 </dev/null
cd $R; python3 - $P > $OUT <<'PY'
import sys, csv, glob, re
P = sys.argv[1]
cases = [l.split() for l in open(P + "/a.log") if l.startswith("LONE")]
rows = []
for f in glob.glob(P + "/a/**/*counter_collection.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
# the k_tail dispatches in order: one per case
disp = {}
for r in rows:
    if "k_tail" in r["Kernel_Name"]:
        disp.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"].split("(")[0]})[r["Counter_Name"]] = float(r["Counter_Value"])
for case, (did, c) in zip(cases, sorted(disp.items())):
    ev = float(case[case.index("events") + 1])
    print(" ".join(case))
    print("  %s: per event VALU %.1f SALU %.1f LDS %.1f | wave-cycles/event %.0f (x4 = SQ cycles) | cycles per instruction %.2f | wait_frac %.2f" % (
        c["name"], c["SQ_INSTS_VALU"] / ev, c["SQ_INSTS_SALU"] / ev, c["SQ_INSTS_LDS"] / ev, 4 * c["SQ_WAVE_CYCLES"] / ev,
        4 * c["SQ_WAVE_CYCLES"] / (c["SQ_INSTS_VALU"] + c["SQ_INSTS_SALU"] + c["SQ_INSTS_LDS"]), c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]))
PY
cat $OUT
