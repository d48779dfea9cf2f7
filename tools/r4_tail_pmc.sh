#!/bin/bash
# round 4: instructions per event of k_tail under load -- every packet handed over at its emission (--tail 1000000), PMC pass
#   tools/r4_tail_pmc.sh <config> <packets> <out.log>
R=${GRAFT_REPO_ROOT:-/root/repo}
C=$1; NP=$2; OUT=$R/$3
P=$R/gpurun_out/prof/tailpmc_$C; rm -rf $P; mkdir -p $P
cd /tmp; export TMPDIR=/tmp
B="python3 $R/bench.py --config $C --no-cpu-baseline --no-extra --packets $NP --tail 1000000 --steps 1 --warmup 0"
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $P/a -o a -- $B > $P/a.log 2>&1 </dev/null
cd $R; python3 - $P $C $NP > $OUT <<'PY'
import sys, csv, glob, json, collections
P, C, NP = sys.argv[1], sys.argv[2], float(sys.argv[3])
line = None
for l in open(P + "/a.log"):
    if l.startswith("{"):
        try: line = json.loads(l)
        except Exception: pass
tot = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(P + "/a/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        tot[r["Kernel_Name"].split("(")[0][:60]][r["Counter_Name"]] += float(r["Counter_Value"])
ev = None
if line:
    cfg = line["config"]
    ev = NP * (cfg["crossings_per_packet"] + cfg["interactions_per_packet"])
    print("config", C, "packets", NP, "events", ev, "kernel_ms", line["roofline"]["kernel_ms"], "tail", line.get("tail"))
for k, c in tot.items():
    print(k, {n: v for n, v in c.items()})
    if ev and "k_tail" in k:
        print("  per event: VALU %.1f SALU %.1f LDS %.1f  wave-cycles %.0f  wait_frac %.2f" % (
            c["SQ_INSTS_VALU"] / ev, c["SQ_INSTS_SALU"] / ev, c["SQ_INSTS_LDS"] / ev, c["SQ_WAVE_CYCLES"] / ev * 4,
            c["SQ_WAIT_ANY"] / max(c["SQ_WAVE_CYCLES"], 1)))
PY
cat $OUT
