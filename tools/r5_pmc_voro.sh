#!/bin/bash
# PMC pass of one Voronoi step (bench.py --config voronoi) per schedule / block size: instructions, wave cycles, waits.
#   tools/r5_pmc_voro.sh "<bench args>" tag     -> gpurun_out/r5/pmc_<tag>.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
ARGS="$1"; TAG="$2"
mkdir -p $R/gpurun_out/r5/prof_$TAG
cd /tmp; export TMPDIR=/tmp
B="python3 $R/bench.py --config voronoi --no-cpu-baseline --steps 1 --warmup 0 --packets 20000000 $ARGS"
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $R/gpurun_out/r5/prof_$TAG/a -o a -- $B > $R/gpurun_out/r5/prof_$TAG/a.log 2>&1 </dev/null
timeout 600 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/r5/prof_$TAG/b -o b -- $B > $R/gpurun_out/r5/prof_$TAG/b.log 2>&1 </dev/null
cd $R
python3 - "$TAG" <<'PY' > gpurun_out/r5/pmc_$TAG.txt
import csv, glob, sys, collections
tag = sys.argv[1]
tot = collections.defaultdict(float)
for f in glob.glob("gpurun_out/r5/prof_%s/*/**/*counter_collection.csv" % tag, recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "thermal_voro" not in k: continue
        tot[(k.split("(")[0][:60], r["Counter_Name"])] += float(r["Counter_Value"])
for (k, c), v in sorted(tot.items()):
    print("%-62s %-24s %.6g" % (k, c, v))
PY
cat gpurun_out/r5/pmc_$TAG.txt
rm -rf gpurun_out/r5/prof_$TAG/*/*/*.csv 2>/dev/null
