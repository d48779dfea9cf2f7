#!/bin/bash
# Memory-pipeline counters of one Voronoi step: tools/r5_pmc_mem.sh "<bench args>" tag
R=${GRAFT_REPO_ROOT:-/root/repo}
ARGS="$1"; TAG="$2"
mkdir -p $R/gpurun_out/r5/mem_$TAG
cd /tmp; export TMPDIR=/tmp
B="python3 $R/bench.py --config voronoi --no-cpu-baseline --steps 1 --warmup 0 --packets 20000000 $ARGS"
i=0
for set in "TA_TA_BUSY_sum TA_BUSY_max TCP_GATE_EN1_sum TCP_GATE_EN2_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCP_TA_TCP_STATE_READ_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TA_BUFFER_WAVEFRONTS_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/r5/mem_$TAG/p$i -o p -- $B > $R/gpurun_out/r5/mem_$TAG/p$i.log 2>&1 </dev/null
done
cd $R
python3 - "$TAG" <<'PY' > gpurun_out/r5/mem_$TAG.txt
import csv, glob, sys, collections
tag = sys.argv[1]
tot = collections.defaultdict(float)
for f in glob.glob("gpurun_out/r5/mem_%s/*/**/*counter_collection.csv" % tag, recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "thermal_voro" not in k: continue
        tot[(k.split("(")[0][:50], r["Counter_Name"])] += float(r["Counter_Value"])
for (k, c), v in sorted(tot.items()):
    print("%-52s %-40s %.6g" % (k, c, v))
PY
cat gpurun_out/r5/mem_$TAG.txt
grep -il "error\|invalid\|not found" gpurun_out/r5/mem_$TAG/*.log | head
rm -rf gpurun_out/r5/mem_$TAG/p*/
