#!/bin/bash
# PMC passes of the pure-flight workload (Pascucci disk: 0.15 interactions per packet): instruction mix and stalls
# of the crossing loop.  Run on the GPU box; summaries land in gpurun_out/.
R=${GRAFT_REPO_ROOT:-/root/repo}
CFG=${1:-pascucci}
mkdir -p $R/gpurun_out
cd /tmp; export TMPDIR=/tmp
run() {
  local name=$1; shift
  rm -rf $R/gpurun_out/q_$name; mkdir -p $R/gpurun_out/q_$name
  timeout 600 rocprofv3 "$@" > $R/gpurun_out/q_$name.log 2>&1 </dev/null
  (cd $R && python3 tools/summarize_prof.py gpurun_out/q_$name gpurun_out pmc_${CFG}_$name > /dev/null 2>&1)
}
B="python3 $R/bench.py --config $CFG --no-cpu-baseline --no-pascucci --steps 1 --warmup 0 --packets 2e7"
run a --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAVES --kernel-trace --output-format csv -d $R/gpurun_out/q_a -o a -- $B
run b --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU --kernel-trace --output-format csv -d $R/gpurun_out/q_b -o b -- $B
run c --pmc SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_INSTS_BRANCH SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 --kernel-trace --output-format csv -d $R/gpurun_out/q_c -o c -- $B
ls $R/gpurun_out/pmc_${CFG}_*.json
