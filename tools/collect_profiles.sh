#!/bin/bash
# Profile collection on the GPU box (run from the repo root through gpurun):
#   tools/collect_profiles.sh [config ...]        (default: pascucci ref41)
# Per configuration: one rocprofv3 kernel trace with --stats of `python3 bench.py --config C` and four PMC passes
# (separate runs, counters only with --kernel-trace, as the MI355X guide prescribes) of the same command with
# --steps 1 --warmup 0.  tools/summarize_prof.py turns them into gpurun_out/r06_kt_C.json / r06_pmc_C.json, which
# are then copied to profiles/ and committed.
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/prof
cd /tmp; export TMPDIR=/tmp
CFGS="$@"; [ -z "$CFGS" ] && CFGS="pascucci ref41"
for C in $CFGS; do
  NP=100000000; [ "$C" = "ref41_mrw" ] && NP=10000000     # (the default line runs the thick disk with 1e7 packets)
  EXTRA=""; [ "$C" = "sed" ] && EXTRA="--sed-observers 10"   # (the default line's SED block runs ten observers, BASELINE config 2's)
  B="python3 $R/bench.py --config $C --no-cpu-baseline --no-extra --packets $NP $EXTRA"
  export PROF_CMD="python3 bench.py --config $C --no-cpu-baseline --no-extra --packets $NP $EXTRA"
  P=$R/gpurun_out/prof/$C
  rm -rf $P; mkdir -p $P
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $P/kt -o kt -- $B > $P/kt.log 2>&1 </dev/null
  timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $P/f -o f -- $B --steps 1 --warmup 0 > $P/f.log 2>&1 </dev/null
  timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $P/w -o w -- $B --steps 1 --warmup 0 > $P/w.log 2>&1 </dev/null
  timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $P/a -o a -- $B --steps 1 --warmup 0 > $P/a.log 2>&1 </dev/null
  timeout 600 rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $P/b -o b -- $B --steps 1 --warmup 0 > $P/b.log 2>&1 </dev/null
  (cd $R && python3 tools/summarize_prof.py gpurun_out/prof/$C $C gpurun_out $NP)
done
ls -la $R/gpurun_out/r06_*.json
