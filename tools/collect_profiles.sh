#!/bin/bash
# Run on the GPU box (gpurun): rocprofv3 kernel traces of the three bench workloads and the two HBM PMC
# passes of the default bench command; summaries land in gpurun_out/ (copy them to profiles/).
# Counters are collected in their own runs, with --kernel-trace only (MI355X guide, HBM section).
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r01}
mkdir -p $R/gpurun_out
cd /tmp; export TMPDIR=/tmp
run() { # name, then the rocprofv3 options, then -- program
  local name=$1; shift
  rm -rf $R/gpurun_out/p_$name; mkdir -p $R/gpurun_out/p_$name
  timeout 900 rocprofv3 "$@" > $R/gpurun_out/p_$name.log 2>&1 </dev/null
  (cd $R && python3 tools/summarize_prof.py gpurun_out/p_$name gpurun_out ${TAG}_$name > /dev/null 2>&1)
}
run kt       --kernel-trace --stats --output-format csv -d $R/gpurun_out/p_kt -o kt -- python3 $R/bench.py --no-cpu-baseline --no-pascucci
run pmc_fetch --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/p_pmc_fetch -o f -- python3 $R/bench.py --no-cpu-baseline --no-pascucci --steps 1 --warmup 0
run pmc_write --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/p_pmc_write -o w -- python3 $R/bench.py --no-cpu-baseline --no-pascucci --steps 1 --warmup 0
run kt_voro  --kernel-trace --stats --output-format csv -d $R/gpurun_out/p_kt_voro -o kt -- python3 $R/bench.py --config voronoi --sites 50000 --packets 2e7 --no-cpu-baseline
run kt_sed   --kernel-trace --stats --output-format csv -d $R/gpurun_out/p_kt_sed -o kt -- python3 $R/bench.py --config sed --packets 2e7 --no-cpu-baseline
ls -la $R/gpurun_out/${TAG}_*.json
