#!/bin/bash
# Round 5, end of round (run through gpurun from the repo root, sources frozen): the kernel traces and counter passes of every
# single-GPU configuration of the default line (tools/collect_profiles.sh), and config 4 once at 1e8 packets, where the
# tail of its longest packet is the share it would have in a production run (the default line runs it at 1e7).
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/r5
cd $R
python3 bench.py --config ref41_mrw --packets 100000000 --steps 1 --warmup 0 --no-cpu-baseline --no-extra > gpurun_out/r5/config4_1e8.json 2> gpurun_out/r5/config4_1e8.err
bash tools/collect_profiles.sh pascucci ref41 ref41_3d voronoi ref41_mrw sed > gpurun_out/r5/collect.log 2>&1
ls -la gpurun_out/r05_*.json | tail -20
