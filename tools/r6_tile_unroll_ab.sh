#!/bin/bash
# round 6: how many rounds of tile reads the deposit's serve loop keeps in flight (TILE_UNROLL builds), config 2 at 3 and 10 inclinations
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/r6_tile; cd $R
for o in 3 10; do
  for v in default tile2 tile8 tile16; do
    if [ $v = default ]; then unset MCGPU_LIB; else export MCGPU_LIB=$R/mcfost_amd/csrc/variants/$v.so; fi
    echo "== $o inclinations, $v: $(python tools/run_config2.py 1e8 10000 4 $o 2>&1 | grep 'SED Monte' | sed 's/.*packets in //; s/ (.*//')"
  done
done | tee gpurun_out/r6_tile/ab.log
