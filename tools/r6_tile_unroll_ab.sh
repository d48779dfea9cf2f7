#!/bin/bash
# round 6: how many rounds of tile reads the deposit's serve loop keeps in flight, config 2 at 3 and 10 inclinations.
# (The variant libraries were built with TILE_UNROLL = 2 / 8 / 16 in mc_mono.hip.h, a constexpr there: 8 gained 1 %, the
# default stayed 4 -- profiles/r06_sed_tile_unroll_ab.log.)
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/r6_tile; cd $R
for o in 3 10; do
  for v in default tile2 tile8 tile16; do
    if [ $v = default ]; then unset MCGPU_LIB; else export MCGPU_LIB=$R/mcfost_amd/csrc/variants/$v.so; fi
    echo "== $o inclinations, $v: $(python tools/run_config2.py 1e8 10000 4 $o 2>&1 | grep 'SED Monte' | sed 's/.*packets in //; s/ (.*//')"
  done
done | tee gpurun_out/r6_tile/ab.log
