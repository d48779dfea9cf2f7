#!/bin/bash
# A/B timing of library variants on the GPU box: tools/ab_bench.sh <out.log> <packets> <lib>:<config>[:ENV=VAL,...] ...
out=$1; shift; n=$1; shift
mkdir -p $(dirname $out); : > $out
for spec in "$@"; do
  lib=$(echo $spec | cut -d: -f1); cfg=$(echo $spec | cut -d: -f2); envs=$(echo $spec | cut -d: -f3 | tr ',' ' ')
  libpath=mcfost_amd/csrc/libmcfost_hip.so
  [ "$lib" != "default" ] && libpath=mcfost_amd/csrc/variants/$lib.so
  extra=""
  case $cfg in *@*) extra="--sites ${cfg#*@}"; cfg=${cfg%@*};; esac   # voronoi@1000000: that many sites
  echo "== $spec" >> $out
  env MCGPU_LIB=$PWD/$libpath $envs python bench.py --config $cfg $extra --packets $n --steps 2 --warmup 1 --no-cpu-baseline --no-pascucci 2>>$out.err | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print('value %.4g pk/s  ms/step %.1f  kernel_ms %.1f  cross/pk %.1f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['crossings_per_packet']))
" >> $out
done
cat $out
