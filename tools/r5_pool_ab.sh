#!/bin/bash
# A/B of pool builds: tools/r5_pool_ab.sh "<lib or ->:<bench args>" ...   (one line per run: packets/s, kernel ms)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for spec in "$@"; do
  lib="${spec%%:*}"; args="${spec#*:}"
  if [ "$lib" = "-" ]; then unset MCGPU_LIB; else export MCGPU_LIB=$R/mcfost_amd/csrc/variants/$lib.so; fi
  out=$(timeout 600 python bench.py --config voronoi --steps 1 --warmup 1 --no-cpu-baseline --packets 4e7 $args 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print(d['value'], d['roofline']['kernel_ms'])")
  echo "$lib | $args | $out"
done
