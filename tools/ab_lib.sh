#!/bin/bash
# usage: ab_lib.sh <config> <steps> <label>:<library under mcfost_amd/csrc/variants>[:ENV=VAL,...] ...
cfg=$1; steps=$2; shift 2
for spec in "$@"; do
  label=$(echo $spec | cut -d: -f1); lib=$(echo $spec | cut -d: -f2); envs=$(echo $spec | cut -s -d: -f3 | tr ',' ' ')
  env MCGPU_LIB=$PWD/mcfost_amd/csrc/variants/$lib $envs python bench.py --config $cfg --steps $steps --warmup 1 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print('$cfg $label', '%.4g pk/s  kernel_ms %.1f' % (d['value'], d['roofline']['kernel_ms']))
"
done
