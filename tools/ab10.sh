#!/bin/bash
# usage: ab10.sh <label>:<ENV=VAL,...> ...   (ref41, 10 steps, tuning library)
for spec in "$@"; do
  label=$(echo $spec | cut -d: -f1); envs=$(echo $spec | cut -d: -f2 | tr ',' ' ')
  env MCGPU_LIB=$PWD/mcfost_amd/csrc/variants/lib_tune.so $envs python bench.py --config ref41 --steps 10 --warmup 2 --no-cpu-baseline --no-pascucci 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print('$label', '%.4g pk/s  kernel_ms %.1f' % (d['value'], d['roofline']['kernel_ms']))
"
done
