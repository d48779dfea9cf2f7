#!/bin/bash
# round-4: compiler-flag variants of the library (python __graft_entry__.py variant fl_<name> <flags>) against the default build
out=$1; shift; mkdir -p $(dirname $out); : > $out
run() {  # label, config
  label=$1; cfg=$2; shift 2
  python bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline --no-extra "$@" 2>>$out.err | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print('%-14s %-9s %.4g pk/s  kernel_ms %.1f' % ('$label', '$cfg', d['value'], d['roofline']['kernel_ms']))
" >> $out
}
for cfg in pascucci ref41 ref41_3d; do
  run default $cfg
  for v in "$@"; do MCGPU_LIB=$PWD/mcfost_amd/csrc/variants/fl_$v.so run $v $cfg; done
done
cat $out
