#!/bin/bash
# round-4 A/B no. 7: the azimuthal sector out of line (default build) against the same build with atan2 (variants/nofast.so)
out=$1; mkdir -p $(dirname $out); : > $out
run() {  # label, config, extra args...
  label=$1; cfg=$2; shift 2
  python bench.py --config $cfg --steps 2 --warmup 1 --no-cpu-baseline --no-extra "$@" 2>>$out.err | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print('%-20s %-9s %.4g pk/s  kernel_ms %.1f  tail %s' % ('$label', '$cfg', d['value'], d['roofline']['kernel_ms'], json.dumps(d.get('tail'))))
" >> $out
}
run "sector(noinline)" ref41_3d
MCGPU_LIB=$PWD/mcfost_amd/csrc/variants/nofast.so run "atan2" ref41_3d
run "sector(noinline)" ref41_3d
MCGPU_LIB=$PWD/mcfost_amd/csrc/variants/nofast.so run "atan2" ref41_3d
run "mrw" ref41_mrw --packets 1e7 --steps 1
cat $out
