#!/bin/bash
# round-4 A/B no. 8: the SED mode's commit pass -- LDS tile and global atomics by address space (default build) against the
# volatile / flat version before it (variants/sedfix3.so)
out=$1; mkdir -p $(dirname $out); : > $out
run() {  # label, extra args...
  label=$1; shift
  python bench.py --config sed --no-cpu-baseline --no-extra "$@" 2>>$out.err | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print('%-12s %.4g pk/s  ms %.1f  records/s %.4g  lines/s %.4g  %s' % ('$label', d['value'], d['ms_per_step'], d['config'].get('records_per_s', 0), d['roofline'].get('atomic_line_ops_per_s', 0), d['config']['workload'][:90]))
" >> $out
}
run new
MCGPU_LIB=$PWD/mcfost_amd/csrc/variants/sedfix3.so run fix3
run new
MCGPU_LIB=$PWD/mcfost_amd/csrc/variants/sedfix3.so run fix3
cat $out
