#!/bin/bash
# round-4 A/B no. 4: the hand-over threshold of the tail kernel
out=$1; mkdir -p $(dirname $out); : > $out
run() {  # label, config, extra args...
  label=$1; cfg=$2; shift 2
  python bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline --no-extra "$@" 2>>$out.err | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print('%-20s %-9s %.4g pk/s  kernel_ms %.1f  tail %s' % ('$label', '$cfg', d['value'], d['roofline']['kernel_ms'], json.dumps(d.get('tail'))))
" >> $out
}
for t in 24 96 192 384 768; do run "tail=$t" ref41 --tail $t; done
for t in 96 384; do run "tail=$t" ref41_3d --tail $t; done
for t in 96 384; do run "tail=$t" ref41_mrw --tail $t --packets 1e7 --steps 1; done
cat $out
