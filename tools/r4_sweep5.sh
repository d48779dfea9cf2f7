#!/bin/bash
# round-4 A/B no. 5: the deferred 3D re-index; the order of the Voronoi sites
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
run "defer" ref41_3d
run "order=morton" voronoi --site-order morton
run "order=file" voronoi
cat $out
