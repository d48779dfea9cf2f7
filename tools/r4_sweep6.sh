#!/bin/bash
# round-4 A/B no. 6: the azimuthal sector in default real (3D), the walk's reuse of the absorption's temperature bracket (config 4)
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
run "sector" ref41_3d
run "sector" ref41_3d
run "reuse" ref41_mrw --packets 1e7 --steps 1
run "base" ref41
cat $out
