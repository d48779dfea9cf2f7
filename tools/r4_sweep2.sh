#!/bin/bash
# round-4 A/B no. 2: the serving-side cuts on every thermal configuration, tail thresholds on ref4.1
out=$1; mkdir -p $(dirname $out); : > $out
run() {  # label, config, extra args...
  label=$1; cfg=$2; shift 2
  python bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline --no-extra "$@" 2>>$out.err | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print('%-28s %-9s %.4g pk/s  kernel_ms %.1f  tail %s' % ('$label', '$cfg', d['value'], d['roofline']['kernel_ms'], json.dumps(d.get('tail'))))
" >> $out
}
run default pascucci
run default ref41
run default ref41_3d
run default ref41_mrw --packets 1e7 --steps 1
cat $out
