#!/bin/bash
mkdir -p gpurun_out/r6_xilog_prof
for obs in 3 6; do for x in 0 2; do
  timeout 600 python bench.py --config sed --steps 1 --warmup 1 --no-cpu-baseline --sed-observers $obs --sed-lambdas 35 --packets 2.5e7 --xi-log $x > gpurun_out/r6_xilog_prof/f${obs}_l35_$x.json 2> gpurun_out/r6_xilog_prof/f${obs}_l35_$x.err
  python -c "
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); print('observers', sys.argv[2], 'lambda 35 xi_log', sys.argv[3], ' %.4g packets/s  %.1f ms/step' % (d['value'], d['ms_per_step']), d.get('xi_log'))
" gpurun_out/r6_xilog_prof/f${obs}_l35_$x.json $obs $x
done; done
