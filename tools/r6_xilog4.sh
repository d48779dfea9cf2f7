#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/r6_xilog_prof
for lam in 5 15 25 35; do
for x in 1 0; do
  timeout 900 python bench.py --config sed --steps 1 --warmup 1 --no-cpu-baseline --sed-observers 10 --xi-log $x --sed-lambdas $lam --packets 2.5e7 > gpurun_out/r6_xilog_prof/d10_l${lam}_log$x.json 2> gpurun_out/r6_xilog_prof/d10_l${lam}_log$x.err
  python -c "
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); print('lambda', sys.argv[3], 'xi_log', sys.argv[2], ' %.4g packets/s  %.1f ms/step  cross/packet %.1f' % (d['value'], d['ms_per_step'], d['config']['crossings_per_packet']), d.get('xi_log'))
" gpurun_out/r6_xilog_prof/d10_l${lam}_log$x.json $x $lam
done
done
