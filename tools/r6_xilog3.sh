#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/r6_xilog_prof
export MCGPU_LIB=$R/mcfost_amd/csrc/variants/xlogtune.so
for t in 256 512 768; do
  export MCGPU_XLOG_CU_THREADS=$t
  timeout 900 python bench.py --config sed --steps 1 --warmup 1 --no-cpu-baseline --sed-observers 10 --xi-log 1 > gpurun_out/r6_xilog_prof/c10_t$t.json 2> gpurun_out/r6_xilog_prof/c10_t$t.err
  python -c "
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); print('cu_threads', sys.argv[2], ' %.4g packets/s  %.1f ms/step' % (d['value'], d['ms_per_step']), d.get('xi_log'))
" gpurun_out/r6_xilog_prof/c10_t$t.json $t
done
