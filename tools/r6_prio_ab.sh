#!/bin/bash
# round 6: the issue priority of a role kernel's waves by role (s_setprio; -DMCGPU_PRIO_SERVE / _FLY builds), A/B on the
# default library
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/r6_prio
cd $R
for cfg in pascucci ref41 ref41_3d; do
  for rep in 1 2; do
    python bench.py --config $cfg --steps 10 --warmup 2 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print('$cfg default', '%.4g pk/s  ms %.2f kernel_ms %.2f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms']))
"
    bash tools/ab_lib.sh $cfg 10 serve2:prio_s2.so fly2:prio_f2.so
  done
done 2>&1 | tee gpurun_out/r6_prio/ab.log
