#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/r6_xilog_prof
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_variable_dust_sed.py tests/test_spherical_sed_modes.py tests/test_image_pipeline.py -x -q -m gpu -k "sed or end_to_end or image" 2>&1 | tail -5
for lam in 5 25 35; do
  timeout 900 python bench.py --config sed --steps 1 --warmup 1 --no-cpu-baseline --sed-observers 10 --sed-lambdas $lam --packets 2.5e7 > gpurun_out/r6_xilog_prof/e10_l${lam}.json 2> gpurun_out/r6_xilog_prof/e10_l${lam}.err
  python -c "
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); print('lambda', sys.argv[2], 'auto  %.4g packets/s  %.1f ms/step' % (d['value'], d['ms_per_step']), d.get('xi_log'))
" gpurun_out/r6_xilog_prof/e10_l${lam}.json $lam
done
timeout 900 python tools/run_config2.py 1e8 10000 4 2>&1 | tee gpurun_out/r6_xilog_prof/config2_auto.log | tail -8
