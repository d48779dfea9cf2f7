#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/r6_packed
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_variable_dust_sed.py tests/test_multi_shared_device.py tests/test_rccl_single_rank.py tests/test_image_pipeline.py -x -q -m gpu -k "sed or end_to_end or image or mono or xI or wavelength" 2>&1 | tail -8
for lam in 5 25 35; do
 for x in 0 1; do
  timeout 900 python bench.py --config sed --steps 1 --warmup 1 --no-cpu-baseline --sed-observers 10 --sed-lambdas $lam --packets 2.5e7 --xi-log $x > gpurun_out/r6_packed/p10_l${lam}_$x.json 2> gpurun_out/r6_packed/p10_l${lam}_$x.err
  python -c "
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); print('lambda', sys.argv[2], 'xi_log', sys.argv[3], ' %.4g packets/s  %.1f ms/step' % (d['value'], d['ms_per_step']), d.get('xi_log'))
" gpurun_out/r6_packed/p10_l${lam}_$x.json $lam $x
 done
done
timeout 900 python bench.py --config sed --steps 1 --warmup 1 --no-cpu-baseline --xi-log 0 > gpurun_out/r6_packed/p3.json 2> gpurun_out/r6_packed/p3.err
python -c "
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); print('3 observers, 4 wavelengths, atomics  %.4g packets/s  %.1f ms/step' % (d['value'], d['ms_per_step']))
" gpurun_out/r6_packed/p3.json
timeout 900 python tools/run_config2.py 1e8 10000 4 2>&1 | tee gpurun_out/r6_packed/config2.log | tail -8
