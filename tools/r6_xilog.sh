#!/bin/bash
# round 6: the SED commit pass's deposits as log + fold (option xi_log 1) against atomics (0)
mkdir -p gpurun_out/r6_xilog
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sed_mode or end_to_end" 2>&1 | tail -15 > gpurun_out/r6_xilog/pytest_sed.log
cat gpurun_out/r6_xilog/pytest_sed.log
for x in 0 1; do
  timeout 900 python bench.py --config sed --steps 1 --warmup 1 --no-cpu-baseline --sed-observers 10 --xi-log $x > gpurun_out/r6_xilog/sed10_log$x.json 2> gpurun_out/r6_xilog/sed10_log$x.err
  timeout 900 python bench.py --config sed --steps 1 --warmup 1 --no-cpu-baseline --xi-log $x > gpurun_out/r6_xilog/sed3_log$x.json 2> gpurun_out/r6_xilog/sed3_log$x.err
done
for f in gpurun_out/r6_xilog/*.json; do echo $f; python -c "
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); print('  %.4g packets/s  %.1f ms/step  %s' % (d['value'], d['ms_per_step'], d['config']['workload'][-70:]))
" $f; done; tail -3 gpurun_out/r6_xilog/*.err
