#!/bin/bash
# Round 5, late: the crossing's 64-bit literals out of the flying loop (copysign for the +-e factor and the wall's sign,
# rare branches for the zero fixes, one select for the azimuthal wall).  Headline and the two ref4.1 configurations, and the
# frozen parity tests of the crossings.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for cfg in pascucci ref41 ref41_3d; do
  for rep in 1 2; do
    out=$(timeout 600 python bench.py --config $cfg --steps 4 --warmup 1 --no-extra --no-cpu-baseline 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('%.4g'%d['value'], '%.1f ms'%d['ms_per_step'], 'kernel', d['roofline'].get('kernel_ms'), 'tail', (d.get('tail') or {}).get('tail_ms'))")
    echo "$cfg | $out"
  done
done
python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "frozen or golden or cross_cell or tail or live_mode" 2>&1 | tail -3
