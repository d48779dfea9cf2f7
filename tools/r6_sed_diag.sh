#!/bin/bash
# round 6: what the SED commit kernel costs without its atomics (diagnostic flag 1 of a -DMCGPU_TUNING build: the deposits
# are staged and served, the atomic instructions are skipped), at 3 and 10 observers
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/r6_diag; cd $R
export MCGPU_LIB=$R/mcfost_amd/csrc/variants/tuning.so
for incl in 3 10; do
  for fl in 0 1 4; do
    MCGPU_DIAG_FLAGS=$fl python tools/mono_timing.py --n2 3000 --incl $incl --xi-bytes 4 --lams 4,10,16,19,22,25,31,40 2>/dev/null | python -c "
import sys,json
t=0
for l in sys.stdin:
    if l.startswith('{'): t+=json.loads(l)['stream_ms']
print('observers $incl diag flags $fl: %.1f ms for 8 wavelengths' % t)"
  done
done | tee gpurun_out/r6_diag/log.txt
