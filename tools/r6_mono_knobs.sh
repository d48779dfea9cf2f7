#!/bin/bash
# round 6: the SED kernel's loop knobs (min_active: leave the crossing loop when fewer lanes than this fraction of 64 still
# fly; inner_iters) at 3 and 10 observers, -DMCGPU_TUNING build
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/r6_knobs
cd $R
export MCGPU_LIB=$R/mcfost_amd/csrc/variants/tuning.so
for incl in 3 10; do
  for ma in 8 16 24 32 40 48; do
    MCGPU_MIN_ACTIVE=$ma python tools/mono_timing.py --n2 3000 --incl $incl --xi-bytes 4 --lams 4,10,16,19,22,25,31,40 2>/dev/null | python -c "
import sys,json
t=0
for l in sys.stdin:
    if l.startswith('{'): t+=json.loads(l)['stream_ms']
print('observers $incl min_active $ma: %.1f ms for 8 wavelengths' % t)"
  done
  for ii in 16 256; do
    MCGPU_INNER_ITERS=$ii python tools/mono_timing.py --n2 3000 --incl $incl --xi-bytes 4 --lams 4,10,16,19,22,25,31,40 2>/dev/null | python -c "
import sys,json
t=0
for l in sys.stdin:
    if l.startswith('{'): t+=json.loads(l)['stream_ms']
print('observers $incl inner_iters $ii: %.1f ms for 8 wavelengths' % t)"
  done
done | tee gpurun_out/r6_knobs/log.txt
