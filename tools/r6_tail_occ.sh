#!/bin/bash
# round 6: k_tail's register budget (waves per SIMD) -- it is throughput-bound while it holds more packets than waves reside
mkdir -p gpurun_out/r6_occ
for v in base tailw3 tailw4; do
  if [ $v = base ]; then unset MCGPU_LIB; else export MCGPU_LIB=$PWD/mcfost_amd/csrc/variants/$v.so; fi
  for rep in 1 2; do
  timeout 600 python bench.py --config ref41 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r6_occ/ref41_${v}_$rep.json 2> gpurun_out/r6_occ/ref41_$v.err
  timeout 600 python bench.py --config ref41_3d --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r6_occ/ref41_3d_${v}_$rep.json 2> gpurun_out/r6_occ/ref41_3d_$v.err
  timeout 600 python bench.py --config ref41_mrw --packets 1e7 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r6_occ/mrw_${v}_$rep.json 2> gpurun_out/r6_occ/mrw_$v.err
  done
done
for f in gpurun_out/r6_occ/*.json; do echo $f; python -c "
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); t=d.get('tail') or {}; print('  %.4g packets/s  %.1f ms/step  tail %.1f ms  host %.1f ms %d packets %d thr %.0f ns/ev  longest %d' % (d['value'], d['ms_per_step'], t.get('tail_ms',0), t.get('host_ms',0), t.get('host_packets',0), t.get('host_threads',0), t.get('host_ns_per_event_per_thread',0) or 0, t.get('longest_packet_events',0)))
" $f; done
