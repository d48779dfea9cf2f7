#!/bin/bash
# round 6: the hand-over thresholds of a launch's tail -- role kernel -> k_tail (option "tail"), k_tail -> host ("tail_host_packets")
mkdir -p gpurun_out/r6_thr
python -c "import __graft_entry__ as g; g.build_hip()" > gpurun_out/r6_thr/build.log 2>&1
for cfg in ref41 ref41_3d; do
  for t in 8 16 24 48; do
    timeout 600 python bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline --tail $t > gpurun_out/r6_thr/${cfg}_t$t.json 2> gpurun_out/r6_thr/${cfg}_t$t.err
  done
done
for t in 8 16 24 48; do
  timeout 600 python bench.py --config ref41_mrw --packets 1e7 --steps 2 --warmup 1 --no-cpu-baseline --tail $t > gpurun_out/r6_thr/mrw_t$t.json 2> gpurun_out/r6_thr/mrw_t$t.err
done
for hp in 256 1024; do
  timeout 600 python bench.py --config ref41_mrw --packets 1e7 --steps 2 --warmup 1 --no-cpu-baseline --tail 16 --tail-host-packets $hp > gpurun_out/r6_thr/mrw_t16_hp$hp.json 2>&1
done
timeout 600 python bench.py --config ref41_mrw --packets 1e7 --steps 2 --warmup 1 --no-cpu-baseline --tail 16 --host-threads 64 --tail-host-packets 1024 > gpurun_out/r6_thr/mrw_t16_ht64.json 2>&1
timeout 600 python bench.py --config ref41_mrw --packets 1e7 --steps 2 --warmup 1 --no-cpu-baseline --tail 16 --host-threads 16 --tail-host-packets 256 > gpurun_out/r6_thr/mrw_t16_ht16.json 2>&1
for f in gpurun_out/r6_thr/*.json; do echo $f; python -c "
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); t=d.get('tail') or {}; print('  %.4g packets/s  %.1f ms/step  tail %.1f ms  host %.1f ms %d packets %d thr %.0f ns/ev  longest %d' % (d['value'], d['ms_per_step'], t.get('tail_ms',0), t.get('host_ms',0), t.get('host_packets',0), t.get('host_threads',0), t.get('host_ns_per_event_per_thread',0) or 0, t.get('longest_packet_events',0)))
" $f; done
