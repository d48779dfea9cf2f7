#!/bin/bash
mkdir -p gpurun_out/r6_q
python - <<'PY'
import os
print("affinity of the main thread:", len(os.sched_getaffinity(0)))
PY
for ht in 16 32 64 128; do
  timeout 600 python bench.py --config ref41_mrw --packets 1e7 --steps 2 --warmup 1 --no-cpu-baseline --host-threads $ht --tail-host-packets $((ht*16)) > gpurun_out/r6_q/mrw_ht$ht.json 2> gpurun_out/r6_q/mrw_ht$ht.err
  timeout 600 python bench.py --config ref41 --steps 3 --warmup 1 --no-cpu-baseline --host-threads $ht --tail-host-packets $((ht*16)) > gpurun_out/r6_q/ref41_ht$ht.json 2> gpurun_out/r6_q/ref41_ht$ht.err
done
for f in gpurun_out/r6_q/*.json; do echo $f; python -c "
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); t=d.get('tail') or {}; print('  %.4g packets/s  %.1f ms/step  tail %.1f ms  host %.1f ms %d packets %d thr %.0f ns/ev  longest %d' % (d['value'], d['ms_per_step'], t.get('tail_ms',0), t.get('host_ms',0), t.get('host_packets',0), t.get('host_threads',0), t.get('host_ns_per_event_per_thread',0) or 0, t.get('longest_packet_events',0)))
" $f; done
