#!/bin/bash
# A/B of the flight-parametric crossing (option "crossing" = 1) against the default: tools/r5_param_ab.sh [config ...]
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for cfg in "$@"; do
  for c in 0 1 0 1; do
    out=$(timeout 600 python bench.py --config $cfg --steps 3 --warmup 1 --no-extra --cpu-seconds 6 --crossing $c 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());t=d.get('tdust_vs_cpu',{});print(d['value'], d['roofline']['kernel_ms'], d['config']['crossings_per_packet'], d['config']['interactions_per_packet'], t.get('rel_rms'), t.get('tolerance_rel_rms'), t.get('p75'), t.get('ok'))")
    echo "$cfg crossing=$c | $out"
  done
done
