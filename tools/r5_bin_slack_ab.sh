#!/bin/bash
# A/B of the deposit log's slack (k_plan_bins: want = demand * S) and the chunk size (launch_binned: c_max = F * log):
# the default (S 1.5, F 0.6) against variants built with (2.0, 0.45) and (1.25, 0.72); config 3, interleaved runs.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for rep in 1 2 3; do
  for v in default bin_a bin_b; do
    L=""; [ "$v" != "default" ] && L="$R/mcfost_amd/csrc/variants/$v.so"
    out=$(MCGPU_LIB=$L timeout 600 python bench.py --config ref41_3d --steps 3 --warmup 1 --no-extra --no-cpu-baseline 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());b=d.get('binned_deposits',{});print('%.4g'%d['value'], '%.1f ms'%d['ms_per_step'], 'chunks', b.get('chunks'), 'overflow', b.get('overflow_blocks'), 'tail', d.get('tail',{}).get('tail_ms'))")
    echo "$v | $out"
  done
done
