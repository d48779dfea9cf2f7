#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/r6_xilog_prof
for x in 1 0; do
  timeout 900 python bench.py --config sed --steps 1 --warmup 1 --no-cpu-baseline --sed-observers 10 --xi-log $x > gpurun_out/r6_xilog_prof/b10_log$x.json 2> gpurun_out/r6_xilog_prof/b10_log$x.err
  python -c "
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); print('xi_log', sys.argv[2], ' %.4g packets/s  %.1f ms/step  cross/packet %.1f' % (d['value'], d['ms_per_step'], d['config']['crossings_per_packet']), d.get('xi_log'))
" gpurun_out/r6_xilog_prof/b10_log$x.json $x
done
cd /tmp; export TMPDIR=/tmp
P=$R/gpurun_out/r6_xilog_prof/log1b
rm -rf $P; mkdir -p $P
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $P -o kt -- python3 $R/bench.py --config sed --steps 1 --warmup 0 --no-cpu-baseline --sed-observers 10 --xi-log 1 > $P/run.log 2>&1 </dev/null
f=$(find $P -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:8]:
    print("%-90s calls %6s total %10.1f ms avg %9.3f ms" % (r["Name"][:90], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6))
PY
