#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/r6_xilog_prof
cd /tmp; export TMPDIR=/tmp
for x in 1 0; do
  P=$R/gpurun_out/r6_xilog_prof/log$x
  rm -rf $P; mkdir -p $P
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $P -o kt -- python3 $R/bench.py --config sed --steps 1 --warmup 0 --no-cpu-baseline --sed-observers 10 --xi-log $x > $P/run.log 2>&1 </dev/null
  f=$(find $P -name "*kernel_stats.csv" | head -1)
  echo "== xi_log $x"; tail -1 $P/run.log | cut -c1-200
  python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    print("%-90s calls %6s total %10.1f ms avg %9.3f ms" % (r["Name"][:90], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6))
PY
  cp "$f" $R/gpurun_out/r6_xilog_prof/kernel_stats_log$x.csv
done
