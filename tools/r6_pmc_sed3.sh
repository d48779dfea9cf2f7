#!/bin/bash
# round 6: counters of the SED commit kernel at THREE observers (one line per crossing: not atomics-bound; what is it?)
R=${GRAFT_REPO_ROOT:-/root/repo}
P=$R/gpurun_out/r6_pmc_sed3
rm -rf $P; mkdir -p $P
cd /tmp; export TMPDIR=/tmp
B="python3 $R/bench.py --config sed --no-cpu-baseline --no-extra --packets 100000000 --sed-observers 3 --steps 1 --warmup 0"
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $P/a -o a -- $B > $P/a.log 2>&1 </dev/null
timeout 600 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $P/b -o b -- $B > $P/b.log 2>&1 </dev/null
cd $R; python3 - <<'PY'
import csv, glob, collections, os
P=os.environ.get('GRAFT_REPO_ROOT','/root/repo')+'/gpurun_out/r6_pmc_sed3'
for sub in 'ab':
    c=collections.defaultdict(float); n=0
    for f in glob.glob(P+'/%s/**/*counter_collection.csv'%sub, recursive=True):
        for r in csv.DictReader(open(f)):
            if 'k_mono' in r['Kernel_Name'] and 'Lb0ELb1ELb0ELb0ELb1ELb0' in r['Kernel_Name'] or ('k_mono<false, true, false, false, true, false>' in r['Kernel_Name']):
                c[r['Counter_Name']]+=float(r['Counter_Value'])
    print(sub, dict(c))
    if 'SQ_WAVE_CYCLES' in c:
        print('  wait_frac', c['SQ_WAIT_ANY']/c['SQ_WAVE_CYCLES'], 'valu_active_per_wave', c['SQ_ACTIVE_INST_VALU']/c['SQ_WAVE_CYCLES'], 'lane util', c['SQ_THREAD_CYCLES_VALU']/(64*c['SQ_ACTIVE_INST_VALU']) if c['SQ_ACTIVE_INST_VALU'] else None, 'busy', c['SQ_BUSY_CYCLES'])
PY
