#!/bin/bash
# round-4 A/B: tools/r4_sweep.sh <out.log>; each line: label, packets/s, kernel ms (bench.py --no-extra, 1e8 packets)
out=$1; mkdir -p $(dirname $out); : > $out
run() {  # label, config, env...
  label=$1; cfg=$2; shift 2
  env "$@" python bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline --no-extra 2>>$out.err | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print('%-40s %-9s %.4g pk/s  kernel_ms %.1f  cross/pk %.1f' % ('$label', '$cfg', d['value'], d['roofline']['kernel_ms'], d['config']['crossings_per_packet']))
" >> $out
}
V=$PWD/mcfost_amd/csrc/variants
run default pascucci
run default ref41
for fi in 16 24 32 48 64; do run "tune fly_iters=$fi" pascucci MCGPU_LIB=$V/lib_tune.so MCGPU_FLY_ITERS=$fi; done
for fd in 8 16 48; do run "tune fly_iters=32 fly_idle=$fd" pascucci MCGPU_LIB=$V/lib_tune.so MCGPU_FLY_ITERS=32 MCGPU_FLY_IDLE=$fd; done
for fi in 24 32; do run "tune fly_iters=$fi" ref41 MCGPU_LIB=$V/lib_tune.so MCGPU_FLY_ITERS=$fi; done
cat $out
