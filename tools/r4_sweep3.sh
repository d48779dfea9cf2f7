#!/bin/bash
# round-4 A/B no. 3: the 3D schedule's knobs (tuning build) and the size of the deposit log
out=$1; mkdir -p $(dirname $out); : > $out
V=$PWD/mcfost_amd/csrc/variants
run() {  # label, config, env...
  label=$1; cfg=$2; shift 2
  env "$@" python bench.py --config $cfg --steps 2 --warmup 1 --no-cpu-baseline --no-extra 2>>$out.err | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print('%-34s %-9s %.4g pk/s  kernel_ms %.1f  tail %.1f  chunks %s' % ('$label', '$cfg', d['value'], d['roofline']['kernel_ms'], (d.get('tail') or {}).get('tail_ms', 0), (d.get('binned_deposits') or {}).get('chunks')))
" >> $out
}
run "tune default" ref41_3d MCGPU_LIB=$V/lib_tune.so
for fi in 8 24 32; do run "fly_iters=$fi" ref41_3d MCGPU_LIB=$V/lib_tune.so MCGPU_FLY_ITERS=$fi; done
for fd in 16 48; do run "fly_idle=$fd" ref41_3d MCGPU_LIB=$V/lib_tune.so MCGPU_FLY_IDLE=$fd; done
for ks in 0 4; do run "k_short=$ks" ref41_3d MCGPU_LIB=$V/lib_tune.so MCGPU_K_SHORT=$ks; done
for ns in 2 4 5; do run "n_srv=$ns" ref41_3d MCGPU_LIB=$V/lib_tune.so MCGPU_N_SRV=$ns; done
cat $out
