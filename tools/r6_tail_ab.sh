#!/bin/bash
# round 6: the tail of a launch -- k_tail alone (tail_where 1) against k_tail + the library's host threads (2)
mkdir -p gpurun_out/r6_tail
nproc > gpurun_out/r6_tail/nproc.txt; lscpu | grep -i "model name\|^CPU(s)\|Thread" >> gpurun_out/r6_tail/nproc.txt
timeout 900 python -m pytest tests/test_binned_deposits.py -x -q -m gpu -k tail_kernel 2>&1 | tail -15 > gpurun_out/r6_tail/pytest_tail.log
for cfg in ref41 ref41_3d; do
  for w in 1 2; do
    timeout 600 python bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline --tail-where $w > gpurun_out/r6_tail/${cfg}_where$w.json 2> gpurun_out/r6_tail/${cfg}_where$w.err
  done
done
for w in 1 2; do
  timeout 600 python bench.py --config ref41_mrw --packets 1e7 --steps 2 --warmup 1 --no-cpu-baseline --tail-where $w > gpurun_out/r6_tail/mrw_where$w.json 2> gpurun_out/r6_tail/mrw_where$w.err
done
timeout 600 python bench.py --config ref41_mrw --packets 1e8 --steps 1 --warmup 1 --no-cpu-baseline --tail-where 2 > gpurun_out/r6_tail/mrw1e8_where2.json 2> gpurun_out/r6_tail/mrw1e8_where2.err
for hp in 64 128 512 1024; do
  timeout 600 python bench.py --config ref41 --steps 3 --warmup 1 --no-cpu-baseline --tail-where 2 --tail-host-packets $hp > gpurun_out/r6_tail/ref41_hp$hp.json 2>&1
done
