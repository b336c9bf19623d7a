#!/bin/bash
out=${1:-gpurun_out/r3_chunks}; mkdir -p $out
for n in 65536 49152 81920; do
  for run in 1 2 8 64 512; do
    NPB_STEP4_RUN=$run NPB_STEP_KERNEL=6 python3 bench.py --plants-per-gpu $n --steps 300 --warmup 50 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$n run $run', d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('kernel'))" | tee -a $out/run_sweep.txt
  done
done
