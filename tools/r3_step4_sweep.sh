#!/bin/bash
# the four-wave kernel (variant 5) against the kernel the launcher picks by size (variant 0), with and without maintenance
#   bash tools/r3_step4_sweep.sh gpurun_out/r3_step4
out=${1:-gpurun_out/r3_step4}; mkdir -p $out
for n in 8192 16384 24576 32768 40960 49152 65536; do
  for v in 0 5; do
    NPB_STEP_KERNEL=$v python3 bench.py --plants-per-gpu $n --steps 300 --warmup 50 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$n variant $v', d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('kernel'))" | tee -a $out/sweep.txt
  done
done
for n in 32768; do
  for v in 0 5; do
    NPB_STEP_KERNEL=$v python3 bench.py --plants-per-gpu $n --steps 300 --warmup 50 --no-cpu-baseline --maintenance 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$n maintenance variant $v', d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('kernel'))" | tee -a $out/sweep.txt
  done
done
