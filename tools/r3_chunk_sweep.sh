#!/bin/bash
# a batch too large for one launch of the four-wave kernel: several launches of it (variant 6) against the single-launch kernels
#   bash tools/r3_chunk_sweep.sh OUT
out=${1:-gpurun_out/r3_chunks}; mkdir -p $out
for n in 40960 49152 57344 65536 73728 81920 90112 98304 131072; do
  for v in 6 2 1 4; do
    if [ $v = 2 ] && [ $n -gt 65536 ]; then continue; fi
    NPB_STEP_KERNEL=$v python3 bench.py --plants-per-gpu $n --steps 300 --warmup 50 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$n variant $v', d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('kernel'))" | tee -a $out/chunk_sweep.txt
  done
done
