#!/bin/bash
# the segmented arena (handles of 32 769 .. 98 304 plants: segments of 32 768) against the one-block arena, per step kernel
#   bash tools/r3_segment_sweep.sh OUT
out=${1:-gpurun_out/r3_segments}; mkdir -p $out
for n in 40960 49152 57344 65536 81920 98304; do
  for segm in 1 0; do
    for v in 6 1 2; do
      if [ $v = 2 ] && [ $n -gt 65536 ]; then continue; fi
      NPB_ARENA_SEGMENT=$segm NPB_STEP_KERNEL=$v python3 bench.py --plants-per-gpu $n --steps 300 --warmup 50 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$n segmented $segm variant $v', d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('kernel'))" | tee -a $out/segment_sweep.txt
    done
  done
done
