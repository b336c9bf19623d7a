#!/bin/bash
# segment size of the arena (NPB_ARENA_SEGMENT=<plants>, 0 = one block) per step kernel: bash tools/r3_segsize_sweep.sh
for n in 65536 32768 131072; do
  for seg in 0 2048 8192 16384 32768 65536; do
    if [ $seg -ge $n ] && [ $seg != 0 ]; then continue; fi
    for v in 5 1; do
      if [ $n = 32768 ] && [ $v = 1 ]; then continue; fi
      NPB_ARENA_SEGMENT=$seg NPB_STEP_KERNEL=$v python3 bench.py --plants-per-gpu $n --steps 300 --warmup 50 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$n segment $seg variant $v', d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel'])"
    done
  done
done
