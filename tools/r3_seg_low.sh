#!/bin/bash
# between 32 768 and 49 152 plants: the two-wave kernel (one-block arena) against the four-wave kernel on segmented arenas
for n in 36864 40960 45056; do
  NPB_ARENA_SEGMENT=0 NPB_STEP_KERNEL=2 python3 bench.py --plants-per-gpu $n --steps 300 --warmup 50 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$n one block variant 2', d['ms_per_step'], d['roofline']['frac'])"
  for seg in 4096 8192 16384; do
  NPB_ARENA_SEGMENT=$seg NPB_STEP_KERNEL=5 python3 bench.py --plants-per-gpu $n --steps 300 --warmup 50 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$n segment $seg variant 5', d['ms_per_step'], d['roofline']['frac'])"
  done
done
