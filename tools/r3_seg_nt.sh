#!/bin/bash
# does the segmented arena also help the kernels of the other size ranges?  (variant 4 = streaming one-wave build, 2 = two-wave)
for n in 131072 262144; do for seg in 0 8192 16384 32768; do
NPB_ARENA_SEGMENT=$seg NPB_STEP_KERNEL=4 python3 bench.py --plants-per-gpu $n --steps 200 --warmup 30 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$n segment $seg variant 4', d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel'])"
done; done
for seg in 0 8192 16384; do
NPB_ARENA_SEGMENT=$seg NPB_STEP_KERNEL=2 python3 bench.py --plants-per-gpu 40960 --steps 300 --warmup 50 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('40960 segment $seg variant 2', d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel'])"
done
