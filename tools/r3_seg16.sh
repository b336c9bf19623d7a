for n in 49152 57344 73728 81920 98304 106496; do for seg in 16384 8192 32768; do
NPB_ARENA_SEGMENT=$seg NPB_STEP_KERNEL=5 python3 bench.py --plants-per-gpu $n --steps 300 --warmup 50 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$n segment $seg variant 5', d['ms_per_step'], d['roofline']['frac'])"
done; done
for n in 49152 106496; do NPB_ARENA_SEGMENT=0 NPB_STEP_KERNEL=0 python3 bench.py --plants-per-gpu $n --steps 300 --warmup 50 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$n one block, by size', d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel'])"; done
