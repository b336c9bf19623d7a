for seg in 0 8192 16384 32768; do for rep in 1 2; do
NPB_ARENA_SEGMENT=$seg python3 bench.py --plants-per-gpu 65536 --storage f32 --steps 400 --warmup 50 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('f32 65536 segment $seg', d['ms_per_step'], d['value'], d['roofline']['kernel'])"
done; done
