#!/bin/bash
# Round-4 measurement pass on the GPU box (repo root): phase stamps of the four-wave kernel, the bench lines, config 2 / 4 tools.
# usage: bash tools/r4_round.sh gpurun_out/r4/round
OUT=$(realpath -m "$1"); mkdir -p "$OUT"
python3 tools/phase_stamps4.py 32768 10 > "$OUT/step4_phase_stamps_32k.txt" 2>&1
python3 tools/phase_stamps4.py 65536 10 > "$OUT/step4_phase_stamps_64k.txt" 2>&1
for n in 32768 40960 49152 65536; do
  python3 bench.py --plants-per-gpu $n --steps 400 --warmup 50 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$n', d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel'][:28])" | tee -a "$OUT/sweep.txt"
done
python3 tools/config2.py > "$OUT/config2.txt" 2>&1
python3 bench.py --steps 20 --warmup 5 > "$OUT/bench_driver_like.json" 2>"$OUT/bench_driver_like.err"
NPB_BENCH_DEVICE=0 NPB_BENCH_BACKEND=gloo NPB_PLACEMENT_PROBE=0 python3 bench.py --gpus 2 --steps 20 --warmup 5 --plants-per-gpu 8192 --no-cpu-baseline > "$OUT/bench_self_launch_2ranks.json" 2>"$OUT/bench_self_launch_2ranks.err"
tail -c 600 "$OUT/bench_self_launch_2ranks.json"
