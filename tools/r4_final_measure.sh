#!/bin/bash
# Round-4 closing measurements of the final build (GPU box, repo root): the six bench lines of tools/round_measure.sh, then the profile passes
# (kernel-trace stats, PMC traffic, SQ counters) at 65 536 and 32 768 plants.   bash tools/r4_final_measure.sh gpurun_out/r4/final
OUT=$(realpath -m "$1"); mkdir -p "$OUT"
python3 bench.py > "$OUT/bench_64k.json" 2>"$OUT/bench_64k.err"
python3 bench.py --plants-per-gpu 32768 > "$OUT/bench_32k.json" 2>/dev/null
python3 bench.py --storage f32 --no-cpu-baseline > "$OUT/bench_f32_64k.json" 2>/dev/null
python3 bench.py --storage f32 --no-cpu-baseline --plants-per-gpu 32768 > "$OUT/bench_f32_32k.json" 2>/dev/null
python3 bench.py --maintenance --no-cpu-baseline > "$OUT/bench_maint_64k.json" 2>/dev/null
python3 bench.py --maintenance --no-cpu-baseline --plants-per-gpu 32768 > "$OUT/bench_maint_32k.json" 2>/dev/null
python3 bench.py --steps 20 --warmup 5 > "$OUT/bench_driver_like.json" 2>/dev/null
for f in "$OUT"/bench_*.json; do python3 -c "
import json,sys
d=json.loads([l for l in open('$f') if l.startswith('{')][0]); print('$(basename $f)', '%.4e' % d['value'], '%.5f ms' % d['ms_per_step'], 'frac %.3f' % d['roofline']['frac'], d['roofline']['kernel'][:26])"; done | tee "$OUT/lines.txt"
python3 tools/config4.py --plants 32768 --steps 24 --dt 5.0 > "$OUT/config4.jsonl" 2>/dev/null
bash tools/profile_round.sh "$OUT/prof64" > "$OUT/prof64.log" 2>&1
NPB_PROFILE_PLANTS=32768 bash tools/profile_round.sh "$OUT/prof32" > "$OUT/prof32.log" 2>&1
tail -3 "$OUT/prof64.log"
