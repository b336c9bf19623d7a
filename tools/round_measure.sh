#!/bin/bash
# The measurements a round's README / DESIGN numbers come from (on the GPU box, from the repo root):
#   bash tools/round_measure.sh gpurun_out/TAG
# bench lines (default = 65 536 plants, 32 768 plants, fp32 storage, with maintenance), then the profiling passes
# (kernel-trace stats, PMC traffic, SQ counters) at 65 536 and 32 768 plants, then the phase stamps of the three kernels.
set -e
OUT=$(realpath -m "$1"); mkdir -p "$OUT"
python3 bench.py > "$OUT/bench_64k.json" 2> "$OUT/bench_64k.err"
python3 bench.py --plants-per-gpu 32768 --no-cpu-baseline > "$OUT/bench_32k.json" 2>> "$OUT/bench_64k.err"
python3 bench.py --storage f32 --no-cpu-baseline > "$OUT/bench_f32_64k.json" 2>> "$OUT/bench_64k.err"
python3 bench.py --storage f32 --plants-per-gpu 32768 --no-cpu-baseline > "$OUT/bench_f32_32k.json" 2>> "$OUT/bench_64k.err"
python3 bench.py --maintenance --no-cpu-baseline > "$OUT/bench_maint_64k.json" 2>> "$OUT/bench_64k.err"
python3 bench.py --maintenance --plants-per-gpu 32768 --no-cpu-baseline > "$OUT/bench_maint_32k.json" 2>> "$OUT/bench_64k.err"
echo "bench lines done"
bash tools/profile_round.sh "$OUT/prof_64k" > "$OUT/prof_64k.log" 2>&1
NPB_PROFILE_PLANTS=32768 bash tools/profile_round.sh "$OUT/prof_32k" > "$OUT/prof_32k.log" 2>&1
# the streaming build of the one-wave kernel (what npb_step takes at 131 072 plants): kernel-trace stats only
( cd /tmp; export TMPDIR=/tmp; rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_131k" -- python3 "$OLDPWD/bench.py" --steps 100 --warmup 5 --no-cpu-baseline --plants-per-gpu 131072 > "$OUT/prof_131k.log" 2>&1 )
echo "profiles done"
NPB_STEP_KERNEL=1 python3 tools/phase_stamps.py 65536 10 > "$OUT/stamps1_64k.txt" 2>&1
NPB_STEP_KERNEL=2 python3 tools/phase_stamps2.py 32768 5 > "$OUT/stamps2_32k.txt" 2>&1
python3 tools/phase_stamps4.py 32768 10 > "$OUT/stamps4_32k.txt" 2>&1
python3 tools/phase_stamps4.py 8192 10 > "$OUT/stamps4_8k.txt" 2>&1
echo "stamps done"
