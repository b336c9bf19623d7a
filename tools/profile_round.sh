#!/bin/bash
# One profiling pass of the current build (on the GPU box, from the repo root): kernel-trace stats, the two PMC
# traffic passes, the SQ counter passes.  usage: bash tools/profile_round.sh gpurun_out/prof_TAG
# Every rocprofv3 run has the program itself after "--" and uses --pmc only together with --kernel-trace.
set -e
OUT=$(realpath -m "$1"); REPO=$(pwd); mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$REPO/bench.py" --steps 100 --warmup 5 --no-cpu-baseline --plants-per-gpu ${NPB_PROFILE_PLANTS:-65536} > "$OUT/stats.log" 2>&1
echo "stats done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 "$REPO/tools/profile_traffic.py" > "$OUT/fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 "$REPO/tools/profile_traffic.py" > "$OUT/write.log" 2>&1
echo "traffic passes done"
cd "$REPO"
python3 tools/profile_traffic.py --summarize "$OUT" > "$OUT/traffic.json"
cat "$OUT/traffic.json"
bash tools/sq_passes.sh "$OUT/sq" | tail -40
find "$OUT/stats" -name "*kernel_stats.csv" | head -1 | xargs head -8
