set -e
REPO=$(pwd); OUT=$REPO/gpurun_out/prof_maint; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s32 -- python3 $REPO/bench.py --steps 100 --warmup 5 --no-cpu-baseline --maintenance --plants-per-gpu 32768 > $OUT/s32.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s64 -- python3 $REPO/bench.py --steps 100 --warmup 5 --no-cpu-baseline --maintenance --plants-per-gpu 65536 > $OUT/s64.log 2>&1
for d in s32 s64; do find $OUT/$d -name "*kernel_stats.csv" | head -1 | xargs head -6; done
