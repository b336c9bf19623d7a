#!/bin/bash
# rocprofv3 kernel statistics of bench.py --maintenance at 32 768 and 65 536 plants: step kernel and rule kernel durations
set -e
REPO=$(pwd); OUT=$REPO/gpurun_out/prof_maint; rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for n in 32768 65536; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s$n -- python3 $REPO/bench.py --steps 100 --warmup 5 --no-cpu-baseline --maintenance --plants-per-gpu $n > $OUT/s$n.log 2>&1
  f=$(find $OUT/s$n -name "*kernel_stats.csv" | xargs ls -S | head -1)
  cp $f $OUT/kernel_stats_$n.csv
  echo "== $n plants"; python3 - $f <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if r["Name"].startswith("npb_"):
        print("%-28s calls %6s  avg %9.0f ns  min %8s  max %9s" % (r["Name"].split("(")[0], r["Calls"], float(r["AverageNs"]), r["MinNs"], r["MaxNs"]))
PY
  find $OUT/s$n -name "*kernel_trace.csv" -delete
done
