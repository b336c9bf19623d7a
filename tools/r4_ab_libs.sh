#!/bin/bash
# A/B of several builds of libnpb.so by bench.py, alternating:  bash tools/r4_ab_libs.sh OUTDIR "32768 65536" lib1.so lib2.so ...
OUT=$(realpath -m "$1"); REPO=$(pwd); mkdir -p "$OUT"; SIZES=$2; shift 2
for n in $SIZES; do
  for rep in 1 2 3; do
    for lib in "$@"; do
      NPB_LIB=$REPO/$lib python3 bench.py --plants-per-gpu $n --steps 400 --warmup 50 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$n', '$lib', d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel'][:24])" | tee -a "$OUT/times.txt"
    done
  done
done
