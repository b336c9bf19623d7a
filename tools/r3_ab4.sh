#!/bin/bash
# A/B of two builds of the library on the four-wave kernel (variant 5): bash tools/r3_ab4.sh <lib A> <lib B> [plants ...]
A=$1; B=$2; shift 2
for n in ${@:-8192 32768}; do
  for rep in 1 2; do
    for lib in $A $B; do
      NPB_LIB=$lib NPB_STEP_KERNEL=5 python3 bench.py --plants-per-gpu $n --steps 400 --warmup 50 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$n', '$lib', d['ms_per_step'], d['roofline']['frac'])"
    done
  done
done
