#!/bin/bash
# Round 4: how the narrow columns that hold nothing but OUTPUT members are stored (NPD_OUT_STORE, npb_kernels.hip):
# 0 = compared with their old bits like every narrow column (round 3: the only reason they are fetched), 1 = plain store without
# compare / load, 2 = the same with the non-temporal bit.  Time by bench.py (four-wave kernel), then FETCH_SIZE / WRITE_SIZE of each.
# usage (GPU box, repo root): bash tools/r4_out_store.sh gpurun_out/r4/out_store
set -e
OUT=$(realpath -m "$1"); REPO=$(pwd); mkdir -p "$OUT"
LIBS="nuclear_sim_amd/libnpb.so nuclear_sim_amd/ablate/libnpb_out1.so nuclear_sim_amd/ablate/libnpb_out2.so"
for n in 65536 32768; do
  for rep in 1 2; do
    for lib in $LIBS; do
      NPB_LIB=$REPO/$lib python3 bench.py --plants-per-gpu $n --steps 400 --warmup 50 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$n', '$lib', d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('kernel','')[:24])" | tee -a "$OUT/times.txt"
    done
  done
done
cd /tmp; export TMPDIR=/tmp
for lib in $LIBS; do
  tag=$(basename $lib .so)
  export NPB_LIB=$REPO/$lib
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/$tag/fetch" -- python3 "$REPO/tools/profile_traffic.py" > "$OUT/$tag.fetch.log" 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/$tag/write" -- python3 "$REPO/tools/profile_traffic.py" > "$OUT/$tag.write.log" 2>&1
  (cd "$REPO" && python3 tools/profile_traffic.py --summarize "$OUT/$tag" > "$OUT/$tag.traffic.json")
  echo "$tag traffic done"
done
cd "$REPO"
