#!/bin/bash
# Round 4: the chain wave of the four-wave kernel stores the stages' efficiency degradation / deposit thickness itself (NPD4_CHAIN_STORES_DEG,
# npd_step4.h) against the post-pass waves fetching those 28 columns a second time (ablate/libnpb_deg0.so).  Time by bench.py, alternating,
# then FETCH_SIZE / WRITE_SIZE of each.   usage (GPU box, repo root): bash tools/r4_chain_deg.sh gpurun_out/r4/chain_deg
set -e
OUT=$(realpath -m "$1"); REPO=$(pwd); mkdir -p "$OUT"
LIBS="nuclear_sim_amd/libnpb.so nuclear_sim_amd/ablate/libnpb_deg0.so"
bash tools/r4_ab_libs.sh "$OUT" "65536 32768" $LIBS
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
