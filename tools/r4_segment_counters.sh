#!/bin/bash
# Round 4: WHY does the segmented arena help?  The same step kernel (four-wave, 65 536 plants) on a segmented arena (default) and on
# one block (NPB_ARENA_SEGMENT=0), one rocprofv3 --pmc pass per counter group (PMC only with --kernel-trace): address translation
# (UTCL1 / UTCL2), L2 hit / miss and tag stalls, fabric read requests by size and by DRAM / Infinity-Cache target.
# usage (GPU box, repo root): bash tools/r4_segment_counters.sh gpurun_out/r4/segctr
OUT=$(realpath -m "$1"); REPO=$(pwd); mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
rocprofv3 --list-avail > "$OUT/avail.txt" 2>&1 || rocprofv3 -L > "$OUT/avail.txt" 2>&1
grep -o "\b\(TCP_UTCL1[A-Z0-9_]*\|TCP_UTCL2[A-Z0-9_]*\|UTCL2[A-Z0-9_]*\|TCC_EA0_RD[A-Z0-9_]*\|TCC_EA0_WR[A-Z0-9_]*\|TCC_TAG_STALL[A-Z0-9_]*\|TCC_HIT[A-Z0-9_]*\|TCC_MISS[A-Z0-9_]*\|TCC_BUBBLE[A-Z0-9_]*\|TCC_MALL[A-Z0-9_]*\|TCP_TCC_[A-Z0-9_]*\|TCP_PENDING_STALL[A-Z0-9_]*\|TCP_TA_TCP_STATE_READ[A-Z0-9_]*\|TCC_REQ[A-Z0-9_]*\|TCC_READ[A-Z0-9_]*\)\b" "$OUT/avail.txt" | sort -u > "$OUT/candidates.txt"
wc -l "$OUT/candidates.txt"
i=0
for group in \
  "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_PERMISSION_MISS_sum" \
  "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
  "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RD_UNCACHED_32B_sum" \
  "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_WRREQ_STALL_sum" \
  "TCC_TAG_STALL_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_BUBBLE_sum" \
  "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum"
do
  i=$((i+1))
  for layout in seg one; do
    if [ $layout = one ]; then export NPB_ARENA_SEGMENT=0; else unset NPB_ARENA_SEGMENT; fi
    rocprofv3 --kernel-trace --pmc $group --output-format csv -d "$OUT/$layout/p$i" -- python3 "$REPO/tools/profile_traffic.py" > "$OUT/$layout.p$i.log" 2>&1 || echo "pass $i ($layout) failed: see $layout.p$i.log"
  done
  echo "pass $i done"
done
unset NPB_ARENA_SEGMENT
cd "$REPO"
for layout in seg one; do echo "== $layout"; python3 tools/profile_traffic.py --summarize-sq "$OUT/$layout"; done | tee "$OUT/summary.txt"
