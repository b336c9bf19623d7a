#!/bin/bash
# SQ counter passes over the step kernel (one rocprofv3 run per counter group; PMC only, no other tracing).
# usage (on the GPU box, from the repo root): bash tools/sq_passes.sh gpurun_out/sq
set -e
OUT=$(realpath -m "$1"); REPO=$(pwd)
mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
i=0
for group in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" \
  "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH" \
  "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT" \
  "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INSTS_VALU_INT64" \
  "SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SMEM SQ_THREAD_CYCLES_VALU SQ_IFETCH"
do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $group --output-format csv -d "$OUT/p$i" -- python3 "$REPO/tools/profile_traffic.py" > "$OUT/p$i.log" 2>&1
  echo "pass $i done"
done
cd "$REPO"; python3 tools/profile_traffic.py --summarize-sq "$OUT" | tee "$OUT/summary.txt"
