#!/bin/bash
# Rehearsal of bench.py's multi-process path on a ONE-GPU box: N ranks (at most 6) share device 0 and talk over gloo, so the
# rank-dependent inputs, the barriers, the max-over-ranks timing and the episode-end gather / reduce all run.  The numbers
# mean nothing (the ranks share one GPU); the driver's real multi-GPU run uses RCCL, one rank per GPU.
#   bash tools/bench_rehearsal.sh 2
N=${1:-2}
NPB_BENCH_DEVICE=0 NPB_BENCH_BACKEND=gloo NPB_PLACEMENT_PROBE=0 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $N \
  --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus $N --steps 20 --warmup 5 --plants-per-gpu 8192
# the same for BASELINE config 4's run (sharded by global seed, histogram of executions summed over the ranks)
NPB_BENCH_DEVICE=0 NPB_BENCH_BACKEND=gloo python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $N \
  --master-addr 127.0.0.1 --master-port 29512 tools/config4.py --plants 4096 --steps 30 --dt 5.0
