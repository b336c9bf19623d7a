#!/usr/bin/env python3
"""BASELINE config 4 as a run: N randomised oil_top_off action-test plants (one per seed, the
data-gen runner's settings: ConstantHeatSource 0.1 % noise seeded 42, automatic maintenance on), sharded by global
seed over the ranks, 120 steps; at the end one all-gather of the observations and one all-reduce of the counters.
Prints one JSON line: set-up time (vectorised initial conditions, nuclear_sim_amd/scenarios.py), plant-env-steps/s of
the step loop, the histogram of per-plant oil_top_off executions and the counter totals.

  python3 tools/config4.py                       # 32 768 plants on one GPU (the per-GPU share of the 262 144 of C4)
  python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 tools/config4.py --plants 262144
Counts against the CPU restatement on sampled seeds are a test (tests/test_gpu_parity.py, test_config4_*)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--plants", type=int, default=32768, help="global number of plants (= seeds 0 .. plants-1)")
    ap.add_argument("--steps", type=int, default=120)
    ap.add_argument("--dt", type=float, default=1.0, help="minutes per step (the runner's unit)")
    args = ap.parse_args()
    import torch
    import torch.distributed as dist
    from nuclear_sim_amd.env import BatchedPlantEnv
    from nuclear_sim_amd.sharding import gather_observations, reduce_counters, shard_range, event_histogram

    rank = int(os.environ.get("RANK", "0")); local_rank = int(os.environ.get("LOCAL_RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    # rehearsal on a one-GPU box (as bench.py's, tools/bench_rehearsal.sh): every rank on device NPB_BENCH_DEVICE, collectives over gloo
    if os.environ.get("NPB_BENCH_DEVICE") is not None:
        local_rank = int(os.environ["NPB_BENCH_DEVICE"])
    backend = os.environ.get("NPB_BENCH_BACKEND", "nccl")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    dev = torch.device("cuda", local_rank)
    lo, hi = shard_range(args.plants, rank, world)
    n = hi - lo
    t0 = time.perf_counter()
    env = BatchedPlantEnv.action_test("oil_top_off", list(range(lo, hi)), dt=args.dt, device=local_rank)
    torch.cuda.synchronize(dev)
    setup_s = time.perf_counter() - t0
    # the runner ramps the heat source towards its profile; here a fixed 90 % target, the noise from the per-plant stream
    target = torch.full((n,), 90.0, dtype=torch.float64, device=dev)
    for _ in range(3):
        env.step(power_setpoint=target)          # warm-up (part of the episode; the count below covers all steps)
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps - 3):
        obs, rew, done, info = env.step(power_setpoint=target)
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    loop_s = time.perf_counter() - t0
    events = env.get_field("maint.maintenance_actions_performed").to(torch.int64)
    created = env.get_field("maint.work_orders_created").to(torch.int64)
    gloo = world > 1 and backend != "nccl"      # gloo moves host tensors
    hist = event_histogram(events.cpu() if gloo else events)
    flags = info["trip_flags"]
    counters = torch.stack([(flags & 1).ne(0).sum(), (flags & 0xF00).ne(0).sum(), events.sum(), created.sum()]).to(torch.int64)
    if world > 1:
        full_obs = gather_observations(obs.cpu() if gloo else obs, args.plants)
        counters = reduce_counters(counters.cpu() if gloo else counters)
        el = torch.tensor([loop_s], dtype=torch.float64, device="cpu" if gloo else dev); dist.all_reduce(el, op=dist.ReduceOp.MAX); loop_s = float(el.item())
        assert full_obs.shape == (args.plants, 22)
    if rank == 0:
        h = hist.cpu().numpy()
        print(json.dumps({
            "config": "BASELINE config 4: %d randomised oil_top_off plants (seeds 0..%d), %d steps of %g min, maintenance on"
                      % (args.plants, args.plants - 1, args.steps, args.dt),
            "n_gpus": world, "plants_per_gpu": n, "setup_seconds_rank0": setup_s,
            "plant_env_steps_per_s": args.plants * (args.steps - 3) / loop_s, "ms_per_step": loop_s / (args.steps - 3) * 1e3,
            "oil_top_off_executions_histogram": {str(k): int(v) for k, v in enumerate(h) if v},
            "scrammed_plants": int(counters[0]), "plants_with_pump_trip": int(counters[1]),
            "oil_top_off_executions": int(counters[2]), "work_orders_created": int(counters[3])}), flush=True)
    if world > 1:
        dist.barrier(); dist.destroy_process_group()


if __name__ == "__main__":
    main()
