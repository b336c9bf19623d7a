#!/usr/bin/env python3
"""bench.py -- plant-env-steps/s of the fused HIP step kernel on MI355X.

A "step" is one pass of the hot path (NuclearPlantSimulator.step for every plant) over one
batch of synthetic per-step inputs already resident in HBM.  Workload at N=1: BASELINE config 3
(65 536 plants, full secondary, ConstantHeatSource with 0.1 % noise, load-following setpoints,
fp64).  Multi-GPU: every rank owns the same number of plants (weak scaling), no collective in the
data path; one RCCL all-gather of the observation block after the timed region (episode end).

Launch:  python bench.py --gpus 1 --steps K --warmup W
         python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PLANTS_PER_GPU = 65536
HBM_PEAK_GBS = 8000.0  # MI355X spec HBM3E bandwidth (guides/MI355X_MICROARCH.md)


def cpu_baseline(seconds_budget=15.0):
    """The CPU oracle (plain-C restatement, 'port') timed on this box's host cores on a bounded
    sample of the same workload.  Reported beside the GPU number; it is a baseline, not a target."""
    from oracle import npo
    n = 32768
    cores = os.cpu_count() or 1
    os.environ.setdefault("OMP_NUM_THREADS", str(cores))
    P = npo.Params(); P.hs_noise_enabled = 1
    ora = npo.OraclePlants(n, P)
    rng = np.random.default_rng(1)
    sp = 90.0 + 10.0 * np.sin(np.arange(n) / 16.0)
    ora.step(setpoint=sp, noise_z=rng.standard_normal(n))  # warm-up
    steps = 0
    t0 = time.perf_counter()
    while True:
        ora.step(setpoint=sp, noise_z=rng.standard_normal(n))
        steps += 1
        dt = time.perf_counter() - t0
        if dt > seconds_budget or steps >= 2000:
            break
    return {"value": n * steps / dt, "unit": "plant-env-steps/s", "cores": cores, "kind": "port",
            "sample": "%d plants x %d steps of the same C3 workload through oracle/libnpo.so (OpenMP over plants, %.1f s)" % (n, steps, dt)}


def measured_traffic(n):
    """HBM bytes per npb_step_kernel launch from the committed rocprofv3 PMC passes (profiles/README.md);
    only meaningful for the plant count it was measured at."""
    best = None
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json"))):
        try:
            d = json.load(open(f))
            if d.get("plants") == n and isinstance(d.get("step_hbm_bytes_per_launch"), (int, float)):
                best = float(d["step_hbm_bytes_per_launch"])
        except Exception:
            pass
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--plants-per-gpu", type=int, default=PLANTS_PER_GPU)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--maintenance", action="store_true",
                    help="also run the automatic oil_top_off maintenance kernel after every step (not the headline workload)")
    ap.add_argument("--storage", choices=["f64", "f32"], default="f64",
                    help="element type of the carried state in HBM (f32 = BASELINE config 5; the headline is f64)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    distributed = world > 1
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)

    from nuclear_sim_amd.env import BatchedPlantEnv
    from nuclear_sim_amd.sharding import gather_observations, reduce_counters

    n = args.plants_per_gpu
    n_global = n * world
    lo = rank * n
    K, W = args.steps, args.warmup

    env = BatchedPlantEnv(n, dt=1.0, heat_source="constant", noise_enabled=True, noise_std_percent=0.1, device=local_rank,
                          maintenance=args.maintenance, storage=args.storage)
    # synthetic inputs, resident in HBM before the timed region: per-plant load-following setpoint
    # trace (90 % + 10 % sin, period 600 + 60*(i mod 16) steps) and N(0,1) noise samples
    gid = torch.arange(lo, lo + n, device=dev, dtype=torch.float64)
    period = 600.0 + 60.0 * (gid % 16)
    gen = torch.Generator(device=dev); gen.manual_seed(1234 + rank)
    total = K + W
    tt = torch.arange(total, device=dev, dtype=torch.float64)[:, None]
    setpoints = (90.0 + 10.0 * torch.sin(2.0 * np.pi * tt / period[None, :])).contiguous()
    noise = torch.randn((total, n), device=dev, dtype=torch.float64, generator=gen)

    def one_step(t):
        return env.step(power_setpoint=setpoints[t], noise_z=noise[t])

    for t in range(W):
        one_step(t)
    torch.cuda.synchronize(dev)
    if distributed:
        dist.barrier()
    torch.cuda.synchronize(dev)

    # timed region: exactly K steps between barrier + synchronize on both sides, nothing else in it
    t0 = time.perf_counter()
    for k in range(K):
        obs, rew, done, info = one_step(W + k)
    torch.cuda.synchronize(dev)
    if distributed:
        dist.barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0

    # roofline leg: the step kernel's own duration, HIP events on the stream it is launched on (torch's current
    # stream), one pair per launch, over a replay of the same steps (outside the timed region so that event
    # bookkeeping does not sit between the timed launches)
    KE = min(K, 100)
    starts = [torch.cuda.Event(enable_timing=True) for _ in range(KE)]
    ends = [torch.cuda.Event(enable_timing=True) for _ in range(KE)]
    for k in range(KE):
        starts[k].record()
        one_step(W + k)
        ends[k].record()
    torch.cuda.synchronize(dev)
    kernel_ms = float(np.mean([s.elapsed_time(e) for s, e in zip(starts, ends)]))
    el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    km = torch.tensor([kernel_ms], dtype=torch.float64, device=dev)
    if distributed:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_reduce(km, op=dist.ReduceOp.MAX)
        # episode end: the only collectives of the path
        full_obs = gather_observations(obs, n_global)
        flags = info["trip_flags"]
        counters = torch.stack([(flags & 1).ne(0).sum(), (flags & 8).ne(0).sum(), (flags & 0xF00).ne(0).sum(),
                                torch.zeros((), device=dev, dtype=torch.int64)]).to(torch.int64)
        counters = reduce_counters(counters)
        assert full_obs.shape == (n_global, 22)
    elapsed = float(el.item()); kernel_ms = float(km.item())

    if rank == 0:
        bytes_per_plant = env.handle_step_bytes_per_plant()
        achieved = bytes_per_plant * n / (kernel_ms * 1e-3) / 1e9
        out = {
            "metric": "plant-env-steps/s", "value": n_global * K / elapsed, "unit": "plant-env-steps/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": elapsed / K * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",   # arithmetic is fp64 in both storage modes
            "config": {"workload": "BASELINE config 3: %d plants per GPU (%d total), full secondary "
                                   "(primary + feedwater + 3 SG + turbine + condenser), ConstantHeatSource 0.1%% noise, "
                                   "load-following setpoints, dt=1.0, obs+reward+done+trip_flags+info written every step" % (n, n_global),
                       "plants_per_gpu": n, "global_plants": n_global, "parallelism": "plants sharded contiguously, no data-path collective",
                       "state_bytes_per_plant": BatchedPlantEnv.state_bytes_per_plant(),
                       "algorithmic_bytes_per_plant_step": bytes_per_plant, "maintenance_kernel": bool(args.maintenance),
                       "state_storage": args.storage},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": measured_traffic(n) if args.storage == "f64" else None,
                         "algorithmic_bytes_per_launch": bytes_per_plant * n,
                         "kernel": "npb_step_kernel", "kernel_ms": kernel_ms},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
