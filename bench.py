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
PRECONDITION_MS = 40.0  # a scratch handle is stepped this long before the warm-up steps, see main()


REFERENCE_PYTHON_STEPS_PER_S_PER_CORE = 147.7  # the reference's own step(), survey container, 1 core


def allowed_cores():
    """Cores this process may actually use: its affinity mask, bounded by the cgroup CPU quota (cgroup v2 cpu.max or
    v1 cfs quota).  os.cpu_count() is the machine, not the allowance."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except Exception:
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except Exception:
            pass
    if quota is not None:
        n = max(1, min(n, int(quota + 0.5)))
    return n


def _time_oracle(npo, n, threads, seconds_budget):
    P = npo.Params(); P.hs_noise_enabled = 1
    ora = npo.OraclePlants(n, P)
    used = int(ora.L.npo_set_threads(int(threads)))
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
    return n * steps / dt, used, steps, dt


def cpu_baseline():
    """The CPU oracle (plain-C restatement, 'port') timed on this box's host cores on a bounded sample of the same
    workload: one thread, then every core this process is allowed (affinity / cgroup quota; OpenMP over plants).
    Reported beside the GPU number; it is a baseline, not a target.  The reference's own Python step() cannot
    travel to this box; its survey-container figure is quoted beside it."""
    from oracle import npo
    cores = allowed_cores()
    v1, _, s1, d1 = _time_oracle(npo, 2048, 1, 6.0)
    n = max(4096, 512 * cores)
    vN, used, sN, dN = _time_oracle(npo, n, cores, 12.0)
    return {"value": vN, "unit": "plant-env-steps/s", "cores": used, "kind": "port",
            "value_1_thread": v1, "machine_cpus": os.cpu_count(),
            "reference_python": {"value_per_core": REFERENCE_PYTHON_STEPS_PER_S_PER_CORE, "unit": "plant-env-steps/s/core",
                                 "where": "reference NuclearPlantSimulator.step(), state management off, survey container "
                                          "(BASELINE.md section 2) -- a different box; the reference never travels to the GPU box"},
            "sample": "the same C3 workload through oracle/libnpo.so: %d plants x %d steps on %d thread(s) (%.1f s), "
                      "2048 plants x %d steps on 1 thread (%.1f s)" % (n, sN, used, dN, s1, d1)}


def measured_traffic(n):
    """HBM bytes per npb_step_kernel launch from the COMMITTED rocprofv3 PMC passes (profiles/README.md) -- a constant
    read from a file of an earlier profiling run of the same workload, not measured by this run; only meaningful
    for the plant count it was measured at.  Returns (bytes, source file) or (None, None)."""
    best = (None, None)
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json"))):
        try:
            d = json.load(open(f))
            if d.get("plants") == n and isinstance(d.get("step_hbm_bytes_per_launch"), (int, float)):
                best = (float(d["step_hbm_bytes_per_launch"]), os.path.relpath(f, ROOT))
        except Exception:
            pass
    return best


def measured_traffic_split(n):
    """(read bytes, write bytes) per launch from the same committed PMC passes (calibrated FETCH_SIZE, WRITE_SIZE), or None"""
    src = measured_traffic(n)[1]
    if not src:
        return None
    try:
        d = json.load(open(os.path.join(ROOT, src)))
        return float(d["FETCH_SIZE"]["step_bytes_calibrated"]), float(d["WRITE_SIZE"]["step_bytes_calibrated"])
    except Exception:
        return None


def algorithmic_read_write_bytes(env, bytes_per_plant):
    """the algorithmic bytes of one plant-step by direction: a carried member is read and written once each (8 B + 8 B; 4 + 4 for an
    int32 member), an output member only written (4 B), the per-step inputs only read, obs / reward / done / flags / info only written.
    The handle reports the sum (npb_handle_step_bytes_per_plant); the write-only part is the schema's."""
    from nuclear_sim_amd.schema import SCHEMA
    info_dim = 17
    write_only = 4 * SCHEMA.n_outputs_step() + 22 * 8 + 8 + 1 + 4 + info_dim * 8
    read_only = 16                                   # power setpoint + noise (the other inputs are passed as NULL)
    both = bytes_per_plant - write_only - read_only  # state read + written
    return both // 2 + read_only, both // 2 + write_only


def c3_noise(lo, n, total):
    """BASELINE config 3's heat-source noise exactly as specified: plant i draws from
    np.random.RandomState(42 + i) -- the reference's own generator (constant_heat_source.py:58-62,178), seeded by
    GLOBAL plant id, so the samples a plant sees do not depend on how many GPUs share the batch.  Drawn on the host
    before the timed region and uploaded as one [total, n] block (inputs resident in HBM)."""
    z = np.empty((n, total))
    for i in range(n):
        z[i] = np.random.RandomState(42 + lo + i).standard_normal(total)
    return np.ascontiguousarray(z.T)


def step_kernel_name(n, storage="f64", forced=None, mode="full", maintenance=False):
    """which step kernel npb_step launches for n plants -- the selection rule of the launcher (npb_kernels.hip,
    NPB_LAUNCHER(step)) restated, for labels made before anything ran; NPB_STEP_KERNEL / npb_set_step_kernel (`forced`)
    override the choice by batch size.  The JSON line itself names what npb_debug_last_step_kernel reports after the run;
    tests/test_abi.py and the GPU tests hold the two together."""
    if forced is None:
        forced = os.environ.get("NPB_STEP_KERNEL", "0")
    variant = int(forced) if str(forced) in ("1", "2", "3", "4", "5") else 0
    npad = (n + 63) // 64 * 64
    if mode == "primary":
        return "npb_step_primary_kernel"
    if variant == 0:
        variant = 5 if npad <= 32768 else (2 if npad <= 45056 else (5 if npad <= 114688 else (4 if npad * (8 if storage == "f64" else 4) > 90112 * 8 else 1)))
    m = "_maint" if (maintenance and mode == "full") else ""       # the builds with the automatic maintenance compiled in
    if variant == 4:
        return "npb_step_nt%s_kernel" % m
    if variant == 5 and mode == "full":
        return "npb_step4%s_kernel" % m
    if variant in (2, 3) and mode == "full":
        return ("npb_step2_wide%s_kernel" if (variant == 2 and npad <= 32768) else "npb_step2%s_kernel") % m
    return "npb_step%s_kernel" % m


def past_the_knee(n, device, storage, bytes_per_plant, K=40):
    import ctypes
    import torch
    from nuclear_sim_amd import _lib
    from nuclear_sim_amd.env import BatchedPlantEnv
    env = BatchedPlantEnv(n, dt=1.0, heat_source="constant", noise_enabled=True, noise_std_percent=0.1, device=device, storage=storage)
    dev = env.device
    gen = torch.Generator(device=dev); gen.manual_seed(7)
    z = torch.randn((8, n), device=dev, dtype=torch.float64, generator=gen)
    sp = torch.full((n,), 92.0, device=dev, dtype=torch.float64)
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]
    W = 150       # ~30 ms of the same launches first: the clock ramp after an idle period (main(), "preconditioning")
    for t in range(K + W):
        if t >= W:
            ev[t - W][0].record()
        _lib.check(env.L.npb_step(env._h, None, None, ctypes.c_void_p(sp.data_ptr()), ctypes.c_void_p(z[t % 8].data_ptr()), None, env._p(env._obs),
                                  env._p(env._reward), env._p(env._done), env._p(env._flags), env._p(env._info), stream), env._h)
        if t >= W:
            ev[t - W][1].record()
    torch.cuda.synchronize(dev)
    ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    achieved = bytes_per_plant * n / (ms * 1e-3) / 1e9
    assert env.last_step_kernel() == step_kernel_name(n, storage), (env.last_step_kernel(), step_kernel_name(n, storage))
    return {"plants": n, "kernel": env.last_step_kernel(), "kernel_ms": ms, "achieved": achieved, "frac": achieved / HBM_PEAK_GBS}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--plants-per-gpu", type=int, default=PLANTS_PER_GPU)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-precondition", action="store_true",
                    help="skip the clock-ramp preconditioning before the warm-up steps (see the comment at its place)")
    ap.add_argument("--maintenance", action="store_true",
                    help="also run the automatic oil_top_off maintenance kernel after every step (not the headline workload)")
    ap.add_argument("--storage", choices=["f64", "f32"], default="f64",
                    help="element type of the carried state in HBM (f32 = BASELINE config 5; the headline is f64)")
    ap.add_argument("--launch-check", action="store_true",
                    help="no stepping, no GPU: the ranks rendezvous (NPB_BENCH_BACKEND, e.g. gloo), all-reduce one number and rank 0 prints "
                         "{n_gpus, launch_check: true} -- a test of the launcher path on a box without GPUs, not a measurement")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Run plainly as `python bench.py --gpus N`: nobody has set up the ranks.  Start them here -- N fresh worker processes
        # through torch's own launcher, one per GPU, each re-entering this file with RANK / LOCAL_RANK / WORLD_SIZE set -- BEFORE
        # anything in this process touches the GPU (no torch.cuda call, no HIP call: the parent stays a plain launcher, it is never
        # re-exec'ed), relay their output (rank 0 prints the JSON line) and exit with the launcher's code.
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        sys.exit(subprocess.run(cmd, env=env).returncode)

    import torch
    import torch.distributed as dist

    if args.launch_check:
        world = int(os.environ.get("WORLD_SIZE", "1"))
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group(os.environ.get("NPB_BENCH_BACKEND", "gloo"))
            one = torch.ones(1, dtype=torch.int64)
            dist.all_reduce(one)
            assert int(one.item()) == world
        if int(os.environ.get("RANK", "0")) == 0:
            print(json.dumps({"launch_check": True, "n_gpus": world, "requested_gpus": args.gpus}), flush=True)
        if world > 1:
            dist.barrier(); dist.destroy_process_group()
        return

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    distributed = world > 1
    # rehearsal on a one-GPU box (tools/bench_rehearsal.sh): every rank on device NPB_BENCH_DEVICE, collectives over gloo
    if os.environ.get("NPB_BENCH_DEVICE") is not None:
        local_rank = int(os.environ["NPB_BENCH_DEVICE"])
    backend = os.environ.get("NPB_BENCH_BACKEND", "nccl")
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
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
    total = K + W
    tt = torch.arange(total, device=dev, dtype=torch.float64)[:, None]
    target = 90.0 + 10.0 * torch.sin(2.0 * np.pi * tt / period[None, :])
    # rate-limited to 0.02 % per step exactly as the data-gen runner's _set_target_power does
    # (maintenance_scenario_runner.py:651-671): the first target is taken as it is, later ones are approached
    setpoints = torch.empty_like(target)
    setpoints[0] = target[0]
    for t in range(1, total):
        d = target[t] - setpoints[t - 1]
        setpoints[t] = torch.where(d.abs() > 0.02, setpoints[t - 1] + 0.02 * torch.sign(d), target[t])
    setpoints = setpoints.contiguous()
    noise = torch.from_numpy(c3_noise(lo, n, total)).to(dev)

    def one_step(t):
        return env.step(power_setpoint=setpoints[t % total], noise_z=noise[t % total])

    # Device preconditioning (not steps of the benchmarked plants: their state is not advanced).  After the host-side input
    # generation above the GPU has idled, and an MI355X needs ~17 ms of continuous work of THIS kind to settle at its steady
    # clocks: the step kernel runs 0.103-0.105 ms for its first ~170 launches after an idle period and 0.098 from then on
    # (tools/first_launches.py, profiles/r2_first_launches.txt), so W = 5 warm-up steps + K = 20 timed steps (2.5 ms) would
    # measure the ramp, not the rate of a job that steps continuously.  A scratch handle of the same size is stepped for
    # PRECONDITION_MS on the same inputs before the W warm-up steps, then freed (sweeping the arena with a copy kernel for as
    # long brings only half of it back: the governor follows the kind of work, profiles/r2_bench_preconditioning.txt).
    # --no-precondition turns it off; the JSON line says what was done.
    precondition_ms = 0.0
    if not args.no_precondition:
        if distributed:
            dist.barrier()        # communicator start-up happens here, not between the preconditioning and the warm-up
        probe = os.environ.get("NPB_PLACEMENT_PROBE")
        os.environ["NPB_PLACEMENT_PROBE"] = "0"        # the scratch handle's own speed does not matter
        scratch_env = BatchedPlantEnv(n, dt=1.0, heat_source="constant", noise_enabled=True, noise_std_percent=0.1, device=local_rank,
                                      maintenance=args.maintenance, storage=args.storage)
        if probe is None:
            os.environ.pop("NPB_PLACEMENT_PROBE")
        else:
            os.environ["NPB_PLACEMENT_PROBE"] = probe
        torch.cuda.synchronize(dev)
        tp = time.perf_counter()
        while (time.perf_counter() - tp) * 1e3 < PRECONDITION_MS:
            for q in range(50):
                scratch_env.step(power_setpoint=setpoints[q % total], noise_z=noise[q % total])
            torch.cuda.synchronize(dev)
        precondition_ms = (time.perf_counter() - tp) * 1e3
        scratch_env.close()
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
    import ctypes
    from nuclear_sim_amd import _lib
    stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)

    def raw_step(t):  # the C-ABI call alone: one launch of the step kernel, no host-side extras
        _lib.check(env.L.npb_step(env._h, None, None, ctypes.c_void_p(setpoints[t % total].data_ptr()),
                                  ctypes.c_void_p(noise[t % total].data_ptr()), None, env._p(env._obs), env._p(env._reward),
                                  env._p(env._done), env._p(env._flags), env._p(env._info), stream), env._h)
    for k in range(KE):
        starts[k].record()
        raw_step(W + k)
        ends[k].record()
    torch.cuda.synchronize(dev)
    kernel_ms = float(np.mean([s.elapsed_time(e) for s, e in zip(starts, ends)]))
    launched_kernel = env.last_step_kernel()       # what npb_step launched, asked of the library (npb_debug_last_step_kernel)
    assert launched_kernel == step_kernel_name(n, args.storage, maintenance=args.maintenance), (launched_kernel, step_kernel_name(n, args.storage, maintenance=args.maintenance))
    # self-check: the same loop as the timed region over >= 0.6 s of launches (the driver's --steps 20 is a 2 ms
    # timed region; one scheduling hiccup there is a 10 % error), inputs cycled
    K_long = max(K, int(np.ceil(0.6 / max(elapsed / K, 1e-6))))
    torch.cuda.synchronize(dev)
    t1 = time.perf_counter()
    for k in range(K_long):
        one_step(W + k)
    torch.cuda.synchronize(dev)
    long_elapsed = time.perf_counter() - t1
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
        # algorithmic bytes of one plant-step for this handle (npb_handle_step_bytes_per_plant: carried members read +
        # written, outputs written, per-step inputs and outputs), minus the three input columns this workload passes
        # as NULL (action 4 B, magnitude 8 B, cooling-water temperature 8 B): they are not read
        null_input_bytes = 4 + 8 + 8
        bytes_per_plant = env.handle_step_bytes_per_plant() - null_input_bytes
        achieved = bytes_per_plant * n / (kernel_ms * 1e-3) / 1e9
        traffic, traffic_src = measured_traffic(n) if args.storage == "f64" else (None, None)
        traffic_rw = measured_traffic_split(n) if args.storage == "f64" else None
        # reads: carried members + int32 members + the inputs passed; writes: the same state + the outputs (floats) + obs / reward / done / flags / info
        algorithmic_rw = algorithmic_read_write_bytes(env, bytes_per_plant)
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
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": ("%s: committed rocprofv3 PMC passes of an earlier run of this workload, "
                                            "not measured by this run" % traffic_src) if traffic_src else None,
                         "frac_of_peak_by_traffic": (traffic / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                         "algorithmic_bytes_per_launch": bytes_per_plant * n,
                         "read_bytes": traffic_rw[0] if traffic_rw else None, "write_bytes": traffic_rw[1] if traffic_rw else None,
                         "algorithmic_read_bytes": algorithmic_rw[0] * n, "algorithmic_write_bytes": algorithmic_rw[1] * n,
                         "bytes_not_moved": "the algorithmic figure counts every carried column as read + written; WRITES come out below "
                                            "it because the kernel skips the store of a column whose bits did not change for any plant of "
                                            "a wave (unchanged-column elision); READS come out ABOVE it (FETCH_SIZE counts what the L2 "
                                            "fetches, Infinity-Cache hits included: the cache does not hide reads from this counter): "
                                            "by the counters these are the kernel's own loads (the vector caches request the same bytes). "
                                            "Columns two waves of a group both read (stage arrays, part of the turbine section, the plant's "
                                            "clock) and the narrow columns that mix output and int32 members are candidates, but removing "
                                            "224 B per plant of such second reads lowered FETCH_SIZE by 35 B per plant (they hit the L2; "
                                            "DESIGN.md section 3, round 4): the excess is not accounted for column by column; NULL inputs "
                                            "(20 B/plant) are already excluded",
                         "kernel": launched_kernel + (" (the build with the automatic maintenance inside: threshold screen in the pump phase, rule by function call for flagged waves)" if args.maintenance else ""),
                         "kernel_ms": kernel_ms},
            "preconditioning": {"ms": precondition_ms, "what": "a scratch handle of the same size stepped on the same inputs before the %d warm-up "
                                "steps of the benchmarked one (whose state is not advanced), so that the timed steps run at the GPU's "
                                "steady clocks rather than in the ~17 ms ramp after the idle input generation "
                                "(profiles/r2_first_launches.txt); --no-precondition measures the ramp" % W},
            "selfcheck": {"steps": K_long, "seconds": long_elapsed, "value": n * K_long / long_elapsed, "ms_per_step": long_elapsed / K_long * 1e3,
                          "what": "rank 0's own rate over a longer run of the same loop (inputs cycled); not the headline"},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
            # the same kernel with the working set well past the 256 MB Infinity Cache (twice the plants): what the fraction is
            # when nothing of the state survives in the memory-side cache between two steps
            out["roofline"]["past_the_cache_knee"] = past_the_knee(2 * n, local_rank, args.storage, bytes_per_plant)
        print(json.dumps(out), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
