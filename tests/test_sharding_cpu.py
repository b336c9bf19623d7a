"""CPU, world_size 2 over gloo: the multi-rank host path (shard ranges, obs all-gather, counter reduce)."""
import os

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from nuclear_sim_amd.sharding import gather_observations, reduce_counters, shard_range


def test_shard_ranges_cover_and_are_contiguous():
    for n in (1, 7, 64, 65536, 262144, 1000003):
        for w in (1, 2, 3, 8):
            rs = [shard_range(n, r, w) for r in range(w)]
            assert rs[0][0] == 0 and rs[-1][1] == n
            assert all(rs[i][1] == rs[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in rs]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, n_global, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_range(n_global, rank, world)
    # a deterministic function of the GLOBAL plant id stands in for the stepped observation
    ids = torch.arange(lo, hi, dtype=torch.float64)
    obs_local = ids[:, None] * 100.0 + torch.arange(22, dtype=torch.float64)[None, :]
    full = gather_observations(obs_local, n_global)
    cnt = reduce_counters(torch.tensor([hi - lo, rank + 1, 0, 5], dtype=torch.int64))
    if rank == 0:
        q.put((full.numpy(), cnt.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_and_reduce_world2():
    n_global = 1001  # ragged on purpose
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_global, q)) for r in range(2)]
    for p in procs:
        p.start()
    full, cnt = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    expect = np.arange(n_global, dtype=np.float64)[:, None] * 100.0 + np.arange(22, dtype=np.float64)[None, :]
    assert np.array_equal(full, expect)
    assert cnt.tolist() == [n_global, 3, 0, 10]


def _episode_inputs(n_global, T):
    """per-plant inputs as a function of the GLOBAL plant id only (what bench.py / tools/config4.py do)"""
    gid = np.arange(n_global)
    sp = 90.0 + 10.0 * np.sin(2.0 * np.pi * np.arange(T)[:, None] / (600.0 + 60.0 * (gid % 16))[None, :])
    z = np.stack([np.random.RandomState(42 + int(i)).standard_normal(T) for i in gid], axis=1)
    oil = 57.0 + (gid % 7) * 0.5          # some pumps cross the 58 % maintenance threshold
    return sp, z, oil


def _stepping_worker(rank, world, port, n_global, T, q):
    """One rank of a sharded episode: steps ITS plants (the CPU oracle stands in for the GPU stepper here -- this test is
    about the host logic: shard ranges, inputs by global id, the episode-end collectives), then gathers."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import npo
    lo, hi = shard_range(n_global, rank, world)
    sp, z, oil = _episode_inputs(n_global, T)
    P = npo.Params(); P.hs_noise_enabled = 1; P.maint_enabled = 1; P.dt = 5.0
    ora = npo.OraclePlants(hi - lo, P)
    ora.set("pump.oil_level", oil[lo:hi], instance=0)
    for t in range(T):
        obs, rew, done, flags, info = ora.step(setpoint=sp[t, lo:hi], noise_z=z[t, lo:hi])
    full = gather_observations(torch.from_numpy(obs), n_global)
    events = sum(ora.get("maint.maintenance_actions_performed", plant=i) for i in range(hi - lo))
    cnt = reduce_counters(torch.tensor([int((flags & 1).astype(bool).sum()), events], dtype=torch.int64))
    if rank == 0:
        q.put((full.numpy(), cnt.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_episode_does_not_depend_on_the_world_size():
    """'results independent of G'.  The same 37-plant episode (ragged shards) stepped by one rank and by two:
    the gathered observation block and the reduced counters must be identical."""
    n_global, T = 37, 12
    ctx = mp.get_context("spawn")
    results = {}
    for world in (1, 2):
        q = ctx.Queue()
        port = 31500 + (os.getpid() % 2000) + world
        procs = [ctx.Process(target=_stepping_worker, args=(r, world, port, n_global, T, q)) for r in range(world)]
        for p in procs:
            p.start()
        results[world] = q.get(timeout=300)
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
    assert np.array_equal(results[1][0], results[2][0])
    assert results[1][1].tolist() == results[2][1].tolist()
    assert results[1][1][1] > 0, "some plants had a maintenance event"


def _config4_worker(rank, world, port, n_global, T, q):
    """One rank of BASELINE config 4's run at test size: ITS seeds' plants from the scenario catalog (seeds = global plant ids),
    stepped by the CPU oracle standing in for the GPU stepper, then the job-wide histogram of executions per plant."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import npo
    from nuclear_sim_amd import scenarios
    from nuclear_sim_amd.sharding import event_histogram
    lo, hi = shard_range(n_global, rank, world)
    P = npo.Params(); P.hs_noise_enabled = 1; P.maint_enabled = 1; P.dt = 5.0
    ora = npo.OraclePlants(hi - lo, P)
    eff = float(ora.get("pump.lubrication_effectiveness"))
    for key, v in scenarios.action_test_fields("oil_top_off", list(range(lo, hi)), eff).items():
        name, inst, k = (key, 0, 0) if not isinstance(key, tuple) else (key[0], key[1], key[2] if len(key) > 2 else 0)
        ora.set(name, v, instance=inst, k=k)
    z = np.random.RandomState(42).standard_normal(T)
    for t in range(T):
        ora.step(setpoint=np.full(hi - lo, 90.0), noise_z=np.full(hi - lo, z[t]))
    events = torch.tensor([ora.get("maint.maintenance_actions_performed", plant=i) for i in range(hi - lo)], dtype=torch.int64)
    hist = event_histogram(events)
    if rank == 0:
        q.put(hist.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_config4_histogram_does_not_depend_on_the_world_size():
    """tools/config4.py's episode-end report: the histogram of oil_top_off executions per plant over all ranks (seeds by global
    plant id, shards contiguous, one all-reduce).  45 seeds as one rank and as two ragged shards: the same histogram, with
    plants that top off and plants that never do."""
    n_global, T = 45, 30
    ctx = mp.get_context("spawn")
    hists = {}
    for world in (1, 2):
        q = ctx.Queue()
        port = 33500 + (os.getpid() % 2000) + world
        procs = [ctx.Process(target=_config4_worker, args=(r, world, port, n_global, T, q)) for r in range(world)]
        for p in procs:
            p.start()
        hists[world] = q.get(timeout=300)
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
    assert np.array_equal(hists[1], hists[2])
    assert hists[1].sum() == n_global and hists[1][0] > 0 and hists[1][1:].sum() > 0


def test_bench_starts_its_own_ranks_when_nobody_has():
    """`python bench.py --gpus 2` run plainly (no WORLD_SIZE in the environment, no outside launcher): the parent must start the
    two worker processes itself, before it touches any GPU, and relay rank 0's line with n_gpus = 2.  Here without a GPU, so with
    --launch-check (rendezvous + one all-reduce over gloo, no stepping); the real path with two ranks sharing one device is
    rehearsed on the GPU box by tools/bench_rehearsal.sh."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["NPB_BENCH_BACKEND"] = "gloo"
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launch-check"], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2 and lines[0]["launch_check"] is True
    text = open(os.path.join(root, "bench.py")).read()
    launch = text.index('if args.gpus > 1 and "WORLD_SIZE" not in os.environ')
    assert "import torch" not in text[text.index("def main():"):launch], "main() starts the ranks before it imports torch: the parent never touches the GPU"
