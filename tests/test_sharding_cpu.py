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
