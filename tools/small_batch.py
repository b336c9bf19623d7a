#!/usr/bin/env python3
"""Diagnostic: per-step time of a plain Python stepping loop at small batch sizes (is it launch / host bound?)."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from nuclear_sim_amd.env import BatchedPlantEnv
for n in (64, 1024, 4096, 16384):
    env = BatchedPlantEnv(n, noise_enabled=True)
    z = torch.randn((64, n), device=env.device, dtype=torch.float64)
    sp = torch.full((n,), 92.0, device=env.device, dtype=torch.float64)
    for t in range(50): env.step(power_setpoint=sp, noise_z=z[t % 64])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 2000
    for t in range(K): env.step(power_setpoint=sp, noise_z=z[t % 64])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record(); env.step(power_setpoint=sp, noise_z=z[0]); b.record(); torch.cuda.synchronize()
    print("n=%6d: %.1f us per step in a Python loop, kernel alone %.1f us" % (n, dt * 1e6, a.elapsed_time(b) * 1e3))
