#!/usr/bin/env python3
"""Diagnostic: does the step kernel's time depend on WHERE the arena was allocated?  Several handles are created in
one process and kept alive (so each arena sits in different physical memory) and timed in turn, twice."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nuclear_sim_amd.env import BatchedPlantEnv
n = 65536
envs = []
for i in range(6):
    envs.append(BatchedPlantEnv(n, dt=1.0, heat_source="constant", noise_enabled=True, noise_std_percent=0.1))
    if i == 2:
        pad = torch.empty(3 * 1024 ** 3, dtype=torch.uint8, device="cuda")   # shift the later arenas by 3 GiB
dev = envs[0].device
z = torch.randn((64, n), device=dev, dtype=torch.float64)
sp = torch.full((n,), 92.0, device=dev, dtype=torch.float64)
def burst(env, K=300):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]
    for t in range(K):
        ev[t][0].record(); env.step(power_setpoint=sp, noise_z=z[t % 64]); ev[t][1].record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)
    return ms[0], ms[len(ms) // 2]
import ctypes
for rnd in range(2):
    for i, env in enumerate(envs):
        real = ctypes.c_void_p(); pitch = ctypes.c_size_t(); seg = ctypes.c_size_t(); ncol = ctypes.c_int(); stor = ctypes.c_int()
        env.L.npb_state_arena_layout(env._h, ctypes.byref(real), ctypes.byref(pitch), ctypes.byref(seg), ctypes.byref(ncol), ctypes.byref(stor))
        mn, md = burst(env)
        # the calibration kernel (reads and rewrites every column in the step's access shape) on the same arena
        from nuclear_sim_amd import _lib
        tev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
        for a, b in tev:
            a.record(); _lib.check(env.L.npb_debug_touch(env._h, env._stream()), env._h); b.record()
        torch.cuda.synchronize()
        tt = sorted(a.elapsed_time(b) for a, b in tev)
        print("round %d handle %d arena at 0x%012x: step min %.5f median %.5f ms; touch min %.5f median %.5f ms" % (rnd, i, real.value, mn, md, tt[0], tt[10]), flush=True)
