#!/usr/bin/env python3
"""BASELINE config 2 as a run: reactor point kinetics + steam generators only (mode="primary_sg"), dt = 0.1, random actuator
actions, in both integrator modes (the reference's clipped Euler; RK4 sub-steps inside the kernel), at 4 096 plants (the config)
and 65 536.  python3 tools/config2.py"""
import sys, time, torch, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nuclear_sim_amd.env import BatchedPlantEnv, equilibrium_state, config2_draws
for integ in ("reference", "rk4"):
    for n in (4096, 65536):
        env = BatchedPlantEnv(n, dt=0.1, heat_source="reactor", mode="primary_sg", integrator=integ)
        env.set_fields(equilibrium_state(*config2_draws(n)))      # one equilibrium state per plant
        a = torch.randint(0, 4, (n,), dtype=torch.int32, device=env.device); m = torch.rand(n, dtype=torch.float64, device=env.device)
        for _ in range(20): env.step(action=a, magnitude=m)
        torch.cuda.synchronize(); t = time.perf_counter(); K = 300
        for _ in range(K): env.step(action=a, magnitude=m)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / K
        print("config 2 shape: %6d plants, integrator %-9s %.4f ms per step, %.3e plant-env-steps/s" % (n, integ, dt * 1e3, n / dt), flush=True)
