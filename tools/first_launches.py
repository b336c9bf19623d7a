#!/usr/bin/env python3
"""Diagnostic: per-launch time of the step kernel over the first launches after handle creation (the driver's bench run
times 20 steps after 5 warm-up steps), and again after the GPU has idled.  python3 tools/first_launches.py [n_plants]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nuclear_sim_amd.env import BatchedPlantEnv
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
K = 200
env = BatchedPlantEnv(n, dt=1.0, heat_source="constant", noise_enabled=True, noise_std_percent=0.1)
dev = env.device
gen = torch.Generator(device=dev); gen.manual_seed(1234)
z = torch.randn((K, n), device=dev, dtype=torch.float64, generator=gen)
gid = torch.arange(n, device=dev, dtype=torch.float64)
tt = torch.arange(K, device=dev, dtype=torch.float64)[:, None]
sp = 90.0 + 10.0 * torch.sin(2 * torch.pi * tt / (600.0 + 60.0 * (gid % 16))[None, :])
torch.cuda.synchronize()


def series(label):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]
    for t in range(K):
        ev[t][0].record(); env.step(power_setpoint=sp[t], noise_z=z[t]); ev[t][1].record()
    torch.cuda.synchronize()
    ms = [a.elapsed_time(b) for a, b in ev]
    print(label + ": " + " ".join("%.4f" % (sum(ms[i:i + 10]) / 10) for i in range(0, K, 10)) + "   (means of 10 launches, ms)")
    print("   first ten: " + " ".join("%.4f" % m for m in ms[:10]), flush=True)


series("right after creation")
series("continuing          ")
time.sleep(0.5); series("after 0.5 s idle    ")
time.sleep(3.0); series("after 3 s idle      ")
