#!/usr/bin/env python3
"""A/B timing of libnpb.so builds on one box (diagnostic).  python3 tools/ab_kernel.py libA.so libB.so ...
Each build runs in its own process (NPB_LIB), on the bench workload (65 536 plants, load-following setpoints, noise);
every launch is bracketed by its own pair of events and the minimum / median / mean over 300 launches are printed,
three alternating rounds.  The minimum is the robust figure: run-to-run spread on one box is +-3 % of the mean."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child():
    sys.path.insert(0, ROOT)
    import torch
    from nuclear_sim_amd.env import BatchedPlantEnv
    n, K = int(os.environ.get("NPB_AB_N", "65536")), 300
    env = BatchedPlantEnv(n, dt=1.0, heat_source="constant", noise_enabled=True, noise_std_percent=0.1,
                          storage=os.environ.get("NPB_AB_STORAGE", "f64"), maintenance=bool(int(os.environ.get("NPB_AB_MAINT", "0"))))
    dev = env.device
    gen = torch.Generator(device=dev); gen.manual_seed(1234)
    z = torch.randn((K + 20, n), device=dev, dtype=torch.float64, generator=gen)
    gid = torch.arange(n, device=dev, dtype=torch.float64)
    period = 600.0 + 60.0 * (gid % 16)
    tt = torch.arange(K + 20, device=dev, dtype=torch.float64)[:, None]
    sp = 90.0 + 10.0 * torch.sin(2 * torch.pi * tt / period[None, :])
    for t in range(20):
        env.step(power_setpoint=sp[t], noise_z=z[t])
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]
    for t in range(K):
        ev[t][0].record(); env.step(power_setpoint=sp[20 + t], noise_z=z[20 + t]); ev[t][1].record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)
    print("min %.5f  median %.5f  mean %.5f ms" % (ms[0], ms[len(ms) // 2], sum(ms) / len(ms)))


if __name__ == "__main__":
    if os.environ.get("NPB_AB_CHILD"):
        child()
    else:
        for rnd in range(3):
            for lib in sys.argv[1:]:
                env = dict(os.environ, NPB_AB_CHILD="1", NPB_LIB=os.path.abspath(lib))
                out = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True)
                print("%-44s %s" % (os.path.basename(lib), out.stdout.strip() or out.stderr.strip()[-300:]), flush=True)
