#!/usr/bin/env python3
"""Is a full-chip batch better stepped as several launches of the four-wave kernel, each with every wave resident at once?
K handles of n/K plants stepped one after the other on one stream, against one handle of n plants (the kernel npb_step picks).

  python3 tools/chunked_step.py [n] [steps]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from nuclear_sim_amd.env import BatchedPlantEnv
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    for chunks in (1, n // 32768, 1, n // 32768):
        m = n // chunks
        envs = [BatchedPlantEnv(m, noise_enabled=True) for _ in range(chunks)]
        gen = torch.Generator(device=envs[0].device); gen.manual_seed(1)
        z = torch.randn((steps + 50, m), device=envs[0].device, dtype=torch.float64, generator=gen)
        sp = torch.full((m,), 95.0, device=envs[0].device, dtype=torch.float64)
        for t in range(50):
            for e in envs:
                e.step(power_setpoint=sp, noise_z=z[t])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in range(steps):
            for e in envs:
                e.step(power_setpoint=sp, noise_z=z[50 + t])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        print("%d plants as %d handle(s) of %d (%s): %.4f ms per step of the whole batch, %.3g plant-env-steps/s" %
              (n, chunks, m, envs[0].last_step_kernel(), dt * 1e3, n / dt))
        del envs


if __name__ == "__main__":
    main()
