#!/usr/bin/env python3
"""Phase attribution of npb_step_kernel on the GPU: runs the diagnostic build
(make -C nuclear_sim_amd/csrc stamps -> nuclear_sim_amd/ablate/libnpb_stamps.so), in which lane 0 of
every wave records s_memtime at phase boundaries, and prints the mean time between stamps.

  NPB_LIB=nuclear_sim_amd/ablate/libnpb_stamps.so python3 tools/phase_stamps.py [plants] [steps]
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("NPB_LIB", os.path.join(ROOT, "nuclear_sim_amd", "ablate", "libnpb_stamps.so"))

NAMES = {0: "start", 1: "primary+coupling done", 2: "pump0 start (fw ctrl done)", 3: "pump1 start", 4: "pump2 start",
         5: "pump3 start", 6: "pumps done", 7: "fw finish done", 8: "sg0 start", 9: "sg1 start", 10: "sg2 start",
         11: "sgs done", 12: "turbine lube done", 13: "stage pass A done", 14: "stage pass B done",
         15: "stage pass C done", 18: "turbine rest done", 19: "condenser done", 20: "chem sidecar done",
         21: "tail scalars done", 22: "obs/info stored"}


def main():
    import numpy as np
    import torch
    from nuclear_sim_amd.env import BatchedPlantEnv
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    env = BatchedPlantEnv(n, noise_enabled=True)
    waves = (n + 63) // 64
    buf = torch.zeros(waves * 32, dtype=torch.int64, device=env.device)
    env.L.npb_debug_set_stamp_buffer.argtypes = [ctypes.c_void_p]
    assert env.L.npb_debug_set_stamp_buffer(ctypes.c_void_p(buf.data_ptr())) == 0
    gen = torch.Generator(device=env.device); gen.manual_seed(1)
    z = torch.randn((K + 5, n), device=env.device, dtype=torch.float64, generator=gen)
    sp = torch.full((n,), 95.0, device=env.device, dtype=torch.float64)
    for t in range(5):
        env.step(power_setpoint=sp, noise_z=z[t])
    torch.cuda.synchronize()
    acc = None
    for t in range(K):
        buf.zero_()
        env.step(power_setpoint=sp, noise_z=z[5 + t])
        torch.cuda.synchronize()
        s = buf.cpu().numpy().reshape(waves, 32).astype(np.float64)
        acc = s if acc is None else acc + s
        last = s
    s = acc / K
    ids = sorted(NAMES)
    print("mean over %d waves x %d steps; s_memtime ticks (shader cycles, MI355X_MICROARCH.md)" % (waves, K))
    total = (last[:, 22] - last[:, 0]).mean()
    prev = ids[0]
    for k in ids[1:]:
        d = (last[:, k] - last[:, prev])
        print("%-32s %9.1f ticks  (%5.1f %%)   min %8.0f max %8.0f" % (NAMES[k], d.mean(), 100 * d.mean() / total, d.min(), d.max()))
        prev = k
    for a, b, what in ((14, 24, "pass C stage 0 (incl. the vmcnt wait)"), (24, 25, "pass C stage 1"), (25, 26, "pass C stages 2-6"),
                       (26, 27, "pass C stages 7-12"), (27, 28, "pass C stage 13"), (28, 29, "return from stage pass"),
                       (29, 30, "lgkmcnt(0) drain"), (30, 15, "issue condenser-group DMA")):
        d = last[:, b] - last[:, a]
        print("  %-40s %9.1f ticks" % (what, d.mean()))
    print("ticks inside the staging pipeline's vmcnt(0) waits: mean %.1f (%.1f %% of the wave lifetime)" %
          (last[:, 23].mean(), 100 * last[:, 23].mean() / total))
    print("wave lifetime (stamp 0 -> 22): %.1f ticks; spread of start stamps across waves: %.0f ticks" %
          (total, last[:, 0].max() - last[:, 0].min()))


if __name__ == "__main__":
    main()
