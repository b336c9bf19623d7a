#!/usr/bin/env python3
"""Where the four wavefronts of npb_step4_kernel spend their time: the diagnostic build (make -C nuclear_sim_amd/csrc stamps)
stamps the cycle counter before and after every barrier, per wave role; this prints work and wait per segment.

  python3 tools/phase_stamps4.py [plants] [steps]
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("NPB_LIB", os.path.join(ROOT, "nuclear_sim_amd", "ablate", "libnpb_stamps.so"))
os.environ["NPB_STEP_KERNEL"] = "5"

# stamp k of each wave (npd_step4.h, NPD4_STAMP): what the wave did between stamp k-1 and stamp k ("wait: ..." = polling a progress word)
LABELS = {
    0: ["primary side", "pump 3", "wait: primary (own)", "SG 0 part 1", "wait: feedwater flow", "SG 0 part 2", "stage arrays preload", "pass B units 0,2,..",
        "wait: verdict; stage post (behind the chain)", "wait: tail", "observation, flags"],
    1: ["chemistry sidecar", "wait: level control; pump 1", "wait: primary", "SG 1 part 1", "wait: feedwater flow", "SG 1 part 2", "stage efficiencies",
        "stage chain (behind pass B), wait: verdict", "-", "turbine lubrication pre-step", "-", "wait: tail, condenser", "reward, write-back"],
    2: ["-", "wait: level control; pump 2", "wait: primary", "SG 2 part 1", "wait: feedwater flow", "SG 2 part 2", "stage arrays preload", "pass B units 1,3,..",
        "load condenser; wait: verdict; stage post (behind the chain)", "wait: chain; condenser"],
    3: ["prelude, level control", "pump 0", "wait: pumps 1-3", "pump tails, system level", "fw store, turbine load, stage arrays preload",
        "wait: steam generators", "SG sums, stage pass A, verdict", "pass B units 12-14, stage post 2,5,8,11 (behind the chain)", "wait: chain; rotor", "wait: stage post",
        "protection, gates, tail", "turbine section to the arena; wait: condenser", "info"],
}


def main():
    import numpy as np
    import torch
    from nuclear_sim_amd.env import BatchedPlantEnv
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    env = BatchedPlantEnv(n, noise_enabled=True)
    groups = (n + 63) // 64
    buf = torch.zeros(groups * 4 * 32, dtype=torch.int64, device=env.device)
    env.L.npb_debug_set_stamp_buffer.argtypes = [ctypes.c_void_p]
    assert env.L.npb_debug_set_stamp_buffer(ctypes.c_void_p(buf.data_ptr())) == 0
    gen = torch.Generator(device=env.device); gen.manual_seed(1)
    z = torch.randn((K + 5, n), device=env.device, dtype=torch.float64, generator=gen)
    sp = torch.full((n,), 95.0, device=env.device, dtype=torch.float64)
    for t in range(5):
        env.step(power_setpoint=sp, noise_z=z[t])
    torch.cuda.synchronize()
    assert env.last_step_kernel() == "npb_step4_kernel", env.last_step_kernel()
    acc = 0
    for t in range(K):
        buf.zero_()
        env.step(power_setpoint=sp, noise_z=z[5 + t])
        torch.cuda.synchronize()
        s = buf.cpu().numpy().reshape(groups, 4, 32).astype(np.float64)
        acc = acc + (s - s[:, :1, :1])      # relative to the group's wave-0 start stamp
    s = acc / K
    print("%d plants, mean over %d groups x %d steps, ticks relative to each group's start" % (n, groups, K))
    total = s[:, :, 31].max(axis=1).mean()
    for w in range(4):
        print("wave %d: start %.0f end %.0f" % (w, s[:, w, 0].mean(), s[:, w, 31].mean()))
        prev = 0
        for k, label in enumerate(LABELS[w], start=1):
            d = (s[:, w, k] - s[:, w, prev]).mean()
            print("  %2d %-58s %8.0f (%4.1f %%)   at %8.0f" % (k, label, d, 100 * d / total, s[:, w, k].mean()))
            prev = k
        d = (s[:, w, 31] - s[:, w, prev]).mean()
        print("     %-58s %8.0f (%4.1f %%)" % ("to the end (maintenance barrier)", d, 100 * d / total))
    print("group lifetime %.0f ticks" % total)


if __name__ == "__main__":
    main()
