#!/usr/bin/env python3
"""Where the two wavefronts of npb_step2_kernel spend their time: the diagnostic build (make -C nuclear_sim_amd/csrc stamps)
stamps s_memtime before and after every barrier, per wave role; this prints work and wait per segment.

  python3 tools/phase_stamps2.py [plants] [steps]
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("NPB_LIB", os.path.join(ROOT, "nuclear_sim_amd", "ablate", "libnpb_stamps.so"))

# segment j ends at barrier j of npd_step2.h (NPD2_SYNCJ); the work listed is what the wave does before that barrier
SEG = {1: ("A: primary, feedwater control", "B: turbine lube, chemistry sidecar"),
       2: ("A: pumps 0,1", "B: pumps 2,3"),
       5: ("A: pump tails, fw finish, SG 0, load turbine", "B: SG 1,2 part 1; wait for fw flow; part 2"),
       7: ("A: SG sums, stage pass A", "B: preload the stage arrays"),
       8: ("A: pass B, stages 0..10", "B: pass B, stages 11..13 and extractions"),
       9: ("A: stage chain, rotor", "B: stage post (behind the chain's flags), load condenser group"),
       11: ("A: protection, turbine store, gates", "B: condenser"),
       12: ("A: write-back, observations", "B: reward, secondary write-back, info")}


def main():
    import numpy as np
    import torch
    from nuclear_sim_amd.env import BatchedPlantEnv
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    env = BatchedPlantEnv(n, noise_enabled=True)
    groups = (n + 63) // 64
    buf = torch.zeros(groups * 2 * 32, dtype=torch.int64, device=env.device)
    env.L.npb_debug_set_stamp_buffer.argtypes = [ctypes.c_void_p]
    assert env.L.npb_debug_set_stamp_buffer(ctypes.c_void_p(buf.data_ptr())) == 0
    gen = torch.Generator(device=env.device); gen.manual_seed(1)
    z = torch.randn((K + 5, n), device=env.device, dtype=torch.float64, generator=gen)
    sp = torch.full((n,), 95.0, device=env.device, dtype=torch.float64)
    for t in range(5):
        env.step(power_setpoint=sp, noise_z=z[t])
    torch.cuda.synchronize()
    acc = 0
    for t in range(K):
        buf.zero_()
        env.step(power_setpoint=sp, noise_z=z[5 + t])
        torch.cuda.synchronize()
        s = buf.cpu().numpy().reshape(groups, 2, 32).astype(np.float64)
        acc = acc + (s - s[:, :1, :1])      # relative to the group's wave-A start stamp
        raw = s
    s = acc / K
    print("%d plants, mean over %d groups x %d steps, ticks relative to each group's start" % (n, groups, K))
    total = s[:, :, 31].max(axis=1).mean()
    for w, role in ((0, "A"), (1, "B")):
        print("wave %s: start %.0f end %.0f" % (role, s[:, w, 0].mean(), s[:, w, 31].mean()))
        prev = 0
        for j in (1, 2, 5, 7, 8, 9, 11):
            work = (s[:, w, 2 * j - 1] - s[:, w, prev]).mean(); wait = (s[:, w, 2 * j] - s[:, w, 2 * j - 1]).mean()
            print("  seg %2d %-36s work %8.0f (%4.1f %%)  wait at barrier %8.0f (%4.1f %%)" %
                  (j, SEG[j][w], work, 100 * work / total, wait, 100 * wait / total))
            prev = 2 * j
        work = (s[:, w, 31] - s[:, w, prev]).mean()
        print("  seg 12 %-36s work %8.0f (%4.1f %%)" % (SEG[12][w], work, 100 * work / total))
    print("group lifetime %.0f ticks" % total)


if __name__ == "__main__":
    main()
