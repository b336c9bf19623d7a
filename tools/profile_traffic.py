#!/usr/bin/env python3
"""Workload for the rocprofv3 PMC passes: K launches of the calibration kernel (known bytes) followed
by K launches of the step kernel on the bench workload.  Run once per counter:

  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d OUT/fetch -- python3 tools/profile_traffic.py
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d OUT/write -- python3 tools/profile_traffic.py
then  python3 tools/profile_traffic.py --summarize OUT  prints per-launch HBM bytes of npb_step_kernel,
scaled by (known bytes of npb_touch_kernel) / (counter reading of npb_touch_kernel).
"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
N = int(os.environ.get("NPB_PROFILE_PLANTS", "65536"))   # which step kernel runs follows from the batch size (npb_set_step_kernel)
K = 10


def run():
    import numpy as np
    import torch
    from nuclear_sim_amd.env import BatchedPlantEnv
    from nuclear_sim_amd import _lib
    env = BatchedPlantEnv(N, noise_enabled=True)
    dev = env.device
    gen = torch.Generator(device=dev); gen.manual_seed(1)
    z = torch.randn((K + 3, N), device=dev, dtype=torch.float64, generator=gen)
    sp = torch.full((N,), 95.0, device=dev, dtype=torch.float64)
    for t in range(3):
        env.step(power_setpoint=sp, noise_z=z[t])
    torch.cuda.synchronize()
    for _ in range(K):
        _lib.check(env.L.npb_debug_touch(env._h, env._stream()), env._h)
    torch.cuda.synchronize()
    for t in range(K):
        env.step(power_setpoint=sp, noise_z=z[3 + t])
    torch.cuda.synchronize()


def counter_per_kernel(d, counter):
    vals = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != counter:
                continue
            name = row["Kernel_Name"].split("(")[0]
            vals.setdefault(name, []).append(float(row["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in vals.items()}, {k: len(v) for k, v in vals.items()}


def summarize(out):
    from nuclear_sim_amd.schema import SCHEMA
    state = SCHEMA.state_bytes()
    known = state * N  # bytes read == bytes written by one touch launch
    res = {"plants": N, "known_bytes_per_touch_launch_each_way": known}
    total = 0.0
    for counter, sub in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
        avg, cnt = counter_per_kernel(os.path.join(out, sub), counter)
        touch = avg.get("npb_touch_kernel"); step = next((avg[k] for k in ("npb_step_kernel", "npb_step2_kernel", "npb_step2_wide_kernel", "npb_step4_kernel") if k in avg), None)
        if touch is None or step is None:
            res[counter] = "missing"; continue
        raw_unit_bytes = 1024.0  # rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB
        scale = known / (touch * raw_unit_bytes)
        res[counter] = {"touch_raw_kib": touch, "step_raw_kib": step, "launches": next((cnt[k] for k in ("npb_step_kernel", "npb_step2_kernel", "npb_step2_wide_kernel", "npb_step4_kernel") if k in cnt), None),
                        "calibration_scale": scale, "step_bytes_calibrated": step * raw_unit_bytes * scale}
        total += step * raw_unit_bytes * scale
    res["step_hbm_bytes_per_launch"] = total
    from nuclear_sim_amd.env import BatchedPlantEnv
    # this workload's handle runs a ConstantHeatSource (the 12 point-kinetics columns are neither read nor written:
    # -12 * 16 B) and passes action / magnitude / cooling-water temperature as NULL (-20 B), as bench.py does
    res["algorithmic_bytes_per_launch"] = (BatchedPlantEnv.step_bytes_per_plant() - 12 * 16 - 20) * N
    print(json.dumps(res, indent=1))


def summarize_sq(out):
    """Per-launch averages of every counter found under OUT (one sub-directory per --pmc pass), for
    npb_step_kernel; SQ counters are summed over the chip, so they are also shown per wave (1 024 waves)."""
    waves = N // 64   # groups of 64 plants: one wave each (npb_step_kernel) or two (npb_step2_kernel)
    res = {}
    for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if not row["Kernel_Name"].startswith(("npb_step_kernel", "npb_step2_kernel", "npb_step2_wide_kernel", "npb_step4_kernel")):
                continue
            res.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    print("%-28s %16s %14s" % ("counter", "per launch", "per wave"))
    for k in sorted(res):
        v = sum(res[k]) / len(res[k])
        print("%-28s %16.0f %14.1f" % (k, v, v / waves))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--summarize-sq":
        summarize_sq(sys.argv[2])
    elif len(sys.argv) > 2 and sys.argv[1] == "--summarize":
        summarize(sys.argv[2])
    else:
        run()
