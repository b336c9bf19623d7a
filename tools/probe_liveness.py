#!/usr/bin/env python3
"""Which state members does the step kernel never read?  Compiles npb_kernels.hip with -DNPB_PROBE (npd_stage.h: every
loaded member passes through a non-volatile asm marker that the compiler deletes when the value is unused) and lists,
per section, the members whose marker is gone: they are dead on entry, pure outputs of the step."""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nuclear_sim_amd.schema import SCHEMA

out = "/tmp/npb_probe.s"
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
                       "-freciprocal-math", "-fapprox-func", "-DNPB_PROBE", "-w", "-S", "--cuda-device-only", "-o", out,
                       os.path.join(ROOT, "nuclear_sim_amd", "csrc", "npb_kernels.hip")])
txt = open(out).read()
a = txt.index("_Z15npb_step_kernel"); b = txt.index(".Lfunc_end0", a)
live = set((t, int(z, 0), int(k, 0)) for t, z, k in re.findall(r"PROBE_([FI]) (\S+) (\S+)", txt[a:b]))
labels = {(k, s): l for k, s, l, _p in SCHEMA.columns()}
fb = ib = tot = 0
for s in SCHEMA.sections:
    base = (fb, ib); fb += s.nf64 * s.count; ib += s.ni32 * s.count
    if not any(z == base[0] for _t, z, _k in live):
        print(s.member, "(not loaded as a struct: members are read by use)"); continue
    livef = {k for t, z, k in live if t == "F" and z == base[0]}; livei = {k for t, z, k in live if t == "I" and z == base[0]}
    deadf = sorted(set(range(s.nf64)) - livef); deadi = sorted(set(range(s.ni32)) - livei); tot += len(deadf) * s.count
    print("%s x%d  dead f64 %d/%d: %s" % (s.member, s.count, len(deadf), s.nf64, [labels[("f64", base[0] + k)].split(".", 1)[1] for k in deadf]))
    print("        dead i32 %d/%d: %s" % (len(deadi), s.ni32, [labels[("i32", base[1] + k)].split(".", 1)[1] for k in deadi]))
print("dead real columns, instances counted:", tot)
