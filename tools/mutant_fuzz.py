#!/usr/bin/env python3
"""Can ANY plant state tell a surviving mutant of the restatement from the original?  (CPU; companion of tools/mutate_oracle.py.)

A mutant that the reference's fixtures do not reject is either a hole in the fixtures or a change that cannot matter -- a clip bound
no state reaches, a rate constant whose term the reference's own clips swallow (its point kinetics runs into the +-10 % flux-rate clip
and pins the precursors at the [0, 1] clip: beta and the six decay constants cannot show), a branch behind a literal input.  This tool
separates the two by brute force on the ORACLE alone: every survivor is built, and original and mutant step the same bank of random
plants -- every fp64 member of a constructed plant scaled by an independent factor in [0.3, 3] with probability 0.7 (so levels, wear,
deposits, temperatures, pressures, timers far outside anything a run visits), flags flipped, pump states redrawn, under random
operator actions, set-points 0-115 %, cooling water 2-45 C, in six configurations (constant / reactor heat source x dt 0.1, 1, 5) -- for
a few steps, and every state member and output is compared BIT FOR BIT.
    indistinguishable   no difference in any of the ~10 000 plant-steps: the mutated token is unobservable at the reference's constants
    distinguishable     some state shows it: the fixtures should too (an example is recorded: configuration, plant, first differing member)
Results go into the record of tools/mutate_oracle.py (profiles/r4_mutation_score.json: each survivor gets "fuzz").
    python3 tools/mutant_fuzz.py [--jobs 6] [--plants 1500]
"""
import argparse
import concurrent.futures as cf
import ctypes
import json
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ORACLE = os.path.join(ROOT, "oracle")
CONFIGS = [(hs, dt) for hs in (0, 1) for dt in (1.0, 5.0, 0.1)]


def load(path):
    from oracle import npo
    os.environ["NPO_LIB"] = path
    npo._LIB = None
    return npo.lib()


def bank(npo, n, hs, dt, seed):
    """n random plants (original library): the raw buffer, parameters and three steps of inputs"""
    rng = np.random.default_rng(seed)
    P = npo.Params(); P.dt = dt; P.heat_source = hs; P.hs_noise_enabled = 1; P.maint_enabled = int(seed % 2)
    o = npo.OraclePlants(n, P)
    F, I = o.state_all()
    scale = np.where(rng.random(F.shape) < 0.7, rng.uniform(0.3, 3.0, F.shape), 1.0)
    F = F * scale
    cols = o.schema.columns()
    for kind, slot, label, _p in cols:
        if kind != "i32" or label.startswith(("maint.", "mpump.")):
            continue
        if label.endswith(".status"):
            I[:, slot] = np.where(rng.random(n) < 0.3, rng.integers(0, 5, n), I[:, slot])
        elif label.endswith(("mask", "count", "ejector", "reason")):
            continue
        else:
            I[:, slot] = np.where(rng.random(n) < 0.2, 1 - (I[:, slot] != 0), I[:, slot])
    for pl in range(n):
        o.set_state(F[pl], I[pl], plant=pl)
    steps = []
    for _t in range(3):
        steps.append(dict(action=rng.choice([0, 1, 2, 3, 4, 5, 8, 9, 10], n).astype(np.int32), magnitude=rng.uniform(0, 1, n),
                          setpoint=np.where(rng.random(n) < 0.5, rng.uniform(0, 115, n), np.nan), noise_z=rng.standard_normal(n),
                          cw_temp=np.where(rng.random(n) < 0.5, rng.uniform(2, 45, n), np.nan)))
    return o, P, steps


def run(npo, o, steps):
    outs = []
    for s in steps:
        obs, rew, done, flags, info = o.step(**s)
        outs += [obs.copy(), rew.copy(), done.copy(), flags.copy(), info.copy()]
    F, I = o.state_all()
    return outs + [F, I]


def first_difference(a, b, cols):
    for k, (x, y) in enumerate(zip(a, b)):
        same = (x == y) | ((x != x) & (y != y)) if x.dtype.kind == "f" else (x == y)
        if not same.all():
            idx = np.argwhere(~same)[0]
            what = ["obs", "reward", "done", "flags", "info"][k % 5] + " step %d" % (k // 5) if k < len(a) - 2 else ("f64 " + [c[2] for c in cols if c[0] == "f64"][idx[1]] if k == len(a) - 2 else "i32 " + [c[2] for c in cols if c[0] == "i32"][idx[1]])
            return "%s, plant %d: %r vs %r" % (what, int(idx[0]), x[tuple(idx)].item(), y[tuple(idx)].item())
    return None


def classify(job):
    rec, n = job
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import mutate_oracle as M
    from oracle import npo
    work = tempfile.mkdtemp(prefix="npo_fz_")
    try:
        for f in os.listdir(ORACLE):
            if f.endswith((".h", ".c")):
                shutil.copy(os.path.join(ORACLE, f), work)
        path = os.path.join(work, rec["file"])
        lines = open(path).read().split("\n")
        site = [s for s in M.sites(os.path.join(ORACLE, rec["file"])) if s[0] == rec["line"] - 1 and s[4] == rec["op"] and s[3] == rec["now"] and lines[s[0]][s[1]:s[1] + s[2]] == rec["was"]]
        if not site:
            return dict(rec, fuzz="site not found in the present text")
        out = dict(rec)
        verdicts = []
        for ln, col, length, rep, _k in site:            # (a line can hold the same token twice: each is its own mutant)
            ml = list(lines); ml[ln] = ml[ln][:col] + rep + ml[ln][col + length:]
            open(path, "w").write("\n".join(ml))
            so = os.path.join(work, "libnpo_%d.so" % col)
            cc = subprocess.run(["gcc", "-O2", "-fPIC", "-std=gnu11", "-ffp-contract=off", "-fno-fast-math", "-fvisibility=hidden", "-fopenmp", "-w", "-I", os.path.join(ROOT, "include"),
                                 "-I", work, "-shared", "-o", so, os.path.join(work, "npo_api.c"), "-lm"], capture_output=True, text=True, cwd=work)
            if cc.returncode != 0:
                verdicts.append("stillborn"); continue
            found = None
            for c, (hs, dt) in enumerate(CONFIGS):
                load(os.path.join(ORACLE, "libnpo.so"))
                o, P, steps = bank(npo, n, hs, dt, 1000 + c)
                raw = o._buf.copy()
                ref = run(npo, o, steps)
                load(so)
                m = npo.OraclePlants(n, P)
                m._buf[:] = raw
                got = run(npo, m, steps)
                d = first_difference(ref, got, o.schema.columns())
                if d:
                    found = "heat source %d, dt %g: %s" % (hs, dt, d); break
            verdicts.append("distinguishable: " + found if found else "indistinguishable")
        out["fuzz"] = verdicts[0] if len(verdicts) == 1 else "; ".join(verdicts)
        return out
    finally:
        shutil.rmtree(work, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--jobs", type=int, default=6)
    ap.add_argument("--plants", type=int, default=1500)
    ap.add_argument("--record", default=os.path.join(ROOT, "profiles", "r4_mutation_score.json"))
    args = ap.parse_args()
    rec = json.load(open(args.record))
    jobs = [(r, args.plants) for r in rec["survivors"]]
    out = []
    with cf.ProcessPoolExecutor(args.jobs) as pool:
        for k, r in enumerate(pool.map(classify, jobs, chunksize=2)):
            out.append(r)
            if (k + 1) % 50 == 0:
                print("%d / %d   indistinguishable %d" % (k + 1, len(jobs), sum(x["fuzz"].startswith("indistinguishable") for x in out)), flush=True)
    rec["survivors"] = out
    ind = sum(x["fuzz"].startswith("indistinguishable") for x in out)
    rec["fuzz"] = {"what": "tools/mutant_fuzz.py: original and mutant stepped on %d random plants x 3 steps x %d configurations, compared bit for bit" % (args.plants, len(CONFIGS)),
                   "indistinguishable": ind, "distinguishable": len(out) - ind}
    json.dump(rec, open(args.record, "w"), indent=1)
    print(json.dumps(rec["fuzz"], indent=1))


if __name__ == "__main__":
    main()
