"""Study behind the fp32-storage mode's list of columns that stay fp64 (include/npb_fields.h, NPB_WIDE_MEMBERS).

Not a test (pytest does not collect it).  Steps the CPU oracle twice on the same inputs -- plain fp64, and with
its state rounded to float after every step except for a set of "wide" columns -- and lists the columns whose
relative deviation grows with the number of steps: slow integrators whose per-step increment is below half a
float ulp of the value stall or drift under float storage.  Run:  python tests/fp32_wide_study.py [steps] [dt]
"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
from oracle import npo
from nuclear_sim_amd.schema import SCHEMA


def wide_mask(names):
    m = np.zeros(SCHEMA.total_f64, dtype=np.uint8)
    for kind, slot, label, _p in SCHEMA.columns():
        base = label.split("[")[0] if not label.startswith(("sg[", "pump[", "chem[")) else label.split("].", 1)[1].split("[")[0]
        sect = label.split(".")[0].split("[")[0]
        if kind == "f64" and ("%s.%s" % (sect, base.split(".")[-1])) in names:
            m[slot] = 1
    return m


def run(steps, dt, wide, n=8, seed=3, report=12, tol=2e-6):
    P = npo.Params(); P.hs_noise_enabled = 1; P.dt = dt
    a = npo.OraclePlants(n, P); b = npo.OraclePlants(n, P)
    keep = wide_mask(wide)
    b.round_state_f32(keep)
    rng = np.random.default_rng(seed)
    worst_obs = 0.0
    for t in range(steps):
        z = rng.standard_normal(n); sp = np.full(n, 90.0 + 8 * np.sin(t * dt / 500.0 + np.arange(n)))
        oa = a.step(setpoint=sp, noise_z=z)[0]; ob = b.step(setpoint=sp, noise_z=z)[0]; b.round_state_f32(keep)
        if t % 50 == 0:
            worst_obs = max(worst_obs, float(np.max(np.abs(oa - ob) / np.maximum(np.abs(oa), 1e-3))))
    fa, ia = a.state_all(); fb, ib = b.state_all()
    rel = (np.abs(fa - fb) / np.maximum(np.abs(fa), 1e-3)).max(0)
    cols = {(k, s): l for k, s, l, _p in SCHEMA.columns()}
    order = np.argsort(rel)[::-1]
    print("steps %d dt %g wide %d columns: worst obs deviation %.2e, int mismatches %d, columns above %.0e: %d"
          % (steps, dt, int(keep.sum()), worst_obs, int((ia != ib).sum()), tol, int((rel > tol).sum())))
    for s in order[:report]:
        if rel[s] > tol:
            print("   %-40s rel %.2e   fp64 %.9g   float-stored %.9g" % (cols[("f64", int(s))], rel[s], fa[0, s], fb[0, s]))
    return rel


if __name__ == "__main__":
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    dt = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    names = set(sys.argv[3].split(",")) if len(sys.argv) > 3 else set()
    run(steps, dt, names, report=60)
