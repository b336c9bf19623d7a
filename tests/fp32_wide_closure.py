"""Closure search behind DESIGN.md's pricing of "fp32 storage with fp64 accumulators" (not a test; pytest does not collect it).

Steps the CPU oracle twice on the same inputs -- plain fp64, and with its state rounded to float after every step except for a set
of columns kept in fp64 -- and grows that set by every column that leaves the tolerance, until none is left or the columns still
outside are already in the set (an integrator fed by float-rounded inputs drifts even when it is itself kept in fp64).
Run:  python tests/fp32_wide_closure.py STEPS DT TOL
"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from oracle import npo
from nuclear_sim_amd.schema import SCHEMA

cols = {(k, s): l for k, s, l, _p in SCHEMA.columns()}
NF = SCHEMA.total_f64

def run(steps, dt, keep, n=8, seed=3, heat=0):
    P = npo.Params(); P.hs_noise_enabled = 1; P.dt = dt
    if heat: P.heat_source = 1
    a = npo.OraclePlants(n, P); b = npo.OraclePlants(n, P)
    b.round_state_f32(keep)
    rng = np.random.default_rng(seed)
    worst = np.zeros(NF)
    for t in range(steps):
        z = rng.standard_normal(n); sp = np.full(n, 90.0 + 8 * np.sin(t * dt / 500.0 + np.arange(n)))
        a.step(setpoint=sp, noise_z=z); b.step(setpoint=sp, noise_z=z); b.round_state_f32(keep)
        if t % 500 == 499 or t == steps - 1:
            fa, ia = a.state_all(); fb, ib = b.state_all()
            rel = (np.abs(fa - fb) / np.maximum(np.abs(fa), 1e-3)).max(0)
            worst = np.maximum(worst, rel)
    return worst, int((ia != ib).sum())

steps = int(sys.argv[1]); dt = float(sys.argv[2]); tol = float(sys.argv[3])
keep = np.zeros(NF, dtype=np.uint8)
for it in range(12):
    w, im = run(steps, dt, keep)
    bad = np.where((w > tol) & (keep == 0))[0]
    print("iter %d wide %d: above tol %d (of which new %d), worst %.2e, int mismatches %d" % (it, keep.sum(), (w > tol).sum(), len(bad), w.max(), im), flush=True)
    if len(bad) == 0: break
    keep[bad] = 1
names = sorted(set(cols[("f64", int(s))] for s in np.where(keep)[0]))
print(len(names)); print("\n".join(names))
