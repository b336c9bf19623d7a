"""Which state columns does one step never read?  (study behind the "output-only" columns of the schema; not a test)

For every real-valued column: perturb it in a batch of plants that sit in different operating states, take one step
with the CPU oracle, and compare EVERYTHING (all state columns, observations, reward, done, trip flags, info) with
the unperturbed step.  A column whose perturbation never shows anywhere -- not even in itself, because the step
overwrites it -- is dead on entry: a pure output of the step.  Candidates only; each is confirmed by reading the code.
Run: python tests/dead_on_entry_study.py
"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
from oracle import npo
from nuclear_sim_amd.schema import SCHEMA


def make(n, heat, seed, warm):
    P = npo.Params(); P.hs_noise_enabled = 1; P.heat_source = heat; P.maint_enabled = 1; P.dt = 1.0
    o = npo.OraclePlants(n, P)
    rng = np.random.default_rng(seed)
    # spread the plants over operating states: low oil, high fuel temperature (scram), low flow, SG level trips
    for k in range(4):
        o.set("pump.oil_level", rng.uniform(5, 100, n), instance=k)
    o.set("prim.coolant_flow_rate", rng.uniform(3000, 25000, n))
    o.set("prim.fuel_temperature", np.where(rng.random(n) < 0.2, 1300.0, rng.uniform(400, 700, n)))
    for k in range(3):
        o.set("sg.water_level", rng.uniform(9.0, 17.0, n), instance=k)
    for t in range(warm):
        o.step(action=rng.integers(0, 15, n).astype(np.int32), magnitude=rng.uniform(0, 1, n), setpoint=rng.uniform(60, 105, n),
               noise_z=rng.standard_normal(n), cw_temp=rng.uniform(15, 35, n))
    return o, P, rng


def snapshot(o, res):
    f, i = o.state_all()
    return [f, i] + [np.asarray(r) for r in res]


def main():
    n = 96
    cols = [c for c in SCHEMA.columns() if c[0] == "f64"]
    dead = np.ones(len(cols), dtype=bool)
    for heat in (0, 1):
        for warm in (0, 3, 40):
            base, P, rng = make(n, heat, 11 + warm, warm)
            f0, i0 = base.state_all()
            inp = dict(action=rng.integers(0, 15, n).astype(np.int32), magnitude=rng.uniform(0, 1, n), setpoint=rng.uniform(60, 105, n),
                       noise_z=rng.standard_normal(n), cw_temp=rng.uniform(15, 35, n))
            ref = npo.OraclePlants(n, P)
            for p in range(n):
                ref.set_state(f0[p], i0[p], plant=p)
            want = snapshot(ref, ref.step(**inp))
            for j, (_k, slot, label, _p) in enumerate(cols):
                if not dead[j]:
                    continue
                o = npo.OraclePlants(n, P)
                f1 = f0.copy(); f1[:, slot] = f1[:, slot] * 1.37 + 0.123
                for p in range(n):
                    o.set_state(f1[p], i0[p], plant=p)
                got = snapshot(o, o.step(**inp))
                same = all(np.array_equal(a, b, equal_nan=True) for a, b in zip(want, got))
                if not same:
                    dead[j] = False
            print("heat %d warm %d: %d candidates left" % (heat, warm, int(dead.sum())), flush=True)
    for j, (_k, slot, label, _p) in enumerate(cols):
        if dead[j]:
            print("  dead on entry:", slot, label)


if __name__ == "__main__":
    main()
