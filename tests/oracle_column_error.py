"""How closely does the CPU restatement reproduce the reference, column by column?  (CPU, needs oracle/libnpo.so.)

Replays every trajectory fixture on the oracle and records, per fp64 state column, the largest relative difference from the
reference's recorded value over every sampled state.  Result (round 4): 645 of 810 columns come out BIT-IDENTICAL in every sample of
every fixture -- both sides evaluate the same IEEE fp64 expressions in the same order -- and the rest stay below 1e-10 (libm's pow /
log10 / exp against numpy's; a reset's sqrt).  Writes tests/golden/oracle_exact_columns.json: the columns that are bit-identical,
which tests/test_oracle_golden.py then holds to 4e-16 (two ulps) instead of to a tolerance a thousandth-of-an-increment error would
slip under (tools/mutate_oracle.py found such survivors).        python tests/oracle_column_error.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, HERE)
from golden_util import Golden, fixture_names, GOLDEN_DIR  # noqa: E402
import test_oracle_golden as tg  # noqa: E402
from oracle import npo  # noqa: E402
from nuclear_sim_amd.schema import SCHEMA  # noqa: E402


# fixtures left out of the intersection (and held to the oracle's relative tolerance only): a step of 0.12 s rounds differently in a few dozen
# columns that are bit-identical at every other dt, and one such fixture should not take the two-ulp check away from the other hundred
NOT_HELD_ON = ("c20_rotor_dynamics",)


def main():
    worst = {}
    for name in fixture_names():
        if name in NOT_HELD_ON:
            continue
        g = Golden(name)
        o = npo.OraclePlants(1, tg._configure(npo, g))
        f0, i0 = o.state(); f, i, fm, im = g.split_state(g.state[0]); f0[fm] = f[fm]; i0[im] = i[im]; o.set_state(f0, i0)
        sampled = {int(s): k for k, s in enumerate(g.state_steps)}
        for t in range(g.T):
            for label, v in g.pokes.get(t, []):
                kind, slot = g.label_slot(label)
                (o.L.npo_set_f64 if kind == "f64" else o.L.npo_set_i32)(o._buf.ctypes.data, 0, slot, float(v) if kind == "f64" else int(v))
            if t in g.resets:
                steady, _obs, _state = g.resets[t]
                o.reset(start_at_steady_state=steady); tg._reapply_initial_conditions(o, g, steady)
            o.step(action=g.action[t], magnitude=g.magnitude[t], setpoint=g.setpoint[t], noise_z=g.noise_z[t], cw_temp=g.cooling[t])
            if t + 1 in sampled:
                fs, _is = o.state(); row = g.state[sampled[t + 1]]
                for (kind, slot, label, _p), v in zip(g.cols, row):
                    if kind != "f64" or np.isnan(v):
                        continue
                    e = abs(fs[slot] - v) / abs(v) if v != 0 else abs(fs[slot])
                    if e > worst.get(label, (0.0,))[0]:
                        worst[label] = (float(e), name, t)
    labels = [c[2] for c in SCHEMA.columns() if c[0] == "f64"]
    exact = sorted(l for l in labels if worst.get(l, (0.0,))[0] == 0.0)
    out = {"source": "tests/oracle_column_error.py over the %d trajectory fixtures" % len(fixture_names()), "fp64_columns": len(labels), "not_held_on": list(NOT_HELD_ON), "bit_identical": exact,
           "largest": {l: list(w) for l, w in sorted(worst.items(), key=lambda x: -x[1][0])[:40]}}
    json.dump(out, open(os.path.join(GOLDEN_DIR, "oracle_exact_columns.json"), "w"), indent=1)
    print("%d of %d fp64 columns bit-identical in every sample; worst other: %s" % (len(exact), len(labels), list(out["largest"].items())[:3]))


if __name__ == "__main__":
    main()
