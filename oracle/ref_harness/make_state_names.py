"""Map the column names of the reference's state log (StateManager DataFrame) onto schema members.

Harness-only (imports the reference from /root/reference).  Runs the data-gen runner's simulator (state management
on, as MaintenanceScenarioRunner builds it) through three eventful runs -- a moving load; the same with a pump trip,
a cooling-water swing and a second pump's NPSH collapse; a degraded plant (seal_replacement scenario, seed 3) -- and
reads the reference's own log (`sim.state_manager.data`, one row per step, names `category.variable` built by
state_manager.py:169-182 from the auto-registered providers, auto_register.py:83-165) and, beside it, every schema
member's reference attribute (the quoted path of include/npb_fields.h).

A LOG COLUMN is mapped to a member when the member's whole series over the three runs equals the column's
(bit for bit), or equals it after one of a few unit factors; a column that never varies proves nothing and is left
out.  Several log columns may map to one member (the reference logs `total_feedwater_flow` three times), and when
several members carry the same series (the stored previous-step copies, the chemistry twins) the one whose label is
closest to the column's name is taken.  Output: nuclear_sim_amd/state_names.json
  names        member label -> ONE log column (the 1:1 map of round 1, kept for callers that select by member)
  log_columns  log column -> [member label, factor]   (everything the state log can reproduce)
and tests/golden/log_m1_oil_top_off_staggered.npz: the reference's log of the m1 fixture's run, the ground truth of the GPU test.
(python -m oracle.ref_harness.make_state_names)
"""
import difflib
import json
import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from nuclear_sim_amd.schema import SCHEMA  # noqa: E402
from oracle.ref_harness import refsim  # noqa: E402
from oracle.ref_harness.trace import _val  # noqa: E402

FACTORS = (1.0, 100.0, 0.01, 1000.0, 0.001, 10.0, 0.1, 60.0, 1.0 / 60.0, 3600.0, 1.0 / 3600.0, 1e6, 1e-6)
P1 = "secondary_physics.feedwater_system.pump_system.pumps['FWP-%d']"


def _run(action, seed, steps, events):
    cols = SCHEMA.columns()
    with refsim.quiet():
        runner, sim = refsim.make_runner_sim(action=action, duration_hours=steps * 5.0 / 60.0, randomization_seed=seed)
        profile = runner._generate_power_profile(steps)
        mine = np.full((steps, len(cols)), np.nan)
        for t in range(steps):
            for path, v in events.get(t, []):
                exec("sim.%s = v" % path, {"sim": sim, "v": v})
            runner._set_target_power(float(np.clip(profile[t] - 8.0 + 8.0 * np.sin(t / 3.0), 50.0, 100.0)))   # a load that moves
            kw = {"cooling_water_temp": 25.0 + 6.0 * np.sin(t / 5.0)} if events else {}
            sim.step(**kw)
            for j, (_kind, _slot, _label, path) in enumerate(cols):
                mine[t, j] = _val(sim, path) if path else np.nan
    df = sim.state_manager.data
    assert len(df) == steps, (len(df), steps)
    log = {}
    for name in df.columns:
        if name == "time":
            continue
        try:
            log[name] = df[name].astype(float).to_numpy()
        except (TypeError, ValueError):
            continue
    return mine, log


def _instance_token(label):
    m = re.match(r"^(\w+)\[(\d+)\]\.", label)
    if not m:
        return None
    sec, inst = m.group(1), int(m.group(2))
    return {"pump": "FWP-%d" % (inst + 1), "mpump": "FWP-%d" % (inst + 1), "sg": "SG-%d" % inst}.get(sec)


def _closeness(label, name):
    base = label.split(".")[-1]
    tail = name.split(".")[-1]
    score = difflib.SequenceMatcher(None, re.sub(r"\[\d+\]", "", base), tail).ratio()
    tok = _instance_token(label)
    if tok is not None:
        score += 1.0 if tok in name else -1.0
    elif re.search(r"(FWP|SG)-\d", name):
        score -= 0.5
    k = re.search(r"\[(\d+)\]$", base)          # array element: the element number usually shows in the column name
    if k and re.search(r"(^|[^0-9])0*%d([^0-9]|$)" % (int(k.group(1)) + 1), tail):
        score += 0.3
    return score


def main():
    refsim.setup()
    cols = SCHEMA.columns()
    runs = [_run("oil_top_off", None, 60, {}),
            _run("oil_top_off", None, 70, {20: [((P1 % 1) + ".lubrication_system.oil_level", 9.0)],
                                            40: [((P1 % 2) + ".state.npsh_available", 11.0)]}),
            _run("seal_replacement", 3, 40, {})]
    mine = np.concatenate([r[0] for r in runs], axis=0)
    names_all = [n for n in runs[0][1] if all(n in r[1] for r in runs)]
    log = {n: np.concatenate([r[1][n] for r in runs]) for n in names_all}
    usable = [j for j in range(len(cols)) if not np.isnan(mine[:, j]).any()]
    log_columns, constant, unmatched = {}, 0, []
    for name, series in log.items():
        if np.isnan(series).any():
            unmatched.append(name); continue
        if np.ptp(series) == 0.0:
            constant += 1; continue
        best = None
        for f in FACTORS:
            hits = [j for j in usable if np.array_equal(mine[:, j] * f if f != 1.0 else mine[:, j], series)]
            if hits:
                j = max(hits, key=lambda j: _closeness(cols[j][2], name))
                best = (cols[j][2], f)
                break
        if best is None:
            unmatched.append(name)
        else:
            log_columns[name] = [best[0], best[1]]
    # the 1:1 map (member -> one column): of the columns that map to a member with factor 1, the closest name
    one = {}
    for name, (label, f) in log_columns.items():
        if f != 1.0:
            continue
        if label not in one or _closeness(label, name) > _closeness(label, one[label]):
            one[label] = name
    path = os.path.join(ROOT, "nuclear_sim_amd", "state_names.json")
    old = json.load(open(path)) if os.path.exists(path) else {}
    with open(path, "w") as fh:
        json.dump({"source": "reference StateManager log: three action-test runs (moving load; pump trip + NPSH collapse + cooling-water swing; "
                             "seal_replacement seed 3), %d steps in all" % len(mine),
                   "reference_log_columns": len(runs[0][1]), "constant_in_all_runs": constant, "unmatched": sorted(unmatched),
                   "names": one, "log_columns": log_columns,
                   "secondary_result_keys": old.get("secondary_result_keys", {}),
                   "secondary_result_numeric_keys": old.get("secondary_result_numeric_keys", 0)}, fh, indent=1, sort_keys=True)
    print("%d of the reference's %d numeric log columns mapped onto %d members (%d constant in every run, %d without a member) -> %s"
          % (len(log_columns), len(log), len(set(v[0] for v in log_columns.values())), constant, len(unmatched), path))
    make_m1_log()


def make_m1_log(fixture="m1_oil_top_off_staggered"):
    """tests/golden/log_<fixture>.npz: the reference's own log (sim.state_manager.data) for the run of a trajectory fixture"""
    from oracle.ref_harness import make_golden, trace
    sc = [s for s in make_golden.scenarios() if s["name"] == fixture][0]
    ref, sim = trace.run_reference(dict(sc), SCHEMA.columns())
    df = sim.state_manager.data
    keep, data = [], []
    for name in df.columns:
        if name == "time":
            continue
        try:
            v = df[name].astype(float).to_numpy()
        except (TypeError, ValueError):
            continue
        keep.append(name); data.append(v)
    out = os.path.join(ROOT, "tests", "golden", "log_%s.npz" % fixture)
    np.savez_compressed(out, names=np.array(keep), log=np.array(data).T)
    print("log of %s: %d rows x %d columns -> %s" % (fixture, len(df), len(keep), out))


def add_constants(fixtures=("m1_oil_top_off_staggered", "e1_eventful_log")):
    """state_names.json "constants": the reference's log columns that hold ONE value in every row of every reference log under
    tests/golden/ (a quiet run and an eventful one: pump trip, NPSH collapse, load and cooling-water swings, worn components, a
    fouled steam generator) -- configuration values, flags at rest, the idle spare's readings -- with that value.  A column that
    moves in any of the logs is not a constant and has to be produced (nuclear_sim_amd/statelog.py)."""
    logs = []
    for fx in fixtures:
        z = np.load(os.path.join(ROOT, "tests", "golden", "log_%s.npz" % fx))
        logs.append(({str(n): j for j, n in enumerate(z["names"])}, z["log"]))
    path = os.path.join(ROOT, "nuclear_sim_amd", "state_names.json")
    d = json.load(open(path))
    const = {}
    for name, j in logs[0][0].items():
        cols = [lg[:, idx[name]] for idx, lg in logs if name in idx]
        v = cols[0][0]
        if len(cols) == len(logs) and all(np.all(c == v) for c in cols):
            const[name] = float(v)
    d["constants"] = dict(sorted(const.items()))
    d["constants_source"] = "constant, with the same value, in every row of %s" % ", ".join("log_%s.npz" % f for f in fixtures)
    with open(path, "w") as fh:
        json.dump(d, fh, indent=1, sort_keys=True)
    print("%d constant columns -> %s" % (len(const), path))


if __name__ == "__main__":
    if sys.argv[1:] == ["constants"]:
        add_constants()
    elif len(sys.argv) > 2 and sys.argv[1] == "log":
        refsim.setup()
        for fx in sys.argv[2:]:
            make_m1_log(fx)
    else:
        main()
