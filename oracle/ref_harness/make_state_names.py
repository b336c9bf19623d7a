"""Map schema columns onto the column names of the reference's state log (StateManager DataFrame).

Harness-only (imports the reference from /root/reference).  Runs the data-gen runner's simulator (state management
on, as MaintenanceScenarioRunner builds it) for a few dozen steps with a varying load, reads the reference's own
log (`sim.state_manager.data`, one row per step, names `category.variable` built by state_manager.py:169-182 from
the auto-registered providers, auto_register.py:83-165) and, beside it, every schema column's reference attribute
(the quoted path of include/npb_fields.h).  A schema column is mapped to the log column whose whole series equals
its own; ambiguous or missing matches are left out.  Output: nuclear_sim_amd/state_names.json
(python -m oracle.ref_harness.make_state_names).
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from nuclear_sim_amd.schema import SCHEMA  # noqa: E402
from oracle.ref_harness import refsim  # noqa: E402
from oracle.ref_harness.trace import _val  # noqa: E402


def main(steps=60):
    refsim.setup()
    cols = SCHEMA.columns()
    with refsim.quiet():
        runner, sim = refsim.make_runner_sim(action="oil_top_off", duration_hours=steps * 5.0 / 60.0)
        profile = runner._generate_power_profile(steps)
        mine = np.full((steps, len(cols)), np.nan)
        results = []
        for t in range(steps):
            runner._set_target_power(float(np.clip(profile[t] - 8.0 + 8.0 * np.sin(t / 3.0), 50.0, 100.0)))   # a load that moves
            results.append(sim.step()["info"]["secondary_system"])
            for j, (_kind, _slot, _label, path) in enumerate(cols):
                mine[t, j] = _val(sim, path) if path and not path.startswith("=") else np.nan
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
    out, ambiguous = {}, 0
    for j, (kind, slot, label, path) in enumerate(cols):
        series = mine[:, j]
        if np.isnan(series).any():
            continue
        hits = [n for n, v in log.items() if np.array_equal(v, series)]
        if len(hits) > 1:
            # identical twins (the three duty pumps, the three SGs, constants): keep the names of this instance,
            # then the name that carries the member's own name
            sec, inst = (label.split("[")[0], int(label.split("[")[1].split("]")[0])) if label.split(".")[0].endswith("]") else (label.split(".")[0], None)
            token = {"pump": "FWP-%d", "sg": "SG-%d"}.get(sec)
            if token is not None and inst is not None:
                narrowed = [n for n in hits if (token % (inst + 1)) in n]
                hits = narrowed or hits
            base = label.split(".")[-1].split("[")[0]
            narrowed = [n for n in hits if n.split(".")[-1] == base] or [n for n in hits if base in n.split(".")[-1]]
            hits = narrowed if len(narrowed) == 1 else hits
        if len(hits) == 1:
            out[label] = hits[0]
        elif len(hits) > 1:
            ambiguous += 1
            if os.environ.get("NPB_NAMES_VERBOSE"):
                print("ambiguous:", label, hits[:6])
    # one member per log column: when two members share a series (speed_percent / speed_setpoint at steady speed) the one
    # whose own name ends the log name keeps it
    taken = {}
    for label, name in list(out.items()):
        base = label.split(".")[-1].split("[")[0]
        if name in taken:
            other = taken[name]
            if name.split(".")[-1] == base:
                del out[other]; taken[name] = label
            else:
                del out[label]
        else:
            taken[name] = label
    # the scalars of step()'s info["secondary_system"] dictionary (secondary/__init__.py:930-1010) that are state members
    result_keys = {}
    numeric = [k for k, v in results[0].items() if isinstance(v, (int, float, bool))]
    for k in numeric:
        series = np.array([float(r[k]) for r in results])
        hits = [cols[j][2] for j in range(len(cols)) if not np.isnan(mine[:, j]).any() and np.array_equal(mine[:, j], series)]
        if len(hits) > 1:   # twins: prefer the section the key names, then the member whose name ends the key
            pref = {"condenser": "cond.", "turbine": "turb.", "feedwater": "fw.", "sg": "sec.sg_", "water_chemistry": "chem[0].", "ph_control": "ph."}
            for word, sec_prefix in pref.items():
                if k.startswith(word):
                    hits = [h for h in hits if h.startswith(sec_prefix)] or hits
            narrowed = [h for h in hits if k.endswith(h.split(".")[-1].split("[")[0])] or [h for h in hits if h.startswith("sec.")]
            hits = narrowed if narrowed else hits
        if hits and np.ptp(series) > 0:      # a constant series proves nothing
            result_keys[k] = hits[0]
    path = os.path.join(ROOT, "nuclear_sim_amd", "state_names.json")
    with open(path, "w") as fh:
        json.dump({"source": "reference StateManager log, oil_top_off action-test run, %d steps" % steps,
                   "log_columns": len(df.columns) - 1, "names": out,
                   "secondary_result_keys": result_keys, "secondary_result_numeric_keys": len(numeric)}, fh, indent=1, sort_keys=True)
    print("%d of %d numeric keys of step()'s secondary_system result are state members" % (len(result_keys), len(numeric)))
    print("%d of %d schema columns mapped onto the reference's %d log columns (%d ambiguous left out) -> %s"
          % (len(out), len(cols), len(df.columns) - 1, ambiguous, path))


if __name__ == "__main__":
    main()
