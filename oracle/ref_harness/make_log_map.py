"""Which state member does a column of the reference's state log SHOW?  Answered by intervention, not by coincidence.

Harness-only (imports the reference from /root/reference).  Round 1-3 matched whole series: a log column was mapped to the
member whose series over three runs it equalled bit for bit.  That finds the right member when both move, and a wrong one
whenever two different quantities happen to carry the same numbers in those runs (an ejector's operating hours and the
vacuum controller's rotation timer both count the run's hours -- until the ejectors rotate), and nothing at all for a
quantity that rests.  Here every assignable member of the schema is POKED on the live reference object, one at a time, to two
different values, and the state manager's providers are asked for their row each time (the loop of
StateManager.collect_states, simulator/state/state_manager.py:152-233, without appending to the log):

    column == factor * poked value, both times      ->  the column shows that member            ("log_columns")
    column moves with the member, but is not it     ->  the column is a function of the member  ("depends", for the review of
                                                        nuclear_sim_amd/statelog.py's derived columns)
    no member moves it                              ->  a value from inside the step, or a configuration value ("untouched")

Written to nuclear_sim_amd/state_names.json beside the series map ("poked": ...); tests/test_statelog_cpu.py holds the
shipped column rules to it.   python -m oracle.ref_harness.make_log_map
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from nuclear_sim_amd.schema import SCHEMA  # noqa: E402
from oracle.ref_harness import refsim, trace  # noqa: E402

FACTORS = (1.0, 100.0, 0.01, 1000.0, 0.001, 1e6, 1e-6, 60.0, 1.0 / 60.0)


def collect(sim):
    """one row of the state log as StateManager.collect_states would build it, numeric columns only"""
    row = {}
    for provider, category in sim.state_manager.providers:
        if hasattr(provider, "get_current_state") and callable(provider.get_current_state):
            row.update(provider.get_current_state())
        elif hasattr(provider, "get_state_dict") and callable(provider.get_state_dict):
            for k, v in provider.get_state_dict().items():
                row["%s.%s" % (category, k)] = v
    out = {}
    for k, v in row.items():
        try:
            out[k] = float(v)
        except (TypeError, ValueError):
            pass
    return out


def poke(sim, path, v):
    """trace._poke, plus the schema's "=list(<dict>.values())[k]" paths (assigned through the dict's k-th key)"""
    import re
    m = re.match(r"^=list\((.+)\.values\(\)\)\[(\d+)\]$", path)
    if m:
        from oracle.ref_harness.leaves import H
        d = eval(m.group(1), {"root": sim, "H": H})
        d[list(d.keys())[int(m.group(2))]] = v
    else:
        trace._poke(sim, path, v)


def same(a, b):
    return a == b or (a != a and b != b)


def main(steps=6):
    refsim.setup()
    cols = SCHEMA.columns()
    with refsim.quiet():
        runner, sim = refsim.make_runner_sim(action="oil_top_off", duration_hours=2.0)
        profile = runner._generate_power_profile(steps)
        for t in range(steps):
            runner._set_target_power(profile[t])
            sim.step()
    direct, depends = {}, {}
    with refsim.quiet():
        row0 = collect(sim)
        row0b = collect(sim)   # the lubrication systems clear their maintenance flags when read: the second read is the baseline
        assert all(same(row0b[k], row0[k]) for k in row0b if "occurred" not in k)
        row0 = row0b
        for kind, _slot, label, path in cols:
            if not path or label.startswith(("mpump.", "maint.")):
                continue
            old = trace._val(sim, path)
            if old != old:
                continue
            obj_old = None
            try:
                from oracle.ref_harness.leaves import resolve
                obj_old = resolve(sim, path)
            except Exception:
                continue
            rows, vals = [], []
            try:
                for trial in (0, 1):
                    if isinstance(obj_old, bool) or (kind == "i32" and old in (0.0, 1.0) and not label.endswith(("status", "mask", "count", "ejector", "reason"))):
                        v = (not bool(old)) if trial == 0 else bool(old)
                        if trial == 1:
                            continue
                    elif kind == "i32":
                        continue            # enums, masks, counters: by name (statelog.py), not by poke
                    else:
                        v = old * (1.37 if trial == 0 else 0.81) + (0.0123 if trial == 0 else -0.0456)
                    poke(sim, path, v)
                    if trace._val(sim, path) != float(v):
                        raise ValueError("the poke did not take: %s" % path)
                    rows.append(collect(sim)); vals.append(float(v))
            except Exception:
                rows = []
            finally:
                try:
                    poke(sim, path, obj_old)
                except Exception:
                    pass
            if not rows:
                continue
            for name in row0:
                moved = [not same(r[name], row0[name]) for r in rows]
                if not any(moved):
                    continue
                hit = None
                for f in FACTORS:
                    if all(r[name] == v * f for r, v in zip(rows, vals)):
                        hit = f
                        break
                if hit is not None:
                    direct.setdefault(name, []).append([label, hit])
                else:
                    depends.setdefault(name, []).append(label)
        after = collect(sim)
        assert all(same(after[k], row0[k]) for k in row0), [k for k in row0 if not same(after[k], row0[k])][:5]
    untouched = sorted(n for n in row0 if n not in direct and n not in depends)
    path = os.path.join(ROOT, "nuclear_sim_amd", "state_names.json")
    d = json.load(open(path))
    d["poked"] = {"source": "oracle/ref_harness/make_log_map.py: every assignable schema member poked on the live reference (the data-gen "
                            "runner's plant after %d steps), the state manager's providers read each time" % steps,
                  "log_columns": {k: v for k, v in sorted(direct.items())}, "depends": {k: sorted(v) for k, v in sorted(depends.items())},
                  "untouched": untouched}
    with open(path, "w") as fh:
        json.dump(d, fh, indent=1, sort_keys=True)
    print("%d log columns; %d show a member, %d more move with members, %d moved by none" % (len(row0), len(direct), len(depends), len(untouched)))
    amb = {k: v for k, v in direct.items() if len(v) > 1}
    print("%d columns shown by more than one member (copies the reference keeps):" % len(amb))
    for k, v in sorted(amb.items()):
        print("   ", k, v)


if __name__ == "__main__":
    main()
