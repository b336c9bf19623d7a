"""CPU: what the reference-generated fixtures actually visit.  Every int32 column of the schema (flags, status codes,
counters, trip reasons) must take more than one value somewhere in tests/golden/ -- a flag that is constant in every
fixture is a branch the reference never pinned -- and the value sets of the state machines must be complete.
`python tests/test_fixture_coverage.py` prints the table."""
import numpy as np

from golden_util import Golden, fixture_names


def coverage():
    seen = {}
    for name in fixture_names():
        g = Golden(name)
        rows = [g.state] + [r[2][None, :] for r in g.resets.values()]
        st = np.concatenate(rows, axis=0)
        for j, (kind, _slot, label, _p) in enumerate(g.cols):
            if kind != "i32":
                continue
            v = st[:, j]
            v = v[~np.isnan(v)]
            d = seen.setdefault(label, {})
            for x in np.unique(v):
                d.setdefault(int(x), set()).add(name)
    return seen


# int32 members that cannot vary, with the reason
CONSTANT_BY_CONSTRUCTION = {
    "ph.controller_enabled": "set by operator action only; no step changes it",
    "ph.ammonia_supply_available": "changes only through the controller's random equipment failures, which use the unseeded global RNG "
                                   "(ph_control_system.py:409-420) and are neutralised in the harness (SURVEY 8c)",
    "ph.morpholine_supply_available": "as above",
}


def test_every_flag_and_code_varies_in_some_fixture():
    seen = coverage()
    constant = sorted(l for l, d in seen.items() if len(d) < 2 and l not in CONSTANT_BY_CONSTRUCTION)
    assert not constant, "int32 columns constant in every fixture: %s" % constant


def test_state_machines_are_visited_completely():
    seen = coverage()
    status = set().union(*[set(seen["pump[%d].status" % k]) for k in range(4)])
    assert status == {0, 1, 2, 3, 4}, status                      # RUNNING STOPPED STARTING STOPPING TRIPPED
    reasons = set().union(*[set(seen["pump[%d].trip_reason" % k]) for k in range(4)])
    # NPSH, low suction, SG high level, severe cavitation, cavitation damage, critical NPSH, lubrication: very low / low oil,
    # component wear, combined wear.  Unreachable by construction (make_golden.py, C6): 1 low flow, 4 high discharge,
    # 12 overfill, 14 seal leakage
    assert {2, 3, 5, 6, 7, 8, 10, 11, 13, 15} <= reasons, reasons
    masks = set(seen["turb.trip_latched_mask"])
    # vibration, thermal expansion, thermal stress.  Unreachable by construction (make_golden.py, C7): 1 overspeed (the rotor
    # model clamps the speed at the trip value), 4 bearing metal temperature (at most 115 C against 120 C), 16 low vacuum (the
    # protection is handed the literal 0.007 MPa)
    assert any(m & 2 for m in masks) and any(m & 8 for m in masks) and any(m & 32 for m in masks), masks
    assert {0, 1} <= set(seen["fw.npsh_low_low_trip_active"]) and {0, 1} <= set(seen["sg[0].tsp_shutdown_required"])
    assert {0, 1} <= set(seen["prim.scram_status"]) and {0, 1} <= set(seen["fw.system_trip_active"])
    assert len(seen["cond.lead_ejector"]) >= 2 and len(seen["cond.ej_operating_mask"]) >= 2


if __name__ == "__main__":
    for label, d in sorted(coverage().items()):
        print("%-40s %s" % (label, {k: len(v) for k, v in sorted(d.items())}))
