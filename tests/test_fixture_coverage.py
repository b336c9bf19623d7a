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


# real-valued physics members that rest at one or two values in EVERY fixture, with the reason (a member that sits on a clamp or on its
# construction value hides whatever computes it: the rotor speed did, 3600 / 3780 rpm in 106 fixtures, until fixture c20 -- found by
# tools/mutate_oracle.py's sample of the turbine's text)
RESTING_BY_CONSTRUCTION = {
    "fw.timer_motor_temp": "the motor temperature it watches (> 130 C) is an output every pump recomputes each step, 60-90 C at the reference's constants",
    "fw.timer_vibration": "the vibration level it watches (> 10 mm/s) is an output every pump recomputes each step, below 6 mm/s with every wear at its trip value",
    "pump[0].head_degradation": "performance factors are computed with a literal cavitation damage of 0.0 (pump_lubrication.py:742): the head loss is 0",
    "pump[1].head_degradation": "as pump 0", "pump[2].head_degradation": "as pump 0", "pump[3].head_degradation": "as pump 0",
    "pump[3].cavitation_time": "the spare pump never runs long enough to cavitate; pumps 0-2 cover the member",
    "sec.previous_feedwater_temp": "a moving average of the literal 227 C the feedwater temperature estimate is",
    "turb.timer_bearing_temp": "bearing metal temperatures are capped at 115 C, the timer starts above 120 C (test_state_machines_are_visited_completely)",
    "turb.timer_overspeed": "the rotor model clamps the speed AT the trip value (> 3780 never holds)",
    "chem[0].dissolved_oxygen": "assigned a literal every step (water_chemistry.py:352)", "chem[1].dissolved_oxygen": "as chem[0]",
    "prim.burnable_poison_worth": "rests at 0 unless poked (fixture c11): no step changes it",
    "prim.coolant_void_fraction": "rests at 0 unless poked (fixture c11): no step changes it",
}


def test_every_real_valued_physics_member_moves_in_some_fixture():
    """fp64 members outside the maintenance bookkeeping must take at least three values across tests/golden/ (timers: two)"""
    seen = {}
    for name in fixture_names():
        g = Golden(name)
        for j, (kind, _slot, label, _p) in enumerate(g.cols):
            if kind != "f64" or label.split(".")[0].split("[")[0] in ("maint", "mpump") or len(seen.get(label, ())) > 8:
                continue
            v = g.state[:, j]
            seen.setdefault(label, set()).update(np.unique(v[~np.isnan(v)]).tolist()[:16])
    # (a protection timer that a step of dt raises to dt and the next step clears has two values when it has run)
    resting = sorted(l for l, s in seen.items() if len(s) < (2 if "timer" in l else 3) and l not in RESTING_BY_CONSTRUCTION)
    assert not resting, "real-valued members that rest in every fixture: %s" % resting
