"""State-log naming: the schema -> reference-log-column map is data generated from the reference
(oracle/ref_harness/make_state_names.py); here only its consistency with the schema is checked (no GPU)."""
import re

import numpy as np

import pytest

from nuclear_sim_amd.schema import SCHEMA
from nuclear_sim_amd.statelog import log_columns, reference_names, reference_log_columns


def test_every_mapped_label_is_a_schema_column_and_names_are_unique():
    names = reference_names()
    labels = {c[2] for c in SCHEMA.columns()}
    assert len(names) >= 150
    assert set(names) <= labels
    assert len(set(names.values())) == len(names), "two members mapped onto one log column"
    # the reference's naming: category.variable, instances spelled as in the plant (FWP-1, SG-0, ...)
    assert names["pump[0].oil_level"] == "secondary.feedwater_FWP-1.oil_level"
    assert all(re.match(r"^(primary|secondary)\.", v) for v in names.values())


def test_column_selection():
    cols = log_columns(["pump.oil_level", "sec.electrical_power_output", "pump[2].status"])
    assert [c[2] for c in cols] == ["pump[0].oil_level", "pump[1].oil_level", "pump[2].oil_level", "pump[3].oil_level",
                                    "sec.electrical_power_output", "pump[2].status"]
    assert cols[-1][3] == "npb.pump[2].status" and cols[-1][0] == "i32"
    assert len(log_columns(["tstg.blade_temperatures"])) == 14
    from nuclear_sim_amd.statelog import derived_log_columns
    members = {label for label, _f in reference_log_columns().values()}
    for need, _fn in derived_log_columns().values():
        members.update(need)
    assert len(log_columns()) == len(members)
    with pytest.raises(KeyError):
        log_columns(["pump.no_such_member"])


def test_log_column_map_is_consistent():
    """log column -> (member, factor): every member exists, the 1:1 names are a subset, and the reference's habit of logging one
    quantity under several names is kept (the feedwater system's total flow appears under its own name, the pump system's and the
    protection's; the intervention map of round 4 sorted these out: round 3's series match had tied them to the secondary side's
    copy, which is a different attribute that happens to carry the same number)"""
    lc = reference_log_columns()
    labels = {c[2] for c in SCHEMA.columns()}
    assert len(lc) >= 250 and {label for label, _f in lc.values()} <= labels
    names = reference_names()
    assert all(lc[name] == (label, 1.0) for label, name in names.items())
    same = [n for n, (label, f) in lc.items() if label == "fw.total_flow_rate"]
    assert len(same) >= 2, same
    assert all(re.match(r"^(primary|secondary)\.", n) for n in lc)


def test_result_log_columns_are_the_references_own_result_keys():
    """The map log column -> key of info["secondary_system"] on the reference's own data: the m1 run's state log
    (log_m1_oil_top_off_staggered.npz) against the result dicts recorded in the same run's trajectory fixture."""
    import os
    from golden_util import Golden, GOLDEN_DIR
    from nuclear_sim_amd.statelog import result_log_columns, derived_log_columns
    g = Golden("m1_oil_top_off_staggered")
    z = np.load(os.path.join(GOLDEN_DIR, "log_m1_oil_top_off_staggered.npz"))
    names = [str(x) for x in z["names"]]; log = z["log"]
    rc = result_log_columns()
    assert len(rc) >= 15 and set(rc) <= set(names) and not set(rc) & set(reference_log_columns()) and not set(rc) & set(derived_log_columns())
    for name, (key, factor) in rc.items():
        want = log[:, names.index(name)]
        np.testing.assert_allclose(g.sec[:, g.sec_keys.index(key)] * factor, want, rtol=1e-9, atol=1e-12, err_msg=name)
        assert np.ptp(want) > 0, name


LOGS = (("m1_oil_top_off_staggered", "composed"), ("e1_eventful_log", "composed"), ("l1_reactor_log", "default"),
        ("l2_feedwater_events_log", "composed"), ("l3_turbine_sg_events_log", "composed"))
SIDE = "secondary.feedwater_SECONDARY-COMP-001-FW.diagnostics_total_wear"


def _load_log(fixture, naming):
    """the reference's own log of a fixture's run, its columns under the rules' (composed) names"""
    import os
    from golden_util import GOLDEN_DIR
    from nuclear_sim_amd import statelog
    z = np.load(os.path.join(GOLDEN_DIR, "log_%s.npz" % fixture))
    names = [str(x) for x in z["names"]]
    return names, z["log"], (lambda rule_name: names.index(statelog.log_column_name(rule_name, naming)))


def test_every_log_column_has_one_rule_and_the_parameters_hold_in_every_log():
    """One rule per produced column; the rules together cover all 784 columns of each of the reference's five logs (under both
    provider namings); and the "parameter" columns -- what the reference sets at construction and never writes on the stepped path
    -- really hold their value in every row of every log, the eventful ones included.  Round 3 had 252 columns in that class by
    harvest; 85 are left, each with the reference line that sets it."""
    from nuclear_sim_amd import statelog
    groups = [set(reference_log_columns()), set(statelog.derived_log_columns()), set(statelog.result_log_columns()),
              set(statelog._all_diagnostic_columns()), set(statelog.diagnostic_function_columns()), set(statelog.clock_log_columns(5.0)),
              set(statelog.parameter_log_columns()), set(statelog.history_log_columns()), set(statelog.output_log_columns()), {SIDE}]
    for a in range(len(groups)):
        for b in range(a + 1, len(groups)):
            assert not groups[a] & groups[b], (a, b, sorted(groups[a] & groups[b])[:5])
    params = statelog.parameter_log_columns()
    assert len(params) <= 90 and all(isinstance(why, str) and (".py" in why) for _v, why in params.values()), "every parameter column cites where the reference sets it"
    for fx, naming in LOGS:
        names, log, col = _load_log(fx, naming)
        assert {statelog.log_column_name(n, naming) for n in set().union(*groups)} == set(names) and len(names) == 784, fx
        for name, (value, _why) in params.items():
            assert np.all(log[:, col(name)] == value), (fx, name, value, np.unique(log[:, col(name)])[:4])
    # none of the columns the round-3 review named as plant state is a literal any more
    for name in ("primary.reactor.neutronics_neutron_flux", "primary.reactor.control_control_rod_position", "primary.reactor.safety_scram_status",
                 "primary.reactor.scram_activated", "secondary.feedwater_FWP-3.trip_active", "secondary.feedwater_FWP-1.cavitation_damage",
                 "secondary.feedwater_FWP-2.impeller_replacement_occurred", "secondary.steam_generator_SG-0.tsp_shutdown_required",
                 "secondary.steam_generator_SG-0.tsp_fouling_stage_numeric", "secondary.turbine_SECONDARY-COMP-001-TURB.rotor_speed",
                 "secondary.turbine_SECONDARY-COMP-001-TURB.vibration_velocity_x", "secondary.condenser.SJE-002_operating",
                 "secondary.feedwater_SECONDARY-COMP-001-FW.protection_system_trip_active", "secondary.ph_control.ph_control_morpholine_level"):
        assert name not in params, name


def test_the_member_map_is_the_intervention_map():
    """reference_log_columns() is state_names.json "poked" (oracle/ref_harness/make_log_map.py: every member poked on the live
    reference, the providers read back) plus the by-construction twins; and every derived rule reads at least the members the
    intervention saw the column move with."""
    import json
    from nuclear_sim_amd import statelog
    poked = json.load(open(statelog._NAMES_PATH))["poked"]
    lc = reference_log_columns()
    for name, hits in poked["log_columns"].items():
        assert lc[name] == (hits[0][0], float(hits[0][1])), name
    assert set(lc) == set(poked["log_columns"]) | set(statelog._aliases())
    derived = statelog.derived_log_columns()
    for name, members in poked["depends"].items():
        if name in statelog.result_log_columns():      # (the secondary side's heat transfer in MW: a key of the step's result dict, carried in fp64)
            continue
        assert name in derived, (name, members)
        need = set(derived[name][0])
        assert set(members) <= need, (name, sorted(set(members) - need))


@pytest.mark.parametrize("fixture,naming", LOGS)
def test_member_derived_and_clock_columns_on_the_oracle_replay(oracle_lib, fixture, naming):
    """All five reference logs replayed on the CPU oracle: every log column that is a state member, a function of end-of-step
    state, a step counter or the step's own output, evaluated on the oracle after every step, against the reference's log row."""
    from golden_util import Golden
    from nuclear_sim_amd import statelog
    import test_oracle_golden as tg
    g = Golden(fixture)
    names, log, col = _load_log(fixture, naming)
    o = oracle_lib.OraclePlants(1, tg._configure(oracle_lib, g))
    f0, i0 = o.state()
    f, i, fm, im = g.split_state(g.state[0])
    f0[fm] = f[fm]; i0[im] = i[im]
    o.set_state(f0, i0)
    slot = {label: (kind, s) for kind, s, label, _p in SCHEMA.columns()}
    rules = dict(statelog.derived_log_columns()); rules.update(statelog.clock_log_columns(g.meta.get("dt", 1.0)))
    for name, (label, factor) in reference_log_columns().items():
        rules[name] = ((label,), lambda v, factor=factor: v * factor)
    assert len(rules) >= 450
    moved = set()
    for t in range(g.T):
        for label, v in g.pokes.get(t, []):
            kind, s = g.label_slot(label)
            (o.L.npo_set_f64 if kind == "f64" else o.L.npo_set_i32)(o._buf.ctypes.data, 0, s, float(v) if kind == "f64" else int(v))
        _obs, _rew, done, _flags, _info = o.step(action=g.action[t], magnitude=g.magnitude[t], setpoint=g.setpoint[t], noise_z=g.noise_z[t], cw_temp=g.cooling[t])
        fs, is_ = o.state()
        for name, (need, fn) in rules.items():
            args = [np.atleast_1d(np.float64(fs[slot[l][1]] if slot[l][0] == "f64" else is_[slot[l][1]])) for l in need]
            want = log[t, col(name)]
            got = float(np.asarray(fn(*args)).reshape(-1)[0])
            assert abs(got - want) <= 1e-6 * abs(want) + 1e-9, (name, t, got, want)
            if want != log[0, col(name)]:
                moved.add(name)
        for name in statelog.output_log_columns():
            assert float(done[0]) == log[t, col(name)], (name, t)



def test_derived_log_columns_on_the_oracle_replay(oracle_lib):
    """The columns that are plain functions of end-of-step state (pump factors, SG averages, TSP aggregates, the turbine
    stages' efficiency and blade condition, the steam generators' flow capacities, restriction factors, pump power and heat flux ...): the m1 run replayed on the CPU oracle, each formula evaluated on the oracle's
    state after every step, against the reference's own log."""
    import os
    from golden_util import Golden, GOLDEN_DIR
    from nuclear_sim_amd.statelog import derived_log_columns
    import test_oracle_golden as tg
    g = Golden("m1_oil_top_off_staggered")
    z = np.load(os.path.join(GOLDEN_DIR, "log_m1_oil_top_off_staggered.npz"))
    names = [str(x) for x in z["names"]]; log = z["log"]
    o = oracle_lib.OraclePlants(1, tg._configure(oracle_lib, g))
    f0, i0 = o.state()
    f, i, fm, im = g.split_state(g.state[0])
    f0[fm] = f[fm]; i0[im] = i[im]
    o.set_state(f0, i0)
    slot = {label: (kind, s) for kind, s, label, _p in SCHEMA.columns()}
    derived = derived_log_columns()
    assert len(derived) >= 80 and set(derived) <= set(names)
    for t in range(g.T):
        o.step(action=g.action[t], magnitude=g.magnitude[t], setpoint=g.setpoint[t], noise_z=g.noise_z[t], cw_temp=g.cooling[t])
        fs, is_ = o.state()
        for name, (need, fn) in derived.items():
            args = [np.float64(fs[slot[l][1]] if slot[l][0] == "f64" else is_[slot[l][1]]) for l in need]
            want = log[t, names.index(name)]
            assert abs(float(fn(*args)) - want) <= 1e-6 * abs(want) + 1e-9, (name, t, float(fn(*args)), want)


@pytest.mark.parametrize("fixture,naming", LOGS)
def test_history_columns_on_the_references_own_logs(fixture, naming):
    """The windowed columns are functions of other LOG columns' history: evaluated on the reference's own series, against the
    reference's own column; most logs are shorter than the pH controller's 100-step window, so its far edge is checked on a
    long random series against the reference's list arithmetic written out (ph_control_system.py:441-455)."""
    from nuclear_sim_amd import statelog
    names, log, col = _load_log(fixture, naming)
    for name, (sources, fn) in statelog.history_log_columns().items():
        got = fn(*[np.stack([log[:, col(src)]] * 2, axis=1) for src in sources])
        np.testing.assert_allclose(got[:, 0], log[:, col(name)], rtol=1e-9, atol=1e-12, err_msg=name)
        np.testing.assert_array_equal(got[:, 0], got[:, 1])
    rng = np.random.default_rng(5)
    series = rng.normal(0, 0.01, (260, 3))
    got = statelog.history_log_columns()["secondary.ph_control.ph_control_deviation_rms"][1](series)
    for lane in range(3):
        history = []
        for t in range(series.shape[0]):
            history.append(abs(series[t, lane]))
            if len(history) > 100:
                history.pop(0)
            assert abs(got[t, lane] - np.sqrt(np.mean(np.square(history)))) <= 1e-12
