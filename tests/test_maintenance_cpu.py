"""CPU: the maintenance vocabulary of include/npb_maint.h against the reference's own configuration
(tests/golden/maint_table.json, dumped from the live StateManager by oracle/ref_harness/make_golden.py maint_table)."""
import json
import os

from golden_util import GOLDEN_DIR


def _ref():
    with open(os.path.join(GOLDEN_DIR, "maint_table.json")) as fh:
        return json.load(fh)


def test_parameter_catalog_is_exactly_what_resolves_in_a_pumps_state_log():
    from nuclear_sim_amd import _lib
    ref = _ref()
    resolving = [t["name"] for t in ref["thresholds"] if t["resolves_to"]]
    assert sorted(resolving) == sorted(_lib.MAINT_PARAMS)
    assert len(ref["thresholds"]) - len(resolving) == 28      # named by the configuration, never fire


def test_default_table_is_the_action_test_configuration(oracle_lib):
    """npb_maint_table_default() (C) == the reference's thresholds dict pushed through the host-side converter"""
    import ctypes
    from nuclear_sim_amd import _lib
    ref = _ref()
    want = _lib.maint_table_from_thresholds({t["name"]: t for t in ref["thresholds"]})
    got = _lib.NpbMaintTable()
    L = oracle_lib.lib()
    assert L.npo_maint_table_size() == ctypes.sizeof(got)
    L.npo_default_maint_table(ctypes.byref(got))
    order_want = sorted(range(_lib.MAINT_NPARAM), key=lambda k: want.rank[k])
    order_got = sorted(range(_lib.MAINT_NPARAM), key=lambda k: got.rank[k])
    assert order_want == order_got                              # same scan order
    for k in range(_lib.MAINT_NPARAM):
        assert got.rank[k] >= 0 and want.rank[k] >= 0
        for f in ("threshold", "cooldown_hours", "comparison", "action", "priority", "bearing"):
            assert getattr(got, f)[k] == getattr(want, f)[k], (_lib.MAINT_PARAMS[k], f)


def test_action_catalog_against_the_references_action_types():
    from nuclear_sim_amd import _lib
    valid = set(_ref()["valid_action_types"])
    not_types = [a for a in _lib.MAINT_ACTIONS if a not in valid]
    assert not_types == ["system_cleaning"]                     # NPB_MAINT_ACTION_IS_TYPE


def test_settings_of_the_action_test_maintenance_system():
    from nuclear_sim_amd import _lib
    s = _ref()["settings"]
    p = None
    from nuclear_sim_amd.schema import PARAMS
    d = {n: v for n, v, _ in PARAMS}
    assert d["maint_check_interval_hours"] == s["check_interval_hours"] and d["maint_work_order_cooldown"] == s["work_order_cooldown_hours"]
    assert d["maint_start_delay_hours"] == s["high_priority_delay_hours"] and d["maint_medium_delay_hours"] == s["medium_priority_delay_hours"]
    assert d["maint_low_delay_hours"] == s["low_priority_delay_hours"] and d["maint_emergency_delay_hours"] == s["emergency_delay_hours"]
