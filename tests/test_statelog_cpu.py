"""State-log naming: the schema -> reference-log-column map is data generated from the reference
(oracle/ref_harness/make_state_names.py); here only its consistency with the schema is checked (no GPU)."""
import re

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
    quantity under several names is kept (the secondary side's total feedwater flow appears twice)"""
    lc = reference_log_columns()
    labels = {c[2] for c in SCHEMA.columns()}
    assert len(lc) >= 250 and {label for label, _f in lc.values()} <= labels
    names = reference_names()
    assert all(lc[name] == (label, 1.0) for label, name in names.items())
    same = [n for n, (label, f) in lc.items() if label == "sec.total_feedwater_flow"]
    assert len(same) >= 2
    assert all(re.match(r"^(primary|secondary)\.", n) for n in lc)
