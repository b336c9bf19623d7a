"""Columnar state log of a batch.

The reference's StateManager appends one pandas row per plant-step -- `collect_states` walks every registered
provider's `get_state_dict()` and `pd.concat`s the row (simulator/state/state_manager.py:152-233), 790 columns named
`category.variable` (auto_register.py:83-165); with physics on a GPU that bookkeeping would be the whole run time.
Here a sample is one gather kernel (`npb_gather_fields`): the chosen members of every plant, widened to double, land
in a device buffer `[sample, field, plant]`; nothing touches the host until `table()` / `write_parquet()`.

Columns carry the reference's own log names, all 784 of its numeric log columns (checked at every step against the reference's
own log of a quiet and of an eventful run, tests/test_gpu_parity.py):
  * `state_names.json` (made by running the reference through three eventful runs and matching whole series value for value; the
    generator script is named in DESIGN.md section 6) maps 276 log columns onto state members -- several log columns can show one
    member (the reference logs the total feedwater flow three times), a few through a unit factor, the idle spare pump by analogy;
  * `derived_log_columns()`: 99 columns that are plain functions of the end-of-step state (pump performance factors, wear sums,
    steam-generator system averages, TSP deposit aggregates, level-control errors ...);
  * `result_log_columns()`: 15 keys of the step's secondary result; `clock_log_columns()`: 4 step counters;
  * `_all_diagnostic_columns()`: 137 step-internal values from the diagnostics build of the step kernel (`StateLog(env,
    diagnostics=True)`: per-stage turbine conditions, SG capacities and heat fluxes, pump health, alarm counts, bearing oil
    temperatures ...);
  * `history_log_columns()`: 1 column that is a window over another logged column's history (emitted when the log holds every
    step since the reset);
  * `constant_log_columns()`: 252 columns that hold one value in every row of both reference logs, with that value.
`StateLog(env)` without a field list records the members all of these need and `table()` emits every such column; members
selected by name that the reference does not log come out as `npb.<section>.<member>`.

    log = StateLog(env, fields=["pump.oil_level", "sec.electrical_power_output"], every=12, capacity=64)
    for t in range(steps):
        env.step(...); log.maybe_record(t + 1, time_minutes=(t + 1) * env.dt)
    log.write_parquet("run.parquet")          # long format: one row per (sample, plant)
"""
from __future__ import annotations

import ctypes
import json
import os
import re
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import _lib
from .schema import SCHEMA

_NAMES_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "state_names.json")


def reference_names() -> Dict[str, str]:
    """schema label -> ONE of the reference's state-log column names, for the members that are logged there."""
    with open(_NAMES_PATH) as fh:
        return json.load(fh)["names"]


def reference_log_columns() -> Dict[str, tuple]:
    """the reference's log column -> (schema label, factor): every column of its log that is a state member (times a unit factor).
    The name map is made from runs in which the spare pump FWP-4 never moves, so its columns cannot be matched by their series;
    they are taken by analogy -- a column FWP-4.x is the member of pump 3 that FWP-1.x is of pump 0 -- and checked like the rest
    against the reference's own logs (where they hold their resting values)."""
    with open(_NAMES_PATH) as fh:
        d = json.load(fh)
    out = {k: (v[0], float(v[1])) for k, v in d["log_columns"].items()}
    for name in d.get("constants", {}):
        if "FWP-4" in name and name not in out:
            twin = out.get(name.replace("FWP-4", "FWP-1"))
            if twin is not None and twin[0].startswith(("pump[0].", "mpump[0].")):
                out[name] = (twin[0].replace("pump[0].", "pump[3].", 1), twin[1])
    return out


def constant_log_columns() -> Dict[str, float]:
    """The reference's log columns that hold one and the same value in every row of every reference log under tests/golden/
    (a quiet run and an eventful one): configuration values, flags at rest, readings of equipment that never runs -- name ->
    value.  Columns that some other rule here produces (a state member, a derived value, a diagnostic) are left to that rule."""
    with open(_NAMES_PATH) as fh:
        const = json.load(fh).get("constants", {})
    taken = set(reference_log_columns()) | set(derived_log_columns()) | set(result_log_columns()) | set(_all_diagnostic_columns()) | set(history_log_columns())
    return {k: float(v) for k, v in const.items() if k not in taken}


def _max(a, b):
    return np.where(b > a, b, a)


def derived_log_columns() -> Dict[str, tuple]:
    """The reference's log columns that are not a state member but a plain function of state members at the moment the log
    is taken: name -> (member labels it needs, function of their arrays).  Each formula restates the provider's
    get_state_dict (cited); all of them were checked series for series on the three runs the name map is built from, and are
    checked against the reference's own log in tests/test_gpu_parity.py.  (Derived values that are left over from inside
    the step -- stage conditions, heat fluxes, the pumps' health factor -- are not functions of the end-of-step state and
    are not here.)"""
    out = {}
    for k in range(4):
        P, N = "pump[%d]." % k, "secondary.feedwater_FWP-%d." % (k + 1)
        # pump_lubrication.py:225-238 (properties), :1596-1620 (state dict)
        out[N + "efficiency_factor"] = ((P + "efficiency_degradation",), lambda d: _max(0.5, 1 - d / 100))
        out[N + "flow_factor"] = ((P + "flow_degradation",), lambda d: _max(0.5, 1 - d / 100))
        out[N + "sum_wear_level"] = ((P + "wear_impeller", P + "wear_motor_bearings", P + "wear_pump_bearings", P + "wear_thrust_bearing",
                                     P + "wear_mechanical_seals"), lambda i, a, b, c, s_: i + _max(_max(a, b), c) + s_)
    S = "secondary.steam_generator_SECONDARY-COMP-001-SG."      # steam_generator/enhanced_physics.py:672-723
    three = lambda f: tuple("sg[%d].%s" % (i, f) for i in range(3))
    mean3 = lambda a, b, c: (0 + a + b + c) / 3
    out[S + "system_avg_tsp_fouling_fraction"] = (three("tsp_fouling_fraction"), mean3)
    out[S + "system_avg_tsp_heat_transfer_degradation"] = (three("tsp_ht_degradation"), mean3)
    out[S + "system_avg_scale_thickness_mm"] = (three("scale_thickness"), mean3)
    out[S + "system_avg_scale_thermal_resistance"] = (three("scale_thermal_resistance"), mean3)
    out[S + "system_total_fouling_impact"] = (three("tsp_ht_degradation") + three("scale_thermal_resistance"),
                                              lambda a, b, c, x, y, z: mean3(a, b, c) + mean3(x, y, z) * 1000.0)
    out[S + "system_total_thermal_power"] = (three("heat_transfer_rate"), lambda a, b, c: (0 + a + b + c) / 1e6)
    out[S + "system_load_demand"] = (("sec.load_demand",), lambda x: x / 100.0)
    F = "secondary.feedwater_SECONDARY-COMP-001-FW."             # feedwater/level_control.py state dict
    errs = tuple("fw.previous_level_errors[%d]" % i for i in range(3))
    out[F + "level_control_avg_error"] = (errs, lambda a, b, c: (np.abs(a) + np.abs(b) + np.abs(c)) / 3)
    out[F + "level_control_max_error"] = (errs, lambda a, b, c: _max(_max(np.abs(a), np.abs(b)), np.abs(c)))
    out[F + "diagnostics_maintenance_urgency"] = (("fw.overall_health_score",), lambda h: 1.0 - h)   # performance_monitoring.py:565
    for i in range(3):                                           # tsp_fouling_model.py:131-152
        need = tuple("sg[%d].tsp_%s[%d]" % (i, sp, l) for l in range(7) for sp in ("magnetite", "copper", "silica", "biological"))

        def totals(*v):
            return [v[4 * l] + v[4 * l + 1] + v[4 * l + 2] + v[4 * l + 3] for l in range(7)]

        def avg(*v):
            s_ = 0.0
            for t_ in totals(*v):
                s_ = s_ + t_
            return s_ / 7

        def mx(*v):
            m_ = 0.0 * v[0]
            for t_ in totals(*v):
                m_ = _max(m_, t_)
            return m_
        out["secondary.steam_generator_SG-%d.tsp_average_deposit_thickness" % i] = (need, avg)
        out["secondary.steam_generator_SG-%d.tsp_maximum_deposit_thickness" % i] = (need, mx)
    # steam generators: what SteamGenerator.get_state_dict recomputes from the fouling state when the log is taken
    # (steam_generator.py:943-985 with _apply_tsp_flow_restrictions :516-547, _calculate_primary_flow_restriction :549-601 and
    # _calculate_pump_energy_consumption :636-662, each called with the design flows of the default configuration: 500 / 500 /
    # 5 700 kg/s, tube inner diameter 19.1 mm)
    def primary_capacity(scale_mm):
        d = 0.0191
        eff = _max(d - 2.0 * (scale_mm / 1000.0), d * 0.5)
        area_ratio = (np.pi * (eff / 2.0) ** 2) / (np.pi * (d / 2.0) ** 2)
        pdr = 1.0 / ((eff / d) ** 4)
        factor = np.where(pdr <= 3.0, area_ratio, area_ratio * (3.0 / pdr) ** 0.5)
        return np.minimum(5700.0, 5700.0 * factor)
    tsp_capacity = lambda pdr: np.minimum(500.0, 500.0 * (1.0 / np.sqrt(pdr)))
    for i in range(3):
        G, N = "sg[%d]." % i, "secondary.steam_generator_SG-%d." % i
        out[N + "max_steam_flow_capacity"] = ((G + "tsp_pressure_drop_ratio",), tsp_capacity)
        out[N + "max_feedwater_flow_capacity"] = ((G + "tsp_pressure_drop_ratio",), tsp_capacity)
        out[N + "secondary_flow_restriction_factor"] = ((G + "tsp_pressure_drop_ratio",), lambda pdr: tsp_capacity(pdr) / 500.0)
        out[N + "max_primary_flow_capacity"] = ((G + "scale_thickness",), primary_capacity)
        out[N + "primary_flow_restriction_factor"] = ((G + "scale_thickness",), lambda sc: primary_capacity(sc) / 5700.0)
        out[N + "fouling_energy_penalty_mw"] = ((G + "tsp_pressure_drop_ratio",), lambda pdr: 5.0 * (pdr - 1.0) * 0.5)
        out[N + "total_pump_power_mw"] = ((G + "tsp_pressure_drop_ratio",), lambda pdr: 5.0 + 5.0 * (pdr - 1.0) * 0.5)
        # steam_generator.py:251-290: the operating heat flux is the step's heat transfer over the design area, floored
        out[N + "heat_flux"] = ((G + "heat_transfer_rate",), lambda q: _max(q / 5000.0, 5000.0))   # heat_transfer_area_per_sg = 5 000 m2
    # turbine stages, stage_system.py:379-393 (state dict) with :294-339 (update_degradation, the last thing a step does to a
    # stage): the logged efficiency and blade condition are recomputed there from the end-of-step degradation state.  The stage
    # system's own dict takes every stage's un-prefixed keys in turn (:1028-1030), so the turbine-level columns are LP-6's.
    stage_names = ["HP-%d" % (k + 1) for k in range(8)] + ["LP-%d" % (k + 1) for k in range(6)]
    for k, sn in enumerate(stage_names):
        targets = ["secondary.turbine_%s." % sn] + (["secondary.turbine_SECONDARY-COMP-001-TURB."] if k == 13 else [])
        for N in targets:
            out[N + "efficiency"] = (("tstg.stage_efficiency_degradation[%d]" % k,), lambda d: _max(0.7, 0.88 - d))
            out[N + "blade_condition"] = (("tstg.stage_deposit_thickness[%d]" % k, "tstg.stage_blade_wear_factor[%d]" % k),
                                          lambda dep, wear: np.minimum(1.0 / (1.0 + dep / 0.5), wear))
    # feedwater pumps: the motor's current from its hydraulic load (pump_system.py:697-706, rated flow 500 kg/s)
    for k in range(4):
        out["secondary.feedwater_FWP-%d.motor_current" % (k + 1)] = (("pump[%d].flow_rate" % k,), lambda q: 200.0 + 100.0 * (q / 500.0))
    out["secondary.feedwater.pump_system_num_running"] = (("fw.running_mask",), lambda m: sum(((np.asarray(m).astype(np.int64) >> i) & 1) for i in range(4)).astype(np.float64))
    out[F + "level_control_performance"] = (errs, lambda a, b, c: _max(0.0, 1.0 - ((np.abs(a) + np.abs(b) + np.abs(c)) / 3) / 2.0))   # level_control.py:346-347
    # the turbine bearing lubrication system's health factor (lubrication_base.py:380-399, components turbine_bearing_lubrication.py:99-169)
    tb_wpf, tb_lpf = (0.02, 0.018, 0.03, 0.025, 0.01), (0.5, 0.45, 0.7, 0.6, 0.2)

    def tb_health(eff, *wear):
        tot = 0.0
        for w, a, b in zip(wear, tb_wpf, tb_lpf):
            tot = tot + _max(0.1, 1.0 - (w * a + (1.0 - eff) * b))
        return tot / 5 * eff
    out["secondary.turbine_TB-LUB-001.system_health_factor"] = (("turb.lub_effectiveness",) + tuple("turb.lub_wear[%d]" % k for k in range(5)), tb_health)
    out["secondary.condenser_SECONDARY-COMP-001-COND.condensate_flow"] = (("sec.total_steam_flow",), lambda q: q - 250.0)   # the main steam less the extraction flows
    # enhanced_physics.py:815-821 with the property fits :1285-1310: steam enthalpy at the header's conditions x steam rate
    def header_enthalpy(p_mpa, temp_c):
        pb = np.clip(p_mpa * 10.0, 0.01, 100.0)
        ts = np.where(p_mpa <= 0.001, 10.0, np.clip(1730.63 / (8.07131 - np.log10(pb)) - 233.426, 10.0, 374.0))
        h_g = 4.18 * ts + 2257.0 * (1.0 - ts / 374.0) ** 0.38
        return np.where(temp_c <= ts, h_g, h_g + 2.1 * (temp_c - ts))
    out["secondary.turbine_SECONDARY-COMP-001-TURB.enhanced_turbine_heat_rate"] = (
        ("sec.sg_avg_pressure", "sec.sg_avg_temperature", "turb.total_power_output", "sec.total_steam_flow"),
        lambda p_, t_, pw, q: np.where(pw > 0, header_enthalpy(p_, t_) * (q / np.where(pw > 0, pw * 1000, 1.0) * 3600) / 1000, 0.0))
    # feedwater/physics.py:789-798: hydraulic power (flow x (design pressure 8.0 - suction 0.5 MPa) ...) over the pumps' power
    out[F + "feedwater_system_efficiency"] = (("fw.total_flow_rate", "fw.total_power_consumption"),
                                              lambda q, pw: np.where(pw > 0, (q * (8.0 - 0.5) * 1e6 * 1000 * 9.81) / 1e6 / np.where(pw > 0, pw, 1.0), 0.0))
    # tube_interior_fouling.py:268-271: each generator's tube-side fouling fraction from its scale resistance
    out[S + "system_avg_tube_fouling_fraction"] = (three("scale_thermal_resistance"), lambda a, b, c: (0 + np.minimum(a / 0.001, 1.0) + np.minimum(b / 0.001, 1.0) + np.minimum(c / 0.001, 1.0)) / 3)
    # the shared WaterChemistry's composite indices, recomputed from its concentrations (water_chemistry.py:277-320; iron 0.1 ppm,
    # silica 20 ppm, alkalinity 120 mg/L and the concentration factor 5 never change), logged once under its own name and once
    # more in the feedwater system's state dict
    def ph_saturation(tds, hardness):
        return (9.3 + (np.log10(tds) - 1) / 10 + (-13.12 * np.log10(25.0 + 273) + 34.55)) - ((np.log10(hardness) - 0.4) + np.log10(120.0))
    for pre in ("secondary.water_chemistry.", F):
        out[pre + "water_chemistry_particle_content"] = (("chem[0].total_dissolved_solids",), lambda tds: np.clip(1.0 + (tds / 500.0 + 0.1 * 2.0 + 20.0 / 20.0) * 0.1, 0.5, 2.0))
        out[pre + "water_chemistry_corrosion_tendency"] = (("chem[0].total_dissolved_solids", "chem[0].hardness", "chem[0].ph"),
                                                           lambda tds, hard, ph: 2 * ph_saturation(tds, hard) - ph)
        out[pre + "water_chemistry_stability_factor"] = (("chem[0].ph", "chem[0].treatment_efficiency"),
                                                         lambda ph, te: np.clip(((1.0 - np.abs(ph - 9.2) / 2.0) + te + (1.0 - abs(5.0 - 2.0) / 3.0)) / 3.0, 0.1, 1.0))
    mapped = set(reference_log_columns())
    return {k: v for k, v in out.items() if k not in mapped}


def clock_log_columns(dt: float) -> Dict[str, tuple]:
    """Log columns that count the steps taken: name -> (member labels, function), for a plant stepped with ``dt``.  The shared
    WaterChemistry is updated twice a step (feedwater/physics.py:708, secondary/__init__.py:644), each time adding its guess of
    dt in hours (water_chemistry.py:335-348) to operating_hours and last_treatment_time; the number of steps is read off the
    secondary side's own hour counter (operating_hours += dt / 3600, secondary/__init__.py:632)."""
    dth = dt / 3600.0 if dt > 100 else (dt / 60.0 if dt > 1 else dt)
    hours = (("sec.operating_hours",), lambda h: 2 * dth * np.round(h * 3600.0 / dt))
    F = "secondary.feedwater_SECONDARY-COMP-001-FW."
    return {pre + k: hours for pre in ("secondary.water_chemistry.", F) for k in ("water_chemistry_operating_hours", "water_chemistry_time_since_treatment")}


def history_log_columns() -> Dict[str, tuple]:
    """Log columns that are a function of another log column's recent history: name -> (source log column, function of the
    [samples, plants] series).  The pH controller's RMS deviation is the root mean square of the last 100 steps' |pH error|
    (ph_control_system.py:441-455), a list the reference keeps on the controller; a log that holds every step since the reset
    holds the same list, so ``table()`` emits the column exactly then (steps 1, 2, 3 ... recorded with every=1)."""
    def rms_of_last_100(series):
        ns = series.shape[0]
        csum = np.concatenate([np.zeros((1,) + series.shape[1:]), np.cumsum(np.square(series), axis=0)], axis=0)
        hi = np.arange(1, ns + 1); lo = np.maximum(0, hi - 100)
        return np.sqrt((csum[hi] - csum[lo]) / (hi - lo).reshape((-1,) + (1,) * (series.ndim - 1)))
    return {"secondary.ph_control.ph_control_deviation_rms": ("secondary.ph_control.ph_control_error", rms_of_last_100)}


def result_log_columns() -> Dict[str, tuple]:
    """The reference's log columns that are scalar keys of the step's info["secondary_system"] (BatchedPlantEnv.secondary_result):
    log column -> (key, factor).  The heat-flow tracker's and the stage system's state dicts hand the same numbers to the state
    manager that the result dict carries (secondary/__init__.py:922-1010, heat_flow_tracker.py:324-351, stage_system.py:1018-1026);
    found by matching the m1 run's log against its recorded result dicts series for series, checked against the reference's
    own log in tests/test_gpu_parity.py."""
    R, T = "secondary.reactor_SECONDARY-COMP-001.", "secondary.turbine_SECONDARY-COMP-001-TURB."
    return {
        "secondary.condenser_SECONDARY-COMP-001-COND.condenser_thermal_performance": ("condenser_thermal_performance", 1.0),
        R + "heat_flow_condenser_heat_rejection": ("heat_flow_condenser_heat_rejection", 1.0),
        R + "heat_flow_energy_balance_error": ("heat_flow_energy_balance_error", 1.0),
        R + "heat_flow_energy_balance_ok": ("heat_flow_balance_ok", 1.0),
        R + "heat_flow_energy_balance_percent": ("heat_flow_energy_balance_percent", 1.0),
        R + "heat_flow_net_electrical_output": ("heat_flow_net_electrical_output", 1.0),
        R + "heat_flow_overall_efficiency": ("heat_flow_overall_efficiency", 1.0),
        R + "heat_flow_turbine_work_output": ("turbine_mechanical_power", 1.0),
        R + "heat_flow_sg_heat_input": ("total_heat_transfer", 1e-6),
        R + "heat_flow_steam_enthalpy_flow": ("total_heat_transfer", 0.98e-6),
        R + "system_total_heat_transfer": ("total_heat_transfer", 1e-6),
        R + "system_total_system_heat_rejection": ("total_system_heat_rejection", 1e-6),
        T + "enhanced_turbine_efficiency": ("turbine_efficiency", 1.0),
        T + "enhanced_turbine_steam_rate": ("turbine_steam_rate", 1.0),
        T + "stage_system_efficiency": ("turbine_efficiency", 1.0),
    }


def log_columns(fields: Optional[Sequence[str]] = None) -> List[tuple]:
    """[(kind, slot, schema label, log column name)] of the members a log would hold.  `fields`: schema names
    ("pump.oil_level" = every instance and element, or a full label such as "pump[2].oil_level"); None = every
    member the reference itself logs."""
    names = reference_names()
    cols = SCHEMA.columns()
    out = []
    if fields is None:
        wanted = {label for label, _f in reference_log_columns().values()}
        for need, _fn in derived_log_columns().values():
            wanted.update(need)
        for need, _fn in clock_log_columns(1.0).values():
            wanted.update(need)
        for kind, slot, label, _path in cols:
            if label in wanted:
                out.append((kind, slot, label, names.get(label, "npb." + label)))
        return out
    for want in fields:
        hit = False
        for kind, slot, label, _path in cols:
            if want == label or want == re.sub(r"\[\d+\]", "", label):
                out.append((kind, slot, label, names.get(label, "npb." + label)))
                hit = True
        if not hit:
            raise KeyError("no state member named %r" % want)
    return out


def diagnostic_log_columns() -> Dict[str, int]:
    """The reference's log columns that are left over from inside the turbine stages' expansion: log column -> row of
    BatchedPlantEnv.diagnostics (include/npb.h NPB_DIAG_*).  TurbineStage.get_state_dict (stage_system.py:379-393) under each
    stage's own name, and once more under the turbine's, where the stage system's dict keeps the last stage's un-prefixed keys
    (:1028-1030).  Columns that are already state members (HP-1's inlet is the steam header) are left to reference_log_columns."""
    stage_names = ["HP-%d" % (k + 1) for k in range(8)] + ["LP-%d" % (k + 1) for k in range(6)]
    out = {}
    for v, value in enumerate(_lib.DIAG_STAGE_VALUES):
        for k, sn in enumerate(stage_names):
            out["secondary.turbine_%s.%s" % (sn, value)] = v * 14 + k
        out["secondary.turbine_SECONDARY-COMP-001-TURB.%s" % value] = v * 14 + 13
    for v, value in enumerate(_lib.DIAG_SG_VALUES):       # SteamGenerator.get_state_dict (steam_generator.py:943-985)
        for i in range(3):
            out["secondary.steam_generator_SG-%d.%s" % (i, value)] = 14 * len(_lib.DIAG_STAGE_VALUES) + v * 3 + i
    known = set(json.load(open(_NAMES_PATH))["unmatched"])
    return {name: row for name, row in out.items() if name in known}


def _all_diagnostic_columns() -> Dict[str, int]:
    """diagnostic_log_columns plus the rows that are not per stage / per steam generator (include/npb.h NPB_DIAG_PUMP_*, NPB_DIAG_FW_*)"""
    out = dict(diagnostic_log_columns())
    base = 14 * len(_lib.DIAG_STAGE_VALUES) + 3 * len(_lib.DIAG_SG_VALUES)
    for v, value in enumerate(_lib.DIAG_PUMP_VALUES):
        for k in range(4):
            out["secondary.feedwater_FWP-%d.%s" % (k + 1, value)] = base + v * 4 + k
    base += 4 * len(_lib.DIAG_PUMP_VALUES)
    for v, value in enumerate(_lib.DIAG_FW_VALUES):
        out["secondary.feedwater_SECONDARY-COMP-001-FW.%s" % value] = base + v
    base += len(_lib.DIAG_FW_VALUES)
    for v, value in enumerate(_lib.DIAG_ROTOR_VALUES):
        out["secondary.turbine_SECONDARY-COMP-001-TURB.%s" % value] = base + v
    base += len(_lib.DIAG_ROTOR_VALUES)
    for name, off in _lib.DIAG_TAIL_COLUMNS:
        out[name] = base + off
    return out


class StateLog:
    """Device-resident ring of samples of selected state members of every plant."""

    def __init__(self, env, fields: Optional[Sequence[str]] = None, every: int = 1, capacity: int = 256, diagnostics: bool = False):
        self.env = env
        self.columns = log_columns(fields)
        # with no field list the table carries the reference's log columns (several per member, unit factors applied)
        self._reference_layout = fields is None
        if not self.columns:
            raise ValueError("no columns to log")
        self.every = max(1, int(every))
        self.capacity = int(capacity)
        nf = len(self.columns)
        self._kinds = (ctypes.c_int * nf)(*[0 if c[0] == "f64" else 1 for c in self.columns])
        self._slots = (ctypes.c_int * nf)(*[c[1] for c in self.columns])
        self._buf = torch.empty((self.capacity, nf, env.n), dtype=torch.float64, device=env.device)
        # the log columns that are keys of the step's secondary result dict (reference layout, plants with a secondary side)
        self._res_keys = sorted({k for k, _f in result_log_columns().values()}) if self._reference_layout and env.params.mode == _lib.MODE_FULL else []
        self._res = torch.empty((self.capacity, len(self._res_keys), env.n), dtype=torch.float64, device=env.device) if self._res_keys else None
        # step-internal diagnostics (reference layout, full mode): switches the env to the diagnostics build of the step kernel
        self._diag = None
        if diagnostics and self._res_keys:
            if getattr(env, "diagnostics", None) is None:
                env.enable_diagnostics(True)
            self._diag = torch.empty((self.capacity, _lib.DIAG_DIM, env.n), dtype=torch.float64, device=env.device)
        self._times: List[float] = []
        self._steps: List[int] = []

    def __len__(self) -> int:
        return len(self._times)

    def record(self, step: int, time_minutes: float) -> None:
        """Sample now (one kernel launch on the env's stream)."""
        row = len(self._times)
        if row >= self.capacity:
            raise RuntimeError("StateLog is full (%d samples): flush it with table() / write_parquet() and clear()" % self.capacity)
        out = self._buf[row]
        _lib.check(self.env.L.npb_gather_fields(self.env._h, len(self.columns), self._kinds, self._slots,
                                                ctypes.c_void_p(out.data_ptr()), self.env._stream()), self.env._h)
        if self._res is not None:     # valid for the step just taken: call record() between that step and the next
            res = self.env.secondary_result()
            for j, k in enumerate(self._res_keys):
                self._res[row, j] = res[k]
        if self._diag is not None:
            self._diag[row].copy_(self.env.diagnostics)
        self._times.append(float(time_minutes)); self._steps.append(int(step))

    def maybe_record(self, step: int, time_minutes: float) -> bool:
        if step % self.every:
            return False
        self.record(step, time_minutes)
        return True

    def clear(self) -> None:
        self._times.clear(); self._steps.clear()

    def array(self) -> np.ndarray:
        """[samples, fields, plants] on the host."""
        return self._buf[:len(self._times)].cpu().numpy()

    def table(self, plants: Optional[Sequence[int]] = None):
        """pyarrow Table in long format: step, time (minutes), plant, then one column per member."""
        import pyarrow as pa
        data = self.array()
        idx = np.arange(self.env.n) if plants is None else np.asarray(plants)
        data = data[:, :, idx]
        ns, nf, npl = data.shape
        cols = {"step": np.repeat(np.asarray(self._steps, dtype=np.int64), npl),
                "time": np.repeat(np.asarray(self._times, dtype=np.float64), npl),
                "plant": np.tile(idx.astype(np.int64), ns)}
        if self._reference_layout:
            index = {c[2]: f for f, c in enumerate(self.columns)}
            for name, (label, factor) in sorted(reference_log_columns().items()):
                v = data[:, index[label], :].reshape(-1)
                cols[name] = v * factor if factor != 1.0 else v
            for name, (need, fn) in sorted(derived_log_columns().items()):
                cols[name] = np.asarray(fn(*[data[:, index[label], :].reshape(-1) for label in need]), dtype=np.float64)
            if self._res is not None:
                res = self._res[:len(self._times)].cpu().numpy()[:, :, idx]
                for name, (key, factor) in sorted(result_log_columns().items()):
                    cols[name] = res[:, self._res_keys.index(key), :].reshape(-1) * factor
            for name, (need, fn) in sorted(clock_log_columns(float(self.env.params.dt)).items()):
                cols[name] = np.asarray(fn(*[data[:, index[label], :].reshape(-1) for label in need]), dtype=np.float64)
            if self._diag is not None:
                dg = self._diag[:len(self._times)].cpu().numpy()[:, :, idx]
                for name, row in sorted(_all_diagnostic_columns().items()):
                    cols[name] = dg[:, row, :].reshape(-1)
            if self._steps == list(range(1, ns + 1)):      # every step since the reset: the windowed columns can be formed
                for name, (source, fn) in sorted(history_log_columns().items()):
                    cols[name] = np.asarray(fn(cols[source].reshape(ns, npl)), dtype=np.float64).reshape(-1)
            for name, value in sorted(constant_log_columns().items()):
                if name not in cols:
                    cols[name] = np.full(ns * npl, value)
            return pa.table(cols)
        for f, (kind, _slot, _label, name) in enumerate(self.columns):
            v = data[:, f, :].reshape(-1)
            cols[name] = v.astype(np.int32) if kind == "i32" else v
        return pa.table(cols)

    def write_parquet(self, path: str, plants: Optional[Sequence[int]] = None) -> None:
        import pyarrow.parquet as pq
        pq.write_table(self.table(plants), path)
