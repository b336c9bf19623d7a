"""Columnar state log of a batch.

The reference's StateManager appends one pandas row per plant-step -- `collect_states` walks every registered
provider's `get_state_dict()` and `pd.concat`s the row (simulator/state/state_manager.py:152-233), 790 columns named
`category.variable` (auto_register.py:83-165); with physics on a GPU that bookkeeping would be the whole run time.
Here a sample is one gather kernel (`npb_gather_fields`): the chosen members of every plant, widened to double, land
in a device buffer `[sample, field, plant]`; nothing touches the host until `table()` / `write_parquet()`.

Columns carry the reference's own log names, all 784 of its numeric log columns, each by ONE rule (checked at every step against
the reference's own logs of five runs -- a quiet one, an eventful one, a ReactorHeatSource run with operator actions and a scram, a
feedwater run with four kinds of maintenance, a pump trip cascade and the pH controller switching chemicals, a turbine / steam
generator run with TSP shutdown, a vibration trip and an ejector rotation -- tests/test_gpu_parity.py, tests/test_statelog_cpu.py):
  * `reference_log_columns()`: columns that SHOW a state member, established by intervention on the live reference
    (the harness script named in DESIGN.md section 6 pokes every member and reads the providers back; `state_names.json` "poked"), plus the
    members an attribute of the reference is assigned together with, by construction (`_ALIASES`, each with its line);
  * `derived_log_columns()`: plain functions of the end-of-step state (pump factors, wear sums, SG system statistics, TSP stage and
    recommendation, vibration components, ejector performance, alarms ...), each restating the provider's get_state_dict;
  * `result_log_columns()`: 15 keys of the step's secondary result; `clock_log_columns()`: hour counters that advance by a fixed
    amount per step; `output_log_columns()`: the step's own outputs (the scram pulse);
  * `_all_diagnostic_columns()`: values from inside the step, from the diagnostics build of the step kernel (`StateLog(env,
    diagnostics=True)`; include/npb.h NPB_DIAG_*): per-stage turbine conditions and extraction flows, SG heat transfer, pump health,
    which maintenance action a pump received, the protection system's trip bookkeeping, per-ejector performance ...;
  * `history_log_columns()`: columns that are a window over another logged column's history (emitted when the log holds every
    step since the reset);
  * `parameter_log_columns()`: configuration values and attributes that the reference sets at construction and no code on the
    stepped path ever writes -- each with the reference line that sets it (and, where include/npb_params.h carries it, read from
    the handle's parameters).  Round 3 had 252 columns here by harvest ("held one value in two logs"); what is left is what
    the reference itself never moves.
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


_F = "secondary.feedwater_SECONDARY-COMP-001-FW."
_T = "secondary.turbine_SECONDARY-COMP-001-TURB."
_C = "secondary.condenser_SECONDARY-COMP-001-COND."
_G = "secondary.steam_generator_SECONDARY-COMP-001-SG."
_R = "secondary.reactor_SECONDARY-COMP-001."


def _names_json() -> dict:
    with open(_NAMES_PATH) as fh:
        return json.load(fh)


# Log columns whose attribute the reference ASSIGNS TOGETHER WITH a schema member (same statement or the same value two lines apart),
# so that after a step the two are one number by construction -- the poke of make_log_map.py cannot see these (it moves the member,
# not the twin).  log column -> (member, reference line that ties them)
def _aliases() -> Dict[str, tuple]:
    out = {
        "secondary.feedwater.pump_system_total_flow": ("fw.total_flow_rate", "feedwater/pump_system.py:1292 -> feedwater/physics.py:775-776"),
        "secondary.feedwater.pump_system_total_power": ("fw.total_power_consumption", "feedwater/pump_system.py:1293 -> feedwater/physics.py:775-776"),
        _F + "diagnostics_reliability": ("fw.overall_health_score", "feedwater/performance_monitoring.py:562"),
        _F + "protection_npsh_current": ("pump[3].npsh_available", "feedwater/protection_system.py:75, visited once per pump in dict order (:404-410): FWP-4's stands"),
        _C + "cooling_water_inlet_temp": ("sec.cooling_water_temperature", "condenser/physics.py:767"),
        _G + "system_total_steam_flow": ("sec.total_steam_flow", "steam_generator/enhanced_physics.py:515 = secondary/__init__.py:559"),
        _T + "stage_system_steam_flow": ("sec.total_steam_flow", "turbine/stage_system.py:977: the inlet flow the secondary side hands the turbine, secondary/__init__.py:559-568"),
        "secondary.ph_control.ph_control_ammonia_dose_rate": ("ph.pending_ammonia_dose", "ph_control_system.py:474-480 -> water_chemistry.py:675 (the controller's outputs, stored as they are)"),
        "secondary.ph_control.ph_control_morpholine_dose_rate": ("ph.pending_morpholine_dose", "ph_control_system.py:474-480 -> water_chemistry.py:675"),
    }
    for k in range(3):
        out[_F + "level_control_sg_%d_error" % (k + 1)] = ("fw.previous_level_errors[%d]" % k, "feedwater/level_control.py: level_errors[i] and previous_level_errors[i] take the same level_error in one pass (:205-330)")
        out["secondary.steam_generator_SG-%d.feedwater_temperature" % k] = ("sec.previous_feedwater_temp", "steam_generator.py:762 <- the smoothed temperature of secondary/__init__.py:395-398")
        out["secondary.steam_generator_SG-%d.tsp_years_since_cleaning" % k] = ("sg[%d].tsp_operating_years" % k, "fouling_model_base.py:104-108: both += dt_years; apart only after a cleaning, which is steam-generator maintenance (outside the path)")
    return out


def reference_log_columns() -> Dict[str, tuple]:
    """the reference's log column -> (schema label, factor): every column of its log that SHOWS a state member.  Established by
    intervention (the generator script is named in DESIGN.md section 6: each member poked on the live reference to two values, the providers read
    back; a column that equals factor x the poked value both times shows that member), plus the by-construction twins above."""
    d = _names_json()
    out = {k: (v[0][0], float(v[0][1])) for k, v in d["poked"]["log_columns"].items()}
    for name, (label, _why) in _aliases().items():
        assert name not in out, name
        out[name] = (label, 1.0)
    return out


def reference_names() -> Dict[str, str]:
    """schema label -> ONE of the reference's state-log column names, for the members that are logged there (several columns can
    show one member: the closest name, as state_names.json "names" chose it, where the intervention map agrees)"""
    lc = reference_log_columns()
    names = {label: name for label, name in _names_json()["names"].items() if lc.get(name) == (label, 1.0)}
    for name, (label, factor) in sorted(lc.items()):
        if factor == 1.0 and label not in names:
            names[label] = name
    return names


def parameter_log_columns(params=None) -> Dict[str, tuple]:
    """Log columns the stepped path never moves: configuration values, and attributes the reference sets when it constructs a
    component and that no code reachable from NuclearPlantSimulator.step() writes again.  name -> (value, where the reference
    sets it).  ``params``: the handle's npb_params_t (BatchedPlantEnv.params); the few values it carries are read from it."""
    g = lambda name, default: float(getattr(params, name, default)) if params is not None else float(default)
    out = {
        "primary.reactor.rated_power_mw": (g("rated_power_mw", 3000.0), "primary/__init__.py:174 (params.rated_power_mw)"),
        _C + "cooling_water_flow": (45000.0, "secondary/__init__.py:375 control_inputs.get('cooling_water_flow', 45000.0); sim.py passes none -> condenser/physics.py:766"),
        _C + "steam_inlet_pressure": (0.007, "turbine/enhanced_physics.py hands back the condenser_pressure=0.007 it was given (secondary/__init__.py:567, :610) -> condenser/physics.py:762"),
        _F + "feedwater_load_demand": (1.0, "feedwater/physics.py:183; only set_load_demand (:896) writes it and the step never calls it"),
        _F + "level_control_auto_mode": (1.0, "feedwater/level_control.py:150"),
        _F + "level_control_manual_setpoint": (3 * g("sg_design_feedwater_flow_per_sg", 500.0), "feedwater/level_control.py:151: design_flow_per_sg * num_steam_generators"),
        _F + "diagnostics_time_since_maintenance": (0.0, "feedwater/performance_monitoring.py:259; += only in the legacy wear-tracking branch (:493-503), not taken by pumps that own a lubrication system"),
        _F + "protection_system_available": (1.0, "feedwater/protection_system.py:184"),
        _F + "protection_false_trip_count": (0.0, "feedwater/protection_system.py:188; incremented by perform_protection_test only"),
        "secondary.ph_control.ph_control_setpoint": (9.2, "ph_control_system.py:204 config.target_ph"),
        "secondary.ph_control.ph_control_alarm_equipment": (0.0, "ph_control_system.py:396-414: set by the random equipment failures, which the deterministic limit of include/npb_fields.h (NPB_PH_FIELDS) leaves out"),
        _R + "system_feedwater_temperature": (227.0, "secondary/__init__.py:373 control_inputs.get('feedwater_temp', 227.0); sim.py passes none"),
        _R + "system_num_steam_generators": (3.0, "secondary/__init__.py:256"),
        _G + "system_efficiency": (0.98, "steam_generator/enhanced_physics.py:139 / config.py:274 design_efficiency"),
        _G + "system_performance_factor": (1.0, "steam_generator/enhanced_physics.py:143; moved by steam-generator maintenance only (:664, :988 ...), outside the path"),
        _G + "system_load_balance_factor": (1.0, "steam_generator/enhanced_physics.py:145; moved by steam-generator maintenance only"),
        _G + "system_num_steam_generators": (3.0, "steam_generator/enhanced_physics.py:685 config.num_steam_generators"),
        _T + "vibration_trend_slope": (0.0, "turbine/rotor_dynamics.py:616; nothing on the path computes a trend"),
        _T + "stage_system_extraction_flow": None,   # produced from the diagnostics (placeholder removed below)
    }
    del out[_T + "stage_system_extraction_flow"]
    for pre in ("secondary.water_chemistry.", _F):     # the shared WaterChemistry, logged under its own name and in the feedwater system's dict
        out[pre + "water_chemistry_iron_concentration"] = (0.1, "water_chemistry.py:234 config.design_iron_concentration")
        out[pre + "water_chemistry_copper_concentration"] = (0.05, "water_chemistry.py:235")
        out[pre + "water_chemistry_silica_concentration"] = (20.0, "water_chemistry.py:236")
        out[pre + "water_chemistry_alkalinity"] = (120.0, "water_chemistry.py:243")
        out[pre + "water_chemistry_concentration_factor"] = (5.0, "water_chemistry.py:373-374: 1 / (blowdown 0.02 + evaporation) capped at config.concentration_factor_max, the cap at every update")
        out[pre + "water_chemistry_blowdown_rate"] = (0.02, "water_chemistry.py:263, re-set from the literal 'blowdown_rate': 0.02 of every caller (condenser/physics.py:773, feedwater/physics.py:705-712)")
    for k in range(3):
        out[_F + "level_control_sg_%d_target" % (k + 1)] = (12.5, "feedwater/level_control.py:136 config.level_setpoint")
        N = "secondary.steam_generator_SG-%d." % k
        out[N + "steam_void_fraction"] = (0.8, "steam_generator.py:483-489: x rho_f / (x rho_f + (1 - x) rho_g) clipped to [0, 0.8]; the quality is itself clipped to >= 0.90 (:478), "
                                               "where the ratio exceeds 0.8 for any rho_f / rho_g > 0.45 -- the upper clip, always")
        out[N + "base_pump_power_mw"] = (5.0, "steam_generator/steam_generator.py:649")
        out[N + "tsp_cleaning_cycles"] = (0.0, "tsp_fouling_model.py total_cleaning_cycles: counted by perform_cleaning, i.e. steam-generator maintenance (outside the path)")
        out[N + "tube_primary_boric_acid_ppm"] = (1000.0, "steam_generator/tube_interior_fouling.py:82")
        out[N + "tube_primary_lithium_ppm"] = (2.0, "steam_generator/tube_interior_fouling.py:83")
        out[N + "tube_primary_ph"] = (7.2, "steam_generator/tube_interior_fouling.py:84")
        out[N + "tube_primary_dissolved_oxygen_ppm"] = (0.005, "steam_generator/tube_interior_fouling.py:86")
    for k in range(4):
        N = "secondary.feedwater_FWP-%d." % (k + 1)
        out[N + "motor_voltage"] = (6.6, "feedwater/pump_system.py:77, :709 ('kV constant')")
        out[N + "seal_water_pressure"] = (0.8, "feedwater/pump_lubrication.py:58, :207 config.seal_water_system_pressure")
        out[N + "npsh_margin_degradation"] = (0.0, "feedwater/pump_lubrication.py:1472 cavitation_damage * 0.5 with the argument the closure never passes (:1659-1852: 'cavitation_damage' is not in pump_conditions -> 0.0)")
        out[N + "maintenance_due"] = (0.0, "lubrication_base.py:159; set by check_maintenance_requirements (:402-432), which only the never-stepped governor calls (turbine/governor_system.py:1014)")
        out[_T + "TB-00%d_vibration_disp" % (k + 1)] = (5.0, "turbine/enhanced_physics.py:572-573 initial_conditions.bearing_vibrations; the bearing model never writes it (rotor_dynamics.py:62)")
        out[_T + "TB-00%d_vibration_vel" % (k + 1)] = (0.0, "turbine/rotor_dynamics.py:63; never written")
    L = "secondary.turbine_TB-LUB-001."          # the turbine lubrication system's performance effects: computed by update_turbine_lubrication_effects
    for name, value, why in (("oil_level", 100.0, "turbine/turbine_bearing_lubrication.py:201 <- turbine/config.py:349 oil_reservoir_level; no consumption model"),   # (:346-382), which only the module's __main__ demo calls (:1260)
                             ("efficiency_degradation_factor", 1.0, "turbine_bearing_lubrication.py:689: 1 - turbine_efficiency_degradation, computed by _calculate_turbine_performance_degradation (:384-421), reached from update_turbine_lubrication_effects only -- which nothing on the path calls"),
                             ("bearing_housing_temperature", 80.0, "turbine_bearing_lubrication.py:181 config.bearing_housing_temperature (:58); rewritten at :421 only"),
                             ("oil_cooling_effectiveness", 1.0, "turbine_bearing_lubrication.py:187; rewritten at :372 only"),
                             ("vibration_increase", 0.0, "turbine_bearing_lubrication.py:186; rewritten at :417 only"),
                             ("steam_contamination_rate", 0.0, "turbine_bearing_lubrication.py:180; rewritten at :367 only"),
                             ("maintenance_due", 0.0, "lubrication_base.py:159; check_maintenance_requirements is never called on the path")):
        out[L + name] = (value, why)
    return out


def constant_log_columns(params=None) -> Dict[str, float]:
    """name -> value of parameter_log_columns (kept under its round-3 name for callers that emit them)"""
    return {k: v[0] for k, v in parameter_log_columns(params).items()}


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
    # ---- round 4: the columns round 3 emitted as literals
    for k in range(4):
        P, N = "pump[%d]." % k, "secondary.feedwater_FWP-%d." % (k + 1)
        out[N + "head_factor"] = ((P + "head_degradation",), lambda d: _max(0.5, 1 - d / 100))                      # pump_lubrication.py:236-238
    # FeedwaterPumpSystem.system_available: enough pumps running, whatever the protection system says (pump_system.py:1297; 3 = minimum_pumps_required)
    out["secondary.feedwater.pump_system_available"] = (("fw.running_mask",), lambda m: (sum(((np.asarray(m).astype(np.int64) >> i) & 1) for i in range(4)) >= 3).astype(np.float64))
    out[F + "protection_npsh_margin"] = (("pump[3].npsh_available",), lambda v: v - 0.1)   # protection_system.py:131: current_npsh - npsh_critical_trip, which resolves to low_suction_pressure_trip = 0.1 (getattr fall-back)
    # steam generators: maxima over the three (enhanced_physics.py:712, :718), the maintenance hint (:722), the TSP stage (tsp_fouling_model.py:394-411,
    # FoulingStage order :53-58) and the replacement recommendation (:700-703: fouling >= 0.8 or older than the 40-year design life)
    max3 = lambda a, b, c: _max(_max(a, b), c)
    out[S + "system_max_tsp_fouling_fraction"] = (three("tsp_fouling_fraction"), max3)
    out[S + "system_max_scale_thickness_mm"] = (three("scale_thickness"), max3)
    out[S + "system_fouling_maintenance_needed"] = (three("tsp_fouling_fraction") + three("scale_thickness"),
                                                    lambda a, b, c, x, y, z: ((mean3(a, b, c) > 0.15) | (mean3(x, y, z) > 0.5)).astype(np.float64))
    for i in range(3):
        G, N = "sg[%d]." % i, "secondary.steam_generator_SG-%d." % i
        out[N + "tsp_fouling_stage_numeric"] = ((G + "tsp_fouling_fraction",), lambda f: np.where(f < 0.4, 0.0, np.where(f < 0.7, 1.0, np.where(f < 0.85, 2.0, 3.0))))
        out[N + "tsp_replacement_recommended"] = ((G + "tsp_fouling_fraction", G + "tsp_operating_years"), lambda f, y: ((f >= 0.8) | (y > 40.0)).astype(np.float64))
    # turbine: availability is "not tripped" (enhanced_physics.py:829); the vibration monitor's other readings are fixed multiples of
    # the displacement it stores (rotor_dynamics.py:660-693: 1X : 2X : 3X = 1 : 0.1 : 0.05, y = 0.8 x, velocity = displacement x omega / 1000,
    # acceleration = velocity x omega / 9.81, omega from the rotor speed of the same update)
    out[_T + "enhanced_turbine_availability"] = (("turb.trip_active",), lambda t: 1.0 - (np.asarray(t) != 0).astype(np.float64))
    rss = np.sqrt(1.0 + 0.1 ** 2 + 0.05 ** 2)
    vib = ("turb.vibration_displacement", "turb.rotor_speed")
    omega = lambda rpm: 2 * np.pi * (rpm / 60.0)
    out[_T + "vibration_displacement_y"] = (vib, lambda d, rpm: d * 0.8)
    out[_T + "vibration_1x_amplitude"] = (vib, lambda d, rpm: d / rss)
    out[_T + "vibration_2x_amplitude"] = (vib, lambda d, rpm: d / rss * 0.1)
    out[_T + "vibration_velocity_x"] = (vib, lambda d, rpm: d * omega(rpm) / 1000.0)
    out[_T + "vibration_velocity_y"] = (vib, lambda d, rpm: d * omega(rpm) / 1000.0 * 0.8)
    out[_T + "vibration_acceleration_x"] = (vib, lambda d, rpm: d * omega(rpm) / 1000.0 * omega(rpm) / 9.81)
    out[_T + "vibration_acceleration_y"] = (vib, lambda d, rpm: d * omega(rpm) / 1000.0 * omega(rpm) / 9.81 * 0.8)
    # condenser: the ejectors' flags and performance factor (vacuum_pump.py:541, :272-275), the condensate at the saturation temperature of
    # the condenser pressure the vacuum system ends the step with (condenser/physics.py:841 with the fit :1824-1838)
    for e in range(2):
        N = "secondary.condenser.SJE-00%d_" % (e + 1)
        out[N + "operating"] = (("cond.ej_operating_mask",), lambda m, e=e: ((np.asarray(m).astype(np.int64) >> e) & 1).astype(np.float64))
        out[N + "performance"] = (("cond.ej_nozzle_fouling[%d]" % e, "cond.ej_diffuser_fouling[%d]" % e, "cond.ej_nozzle_erosion[%d]" % e), lambda a, b, c: a * b * c)

    def cond_tsat(p_mpa):
        t = 1730.63 / (8.07131 - np.log10(np.clip(p_mpa * 10.0, 0.01, 100.0))) - 233.426
        t = np.where((p_mpa >= 0.005) & (p_mpa <= 0.01), np.clip(t, 35.0, 45.0), t)
        return np.where(p_mpa <= 0.001, 10.0, np.clip(t, 10.0, 374.0))
    out[_C + "condensate_temperature"] = (("cond.condenser_pressure",), cond_tsat)
    # pH controller (ph_control_system.py:243, :416-424, :345-347, :659): the error it stores is setpoint - measured of the same update; the
    # pH alarms compare that measurement with config.ph_alarm_low / _high (8.8 / 9.6); the consumption is the two dose rates together
    out["secondary.ph_control.ph_control_error"] = (("ph.measured_ph",), lambda m: 9.2 - m)
    out["secondary.ph_control.ph_control_alarm_low"] = (("ph.measured_ph",), lambda m: (m < 8.8).astype(np.float64))
    out["secondary.ph_control.ph_control_alarm_high"] = (("ph.measured_ph",), lambda m: (m > 9.6).astype(np.float64))
    out["secondary.ph_control.ph_control_total_consumption"] = (("ph.pending_ammonia_dose", "ph.pending_morpholine_dose"), lambda a, b: a + b)
    out["secondary.ph_control.ph_control_mode_auto"] = (("ph.controller_enabled",), lambda e: (np.asarray(e) != 0).astype(np.float64))   # :426-432: AUTO -> FAILED and enabled -> False in one statement pair
    mapped = set(reference_log_columns())
    return {k: v for k, v in out.items() if k not in mapped}


def clock_log_columns(dt: float) -> Dict[str, tuple]:
    """Log columns that count the steps taken: name -> (member labels, function), for a plant stepped with ``dt``.  The number of
    steps n is read off the secondary side's own hour counter (operating_hours += dt / 3600, secondary/__init__.py:632).
    The shared WaterChemistry is updated twice a step (feedwater/physics.py:708, secondary/__init__.py:644), each time adding its
    guess of dt in hours (water_chemistry.py:335-348) to operating_hours and last_treatment_time.  Every other counter adds the
    same dt / 60 per step: the turbine, its stage system, each of the 14 stages and 4 bearings, the vacuum system and the condenser
    are handed dt / 60 "hours" (secondary/__init__.py:568, :620; enhanced_physics.py:832, stage_system.py:329, :996,
    rotor_dynamics.py:306, vacuum_system.py:502, condenser/physics.py:849), the feedwater system adds dt / 60 itself
    (feedwater/physics.py:809) and the steam-generator system (60 dt) / 3600 (steam_generator/enhanced_physics.py:525) --
    unconditionally, tripped or not."""
    dth = dt / 3600.0 if dt > 100 else (dt / 60.0 if dt > 1 else dt)
    steps = lambda h: np.round(h * 3600.0 / dt)
    chem = (("sec.operating_hours",), lambda h: 2 * dth * steps(h))
    hours = (("sec.operating_hours",), lambda h: (dt / 60.0) * steps(h))
    out = {pre + k: chem for pre in ("secondary.water_chemistry.", _F) for k in ("water_chemistry_operating_hours", "water_chemistry_time_since_treatment")}
    stage_names = ["HP-%d" % (k + 1) for k in range(8)] + ["LP-%d" % (k + 1) for k in range(6)]
    for name in ["secondary.turbine_%s.operating_hours" % sn for sn in stage_names] + \
                [_T + k for k in ("operating_hours", "enhanced_turbine_operating_hours", "stage_system_operating_hours")] + \
                [_T + "TB-00%d_operating_hours" % (k + 1) for k in range(4)] + \
                ["secondary.condenser.vacuum_system_operating_hours", _C + "condenser_operating_hours", _F + "feedwater_operating_hours",
                 _G + "system_operating_hours"]:
        out[name] = hours
    # the pH controller's chemical alarm looks at the tank levels BEFORE this update's consumption (ph_control_system.py:246 runs
    # _update_alarms_and_trips, :262 _update_chemical_supplies): the levels the step ends with, plus what it dosed -- dose [kg/h] x dt
    # "hours" (the controller is handed the simulator's dt as hours, secondary/__init__.py:647-650) over the 1 000 / 2 000 kg tanks
    out["secondary.ph_control.ph_control_alarm_chemical"] = (
        ("ph.ammonia_tank_level", "ph.morpholine_tank_level", "ph.pending_ammonia_dose", "ph.pending_morpholine_dose"),
        lambda la, lm, da, dm: (((la + np.where(da > 0, da * dt / 1000.0 * 100.0, 0.0)) < 20.0) | ((lm + np.where(dm > 0, dm * dt / 2000.0 * 100.0, 0.0)) < 20.0)).astype(np.float64))
    return out


def output_log_columns() -> Dict[str, str]:
    """Log columns that are an OUTPUT of the step just taken, not state: PrimaryReactorPhysics.scram_activated is the result of this
    step's safety check (primary/__init__.py:255-262: True on the step the scram fires, False again on the next) -- the step's
    ``done`` column (sim.py:256).  name -> output"""
    return {"primary.reactor.scram_activated": "done", "primary.reactor.safety_scram_activated": "done"}


def history_log_columns() -> Dict[str, tuple]:
    """Log columns that are a function of other log columns' recent history: name -> (source log columns, function of their
    [samples, plants] series).  ``table()`` emits them when the log holds every step since the reset (steps 1, 2, 3 ... recorded
    with every=1): then it holds the lists the reference keeps.
      * the pH controller's RMS deviation: root mean square of the last 100 steps' |pH error| (ph_control_system.py:441-455), and its
        time in control: the share of the steps so far whose |error| was within the 0.05 deadband (:457-467; the steps are equally long);
      * the NPSH trend of the feedwater protection: its one NPSHProtection object is visited once per pump per step and keeps the last
        ten readings whoever they belong to (protection_system.py:77-85): (latest - oldest kept) / number kept."""
    def rms_of_last_100(series):
        ns = series.shape[0]
        csum = np.concatenate([np.zeros((1,) + series.shape[1:]), np.cumsum(np.square(series), axis=0)], axis=0)
        hi = np.arange(1, ns + 1); lo = np.maximum(0, hi - 100)
        return np.sqrt((csum[hi] - csum[lo]) / (hi - lo).reshape((-1,) + (1,) * (series.ndim - 1)))

    def time_in_control(series):
        inside = (np.abs(series) <= 0.05).astype(np.float64)
        n = np.arange(1, series.shape[0] + 1).reshape((-1,) + (1,) * (series.ndim - 1))
        return np.cumsum(inside, axis=0) / n * 100.0

    def npsh_trend(a, b, c, d):
        seq = np.stack([a, b, c, d], axis=1).reshape((4 * a.shape[0],) + a.shape[1:])     # reading 4 (t - 1) + k = pump k at step t
        out = np.empty_like(a)
        for t in range(a.shape[0]):
            n = 4 * (t + 1); kept = min(10, n)
            out[t] = (seq[n - 1] - seq[n - kept]) / kept
        return out
    E = "secondary.ph_control.ph_control_error"
    return {"secondary.ph_control.ph_control_deviation_rms": ((E,), rms_of_last_100),
            "secondary.ph_control.ph_control_time_in_control": ((E,), time_in_control),
            _F + "protection_npsh_trend": (tuple("secondary.feedwater_FWP-%d.npsh_available" % (k + 1) for k in range(4)), npsh_trend)}


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
        for rows, _fn in diagnostic_function_columns().values():
            wanted.update(r for r in rows if isinstance(r, str))
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
    shown = set(reference_log_columns())     # e.g. LP-6's outlet is no member, HP-1's inlet is none either: all of them come from here
    return {name: row for name, row in out.items() if name not in shown}


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


MAINTENANCE_FLAG_ACTIONS = ("oil_change", "oil_top_off", "bearing_replacement", "seal_replacement", "component_overhaul", "system_cleaning",
                            "bearing_inspection", "impeller_inspection", "impeller_replacement", "lubrication_system_check", "motor_inspection",
                            "oil_analysis", "vibration_analysis")    # FeedwaterPumpLubricationSystem.maintenance_action_flags, pump_lubrication.py:90-104


def diagnostic_function_columns() -> Dict[str, tuple]:
    """Log columns that are a function of diagnostics rows (include/npb.h NPB_DIAG_*, round 4): name -> (rows, function).
      * per pump the thirteen <action>_occurred flags of its state dict (pump_lubrication.py:642-643 sets the flag of the action the
        dispatcher is handed, :1636-1641 logs and clears them): from the row that names the action the maintenance rule carried out on
        the pump in this step (oil_top_off_occurred keeps its own row, NPB_DIAG_PUMP_OIL_TOP_OFF_OCCURRED);
      * the stage system's total extraction (stage_system.py:978): the fourteen stages' extraction flows, summed in stage order;
      * its total power (stage_system.py:976); the protection system's trip bookkeeping; the ejectors."""
    out = {}
    actions = list(_lib.MAINT_ACTIONS)
    for k in range(4):
        for a in MAINTENANCE_FLAG_ACTIONS:
            if a == "oil_top_off":
                continue
            out["secondary.feedwater_FWP-%d.%s_occurred" % (k + 1, a)] = ((_lib.DIAG_PUMP_MAINTENANCE_ACTION + k,), lambda v, code=actions.index(a) + 1: (v == code).astype(np.float64))

    def total(*flows):
        acc = 0.0 * flows[0]
        for f in flows:
            acc = acc + f
        return acc
    out[_T + "stage_system_extraction_flow"] = (tuple(_lib.DIAG_STAGE_EXTRACTION_FLOW + k for k in range(14)), total)
    ident = lambda v: v
    out[_T + "stage_system_total_power"] = ((_lib.DIAG_STAGE_SYSTEM_TOTAL_POWER,), ident)
    stage_names = ["HP-%d" % (k + 1) for k in range(8)] + ["LP-%d" % (k + 1) for k in range(6)]
    for k, sn in enumerate(stage_names):
        out["secondary.turbine_%s.extraction_flow" % sn] = ((_lib.DIAG_STAGE_EXTRACTION_FLOW + k,), ident)
    out[_T + "extraction_flow"] = ((_lib.DIAG_STAGE_EXTRACTION_FLOW + 13,), ident)     # the stage system's dict keeps the last stage's un-prefixed keys
    out[_F + "protection_active_trips_count"] = ((_lib.DIAG_FW_ACTIVE_TRIPS,), ident)
    out[_F + "protection_valid_trip_count"] = ((_lib.DIAG_FW_VALID_TRIP_COUNT,), ident)
    out[_F + "protection_emergency_feedwater"] = ((_lib.DIAG_FW_EMERGENCY_FEEDWATER,), ident)
    out[_F + "protection_steam_dump"] = ((_lib.DIAG_FW_STEAM_DUMP,), ident)
    for e in range(2):
        N = "secondary.condenser.SJE-00%d_" % (e + 1)
        out[N + "capacity"] = ((_lib.DIAG_COND_SJE_CAPACITY + e,), ident)
        out[N + "steam_flow"] = ((_lib.DIAG_COND_SJE_STEAM_FLOW + e,), ident)
        out[N + "steam_consumption"] = ((_lib.DIAG_COND_SJE_STEAM_CONSUMPTION + e,), ident)
        out[N + "compression_ratio"] = ((_lib.DIAG_COND_SJE_COMPRESSION_RATIO + e,), ident)
        out[N + "operating_hours"] = ((_lib.DIAG_COND_SJE_OPERATING_HOURS + e,), ident)
    out["secondary.condenser.vacuum_system_air_removal"] = ((_lib.DIAG_COND_AIR_REMOVAL,), ident)
    # the condenser is handed the main steam less what the stages extracted (enhanced_physics.py effective_steam_flow -> condenser/physics.py:842);
    # a source that is a string is a state member sampled with the log
    out[_C + "condensate_flow"] = (("sec.total_steam_flow",) + tuple(_lib.DIAG_STAGE_EXTRACTION_FLOW + k for k in range(14)), lambda q, *flows: q - total(*flows))
    plain = _all_diagnostic_columns()         # the first ejector's steam flow / consumption have had rows of their own since round 3
    return {k: v for k, v in out.items() if k not in plain}


def log_column_name(name: str, naming: str) -> str:
    """A rule's column name (written with the data-gen composer's ids) as a plant of the given naming logs it"""
    if naming == "default":
        return name.replace("secondary.reactor_SECONDARY-COMP-001.", "secondary.reactor.").replace("_SECONDARY-COMP-001-", "_SECONDARY-001-")
    return name


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
        # the step's own outputs that the reference logs (the scram pulse = the step's done column)
        self._done = torch.empty((self.capacity, env.n), dtype=torch.uint8, device=env.device) if self._reference_layout else None
        # how the reference names the plant's secondary-side providers: NuclearPlantSimulator's default configuration gives
        # "secondary.<subsystem>_SECONDARY-001-<X>." (and "secondary.reactor." for the secondary side itself), the data-gen composer's
        # configuration "SECONDARY-COMP-001" (auto_register.py:83-165 with each config's system_id)
        self.naming = getattr(env, "log_naming", "default")
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
        if self._done is not None:
            self._done[row].copy_(self.env._done)
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
                for name, (rows, fn) in sorted(diagnostic_function_columns().items()):
                    cols[name] = np.asarray(fn(*[(data[:, index[r], :] if isinstance(r, str) else dg[:, r, :]).reshape(-1) for r in rows]), dtype=np.float64)
            done = self._done[:len(self._times)].cpu().numpy()[:, idx].reshape(-1).astype(np.float64)
            for name, what in sorted(output_log_columns().items()):
                cols[name] = done
            if self._steps == list(range(1, ns + 1)):      # every step since the reset: the windowed columns can be formed
                for name, (sources, fn) in sorted(history_log_columns().items()):
                    if all(src in cols for src in sources):
                        cols[name] = np.asarray(fn(*[cols[src].reshape(ns, npl) for src in sources]), dtype=np.float64).reshape(-1)
            for name, value in sorted(constant_log_columns(self.env.params).items()):
                if name not in cols:
                    cols[name] = np.full(ns * npl, value)
            for name, values in sorted(getattr(self.env, "log_side_columns", {}).items()):     # per-plant values the constructor fixed (below)
                cols[name] = np.tile(np.broadcast_to(np.asarray(values, dtype=np.float64), (self.env.n,))[idx], ns)
            if self.naming == "default":
                cols = {log_column_name(k, "default"): v for k, v in cols.items()}
            return pa.table(cols)
        for f, (kind, _slot, _label, name) in enumerate(self.columns):
            v = data[:, f, :].reshape(-1)
            cols[name] = v.astype(np.int32) if kind == "i32" else v
        return pa.table(cols)

    def write_parquet(self, path: str, plants: Optional[Sequence[int]] = None) -> None:
        import pyarrow.parquet as pq
        pq.write_table(self.table(plants), path)
