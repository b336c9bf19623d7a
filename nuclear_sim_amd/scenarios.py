"""Initial conditions of the data-generation scenarios, as columns (BASELINE config 4).

The reference builds one plant at a time: ``ComprehensiveComposer.compose_action_test_scenario``
(data_gen/config_engine/composers/comprehensive_composer.py:75-295) deep-copies a YAML template, overwrites
the target subsystem's ``initial_conditions`` with a catalog entry, optionally randomised per seed
(initial_conditions/randomization_utils.py:799-1017), and the constructors then translate those
dictionaries into object state (feedwater/physics.py:185-437, feedwater/pump_system.py:1177-1233).
For 10^5 - 10^6 plants that dictionary shuffling dominates set-up, so this module produces the same
state directly as struct-of-arrays columns for ``BatchedPlantEnv.set_fields``:

    fields = action_test_fields("oil_top_off", seeds)        # {column: array [n], or [1] where every plant has the same value}
    env = BatchedPlantEnv(n, dt=5.0, noise_enabled=True, maintenance=True); env.set_fields(fields)

All ten actions the composer maps to the feedwater subsystem are covered (``FEEDWATER_ACTIONS``); their catalog
entries, scenario tables and the template's ``initial_conditions`` section are data (``feedwater_catalog.json``,
dumped from the reference by the harness script named in DESIGN.md section 6), the logic that consumes them --
the composer's filter, the scenario-based randomiser, the constructors' mapping onto state -- is restated here.
Every column this module produces is checked against the reference's own constructor, per action, for the
catalog entry and several seeds (tests/golden/ic_*.npz, tests/test_scenarios.py).
"""
from __future__ import annotations

import copy
import json
import os
import random
import re
from typing import Dict, Optional, Sequence

import numpy as np

with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "feedwater_catalog.json")) as _fh:
    _CATALOG = json.load(_fh)
FEEDWATER_ACTIONS = tuple(_CATALOG["conditions"])
with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "action_state_deltas.json")) as _fh:
    _DELTAS = json.load(_fh)
ALL_ACTIONS = tuple(_DELTAS["actions"])      # every action of the composer's action -> subsystem map that the reference can build

NUM_PUMPS = 4
NUM_SG = 3
PUMP_RATED_FLOW = 500.0          # feedwater pump config.rated_flow
MIN_PUMPS_REQUIRED = 3           # FeedwaterPumpSystem.minimum_pumps_required

# FeedwaterInitialConditions defaults that the state mapping below needs (feedwater/config.py)
FEEDWATER_IC_DEFAULTS = {
    "sg_steam_flows": [500.0, 500.0, 500.0],
    "pump_oil_levels": [100.0, 100.0, 100.0, 100.0],
    "pump_oil_contamination": 5.0,
    "motor_bearing_wear": [0.0, 0.0, 0.0, 0.0],
    "pump_bearing_wear": [0.0, 0.0, 0.0, 0.0],
    "thrust_bearing_wear": [0.0, 0.0, 0.0, 0.0],
    "seal_face_wear": [0.3, 0.3, 0.3, 0.3],
    "seal_leakage_rate": [0.001, 0.001, 0.001, 0.0],
    "impeller_cavitation_damage": [0.1, 0.1, 0.1, 0.1],
    "motor_temperature": [70.0, 70.0, 70.0, 70.0],
    # keys beyond oil_top_off's: None = the constructor leaves the construction-time value alone
    "pump_oil_water_content": None, "pump_oil_acid_number": None, "oil_temperature": None,
    "suction_pressure": None, "discharge_pressure": None, "npsh_available": None, "cavitation_intensity": None,
}

# What the comprehensive template (data_gen/config_engine/templates/nuclear_plant_comprehensive_config.yaml)
# sets differently from the dataclass defaults, for the keys that reach carried state.
ACTION_TEST_TEMPLATE = {
    "feedwater": {
        "sg_steam_flows": [450.0, 450.0, 450.0],
        "seal_leakage_rate": [0.0, 0.0, 0.0, 0.0],
        "impeller_cavitation_damage": [0.0, 0.0, 0.0, 0.0],
    },
    "steam_generator": {"sg_steam_flows": [450.0, 450.0, 450.0]},
    "turbine": {"rotor_temperature": 350.0, "bearing_temperatures": [70.0, 68.0, 72.0, 69.0]},
}

# FEEDWATER_CONDITIONS["oil_top_off"]  initial_conditions/feedwater_conditions.py:81-96
OIL_TOP_OFF_CONDITIONS = {
    "pump_oil_levels": [60.3, 98.0, 98.0, 100.0],
    "seal_face_wear": [12.0, 0.1, 0.1, 0.1],
    "pump_oil_contamination": 8.0,
    "motor_bearing_wear": [1.0, 0.1, 0.1, 0.0],
    "pump_bearing_wear": [1.0, 0.1, 0.1, 0.0],
    "thrust_bearing_wear": [0.5, 0.1, 0.1, 0.0],
    "motor_temperature": [70.0, 30.0, 30.0, 25.0],
}
# ACTION_SCENARIOS["oil_top_off"]  randomization_utils.py:770-797: (probability, low, high) of pump_oil_levels[0]
OIL_TOP_OFF_SCENARIOS = [(0.3, 59.2, 59.6), (0.4, 60.8, 61.2), (0.3, 61.5, 63.0)]


def randomized_oil_top_off_levels(seeds: Sequence[int]) -> np.ndarray:
    """pump_oil_levels[n, 4] of ``get_randomized_feedwater_conditions("oil_top_off", seed)`` for every seed
    (randomization_utils.py:799-841, 844-895, 897-917).  The reference draws from the stdlib generator seeded
    with the scenario seed: one ``random()`` picks the weighted scenario, one ``uniform()`` the first pump's
    level, and "preserve_pattern" scales all four levels by the same factor.  The generator's streams come from
    libnpb.so for all seeds at once (include/npb_seeds.h; checked against random.Random itself)."""
    return np.ascontiguousarray(randomized_conditions_columns("oil_top_off", seeds)["pump_oil_levels"])


def _col(v, n):
    a = np.asarray(v, dtype=np.float64)
    return np.broadcast_to(a, (n,) + a.shape[-1:]) if a.ndim <= 1 and a.shape != (n,) else a


def feedwater_fields(ic: Dict[str, object], n: int, lubrication_effectiveness: float) -> Dict[object, np.ndarray]:
    """EnhancedFeedwaterPhysics._apply_initial_conditions (feedwater/physics.py:185-437) followed by
    FeedwaterPumpSystem._initialize_pumps (pump_system.py:1177-1233), for the initial-condition keys above,
    on arrays.  ``ic`` values are scalars, per-pump lists, or arrays [n, 4]; ``lubrication_effectiveness`` is
    the value the lubrication system computed at construction, before any initial condition is applied
    (pump_lubrication.py:204-222) -- it enters the performance factors."""
    g = dict(FEEDWATER_IC_DEFAULTS); g.update(ic)
    # a value that is the same for every plant stays ONE row here and ONE element in the result (shape (1,) instead of (n,)):
    # set_fields fills such a column on the device, and a batch of 10^5 plants does not spend its set-up time copying constants
    def per_pump(k):
        a = np.asarray(g[k], dtype=np.float64)
        return np.broadcast_to(a, (1, NUM_PUMPS)) if a.ndim <= 1 else a
    def scalar(k):                                                                           # one value per plant
        a = np.asarray(g[k], dtype=np.float64)
        return a.reshape(1) if a.ndim == 0 else a
    lubrication_effectiveness = np.asarray(lubrication_effectiveness, dtype=np.float64)
    f: Dict[object, np.ndarray] = {}
    motor, pumpb, thrust, seals = (per_pump(k) for k in ("motor_bearing_wear", "pump_bearing_wear", "thrust_bearing_wear", "seal_face_wear"))
    # _calculate_pump_performance_factors(cavitation_damage=0.0)  pump_lubrication.py:1412-1478
    bearing_efficiency_loss = ((motor / 100.0) * 0.01 + (pumpb / 100.0) * 0.015 + (thrust / 100.0) * 0.02)
    seal_efficiency_loss = (seals / 100.0) * 0.01
    lubrication_efficiency_loss = (1.0 - lubrication_effectiveness) * 0.02
    total_efficiency_loss = (bearing_efficiency_loss + seal_efficiency_loss + lubrication_efficiency_loss + 0.0 + 0.0)
    total_flow_loss = (0.0 + 0.0 + bearing_efficiency_loss * 0.3)
    efficiency_degradation = np.minimum(50.0, total_efficiency_loss * 100.0)
    flow_degradation = np.minimum(50.0, total_flow_loss * 100.0)
    vibration_increase = (motor + pumpb + thrust) * 0.1 + 0.0
    # _initialize_pumps: demand from the SG steam flows, shared over the three duty pumps, speed from the pump law
    flow_factor = np.maximum(0.5, 1.0 - flow_degradation / 100.0)                     # pump_lubrication.py:230-233
    degradation_factor = np.maximum(0.5, (0.0 + flow_factor[:, 0] + flow_factor[:, 1] + flow_factor[:, 2] + flow_factor[:, 3]) / NUM_PUMPS)
    sg_flows = np.asarray(g["sg_steam_flows"], dtype=np.float64)
    sg_flows = np.broadcast_to(sg_flows, (1, NUM_SG)) if sg_flows.ndim <= 1 else sg_flows
    total_steam_flow = 0.0 + sg_flows[:, 0] + sg_flows[:, 1] + sg_flows[:, 2]
    flow_per_pump = ((total_steam_flow * 1.02) / MIN_PUMPS_REQUIRED) / degradation_factor
    for k in range(NUM_PUMPS):
        f[("pump.oil_level", k)] = per_pump("pump_oil_levels")[:, k]
        f[("pump.oil_contamination", k)] = scalar("pump_oil_contamination")
        f[("pump.wear_motor_bearings", k)] = motor[:, k]
        f[("pump.wear_pump_bearings", k)] = pumpb[:, k]
        f[("pump.wear_thrust_bearing", k)] = thrust[:, k]
        f[("pump.wear_mechanical_seals", k)] = seals[:, k]
        f[("pump.seal_leakage_rate", k)] = per_pump("seal_leakage_rate")[:, k]
        f[("pump.efficiency_degradation", k)] = efficiency_degradation[:, k]
        f[("pump.flow_degradation", k)] = flow_degradation[:, k]
        f[("pump.vibration_increase", k)] = vibration_increase[:, k]
        f[("pump.cavitation_damage", k)] = per_pump("impeller_cavitation_damage")[:, k] * 10.0
        f[("pump.motor_temperature", k)] = per_pump("motor_temperature")[:, k]
        # lubrication oil state and pump hydraulics (physics.py:236-262, 283-305): scalars apply to every pump
        for key, col in (("pump_oil_water_content", "pump.oil_moisture"), ("pump_oil_acid_number", "pump.oil_acidity"),
                         ("oil_temperature", "pump.oil_temperature"), ("suction_pressure", "pump.suction_pressure"),
                         ("discharge_pressure", "pump.discharge_pressure")):
            if g.get(key) is not None:
                f[(col, k)] = scalar(key)
        for key, col in (("npsh_available", "pump.npsh_available"), ("cavitation_intensity", "pump.cavitation_intensity")):
            if g.get(key) is not None:
                f[(col, k)] = per_pump(key)[:, k]
        if k < MIN_PUMPS_REQUIRED:
            f[("pump.flow_demand", k)] = np.clip(flow_per_pump, 0.0, PUMP_RATED_FLOW * 1.2)    # set_flow_demand :422-447
            required_speed = np.sqrt(flow_per_pump / (PUMP_RATED_FLOW * flow_factor[:, k])) * 100.0
            speed = np.minimum(100.0, np.maximum(30.0, required_speed))                         # _calculate_safe_speed :1347-1368
            f[("pump.speed_percent", k)] = speed
            f[("pump.speed_setpoint", k)] = speed
    f["fw.total_flow_rate"] = total_steam_flow                                                    # physics.py:196-197
    icd = per_pump("impeller_cavitation_damage")
    f["fw.cav_accumulated_damage"] = ((0.0 + icd[:, 0] + icd[:, 1] + icd[:, 2] + icd[:, 3]) / NUM_PUMPS) * 10.0
    out = {}
    for k, v in f.items():
        v = np.ascontiguousarray(v, dtype=np.float64)
        assert v.shape in ((1,), (n,)), (k, v.shape)
        out[k] = v
    return out


def feedwater_reset_fields(ic: Dict[str, object], n: int, lubrication_effectiveness, start_at_steady_state: bool) -> Dict[object, np.ndarray]:
    """The columns EnhancedFeedwaterPhysics.reset (feedwater/physics.py:1286-1323) puts back when it re-applies the
    configured initial conditions after NuclearPlantSimulator.reset(): ``feedwater_fields`` evaluated with the
    lubrication effectiveness the history left (the lubrication system is never reset; array [n, 4] or scalar),
    minus what the reset decides itself -- the system flow is the sum of the 555 kg/s the reset has just written, and
    with ``start_at_steady_state`` the pump hydraulics, speeds, demands, cavitation state and motor temperatures are
    force-set afterwards by _initialize_feedwater_system_to_steady_state (secondary/__init__.py:1247-1357)."""
    f = feedwater_fields(ic, n, lubrication_effectiveness)
    f.pop("fw.total_flow_rate", None)
    if start_at_steady_state:
        forced = ("pump.speed_percent", "pump.speed_setpoint", "pump.flow_demand", "pump.cavitation_damage", "pump.motor_temperature",
                  "pump.suction_pressure", "pump.discharge_pressure", "pump.npsh_available", "pump.cavitation_intensity")
        f = {k: v for k, v in f.items() if not (isinstance(k, tuple) and k[0] in forced)}
    return f


def catalog_conditions(action: str) -> Dict[str, object]:
    """FEEDWATER_CONDITIONS[action] (numbers and lists only)  initial_conditions/feedwater_conditions.py"""
    if action not in _CATALOG["conditions"]:
        raise NotImplementedError("%r is not an action the composer maps to the feedwater subsystem (%s)"
                                  % (action, ", ".join(FEEDWATER_ACTIONS)))
    return copy.deepcopy(_CATALOG["conditions"][action])


def _apply_to_array(base, value, handling):
    """apply_scenario_to_array_parameter  randomization_utils.py:880-895"""
    if not base:
        return base
    if handling == "preserve_pattern":
        if base[0] != 0:
            scale_factor = value / base[0]
            return [v * scale_factor for v in base]
        return [value] + base[1:]
    if handling == "first_element_only":
        return [value] + base[1:]
    return [value] * len(base)


def _is_number(v) -> bool:
    return isinstance(v, (int, float)) and not isinstance(v, bool)


def _jitter(conditions: Dict[str, object], rules: Dict[str, dict], default_scale: float, seed: Optional[int]) -> Dict[str, object]:
    """add_randomness_to_conditions  randomization_utils.py:13-122: every numeric leaf of the entry -- nested dictionaries
    included, in the entry's own key order -- is scaled by 1 + U(-s, s), s and the clamp from the parameter's rule or the
    default scale; one draw per leaf from numpy's legacy generator seeded with the scenario seed (the same generator
    here: RandomState(seed) is that stream)."""
    rs = np.random.RandomState(seed)

    def single(value, rule):                               # _apply_single_rule :73-89
        new = value * (1.0 + rs.uniform(-rule.get("scale_factor", 0.1), rule.get("scale_factor", 0.1)))
        if rule.get("min_value") is not None:
            new = max(new, rule["min_value"])
        if rule.get("max_value") is not None:
            new = min(new, rule["max_value"])
        return new

    def default(value):                                    # _apply_default_scaling :91-94
        return value * (1.0 + rs.uniform(-default_scale, default_scale))

    def array(arr):                                        # _randomize_array :96-122 with an empty rule
        if not arr or not isinstance(arr[0], (int, float)):
            return arr
        return [default(v) for v in arr]

    def rec(obj):                                          # _randomize_recursive :46-64
        if isinstance(obj, dict):
            for key, value in obj.items():
                if key in rules:
                    obj[key] = [single(v, rules[key]) for v in value] if isinstance(value, list) else single(value, rules[key])
                elif _is_number(value):
                    obj[key] = default(value)
                elif isinstance(value, list):
                    obj[key] = array(value)
                else:
                    rec(value)
        elif isinstance(obj, list):
            for item in obj:
                rec(item)

    out = copy.deepcopy(conditions)
    rec(out)
    return out


def _violates(conditions: Dict[str, object], rules: Dict[str, dict]) -> bool:
    """validate_safety_limits :214-262, for rule tables whose entries are all safety limits"""
    for key, value in conditions.items():
        if key in rules:
            lim, less = rules[key]["safety_limit"], rules[key].get("safety_direction", "greater_than") == "less_than"
            for v in (value if isinstance(value, list) else [value]):
                if _is_number(v) and ((v < lim) if less else (v > lim)):
                    return True
        elif isinstance(value, dict) and _violates(value, rules):
            return True
    return False


def randomized_conditions(action: str, seed: Optional[int]) -> Dict[str, object]:
    """get_randomized_feedwater_conditions(action, seed) for the actions that have a scenario table
    (randomization_utils.py:799-842 scenario pick and per-parameter draws, :897-941 array handling).  The reference
    seeds both the stdlib generator and numpy's legacy global generator with the scenario seed; uniform parameters
    draw from the former, normal ones from the latter (clipped to the range), in the table's parameter order --
    the same two generators are used here, so the draws are the reference's by construction."""
    base = catalog_conditions(action)
    table = _CATALOG["scenarios"].get(action)
    jit = _CATALOG["jitter"]
    if not table:      # no scenario table: the generic jitter over the complete entry (get_scenario_based_conditions :816-818)
        out = _jitter(jit["full_conditions"][action], jit["default_rules"], jit["default_scale"], seed)
        return base if _violates(out, jit["feedwater_safety_rules"]) else out
    r = random.Random(seed)
    nr = np.random.RandomState(seed)
    total = sum(sc["probability"] for sc in table)
    rand_val, cumulative, pick = r.random(), 0.0, table[-1]
    for sc in table:
        cumulative += sc["probability"] / total
        if rand_val <= cumulative:
            pick = sc
            break
    out = copy.deepcopy(base)
    for name, cfg in pick["parameters"].items():
        if name not in out:
            continue
        lo, hi = cfg["range"]
        if cfg["distribution"] == "normal":
            value = float(nr.normal((lo + hi) / 2, (hi - lo) / 4))
            value = max(lo, min(hi, value))
        else:
            value = r.uniform(lo, hi)
        if isinstance(out[name], list):
            out[name] = _apply_to_array(out[name], value, cfg["array_handling"] or "preserve_pattern")
        else:
            out[name] = value
    for name, (handling, default) in _CATALOG["array_parameters"].items():
        if name in out and isinstance(out[name], (int, float)):
            out[name] = _apply_to_array(base.get(name, default), out[name], handling)
    # a draw beyond a safety limit makes the reference raise, and the composer then uses the plain catalog entry
    # (get_randomized_feedwater_conditions :962-965, comprehensive_composer.py:251-254)
    return base if _violates(out, jit["feedwater_safety_rules"]) else out


# ---------------------------------------------------------------------------------------------------------------------
# The same randomisers for a whole array of seeds ("SoA IC arrays for 10^5 - 10^6 plants").  The functions
# above follow the reference one plant at a time and are kept as the readable statement (and as the check of what follows,
# tests/test_scenarios.py); below, the two generators' streams come from libnpb.so for all seeds at once
# (include/npb_seeds.h) and the control flow of the functions above is walked ONCE per scenario with whole columns in place
# of the scalars: which draws are consumed, and in which order, depends only on the catalog entry's structure, never on a
# drawn value, so every seed of a scenario takes the same path.
def _seed_streams(kind: str, seeds: np.ndarray, k: int) -> np.ndarray:
    """[n, k]: the first k values of random.Random(seed).random() ("py"), numpy.random.RandomState(seed).random_sample()
    ("np") or .standard_normal() ("gauss") for every seed"""
    import ctypes
    from . import _lib
    L = _lib.load()
    seeds = np.ascontiguousarray(seeds, dtype=np.int64)
    out = np.empty((len(seeds), max(k, 1)))
    if k <= 0 or len(seeds) == 0:
        return out[:, :0]
    fn = getattr(L, {"py": "npb_seed_py_random", "np": "npb_seed_np_random", "gauss": "npb_seed_np_gauss"}[kind])
    fn.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
    if fn(seeds.ctypes.data, len(seeds), int(k), out.ctypes.data) != 0:
        raise ValueError("scenario seeds must be non-negative integers (below 2**32 for numpy's generator)")
    return out


def _is_value(v) -> bool:
    """a number, or a column of them"""
    return _is_number(v) or (isinstance(v, np.ndarray) and v.ndim == 1)


def _columns(struct, m: int):
    """a conditions dictionary whose leaves are numbers, columns [m] or lists of either -> {key: array [m] / [m, len]}
    (numeric entries only: the only ones that can reach plant state)"""
    out = {}
    for k, v in struct.items():
        if _is_value(v):
            out[k] = np.broadcast_to(np.asarray(v, dtype=np.float64), (m,))
        elif isinstance(v, list) and v and all(_is_value(e) for e in v):
            out[k] = np.stack([np.broadcast_to(np.asarray(e, dtype=np.float64), (m,)) for e in v], axis=1)
    return out


def _violates_columns(conditions: Dict[str, object], rules: Dict[str, dict], m: int) -> np.ndarray:
    """_violates for columns: bool [m]"""
    bad = np.zeros(m, dtype=bool)
    for key, value in conditions.items():
        if key in rules:
            lim, less = rules[key]["safety_limit"], rules[key].get("safety_direction", "greater_than") == "less_than"
            for v in (value if isinstance(value, list) else [value]):
                if _is_value(v):
                    bad |= (np.asarray(v) < lim) if less else (np.asarray(v) > lim)
        elif isinstance(value, dict):
            bad |= _violates_columns(value, rules, m)
    return bad


def _jitter_columns(conditions: Dict[str, object], rules: Dict[str, dict], default_scale: float, seeds: np.ndarray) -> Dict[str, object]:
    """_jitter for every seed: the same traversal, each numeric leaf's draw being column j of the numpy stream"""
    def count(obj):                                        # how many draws the traversal consumes
        c = 0
        if isinstance(obj, dict):
            for key, value in obj.items():
                if key in rules:
                    c += len(value) if isinstance(value, list) else 1
                elif _is_number(value):
                    c += 1
                elif isinstance(value, list):
                    c += len(value) if (value and isinstance(value[0], (int, float))) else 0
                else:
                    c += count(value)
        elif isinstance(obj, list):
            for item in obj:
                c += count(item)
        return c

    U = _seed_streams("np", seeds, count(conditions))
    cursor = [0]

    def uniform(scale):                                    # RandomState.uniform(-scale, scale) = low + (high - low) * u
        u = U[:, cursor[0]]; cursor[0] += 1
        return -scale + (scale - -scale) * u

    def single(value, rule):
        new = value * (1.0 + uniform(rule.get("scale_factor", 0.1)))
        if rule.get("min_value") is not None:
            new = np.maximum(new, rule["min_value"])
        if rule.get("max_value") is not None:
            new = np.minimum(new, rule["max_value"])
        return new

    def default(value):
        return value * (1.0 + uniform(default_scale))

    def rec(obj):
        if isinstance(obj, dict):
            for key, value in obj.items():
                if key in rules:
                    obj[key] = [single(v, rules[key]) for v in value] if isinstance(value, list) else single(value, rules[key])
                elif _is_number(value):
                    obj[key] = default(value)
                elif isinstance(value, list):
                    if value and isinstance(value[0], (int, float)):
                        obj[key] = [default(v) for v in value]
                else:
                    rec(value)
        elif isinstance(obj, list):
            for item in obj:
                rec(item)

    out = copy.deepcopy(conditions)
    rec(out)
    assert cursor[0] == U.shape[1]
    return out


def randomized_conditions_columns(action: str, seeds: Sequence[int], keys: Optional[Sequence[str]] = None) -> Dict[str, np.ndarray]:
    """``randomized_conditions(action, seed)`` for every seed at once: {parameter: array [n] or [n, len]} of the numeric
    entries; a parameter that no draw of the action's table reaches comes back as the catalog entry's number or [len] array,
    the same for every seed (``keys``: only these -- every draw is still consumed in its turn, but a parameter nobody asked for and no safety
    limit looks at is not evaluated).  Bit-identical to the per-seed function (tests/test_scenarios.py), at about a million
    seeds per second."""
    seeds = np.ascontiguousarray(seeds, dtype=np.int64)
    n = len(seeds)
    base = catalog_conditions(action)
    wanted = None if keys is None else set(keys) | set(_CATALOG["jitter"]["feedwater_safety_rules"])
    base_cols = _columns(base if wanted is None else {k: v for k, v in base.items() if k in wanted}, n)
    table = _CATALOG["scenarios"].get(action)
    jit = _CATALOG["jitter"]
    if not table:
        out = _jitter_columns(jit["full_conditions"][action], jit["default_rules"], jit["default_scale"], seeds)
        bad = _violates_columns(out, jit["feedwater_safety_rules"], n)
        cols = _columns(out, n)
    else:
        n_uniform = max(sum(1 for nm, c in sc["parameters"].items() if nm in base and c["distribution"] != "normal") for sc in table)
        n_normal = max(sum(1 for nm, c in sc["parameters"].items() if nm in base and c["distribution"] == "normal") for sc in table)
        U = _seed_streams("py", seeds, 1 + n_uniform)
        G = _seed_streams("gauss", seeds, n_normal)
        total = sum(sc["probability"] for sc in table)
        pick = np.full(n, len(table) - 1, dtype=np.int64)
        undecided = np.ones(n, dtype=bool)
        cumulative = 0.0
        for s_i, sc in enumerate(table):
            cumulative += sc["probability"] / total
            hit = undecided & (U[:, 0] <= cumulative)
            pick[hit] = s_i
            undecided &= ~hit
        touched = set()
        for sc in table:
            touched |= {nm for nm in sc["parameters"] if nm in base}
        touched |= {nm for nm in _CATALOG["array_parameters"] if nm in base and _is_number(base[nm])}
        # parameters no scenario draws keep the catalog entry's value for every seed: they stay what they are (a number or a
        # list), and only the drawn ones become columns
        cols = {k: np.array(v) for k, v in base_cols.items() if k in touched}
        bad = np.zeros(n, dtype=bool)
        for s_i, sc in enumerate(table):
            rows = np.nonzero(pick == s_i)[0]
            if not len(rows):
                continue
            out = copy.deepcopy(base)
            iu, ig = 1, 0
            for name, cfg in sc["parameters"].items():
                if name not in out:
                    continue
                lo, hi = cfg["range"]
                skip = wanted is not None and name not in wanted
                if cfg["distribution"] == "normal":
                    ig += 1
                    if skip:
                        continue
                    value = (lo + hi) / 2 + ((hi - lo) / 4) * G[rows, ig - 1]          # RandomState.normal: loc + scale * gauss
                    value = np.maximum(lo, np.minimum(hi, value))
                else:
                    iu += 1
                    if skip:
                        continue
                    value = lo + (hi - lo) * U[rows, iu - 1]                           # Random.uniform: a + (b - a) * random()
                if isinstance(out[name], list):
                    out[name] = _apply_to_array(out[name], value, cfg["array_handling"] or "preserve_pattern")
                else:
                    out[name] = value
            for name, (handling, default) in _CATALOG["array_parameters"].items():
                if name in out and _is_value(out[name]):
                    out[name] = _apply_to_array(base.get(name, default), out[name], handling)
            bad[rows] = _violates_columns(out, jit["feedwater_safety_rules"], len(rows))
            for k, v in _columns({k: v for k, v in out.items() if k in touched and (wanted is None or k in wanted)}, len(rows)).items():
                if k not in cols or cols[k].shape[1:] != v.shape[1:]:
                    cols[k] = np.array(np.broadcast_to(np.zeros(1), (n,) + v.shape[1:]))   # (a parameter the entry holds as a scalar, made an array)
                    if k in base_cols and base_cols[k].shape[1:] == v.shape[1:]:
                        cols[k][...] = base_cols[k]
                cols[k][rows] = v
        for k, v in base.items():
            if k not in cols and (wanted is None or k in wanted) and (_is_number(v) or (isinstance(v, list) and v and all(_is_number(e) for e in v))):
                cols[k] = np.asarray(v, dtype=np.float64)          # shape () or (len,): the same for every seed
    if bad.any():       # a draw beyond a safety limit: the reference raises and the composer uses the plain catalog entry
        for k in cols:
            if k in base_cols and base_cols[k].shape == cols[k].shape:
                cols[k] = np.where(bad.reshape((n,) + (1,) * (cols[k].ndim - 1)), base_cols[k], cols[k])
    return cols


def composed_feedwater_ic_columns(cols: Dict[str, np.ndarray]) -> Dict[str, object]:
    """composed_feedwater_ic for columns"""
    tpl = _CATALOG["template_ic"]["feedwater"]
    ic = {k: copy.deepcopy(tpl[k]) for k in FEEDWATER_IC_DEFAULTS if k in tpl}
    ic["sg_steam_flows"] = list(ACTION_TEST_TEMPLATE["feedwater"]["sg_steam_flows"])
    for k, v in cols.items():
        if k in tpl and k in FEEDWATER_IC_DEFAULTS:
            ic[k] = v
    return ic


def composed_feedwater_ic(conditions: Dict[str, object]) -> Dict[str, object]:
    """The feedwater initial_conditions the composer hands to the simulator: the template's section with the catalog
    parameters that exist in it overwritten (comprehensive_composer.py:284-293), reduced to the keys that reach state."""
    tpl = _CATALOG["template_ic"]["feedwater"]
    ic = {k: copy.deepcopy(tpl[k]) for k in FEEDWATER_IC_DEFAULTS if k in tpl}
    ic["sg_steam_flows"] = list(ACTION_TEST_TEMPLATE["feedwater"]["sg_steam_flows"])   # what the composed configuration hands physics.py:196
    for k, v in conditions.items():
        if k in tpl and k in FEEDWATER_IC_DEFAULTS:
            ic[k] = v
    return ic


def _stack(ics: Sequence[Dict[str, object]]) -> Dict[str, object]:
    """per-plant dictionaries -> one dictionary of arrays [n] / [n, 4]"""
    out = {}
    for k in ics[0]:
        out[k] = None if ics[0][k] is None else np.asarray([ic[k] for ic in ics], dtype=np.float64)
    return out


def _label_key(label: str):
    """"pump[2].oil_level" / "sg[0].tsp_magnetite[3]" / "cond.scale_thickness" -> the key form of set_fields"""
    m = re.match(r"^(\w+)(?:\[(\d+)\])?\.(\w+)(?:\[(\d+)\])?$", label)
    sec, inst, name, k = m.group(1), m.group(2), m.group(3), m.group(4)
    if inst is None and k is None:
        return "%s.%s" % (sec, name)
    if k is None:
        return ("%s.%s" % (sec, name), int(inst))           # the same key form feedwater_fields / the template entries use
    return ("%s.%s" % (sec, name), int(inst or 0), int(k))


# steam-generator initial conditions that are plain per-SG state (steam_generator/system.py _apply_initial_conditions):
_SG_DIRECT = {"sg_levels": "sg.water_level", "sg_pressures": "sg.secondary_pressure", "sg_temperatures": "sg.secondary_temperature",
              "sg_steam_qualities": "sg.steam_quality", "sg_steam_flows": "sg.steam_flow_rate"}
_SG_AVERAGES = {"sg_pressures": "sec.sg_avg_pressure", "sg_temperatures": "sec.sg_avg_temperature", "sg_steam_qualities": "sec.sg_avg_quality"}


NUM_TSP = 7


def _sg_deposit_fields(tsp_thickness: Optional[np.ndarray], scale_thickness: Optional[np.ndarray]) -> Dict[object, np.ndarray]:
    """EnhancedSteamGeneratorPhysics._apply_tsp_fouling_initial_conditions / _apply_scale_initial_conditions
    (steam_generator/enhanced_physics.py:156-226) on arrays [n, 3]: the total TSP deposit thickness spread over the seven
    support plates (lower plates foul more) and four species, then the flow restriction and heat-transfer degradation
    the fouling model derives from them (tsp_fouling_model.py:302-367); tube scale with its composition and thermal
    resistance (tube_interior_fouling.py:190-243).  A thickness of zero leaves the construction-time values."""
    f: Dict[object, np.ndarray] = {}
    if tsp_thickness is not None:
        level_factors = [1.0 + 0.3 * (NUM_TSP - level - 1) / (NUM_TSP - 1) for level in range(NUM_TSP)]
        factor_sum = sum(level_factors)
        hole = 0.023 * 1000.0
        for i in range(NUM_SG):
            t = tsp_thickness[:, i]
            on = t > 0
            total_restriction = np.zeros_like(t)
            for level in range(NUM_TSP):
                lt = t * (level_factors[level] / factor_sum)
                parts = (lt * 0.50, lt * 0.20, lt * 0.25, lt * 0.05)
                for name, v in zip(("tsp_magnetite", "tsp_copper", "tsp_silica", "tsp_biological"), parts):
                    f[("sg." + name, i, level)] = np.where(on, v, 0.0)
                total = parts[0] + parts[1] + parts[2] + parts[3]
                eff = np.maximum(hole - 2.0 * total, hole * 0.1)
                area_ratio = (np.pi * (eff / 2.0) ** 2) / (np.pi * (hole / 2.0) ** 2)
                total_restriction = total_restriction + (1.0 - area_ratio)
            ff = total_restriction / NUM_TSP
            pdr = (1.0 / np.maximum(1.0 - ff, 0.1)) ** 2
            ht = np.minimum((ff ** 1.5 + ff * 0.3) * 0.6, 0.9)
            f[("sg.tsp_fouling_fraction", i)] = np.where(on, ff, 0.0)
            f[("sg.tsp_pressure_drop_ratio", i)] = np.where(on, pdr, 1.0)
            f[("sg.tsp_ht_degradation", i)] = np.where(on, ht, 0.0)
    if scale_thickness is not None:
        for i in range(NUM_SG):
            t = scale_thickness[:, i]
            on = t > 0
            iron, crud, corr = t * 0.6, t * 0.3, t * 0.1
            tot = np.maximum(t, 0.001)
            k = np.maximum((iron / tot) * .5 + (crud / tot) * 0.15 + (corr / tot) * 0.3, 0.05)
            r = (t / 1000.0) / k + 1e-5 + (t / 1000.0) * 0.001
            for name, v in (("scale_thickness", t), ("scale_iron_oxide", iron), ("scale_crud", crud), ("scale_corrosion", corr),
                            ("scale_thermal_resistance", r)):
                f[("sg." + name, i)] = np.where(on, v, 0.0)
    return f


def _sg_randomized_fields(action: str, seeds: Sequence[int]) -> Dict[object, np.ndarray]:
    """get_randomized_sg_conditions(action, seed) (steam_generator_conditions.py:191-300: the generic jitter with the
    steam-generator rule table) -> the per-SG state members its parameters reach, for the parameters that exist in the
    template's steam_generator section."""
    jit = _CATALOG["jitter"]
    tpl = _CATALOG["template_ic"]["steam_generator"]
    seeds = np.ascontiguousarray(seeds, dtype=np.int64)
    cond = _columns(_jitter_columns(jit["full_conditions"][action], jit["sg_rules"], jit["sg_scale"], seeds), len(seeds))
    vals = {key: cond[key] for key in list(_SG_DIRECT) + ["tsp_fouling_thicknesses", "scale_thicknesses"] if key in cond and key in tpl}
    f: Dict[object, np.ndarray] = _sg_deposit_fields(
        np.asarray(vals.pop("tsp_fouling_thicknesses"), dtype=np.float64) if "tsp_fouling_thicknesses" in vals else None,
        np.asarray(vals.pop("scale_thicknesses"), dtype=np.float64) if "scale_thicknesses" in vals else None)
    for key, rows in vals.items():
        a = np.asarray(rows, dtype=np.float64)
        for k in range(NUM_SG):
            f[(_SG_DIRECT[key], k)] = a[:, k]
        if key in _SG_AVERAGES:
            f[_SG_AVERAGES[key]] = (0.0 + a[:, 0] + a[:, 1] + a[:, 2]) / NUM_SG
    return f


def log_side_columns(action: Optional[str] = None, seeds: Optional[Sequence[int]] = None, randomize: bool = True) -> Dict[str, np.ndarray]:
    """Per-plant values of the reference's state log that the CONSTRUCTOR fixes from the configured initial conditions and no step
    moves: the feedwater diagnostics' wear tracker takes 100 x the mean configured impeller / bearing / seal-face wear
    (feedwater/physics.py:400-412) and is never updated again (its update is the legacy branch of performance_monitoring.py:493-503,
    not taken by pumps that own a lubrication system).  ``action`` None: NuclearPlantSimulator's default configuration
    (FeedwaterInitialConditions: impeller_wear and seal_face_wear 0.3 each, no bearing_wear attribute, feedwater/config.py:270-276)."""
    name = "secondary.feedwater_SECONDARY-COMP-001-FW.diagnostics_total_wear"
    if action is None:
        return {name: np.full(1, 0.3 * 100.0 + 0.3 * 100.0)}
    tpl = _CATALOG["template_ic"]["feedwater"]
    keys = [k for k in ("impeller_wear", "bearing_wear", "seal_face_wear") if tpl.get(k) is not None]
    info = _DELTAS["actions"].get(action, {})
    if info.get("subsystem") == "feedwater":
        cond = randomized_conditions_columns(action, seeds, keys=keys) if randomize else catalog_conditions(action)
    else:
        cond = {}
    n = len(seeds)
    total = np.zeros(n)
    for k in keys:
        v = np.asarray(cond[k] if k in cond else tpl[k], dtype=np.float64)
        total = total + (np.broadcast_to(v, (n, v.shape[-1])).sum(axis=1) / v.shape[-1]) * 100.0
    return {name: total}


def action_test_fields(action: str, seeds: Sequence[int], lubrication_effectiveness: float, randomize: bool = True) -> Dict[object, np.ndarray]:
    """Columns that turn freshly constructed plants (default configuration) into the plants
    ``MaintenanceScenarioRunner`` would build for ``compose_action_test_scenario(action, randomize=True,
    randomization_seed=seed)``, one per seed.

    Feedwater actions: restated (catalog + randomiser + constructor mapping, above).  Every other action of the
    composer's map (turbine, condenser, steam generator, generic; 100 of them): the composed template plus the few
    state members the action's catalog entry reaches (``action_state_deltas.json``: for 77 of them none at all -- their
    catalog parameters are not in the template's sections or are never read by a constructor).  For turbine, condenser
    and generic actions the composer's randomisation never reaches plant state, so that is exact for every seed; the
    ten steam-generator actions whose randomisation does are available un-randomised only."""
    if action not in _DELTAS["actions"]:
        raise NotImplementedError("%r is not an action-test scenario the reference can build%s" % (
            action, " (its own composition raises)" if action in _DELTAS["failed_in_reference"] else ""))
    info = _DELTAS["actions"][action]
    if info["subsystem"] != "feedwater":
        sg_rand = randomize and info["randomisation_reaches_state"]
        n = len(seeds)
        f = feedwater_fields(composed_feedwater_ic({}), n, lubrication_effectiveness)
        for k in range(NUM_SG):
            f[("sg.steam_flow_rate", k)] = np.full(1, ACTION_TEST_TEMPLATE["steam_generator"]["sg_steam_flows"][k])
        f["turb.rotor_temperature"] = np.full(1, ACTION_TEST_TEMPLATE["turbine"]["rotor_temperature"])
        for k in range(4):
            f[("turb.bearing_metal_temp", 0, k)] = np.full(1, ACTION_TEST_TEMPLATE["turbine"]["bearing_temperatures"][k])
        for label, value in info["delta"].items():
            f[_label_key(label)] = np.full(1, value)
        if sg_rand:
            f.update(_sg_randomized_fields(action, seeds))
        return f
    n = len(seeds)
    if randomize:
        tpl = _CATALOG["template_ic"]["feedwater"]
        ic = composed_feedwater_ic_columns(randomized_conditions_columns(action, seeds, keys=[k for k in FEEDWATER_IC_DEFAULTS if k in tpl]))
    else:
        ic = composed_feedwater_ic(catalog_conditions(action))
    f = feedwater_fields(ic, n, lubrication_effectiveness)
    for k in range(NUM_SG):
        f[("sg.steam_flow_rate", k)] = np.full(1, ACTION_TEST_TEMPLATE["steam_generator"]["sg_steam_flows"][k])
    f["turb.rotor_temperature"] = np.full(1, ACTION_TEST_TEMPLATE["turbine"]["rotor_temperature"])
    for k in range(4):
        f[("turb.bearing_metal_temp", 0, k)] = np.full(1, ACTION_TEST_TEMPLATE["turbine"]["bearing_temperatures"][k])
    return f
