"""Initial conditions of the data-generation scenarios, as columns (BASELINE config 4).

The reference builds one plant at a time: ``ComprehensiveComposer.compose_action_test_scenario``
(data_gen/config_engine/composers/comprehensive_composer.py:75-295) deep-copies a YAML template, overwrites
the target subsystem's ``initial_conditions`` with a catalog entry, optionally randomised per seed
(initial_conditions/randomization_utils.py:799-1017), and the constructors then translate those
dictionaries into object state (feedwater/physics.py:185-437, feedwater/pump_system.py:1177-1233).
For 10^5 - 10^6 plants that dictionary shuffling dominates set-up, so this module produces the same
state directly as struct-of-arrays columns for ``BatchedPlantEnv.set_fields``:

    fields = action_test_fields("oil_top_off", seeds)        # {column: array[n]}
    env = BatchedPlantEnv(n, dt=5.0, noise_enabled=True, maintenance=True); env.set_fields(fields)

Only the ``oil_top_off`` action is catalogued here (the scenario BASELINE config 4 names); the mapping
functions take general initial-condition dictionaries, so further catalog entries are data.
Every column this module produces is checked against the reference's own constructor on several seeds
(tests/golden/ic_oil_top_off.npz, tests/test_scenarios.py).
"""
from __future__ import annotations

import random
from typing import Dict, Sequence

import numpy as np

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
    level, and "preserve_pattern" scales all four levels by the same factor.  The stdlib generator is used
    here too, so the draws are the reference's by construction (~10 us per seed)."""
    base = OIL_TOP_OFF_CONDITIONS["pump_oil_levels"]
    total = sum(p for p, _lo, _hi in OIL_TOP_OFF_SCENARIOS)
    out = np.empty((len(seeds), NUM_PUMPS))
    for row, seed in enumerate(seeds):
        r = random.Random(int(seed))
        rand_val = r.random()
        cumulative, pick = 0.0, OIL_TOP_OFF_SCENARIOS[-1]
        for sc in OIL_TOP_OFF_SCENARIOS:
            cumulative += sc[0] / total
            if rand_val <= cumulative:
                pick = sc
                break
        value = r.uniform(pick[1], pick[2])
        scale_factor = value / base[0]
        out[row] = [v * scale_factor for v in base]
    return out


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
    per_pump = lambda k: np.broadcast_to(np.asarray(g[k], dtype=np.float64), (n, NUM_PUMPS))
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
    sg_flows = np.broadcast_to(np.asarray(g["sg_steam_flows"], dtype=np.float64), (n, NUM_SG))
    total_steam_flow = 0.0 + sg_flows[:, 0] + sg_flows[:, 1] + sg_flows[:, 2]
    flow_per_pump = ((total_steam_flow * 1.02) / MIN_PUMPS_REQUIRED) / degradation_factor
    for k in range(NUM_PUMPS):
        f[("pump.oil_level", k)] = per_pump("pump_oil_levels")[:, k]
        f[("pump.oil_contamination", k)] = np.broadcast_to(np.float64(g["pump_oil_contamination"]), (n,))
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
        if k < MIN_PUMPS_REQUIRED:
            f[("pump.flow_demand", k)] = np.clip(flow_per_pump, 0.0, PUMP_RATED_FLOW * 1.2)    # set_flow_demand :422-447
            required_speed = np.sqrt(flow_per_pump / (PUMP_RATED_FLOW * flow_factor[:, k])) * 100.0
            speed = np.minimum(100.0, np.maximum(30.0, required_speed))                         # _calculate_safe_speed :1347-1368
            f[("pump.speed_percent", k)] = speed
            f[("pump.speed_setpoint", k)] = speed
    f["fw.total_flow_rate"] = total_steam_flow                                                    # physics.py:196-197
    icd = per_pump("impeller_cavitation_damage")
    f["fw.cav_accumulated_damage"] = ((0.0 + icd[:, 0] + icd[:, 1] + icd[:, 2] + icd[:, 3]) / NUM_PUMPS) * 10.0
    return {k: np.ascontiguousarray(v, dtype=np.float64) for k, v in f.items()}


def action_test_fields(action: str, seeds: Sequence[int], lubrication_effectiveness: float, randomize: bool = True) -> Dict[object, np.ndarray]:
    """Columns that turn freshly constructed plants (default configuration) into the plants
    ``MaintenanceScenarioRunner`` would build for ``compose_action_test_scenario(action, randomize=True,
    randomization_seed=seed)``, one per seed."""
    if action != "oil_top_off":
        raise NotImplementedError("only the oil_top_off catalog entry is restated (SURVEY.md 8f-2)")
    n = len(seeds)
    ic = dict(ACTION_TEST_TEMPLATE["feedwater"]); ic.update(OIL_TOP_OFF_CONDITIONS)
    if randomize:
        ic["pump_oil_levels"] = randomized_oil_top_off_levels(seeds)
    f = feedwater_fields(ic, n, lubrication_effectiveness)
    for k in range(NUM_SG):
        f[("sg.steam_flow_rate", k)] = np.full(n, ACTION_TEST_TEMPLATE["steam_generator"]["sg_steam_flows"][k])
    f["turb.rotor_temperature"] = np.full(n, ACTION_TEST_TEMPLATE["turbine"]["rotor_temperature"])
    for k in range(4):
        f[("turb.bearing_metal_temp", 0, k)] = np.full(n, ACTION_TEST_TEMPLATE["turbine"]["bearing_temperatures"][k])
    return f
