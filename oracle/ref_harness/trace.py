"""Run a scenario on the *reference* simulator and record a trace: per-step inputs,
obs/reward/done/info and the value of every schema column's reference attribute.
Harness-only (needs /root/reference; see refsim.py).

Scenario dict keys:
  name, steps, dt, heat_source ("constant"|"reactor"), noise (bool), noise_seed,
  noise_std_percent, secondary (IC override dict for SecondarySystemConfig.from_dict),
  enable_secondary: False -> NuclearPlantSimulator(enable_secondary=False): 12 observations (recorded in obs[:, :12], the
          rest 0), no secondary keys in info (recorded as NaN),
  equilibrium: (power, rods) -> start ReactorState from create_equilibrium_state,
  actions: {step: (action_id, magnitude)} or callable(step)->(action, magnitude),
  setpoints: callable(step)->percent or None,
  cooling: callable(step)->degC or None,
  pokes: {step: [(python_path, value), ...]} applied to the sim before that step; value may be callable(sim) -> number; a path written
         "~path" is applied to the reference only (a cached copy of a poked member that no schema column holds) and not recorded for replay.
  resets: {step: start_at_steady_state} -> sim.reset(start_at_steady_state) is called before that step (after the
          step's pokes); the observation it returns and the state it leaves are recorded (reset_obs / reset_state).
  runner: {"action": name, "duration_hours": h, "feedwater_ic": {...}} -> build the simulator the
          way data_gen's MaintenanceScenarioRunner does (state management + AutoMaintenanceSystem
          on, ComprehensiveComposer action-test config, maintenance_scenario_runner.py:210-244)
          and drive it with the runner's own ramped power profile (:383-411, :651-671).
"""
import enum

import numpy as np

from . import refsim
from .leaves import resolve


def _val(sim, path):
    if not path:
        return np.nan
    try:
        v = resolve(sim, path)
    except (AttributeError, KeyError, IndexError):
        return np.nan
    if isinstance(v, enum.Enum):
        # numeric enums by value, string-valued enums by declaration index
        v = v.value if isinstance(v.value, (int, float)) else list(type(v)).index(v)
    try:
        return float(v)
    except (TypeError, ValueError):
        return np.nan


def _poke(sim, path, v):
    """assign a reference attribute: a plain attribute chain, or one of the schema's "=expression" paths when the expression is
    itself assignable (list(root....values())[k].member)"""
    if path.startswith("="):
        from .leaves import H
        exec(path[1:] + " = v", {"root": sim, "H": H, "v": v})
    else:
        exec("sim.%s = v" % path, {"sim": sim, "v": v})


def poke_value(v):
    """a poke's value: a number, or "=<expr>" evaluated with the reference's enums in scope (e.g. "=PumpStatus.STOPPING")"""
    if isinstance(v, str) and v.startswith("="):
        from systems.primary.coolant.pump_models import PumpStatus
        return eval(v[1:], {"PumpStatus": PumpStatus, "nan": float("nan"), "inf": float("inf"), "True": True, "False": False})
    return v


def poke_number(v):
    """the same value as the number a schema column holds (enums by declaration index, as _val reads them)"""
    v = poke_value(v)
    if isinstance(v, enum.Enum):
        return float(v.value) if isinstance(v.value, (int, float)) else float(list(type(v)).index(v))
    return float(v)


def run_reference(sc, columns):
    """columns: SCHEMA.columns(). Returns dict of arrays."""
    refsim.setup()
    runner = None
    if sc.get("runner") is not None:
        runner, sim = refsim.make_runner_sim(**sc["runner"])
        profile = runner._generate_power_profile(sc["steps"])
    else:
      sim = refsim.make_sim(dt=sc.get("dt", 1.0), heat_source=sc.get("heat_script") or sc.get("heat_source", "constant"),
                          noise=sc.get("noise", False), noise_std_percent=sc.get("noise_std_percent", 0.1),
                          noise_seed=sc.get("noise_seed", 42), secondary=sc.get("secondary"),
                          enable_secondary=sc.get("enable_secondary", True), state_management=sc.get("state_management", False))
    from systems.primary import ControlAction
    if sc.get("feedwater_thresholds_only") and getattr(sim, "state_manager", None) is not None:
        # a maintenance configuration that names the feedwater pumps only: the automatic maintenance of steam generators, turbine
        # and condenser (TSP cleaning, bearing work ...) is outside the path this repository restates, and a run whose log is to
        # be followed column by column must not have the reference execute it
        th = sim.state_manager.maintenance_thresholds
        for cid in [c for c in th if not c.startswith("FWP-")]:
            del th[cid]
    if sc.get("thresholds_override"):
        # edit the live maintenance thresholds of every feedwater pump (what another maintenance configuration would load)
        for cid, th in sim.state_manager.maintenance_thresholds.items():
            if cid.startswith("FWP-"):
                for name, changes in sc["thresholds_override"]:
                    th[name].update(changes)
        sc["_maint_thresholds"] = [[n, {k: c.get(k) for k in ("threshold", "comparison", "action", "cooldown_hours", "priority", "component_id")}]
                                   for n, c in sim.state_manager.maintenance_thresholds["FWP-1"].items()]
    if sc.get("state_management") and getattr(sim, "maintenance_system", None) is not None:
        # a simulator constructed with state management but without a maintenance configuration: the thresholds the state
        # manager's factory default gives a feedwater pump and the execution delays of the non-aggressive mode, as the run used them
        ms = sim.maintenance_system
        sc["_maint_thresholds"] = [[n, {k: c.get(k) for k in ("threshold", "comparison", "action", "cooldown_hours", "priority", "component_id")}]
                                   for n, c in sim.state_manager.maintenance_thresholds["FWP-1"].items()]
        sc["_maint_params"] = dict(maint_emergency_delay_hours=float(ms.emergency_delay_hours), maint_start_delay_hours=float(ms.high_priority_delay_hours),
                                   maint_medium_delay_hours=float(ms.medium_priority_delay_hours), maint_low_delay_hours=float(ms.low_priority_delay_hours))
    if sc.get("equilibrium") is not None:
        from systems.primary.reactor.reactivity_model import create_equilibrium_state
        p, rods = sc["equilibrium"]
        with refsim.quiet():
            st = create_equilibrium_state(power_level=p, control_rod_position=rods, auto_balance=True)
        sim.primary_physics.state = st
        sim.state = st
    for path, v in sc.get("init_pokes", []):
        _poke(sim, path, poke_value(v))
    T = sc["steps"]
    paths = [c[3] for c in columns]
    state = np.full((T + 1, len(paths)), np.nan)
    state[0] = [_val(sim, p) for p in paths]
    obs = np.zeros((T, 22)); rew = np.zeros(T); done = np.zeros(T, dtype=np.uint8)
    info = np.full((T, 10), np.nan)
    act = np.full(T, 8, dtype=np.int32); mag = np.ones(T); sp = np.full(T, np.nan)
    cw = np.full(T, np.nan); z = np.zeros(T)
    # pre-draw the heat-source noise exactly as numpy's legacy RandomState would
    if sc.get("noise", False):
        z[:] = np.random.RandomState(sc.get("noise_seed", 42)).standard_normal(T)
    actions = sc.get("actions")
    sec_keys, sec_rows = None, []
    rc_keys, rc_rows = None, []     # info["reactivity_components"] (sim.py:205), in the dict's own order
    resets = sc.get("resets", {})
    reset_steps, reset_modes, reset_obs, reset_state = [], [], [], []
    for t in range(T):
        lst = sc.get("pokes", {}).get(t, [])
        for j, (path, v) in enumerate(lst):
            if callable(v):          # a value computed from the live simulator (e.g. the boron that puts the reactivity on a threshold):
                v = float(v(sim))    # resolved here and written back, so the fixture records the number that was poked
                lst[j] = (path, v)
            _poke(sim, path.lstrip("~"), poke_value(v))     # "~path": a derived copy the reference caches (not a schema member), kept consistent with the poke beside it
        if t in resets:
            with refsim.quiet():
                ob = sim.reset(start_at_steady_state=bool(resets[t]))
            reset_steps.append(t); reset_modes.append(int(bool(resets[t]))); reset_obs.append(np.concatenate([np.asarray(ob, dtype=np.float64), np.zeros(22 - len(ob))]))
            reset_state.append([_val(sim, p) for p in paths])
        if actions is not None:
            a = actions(t) if callable(actions) else actions.get(t)
            if a is not None:
                act[t], mag[t] = a
        if runner is not None:
            with refsim.quiet():
                runner._set_target_power(profile[t])
            sp[t] = sim.primary_physics.heat_source.power_setpoint_percent
        if sc.get("setpoints") is not None:
            v = sc["setpoints"](t)
            if v is not None:
                sp[t] = v
                sim.primary_physics.heat_source.set_power_setpoint(v)
        kw = {}
        if sc.get("cooling") is not None:
            v = sc["cooling"](t)
            if v is not None:
                cw[t] = v; kw["cooling_water_temp"] = v
        with refsim.quiet():
            r = sim.step(ControlAction(int(act[t])), magnitude=float(mag[t]), **kw)
        if sc.get("heat_script") is not None:    # the plugin's result for this step, in the columns the C ABI carries it in (NPB_HEAT_EXTERNAL)
            z[t], sp[t] = sim.primary_physics.heat_source.results[-1]
        obs[t, :len(r["observation"])] = r["observation"]; rew[t] = r["reward"]; done[t] = bool(r["done"])
        i = r["info"]
        info[t] = [i["thermal_power"], i["reactivity"], i.get("electrical_power", np.nan),
                   i.get("thermal_efficiency", np.nan), i.get("steam_flow", np.nan),
                   i.get("steam_pressure", np.nan), i.get("condenser_pressure", np.nan),
                   i.get("condenser_heat_rejection", np.nan), i["time"],
                   i["secondary_system"]["feedwater_total_flow"] if "secondary_system" in i else np.nan]
        rc = i.get("reactivity_components") or {}
        if rc_keys is None:
            rc_keys = list(rc.keys())
        rc_rows.append([float(rc[k]) for k in rc_keys])
        ss = i.get("secondary_system", {})
        if sec_keys is None:   # every scalar key of SecondaryReactorPhysics.update_system's result (secondary/__init__.py:922-1010)
            sec_keys = [k for k, v in ss.items() if isinstance(v, (bool, int, float, np.floating, np.integer, np.bool_))]
        sec_rows.append([float(ss[k]) for k in sec_keys])
        state[t + 1] = [_val(sim, p) for p in paths]
    return dict(state=state, obs=obs, reward=rew, done=done, info=info,
                action=act, magnitude=mag, setpoint=sp, cooling=cw, noise_z=z,
                rc_keys=np.array(rc_keys or []), rc=np.array(rc_rows, dtype=np.float64).reshape(T, len(rc_keys or [])),
                sec_keys=np.array(sec_keys or []), sec=np.array(sec_rows, dtype=np.float64).reshape(T, len(sec_keys or [])),
                reset_steps=np.array(reset_steps, dtype=np.int64), reset_modes=np.array(reset_modes, dtype=np.int64),
                reset_obs=np.array(reset_obs, dtype=np.float64).reshape(len(reset_steps), 22),
                reset_state=np.array(reset_state, dtype=np.float64).reshape(len(reset_steps), len(paths))), sim
