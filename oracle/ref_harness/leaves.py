"""Walk the reference simulator's object graph and list every numeric leaf
(float/int/bool/Enum/np scalar/np array element) with an attribute path that
`resolve()` can evaluate again.  Harness-only tooling (see refsim.py)."""
import enum
import numpy as np

SKIP_ATTRS = {"config", "heat_flow_tracker", "chemistry_flow_tracker", "state_manager",
              "maintenance_system", "state_df", "rng", "_state_provider",
              "component_registry"}


def walk(obj, path="", out=None, seen=None, depth=0, skip=SKIP_ATTRS):
    if out is None:
        out, seen = {}, set()
    if depth > 12:
        return out
    if isinstance(obj, bool):
        out[path] = float(obj); return out
    if isinstance(obj, (int, float, np.integer, np.floating)):
        out[path] = float(obj); return out
    if isinstance(obj, enum.Enum):
        v = obj.value
        out[path] = float(v) if isinstance(v, (int, float)) else float(list(type(obj)).index(obj))
        return out
    if obj is None or isinstance(obj, (str, bytes)):
        return out
    if isinstance(obj, np.ndarray):
        if obj.dtype.kind in "fiub" and obj.size <= 64:
            for i, v in enumerate(obj.ravel()):
                out["%s[%d]" % (path, i)] = float(v)
        return out
    if id(obj) in seen:
        return out
    seen.add(id(obj))
    if isinstance(obj, dict):
        for k, v in obj.items():
            if isinstance(k, (str, int)):
                walk(v, "%s[%r]" % (path, k), out, seen, depth + 1, skip)
        return out
    if isinstance(obj, (list, tuple)):
        if len(obj) <= 64:
            for i, v in enumerate(obj):
                walk(v, "%s[%d]" % (path, i), out, seen, depth + 1, skip)
        return out
    if isinstance(obj, (set, frozenset)):
        out[path + ".__len__"] = float(len(obj)); return out
    d = getattr(obj, "__dict__", None)
    if d is None:
        return out
    for k, v in d.items():
        if k in skip or callable(v) and not hasattr(v, "__dict__"):
            continue
        if callable(v) and type(v).__name__ in ("function", "method", "builtin_function_or_method"):
            continue
        walk(v, (path + "." + k) if path else k, out, seen, depth + 1, skip)
    return out


def _maint_catalog():
    """parameter / action names of include/npb_maint.h, in catalog order"""
    import os, re
    text = open(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "include", "npb_maint.h")).read()
    params = re.findall(r'^\s*X\(\w+,\s*"(\w+)"\)', text, flags=re.M)
    actions = re.findall(r'^\s*X\(\w+,\s*"(\w+)",\s*[01]\)', text, flags=re.M)
    return params, actions


class H:
    """Lookups the mpump.* / maint.* schema columns use (include/npb_fields.h): the reference keeps this state in dicts
    keyed by component id, absent until first use.  k indexes the parameter / action catalog of include/npb_maint.h."""
    PARAMS, ACTIONS = _maint_catalog()
    BEARINGS = {None: 0.0, "all": 0.0, "motor_bearings": 1.0, "pump_bearings": 2.0, "thrust_bearing": 3.0}

    @staticmethod
    def _open(root, pump, action):
        ms = root.maintenance_system
        for w in ms.work_order_manager.work_orders.values():
            if w.component_id == "FWP-%d" % (pump + 1) and w.status.name == "SCHEDULED" and \
                    any(a.action_type == action for a in w.maintenance_actions):
                return w
        return None

    @staticmethod
    def open_wo(root, pump, k, what):
        w = H._open(root, pump, H.ACTIONS[k])
        if w is None:
            return 0.0
        if what == "order":
            return float(int(w.work_order_id.split("-")[1]))
        return float(getattr(w, what))

    @staticmethod
    def open_wo_bearing(root, pump):
        w = H._open(root, pump, "bearing_replacement")
        if w is None:
            return 0.0
        return H.BEARINGS[(getattr(w, "metadata", None) or {}).get("extracted_component_id")]

    @staticmethod
    def last_violation(root, pump, k):
        return root.state_manager.threshold_last_violation_times.get("FWP-%d" % (pump + 1), {}).get(H.PARAMS[k], -1.0)

    @staticmethod
    def last_trigger(root, pump, k):
        return root.maintenance_system.recent_work_order_triggers.get("FWP-%d:%s" % (pump + 1, H.ACTIONS[k]), -1.0)

    # ---- trip reasons: the reference keeps strings; the schema keeps a code / a bit mask
    PUMP_TRIP_CODES = (("Low Flow", 1), ("NPSH Violation", 2), ("Low Suction Pressure", 3), ("High Discharge Pressure", 4),
                       ("Steam Generator High Level", 5), ("Severe Cavitation", 6), ("Cavitation Damage Limit", 7),
                       ("Critical NPSH Violation", 8), ("Lubrication: Very Low Oil Level", 10), ("Lubrication: Low Oil Level", 11),
                       ("Lubrication: Oil System Overfill", 12), ("Excessive Wear", 13), ("Lubrication: Excessive Seal Leakage", 14),
                       ("Lubrication: Combined Wear Limit", 15), ("Lubrication: Performance Degradation", 16))
    TURBINE_TRIP_BITS = {"Overspeed": 1, "High Vibration": 2, "High Bearing Temperature": 4, "Thrust Bearing Displacement": 8,
                         "Low Vacuum": 16, "High Thermal Stress": 32}

    @staticmethod
    def pump_trip_reason(root, pump):
        """FeedwaterPumpState.trip_reason (pump_models.py:261-267, pump_system.py:277-359) -> NPD_TRIP_* code (npd_feedwater.h)"""
        r = root.secondary_physics.feedwater_system.pump_system.pumps["FWP-%d" % (pump + 1)].state.trip_reason or ""
        if not r:
            return 0.0
        for text, code in H.PUMP_TRIP_CODES:
            if (r.startswith(text) if not text.startswith("Excessive Wear") else r.endswith("Excessive Wear")):
                return float(code)
        raise ValueError("unmapped pump trip reason %r" % r)

    @staticmethod
    def turbine_trip_mask(root):
        """TurbineProtectionSystem.trip_reasons (turbine/enhanced_physics.py:348-436: appended once, never removed until a
        reset) -> bit mask in the order of the six checks"""
        return float(sum(H.TURBINE_TRIP_BITS[r] for r in root.secondary_physics.turbine.protection_system.trip_reasons))

    @staticmethod
    def executed(root, k):
        n = 0
        for w in root.maintenance_system.work_order_manager.work_orders.values():
            if w.component_id.startswith("FWP-") and w.status.name not in ("SCHEDULED", "PLANNED") and \
                    any(a.action_type == H.ACTIONS[k] for a in w.maintenance_actions):
                n += 1
        # completed orders may also be moved to a history list
        for w in getattr(root.maintenance_system.work_order_manager, "completed_work_orders", []) or []:
            if w.component_id.startswith("FWP-") and any(a.action_type == H.ACTIONS[k] for a in w.maintenance_actions):
                n += 1
        return float(n)


def resolve(root, path):
    if path.startswith("="):
        return eval(path[1:], {"root": root, "H": H})
    return eval("root." + path if not path.startswith("[") else "root" + path, {"root": root})
