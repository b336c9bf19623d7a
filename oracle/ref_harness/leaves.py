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


class H:
    """Lookups the maint.* schema columns use (include/npb_fields.h NPB_MAINT_FIELDS): the
    reference keeps this state in dicts keyed by component id, absent until first use."""

    @staticmethod
    def open_wo(root, k, what):
        ms = root.maintenance_system
        for w in ms.work_order_manager.work_orders.values():
            if w.component_id == "FWP-%d" % (k + 1) and w.status.name == "SCHEDULED":
                if what == "order":
                    return float(int(w.work_order_id.split("-")[1]))
                return float(getattr(w, what))
        return 0.0

    @staticmethod
    def last_violation(root, k):
        return root.state_manager.threshold_last_violation_times.get("FWP-%d" % (k + 1), {}).get("oil_level", -1.0)

    @staticmethod
    def last_trigger(root, k):
        return root.maintenance_system.recent_work_order_triggers.get("FWP-%d:oil_top_off" % (k + 1), -1.0)


def resolve(root, path):
    if path.startswith("="):
        return eval(path[1:], {"root": root, "H": H})
    return eval("root." + path if not path.startswith("[") else "root" + path, {"root": root})
