"""ctypes binding of libnpb.so (the HIP stepper's C ABI, include/npb.h).

There is no CPU fallback: if the shared library is missing or does not load, importing
this module raises.  Build it with ``make -C nuclear_sim_amd/csrc`` (or
``__graft_entry__.build()``).
"""
from __future__ import annotations

import ctypes
import os

from .schema import PARAMS, SCHEMA

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NPB_LIB", os.path.join(_HERE, "libnpb.so"))

NPB_KIND_F64, NPB_KIND_I32 = 0, 1
HEAT_CONSTANT, HEAT_REACTOR, HEAT_EXTERNAL = 0, 1, 2
STORAGE_F64, STORAGE_F32 = 0, 1
MODE_FULL, MODE_PRIMARY_SG, MODE_PRIMARY = 0, 1, 2
OBS_DIM, INFO_DIM = 22, 17   # include/npb.h NPB_OBS_DIM / NPB_INFO_DIM; checked against the library in load()
INFO_NRHO = 10   # include/npb.h NPB_INFO_NRHO
# include/npb.h NPB_DIAG_*: step-internal diagnostics, fourteen turbine stages each (TurbineStage.get_state_dict, stage_system.py:379-393)
DIAG_STAGE_VALUES = ("inlet_pressure", "inlet_temperature", "outlet_pressure", "outlet_temperature", "power_output", "loading_factor")
DIAG_SG_VALUES = ("primary_inlet_temp", "primary_outlet_temp", "overall_htc", "feedwater_flow_rate")   # steam_generator.py:943-985
DIAG_PUMP_VALUES = ("system_health_factor", "maintenance_action_occurred", "oil_top_off_occurred")   # per pump, FWP-1..4 (NPB_DIAG_PUMP_*)
DIAG_FW_VALUES = ("feedwater_avg_sg_level", "feedwater_avg_sg_pressure", "feedwater_total_steam_flow", "feedwater_avg_steam_quality")   # NPB_DIAG_FW_*
DIAG_ROTOR_VALUES = ("friction_torque", "net_torque", "rotor_acceleration")   # NPB_DIAG_ROTOR_*
# NPB_DIAG_COND_*, NPB_DIAG_SG_SCALE_FORMATION_RATE (x3), NPB_DIAG_FW_PERFORMANCE_FACTOR: (log column, row offset behind the rotor's)
DIAG_TAIL_COLUMNS = (("secondary.condenser_SECONDARY-COMP-001-COND.condenser_overall_htc", 0), ("secondary.condenser_SECONDARY-COMP-001-COND.tube_leak_rate", 1),
                     ("secondary.condenser.SJE-001_steam_flow", 2), ("secondary.condenser.SJE-001_steam_consumption", 3),
                     ("secondary.condenser.vacuum_system_steam_consumption", 4),
                     ("secondary.steam_generator_SG-0.tube_scale_formation_rate_mm_per_year", 5), ("secondary.steam_generator_SG-1.tube_scale_formation_rate_mm_per_year", 6),
                     ("secondary.steam_generator_SG-2.tube_scale_formation_rate_mm_per_year", 7),
                     ("secondary.feedwater_SECONDARY-COMP-001-FW.feedwater_performance_factor", 8),
                     # accumulators (NPB_DIAG_ROTOR_CLEARANCE_INCREASE x4, NPB_DIAG_ROTOR_OVERSPEED_EVENTS): since the diagnostics were switched on
                     ("secondary.turbine_SECONDARY-COMP-001-TURB.TB-001_clearance_increase", 9), ("secondary.turbine_SECONDARY-COMP-001-TURB.TB-002_clearance_increase", 10),
                     ("secondary.turbine_SECONDARY-COMP-001-TURB.TB-003_clearance_increase", 11), ("secondary.turbine_SECONDARY-COMP-001-TURB.TB-004_clearance_increase", 12),
                     ("secondary.turbine_SECONDARY-COMP-001-TURB.overspeed_events", 13),
                     # NPB_DIAG_BEARING_OIL_TEMP x4
                     ("secondary.turbine_SECONDARY-COMP-001-TURB.TB-001_oil_temp", 14), ("secondary.turbine_SECONDARY-COMP-001-TURB.TB-002_oil_temp", 15),
                     ("secondary.turbine_SECONDARY-COMP-001-TURB.TB-003_oil_temp", 16), ("secondary.turbine_SECONDARY-COMP-001-TURB.TB-004_oil_temp", 17),
                     # offset 18 = NPB_DIAG_STAGE_SYSTEM_EFFICIENCY: carried from step to step, not a log column of the reference's;
                     # NPB_DIAG_TURBINE_PERFORMANCE_FACTOR, NPB_DIAG_FW_ACTIVE_ALARMS
                     ("secondary.turbine_SECONDARY-COMP-001-TURB.enhanced_turbine_performance", 19),
                     ("secondary.feedwater_SECONDARY-COMP-001-FW.protection_active_alarms_count", 20))
# round 4 (include/npb.h NPB_DIAG_PUMP_MAINTENANCE_ACTION ...): rows by number
DIAG_PUMP_MAINTENANCE_ACTION = 136          # x4: catalog index + 1 of the action carried out on the pump in this step, 0 = none
DIAG_FW_ACTIVE_TRIPS, DIAG_FW_VALID_TRIP_COUNT, DIAG_FW_EMERGENCY_FEEDWATER, DIAG_FW_STEAM_DUMP = 140, 141, 142, 143
DIAG_STAGE_EXTRACTION_FLOW = 144            # x14
DIAG_COND_SJE_CAPACITY, DIAG_COND_SJE_STEAM_FLOW, DIAG_COND_SJE_STEAM_CONSUMPTION = 158, 160, 162    # x2 each
DIAG_COND_SJE_COMPRESSION_RATIO, DIAG_COND_SJE_OPERATING_HOURS, DIAG_COND_AIR_REMOVAL = 164, 166, 168
DIAG_STAGE_SYSTEM_TOTAL_POWER = 169
DIAG_STAGE_POWER_OUTPUT = 56                # x14 (NPB_DIAG_STAGE_POWER_OUTPUT)
# rows the step CARRIES in the caller's buffer from one step to the next (accumulators, latches, values kept while equipment
# rests): row -> value of a freshly constructed plant.  BatchedPlantEnv.enable_diagnostics / reset put them there.
DIAG_CARRIED_ROWS = {**{124 + q: 0.0 for q in range(5)}, 133: 0.0, DIAG_FW_VALID_TRIP_COUNT: 0.0, DIAG_FW_EMERGENCY_FEEDWATER: 0.0,
                     DIAG_FW_STEAM_DUMP: 0.0, DIAG_COND_SJE_COMPRESSION_RATIO: 1.0, DIAG_COND_SJE_COMPRESSION_RATIO + 1: 1.0,
                     DIAG_COND_SJE_OPERATING_HOURS: 0.0, DIAG_COND_SJE_OPERATING_HOURS + 1: 0.0}
DIAG_DIM = 170
assert 14 * len(DIAG_STAGE_VALUES) + 3 * len(DIAG_SG_VALUES) + 4 * len(DIAG_PUMP_VALUES) + len(DIAG_FW_VALUES) + len(DIAG_ROTOR_VALUES) + len(DIAG_TAIL_COLUMNS) + 1 == DIAG_PUMP_MAINTENANCE_ACTION
REACTIVITY_COMPONENTS = ("control_rods", "boron", "doppler", "moderator_temp", "moderator_void", "pressure", "xenon", "samarium",
                         "fuel_depletion", "burnable_poisons")   # reactivity_model.py:87-121, NPB_RHO_*


class NpbError(RuntimeError):
    pass


def _make_params_struct():
    fields = [(name, ctypes.c_double) for name, _d, _p in PARAMS]
    fields += [("dt", ctypes.c_double), ("heat_source", ctypes.c_int), ("hs_noise_enabled", ctypes.c_int),
               ("mode", ctypes.c_int), ("maint_enabled", ctypes.c_int), ("info_reactivity_components", ctypes.c_int), ("kinetics_rk4_substeps", ctypes.c_int)]
    return type("NpbParams", (ctypes.Structure,), {"_fields_": fields})


NpbParams = _make_params_struct()

MAINT_NPARAM, MAINT_NACT = 16, 18


class NpbMaintTable(ctypes.Structure):
    """npb_maint_table_t (include/npb_maint.h): one row per catalogued threshold parameter"""
    _fields_ = [("threshold", ctypes.c_double * MAINT_NPARAM), ("cooldown_hours", ctypes.c_double * MAINT_NPARAM),
                ("rank", ctypes.c_int * MAINT_NPARAM), ("comparison", ctypes.c_int * MAINT_NPARAM),
                ("action", ctypes.c_int * MAINT_NPARAM), ("priority", ctypes.c_int * MAINT_NPARAM),
                ("bearing", ctypes.c_int * MAINT_NPARAM)]


def __getattr__(name):
    """MAINT_PARAMS / MAINT_ACTIONS: the catalogs of include/npb_maint.h, read from the library itself (npb_maint_param_name /
    npb_maint_action_name) the first time they are asked for -- the package needs no header beside it."""
    if name in ("MAINT_PARAMS", "MAINT_ACTIONS"):
        L = load()
        params = [L.npb_maint_param_name(k).decode() for k in range(L.npb_maint_num_params())]
        actions = [L.npb_maint_action_name(a).decode() for a in range(L.npb_maint_num_actions())]
        if len(params) != MAINT_NPARAM or len(actions) != MAINT_NACT:
            raise NpbError("libnpb.so's maintenance catalogs (%d parameters, %d actions) are not the %d / %d this binding's "
                           "NpbMaintTable is laid out for: rebuild" % (len(params), len(actions), MAINT_NPARAM, MAINT_NACT))
        globals()["MAINT_PARAMS"], globals()["MAINT_ACTIONS"] = params, actions
        return globals()[name]
    raise AttributeError(name)


MAINT_COMPARISONS = ("greater_than", "less_than", "greater_equal", "less_equal", "equals", "not_equals")
MAINT_PRIORITIES = {"LOW": 1, "MEDIUM": 2, "HIGH": 3, "CRITICAL": 4, "EMERGENCY": 5}
MAINT_BEARINGS = {None: 0, "all": 0, "motor_bearings": 1, "pump_bearings": 2, "thrust_bearing": 3}


def maint_table_from_thresholds(thresholds: dict) -> "NpbMaintTable":
    """The reference's thresholds dict of a feedwater pump (maintenance_system.component_configs.feedwater.thresholds of
    the configuration, = StateManager.maintenance_thresholds['FWP-1'], in ITS order) -> table.  Names that do not
    resolve in a pump's state log are dropped, as the reference's scan drops them (state_manager.py:1371-1411)."""
    t = NpbMaintTable()
    MAINT_PARAMS, MAINT_ACTIONS = __getattr__("MAINT_PARAMS"), __getattr__("MAINT_ACTIONS")
    for k in range(MAINT_NPARAM):
        t.rank[k] = -1
    for rank, (name, cfg) in enumerate(thresholds.items()):
        if name not in MAINT_PARAMS or cfg.get("threshold") is None:
            continue
        k = MAINT_PARAMS.index(name)
        action = cfg.get("action")
        if action not in MAINT_ACTIONS:
            raise NpbError("maintenance action %r of threshold %r is not in the action catalog (include/npb_maint.h)" % (action, name))
        t.rank[k] = rank
        t.threshold[k] = float(cfg["threshold"])
        t.cooldown_hours[k] = float(cfg.get("cooldown_hours", 24.0))
        t.comparison[k] = MAINT_COMPARISONS.index(cfg.get("comparison", "greater_than"))
        t.action[k] = MAINT_ACTIONS.index(action)
        t.priority[k] = MAINT_PRIORITIES.get(str(cfg.get("priority", "MEDIUM")).upper(), 2)
        t.bearing[k] = MAINT_BEARINGS.get(cfg.get("component_id"), 0)
    return t

_lib = None


def load():
    """Load libnpb.so and declare its entry points; raises NpbError when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NpbError("%s not found: build the HIP extension first (make -C nuclear_sim_amd/csrc); "
                       "there is no CPU fallback" % LIB_PATH)
    # PyTorch-ROCm bundles its own HIP runtime; import it first so that libnpb.so binds to the one
    # runtime already in the process (two HIP runtimes in one process do not see the device).
    import torch  # noqa: F401
    L = ctypes.CDLL(LIB_PATH)
    vp, ci = ctypes.c_void_p, ctypes.c_int
    L.npb_version.restype = ci
    L.npb_num_f64.restype = ci
    L.npb_num_i32.restype = ci
    L.npb_state_bytes.restype = ctypes.c_size_t
    L.npb_step_bytes_per_plant.restype = ctypes.c_size_t
    L.npb_default_params.argtypes = [ctypes.POINTER(NpbParams)]
    L.npb_create.argtypes = [ctypes.POINTER(NpbParams), ci, ci, ctypes.POINTER(vp)]
    L.npb_create_storage.argtypes = [ctypes.POINTER(NpbParams), ci, ci, ci, ctypes.POINTER(vp)]
    L.npb_storage.argtypes = [vp]
    L.npb_handle_step_bytes_per_plant.argtypes = [vp]
    L.npb_handle_step_bytes_per_plant.restype = ctypes.c_size_t
    L.npb_destroy.argtypes = [vp]
    L.npb_last_error.argtypes = [vp]
    L.npb_last_error.restype = ctypes.c_char_p
    L.npb_num_plants.argtypes = [vp]
    L.npb_set_params.argtypes = [vp, ctypes.POINTER(NpbParams)]
    L.npb_reset.argtypes = [vp, vp, vp]
    L.npb_set_step_kernel.argtypes = [vp, ci]
    if hasattr(L, "npb_debug_last_step_kernel"):    # ABI 140
        L.npb_debug_last_step_kernel.argtypes = [vp]
        L.npb_step_kernel_name.argtypes = [ci]
        L.npb_step_kernel_name.restype = ctypes.c_char_p
        for f in ("npb_maint_param_name", "npb_maint_action_name"):
            getattr(L, f).argtypes = [ci]
            getattr(L, f).restype = ctypes.c_char_p
        # the widths this binding allocates its output blocks with must be the library's: a library that writes more info
        # columns than the caller allocated overruns the buffer (what crashed a round-2 test run on the host side)
        got = (L.npb_obs_dim(), L.npb_info_dim(), L.npb_info_nrho(), L.npb_diag_dim())
        if got != (OBS_DIM, INFO_DIM, INFO_NRHO, DIAG_DIM):
            raise NpbError("libnpb.so writes obs / info / reactivity / diagnostics blocks of width %r, this binding allocates %r: "
                           "rebuild the library or update nuclear_sim_amd/_lib.py" % (got, (OBS_DIM, INFO_DIM, INFO_NRHO, DIAG_DIM)))
    if hasattr(L, "npb_set_diagnostics"):     # (absent from builds older than ABI 133: tools/ab_kernel.py loads those)
        L.npb_set_diagnostics.argtypes = [vp, vp, ctypes.c_size_t]
    L.npb_set_maintenance_table.argtypes = [vp, ctypes.POINTER(NpbMaintTable)]
    if hasattr(L, "npb_set_maintenance_count_buffer"):
        L.npb_set_maintenance_count_buffer.argtypes = [vp, vp]
    L.npb_default_maintenance_table.argtypes = [ctypes.POINTER(NpbMaintTable)]
    L.npb_reset_reference.argtypes = [vp, vp, ci, vp]
    L.npb_get_field.argtypes = [vp, ci, ci, vp, ci, vp]
    L.npb_set_field.argtypes = [vp, ci, ci, vp, ci, vp]
    L.npb_state_arena.argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ci)]
    if hasattr(L, "npb_state_arena_layout"):     # ABI 142
        L.npb_state_arena_layout.argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ci), ctypes.POINTER(ci)]
    if hasattr(L, "npb_state_arena_segment"):    # ABI 141
        L.npb_state_arena_segment.argtypes = [vp]
        L.npb_state_arena_segment.restype = ctypes.c_size_t
    L.npb_gather_fields.argtypes = [vp, ci, vp, vp, vp, vp]
    L.npb_locate_field.argtypes = [vp, ci, ci, ctypes.POINTER(ci), ctypes.POINTER(ci), ctypes.POINTER(ci)]
    L.npb_step.argtypes = [vp] + [vp] * 11
    L.npb_observe.argtypes = [vp, vp, vp]
    L.npb_debug_touch.argtypes = [vp, vp]
    if L.npb_num_f64() != SCHEMA.total_f64 or L.npb_num_i32() != SCHEMA.total_i32:
        raise NpbError("libnpb.so was built against a different include/npb_fields.h (%d/%d vs %d/%d): rebuild"
                       % (L.npb_num_f64(), L.npb_num_i32(), SCHEMA.total_f64, SCHEMA.total_i32))
    _lib = L
    return L


def default_params() -> "NpbParams":
    p = NpbParams()
    load().npb_default_params(ctypes.byref(p))
    return p


def check(rc, handle=None):
    if rc != 0:
        msg = load().npb_last_error(handle)
        raise NpbError("libnpb error %d: %s" % (rc, msg.decode() if msg else "?"))
