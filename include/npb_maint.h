/*
 * npb_maint.h -- automatic maintenance of the feedwater pumps (public ABI vocabulary).
 *
 * The reference's control plane for this is three layers (state manager -> orchestrator -> AutoMaintenanceSystem ->
 * component handlers); what a run observes of it is restated in nuclear_sim_amd/csrc/npd_maintenance.h.  This header
 * holds the vocabulary both sides of the C ABI share:
 *
 *  - the PARAMETER catalog: the threshold names of the maintenance configuration that resolve to a value in the
 *    state log of a feedwater pump (StateManager._find_parameter_in_row_data, simulator/state/state_manager.py:1371-1411,
 *    against FeedwaterPump.get_state_dict feedwater/pump_system.py:1062-1086 + FeedwaterPumpLubricationSystem.get_state_dict
 *    feedwater/pump_lubrication.py:1582-1644).  The configuration names 28 more thresholds for the same component
 *    (pump_head, suction_pressure, oil_water_content ...) that never resolve and therefore never fire; they have no
 *    entry here.
 *  - the ACTION catalog: every action those thresholds, the orchestrator's promotions (maintenance_orchestrator.py
 *    :469-524) and the lubrication system's dispatcher (pump_lubrication.py:625-674) can produce.
 *  - the threshold TABLE a handle carries (npb_set_maintenance_table): per catalogued parameter its rank in the
 *    configuration's dict order (the scan order; -1 = absent), threshold, comparison, action, priority, cooldown and,
 *    for bearing_replacement, which bearing.  npb_maint_table_default() is the table of the data-gen action-test
 *    configuration (data_gen/config_engine/templates/nuclear_plant_comprehensive_config.yaml:682-1010 after the
 *    composer, as the live StateManager.maintenance_thresholds['FWP-1'] holds it; tests/golden/maint_table.json is
 *    that dict dumped from the reference and tests/test_maintenance_cpu.py checks this function against it).
 */
#ifndef NPB_MAINT_H
#define NPB_MAINT_H

#define NPB_MAINT_NPARAM 16
#define NPB_MAINT_NACT 18

/* parameter catalog: X(id, "name in the state log") */
#define NPB_MAINT_PARAMS(X) \
  X(OIL_LEVEL,               "oil_level") \
  X(OIL_CONTAMINATION_LEVEL, "oil_contamination_level") \
  X(LUBRICATION_EFFECTIVENESS, "lubrication_effectiveness") \
  X(IMPELLER_WEAR,           "impeller_wear") \
  X(CAVITATION_DAMAGE,       "cavitation_damage") \
  X(CAVITATION_INTENSITY,    "cavitation_intensity") \
  X(NPSH_AVAILABLE,          "npsh_available") \
  X(MOTOR_BEARING_WEAR,      "motor_bearing_wear") \
  X(PUMP_BEARING_WEAR,       "pump_bearing_wear") \
  X(THRUST_BEARING_WEAR,     "thrust_bearing_wear") \
  X(SEAL_WEAR,               "seal_wear") \
  X(VIBRATION_LEVEL,         "vibration_level") \
  X(OIL_TEMPERATURE,         "oil_temperature") \
  X(MOTOR_TEMPERATURE,       "motor_temperature") \
  X(SEAL_LEAKAGE_RATE,       "seal_leakage_rate") \
  X(SUM_WEAR_LEVEL,          "sum_wear_level")

/* action catalog: X(id, "MaintenanceActionType value", has_handler) -- has_handler = the lubrication system's
 * perform_maintenance knows it (pump_lubrication.py:642-656); the others execute as "Unknown maintenance type",
 * change no plant state, but are created, queued and counted like any other work order */
#define NPB_MAINT_ACTIONS(X) \
  X(OIL_CHANGE,               "oil_change", 1) \
  X(OIL_TOP_OFF,              "oil_top_off", 1) \
  X(LUBRICATION_SYSTEM_CHECK, "lubrication_system_check", 1) \
  X(IMPELLER_INSPECTION,      "impeller_inspection", 1) \
  X(IMPELLER_REPLACEMENT,     "impeller_replacement", 1) \
  X(CAVITATION_ANALYSIS,      "cavitation_analysis", 0) \
  X(NPSH_ANALYSIS,            "npsh_analysis", 0) \
  X(BEARING_REPLACEMENT,      "bearing_replacement", 1) \
  X(SEAL_REPLACEMENT,         "seal_replacement", 1) \
  X(VIBRATION_ANALYSIS,       "vibration_analysis", 1) \
  X(LUBRICATION_INSPECTION,   "lubrication_inspection", 0) \
  X(MOTOR_INSPECTION,         "motor_inspection", 1) \
  X(COMPONENT_OVERHAUL,       "component_overhaul", 1) \
  X(COMPREHENSIVE_SYSTEM_INSPECTION, "comprehensive_system_inspection", 0) \
  X(BEARING_INSPECTION,       "bearing_inspection", 1) \
  X(OIL_ANALYSIS,             "oil_analysis", 1) \
  X(SYSTEM_CLEANING,          "system_cleaning", 1) \
  X(ROUTINE_MAINTENANCE,      "routine_maintenance", 0)

/* system_cleaning is a handler of the lubrication system but not a value of the reference's MaintenanceActionType enum
 * (systems/maintenance/maintenance_actions.py): a threshold naming it fires, is recorded, and creates no work order */
#define NPB_MAINT_ACTION_IS_TYPE(a) ((a) != NPB_MA_SYSTEM_CLEANING)

enum {
#define NPB__X(id, name) NPB_MP_##id,
  NPB_MAINT_PARAMS(NPB__X)
#undef NPB__X
  NPB_MP_COUNT_
};
enum {
#define NPB__X(id, name, handler) NPB_MA_##id,
  NPB_MAINT_ACTIONS(NPB__X)
#undef NPB__X
  NPB_MA_COUNT_
};
enum { NPB_CMP_GREATER_THAN = 0, NPB_CMP_LESS_THAN = 1, NPB_CMP_GREATER_EQUAL = 2, NPB_CMP_LESS_EQUAL = 3, NPB_CMP_EQUALS = 4, NPB_CMP_NOT_EQUALS = 5 };
enum { NPB_PRIO_LOW = 1, NPB_PRIO_MEDIUM = 2, NPB_PRIO_HIGH = 3, NPB_PRIO_CRITICAL = 4, NPB_PRIO_EMERGENCY = 5 };
enum { NPB_BEARING_ALL = 0, NPB_BEARING_MOTOR = 1, NPB_BEARING_PUMP = 2, NPB_BEARING_THRUST = 3 };

typedef struct npb_maint_table_t {
  double threshold[NPB_MAINT_NPARAM];
  double cooldown_hours[NPB_MAINT_NPARAM];
  int rank[NPB_MAINT_NPARAM];        /* position in the configuration's dict order; -1 = no such threshold */
  int comparison[NPB_MAINT_NPARAM];
  int action[NPB_MAINT_NPARAM];
  int priority[NPB_MAINT_NPARAM];
  int bearing[NPB_MAINT_NPARAM];     /* threshold's component_id for bearing_replacement (NPB_BEARING_*; 0 = none) */
} npb_maint_table_t;

static inline void npb_maint_table_default(npb_maint_table_t *t) {
#define NPB__T(rk, id, thr, cmp, act, cool, prio, brg) \
  t->rank[NPB_MP_##id] = (rk); t->threshold[NPB_MP_##id] = (thr); t->comparison[NPB_MP_##id] = NPB_CMP_##cmp; \
  t->action[NPB_MP_##id] = NPB_MA_##act; t->cooldown_hours[NPB_MP_##id] = (cool); t->priority[NPB_MP_##id] = NPB_PRIO_##prio; \
  t->bearing[NPB_MP_##id] = NPB_BEARING_##brg;
  NPB__T(0,  OIL_LEVEL,               58.0, LESS_THAN,    OIL_TOP_OFF,              168.0,  HIGH,     ALL)
  NPB__T(1,  OIL_CONTAMINATION_LEVEL, 15.2, GREATER_THAN, OIL_CHANGE,               720.0,  MEDIUM,   ALL)
  NPB__T(2,  LUBRICATION_EFFECTIVENESS, 0.35, LESS_THAN,  LUBRICATION_SYSTEM_CHECK, 720.0,  MEDIUM,   ALL)
  NPB__T(3,  IMPELLER_WEAR,           8.0,  GREATER_THAN, IMPELLER_INSPECTION,      2190.0, MEDIUM,   ALL)
  NPB__T(4,  CAVITATION_DAMAGE,       8.0,  GREATER_THAN, IMPELLER_REPLACEMENT,     4380.0, HIGH,     ALL)
  NPB__T(5,  CAVITATION_INTENSITY,    0.25, GREATER_THAN, CAVITATION_ANALYSIS,      168.0,  HIGH,     ALL)
  NPB__T(6,  NPSH_AVAILABLE,          18.0, LESS_THAN,    NPSH_ANALYSIS,            168.0,  HIGH,     ALL)
  NPB__T(7,  MOTOR_BEARING_WEAR,      8.5,  GREATER_THAN, BEARING_REPLACEMENT,      2190.0, HIGH,     MOTOR)
  NPB__T(8,  PUMP_BEARING_WEAR,       6.5,  GREATER_THAN, BEARING_REPLACEMENT,      2190.0, HIGH,     PUMP)
  NPB__T(9,  THRUST_BEARING_WEAR,     4.5,  GREATER_THAN, BEARING_REPLACEMENT,      2190.0, CRITICAL, THRUST)
  NPB__T(10, SEAL_WEAR,               16.0, GREATER_THAN, SEAL_REPLACEMENT,         2190.0, MEDIUM,   ALL)
  NPB__T(11, VIBRATION_LEVEL,         20.0, GREATER_THAN, VIBRATION_ANALYSIS,       168.0,  HIGH,     ALL)
  NPB__T(12, OIL_TEMPERATURE,         55.0, GREATER_THAN, LUBRICATION_INSPECTION,   168.0,  MEDIUM,   ALL)
  NPB__T(13, MOTOR_TEMPERATURE,       90.0, GREATER_THAN, MOTOR_INSPECTION,         168.0,  MEDIUM,   ALL)
  NPB__T(14, SEAL_LEAKAGE_RATE,       0.15, GREATER_THAN, SEAL_REPLACEMENT,         2190.0, MEDIUM,   ALL)
  NPB__T(15, SUM_WEAR_LEVEL,          75.0, GREATER_THAN, COMPONENT_OVERHAUL,       4380.0, HIGH,     ALL)
#undef NPB__T
}

#endif /* NPB_MAINT_H */
