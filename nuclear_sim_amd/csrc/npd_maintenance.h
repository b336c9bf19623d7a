/*
 * npd_maintenance.h -- device side of the automatic maintenance of the feedwater pumps.
 *
 * What the reference does, in the order NuclearPlantSimulator.step calls it (sim.py:208-223):
 *   AutoMaintenanceSystem.update(t)   systems/maintenance/auto_maintenance.py:200-236, :468-580
 *     every check_interval (first call immediately) walk the SCHEDULED work orders in creation order and execute the
 *     ones that are due -- but _execute_work_order ends in state_manager.record_maintenance_result, which raises
 *     (state_manager.py:1657) and sim.py:215 swallows it, so exactly ONE due order is executed per check, and
 *     maintenance_actions_performed (not work_orders_executed) counts it.  Execution = the pump's perform_maintenance,
 *     i.e. the lubrication system's dispatcher (feedwater/pump_lubrication.py:625-674): thirteen handlers; any other
 *     action type "succeeds" at nothing but is queued and counted all the same.
 *   StateManager.collect_states(t)    simulator/state/state_manager.py:1307-1369
 *     per pump: every configured threshold whose parameter resolves in the pump's state log, outside its own
 *     cooldown, is compared; the violations of one pump go to the orchestrator
 *     (maintenance_orchestrator.py:81-140,188-330: promote to component_overhaul / comprehensive_system_inspection,
 *     coordinate, promote by rule, else the first violation's action), and ONE event per pump reaches
 *     AutoMaintenanceSystem._handle_state_manager_threshold (:239-329), which creates a work order unless the same
 *     (pump, action) was triggered less than work_order_cooldown_hours ago -- a number of hours the reference compares
 *     with minutes -- or the pump already has an open order with that action (:331-456).
 * Array indices are the parameter / action catalog of include/npb_maint.h.
 */
#ifndef NPD_MAINTENANCE_H
#define NPD_MAINTENANCE_H
#include "npd_common.h"
#include "npd_feedwater.h"
#include "../../include/npb_maint.h"

/* action types the lubrication system's dispatcher has a handler for (include/npb_maint.h, third column) */
static constexpr uint32_t NPD_MA_HANDLER_MASK = 0u
#define NPB__X(id, name, handler) | ((uint32_t)(handler) << NPB_MA_##id)
    NPB_MAINT_ACTIONS(NPB__X)
#undef NPB__X
    ;

/* ---- the pump's state log, as far as thresholds read it  pump_system.py:1062-1086, pump_lubrication.py:1582-1638 */
NPD_FN void npd_maint_values(const npb_pump_t *p, double *v) {
  v[NPB_MP_OIL_LEVEL] = p->oil_level;
  v[NPB_MP_OIL_CONTAMINATION_LEVEL] = p->oil_contamination;
  v[NPB_MP_LUBRICATION_EFFECTIVENESS] = p->lubrication_effectiveness;
  v[NPB_MP_IMPELLER_WEAR] = p->wear_impeller;
  v[NPB_MP_CAVITATION_DAMAGE] = p->cavitation_damage;
  v[NPB_MP_CAVITATION_INTENSITY] = p->cavitation_intensity;
  v[NPB_MP_NPSH_AVAILABLE] = p->npsh_available;
  v[NPB_MP_MOTOR_BEARING_WEAR] = p->wear_motor_bearings;
  v[NPB_MP_PUMP_BEARING_WEAR] = p->wear_pump_bearings;
  v[NPB_MP_THRUST_BEARING_WEAR] = p->wear_thrust_bearing;
  v[NPB_MP_SEAL_WEAR] = p->wear_mechanical_seals;
  v[NPB_MP_VIBRATION_LEVEL] = p->vibration_level;
  v[NPB_MP_OIL_TEMPERATURE] = p->oil_temperature;
  v[NPB_MP_MOTOR_TEMPERATURE] = p->motor_temperature;
  v[NPB_MP_SEAL_LEAKAGE_RATE] = p->seal_leakage_rate;
  const double max_bearing_wear = npd_pymax3(p->wear_motor_bearings, p->wear_pump_bearings, p->wear_thrust_bearing);
  v[NPB_MP_SUM_WEAR_LEVEL] = p->wear_impeller + max_bearing_wear + p->wear_mechanical_seals;
}

/* StateManager._check_threshold_condition  state_manager.py:1413-1442 */
NPD_FN int npd_maint_violates(double value, double threshold, int comparison) {
  if (comparison == NPB_CMP_GREATER_THAN) return value > threshold;
  if (comparison == NPB_CMP_LESS_THAN) return value < threshold;
  if (comparison == NPB_CMP_GREATER_EQUAL) return value >= threshold;
  if (comparison == NPB_CMP_LESS_EQUAL) return value <= threshold;
  if (comparison == NPB_CMP_EQUALS) return fabs(value - threshold) < 0.001;
  return fabs(value - threshold) >= 0.001;
}

/* ---- MaintenanceOrchestrator._make_maintenance_decision for component type 'feedwater_pump'
 * maintenance_orchestrator.py:188-330 with the hierarchy of :472-524.  viol = bit mask over catalogued parameters,
 * values = their current values, requested = the first violation's action (state_manager.py:1553). */
NPD_FN int npd_maint_in_set(int action, uint32_t set) { return (set >> action) & 1u; }
#define NPD_MA_BIT(a) (1u << NPB_MA_##a)
NPD_FN int npd_maint_decide(const npb_maint_table_t *T, uint32_t viol, const double *values, int requested) {
  /* _check_comprehensive_promotion: component_overhaul, then comprehensive_system_inspection (dict order) */
  const uint32_t overhaul_set = NPD_MA_BIT(BEARING_REPLACEMENT) | NPD_MA_BIT(SEAL_REPLACEMENT) | NPD_MA_BIT(OIL_CHANGE) |
                                NPD_MA_BIT(MOTOR_INSPECTION) | NPD_MA_BIT(IMPELLER_REPLACEMENT) | NPD_MA_BIT(BEARING_INSPECTION) |
                                NPD_MA_BIT(OIL_ANALYSIS) | NPD_MA_BIT(VIBRATION_ANALYSIS);
  const uint32_t inspection_set = NPD_MA_BIT(BEARING_INSPECTION) | NPD_MA_BIT(MOTOR_INSPECTION) | NPD_MA_BIT(IMPELLER_INSPECTION) |
                                  NPD_MA_BIT(VIBRATION_ANALYSIS) | NPD_MA_BIT(OIL_ANALYSIS) | NPD_MA_BIT(LUBRICATION_INSPECTION);
  int total = 0, in_overhaul = 0, in_inspection = 0;
  uint32_t actions = 0;   /* set of the violations' actions */
#pragma unroll
  for (int k = 0; k < NPB_MAINT_NPARAM; k++) {
    if (!((viol >> k) & 1u)) continue;
    total++;
    in_overhaul += npd_maint_in_set(T->action[k], overhaul_set);
    in_inspection += npd_maint_in_set(T->action[k], inspection_set);
    actions |= 1u << T->action[k];
  }
  /* trigger conditions 'bearing_wear_threshold' / 'system_health_factor_threshold' name parameters ('bearing_wear',
   * 'system_health_factor') that no threshold carries, so they never match a violation (:330-337) */
  if (in_overhaul >= 2 || total >= 4) return NPB_MA_COMPONENT_OVERHAUL;
  if (in_inspection >= 3 || total >= 5) return NPB_MA_COMPREHENSIVE_SYSTEM_INSPECTION;
  /* _check_action_coordination keeps the requested action (:260-283) */
  if (requested == NPB_MA_BEARING_REPLACEMENT && (actions & (NPD_MA_BIT(OIL_CHANGE) | NPD_MA_BIT(OIL_ANALYSIS) | NPD_MA_BIT(VIBRATION_ANALYSIS)))) return requested;
  if (requested == NPB_MA_IMPELLER_REPLACEMENT && (actions & (NPD_MA_BIT(CAVITATION_ANALYSIS) | NPD_MA_BIT(NPSH_ANALYSIS) | NPD_MA_BIT(BEARING_INSPECTION)))) return requested;
  if (requested == NPB_MA_SEAL_REPLACEMENT && (actions & (NPD_MA_BIT(OIL_ANALYSIS) | NPD_MA_BIT(LUBRICATION_INSPECTION)))) return requested;
  if (requested == NPB_MA_MOTOR_INSPECTION && (actions & (NPD_MA_BIT(BEARING_INSPECTION) | NPD_MA_BIT(VIBRATION_ANALYSIS)))) return requested;
  /* _check_action_promotion (:285-310, conditions :339-357: "param > x" holds when a VIOLATION of that parameter has a
   * larger value; 'bearing_wear', 'vibration_increase', 'oil_acidity_number' are not threshold parameters) */
  const int motor_hot = ((viol >> NPB_MP_MOTOR_TEMPERATURE) & 1u) && values[NPB_MP_MOTOR_TEMPERATURE] > 80.0;
  const int dirty12 = ((viol >> NPB_MP_OIL_CONTAMINATION_LEVEL) & 1u) && values[NPB_MP_OIL_CONTAMINATION_LEVEL] > 12.0;
  const int dirty15 = ((viol >> NPB_MP_OIL_CONTAMINATION_LEVEL) & 1u) && values[NPB_MP_OIL_CONTAMINATION_LEVEL] > 15.0;
  if (requested == NPB_MA_OIL_CHANGE && motor_hot) return NPB_MA_BEARING_REPLACEMENT;
  if (requested == NPB_MA_OIL_TOP_OFF && dirty12) return NPB_MA_OIL_CHANGE;
  if (requested == NPB_MA_LUBRICATION_SYSTEM_CHECK && dirty12) return NPB_MA_OIL_CHANGE;
  if (requested == NPB_MA_OIL_ANALYSIS && dirty15) return NPB_MA_OIL_CHANGE;
  return requested;
}

/* AutoMaintenanceSystem._calculate_start_time  auto_maintenance.py:458-466 */
NPD_FN double npd_maint_start_time(const npb_params_t *P, double t, int priority) {
  if (priority == NPB_PRIO_EMERGENCY) return t + (P->maint_emergency_delay_hours * 60);
  if (priority == NPB_PRIO_CRITICAL) return t + (P->maint_start_delay_hours * 0.5 * 60);
  if (priority == NPB_PRIO_HIGH) return t + (P->maint_start_delay_hours * 60);
  if (priority == NPB_PRIO_MEDIUM) return t + (P->maint_medium_delay_hours * 60);
  return t + (P->maint_low_delay_hours * 60);
}

/* ---- one pump's part of StateManager._check_maintenance_thresholds + _emit_batched_threshold_violation +
 * AutoMaintenanceSystem._handle_state_manager_threshold / _create_automatic_work_order.  Returns 1 when mp changed. */
NPD_FN int npd_maint_scan_pump(npb_mpump_t *mp, npb_maint_t *m, const npb_params_t *P, const npb_maint_table_t *T,
                               const npb_pump_t *p, double t) {
  double values[NPB_MAINT_NPARAM];
  npd_maint_values(p, values);
  uint32_t viol = 0;
  int first_rank = 1 << 30, requested = -1, priority = 0;
#pragma unroll
  for (int k = 0; k < NPB_MAINT_NPARAM; k++) {
    if (T->rank[k] < 0) continue;
    /* _is_threshold_in_cooldown  state_manager.py:1267-1293 */
    if (mp->last_violation_time[k] >= 0.0 && t - mp->last_violation_time[k] < T->cooldown_hours[k] * 60) continue;
    if (!npd_maint_violates(values[k], T->threshold[k], T->comparison[k])) continue;
    mp->last_violation_time[k] = t;
    viol |= 1u << k;
    if (T->rank[k] < first_rank) { first_rank = T->rank[k]; requested = T->action[k]; }
    if (T->priority[k] > priority) priority = T->priority[k];     /* batched event: the highest priority (:1599) */
  }
  if (!viol) return 0;
  const int action = npd_maint_decide(T, viol, values, requested);
  /* the bearing named by the first violation (dict order) whose action is the selected one (auto_maintenance.py:257-266) */
  int bearing = NPB_BEARING_ALL, bearing_rank = 1 << 30;
#pragma unroll
  for (int k = 0; k < NPB_MAINT_NPARAM; k++)
    if (((viol >> k) & 1u) && T->action[k] == action && T->bearing[k] != NPB_BEARING_ALL && T->rank[k] < bearing_rank) {
      bearing_rank = T->rank[k]; bearing = T->bearing[k];
    }
  /* duplicate prevention  auto_maintenance.py:344-367: hours compared with minutes, then an open order with this action */
  /* _create_automatic_work_order first converts the name to a MaintenanceActionType and gives up when there is none
   * (auto_maintenance.py:336-343) */
  int create = NPB_MAINT_ACTION_IS_TYPE(action);
#pragma unroll
  for (int a = 0; a < NPB_MAINT_NACT; a++) {
    if (a != action) continue;
    if (mp->last_trigger_time[a] >= 0.0 && t - mp->last_trigger_time[a] < P->maint_work_order_cooldown) create = 0;
    if (mp->wo_order[a] > 0.0) create = 0;
  }
  if (create) {
    m->work_orders_created += 1;
#pragma unroll
    for (int a = 0; a < NPB_MAINT_NACT; a++) {
      if (a != action) continue;
      mp->wo_order[a] = (double)m->work_orders_created;
      mp->wo_planned_start[a] = npd_maint_start_time(P, t, priority);
      mp->last_trigger_time[a] = t;
    }
    if (action == NPB_MA_BEARING_REPLACEMENT) mp->wo_bearing = (double)bearing;
  }
  return 1;
}

/* ---- the lubrication system's handlers  feedwater/pump_lubrication.py:676-1410 (state effects only) */
NPD_FN void npd_oil_top_off(npb_pump_t *p, double target_level) {   /* _perform_oil_top_off :710-753 */
  double oil_added = npd_pymax(0.0, target_level - p->oil_level);
  if (oil_added > 0) {
    p->oil_level = npd_pymin(100.0, target_level);
    double dilution_factor = oil_added / 100.0;
    p->oil_contamination *= (1.0 - dilution_factor * 0.5);
    p->oil_acidity *= (1.0 - dilution_factor * 0.3);
    p->oil_moisture *= (1.0 - dilution_factor * 0.4);
    npd_pump_lubrication_effectiveness(p);
    npd_pump_performance_factors(p, 0.0); /* default cavitation_damage argument */
  }
}
NPD_FN void npd_maint_execute(npb_pump_t *p, const npb_params_t *P, int action, int bearing) {
  if (action == NPB_MA_OIL_CHANGE) {                               /* _perform_oil_change :676-708 */
    p->oil_level = 100.0; p->oil_temperature = 40.0; p->oil_contamination = 5.0; p->oil_acidity = 0.5; p->oil_moisture = 0.02;
    npd_pump_lubrication_effectiveness(p);
    npd_pump_performance_factors(p, 0.0);
    p->seal_leakage_rate = npd_pymax(0.0, p->seal_leakage_rate * 0.5);
  } else if (action == NPB_MA_OIL_TOP_OFF) {
    npd_oil_top_off(p, P->maint_top_off_target);
  } else if (action == NPB_MA_BEARING_REPLACEMENT) {               /* _perform_bearing_replacement :755-808 */
    double total_wear_removed;
    if (bearing == NPB_BEARING_ALL) {
      total_wear_removed = p->wear_motor_bearings + p->wear_pump_bearings + p->wear_thrust_bearing;
      p->wear_motor_bearings = 0.0; p->wear_pump_bearings = 0.0; p->wear_thrust_bearing = 0.0;
    } else if (bearing == NPB_BEARING_MOTOR) { total_wear_removed = p->wear_motor_bearings; p->wear_motor_bearings = 0.0; }
    else if (bearing == NPB_BEARING_PUMP) { total_wear_removed = p->wear_pump_bearings; p->wear_pump_bearings = 0.0; }
    else { total_wear_removed = p->wear_thrust_bearing; p->wear_thrust_bearing = 0.0; }
    npd_pump_lubrication_effectiveness(p);
    npd_pump_performance_factors(p, 0.0);
    p->vibration_increase = npd_pymax(0.0, p->vibration_increase - total_wear_removed * 0.1);
  } else if (action == NPB_MA_SEAL_REPLACEMENT) {                  /* _perform_seal_replacement :810-838 */
    p->wear_mechanical_seals = 0.0; p->seal_leakage_rate = 0.0;
    npd_pump_lubrication_effectiveness(p);
    npd_pump_performance_factors(p, 0.0);
  } else if (action == NPB_MA_COMPONENT_OVERHAUL) {                /* _perform_component_overhaul :840-881 */
    p->wear_impeller = 0.0; p->wear_motor_bearings = 0.0; p->wear_pump_bearings = 0.0; p->wear_thrust_bearing = 0.0;
    p->wear_mechanical_seals = 0.0; p->wear_coupling_system = 0.0;
    p->oil_level = 100.0; p->oil_temperature = 40.0; p->oil_contamination = 5.0; p->oil_acidity = 0.5; p->oil_moisture = 0.02;
    p->seal_leakage_rate = 0.0; p->vibration_increase = 0.0;
    npd_pump_lubrication_effectiveness(p);
    npd_pump_performance_factors(p, 0.0);
  } else if (action == NPB_MA_SYSTEM_CLEANING) {                   /* _perform_system_cleaning :883-915 */
    double old_contamination = p->oil_contamination;
    double contamination_reduction = npd_pymin(old_contamination * 0.7, 50.0);
    p->oil_contamination = npd_pymax(5.0, old_contamination - contamination_reduction);
    p->oil_acidity *= 0.8; p->oil_moisture *= 0.9;
    p->wear_impeller = npd_pymax(0.0, p->wear_impeller - 0.5); p->wear_motor_bearings = npd_pymax(0.0, p->wear_motor_bearings - 0.5);
    p->wear_pump_bearings = npd_pymax(0.0, p->wear_pump_bearings - 0.5); p->wear_thrust_bearing = npd_pymax(0.0, p->wear_thrust_bearing - 0.5);
    p->wear_mechanical_seals = npd_pymax(0.0, p->wear_mechanical_seals - 0.5); p->wear_coupling_system = npd_pymax(0.0, p->wear_coupling_system - 0.5);
    npd_pump_lubrication_effectiveness(p);
    npd_pump_performance_factors(p, 0.0);
  } else if (action == NPB_MA_BEARING_INSPECTION) {                /* _perform_bearing_inspection :917-955 */
    if (npd_pymax3(p->wear_motor_bearings, p->wear_pump_bearings, p->wear_thrust_bearing) > 5.0) {
      p->wear_motor_bearings *= 0.9; p->wear_pump_bearings *= 0.9; p->wear_thrust_bearing *= 0.9;
      npd_pump_performance_factors(p, 0.0);
    }
  } else if (action == NPB_MA_IMPELLER_INSPECTION) {               /* _perform_impeller_inspection :957-1035 */
    double impeller_wear = p->wear_impeller;
    double max_bearing_wear = npd_pymax3(p->wear_motor_bearings, p->wear_pump_bearings, p->wear_thrust_bearing);
    if (impeller_wear > 3.0 || max_bearing_wear > 5.0) {
      p->wear_impeller = npd_pymax(0.0, impeller_wear * 0.9);
      p->wear_motor_bearings = npd_pymax(0.0, p->wear_motor_bearings - 0.5);
      p->wear_pump_bearings = npd_pymax(0.0, p->wear_pump_bearings - 0.5);
      p->wear_thrust_bearing = npd_pymax(0.0, p->wear_thrust_bearing - 0.5);
      npd_pump_performance_factors(p, 0.0);
    }
  } else if (action == NPB_MA_IMPELLER_REPLACEMENT) {              /* _perform_impeller_replacement :1037-1090 */
    double old_impeller_wear = p->wear_impeller;
    p->wear_impeller = 0.0;
    npd_pump_performance_factors(p, 0.0);
    p->vibration_increase = npd_pymax(0.0, p->vibration_increase - old_impeller_wear * 0.08);
  } else if (action == NPB_MA_LUBRICATION_SYSTEM_CHECK) {          /* _perform_lubrication_system_check :1092-1260 */
    if (p->oil_level < 95.0) {
      double target_level = npd_pymin(95.0, p->oil_level + 5.0);
      double oil_added = target_level - p->oil_level;
      p->oil_level = target_level;
      if (oil_added > 0) {
        double dilution_factor = oil_added / 100.0;
        double contamination_dilution = dilution_factor * 0.5;
        p->oil_contamination = npd_pymax(1.0, p->oil_contamination * (1.0 - contamination_dilution));
        double additive_boost = dilution_factor * 15.0;
        p->antioxidant_level = npd_pymin(100.0, p->antioxidant_level + additive_boost);
        p->anti_wear_level = npd_pymin(100.0, p->anti_wear_level + additive_boost * 0.8);
      }
    }
    double contamination_reduction = npd_pymin(p->oil_contamination * 0.3, 5.0);
    p->oil_contamination = npd_pymax(1.0, p->oil_contamination - contamination_reduction);
    const double additive_restoration = 15.0;
    p->antioxidant_level = npd_pymin(100.0, p->antioxidant_level + additive_restoration);
    p->anti_wear_level = npd_pymin(100.0, p->anti_wear_level + additive_restoration * 0.8);
    p->corrosion_inhibitor_level = npd_pymin(100.0, p->corrosion_inhibitor_level + additive_restoration * 0.6);
    p->wear_motor_bearings = npd_pymax(0.0, p->wear_motor_bearings - 0.5);
    p->wear_pump_bearings = npd_pymax(0.0, p->wear_pump_bearings - 0.5);
    p->wear_thrust_bearing = npd_pymax(0.0, p->wear_thrust_bearing - 0.5);
    p->seal_leakage_rate = npd_pymax(0.0, p->seal_leakage_rate * 0.9);
    npd_pump_lubrication_effectiveness(p);                         /* the performance factors are NOT recomputed here */
  } else if (action == NPB_MA_MOTOR_INSPECTION) {                  /* _perform_motor_inspection :1262-1290 */
    if (p->wear_motor_bearings > 3.0) { p->wear_motor_bearings *= 0.95; npd_pump_performance_factors(p, 0.0); }
  }
  /* oil_analysis :1292-1335 and vibration_analysis :1337-1385 read state only; every other action type is
   * "Unknown maintenance type" (:661-668) */
}

/* ---- AutoMaintenanceSystem.update: is a check due, and which open order (pump, action) is executed at it */
NPD_FN int npd_maint_check_due(npb_maint_t *m, const npb_params_t *P, double t) {
  double check_interval_minutes = P->maint_check_interval_hours * 60;
  if (m->last_check_time > 0.0 && t - m->last_check_time < check_interval_minutes) return 0;
  m->last_check_time = t;
  return 1;
}
/* the earliest-created due order of one pump: returns its order number (0 = none) and its action */
NPD_FN double npd_maint_first_due(const npb_mpump_t *mp, double t, int *action) {
  double best = 0.0; int act = -1;
#pragma unroll
  for (int a = 0; a < NPB_MAINT_NACT; a++) {
    int due = mp->wo_order[a] > 0.0 && mp->wo_planned_start[a] != 0.0 && t >= mp->wo_planned_start[a];
    if (due && (best == 0.0 || mp->wo_order[a] < best)) { best = mp->wo_order[a]; act = a; }
  }
  *action = act;
  return best;
}
/* completed orders leave WorkOrderManager.work_orders (work_orders.py:341-350) */
NPD_FN int npd_maint_close_order(npb_mpump_t *mp, npb_maint_t *m, int action) {
  int bearing = NPB_BEARING_ALL;
#pragma unroll
  for (int a = 0; a < NPB_MAINT_NACT; a++) {
    if (a != action) continue;
    mp->wo_order[a] = 0.0; mp->wo_planned_start[a] = 0.0;
    m->executed[a] += 1.0;
  }
  if (action == NPB_MA_BEARING_REPLACEMENT) { bearing = (int)mp->wo_bearing; mp->wo_bearing = 0.0; }
  m->maintenance_actions_performed += 1;
  return bearing;
}

/* ---- the threshold screen INSIDE the step kernels.  What nearly every step of nearly every plant ends with is "nothing new":
 * no threshold is crossed, or the crossed ones are inside their cooldown (a running pump's oil temperature sits above its
 * 55 C row for the whole run, one violation per week).  The pump phase of the step holds the sixteen values the thresholds
 * look at in registers when it has just updated a pump, so it answers that question there -- for (64 plants, pump) at a time,
 * one flag word -- instead of a separate launch reading them back (0.5 KB per plant and step, which at 65 536 plants pushed
 * the step's working set further past the Infinity Cache and slowed the step kernel itself by a quarter).
 *   The table is folded by the host (npb_kernels.hip, npd_maint_fold_table) so that a strict comparison "value > threshold" /
 * "value < threshold" is the sign of d = fma(value, sgn, c) with (sgn, c) = (+1, -threshold) / (-1, +threshold): exact -- the
 * difference of two distinct doubles never rounds to zero, a NaN value gives a NaN that max() drops as `NaN > x` is false --
 * and a row outside the scan has (0, -1).  Rows with any other comparison (>=, <=, ==, !=) make the whole table "always"
 * flagged and leave the decision to the rule kernel's own evaluation.
 *   Cooldowns: which rows of a (plant, pump) are inside their cooldown, and until when at the earliest, is kept in a side
 * cache the handle owns (NOT plant state: it is a function of the mpump.last_violation_time stamps, the table and the clock,
 * rebuilt by the rule kernel whenever it looks at a wave, and zeroed -- "look now" -- by every call that can change any of
 * those).  A cooling row's d gets its sign bit set; the pump is flagged when a row outside the mask is crossed or when the
 * earliest expiry has come (taken a hair early: the rule kernel then decides with the reference's own comparison).
 * A flag only says "look": cooldowns, the orchestrator and the work orders are the rule kernel's (npb_maint_kernel), which
 * re-evaluates the flagged waves from the stored state.
 *   tab[0..15] sgn   tab[16..31] c   tab[32] != 0: always flagged   tab[33] check interval [min] */
#define NPD_MH_N 34
struct npd_maint_hot_t { double tab[NPD_MH_N]; };
/* flag words per wave of 64 plants: [0..3] pump k has a lane with something new, [4] a lane whose check is due finds open
 * work orders; [5..7] unused */
#define NPD_MAINT_FLAG_WORDS 8
/* the side cache: entry[plant][pump] = {uint32 mask: bit q = row q of the table is inside its cooldown; float until: no masked
 * row leaves its cooldown before this time [min] (rounded DOWN to float; +inf with an empty mask; 0 = unknown, look)}: 32 bytes
 * per plant, two 16-byte loads.  The step kernels fetch them with asm loads when they start and first look at them two phases
 * later: left to the compiler the loads sink to their first use, and a lone wave then waits out a full memory latency per
 * pump (measured: +4 us at 65 536 plants). */
typedef uint32_t npd_u32x4 __attribute__((ext_vector_type(4)));
struct npd_maint_cache_t { npd_u32x4 *entry;      /* [pitch][2]: pumps 0,1 | pumps 2,3 */
                           int32_t *counts;       /* the caller's maintenance_actions_performed column (npb_set_maintenance_count_buffer), or NULL */
                           int n_plants;
                           double *diag; size_t diag_pitch;   /* npb_set_diagnostics buffer (the maintenance flags of NPB_DIAG_PUMP_*), or NULL */ };
__device__ __forceinline__ npd_u32x4 npd_maint_cache_fetch(const npd_maint_cache_t &C, size_t p, int half) {
  npd_u32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(C.entry + p * 2 + half) : "memory");
  return v;
}
/* after the wait that covers the fetch (an explicit s_waitcnt vmcnt(0), or the staging pipeline's own): ties the register to
 * this point of the instruction stream */
__device__ __forceinline__ void npd_maint_cache_landed(npd_u32x4 &v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ float npd_float_below(double x) {     /* the largest float <= x */
  float f = (float)x;
  if ((double)f > x) f = __uint_as_float(f > 0.0f ? __float_as_uint(f) - 1u : (f < 0.0f ? __float_as_uint(f) + 1u : 0x80000001u));
  return f;
}
/* tl: the table in LDS (every lane reads the same word: a broadcast).  (npd_real_t): the value as the arena will hold it
 * (rounded to float under fp32 storage), which is what the rule kernel will see.  t: the plant's clock after this step */
__device__ __forceinline__ bool npd_maint_pump_hit(const npb_pump_t *pm, const double *table_lds, uint32_t cooling_mask, float cooling_until, double t) {
  /* read where it is used, every time (volatile): hoisted out of the rolled pump loop the 34 entries would sit in 68 registers
   * across the whole pump phase, which spills */
  const volatile __attribute__((address_space(3))) double *tl = (const volatile __attribute__((address_space(3))) double *)table_lds;
  double v[NPB_MAINT_NPARAM];
  npd_maint_values(pm, v);
  double m = -1.0;
#pragma unroll
  for (int q = 0; q < NPB_MAINT_NPARAM; q++) {
    const double d = __builtin_fma((double)(npd_real_t)v[q], tl[q], tl[NPB_MAINT_NPARAM + q]);
    /* a row inside its cooldown cannot fire: its d is made negative (sign bit from bit q of the mask) */
    const uint32_t hi = (uint32_t)__double2hiint(d) | ((cooling_mask << (31 - q)) & 0x80000000u);
    m = __builtin_fmax(m, __hiloint2double((int)hi, __double2loint(d)));
  }
  return (m > 0.0) | (tl[2 * NPB_MAINT_NPARAM] != 0.0) | !(t < (double)cooling_until);
}
/* the cache entry of one (plant, pump) from its stamps, as of time t (the rule kernel, after it has looked at a wave) */
__device__ __forceinline__ void npd_maint_cache_entry(const double *last_violation_time, const double *cooldown_minutes, uint32_t scan_mask, double t,
                                                      uint32_t *mask_out, float *until_out) {
  uint32_t mask = 0; double until = __builtin_inf();
#pragma unroll
  for (int q = 0; q < NPB_MAINT_NPARAM; q++) {
    const double lv = last_violation_time[q];
    const bool cooling = ((scan_mask >> q) & 1u) && (lv >= 0.0) && (t - lv < cooldown_minutes[q]);     /* _is_threshold_in_cooldown */
    if (cooling) {
      mask |= 1u << q;
      const double ends = lv + cooldown_minutes[q];
      until = fmin(until, ends - (fabs(ends) * 1e-9 + 1e-9));     /* a hair early: rounding of (t - lv) must never hide an expiry */
    }
  }
  *mask_out = mask; *until_out = npd_float_below(until);
}
/* AutoMaintenanceSystem.update as far as it needs no order (auto_maintenance.py:200-236): a check that falls due with nothing
 * open only moves last_check_time; one that finds open orders is left, untouched, to the rule kernel.  In two parts, so that a
 * step kernel can issue the three loads when it starts and decide once the plant's clock is known: a lone wave that uses a
 * load at once waits out the whole memory latency. */
struct npd_maint_due_t { npd_real_t *last_check_time_p; double last_check_time; int created, performed; };
__device__ __forceinline__ void npd_maint_due_load(npd_maint_due_t *d, npd_real_t *f64, size_t N, size_t p) {
  d->last_check_time_p = (npd_real_t *)((char *)(f64 + (size_t)(NPD_SEC_COL(MAINT, 0) + NPB_F64_SLOT(npb_maint_t, last_check_time)) * N + p));
  const char *cnt = (const char *)(f64 + (size_t)(NPD_SEC_COL(MAINT, 0) + NPB_MAINT_NCARRY) * N + p);
  d->last_check_time = (double)*d->last_check_time_p;
  d->created = *(const int32_t *)(cnt + ((NPB_MAINT_NOUT + NPB_I32_SLOT(npb_maint_t, MAINT, work_orders_created)) / NPD_NPC) * N * sizeof(npd_real_t) +
                                  ((NPB_MAINT_NOUT + NPB_I32_SLOT(npb_maint_t, MAINT, work_orders_created)) % NPD_NPC) * 4);
  d->performed = *(const int32_t *)(cnt + ((NPB_MAINT_NOUT + NPB_I32_SLOT(npb_maint_t, MAINT, maintenance_actions_performed)) / NPD_NPC) * N * sizeof(npd_real_t) +
                                    ((NPB_MAINT_NOUT + NPB_I32_SLOT(npb_maint_t, MAINT, maintenance_actions_performed)) % NPD_NPC) * 4);
}
/* returns "work": the check is due and finds open orders */
__device__ __forceinline__ bool npd_maint_due_decide(const npd_maint_due_t *d, double t, double check_interval_minutes) {
  const bool due = !(d->last_check_time > 0.0 && t - d->last_check_time < check_interval_minutes);    /* npd_maint_check_due */
  const bool work = due & (d->created > d->performed);
  if (due & !work) *d->last_check_time_p = (npd_real_t)t;
  return work;
}

NPD_FN void npd_mpump_init(npb_mpump_t *mp) {
  memset(mp, 0, sizeof(*mp));
#pragma unroll
  for (int k = 0; k < NPB_MAINT_NPARAM; k++) mp->last_violation_time[k] = -1.0;
#pragma unroll
  for (int a = 0; a < NPB_MAINT_NACT; a++) mp->last_trigger_time[a] = -1.0;
}
NPD_FN void npd_maint_init(npb_maint_t *m) { memset(m, 0, sizeof(*m)); }

#endif
