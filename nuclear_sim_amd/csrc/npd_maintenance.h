/*
 * npd_maintenance.h -- device side of the automatic oil_top_off maintenance rule.
 *
 * What the reference does, in the order NuclearPlantSimulator.step calls it (sim.py:208-223):
 *   AutoMaintenanceSystem.update(t)   auto_maintenance.py:200-236, :468-488, :504-580
 *     every check_interval (first call immediately) execute SCHEDULED work orders that are due, in
 *     creation order -- but _execute_work_order ends in state_manager.record_maintenance_result,
 *     which raises (state_manager.py:1657) and sim.py:215 swallows it, so exactly one due order is
 *     executed per check and maintenance_actions_performed (not work_orders_executed) counts it;
 *   StateManager.collect_states(t)    state_manager.py:1267-1369 + auto_maintenance.py:392-456
 *     oil_level < threshold outside the per-threshold cooldown records a violation and creates a
 *     work order unless one is open or the (component, action) trigger is younger than
 *     work_order_cooldown_hours, which the reference compares against minutes.
 * Only the feedwater pumps' oil_level threshold is covered ("maintenance").
 */
#ifndef NPD_MAINTENANCE_H
#define NPD_MAINTENANCE_H
#include "npd_common.h"
#include "npd_feedwater.h"

/* FeedwaterPumpLubricationSystem._perform_oil_top_off  pump_lubrication.py:710-753 */
NPD_FN void npd_oil_top_off(npb_pump_t *p, double target_level) {
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

/* update(t): which pump's work order is executed at this check, or -1.  Sets *dirty when m changed. */
NPD_FN int npd_maint_pick_due(npb_maint_t *m, const npb_params_t *P, double t, int *dirty) {
  double check_interval_minutes = P->maint_check_interval_hours * 60;
  if (m->last_check_time > 0.0 && t - m->last_check_time < check_interval_minutes) return -1;
  m->last_check_time = t; *dirty = 1;
  int pick = -1; double pick_order = 0.0;
#pragma unroll
  for (int k = 0; k < NPB_NUM_PUMPS; k++) {
    bool due = m->wo_order[k] > 0.0 && m->wo_planned_start[k] != 0.0 && t >= m->wo_planned_start[k];
    if (due && (pick < 0 || m->wo_order[k] < pick_order)) { pick = k; pick_order = m->wo_order[k]; }
  }
  if (pick >= 0) {
    m->maintenance_actions_performed += 1;
#pragma unroll
    for (int k = 0; k < NPB_NUM_PUMPS; k++)
      if (k == pick) { m->wo_order[k] = 0.0; m->wo_planned_start[k] = 0.0; }
  }
  return pick;
}

/* collect_states(t): threshold scan over the pumps' oil levels and work-order creation */
NPD_FN void npd_maint_scan(npb_maint_t *m, const npb_params_t *P, double t, const double *oil_level, int *dirty) {
#pragma unroll
  for (int k = 0; k < NPB_NUM_PUMPS; k++) {
    if (m->last_violation_time[k] >= 0.0 && t - m->last_violation_time[k] < P->maint_oil_level_cooldown_hours * 60) continue;
    if (!(oil_level[k] < P->maint_oil_level_threshold)) continue;
    m->last_violation_time[k] = t; *dirty = 1;
    if (m->last_trigger_time[k] >= 0.0 && t - m->last_trigger_time[k] < P->maint_work_order_cooldown) continue;
    if (m->wo_order[k] > 0.0) continue;
    m->work_orders_created += 1;
    m->wo_order[k] = (double)m->work_orders_created;
    m->wo_planned_start[k] = t + P->maint_start_delay_hours * 60;
    m->last_trigger_time[k] = t;
  }
}

NPD_FN void npd_maint_init(npb_maint_t *m) {
  memset(m, 0, sizeof(*m));
#pragma unroll
  for (int k = 0; k < NPB_NUM_PUMPS; k++) { m->last_violation_time[k] = -1.0; m->last_trigger_time[k] = -1.0; }
}

#endif
