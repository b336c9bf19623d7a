/*
 * npo_maintenance.h -- CPU oracle: the automatic-maintenance rule for the feedwater pumps'
 * oil_top_off action.  TEST INFRASTRUCTURE ONLY (see npo_common.h).
 *
 * The reference spreads this over an event bus, a work-order manager and the state manager; what
 * a data-generation run actually observes for the oil_level threshold is the rule below, restated
 * from the call order of NuclearPlantSimulator.step (sim.py:208-223):
 *   physics -> AutoMaintenanceSystem.update(t) -> StateManager.collect_states(t)
 * with t = elapsed minutes after this step.
 *
 *  update(t)            auto_maintenance.py:200-236
 *    runs when last_check_time == 0 or t - last_check_time >= check_interval_hours*60, and
 *    executes SCHEDULED work orders whose planned_start_date (truthy and) <= t in creation order
 *    (:468-488).  _execute_work_order (:504-580) performs the action, completes the order, then
 *    calls state_manager.record_maintenance_result, which raises AttributeError
 *    (state_manager.py:1657 reads a current_time attribute that does not exist); sim.py:215 swallows
 *    it.  Net effect: exactly ONE due work order is executed per check, maintenance_actions_performed
 *    is incremented, work_orders_executed never is.
 *  oil_top_off          pump_lubrication.py:710-744 (via FeedwaterPump.perform_maintenance
 *    pump_system.py:750-766): level -> min(100, 95), contamination/acidity/moisture diluted,
 *    _calculate_lubrication_effectiveness (:240-269) and _calculate_pump_performance_factors()
 *    with its default cavitation_damage = 0.0 (:1412-1478).
 *  collect_states(t)    state_manager.py:1307-1369: for every pump, unless the threshold is inside
 *    its cooldown (time since the last violation < cooldown_hours*60, :1267-1291), oil_level <
 *    threshold records the violation time and emits an event; the handler creates a work order
 *    (auto_maintenance.py:392-456) unless the same (component, action) was triggered less than
 *    work_order_cooldown_hours ago -- compared against MINUTES, a unit slip reproduced here -- or
 *    the pump already has an open oil_top_off order.  planned_start_date = t + HIGH-priority delay.
 *
 * Scope: only the oil_level -> oil_top_off threshold of the four feedwater pumps is restated; the
 * scenario's 40-odd other thresholds never fire in the action-test runs this is pinned against
 * (tests/golden/m1*.npz) and are the control plane the tier framing leaves out.
 */
#ifndef NPO_MAINTENANCE_H
#define NPO_MAINTENANCE_H
#include "npo_common.h"
#include "npo_plant.h"
#include "npo_feedwater.h"

/* FeedwaterPumpLubricationSystem._perform_oil_top_off  pump_lubrication.py:710-753 */
NPO_FN void npo_oil_top_off(npb_pump_t *p, double target_level) {
  double oil_added = npo_pymax(0.0, target_level - p->oil_level);
  if (oil_added > 0) {
    p->oil_level = npo_pymin(100.0, target_level);
    double dilution_factor = oil_added / 100.0;
    p->oil_contamination *= (1.0 - dilution_factor * 0.5);
    p->oil_acidity *= (1.0 - dilution_factor * 0.3);
    p->oil_moisture *= (1.0 - dilution_factor * 0.4);
    npo_pump_lubrication_effectiveness(p);
    npo_pump_performance_factors(p, 0.0);
  }
}

NPO_FN void npo_maintenance_update(npo_plant_t *pl, const npb_params_t *P) {
  npb_maint_t *m = &pl->maint;
  double t = pl->prim.sim_time; /* elapsed minutes, already advanced by this step */
  /* ---- AutoMaintenanceSystem.update */
  double check_interval_minutes = P->maint_check_interval_hours * 60;
  if (!(m->last_check_time > 0.0 && t - m->last_check_time < check_interval_minutes)) {
    m->last_check_time = t;
    int pick = -1;
    for (int k = 0; k < NPB_NUM_PUMPS; k++) {
      if (m->wo_order[k] > 0.0 && m->wo_planned_start[k] != 0.0 && t >= m->wo_planned_start[k])
        if (pick < 0 || m->wo_order[k] < m->wo_order[pick]) pick = k;
    }
    if (pick >= 0) {
      npo_oil_top_off(&pl->pump[pick], P->maint_top_off_target);
      m->maintenance_actions_performed += 1;
      m->wo_order[pick] = 0.0; /* completed orders leave WorkOrderManager.work_orders */
      m->wo_planned_start[pick] = 0.0;
    }
  }
  /* ---- StateManager._check_maintenance_thresholds + the work-order handler */
  for (int k = 0; k < NPB_NUM_PUMPS; k++) {
    if (m->last_violation_time[k] >= 0.0 && t - m->last_violation_time[k] < P->maint_oil_level_cooldown_hours * 60) continue;
    if (!(pl->pump[k].oil_level < P->maint_oil_level_threshold)) continue;
    m->last_violation_time[k] = t;
    if (m->last_trigger_time[k] >= 0.0 && t - m->last_trigger_time[k] < P->maint_work_order_cooldown) continue;
    if (m->wo_order[k] > 0.0) continue;
    m->work_orders_created += 1;
    m->wo_order[k] = (double)m->work_orders_created;
    m->wo_planned_start[k] = t + P->maint_start_delay_hours * 60;
    m->last_trigger_time[k] = t;
  }
}

#endif
