/*
 * npd_ph.h -- device physics: the chemistry sidecar of the secondary system: shared WaterChemistry (updated
 * twice per step) + PHControlSystem.
 *
 * Follows secondary/__init__.py:634-665, feedwater/physics.py:706-714, water_chemistry.py:322-389,659-745,
 * ph_control_system.py:219-470,535-566.  Nothing here feeds an observation, reward, flow, pressure or
 * trip; it only evolves its own columns.  The controller's RNG-driven sensor noise / random failures
 * are taken at their deterministic limit (normal -> 0, random -> 1.0), see npb_fields.h.
 */
#ifndef NPD_PH_H
#define NPD_PH_H
#include "npd_common.h"
#include "npd_chem.h"

/* WaterChemistry.update_chemistry's "pending effects" prologue  water_chemistry.py:349-359 with
 * _apply_ph_control_effects :683-722 and _apply_chemical_additions :724-745 */
NPD_FN void npd_chem_apply_pending(npb_chem_t *c, npb_ph_t *ph) {
  if (!ph->has_pending_effects) return;
  const double ph_setpoint = 9.2, ph_max = 9.6;
  double ammonia_dose = ph->pending_ammonia_dose, morpholine_dose = ph->pending_morpholine_dose;
  const double system_volume_m3 = 1000.0, water_density = 1000.0;
  if (ammonia_dose > 0) {
    double concentration_increase_ppm = (ammonia_dose / 3600.0) / (system_volume_m3 * water_density) * 1e6;
    c->ph = npd_pymin(c->ph + concentration_increase_ppm * 0.1, ph_max);
  }
  if (morpholine_dose > 0) {
    double concentration_increase_ppm = (morpholine_dose / 3600.0) / (system_volume_m3 * water_density) * 1e6;
    c->ph = npd_pymin(c->ph + concentration_increase_ppm * 0.05, ph_max);
  }
  if (fabs(c->ph - ph_setpoint) > 0.01) c->ph += (ph_setpoint - c->ph) * 0.3;
  /* chemical_additions: kg/s = dose / 3600 (secondary/__init__.py:660-663) */
  c->antiscalant_concentration += (ammonia_dose / 3600.0) * 3600.0 * 0.1;
  c->corrosion_inhibitor_level += (morpholine_dose / 3600.0) * 3600.0 * 0.05;
  ph->has_pending_effects = 0;
}

/* PHControlSystem.update_system -> PHController.update_controller  ph_control_system.py:219-272 (dt in hours) */
NPD_FN void npd_ph_update(npb_ph_t *s, double current_ph, double dt) {
  double dt_minutes = dt * 60.0;
  /* _apply_sensor_dynamics :274-291 (sensor never fails in the deterministic limit) */
  double alpha = dt_minutes / (1.0 + dt_minutes);
  double filtered_ph = s->measured_ph + alpha * (current_ph - s->measured_ph);
  double drift = 0.001 * dt_minutes / 60.0;
  s->measured_ph = filtered_ph + 0.0 + drift;
  double ph_error = 9.2 - s->measured_ph;
  /* _update_alarms_and_trips :424-444: pH trip fails the controller for good */
  if ((s->measured_ph < 8.5 || s->measured_ph > 10.0) && s->controller_enabled) s->controller_enabled = 0;
  double controller_output;
  if (s->controller_enabled) { /* _calculate_pid_output :293-325 */
    double proportional = 2.0 * ph_error;
    s->integral_sum += ph_error * dt_minutes;
    s->integral_sum = npd_clip(s->integral_sum, -50.0, 50.0);
    double integral = 0.1 * s->integral_sum;
    double derivative = (dt_minutes > 0) ? 0.5 * ((ph_error - s->previous_error) / dt_minutes) : 0.0;
    double output = (proportional + integral + derivative);
    s->previous_error = ph_error;
    controller_output = npd_clip(output, 0.0, 100.0);
  } else {
    controller_output = 0.0;
  }
  /* _apply_rate_limiting :327-350 returns early while its history is empty, and only that function
   * ever appends to the history -> the rate limit never engages */
  s->controller_output = controller_output;
  /* _calculate_dosing_rates :352-381 (pumps never fail in the deterministic limit) */
  double ammonia_dose_rate = 0.0, morpholine_dose_rate = 0.0;
  if (s->ammonia_supply_available && controller_output > 0) ammonia_dose_rate = (controller_output / 100.0) * 5.0 * 0.95;
  if (!s->ammonia_supply_available && s->morpholine_supply_available && controller_output > 0)
    morpholine_dose_rate = (controller_output / 100.0) * 10.0 * 0.98;
  /* _update_chemical_supplies :383-407 */
  if (ammonia_dose_rate > 0) s->ammonia_tank_level = npd_pymax(0.0, s->ammonia_tank_level - ((ammonia_dose_rate * dt) / 1000.0) * 100.0);
  if (morpholine_dose_rate > 0) s->morpholine_tank_level = npd_pymax(0.0, s->morpholine_tank_level - ((morpholine_dose_rate * dt) / 2000.0) * 100.0);
  s->ammonia_supply_available = s->ammonia_tank_level > 5.0;
  s->morpholine_supply_available = s->morpholine_tank_level > 5.0;
  /* update_chemistry_effects :659-681: stored for the next update_chemistry call */
  s->pending_ammonia_dose = ammonia_dose_rate;
  s->pending_morpholine_dose = morpholine_dose_rate;
  s->has_pending_effects = 1;
}

/* the whole sidecar for one step: update #1 happens inside the feedwater system (with the effects left
 * pending by the previous step), update #2 and the controller after the condenser; nothing else reads
 * this state, so they are run back to back */
NPD_FN void npd_chemistry_sidecar(npb_chem_t *c, npb_ph_t *ph, double dt, double *aggressiveness_in_feedwater = nullptr) {
  npd_chem_apply_pending(c, ph);
  npd_chem_update(c, dt);   /* feedwater/physics.py:708 */
  if (aggressiveness_in_feedwater) *aggressiveness_in_feedwater = c->water_aggressiveness;   /* what the feedwater system's performance factor sees (state-log diagnostics) */
  npd_chem_update(c, dt);   /* secondary/__init__.py:644 */
  npd_ph_update(ph, c->ph, dt); /* :647-650, dt taken as hours */
}

#endif
