/*
 * npo_step.h -- CPU oracle: one NuclearPlantSimulator.step() for one plant.
 * TEST INFRASTRUCTURE ONLY (see npo_common.h).
 *
 * Follows simulator/core/sim.py:130-258 (step), :290-333 (get_observation),
 * :429-498 (_apply_secondary_to_primary_feedback), :500-544 (calculate_reward).
 */
#ifndef NPO_STEP_H
#define NPO_STEP_H
#include "npo_common.h"
#include "npo_plant.h"
#include "npo_primary.h"
#include "npo_sg.h"
#include "npo_secondary.h"
#include "npo_maintenance.h"
/* thresholds of the automatic maintenance: one table per process, set through npo_set_maint_table (npo_api.c) */
static npb_maint_table_t npo_maint_table;
static int npo_maint_table_set = 0;
static int npo_maint_table_custom = 0;   /* a table given by the caller is taken as it is (as npb_set_maintenance_table) */

/* get_observation  sim.py:290-333 */
NPO_FN void npo_observation(const npo_plant_t *pl, int mode, double *obs) {
  const npb_prim_t *s = &pl->prim;
  const npb_sec_t *sec = &pl->sec;
  obs[0] = s->neutron_flux / 1e12;
  obs[1] = s->fuel_temperature / 1000;
  obs[2] = s->coolant_temperature / 300;
  obs[3] = s->coolant_pressure / 20;
  obs[4] = s->coolant_flow_rate / 50000;
  obs[5] = s->steam_temperature / 300;
  obs[6] = s->steam_pressure / 10;
  obs[7] = s->steam_flow_rate / 3000;
  obs[8] = s->control_rod_position / 100;
  obs[9] = s->steam_valve_position / 100;
  obs[10] = s->power_level / 100;
  obs[11] = (double)(s->scram_status != 0);
  if (mode == NPB_MODE_PRIMARY) { for (int k = 12; k < NPB_OBS_DIM; k++) obs[k] = 0.0; return; }   /* sim.py:333: twelve values */
  obs[12] = sec->electrical_power_output / 1100;
  obs[13] = sec->thermal_efficiency / 0.35;
  obs[14] = sec->total_steam_flow / 1665;
  obs[15] = sec->load_demand / 100;
  obs[16] = 227.0 / 250; /* secondary feedwater_temperature is always the 227.0 sim.py:166 passes */
  obs[17] = sec->cooling_water_temperature / 35;
  double fw_flow, fw_power; int fw_avail;
  npo_feedwater_obs(pl, mode, &fw_flow, &fw_power, &fw_avail);
  obs[18] = fw_flow / 1665;
  obs[19] = fw_power / 40;
  obs[20] = (double)fw_avail;
  obs[21] = fw_flow / 1665;
}

/* calculate_reward  sim.py:500-544 */
NPO_FN double npo_reward(const npo_plant_t *pl, const npo_secondary_result_t *r) {
  const npb_prim_t *s = &pl->prim;
  double power_reward = -fabs(s->power_level - 100) / 100;
  double temp_penalty = 0, pressure_penalty = 0;
  if (s->fuel_temperature > 800) temp_penalty = -(s->fuel_temperature - 800) / 100;
  if (s->coolant_pressure > 16) pressure_penalty = -(s->coolant_pressure - 16);
  double scram_penalty = s->scram_status ? -100 : 0;
  double base_reward = power_reward + temp_penalty + pressure_penalty + scram_penalty;
  if (!r) return base_reward;
  double efficiency_reward = (r->thermal_efficiency - 0.30) * 10;
  double target_electrical_power = pl->sec.load_demand / 100.0 * 1100.0; /* sim.load_demand == power_level, sim.py:161 */
  double electrical_reward = -fabs(r->electrical_power_mw - target_electrical_power) / 100;
  double steam_pressure_penalty = 0;
  if (r->sg_avg_pressure < 5.0 || r->sg_avg_pressure > 8.0) steam_pressure_penalty = -fabs(r->sg_avg_pressure - 6.895) * 5;
  double condenser_penalty = 0;
  if (r->condenser_pressure > 0.01) condenser_penalty = -(r->condenser_pressure - 0.007) * 100;
  double secondary_reward = efficiency_reward + electrical_reward + steam_pressure_penalty + condenser_penalty;
  return base_reward + secondary_reward * 0.5;
}

NPO_FN double npo_finite_or(double x, double dflt) { return isfinite(x) ? x : dflt; }

/* NuclearPlantSimulator.step  sim.py:130-258 */
NPO_FN void npo_step(npo_plant_t *pl, const npb_params_t *P, const npo_inputs_t *in, npo_outputs_t *out) {
  npb_prim_t *s = &pl->prim;
  /* heat_source.set_power_setpoint  constant_heat_source.py:93-102 (called by the driver loop before step) */
  if (P->heat_source != NPB_HEAT_EXTERNAL && !isnan(in->power_setpoint)) s->hs_setpoint_percent = npo_clip(in->power_setpoint, 0.0, 150.0);
  if (!isnan(in->cooling_water_temp)) pl->sec.cooling_water_temperature = in->cooling_water_temp; /* sim.py:138-139 */

  int nan_reset = 0;
  for (int k = 0; k < NPB_INFO_NRHO; k++) out->rho[k] = NAN;
  int scram_fired = npo_primary_update(s, P, in, &nan_reset, out->rho);

  if (P->mode == NPB_MODE_PRIMARY) {   /* enable_secondary=False: sim.py:155 skips coupling, secondary update and feedback */
    s->sim_time += P->dt;
    npo_observation(pl, P->mode, out->obs);
    out->reward = npo_reward(pl, 0);
    out->done = (uint8_t)scram_fired;
    out->trip_flags = (s->scram_status ? NPB_TRIP_SCRAM : 0) | (scram_fired ? NPB_TRIP_SCRAM_FIRED : 0) | (nan_reset ? NPB_TRIP_NAN_RESET : 0);
    for (int k = 0; k < NPB_INFO_DIM; k++) out->info[k] = NAN;   /* the secondary keys are absent from the reference's dict */
    out->info[NPB_INFO_THERMAL_POWER] = s->thermal_power_mw;
    out->info[NPB_INFO_REACTIVITY_PCM] = s->total_reactivity_pcm;
    out->info[NPB_INFO_TIME] = s->sim_time;
    return;
  }

  npo_coupling_t c;
  npo_primary_to_secondary(s, &c);
  pl->sec.load_demand = s->power_level; /* sim.py:161: the caller's load_demand is overwritten */

  npo_secondary_result_t r;
  npo_secondary_update(pl, P, &c, &r);

  /* _apply_secondary_to_primary_feedback  sim.py:429-498 */
  double heat_removal_factor = r.total_steam_flow / 1665.0;
  if (!r.feedwater_system_available) heat_removal_factor *= 0.5;
  s->steam_flow_rate = r.total_steam_flow;
  s->last_heat_removal_factor = heat_removal_factor;
  s->has_heat_removal_factor = 1;

  s->sim_time += P->dt; /* sim.py:189-193 (state management disabled) */

  npo_observation(pl, P->mode, out->obs);
  out->reward = npo_reward(pl, &r);
  out->done = (uint8_t)scram_fired;
  uint32_t flags = r.trip_flags;
  if (s->scram_status) flags |= NPB_TRIP_SCRAM;
  if (scram_fired) flags |= NPB_TRIP_SCRAM_FIRED;
  if (nan_reset) flags |= NPB_TRIP_NAN_RESET;
  out->trip_flags = flags;
  /* info  sim.py:199-250 with the non-finite substitutions of :231-240 */
  out->info[NPB_INFO_THERMAL_POWER] = s->thermal_power_mw;
  out->info[NPB_INFO_REACTIVITY_PCM] = s->total_reactivity_pcm;
  out->info[NPB_INFO_ELECTRICAL_POWER] = npo_finite_or(r.electrical_power_mw, 0.0);
  out->info[NPB_INFO_THERMAL_EFFICIENCY] = npo_pymax(0.0, npo_pymin(npo_finite_or(r.thermal_efficiency, 0.0), 0.35));
  out->info[NPB_INFO_STEAM_FLOW] = npo_finite_or(r.total_steam_flow, 1665.0);
  out->info[NPB_INFO_STEAM_PRESSURE] = npo_finite_or(r.sg_avg_pressure, 6.895);
  out->info[NPB_INFO_CONDENSER_PRESSURE] = npo_finite_or(r.condenser_pressure, 0.007);
  out->info[NPB_INFO_CONDENSER_HEAT_REJECTION] = npo_finite_or(r.total_system_heat_rejection, 0.0);
  out->info[NPB_INFO_TIME] = s->sim_time;
  out->info[NPB_INFO_FEEDWATER_FLOW] = r.feedwater_total_flow;
  out->info[NPB_INFO_SG_HEAT_TRANSFER] = r.sg_total_heat_transfer; out->info[NPB_INFO_TURBINE_POWER] = r.turbine_power_output;
  out->info[NPB_INFO_FEEDWATER_POWER] = r.feedwater_total_power; out->info[NPB_INFO_PRIMARY_THERMAL_POWER] = r.primary_thermal_power;
  out->info[NPB_INFO_TURBINE_EFFICIENCY] = r.turbine_efficiency;
  out->info[NPB_INFO_TURBINE_HP_POWER] = r.turbine_hp_power; out->info[NPB_INFO_TURBINE_LP_POWER] = r.turbine_lp_power;

  /* maintenance_system.update + state_manager.collect_states  sim.py:208-223; nothing they touch
   * feeds the observation, reward or info built above */
  if (P->maint_enabled) {
    npb_maint_table_t table = npo_maint_table;   /* as npb_step: with the default table the two oil_level params set their row */
    if (!npo_maint_table_custom) {
      table.threshold[NPB_MP_OIL_LEVEL] = P->maint_oil_level_threshold;
      table.cooldown_hours[NPB_MP_OIL_LEVEL] = P->maint_oil_level_cooldown_hours;
    }
    npo_maintenance_update(pl, P, &table);
  }
}

#endif
