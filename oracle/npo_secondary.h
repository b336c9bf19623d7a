/*
 * npo_secondary.h -- CPU oracle: SecondaryReactorPhysics.update_system orchestration.
 * TEST INFRASTRUCTURE ONLY (see npo_common.h).
 *
 * Follows systems/secondary/__init__.py:340-1021: step ordering FW -> SG -> turbine ->
 * condenser -> chemistry, the dt fan-out (:358-370), feedwater-temperature smoothing
 * (:385-398), load fraction clamp (:420-440) and the electrical-power gates (:804-932).
 */
#ifndef NPO_SECONDARY_H
#define NPO_SECONDARY_H
#include "npo_common.h"
#include "npo_plant.h"
#include "npo_primary.h"
#include "npo_sg.h"
#include "npo_feedwater.h"
#include "npo_turbine.h"

typedef struct npo_secondary_result_t {
  double electrical_power_mw, thermal_efficiency, total_steam_flow, sg_avg_pressure;
  double condenser_pressure, total_system_heat_rejection;
  double feedwater_total_flow, feedwater_total_power;
  int feedwater_system_available;
  uint32_t trip_flags;
} npo_secondary_result_t;

/* what get_observation() reads from feedwater_state (sim.py:323-329) */
NPO_FN void npo_feedwater_obs(const npo_plant_t *pl, double *flow, double *power, int *avail) {
  *flow = pl->fw.total_flow_rate;
  *power = pl->fw.total_power_consumption;
  *avail = pl->fw.system_availability;
}

/* SecondaryReactorPhysics._saturation_temperature  secondary/__init__.py:1455 ff */
NPO_FN double npo_sec_tsat(double pressure_mpa);

/* turbine -> condenser -> gates; defined after the subsystem headers exist */
NPO_FN void npo_secondary_tail(npo_plant_t *pl, const npb_params_t *P, const npo_coupling_t *c,
                               const npo_sgsys_result_t *sgr, const npo_fw_result_t *fwr, npo_secondary_result_t *r);

NPO_FN void npo_secondary_update(npo_plant_t *pl, const npb_params_t *P, const npo_coupling_t *c,
                                 npo_secondary_result_t *r) {
  npb_sec_t *sec = &pl->sec;
  const double dt = P->dt;
  /* feedwater temperature smoothing :385-398 (estimate 40 + 187, alpha 0.1) */
  double estimated_feedwater_temp = 40.0 + 187.0;
  double alpha = 0.1;
  double actual_feedwater_temp = (alpha * estimated_feedwater_temp + (1 - alpha) * sec->previous_feedwater_temp);
  sec->previous_feedwater_temp = actual_feedwater_temp;
  /* load fraction :420-440 -- sg_X_thermal_power is present, so the direct sum takes precedence */
  double total_thermal_power_mw = 0.0;
  for (int i = 0; i < NPB_NUM_SG; i++) total_thermal_power_mw += c->thermal_power[i];
  double load_demand_fraction = npo_pymin(1.0, total_thermal_power_mw / 3000.0);
  load_demand_fraction = npo_pymax(load_demand_fraction, 0.2);

  memset(r, 0, sizeof(*r));
  if (P->mode == NPB_MODE_PRIMARY_SG) {
    /* BASELINE config 2: primary + steam generators only; feedwater = steam demand
     * ("perfect mass balance" fallback, enhanced_physics.py:495-497) */
    npo_sgsys_result_t sgr;
    npo_sgsys_update(pl->sg, sec, P, c, load_demand_fraction, 0, actual_feedwater_temp, dt * 60, &sgr);
    sec->has_previous_sg_conditions = 1;
    sec->total_steam_flow = sgr.total_steam_flow;
    sec->total_heat_transfer = sgr.total_thermal_power;
    sec->total_feedwater_flow = sgr.total_steam_flow;
    sec->electrical_power_output = 0.0;
    sec->thermal_efficiency = 0.0;
    sec->operating_hours += dt / 3600.0;
    r->total_steam_flow = sgr.total_steam_flow;
    r->sg_avg_pressure = sgr.avg_pressure;
    r->condenser_pressure = 0.007;
    r->feedwater_total_flow = sgr.total_steam_flow;
    r->feedwater_system_available = 1;
    return;
  }
  /* ---- STEP 1: feedwater system first, fed with the PREVIOUS step's SG conditions (:442-491) */
  double estimated_steam_flow_per_sg = 555.0 * load_demand_fraction;
  double prev_levels[NPB_NUM_SG], prev_flows[NPB_NUM_SG], prev_quals[NPB_NUM_SG];
  for (int i = 0; i < NPB_NUM_SG; i++) {
    if (sec->has_previous_sg_conditions) {
      prev_levels[i] = pl->sg[i].water_level; prev_flows[i] = pl->sg[i].steam_flow_rate; prev_quals[i] = pl->sg[i].steam_quality;
    } else { /* :447-453 hard-coded first-step values, not the SG initial conditions */
      prev_levels[i] = 12.5; prev_flows[i] = estimated_steam_flow_per_sg; prev_quals[i] = 0.99;
    }
  }
  npo_fw_result_t fwr;
  npo_feedwater_update(pl->pump, &pl->fw, prev_levels, prev_flows, prev_quals, /*condensate temp*/ 40.0,
                       /*suction*/ 0.5, /*discharge*/ 7.4, dt, &fwr);
  /* ---- STEP 2: steam generators with ACTUAL feedwater flows (:493-535). The pump system writes
   * 'sg_N_flow' keys but the consumer looks up 'sg_N', so the split is always equal (:500-506). */
  double fw_flows[NPB_NUM_SG];
  for (int i = 0; i < NPB_NUM_SG; i++) fw_flows[i] = fwr.total_flow_rate / NPB_NUM_SG;
  npo_sgsys_result_t sgr;
  npo_sgsys_update(pl->sg, sec, P, c, load_demand_fraction, fw_flows, actual_feedwater_temp, dt * 60, &sgr);
  sec->has_previous_sg_conditions = 1;
  double avg_steam_pressure = sgr.avg_pressure;
  double total_steam_flow = sgr.total_steam_flow;

  npo_secondary_tail(pl, P, c, &sgr, &fwr, r);
  (void)avg_steam_pressure; (void)total_steam_flow;
}


NPO_FN void npo_secondary_tail(npo_plant_t *pl, const npb_params_t *P, const npo_coupling_t *c,
                               const npo_sgsys_result_t *sgr, const npo_fw_result_t *fwr, npo_secondary_result_t *r) {
  npb_sec_t *sec = &pl->sec;
  (void)c;
  /* ---- STEP 5: turbine (dt in hours; load_demand handed over in PERCENT, :564-569) */
  double sg_pressures[NPB_NUM_SG];
  for (int i = 0; i < NPB_NUM_SG; i++) sg_pressures[i] = pl->sg[i].secondary_pressure;
  npo_turbine_result_t tr;
  npo_turbine_update(&pl->turb, sgr->avg_pressure, sgr->avg_temperature, sgr->total_steam_flow, sg_pressures,
                     sec->sg_system_availability, sec->load_demand, 0.007, P->dt / 60.0, &tr);
  /* TODO condenser */
  sec->total_steam_flow = sgr->total_steam_flow;
  sec->total_heat_transfer = sgr->total_thermal_power;
  sec->total_feedwater_flow = fwr->total_flow_rate;
  sec->operating_hours += P->dt / 3600.0;
  r->total_steam_flow = sgr->total_steam_flow;
  r->sg_avg_pressure = sgr->avg_pressure;
  r->condenser_pressure = 0.007;
  r->feedwater_total_flow = fwr->total_flow_rate;
  r->feedwater_total_power = fwr->total_power_consumption;
  r->feedwater_system_available = fwr->system_availability;
  r->trip_flags = (fwr->pump_trip_mask << 8) | (pl->fw.system_trip_active ? NPB_TRIP_FW_SYSTEM : 0);
}

#endif
