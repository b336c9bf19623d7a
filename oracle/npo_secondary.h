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
#include "npo_condenser.h"
#include "npo_ph.h"

typedef struct npo_secondary_result_t {
  double electrical_power_mw, thermal_efficiency, total_steam_flow, sg_avg_pressure;
  double condenser_pressure, total_system_heat_rejection;
  double feedwater_total_flow, feedwater_total_power;
  double sg_total_heat_transfer, turbine_power_output, primary_thermal_power;
  double turbine_efficiency, turbine_hp_power, turbine_lp_power;   /* secondary/__init__.py:955-958 */
  int feedwater_system_available;
  uint32_t trip_flags;
} npo_secondary_result_t;

/* what get_observation() reads from feedwater_state (sim.py:323-329) */
NPO_FN void npo_feedwater_obs(const npo_plant_t *pl, int mode, double *flow, double *power, int *avail) {
  if (mode == NPB_MODE_PRIMARY_SG) { *flow = pl->sec.total_feedwater_flow; *power = 0.0; *avail = 1; return; }
  *flow = pl->fw.total_flow_rate;
  *power = pl->fw.total_power_consumption;
  *avail = pl->fw.system_availability;
}

/* SecondaryReactorPhysics._saturation_temperature  secondary/__init__.py:1455 ff */
NPO_FN double npo_sec_tsat(double pressure_mpa);

NPO_FN void npo_secondary_chemistry(npo_plant_t *pl, const npb_params_t *P) { npo_chemistry_sidecar(&pl->chem[0], &pl->ph, P->dt); }

/* turbine -> condenser -> gates */
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
    for (int i = 0; i < NPB_NUM_SG; i++) {
      sec->prev_sg_levels[i] = pl->sg[i].water_level; sec->prev_sg_steam_flows[i] = sgr.sg_steam_flow[i];
      sec->prev_sg_qualities[i] = pl->sg[i].steam_quality;
    }
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
    r->sg_total_heat_transfer = sgr.total_thermal_power;
    for (int i = 0; i < NPB_NUM_SG; i++) r->primary_thermal_power += c->thermal_power[i];
    return;
  }
  /* ---- STEP 1: feedwater system first, fed with the PREVIOUS step's SG conditions (:442-491) */
  double estimated_steam_flow_per_sg = 555.0 * load_demand_fraction;
  double prev_levels[NPB_NUM_SG], prev_flows[NPB_NUM_SG], prev_quals[NPB_NUM_SG];
  for (int i = 0; i < NPB_NUM_SG; i++) {
    if (sec->has_previous_sg_conditions) { /* the stored copy (:530-535), not the SG objects */
      prev_levels[i] = sec->prev_sg_levels[i]; prev_flows[i] = sec->prev_sg_steam_flows[i]; prev_quals[i] = sec->prev_sg_qualities[i];
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
  for (int i = 0; i < NPB_NUM_SG; i++) {
    sec->prev_sg_levels[i] = pl->sg[i].water_level; sec->prev_sg_steam_flows[i] = sgr.sg_steam_flow[i];
    sec->prev_sg_qualities[i] = pl->sg[i].steam_quality;
  }
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
  npo_turbine_update(&pl->turb, &pl->tstg, sgr->avg_pressure, sgr->avg_temperature, sgr->total_steam_flow, sg_pressures,
                     sec->sg_system_availability, sec->load_demand, 0.007, P->dt / 60.0, &tr);
  /* ---- condenser with the ACTUAL LP exhaust quality from LP-6's outlet enthalpy (:591-621) */
  double lp_exhaust_quality = 0.90;
  {
    double h_f = npo_cond_hf(tr.condenser_pressure), h_g = npo_cond_hg(tr.condenser_pressure);
    double h_fg = h_g - h_f;
    if (h_fg > 0) {
      lp_exhaust_quality = (tr.lp6_outlet_enthalpy - h_f) / h_fg;
      lp_exhaust_quality = npo_pymax(0.0, npo_pymin(1.0, lp_exhaust_quality));
    }
  }
  npo_condenser_result_t cr;
  npo_condenser_update(&pl->cond, &pl->chem[1], tr.condenser_pressure, tr.effective_steam_flow, lp_exhaust_quality,
                       45000.0, sec->cooling_water_temperature, 1.2, 185.0, P->dt / 60.0, &cr);
  sec->total_feedwater_flow = fwr->total_flow_rate;
  sec->operating_hours += P->dt / 3600.0;

  /* ---- chemistry sidecar (:634-665): shared WaterChemistry + pH controller */
  npo_secondary_chemistry(pl, P);

  sec->total_steam_flow = sgr->total_steam_flow;
  sec->total_heat_transfer = sgr->total_thermal_power;
  /* ---- energy accounting and electrical-power gates (:750-932) */
  double primary_thermal_power = 0.0;
  for (int i = 0; i < NPB_NUM_SG; i++) primary_thermal_power += c->thermal_power[i];
  double thermal_power_mw = primary_thermal_power;
  double turbine_electrical_power = tr.electrical_power_net;
  double total_system_heat_rejection_mw = primary_thermal_power - turbine_electrical_power;
  double actual_feedwater_flow = fwr->total_flow_rate;
  double power_reduction_factor = 1.0;
  if (actual_feedwater_flow < 300.0) power_reduction_factor = 0.0;
  if (power_reduction_factor > 0.0) {
    if (sgr->total_steam_flow < (300.0 * 0.5)) power_reduction_factor *= 0.1;
    if (sgr->avg_pressure < (1.0 * 0.5)) power_reduction_factor *= 0.1;
    if (thermal_power_mw > (primary_thermal_power * 1.1)) power_reduction_factor = 0.0;
  }
  sec->electrical_power_output = turbine_electrical_power * power_reduction_factor;
  if (primary_thermal_power > 0) sec->thermal_efficiency = sec->electrical_power_output / primary_thermal_power;
  else sec->thermal_efficiency = 0.0;

  r->electrical_power_mw = sec->electrical_power_output;
  r->thermal_efficiency = sec->thermal_efficiency;
  r->total_steam_flow = sgr->total_steam_flow;
  r->sg_avg_pressure = sgr->avg_pressure;
  r->condenser_pressure = cr.condenser_pressure;
  r->total_system_heat_rejection = total_system_heat_rejection_mw * 1e6;
  r->feedwater_total_flow = fwr->total_flow_rate;
  r->feedwater_total_power = fwr->total_power_consumption;
  r->feedwater_system_available = fwr->system_availability;
  r->sg_total_heat_transfer = sgr->total_thermal_power; r->turbine_power_output = tr.electrical_power_gross;
  r->primary_thermal_power = primary_thermal_power;
  r->turbine_efficiency = tr.overall_efficiency; r->turbine_hp_power = tr.hp_power; r->turbine_lp_power = tr.lp_power;
  r->trip_flags = (fwr->pump_trip_mask << 8) | (pl->fw.system_trip_active ? NPB_TRIP_FW_SYSTEM : 0) |
                  (tr.trip_active ? NPB_TRIP_TURBINE : 0);
}

#endif
