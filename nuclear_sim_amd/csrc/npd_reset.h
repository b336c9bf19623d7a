/*
 * npd_reset.h -- device physics: NuclearPlantSimulator.reset(start_at_steady_state)  simulator/core/sim.py:546-581.
 *
 * The reference's reset is NOT "back to the constructed state": every subsystem has its own reset() that puts
 * some attributes back to literals (not to the configured initial conditions), re-applies the initial conditions
 * for others, and leaves a third group untouched (lubrication systems, the turbine's metal-temperature tracker,
 * the pH controller, tube-interior scale, _previous_sg_conditions ...), so the state after reset() depends on the
 * history before it.  With start_at_steady_state=True it then advances the three steam generators by one
 * update (dt = 1 s) as a side effect of computing an "equilibrium point" and force-sets every feedwater pump.
 * Each function below takes the section as it is (history) and leaves it as the reference's reset leaves it.
 * Pinned by tests/golden/r1_*.npz (reference run: construct, step, reset, step).
 */
#ifndef NPD_RESET_H
#define NPD_RESET_H
#include "npd_init.h"

/* PrimaryReactorPhysics.reset_system  primary/__init__.py:409-426: heat_source.reset()
 * (constant_heat_source.py:185-194 / reactor_heat_source.py:139-142; the noise generator is NOT re-seeded), a fresh
 * ReactorState(), thermal_power_mw = total_reactivity_pcm = 0; sim.time = 0 (sim.py:569).  The simulator's
 * _last_heat_removal_factor attribute (sim.py:391-399,495) survives a reset. */
NPD_FN void npd_prim_reset(npb_prim_t *s) {
  const double factor = s->last_heat_removal_factor;
  const int has_factor = s->has_heat_removal_factor;
  npd_prim_init(s);
  s->last_heat_removal_factor = factor; s->has_heat_removal_factor = has_factor;
}

/* SteamGenerator.reset  steam_generator.py:1328-1346 (literals, not the configured initial conditions) with
 * TSPFoulingModel.reset tsp_fouling_model.py:758-778 / FoulingModelBase.reset fouling_model_base.py:247-256.
 * The tube-interior fouling model is not reset: scale thickness, composition, resistance and its operating
 * years carry over. */
NPD_FN void npd_sg_reset(npb_sg_t *g) {
  g->secondary_pressure = 6.895; g->secondary_temperature = 285.8; g->steam_quality = 0.99; g->water_level = 12.5;
  g->steam_flow_rate = 555.0; g->tube_wall_temp = 310.0; g->heat_transfer_rate = 1085.0e6;
  for (int k = 0; k < NPB_NUM_TSP; k++) { g->tsp_magnetite[k] = 0.0; g->tsp_copper[k] = 0.0; g->tsp_silica[k] = 0.0; g->tsp_biological[k] = 0.0; }
  g->tsp_fouling_fraction = 0.0; g->tsp_ht_degradation = 0.0; g->tsp_operating_years = 0.0;
  g->tsp_pressure_drop_ratio = 1.0; g->tsp_shutdown_required = 0;
}

/* FeedwaterPump.reset pump_system.py:1010-1058, FeedwaterPumpSystem.reset :1413-1424 (_initialize_pumps), then
 * EnhancedFeedwaterPhysics.reset physics.py:1286-1323, which re-applies the configured initial conditions
 * (_apply_initial_conditions :185-437 -> _initialize_pumps(sg_conditions)): the pump comes out as constructed,
 * except that the lubrication system is never reset -- oil viscosity change, the three additive levels, the
 * lubrication effectiveness and the impeller / coupling wear carry over, and the performance factors are
 * recomputed with that effectiveness (pump_lubrication.py:1412-1478) -- and that the spare's speed setpoint is the
 * 0 of pump.reset() rather than the dataclass default.  (Default FeedwaterInitialConditions; a caller with other
 * initial conditions re-applies its columns afterwards, as at construction.) */
NPD_FN void npd_pump_reset(npb_pump_t *p, int index) {
  const double viscosity = p->oil_viscosity_change, antioxidant = p->antioxidant_level, anti_wear = p->anti_wear_level;
  const double inhibitor = p->corrosion_inhibitor_level, effectiveness = p->lubrication_effectiveness;
  const double impeller = p->wear_impeller, coupling = p->wear_coupling_system;
  npd_pump_init(p, index);
  p->oil_viscosity_change = viscosity; p->antioxidant_level = antioxidant; p->anti_wear_level = anti_wear;
  p->corrosion_inhibitor_level = inhibitor; p->lubrication_effectiveness = effectiveness;
  p->wear_impeller = impeller; p->wear_coupling_system = coupling;
  npd_pump_performance_factors(p, 0.0);
  if (index >= 3) p->speed_setpoint = 0.0;
}

/* _initialize_feedwater_system_to_steady_state  secondary/__init__.py:1247-1357: every pump gets "perfect"
 * hydraulic / mechanical conditions; the first pumps_needed (3 at any power: min(4, max(3, ceil(flow / 555))))
 * are put straight to RUNNING at the equilibrium speed and flow.  _calculate_power_consumption runs while the
 * status is still STOPPED, so the power comes out 0; set_flow_demand overwrites the speed setpoint. */
NPD_FN void npd_pump_steady_state(npb_pump_t *p, int index, double steam_pressure, double feedwater_flow, int pumps_needed,
                                  double pump_speed) {
  p->status = NPD_PUMP_STOPPED; p->available = 1; p->trip_active = 0; p->trip_reason = 0;
  p->suction_pressure = 0.5; p->discharge_pressure = steam_pressure + 0.5; p->npsh_available = 25.0;
  p->differential_pressure = p->discharge_pressure - p->suction_pressure;
  p->motor_temperature = 65.0; p->vibration_level = 1.5;
  p->cavitation_intensity = 0.0; p->cavitation_damage = 0.0; p->cavitation_time = 0.0;
  if (index < pumps_needed) {
    p->speed_setpoint = pump_speed;
    npd_pump_set_flow_demand(p, feedwater_flow / pumps_needed);
    p->speed_percent = pump_speed; p->flow_rate = feedwater_flow / pumps_needed;
    p->power_consumption = 0.0;
    p->status = NPD_PUMP_RUNNING;
  } else {
    p->speed_setpoint = 0.0;
    npd_pump_set_flow_demand(p, 0.0);
    p->speed_percent = 0.0; p->flow_rate = 0.0; p->power_consumption = 0.0;
  }
}

/* EnhancedFeedwaterPhysics.reset physics.py:1286-1323 (level control level_control.py:107-110,502-514, diagnostics
 * performance_monitoring.py:663-673, protection protection_system.py:790-803): as constructed, except that
 * _apply_initial_conditions sums the sg_steam_flows that reset() has just set to 555 kg/s each (:1307 then :196-197) */
NPD_FN void npd_fw_reset(npb_fw_t *fw) {
  npd_fw_init(fw);
  fw->total_flow_rate = 0 + 555.0 + 555.0 + 555.0;
}

/* EnhancedTurbinePhysics.reset  turbine/enhanced_physics.py:1269-1283: stage system (stage_system.py:1034-1050,
 * 395-415), rotor dynamics (rotor_dynamics.py:1095-1120, bearings :566-583) and protection (:473-479) go to
 * literals; the bearing lubrication system and the metal-temperature tracker are not reset.  steady: secondary/
 * __init__.py:1359-1367 (load demand as a fraction, power = the equilibrium's electrical power). */
NPD_FN void npd_turb_reset(npb_turb_t *t, npb_tstg_t *g, int steady, double load_demand_percent, double electrical_power) {
  for (int k = 0; k < NPB_NUM_STAGES; k++) { g->stage_efficiency_degradation[k] = 0.0; g->stage_deposit_thickness[k] = 0.0; g->stage_blade_wear_factor[k] = 1.0; }
  t->rotor_speed = 0.0; t->rotor_temperature = 450.0; t->thermal_expansion = 0.0; t->thermal_bow = 0.0;
  for (int i = 0; i < NPB_NUM_BEARINGS; i++) { t->bearing_load[i] = 0.0; t->bearing_metal_temp[i] = 90.0; t->bearing_wear_factor[i] = 1.0; }
  t->vibration_displacement = 0.0;
  t->trip_active = 0; t->trip_latched_mask = 0;
  t->timer_overspeed = 0.0; t->timer_vibration = 0.0; t->timer_bearing_temp = 0.0;
  t->total_power_output = 0.0; t->load_demand = 1.0;
  if (steady) { t->load_demand = load_demand_percent / 100.0; t->total_power_output = electrical_power; }
}

/* WaterChemistry.reset  water_chemistry.py:571-609: design values and the composite indices recomputed from them.
 * For the condenser's instance that is NOT its constructed state (the condenser's initial conditions -- pH 7.5,
 * chlorine 1.0, oxygen 8.0 -- are not re-applied, condenser/physics.py:1556-1595) */
NPD_FN void npd_chem_reset(npb_chem_t *c) { npd_chem_init(c, 0); }

/* EnhancedCondenserPhysics.reset  condenser/physics.py:1556-1595 with VacuumSystem.reset vacuum_system.py:567-588
 * and SteamJetEjector.reset vacuum_pump.py:551-561: literals equal to the constructed state except the cooling-water
 * outlet temperature (35.0, the constructor computes it); the vacuum system's efficiency is not touched */
NPD_FN void npd_cond_reset(npb_cond_t *cd) {
  const double efficiency = cd->vacuum_system_efficiency;
  npd_cond_init(cd);
  cd->cooling_water_outlet_temp = 35.0;
  cd->vacuum_system_efficiency = efficiency;
}

/* SecondaryReactorPhysics.reset_system  secondary/__init__.py:1041-1072 and EnhancedSteamGeneratorPhysics.reset
 * steam_generator/enhanced_physics.py:1193-1214: system-level outputs to literals; _previous_sg_conditions and
 * _previous_feedwater_temp are attributes the reset never touches */
NPD_FN void npd_sec_reset(npb_sec_t *sec) {
  sec->total_steam_flow = 0.0; sec->total_heat_transfer = 0.0; sec->electrical_power_output = 0.0; sec->thermal_efficiency = 0.0;
  sec->total_feedwater_flow = 0.0; sec->load_demand = 100.0; sec->cooling_water_temperature = 25.0; sec->operating_hours = 0.0;
  sec->sg_avg_pressure = 6.9; sec->sg_avg_temperature = 285.8; sec->sg_avg_quality = 0.99; /* config.design_steam_pressure / _temperature  steam_generator/config.py:197-198 */
  sec->sg_system_availability = 1;
}

/* initialize_to_steady_state  secondary/__init__.py:1074-1101 with _calculate_equilibrium_point :1103-1238 for the
 * thermal power sim.reset() passes: primary_physics.thermal_power_mw was just zeroed, so it is always the rated
 * power (sim.py:558-563).  The "equilibrium" advances the steam generators by one update of dt = 1 s with the
 * perfect-mass-balance feedwater fallback (:1181-1187); the results feed the pump initialisation. */
typedef struct npd_equilibrium_t {
  double load_demand, steam_flow, steam_pressure, feedwater_flow, electrical_power, thermal_efficiency, heat_transfer, pump_speed;
  int pumps_needed;
} npd_equilibrium_t;

NPD_FN void npd_steady_state_equilibrium(npb_sg_t *sg, npb_sec_t *sec, const npb_params_t *P, double thermal_power_mw, npd_equilibrium_t *eq) {
  const double load_demand = npd_pymin(100.0, (thermal_power_mw / 3000.0) * 100.0);
  const double thermal_power_per_sg = thermal_power_mw / NPB_NUM_SG;
  const double primary_flow_per_sg = 5700.0 * (load_demand / 100.0);
  /* _calculate_temperatures_from_power :1378-1451 with power_fraction = load_demand / 100 */
  npd_coupling_t c;
  for (int i = 0; i < NPB_NUM_SG; i++) {
    const double power_fraction = load_demand / 100.0;
    double delta_t = (primary_flow_per_sg > 0) ? (thermal_power_per_sg * 1000.0) / (primary_flow_per_sg * 5.2) : 0.0;
    const double cold_leg_temp = 293.0;
    const double hot_leg_temp = cold_leg_temp + (34.0 * power_fraction);
    const double realistic_delta_t = hot_leg_temp - cold_leg_temp;
    if (fabs(delta_t - realistic_delta_t) > 10.0) {
      if (thermal_power_per_sg > 0) delta_t = realistic_delta_t * npd_pymin(1.0, thermal_power_per_sg / 1000.0);
      else delta_t = 0.0;
    }
    double outlet_temp = cold_leg_temp;
    double inlet_temp = outlet_temp + delta_t;
    inlet_temp = npd_clip(inlet_temp, 293.0, 350.0);
    outlet_temp = npd_clip(outlet_temp, 280.0, 300.0);
    if (inlet_temp <= outlet_temp) inlet_temp = outlet_temp + 5.0;
    c.inlet_temp[i] = inlet_temp; c.outlet_temp[i] = outlet_temp; c.flow[i] = primary_flow_per_sg;
    c.thermal_power[i] = thermal_power_per_sg;
  }
  npd_sgsys_result_t r;
  npd_sgsys_update(sg, sec, P, &c, load_demand / 100.0, 0, 227.0, 1.0, &r);
  eq->load_demand = load_demand;
  eq->steam_flow = r.total_steam_flow; eq->steam_pressure = r.avg_pressure;
  eq->feedwater_flow = r.total_steam_flow;
  if (load_demand >= 100.0) eq->thermal_efficiency = 0.34;
  else if (load_demand >= 75.0) eq->thermal_efficiency = 0.32 + 0.02 * (load_demand - 75.0) / 25.0;
  else if (load_demand >= 50.0) eq->thermal_efficiency = 0.28 + 0.04 * (load_demand - 50.0) / 25.0;
  else eq->thermal_efficiency = 0.20 + 0.08 * (load_demand / 50.0);
  eq->electrical_power = thermal_power_mw * eq->thermal_efficiency;
  eq->heat_transfer = thermal_power_mw * 1e6;
  int needed = (int)ceil(eq->feedwater_flow / 555.0);
  needed = needed < 3 ? 3 : needed; needed = needed > 4 ? 4 : needed;
  eq->pumps_needed = needed;
  eq->pump_speed = npd_pymin(100.0, ((eq->feedwater_flow / needed) / 555.0) * 100.0);
}

/* system-level variables after initialize_to_steady_state  secondary/__init__.py:1095-1101 */
NPD_FN void npd_sec_steady_state(npb_sec_t *sec, const npd_equilibrium_t *eq) {
  sec->total_steam_flow = eq->steam_flow; sec->total_heat_transfer = eq->heat_transfer;
  sec->electrical_power_output = eq->electrical_power; sec->thermal_efficiency = eq->thermal_efficiency;
  sec->total_feedwater_flow = eq->feedwater_flow; sec->load_demand = eq->load_demand;
}

#endif
