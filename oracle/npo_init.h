/*
 * npo_init.h -- CPU oracle: plant state as the reference's constructors leave it
 * (NuclearPlantSimulator.__init__ with the default SecondarySystemConfig; the
 * data-gen runner starts episodes from this state without reset(),
 * maintenance_scenario_runner.py:243-244).  TEST INFRASTRUCTURE ONLY.
 */
#ifndef NPO_INIT_H
#define NPO_INIT_H
#include "npo_plant.h"
#include "npo_sg.h"
#include "npo_feedwater.h"
#include "npo_turbine.h"
#include "npo_condenser.h"
#include "npo_maintenance.h"

/* ReactorState defaults  systems/primary/__init__.py:48-106 */
NPO_FN void npo_prim_init(npb_prim_t *s) {
  memset(s, 0, sizeof(*s));
  s->neutron_flux = 1e13; s->reactivity = 0.0;
  const double prec[6] = {0.0002, 0.0011, 0.0010, 0.0030, 0.0096, 0.0003};
  for (int i = 0; i < 6; i++) s->precursors[i] = prec[i];
  s->fuel_temperature = 410.0; s->coolant_temperature = 310.0; s->coolant_pressure = 15.5;
  s->coolant_flow_rate = 20000.0; s->coolant_void_fraction = 0.0;
  s->steam_temperature = 285.0; s->steam_pressure = 7.0; s->steam_flow_rate = 1000.0;
  s->feedwater_flow_rate = 1000.0;
  s->control_rod_position = 95.0; s->steam_valve_position = 50.0; s->boron_concentration = 12000.0;
  s->xenon_concentration = 2.8e15; s->iodine_concentration = 1.5e16; s->samarium_concentration = 1.0e15;
  s->burnable_poison_worth = 0.0; s->fuel_burnup = 15000.0; s->power_level = 100.0;
  s->thermal_power_mw = 0.0; s->total_reactivity_pcm = 0.0;   /* __init__.py:169-170 */
  s->hs_setpoint_percent = 100.0; s->hs_filtered_noise_mw = 0.0; /* constant_heat_source.py:48,65 */
  s->last_heat_removal_factor = 0.0; s->has_heat_removal_factor = 0;
  s->sim_time = 0.0; s->scram_status = 0;
}

/* SteamGenerator.__init__ steam_generator.py:87-112 then
 * EnhancedSteamGeneratorPhysics._apply_initial_conditions enhanced_physics.py:231-334
 * with SteamGeneratorInitialConditions defaults (config.py:102-122) */
NPO_FN void npo_sg_init(npb_sg_t *g) {
  memset(g, 0, sizeof(*g));
  g->water_level = 12.5; g->secondary_pressure = 6.9; g->secondary_temperature = 285.8;
  g->steam_quality = 0.99; g->steam_flow_rate = 500.0;
  g->tube_wall_temp = 300.0; g->heat_transfer_rate = 1085.0e6;
  g->tsp_fouling_fraction = 0.0; g->tsp_pressure_drop_ratio = 1.0;
  g->tsp_ht_degradation = npo_tsp_ht_degradation(g->tsp_fouling_fraction);
  g->scale_thermal_resistance = npo_scale_thermal_resistance(g);
}

/* SecondaryReactorPhysics.__init__  secondary/__init__.py:300-338 */
NPO_FN void npo_sec_init(npb_sec_t *sec) {
  memset(sec, 0, sizeof(*sec));
  sec->previous_feedwater_temp = 227.0;  /* first use :385-386 seeds it with feedwater_temp = 227 */
  sec->load_demand = 100.0; sec->cooling_water_temperature = 25.0;
  sec->sg_avg_pressure = (0 + 6.9 + 6.9 + 6.9) / 3; sec->sg_avg_temperature = (0 + 285.8 + 285.8 + 285.8) / 3;
  sec->sg_avg_quality = (0 + 0.99 + 0.99 + 0.99) / 3; /* enhanced_physics.py:327-329 */
  sec->sg_system_availability = 1;
  sec->has_previous_sg_conditions = 0;
}

/* BaseLubricationSystem.__init__ lubrication_base.py:150-176 + FeedwaterPumpLubricationSystem.__init__
 * pump_lubrication.py:204-222, then EnhancedFeedwaterPhysics._apply_initial_conditions
 * feedwater/physics.py:185-437 with FeedwaterInitialConditions defaults (feedwater/config.py) and
 * FeedwaterPumpSystem._initialize_pumps pump_system.py:1177-1233 (3 running at 510 kg/s demand, 1 spare) */
NPO_FN void npo_pump_init(npb_pump_t *p, int index) {
  memset(p, 0, sizeof(*p));
  /* lubrication system as constructed */
  p->oil_level = 90.0; p->oil_temperature = 55.0; p->oil_contamination = 5.0; p->oil_moisture = 0.02;
  p->oil_acidity = 0.15; p->oil_viscosity_change = 0.0;
  p->antioxidant_level = 100.0; p->anti_wear_level = 100.0; p->corrosion_inhibitor_level = 100.0;
  npo_pump_lubrication_effectiveness(p); /* computed once at construction, NOT after the ICs below */
  /* initial conditions (pump_oil_contamination 5.0, water 0.05, acid 1.0, levels 100, oil temp 45, seal wear 0.3) */
  p->oil_contamination = 5.0; p->oil_moisture = 0.05; p->oil_acidity = 1.0; p->oil_level = 100.0; p->oil_temperature = 45.0;
  p->wear_mechanical_seals = 0.3;
  p->seal_leakage_rate = (index < 3) ? 0.001 : 0.0;
  npo_pump_performance_factors(p, 0.0);
  p->suction_pressure = 0.5; p->discharge_pressure = 8.0; p->npsh_available = 20.0; p->differential_pressure = 0.0;
  p->cavitation_intensity = 0.05; p->cavitation_damage = 0.1 * 10.0; p->cavitation_time = 0.0;
  p->motor_temperature = 70.0;
  p->available = 1; p->trip_active = 0; p->trip_reason = 0;
  p->power_consumption = 1.0 * NPO_PUMP_RATED_POWER;
  if (index < 3) {
    p->status = NPO_PUMP_RUNNING; p->flow_rate = 500.0; p->vibration_level = 5.0;
    double flow_per_pump = ((500.0 + 500.0 + 500.0) * 1.02 / 3) / 1.0; /* degradation factor 1.0: flow_degradation is 0 */
    npo_pump_set_flow_demand(p, flow_per_pump);
    double required_speed = sqrt(flow_per_pump / (NPO_PUMP_RATED_FLOW * npo_pump_flow_factor(p))) * 100.0;
    double safe_speed = npo_pymin(100.0, npo_pymax(30.0, required_speed));
    p->speed_percent = safe_speed; p->speed_setpoint = safe_speed;
  } else {
    p->status = NPO_PUMP_STOPPED; p->flow_rate = 0.0; p->vibration_level = 0.0;
    p->speed_percent = 0.0; p->speed_setpoint = 100.0; /* BasePumpState default, pump_models.py:39 */
    p->flow_demand = NPO_PUMP_RATED_FLOW;               /* FeedwaterPump.__init__ pump_system.py:131 */
  }
}

NPO_FN void npo_fw_init(npb_fw_t *fw) {
  memset(fw, 0, sizeof(*fw));
  fw->total_flow_rate = 500.0 + 500.0 + 500.0; /* physics.py:207-208 */
  fw->total_power_consumption = 0.0;
  fw->cav_accumulated_damage = (0.1 + 0.1 + 0.1 + 0.1) / 4 * 10.0; /* physics.py:396-398 */
  fw->overall_health_score = 1.0;
  fw->system_availability = 1;
  fw->running_mask = 0; /* FeedwaterPumpSystem.running_pumps starts empty, pump_system.py:1172 */
}

/* EnhancedTurbinePhysics.__init__ + _apply_initial_conditions  turbine/enhanced_physics.py:505-612 with
 * TurbineInitialConditions defaults (turbine/config.py); TurbineStage.__init__ stage_system.py:49-96;
 * BearingModel.__init__ rotor_dynamics.py:55-82; apply_unified_initial_conditions
 * turbine_bearing_lubrication.py:187-228 */
NPO_FN void npo_turb_init(npb_turb_t *t, npb_tstg_t *g) {
  memset(t, 0, sizeof(*t)); memset(g, 0, sizeof(*g));
  for (int k = 0; k < 14; k++) { g->stage_efficiency_degradation[k] = 0.0; g->stage_deposit_thickness[k] = 0.0; g->stage_blade_wear_factor[k] = 1.0; }
  t->rotor_speed = 3600.0; t->rotor_temperature = 450.0; t->thermal_bow = 0.0; t->thermal_expansion = 0.0;
  for (int i = 0; i < 4; i++) { t->bearing_load[i] = 0.0; t->bearing_metal_temp[i] = 80.0; t->bearing_wear_factor[i] = 1.0; }
  for (int i = 0; i < 8; i++) g->rotor_temperatures[i] = 450.0;
  const double casing[6] = {380.0, 360.0, 340.0, 320.0, 300.0, 280.0};
  const double blade[14] = {500.0, 480.0, 460.0, 440.0, 420.0, 400.0, 380.0, 360.0, 340.0, 320.0, 300.0, 280.0, 260.0, 240.0};
  for (int i = 0; i < 6; i++) g->casing_temperatures[i] = casing[i];
  for (int i = 0; i < 14; i++) g->blade_temperatures[i] = blade[i];
  t->load_demand = 1.0; t->total_power_output = 1000.0; t->vibration_displacement = 0.0;
  t->lub_oil_temperature = 45.0; t->lub_oil_contamination = 5.0; t->lub_oil_moisture = 0.02; t->lub_oil_acidity = 0.15;
  t->lub_oil_viscosity_change = 0.0; t->lub_antioxidant_level = 100.0; t->lub_anti_wear_level = 100.0;
  t->lub_corrosion_inhibitor_level = 100.0; t->lub_effectiveness = 1.0;
  const double wear[5] = {2.0, 1.5, 3.0, 2.5, 1.0};
  for (int i = 0; i < 5; i++) t->lub_wear[i] = wear[i];
}

/* WaterChemistry.__init__ water_chemistry.py:222-275; index 1 then receives the condenser's initial
 * conditions (condenser/physics.py:1476-1500: ph 7.5, hardness 150, chlorine 1.0, DO 8.0) AFTER the
 * composite indices were computed from the design values */
NPO_FN void npo_chem_init(npb_chem_t *c, int index) {
  memset(c, 0, sizeof(*c));
  c->ph = 9.2; c->hardness = 150.0; c->total_dissolved_solids = 500.0; c->chloride = 50.0; c->dissolved_oxygen = 0.005;
  c->chlorine_residual = 0.5; c->antiscalant_concentration = 5.0; c->corrosion_inhibitor_level = 10.0;
  c->treatment_efficiency = 0.95;
  npo_chem_composites(c);
  if (index == 1) { c->ph = 7.5; c->hardness = 150.0; c->chlorine_residual = 1.0; c->dissolved_oxygen = 8.0; }
}

/* PHControllerState defaults  ph_control_system.py:131-190 */
NPO_FN void npo_ph_init(npb_ph_t *s) {
  memset(s, 0, sizeof(*s));
  s->measured_ph = 9.2; s->ammonia_tank_level = 80.0; s->morpholine_tank_level = 80.0;
  s->controller_enabled = 1; s->ammonia_supply_available = 1; s->morpholine_supply_available = 1;
}

/* EnhancedCondenserPhysics.__init__ + _apply_initial_conditions condenser/physics.py:486-562,1374-1555 with
 * CondenserInitialConditions defaults (condenser/config.py); VacuumSystem.__init__ vacuum_system.py:252-300 */
NPO_FN void npo_cond_init(npb_cond_t *cd) {
  memset(cd, 0, sizeof(*cd));
  cd->cooling_water_outlet_temp = 35.632642211589584; cd->heat_rejection_rate = 2000000000.0;
  cd->active_tube_count = 84000; cd->plugged_tube_count = 0; cd->average_wall_thickness = 0.00159;
  cd->fouling_distribution_factor = 1.0;
  cd->condenser_pressure = 0.007; cd->air_partial_pressure = 0.0005; cd->current_air_leakage = 0.05;
  cd->air_mass_in_condenser = 0.1; cd->vacuum_system_efficiency = 1.0; cd->rotation_timer = 0.0;
  for (int e = 0; e < 2; e++) { cd->ej_nozzle_fouling[e] = 1.0; cd->ej_diffuser_fouling[e] = 1.0; cd->ej_nozzle_erosion[e] = 1.0; }
  cd->ej_operating_mask = 0; cd->lead_ejector = -1; cd->lag_ejector = -1;
}

NPO_FN void npo_plant_init(npo_plant_t *pl, const npb_params_t *P) {
  (void)P;
  npo_prim_init(&pl->prim);
  for (int i = 0; i < NPB_NUM_SG; i++) npo_sg_init(&pl->sg[i]);
  for (int i = 0; i < NPB_NUM_PUMPS; i++) npo_pump_init(&pl->pump[i], i);
  npo_fw_init(&pl->fw);
  npo_turb_init(&pl->turb, &pl->tstg);
  for (int i = 0; i < 2; i++) npo_chem_init(&pl->chem[i], i);
  npo_ph_init(&pl->ph);
  npo_cond_init(&pl->cond);
  npo_sec_init(&pl->sec);
  for (int i = 0; i < NPB_NUM_PUMPS; i++) npo_mpump_init(&pl->mpump[i]);
  npo_maint_init(&pl->maint);
}
#endif
