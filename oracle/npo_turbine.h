/*
 * npo_turbine.h -- CPU oracle: turbine (14-stage expansion chain, rotor dynamics, 4 bearings,
 * metal-temperature tracker, protection trips, bearing lubrication system).
 * TEST INFRASTRUCTURE ONLY (see npo_common.h).
 *
 * Follows EnhancedTurbinePhysics.update_state  turbine/enhanced_physics.py:694-890 as wrapped at
 * construction by integrate_lubrication_with_turbine  turbine/turbine_bearing_lubrication.py:704-797.
 * dt is in HOURS here (secondary/__init__.py:568 passes dt/60).
 */
#ifndef NPO_TURBINE_H
#define NPO_TURBINE_H
#include "npo_common.h"
#include "npo_lube.h"

/* Antoine-form saturation temperature used by the turbine and condenser-side helpers
 * stage_system.py:458-466, enhanced_physics.py:1295-1303 */
NPO_FN double npo_tsat_antoine(double pressure_mpa) {
  if (pressure_mpa <= 0.001) return 10.0;
  const double A = 8.07131, B = 1730.63, C = 233.426;
  double pressure_bar = npo_clip(pressure_mpa * 10.0, 0.01, 100.0);
  double temp_c = B / (A - log10(pressure_bar)) - C;
  return npo_clip(temp_c, 10.0, 374.0);
}
/* _saturation_enthalpy_vapor  stage_system.py:468-473 */
NPO_FN double npo_hg_antoine(double pressure_mpa) {
  double temp = npo_tsat_antoine(pressure_mpa);
  double h_f = 4.18 * temp;
  double h_fg = 2257.0 * pow(1.0 - temp / 374.0, 0.38);
  return h_f + h_fg;
}
/* TurbineStage._steam_enthalpy  stage_system.py:418-443 */
NPO_FN double npo_stage_steam_enthalpy(double temp_c, double pressure_mpa) {
  pressure_mpa = npo_pymax(0.001, npo_pymin(pressure_mpa, 22.0));
  temp_c = npo_pymax(0.0, npo_pymin(temp_c, 800.0));
  double sat_temp = npo_tsat_antoine(pressure_mpa);
  if (temp_c <= sat_temp) return npo_hg_antoine(pressure_mpa);
  double h_g = npo_hg_antoine(pressure_mpa);
  double superheat = temp_c - sat_temp;
  double cp = (pressure_mpa > 10.0) ? 2.5 : ((pressure_mpa > 1.0) ? 2.2 : 2.0);
  return h_g + cp * superheat;
}
/* _enthalpy_to_temperature  stage_system.py:482-491 */
NPO_FN double npo_stage_enthalpy_to_temperature(double enthalpy, double pressure) {
  double sat_temp = npo_tsat_antoine(pressure);
  double h_g = npo_hg_antoine(pressure);
  if (enthalpy <= h_g) return sat_temp;
  return sat_temp + (enthalpy - h_g) / 2.1;
}

/* default stage table  TurbineStageSystem._create_default_stages  stage_system.py:683-756
 * (np.linspace(6.895, 1.2, 9) and np.linspace(1.15, 0.007, 7): y[i] = i*step + start, y[-1] = stop) */
NPO_FN void npo_stage_design(int k, double *p_in, double *p_out, double *design_flow, int *has_extraction, int *is_lp) {
  if (k < 8) {
    const double start = 6.895, stop = 1.2; const double step = (stop - start) / 8;
    *p_in = k * step + start;
    *p_out = (k + 1 == 8) ? stop : (k + 1) * step + start;
    *design_flow = 555.0; *has_extraction = (k >= 2 && k < 6); *is_lp = 0;
  } else {
    int i = k - 8;
    const double start = 1.15, stop = 0.007; const double step = (stop - start) / 6;
    *p_in = i * step + start;
    *p_out = (i + 1 == 6) ? stop : (i + 1) * step + start;
    *design_flow = 555.0 * 2; *has_extraction = (i < 3); *is_lp = 1;
  }
}

typedef struct npo_stage_out_t {
  double power_output, outlet_pressure, outlet_temperature, outlet_enthalpy, outlet_flow, extraction_flow, loading_factor;
} npo_stage_out_t;

/* TurbineStage.calculate_stage_expansion  stage_system.py:98-292 */
NPO_FN void npo_stage_expansion(int k, double actual_efficiency, double blade_condition_factor, double fouling_factor,
                                double blade_wear_factor, double inlet_pressure, double inlet_temperature, double inlet_flow,
                                double outlet_pressure, double extraction_demand, npo_stage_out_t *o) {
  double d_in, d_out, design_flow; int has_extraction, is_lp;
  npo_stage_design(k, &d_in, &d_out, &design_flow, &has_extraction, &is_lp);
  double design_pressure_ratio = d_out / d_in;
  double load_factor = (design_flow > 0) ? inlet_flow / design_flow : 1.0;
  load_factor = npo_clip(load_factor, 0.3, 1.5);
  double temp_factor = (inlet_temperature + 273.15) / (285.8 + 273.15);
  temp_factor = npo_clip(temp_factor, 0.8, 1.2);
  double load_adjustment = 0.9 + 0.2 * load_factor;
  double temp_adjustment = 0.95 + 0.1 * (temp_factor - 1.0);
  double adjusted_pressure_ratio = design_pressure_ratio * load_adjustment * temp_adjustment;
  adjusted_pressure_ratio = npo_clip(adjusted_pressure_ratio, design_pressure_ratio * 0.85, design_pressure_ratio * 1.15);
  double physics_based_outlet_pressure = inlet_pressure * adjusted_pressure_ratio;
  double self_outlet_pressure;
  if (outlet_pressure >= inlet_pressure) {
    self_outlet_pressure = physics_based_outlet_pressure;
  } else {
    double min_allowed, max_allowed;
    if (k == 13) { min_allowed = 0.002; max_allowed = 0.009; } /* LP-6 */
    else { min_allowed = inlet_pressure * (design_pressure_ratio * 0.7); max_allowed = inlet_pressure * (design_pressure_ratio * 1.3); }
    if (outlet_pressure < min_allowed) self_outlet_pressure = min_allowed;
    else if (outlet_pressure > max_allowed) self_outlet_pressure = max_allowed;
    else self_outlet_pressure = outlet_pressure;
  }
  double inlet_enthalpy = npo_stage_steam_enthalpy(inlet_temperature, inlet_pressure);
  double extraction_flow = 0.0, extraction_enthalpy = 0.0;
  if (has_extraction && extraction_demand > 0) {
    extraction_flow = npo_clip(extraction_demand, 5.0, npo_pymin(50.0, inlet_flow * 0.3));
    double extraction_pressure = inlet_pressure * 0.7 + outlet_pressure * (1 - 0.7);
    double extraction_temp = npo_tsat_antoine(extraction_pressure);
    extraction_enthalpy = npo_stage_steam_enthalpy(extraction_temp, extraction_pressure);
  }
  double outlet_flow = inlet_flow - extraction_flow;
  double pr = self_outlet_pressure / inlet_pressure;
  double outlet_temp_isentropic = (inlet_temperature + 273.15) * pow(pr, 0.25) - 273.15;
  double outlet_enthalpy_isentropic = npo_stage_steam_enthalpy(outlet_temp_isentropic, self_outlet_pressure);
  double quality_efficiency_factor = 1.0; /* steam_quality is the hard-coded 0.99 (:206) */
  double total_efficiency = (actual_efficiency * blade_condition_factor * fouling_factor * blade_wear_factor * quality_efficiency_factor);
  double isentropic_enthalpy_drop = inlet_enthalpy - outlet_enthalpy_isentropic;
  if (isentropic_enthalpy_drop <= 0) {
    double min_enthalpy_drop = 50.0 * (1.0 - self_outlet_pressure / inlet_pressure);
    isentropic_enthalpy_drop = npo_pymax(min_enthalpy_drop, 10.0);
  }
  double actual_enthalpy_drop = total_efficiency * isentropic_enthalpy_drop;
  if (actual_enthalpy_drop <= 0) actual_enthalpy_drop = npo_pymax(1.0, isentropic_enthalpy_drop * 0.5);
  double outlet_enthalpy = inlet_enthalpy - actual_enthalpy_drop;
  double outlet_temperature = npo_stage_enthalpy_to_temperature(outlet_enthalpy, outlet_pressure);
  double main_power = outlet_flow * actual_enthalpy_drop / 1000.0;
  if (main_power < 0) main_power = 0.0;
  double extraction_power = 0.0;
  if (extraction_flow > 0) extraction_power = extraction_flow * (inlet_enthalpy - extraction_enthalpy) / 1000.0;
  double design_enthalpy_drop = 0.88 * isentropic_enthalpy_drop;
  o->power_output = main_power + extraction_power;
  o->outlet_pressure = self_outlet_pressure; o->outlet_temperature = outlet_temperature;
  o->outlet_enthalpy = outlet_enthalpy; o->outlet_flow = outlet_flow; o->extraction_flow = extraction_flow;
  o->loading_factor = actual_enthalpy_drop / npo_pymax(1.0, design_enthalpy_drop);
}

/* get_dynamic_pressure_ratio closure  stage_system.py:794-868 */
NPO_FN double npo_stage_dynamic_pressure_ratio(int k, double current_pressure, double inlet_flow) {
  double d_in, d_out, design_flow; int has_extraction, is_lp;
  npo_stage_design(k, &d_in, &d_out, &design_flow, &has_extraction, &is_lp);
  double design_pressure_ratio = d_out / d_in;
  double load_factor = (design_flow > 0) ? inlet_flow / design_flow : 1.0;
  load_factor = npo_clip(load_factor, 0.3, 1.5);
  double pressure_factor = (d_in > 0) ? current_pressure / d_in : 1.0;
  pressure_factor = npo_clip(pressure_factor, 0.5, 1.5);
  double load_adjustment = 0.90 + 0.2 * (load_factor - 1.0);
  double pressure_adjustment = 0.95 + 0.1 * (pressure_factor - 1.0);
  double dynamic_ratio = design_pressure_ratio * load_adjustment * pressure_adjustment;
  double min_ratio, max_ratio;
  if (!is_lp) { min_ratio = 0.70; max_ratio = 0.95; } else { min_ratio = 0.50; max_ratio = 0.85; }
  if (is_lp) {
    int remaining_stages = 14 - k - 1;
    if (remaining_stages > 0) {
      double min_outlet_pressure = 0.007 / pow(0.85, (double)remaining_stages);
      double max_allowable_ratio = min_outlet_pressure / current_pressure;
      min_ratio = npo_pymax(min_ratio, max_allowable_ratio);
    }
  }
  if (k == 13) dynamic_ratio = npo_pymax(0.007 / current_pressure, 0.05);
  else dynamic_ratio = npo_clip(dynamic_ratio, min_ratio, max_ratio);
  return dynamic_ratio;
}

typedef struct npo_stagesys_out_t {
  double total_power, total_extraction, lp6_outlet_enthalpy;
  double stage_outlet_temperature[14];
  double hp_power, lp_power;     /* sums of the HP-1..8 / LP-1..6 stage outputs (enhanced_physics.py:879-880) */
  double overall_efficiency;     /* stage_system.py:983-993 */
} npo_stagesys_out_t;

/* TurbineStageSystem.update_state  stage_system.py:928-1016 (+ calculate_stage_by_stage_expansion :760-926,
 * TurbineStage.update_degradation :294-339).  The control-logic pass (:525-651) only fills a command dict. */
NPO_FN void npo_stage_system_update(npb_tstg_t *t, double inlet_pressure, double inlet_temperature, double inlet_flow,
                                    double load_demand, double pressure_stability_factor, double dt, npo_stagesys_out_t *out) {
  /* extraction_demands dict  enhanced_physics.py:729-735 */
  double extraction_demand[14] = {0};
  extraction_demand[2] = 25.0 * load_demand; extraction_demand[3] = 30.0 * load_demand; extraction_demand[4] = 20.0 * load_demand;
  extraction_demand[8] = 15.0 * load_demand; extraction_demand[9] = 10.0 * load_demand;
  double current_pressure = inlet_pressure, current_temperature = inlet_temperature, current_flow = inlet_flow;
  const double final_pressure = 0.007;
  double total_power = 0.0, total_extraction = 0.0;
  double hp_power = 0.0, lp_power = 0.0, steam_enthalpy_in = 0.0;
  double loading[14];
  for (int k = 0; k < 14; k++) {
    double pressure_ratio = npo_stage_dynamic_pressure_ratio(k, current_pressure, inlet_flow);
    double outlet_pressure = current_pressure * pressure_ratio;
    outlet_pressure = npo_pymax(outlet_pressure, final_pressure);
    int remaining_stages = 14 - k - 1;
    if (remaining_stages == 0) outlet_pressure = final_pressure;
    else if (remaining_stages == 1) outlet_pressure = npo_pymax(outlet_pressure, final_pressure / 0.5);
    if (outlet_pressure >= current_pressure) {
      outlet_pressure = current_pressure * 0.95;
      outlet_pressure = npo_pymax(outlet_pressure, final_pressure);
    }
    /* derived per-stage factors (see npb_fields.h) */
    double fouling_factor = 1.0 / (1.0 + t->stage_deposit_thickness[k] / 0.5);
    double blade_wear_factor = t->stage_blade_wear_factor[k];
    double blade_condition_factor = npo_pymin(fouling_factor, blade_wear_factor);
    double actual_efficiency = npo_pymax(0.7, 0.88 - t->stage_efficiency_degradation[k]);
    npo_stage_out_t so;
    npo_stage_expansion(k, actual_efficiency, blade_condition_factor, fouling_factor, blade_wear_factor, current_pressure,
                        current_temperature, current_flow, outlet_pressure, extraction_demand[k], &so);
    total_power += so.power_output; total_extraction += so.extraction_flow;
    if (k < 8) hp_power += so.power_output; else lp_power += so.power_output;
    if (k == 0) steam_enthalpy_in = npo_stage_steam_enthalpy(inlet_temperature, inlet_pressure);
    out->stage_outlet_temperature[k] = so.outlet_temperature;
    if (k == 13) out->lp6_outlet_enthalpy = so.outlet_enthalpy;
    loading[k] = so.loading_factor;
    current_pressure = so.outlet_pressure; current_temperature = so.outlet_temperature; current_flow = so.outlet_flow;
  }
  for (int k = 0; k < 14; k++) { /* update_degradation */
    t->stage_efficiency_degradation[k] += 1e-05 * dt;
    t->stage_deposit_thickness[k] += 5e-05 * dt;
    double blade_wear = (1e-06 * dt) * pow(loading[k], 2.0);
    t->stage_blade_wear_factor[k] = npo_pymax(0.7, t->stage_blade_wear_factor[k] - blade_wear);
  }
  out->total_power = total_power * pressure_stability_factor;
  out->total_extraction = total_extraction;
  out->hp_power = hp_power; out->lp_power = lp_power;
  if (inlet_flow > 0) {   /* stage_system.py:984-991: the last stage's outlet conditions as handed on by the expansion loop */
    double outlet_enthalpy = npo_stage_steam_enthalpy(current_temperature, current_pressure);
    out->overall_efficiency = (steam_enthalpy_in - outlet_enthalpy) / steam_enthalpy_in;
  } else out->overall_efficiency = 0.0;
}

/* _calculate_pressure_variation_effects  enhanced_physics.py:1312-1350 */
NPO_FN double npo_pressure_stability_factor(const double *sg_pressures) {
  double avg_pressure = (0.0 + sg_pressures[0] + sg_pressures[1] + sg_pressures[2]) / 3;
  double max_deviation = fabs(sg_pressures[0] - avg_pressure);
  for (int i = 1; i < 3; i++) max_deviation = npo_pymax(max_deviation, fabs(sg_pressures[i] - avg_pressure));
  double variation_factor = max_deviation / 0.1;
  double stability_factor;
  if (max_deviation < 0.02) stability_factor = 1.0;
  else if (max_deviation < 0.05) stability_factor = 1.0 - (max_deviation - 0.02) / 0.03 * 0.05;
  else stability_factor = 0.95 - npo_pymin(variation_factor - 0.5, 0.25);
  return npo_clip(stability_factor, 0.7, 1.0);
}

/* LubricationComponent tables  turbine_bearing_lubrication.py:97-173 */
static const double NPO_TLUB_BASE[5] = {0.0003, 0.0004, 0.0006, 0.0008, 0.0001};
static const double NPO_TLUB_LOAD_EXP[5] = {1.8, 1.6, 2.5, 1.4, 1.0};
static const double NPO_TLUB_SPEED_EXP[5] = {1.5, 1.5, 1.3, 1.1, 0.5};
static const double NPO_TLUB_CONTAM[5] = {3.0, 2.8, 4.0, 3.5, 1.5};
static const double NPO_TLUB_OIL_FLOW[4] = {25.0, 30.0, 40.0, 15.0}; /* oil_flow_requirement injected into TB-001..004 */

typedef struct npo_turbine_result_t {
  double electrical_power_net, electrical_power_gross, mechanical_power, effective_steam_flow;
  double condenser_pressure, condenser_temperature, lp6_outlet_enthalpy;
  double hp_power, lp_power, overall_efficiency;   /* enhanced_physics.py:840,879-880 */
  int trip_active;
} npo_turbine_result_t;

NPO_FN void npo_turbine_update(npb_turb_t *t, npb_tstg_t *g, double steam_pressure, double steam_temperature, double steam_flow,
                               const double *sg_pressures, int sg_system_availability, double load_demand,
                               double condenser_pressure, double dt, npo_turbine_result_t *res) {
  /* ================= lubrication wrapper (pre-step, previous step's bearing state) =============
   * update_with_lubrication  turbine_bearing_lubrication.py:715-784: called with keyword arguments
   * only, so every rotor quantity takes its default and the load factor is turbine.load_demand
   * as left by the PREVIOUS step (:744-746). */
  double lub_load_factor = t->load_demand;
  double friction_heat[4], b_load_factor[4], b_speed_factor, b_temperature[4];
  b_speed_factor = t->rotor_speed / 3600.0;
  double total_heat_generation = 0.0;
  for (int i = 0; i < 4; i++) { /* collect_bearing_states :800-838, calculate_bearing_friction_heat :841-863 */
    double load_n = t->bearing_load[i] * 1000.0;
    double clearance_m = 0.15 / 1000.0;
    double omega = t->rotor_speed * 2 * NPO_PI / 60.0;
    double friction_power = load_n * 0.001 * omega * clearance_m;
    friction_heat[i] = npo_pymax(0.0, friction_power);
    b_load_factor[i] = t->bearing_load[i] / npo_pymax(500.0, 1.0);
    b_temperature[i] = t->bearing_metal_temp[i];
    total_heat_generation += friction_heat[i];
  }
  /* update_lubrication_with_feedback :866-928 */
  double base_oil_temp = 40.0 + lub_load_factor * 15.0;
  double system_oil_temp;
  if (total_heat_generation > 0) {
    double oil_mass_flow = 100.0 / 60.0 * 0.85;
    system_oil_temp = base_oil_temp + total_heat_generation / (oil_mass_flow * 2000.0);
  } else {
    system_oil_temp = base_oil_temp;
  }
  double contamination_input = lub_load_factor * 0.02 + (1.0 - 0.99) * 0.5;
  double moisture_input = (1.0 - 0.99) * 0.01;
  double avg_wear = 0.0;
  for (int i = 0; i < 5; i++) avg_wear += t->lub_wear[i];
  avg_wear = avg_wear / 5;
  npo_oil_t oil = {&t->lub_oil_temperature, &t->lub_oil_contamination, &t->lub_oil_moisture, &t->lub_oil_acidity,
                   &t->lub_oil_viscosity_change, &t->lub_antioxidant_level, &t->lub_anti_wear_level,
                   &t->lub_corrosion_inhibitor_level, &t->lub_effectiveness};
  const npo_oil_limits_t lim = {8.0, 0.3, 0.05, 20.0};
  npo_update_oil_quality(&oil, &lim, avg_wear, system_oil_temp, contamination_input, moisture_input, dt);
  /* update_component_wear with calculate_component_wear :261-333; map_bearing_to_lubrication_components :931-964 */
  for (int i = 0; i < 5; i++) {
    double wear_rate;
    if (i == 0) {
      double steam_temp_factor = npo_pymax(1.0, (b_temperature[0] - 70.0) / 20.0);
      double load_factor_adj = b_load_factor[0] * 1.2;
      wear_rate = (NPO_TLUB_BASE[0] * pow(load_factor_adj, NPO_TLUB_LOAD_EXP[0]) * pow(b_speed_factor, NPO_TLUB_SPEED_EXP[0]) * steam_temp_factor);
    } else if (i == 1) {
      double moisture_factor = npo_pymax(1.0, (1.0 - 0.99) * 10.0);
      double temp_factor = npo_pymax(1.0, (b_temperature[1] - 60.0) / 25.0);
      wear_rate = (NPO_TLUB_BASE[1] * pow(b_load_factor[1], NPO_TLUB_LOAD_EXP[1]) * pow(b_speed_factor, NPO_TLUB_SPEED_EXP[1]) * moisture_factor * temp_factor);
    } else if (i == 2) {
      double axial_load_factor = b_load_factor[2] * 1.0;
      double temp_factor = npo_pymax(1.0, (b_temperature[2] - 50.0) / 30.0);
      wear_rate = (NPO_TLUB_BASE[2] * pow(axial_load_factor, NPO_TLUB_LOAD_EXP[2]) * pow(b_speed_factor, NPO_TLUB_SPEED_EXP[2]) * temp_factor);
    } else if (i == 3) {
      double contamination_factor = 1.0 + t->lub_oil_contamination / 10.0;
      wear_rate = (NPO_TLUB_BASE[3] * pow(1.0, NPO_TLUB_LOAD_EXP[3]) * contamination_factor);
    } else {
      wear_rate = (NPO_TLUB_BASE[4] * 1.0 * 1.0); /* oil_coolers: no bearing maps to it -> defaults */
    }
    double lubrication_wear_factor = 1.0 + (1.0 - t->lub_effectiveness) * NPO_TLUB_CONTAM[i];
    t->lub_wear[i] += (wear_rate * lubrication_wear_factor) * dt;
  }

  /* ================= EnhancedTurbinePhysics.update_state  enhanced_physics.py:694-890 ========= */
  t->load_demand = load_demand;
  double pressure_stability_factor = npo_pressure_stability_factor(sg_pressures);
  npo_stagesys_out_t ss = {0};
  npo_stage_system_update(g, steam_pressure, steam_temperature, steam_flow, load_demand, pressure_stability_factor, dt, &ss);
  double stage_power_mw = ss.total_power;
  double applied_torque = stage_power_mw * 1e6 / (2 * NPO_PI * 3600 / 60);

  /* ---- RotorDynamicsModel.update_state  rotor_dynamics.py:956-1070 */
  /* calculate_rotor_dynamics :855-911 */
  double dt_seconds = dt * 3600.0;
  double total_friction = 0.0;
  for (int i = 0; i < 4; i++) total_friction += (t->bearing_load[i] * 1000.0 * 0.001 * 0.15 / 1000.0);
  double net_torque = applied_torque - total_friction;
  double angular_acceleration = net_torque / 45000.0;
  double rotor_acceleration = angular_acceleration * 60.0 / (2 * NPO_PI);
  t->rotor_speed += rotor_acceleration * dt_seconds;
  t->rotor_speed = npo_pymax(0.0, npo_pymin(t->rotor_speed, 3780.0));
  /* calculate_thermal_effects :913-954 */
  double temp_change = (steam_temperature - t->rotor_temperature) / 2.0 * dt;
  t->rotor_temperature += temp_change;
  double temp_difference = t->rotor_temperature - 25.0;
  t->thermal_expansion = (temp_difference * 1.2e-05 * 12.0 * 1000.0);
  if (t->rotor_speed < 100.0) {
    double thermal_gradient = (dt > 0) ? fabs(temp_change) / dt : 0;
    double bow_increase = thermal_gradient * 0.001 * dt;
    t->thermal_bow = npo_pymin(2.0, t->thermal_bow + bow_increase);
  } else {
    t->thermal_bow *= 0.95;
  }
  /* bearings */
  double steam_thrust = 100.0 * load_demand;
  double rotor_weight_per_bearing = 150000.0 * 9.81 / 1000.0 / 4;
  double max_bearing_metal = 0.0;
  for (int i = 0; i < 4; i++) {
    /* calculate_bearing_loads :83-130 */
    double static_load = rotor_weight_per_bearing;
    double thrust_load = (i == 2) ? steam_thrust / 4 : 0.0;
    double thermal_load = fabs(t->thermal_expansion) * 100000000.0 / 1000.0;
    double unbalance_force = pow(t->rotor_speed / 3600.0, 2.0) * 0.1;
    double total_load = static_load + thrust_load + thermal_load + unbalance_force;
    total_load *= (2.0 - t->bearing_wear_factor[i]);
    t->bearing_load[i] = total_load;
    /* calculate_bearing_temperature :157-270 (oil inlet 40 C, flow injected by the lubrication wrapper) */
    double oil_inlet_temp = npo_pymax(20.0, npo_pymin(150.0, 40.0));
    double bearing_load = npo_pymax(0.0, total_load);
    double rotor_speed = npo_pymax(0.0, t->rotor_speed);
    double bearing_load_n = bearing_load * 1000.0;
    double angular_velocity = rotor_speed * 2 * NPO_PI / 60.0;
    double friction_torque = 0.001 * bearing_load_n * 0.15;
    double friction_power = friction_torque * angular_velocity;
    friction_power = npo_pymin(friction_power, 50000.0);
    if (!isfinite(friction_power) || friction_power < 0) friction_power = 0.0;
    double oil_mass_flow = NPO_TLUB_OIL_FLOW[i] / 60.0 * 850.0 / 1000.0;
    double temp_rise = friction_power / (oil_mass_flow * 2000.0);
    temp_rise = npo_pymin(50.0, npo_pymax(0.0, temp_rise));
    t->bearing_metal_temp[i] = npo_pymax(30.0, npo_pymin(200.0, oil_inlet_temp + temp_rise * 1.5));
    /* update_bearing_wear :272-315 (oil_contamination argument is the constant 5.0) */
    double lf = total_load / 500.0;
    double load_wear_rate = 0.00001 * pow(lf, 2.0) * dt;
    double contamination_wear_rate = 0.000005 * 5.0 * dt;
    t->bearing_wear_factor[i] = npo_pymax(0.5, t->bearing_wear_factor[i] - (load_wear_rate + contamination_wear_rate));
    max_bearing_metal = npo_pymax(max_bearing_metal, t->bearing_metal_temp[i]);
  }
  /* VibrationMonitor.calculate_vibration_response :624-704 */
  double avg_stiffness = (0.0 + 1e8 + 1e8 + 1e8 + 1e8) / 4, avg_damping = (0.0 + 1e5 + 1e5 + 1e5 + 1e5) / 4;
  double vib_unbalance_force = pow(t->rotor_speed / 60.0, 2.0) * 0.1;
  double rotation_frequency = t->rotor_speed / 60.0;
  const double rotor_mass = 15000.0;
  double natural_frequency = sqrt(avg_stiffness / rotor_mass) / (2 * NPO_PI);
  double frequency_ratio = rotation_frequency / natural_frequency;
  double critical_damping = 2 * sqrt(avg_stiffness * rotor_mass);
  double damping_ratio = avg_damping / critical_damping;
  double denominator = sqrt(pow(1 - pow(frequency_ratio, 2.0), 2.0) + pow(2 * damping_ratio * frequency_ratio, 2.0));
  double unbalance_response = vib_unbalance_force / avg_stiffness / denominator;
  double thermal_response = t->thermal_bow * pow(frequency_ratio, 2.0) / denominator;
  double displacement_1x = (unbalance_response + thermal_response) * 39.37;
  double displacement_2x = displacement_1x * 0.1, displacement_3x = displacement_1x * 0.05;
  double total_displacement = sqrt(pow(displacement_1x, 2.0) + pow(displacement_2x, 2.0) + pow(displacement_3x, 2.0));
  t->vibration_displacement = total_displacement;

  /* ---- MetalTemperatureTracker.update_temperatures  enhanced_physics.py:73-166 (time constant 3600 s = 1 h) */
  const double time_constant = 3600.0 / 3600.0, ambient = 25.0;
  double max_temp_rate = 0.0, max_stress = 0.0;
  for (int i = 0; i < 8; i++) {
    double target_temp = ss.stage_outlet_temperature[i] - 50.0;
    double tc = (target_temp - g->rotor_temperatures[i]) / time_constant * dt;
    double max_rate = 5.0 * dt;
    tc = npo_clip(tc, -max_rate, max_rate);
    g->rotor_temperatures[i] += tc;
    double rate = tc / dt * 60.0;
    max_temp_rate = (i == 0) ? fabs(rate) : npo_pymax(max_temp_rate, fabs(rate));
  }
  for (int i = 0; i < 6; i++) {
    double target_temp = ss.stage_outlet_temperature[i] - 80.0;
    double tc = (target_temp - g->casing_temperatures[i]) / time_constant * dt;
    tc = npo_clip(tc, -3.0 * dt, 3.0 * dt);
    g->casing_temperatures[i] += tc;
  }
  for (int i = 0; i < 14; i++) {
    double target_temp = ss.stage_outlet_temperature[i] - 20.0;
    double tc = (target_temp - g->blade_temperatures[i]) / (time_constant * 0.5) * dt;
    tc = npo_clip(tc, -10.0 * dt, 10.0 * dt);
    g->blade_temperatures[i] += tc;
  }
  for (int i = 0; i < 8; i++) {
    double temp_diff = g->rotor_temperatures[i] - ambient;
    double thermal_strain = 1.2e-05 * temp_diff;
    double stress = thermal_strain * 200000000000.0 * 0.1;
    max_stress = (i == 0) ? stress : npo_pymax(max_stress, stress);
  }
  (void)max_temp_rate;

  /* ---- TurbineProtectionSystem.check_trip_conditions  enhanced_physics.py:348-436 */
  int trips = 0, latched = t->trip_latched_mask;
  if (t->rotor_speed > 3780.0) { t->timer_overspeed += dt_seconds; if (t->timer_overspeed >= 0.1) { trips |= 1; latched |= 1; } }
  else t->timer_overspeed = 0.0;
  if (total_displacement > 25.0) { t->timer_vibration += dt_seconds; if (t->timer_vibration >= 2.0) { trips |= 2; latched |= 2; } }
  else t->timer_vibration = 0.0;
  if (max_bearing_metal > 120.0) { t->timer_bearing_temp += dt_seconds; if (t->timer_bearing_temp >= 10.0) { trips |= 4; latched |= 4; } }
  else t->timer_bearing_temp = 0.0;
  if (t->thermal_expansion > 50.0) { trips |= 8; latched |= 8; }
  if (condenser_pressure > 0.012) { trips |= 16; latched |= 16; }
  if (max_stress > 800000000.0) { trips |= 32; latched |= 32; }
  t->trip_active = trips != 0;
  t->trip_latched_mask = latched;
  double power_reduction = trips ? (t->trip_active ? 0.0 : 1.0) : 1.0;
  double sg_availability_factor = sg_system_availability ? 1.0 : 0.5;
  t->total_power_output = stage_power_mw * (power_reduction * sg_availability_factor);

  res->electrical_power_gross = t->total_power_output;
  res->mechanical_power = t->total_power_output / 0.985;
  res->electrical_power_net = t->total_power_output * 0.98;
  res->effective_steam_flow = steam_flow - ss.total_extraction;
  res->condenser_pressure = condenser_pressure;
  res->condenser_temperature = npo_tsat_antoine(condenser_pressure);
  res->lp6_outlet_enthalpy = ss.lp6_outlet_enthalpy;
  res->hp_power = ss.hp_power; res->lp_power = ss.lp_power; res->overall_efficiency = ss.overall_efficiency;
  res->trip_active = t->trip_active;
}

#endif
