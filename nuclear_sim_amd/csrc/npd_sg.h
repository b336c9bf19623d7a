/*
 * npd_sg.h -- device physics: three U-tube steam generators with TSP and tube-interior fouling.
 *
 * Follows EnhancedSteamGeneratorPhysics.update_system  steam_generator/enhanced_physics.py:433-547
 * and SteamGenerator.update_state  steam_generator/steam_generator.py:664-848.
 */
#ifndef NPD_SG_H
#define NPD_SG_H
#include "npd_common.h"

/* SteamGenerator._saturation_temperature  steam_generator.py:854-888 */
NPD_FN double npd_sg_tsat(double pressure_mpa) {
  if (pressure_mpa <= 0.001) return 10.0;
  double pressure_bar = pressure_mpa * 10.0;
  double ln_p = npd_log_pos(pressure_bar);   /* > 0.01 bar here (or NaN); the lanes the early return covers discard it */
  double temp_c = 42.6776 + 34.5194 * ln_p + 2.8896 * npd_sq(ln_p) + 0.1153 * (ln_p * ln_p * ln_p);
  return npd_clip(temp_c, 10.0, 374.0);
}
/* :890-906 */
NPD_FN double npd_sg_hf(double p) { return 4.18 * npd_sg_tsat(p); }
NPD_FN double npd_sg_hg(double p) {
  double temp = npd_sg_tsat(p);
  double h_f = npd_sg_hf(p);
  double h_fg = 2257.0 * npd_powc(1.0 - temp / 374.0, 0.38);
  return h_f + h_fg;
}
/* :908-941 */
NPD_FN double npd_sg_water_enthalpy(double temp_c, double p) {
  return 4.18 * temp_c + 0.001 * (p - 0.1) * temp_c;
}
NPD_FN double npd_sg_water_density(double temp_c, double p) {
  double rho_temp = 1000.0 * (1.0 - 0.0003 * temp_c);
  double pressure_effect = 1.0 + 4.5e-10 * p * 1e6;
  return rho_temp * pressure_effect;
}
NPD_FN double npd_sg_steam_density(double temp_c, double p) {
  return (p * 1e6) / (461.5 * (temp_c + 273.15));
}

/* DepositState helpers  tsp_fouling_model.py:110-123 */
NPD_FN double npd_tsp_total_thickness(const npb_sg_t *g, int level) {
  return g->tsp_magnetite[level] + g->tsp_copper[level] + g->tsp_silica[level] + g->tsp_biological[level];
}
NPD_FN double npd_tsp_average_thickness(const npb_sg_t *g) {
  double total = 0.0;
  for (int i = 0; i < NPB_NUM_TSP; i++) total += npd_tsp_total_thickness(g, i);
  return total / NPB_NUM_TSP;
}

/* TSPFoulingModel.calculate_heat_transfer_degradation  tsp_fouling_model.py:342-367 */
NPD_FN double npd_tsp_ht_degradation(double ff) {
  double mixing = ff * npd_sqrt(ff);   /* ff ** 1.5 */
  double maldist = ff * 0.3;
  double total = (mixing + maldist) * 0.6;
  return npd_pymin(total, 0.9);
}

/* TSPFoulingModel.calculate_flow_restriction  tsp_fouling_model.py:302-340 */
NPD_FN void npd_tsp_flow_restriction(const npb_sg_t *g, double *ff, double *pdr, double *levels) {
  double total_restriction = 0.0;
  for (int level = 0; level < NPB_NUM_TSP; level++) {
    double total_thickness = npd_tsp_total_thickness(g, level);
    double hole_diameter_mm = 0.023 * 1000.0;
    double effective_diameter = hole_diameter_mm - 2.0 * total_thickness;
    effective_diameter = npd_pymax(effective_diameter, hole_diameter_mm * 0.1);
    double original_area = NPD_PI * npd_sq(hole_diameter_mm / 2.0);
    double effective_area = NPD_PI * npd_sq(effective_diameter / 2.0);
    double area_ratio = effective_area / original_area;
    double restriction = 1.0 - area_ratio;
    levels[level] = restriction;
    total_restriction += restriction;
  }
  *ff = total_restriction / NPB_NUM_TSP;
  double avg_area_ratio = npd_pymax(1.0 - *ff, 0.1);
  *pdr = npd_sq(1.0 / avg_area_ratio);
}

/* TSPFoulingModel.update_fouling_state  tsp_fouling_model.py:654-724
 * (+ calculate_deposit_formation_rates :195-262, update_deposit_accumulation :264-300,
 *  calculate_flow_maldistribution :369-392, evaluate_shutdown_conditions :413-445) */
NPD_FN void npd_tsp_update(npb_sg_t *g, const npb_params_t *P, double temperature, double flow_velocity, double dt_hours) {
  double dt_seconds = dt_hours * 3600.0;
  g->tsp_operating_years += dt_seconds / (365.25 * 24.0 * 3600.0); /* fouling_model_base.py:98-108 */
  double dt_years = dt_hours / (365.25 * 24.0);

  double temp_kelvin = temperature + 273.15;
  double temp_factor = npd_exp_bounded(-45000.0 / (8.314 * temp_kelvin));   /* a saturation temperature: 283 .. 647 K */
  temp_factor = temp_factor / npd_exp(-45000.0 / (8.314 * 573.15));
  double ph_factor = 1.0 + 0.5 * fabs(P->sgchem_ph - 9.2);
  double velocity_factor = npd_sqrt(flow_velocity / 3.0);
  velocity_factor = npd_clip(velocity_factor, 0.5, 2.0);
  double magnetite_rate = 2.5 * (1.0 + P->sgchem_iron * 1.5) * temp_factor * ph_factor * velocity_factor;
  double copper_rate = 0.8 * (1.0 + P->sgchem_copper * 2.0) * temp_factor * velocity_factor;
  double silica_rate = 1.2 * (1.0 + P->sgchem_silica / 100.0 * 1.8) * temp_factor * ph_factor;
  double bio_temp_factor = (temperature < 60) ? 1.0 : npd_exp_bounded(-(temperature - 60) / 20);
  double biological_rate = 0.5 * (1.0 + P->sgchem_dissolved_oxygen * 10.0) * bio_temp_factor * velocity_factor;

  const double max_thickness = 0.023 / 2.0 * 1000.0 * 0.9;
  for (int level = 0; level < NPB_NUM_TSP; level++) {
    double level_factor = 1.0 + 0.3 * (NPB_NUM_TSP - level - 1) / (NPB_NUM_TSP - 1);
    double magnetite_increase = ((magnetite_rate * level_factor) / 1000.0) / 5.2 * 10.0;
    double copper_increase = ((copper_rate * level_factor) / 1000.0) / 8.9 * 10.0;
    double silica_increase = ((silica_rate * level_factor) / 1000.0) / 2.2 * 10.0;
    double bio_increase = ((biological_rate * level_factor) / 1000.0) / 1.2 * 10.0;
    g->tsp_magnetite[level] += magnetite_increase * dt_years;
    g->tsp_copper[level] += copper_increase * dt_years;
    g->tsp_silica[level] += silica_increase * dt_years;
    g->tsp_biological[level] += bio_increase * dt_years;
    g->tsp_magnetite[level] = npd_pymin(g->tsp_magnetite[level], max_thickness * 0.4);
    g->tsp_copper[level] = npd_pymin(g->tsp_copper[level], max_thickness * 0.2);
    g->tsp_silica[level] = npd_pymin(g->tsp_silica[level], max_thickness * 0.3);
    g->tsp_biological[level] = npd_pymin(g->tsp_biological[level], max_thickness * 0.1);
  }
  double levels[NPB_NUM_TSP];
  npd_tsp_flow_restriction(g, &g->tsp_fouling_fraction, &g->tsp_pressure_drop_ratio, levels);
  g->tsp_ht_degradation = npd_tsp_ht_degradation(g->tsp_fouling_fraction);
  /* flow maldistribution: np.mean / np.std (population) */
  double mean = 0.0;
  for (int i = 0; i < NPB_NUM_TSP; i++) mean += levels[i];
  mean /= NPB_NUM_TSP;
  double var = 0.0;
  for (int i = 0; i < NPB_NUM_TSP; i++) var += (levels[i] - mean) * (levels[i] - mean);
  double stdv = npd_sqrt(var / NPB_NUM_TSP);
  double maldistribution = npd_pymin(stdv / (mean + 0.01), 1.0);
  /* evaluate_shutdown_conditions */
  int shutdown = 0;
  if (g->tsp_fouling_fraction >= 0.85) shutdown = 1;
  if (g->tsp_ht_degradation >= (1.0 - 0.60)) shutdown = 1;
  if (g->tsp_pressure_drop_ratio >= 5.0) shutdown = 1;
  if (maldistribution >= 0.30) shutdown = 1;
  if (g->tsp_operating_years > 40.0 && g->tsp_fouling_fraction > 0.5) shutdown = 1;
  g->tsp_shutdown_required = shutdown;
}

/* TubeInteriorFouling.calculate_thermal_resistance / get_effective_thermal_conductivity
 * tube_interior_fouling.py:190-243 */
NPD_FN double npd_scale_thermal_resistance(const npb_sg_t *g) {
  if (g->scale_thickness <= 0) return 0.0;
  double thickness_m = g->scale_thickness / 1000.0;
  double total_thickness = npd_pymax(g->scale_thickness, 0.001);
  double k = (g->scale_iron_oxide / total_thickness) * .5 + (g->scale_crud / total_thickness) * 0.15 +
             (g->scale_corrosion / total_thickness) * 0.3;
  k = npd_pymax(k, 0.05);
  double r_conduction = thickness_m / k;
  double r_contact = 1e-5;
  double r_fouling = thickness_m * 0.001;
  return r_conduction + r_contact + r_fouling;
}

/* TubeInteriorFouling.update_fouling_state  tube_interior_fouling.py:273-325
 * (+ calculate_scale_formation_rate :117-188, update_scale_buildup :245-271).
 * Primary chemistry passed by SteamGenerator.update_state :721-730 is constant. */
/* calculate_scale_formation_rate :117-188 [mm / year], from the scale thickness the step begins with */
NPD_FN double npd_scale_formation_rate(const npb_sg_t *g, double temperature, double flow_velocity) {
  const double boric_acid = 1000.0, lithium = 2.0, ph = 7.2, dissolved_oxygen = 0.005;
  double temp_kelvin = temperature + 273.15, ref_kelvin = 320.0 + 273.15;
  double temp_factor = npd_exp_bounded(-65000.0 / (8.314 * temp_kelvin)) / npd_exp(-65000.0 / (8.314 * ref_kelvin));
  double boric_acid_factor = 1.0 / (1.0 + boric_acid / 1000.0 * 0.5);
  double lithium_factor = npd_pymax(0.5, 1.0 + (lithium - 2.0) * 0.1);
  double ph_factor = 1.0 + 0.5 * fabs(ph - 7.2);
  double velocity_factor = npd_clip(npd_powc(flow_velocity / 5.0, -0.6), 0.5, 2.0);
  double oxygen_factor = 1.0 + dissolved_oxygen * 10.0;
  double saturation_factor = npd_exp_bounded(-g->scale_thickness / 2.0);
  double formation_rate = 0.001 * temp_factor * boric_acid_factor * lithium_factor * ph_factor *
                          velocity_factor * oxygen_factor * saturation_factor;
  return npd_clip(formation_rate, 0.0, 0.1);
}
NPD_FN void npd_scale_update(npb_sg_t *g, double temperature, double flow_velocity, double dt_seconds) {
  g->scale_operating_years += dt_seconds / (365.25 * 24.0 * 3600.0);
  const double boric_acid = 1000.0, lithium = 2.0, ph = 7.2, dissolved_oxygen = 0.005;
  double temp_kelvin = temperature + 273.15, ref_kelvin = 320.0 + 273.15;
  double temp_factor = npd_exp_bounded(-65000.0 / (8.314 * temp_kelvin)) / npd_exp(-65000.0 / (8.314 * ref_kelvin));
  double boric_acid_factor = 1.0 / (1.0 + boric_acid / 1000.0 * 0.5);
  double lithium_factor = npd_pymax(0.5, 1.0 + (lithium - 2.0) * 0.1);
  double ph_factor = 1.0 + 0.5 * fabs(ph - 7.2);
  double velocity_factor = npd_clip(npd_powc(flow_velocity / 5.0, -0.6), 0.5, 2.0);
  double oxygen_factor = 1.0 + dissolved_oxygen * 10.0;
  double saturation_factor = npd_exp_bounded(-g->scale_thickness / 2.0);
  double formation_rate = 0.001 * temp_factor * boric_acid_factor * lithium_factor * ph_factor *
                          velocity_factor * oxygen_factor * saturation_factor;
  formation_rate = npd_clip(formation_rate, 0.0, 0.1);
  double dt_years = dt_seconds / (365.25 * 24.0 * 3600.0);
  double scale_increase = formation_rate * dt_years;
  g->scale_thickness += scale_increase;
  g->scale_iron_oxide += scale_increase * 0.6;
  g->scale_crud += scale_increase * 0.3;
  g->scale_corrosion += scale_increase * 0.1;
  g->scale_thermal_resistance = npd_scale_thermal_resistance(g);
}

typedef struct npd_sg_result_t {
  double heat_transfer_rate, steam_flow_rate, thermal_efficiency;
} npd_sg_result_t;

/* SteamGenerator.update_state  steam_generator.py:664-848, in two parts.  Part 1 -- heat transfer, tube-wall
 * temperature, TSP and tube-interior fouling -- needs the primary-side conditions only; part 2 -- flow restrictions and
 * the secondary-side dynamics -- is where the actual feedwater flow first enters (:744-760).  The two-wave step
 * kernel runs part 1 while the feedwater pumps are still being updated; everyone else calls them back to back. */
NPD_FN double npd_sg_part1(npb_sg_t *g, const npb_params_t *P, double primary_temp_in, double primary_temp_out,
                          double primary_flow, double dt) {
  /* ---- calculate_heat_transfer :150-314 */
  double sat_temp = npd_sg_tsat(g->secondary_pressure);
  double delta_t1 = primary_temp_in - sat_temp;
  double delta_t2 = primary_temp_out - sat_temp;
  double lmtd;
  if (fabs(delta_t1 - delta_t2) < 1.0) lmtd = (delta_t1 + delta_t2) / 2.0;
  else lmtd = (delta_t1 - delta_t2) / npd_log(delta_t1 / delta_t2);
  double flow_factor = npd_powc(primary_flow / P->sg_primary_design_flow, 0.8);
  double h_primary = P->sg_primary_htc * flow_factor;
  double pressure_factor = npd_powc(g->secondary_pressure / P->sg_design_pressure_secondary, 0.15);
  double h_secondary = P->sg_secondary_htc * pressure_factor;
  double r_primary = 1.0 / h_primary;
  double r_wall = P->sg_tube_wall_thickness / P->sg_tube_conductivity;
  double r_secondary = 1.0 / h_secondary;
  double overall_htc = 1.0 / (r_primary + r_wall + r_secondary);
  double htc_tsp = overall_htc * (1.0 - g->tsp_ht_degradation);
  double htc_all;
  if (g->scale_thermal_resistance > 0) htc_all = 1.0 / (1.0 / htc_tsp + g->scale_thermal_resistance);
  else htc_all = htc_tsp;
  /* calculate_effective_heat_transfer_area :114-148 */
  double level_factor;
  if (g->water_level >= 12.5) level_factor = 1.0;
  else if (g->water_level <= 8.0) level_factor = 0.1;
  else level_factor = 0.1 + 0.9 * (g->water_level - 8.0) / (12.5 - 8.0);
  double effective_area = P->sg_heat_transfer_area * level_factor;
  double heat_transfer = htc_all * effective_area * lmtd;
  double max_heat_from_primary = primary_flow * 5200.0 * (primary_temp_in - primary_temp_out);
  double temp_difference = primary_temp_in - primary_temp_out;
  if (temp_difference < 1.0) heat_transfer = 0.0;
  else if (temp_difference < 5.0) heat_transfer = npd_pymin(heat_transfer, max_heat_from_primary * 0.1);
  else heat_transfer = npd_pymin(heat_transfer, max_heat_from_primary);
  if (primary_flow < 100.0) heat_transfer = 0.0;
  if (heat_transfer < 0) heat_transfer = 0.0;
  double operating_heat_flux = npd_pymax(heat_transfer / P->sg_heat_transfer_area, 5000.0);
  /* _calculate_tsp_scale_thermal_resistance :603-634 */
  double avg_dep = npd_tsp_average_thickness(g);
  double r_scale_secondary = (avg_dep > 0) ? ((avg_dep / 1000.0) / 3.0) * g->tsp_fouling_fraction : 0.0;
  double r_to_wall = 1.0 / h_secondary + r_scale_secondary + (r_wall / 2.0) + g->scale_thermal_resistance;
  g->tube_wall_temp = sat_temp + (operating_heat_flux * r_to_wall);

  /* ---- update_state body */
  g->secondary_temperature = npd_sg_tsat(g->secondary_pressure);
  double tube_cross_section = NPD_PI * npd_sq(P->sg_tube_inner_diameter / 2.0);
  double total_flow_area = P->sg_tube_count * tube_cross_section;
  double avg_velocity = primary_flow / (1000.0 * total_flow_area);
  npd_tsp_update(g, P, g->secondary_temperature, avg_velocity, dt / 3600.0);
  npd_scale_update(g, (primary_temp_in + primary_temp_out) / 2.0, avg_velocity, dt);
  return heat_transfer;
}

NPD_FN void npd_sg_part2(npb_sg_t *g, const npb_params_t *P, double heat_transfer, double steam_flow_out, double feedwater_flow_in,
                         double feedwater_temp, double dt, npd_sg_result_t *res) {
  /* _apply_tsp_flow_restrictions :516-547 */
  double flow_capacity_factor = 1.0 / npd_sqrt(g->tsp_pressure_drop_ratio);
  double actual_steam_flow = npd_pymin(steam_flow_out, P->sg_design_steam_flow_per_sg * flow_capacity_factor);
  double actual_feedwater_flow = npd_pymin(feedwater_flow_in, P->sg_design_feedwater_flow_per_sg * flow_capacity_factor);
  /* (_calculate_primary_flow_restriction :549-601 only feeds the result dict) */

  /* ---- calculate_secondary_side_dynamics :316-514 with ACTUAL flows */
  double p = g->secondary_pressure;
  double tsat = npd_sg_tsat(p);
  double h_f = npd_sg_hf(p), h_g = npd_sg_hg(p);
  double h_fg = h_g - h_f;
  double h_fw = npd_sg_water_enthalpy(feedwater_temp, p);
  double rho_f = npd_sg_water_density(tsat, p);
  double rho_g = npd_sg_steam_density(tsat, p);
  double mass_change_rate = actual_feedwater_flow - actual_steam_flow;
  double heat_input_kj = heat_transfer / 1000.0;
  double energy_for_steam_gen = heat_input_kj - actual_feedwater_flow * (h_f - h_fw);
  double steam_generation_rate = npd_pymax(0.0, energy_for_steam_gen / h_fg);
  if (actual_feedwater_flow < 0.1) steam_generation_rate = 0.0;
  double design_heat_input = P->sg_design_thermal_power_per_sg / 1000.0;
  double heat_input_factor = (design_heat_input > 0) ? heat_input_kj / design_heat_input : 0.0;
  double equilibrium_pressure = P->sg_design_pressure_secondary * (0.7 + 0.3 * heat_input_factor);
  equilibrium_pressure = npd_clip(equilibrium_pressure, 3.0, 8.5);
  double steam_demand_factor = (P->sg_secondary_design_flow > 0) ? actual_steam_flow / P->sg_secondary_design_flow : 0.0;
  equilibrium_pressure += -steam_demand_factor * 0.5;
  equilibrium_pressure = npd_clip(equilibrium_pressure, 3.0, 8.5);
  double decay_factor = npd_exp_bounded(-dt / 60.0);
  double base_new_pressure = equilibrium_pressure + (p - equilibrium_pressure) * decay_factor;
  double pressure_corrections = 0.0;
  if (actual_feedwater_flow < 0.1 && actual_steam_flow > 100.0) {
    double inventory_depletion_rate = -actual_steam_flow / P->sg_secondary_water_mass;
    pressure_corrections += inventory_depletion_rate * p * 2.0 * dt;
  }
  double steam_supply_factor = (P->sg_secondary_design_flow > 0) ? steam_generation_rate / P->sg_secondary_design_flow : 0.0;
  double supply_demand_imbalance = steam_supply_factor - steam_demand_factor;
  pressure_corrections += supply_demand_imbalance * 0.005 * dt;
  pressure_corrections = npd_clip(pressure_corrections, -0.2, 0.2);
  double new_pressure = npd_clip(base_new_pressure + pressure_corrections, 1.0, 8.0);
  double sg_cross_section = NPD_PI * npd_sq(4.0 / 2.0);
  double level_change_mass = mass_change_rate * dt / (rho_f * sg_cross_section);
  double volume_expansion = steam_generation_rate * dt * (1.0 / rho_g - 1.0 / rho_f);
  double level_change_swell = volume_expansion / sg_cross_section;
  double new_water_level = npd_clip(g->water_level + (level_change_mass + level_change_swell), 8.0, 16.0);
  double quality_degradation = 0.0;
  if (new_water_level < 11.0) quality_degradation += ((11.0 - new_water_level) / 3.0) * 0.02;
  double q_flow_factor = actual_steam_flow / P->sg_secondary_design_flow;
  if (q_flow_factor > 1.1) quality_degradation += npd_pymin((q_flow_factor - 1.1) * 0.01, 0.03);
  double design_heat_flux = P->sg_design_thermal_power_per_sg / P->sg_heat_transfer_area;
  double heat_flux_ratio = (heat_transfer / P->sg_heat_transfer_area) / design_heat_flux;
  if (heat_flux_ratio > 1.2) quality_degradation += npd_pymin((heat_flux_ratio - 1.2) * 0.005, 0.02);
  double target_quality = npd_clip(0.995 - quality_degradation, 0.90, 1.0);
  double quality_change_rate = (target_quality - g->steam_quality) / 30.0;
  double new_steam_quality = npd_clip(g->steam_quality + quality_change_rate * dt, 0.90, 1.0);

  g->secondary_pressure = new_pressure;
  g->water_level = new_water_level;
  g->steam_quality = new_steam_quality;
  g->steam_flow_rate = actual_steam_flow;
  g->heat_transfer_rate = heat_transfer;
  res->heat_transfer_rate = heat_transfer;
  res->steam_flow_rate = actual_steam_flow;
  res->thermal_efficiency = heat_transfer / P->sg_design_thermal_power_per_sg;
}

NPD_FN void npd_sg_update(npb_sg_t *g, const npb_params_t *P, double primary_temp_in, double primary_temp_out,
                          double primary_flow, double steam_flow_out, double feedwater_flow_in,
                          double feedwater_temp, double dt, npd_sg_result_t *res) {
  const double heat_transfer = npd_sg_part1(g, P, primary_temp_in, primary_temp_out, primary_flow, dt);
  npd_sg_part2(g, P, heat_transfer, steam_flow_out, feedwater_flow_in, feedwater_temp, dt, res);
}

typedef struct npd_sgsys_result_t {
  double total_thermal_power, total_steam_flow;
  double avg_pressure, avg_temperature, avg_quality;
  double sg_steam_flow[NPB_NUM_SG];
} npd_sgsys_result_t;

/* EnhancedSteamGeneratorPhysics.update_system  enhanced_physics.py:433-547
 * (_calculate_load_distribution :549-584, _check_system_availability :589-602).
 * fw_flows == NULL means "perfect mass balance" fallback (:495-497). */
NPD_FN void npd_sgsys_update(npb_sg_t *sg, npb_sec_t *sec, const npb_params_t *P, const npd_coupling_t *c,
                             double load_demand_fraction, const double *fw_flows, double feedwater_temperature,
                             double dt, npd_sgsys_result_t *out) {
  double actual_total_steam_flow = P->sg_design_total_steam_flow * load_demand_fraction;
  double demands[NPB_NUM_SG];
  double total_primary_flow = 0.0;
  for (int i = 0; i < NPB_NUM_SG; i++) total_primary_flow += c->flow[i];
  for (int i = 0; i < NPB_NUM_SG; i++) {
    if (total_primary_flow > 0) demands[i] = actual_total_steam_flow * (c->flow[i] / total_primary_flow);
    else demands[i] = actual_total_steam_flow / NPB_NUM_SG;
  }
  npd_sg_result_t r[NPB_NUM_SG];
  for (int i = 0; i < NPB_NUM_SG; i++) {
    double fw = fw_flows ? fw_flows[i] : demands[i];
    npd_sg_update(&sg[i], P, c->inlet_temp[i], c->outlet_temp[i], c->flow[i], demands[i], fw,
                  feedwater_temperature, dt, &r[i]);
  }
  double tp = 0.0, ts = 0.0, ap = 0.0, at = 0.0, aq = 0.0;
  int effective = 0;
  for (int i = 0; i < NPB_NUM_SG; i++) {
    tp += r[i].heat_transfer_rate; ts += r[i].steam_flow_rate;
    ap += sg[i].secondary_pressure; at += sg[i].secondary_temperature; aq += sg[i].steam_quality;
    out->sg_steam_flow[i] = r[i].steam_flow_rate;
    if (r[i].thermal_efficiency > 0.1) effective++;
  }
  out->total_thermal_power = tp; out->total_steam_flow = ts;
  out->avg_pressure = ap / NPB_NUM_SG; out->avg_temperature = at / NPB_NUM_SG; out->avg_quality = aq / NPB_NUM_SG;
  sec->sg_avg_pressure = out->avg_pressure; sec->sg_avg_temperature = out->avg_temperature;
  sec->sg_avg_quality = out->avg_quality;
  sec->sg_system_availability = effective >= (NPB_NUM_SG - 1);
}

#endif
