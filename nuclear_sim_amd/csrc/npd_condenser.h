/*
 * npd_condenser.h -- device physics: condenser (tube degradation, 3-species fouling, vacuum system with
 * two steam-jet ejectors, LMTD energy balance).
 *
 * Follows EnhancedCondenserPhysics.update_state  condenser/physics.py:730-910.
 * dt is in HOURS (secondary/__init__.py:620 passes dt/60).
 */
#ifndef NPD_CONDENSER_H
#define NPD_CONDENSER_H
#include "npd_common.h"
#include "npd_chem.h"
#include "npd_turbine.h"

/* EnhancedCondenserPhysics._saturation_temperature  condenser/physics.py:1824-1838 */
NPD_FN double npd_cond_tsat(double pressure_mpa) {
  if (pressure_mpa <= 0.001) return 10.0;
  const double A = 8.07131, B = 1730.63, C = 233.426;
  double pressure_bar = npd_clip(pressure_mpa * 10.0, 0.01, 100.0);
  double temp_c = B / (A - npd_log_pos(pressure_bar) * 4.34294481903251827651e-01) - C;   /* pressure_bar in [0.01, 100] (or NaN) */
  if (pressure_mpa >= 0.005 && pressure_mpa <= 0.01) temp_c = npd_clip(temp_c, 35.0, 45.0);
  return npd_clip(temp_c, 10.0, 374.0);
}
NPD_FN double npd_cond_hf(double p) { return 4.18 * npd_cond_tsat(p); }      /* :1840-1843 */
NPD_FN double npd_cond_hg(double p) {                                         /* :1845-1850 */
  double temp = npd_cond_tsat(p);
  double h_f = npd_cond_hf(p);
  double h_fg = 2257.0 * npd_powc_pos(1.0 - temp / 374.0, 0.38);   /* temp in [10, 51.6] */
  return h_f + h_fg;
}

NPD_FN double npd_cond_lmtd(double d1, double d2) { /* physics.py:623-638 */
  d1 = npd_pymax(d1, 0.1); d2 = npd_pymax(d2, 0.1);
  if (fabs(d1 - d2) < 0.1) return (d1 + d2) / 2.0;
  if (d1 > 0 && d2 > 0) return (d1 - d2) / npd_log(d1 / d2);
  return (d1 + d2) / 2.0;
}

typedef struct npd_condenser_result_t {
  double heat_rejection_rate, condenser_pressure, condensate_temperature;
} npd_condenser_result_t;

NPD_FN void npd_condenser_update(npb_cond_t *cd, npb_chem_t *chem, double steam_pressure, double steam_flow,
                                 double steam_quality, double cooling_water_flow, double cooling_water_temp_in,
                                 double motive_steam_pressure, double motive_steam_temperature, double dt,
                                 npd_condenser_result_t *res, double *diag = nullptr) {
  /* diag (state-log diagnostics, NPB_DIAG_COND_*): [0] overall heat-transfer coefficient, [1] tube leak rate, [2] the first
   * ejector's motive steam flow, [3] its steam consumption rate, [4] the vacuum system's total motive steam */
  /* own WaterChemistry (:769-800) */
  npd_chem_update(chem, dt);
  double water_aggressiveness = chem->water_aggressiveness;
  double nutrient_level = npd_pymin(2.0, chem->total_dissolved_solids / 500.0);

  const double tube_inner_diameter = 0.0254, initial_tube_count = 84000;
  double tube_area = NPD_PI * npd_sq(tube_inner_diameter / 2.0);
  double total_flow_area = tube_area * cd->active_tube_count;
  double cooling_water_velocity = (cooling_water_flow / 1000.0) / total_flow_area;

  /* ---- TubeDegradationModel.update_tube_failures :73-145 */
  if (cooling_water_velocity > 3.0) cd->vibration_damage += (npd_sq(cooling_water_velocity - 3.0) * 0.001) * dt;
  double corrosion_rate = (1e-07 * water_aggressiveness * (1.0 + cd->vibration_damage));
  double wall_thickness_loss = corrosion_rate * dt;
  cd->average_wall_thickness = npd_pymax(0.001, cd->average_wall_thickness - wall_thickness_loss);
  cd->corrosion_damage += wall_thickness_loss;
  double vibration_factor = 1.0 + 10.0 * cd->vibration_damage;
  double corrosion_factor = 1.0 + 5.0 * (cd->corrosion_damage / 0.00159);
  double chemistry_factor = 1.0 + water_aggressiveness;
  double effective_failure_rate = (1e-06 * vibration_factor * corrosion_factor * chemistry_factor);
  double tubes_failed = effective_failure_rate * cd->active_tube_count * dt;
  tubes_failed = npd_pymin(tubes_failed, cd->active_tube_count * 0.01);
  cd->plugged_tube_count += tubes_failed;
  if (diag) diag[1] = npd_pymin(tubes_failed * 0.1, npd_pymax(1000.0, 84000.0 - cd->plugged_tube_count) * 0.001) * 0.001;   /* physics.py:121-131 (the new active tube count) */
  cd->active_tube_count = npd_pymax(1000.0, initial_tube_count - cd->plugged_tube_count);
  double area_factor = cd->active_tube_count / initial_tube_count;
  double pressure_drop_factor = npd_powc(initial_tube_count / cd->active_tube_count, 1.8);

  /* ---- AdvancedFoulingModel.update_fouling :324-384 */
  double water_temp = (cooling_water_temp_in + cd->cooling_water_outlet_temp) / 2.0;
  {
    double temp_factor = npd_exp_bounded(0.1 * (water_temp - 25.0));
    double chlorine_factor = 1.0 / (1.0 + chem->chlorine_residual * 2.0);
    double nutrient_factor = nutrient_level * 1.0;
    double growth_rate = (0.001 * temp_factor * chlorine_factor * nutrient_factor);
    double thickness_factor = 1.0 / (1.0 + cd->biofouling_thickness / 2.0);
    double bio_increase = npd_pymax(0.0, growth_rate * thickness_factor * (dt / 1000.0));
    double s_temp_factor = npd_exp_bounded(0.15 * (water_temp - 25.0) / 10.0);
    double hardness_factor = (chem->hardness / 150.0) * 0.002;
    double ph_factor = npd_pymax(0.1, (chem->ph - 6.0) / 2.0);
    double antiscalant_factor = 1.0 / (1.0 + chem->antiscalant_concentration / 5.0);
    double formation_rate = (0.0005 * s_temp_factor * hardness_factor * ph_factor * antiscalant_factor);
    double s_thickness_factor = 1.0 / (1.0 + cd->scale_thickness / 1.0);
    double scale_increase = npd_pymax(0.0, formation_rate * s_thickness_factor * (dt / 1000.0));
    double c_temp_factor = npd_exp_bounded((water_temp - 25.0) / 20.0);
    double oxygen_factor = 8.0 * 0.01; /* 'dissolved_oxygen': 8.0 is hard-coded at :791 */
    double c_ph_factor = 1.0 + fabs(chem->ph - 7.5) / 2.0;
    double inhibitor_factor = 1.0 / (1.0 + chem->corrosion_inhibitor_level / 10.0);
    double velocity_factor = 1.0 / (1.0 + cooling_water_velocity / 2.0);
    double c_formation_rate = (0.0002 * c_temp_factor * oxygen_factor * c_ph_factor * inhibitor_factor * velocity_factor);
    double corrosion_increase = npd_pymax(0.0, c_formation_rate * (dt / 1000.0));
    cd->biofouling_thickness += bio_increase;
    cd->scale_thickness += scale_increase;
    cd->corrosion_product_thickness += corrosion_increase;
    cd->time_since_cleaning += dt;
    double total_resistance = (cd->biofouling_thickness / 1000.0) / 0.5 + (cd->scale_thickness / 1000.0) / 2.0 +
                              (cd->corrosion_product_thickness / 1000.0) / 1.0;
    total_resistance *= cd->fouling_distribution_factor;
    cd->total_fouling_resistance = total_resistance;
    cd->fouling_distribution_factor = npd_pymin(1.5, 1.0 + cd->time_since_cleaning / 8760.0);
  }

  /* ---- VacuumSystem.update_state  vacuum_system.py:425-546 */
  const double target_pressure = 0.007;
  double motive_p = motive_steam_pressure - 0.1;
  int motive_available = motive_p > 0.9;
  (void)motive_available; /* the ejectors' own motive_steam_available flag is never cleared */
  cd->current_air_leakage += 1e-05 * dt;                       /* update_air_leakage :358-376 */
  cd->current_air_leakage = npd_pymin(cd->current_air_leakage, 0.05 * 3.0);
  double pressure_error = cd->condenser_pressure - target_pressure; /* calculate_required_capacity :378-405 */
  double required_capacity = cd->current_air_leakage + 50.0 * pressure_error;
  required_capacity = npd_clip(required_capacity, 0.0, (0.0 + 25.0 + 25.0) * 1.2);
  /* VacuumControlLogic.update_control_logic :61-118 (lead_lag strategy) */
  int cmd[2] = {-1, -1}; /* -1 none, 0 stop, 1 start */
  cd->rotation_timer += dt;
  if (cd->lead_ejector < 0) { cd->lead_ejector = 0; cd->lag_ejector = 1; }
  if (!((cd->ej_operating_mask >> cd->lead_ejector) & 1)) cmd[cd->lead_ejector] = 1;
  if (cd->lag_ejector >= 0) {
    int lag_operating = (cd->ej_operating_mask >> cd->lag_ejector) & 1;
    if (cd->condenser_pressure > 0.008 && !lag_operating) cmd[cd->lag_ejector] = 1;
    else if (cd->condenser_pressure < 0.006 && lag_operating) cmd[cd->lag_ejector] = 0;
  }
  if (cd->rotation_timer >= 168.0) { /* _rotate_ejectors :193-229 with both ejectors available */
    int lead_index = cd->lead_ejector, lag_index = cd->lag_ejector;
    int new_lead_index = (lead_index + 1) % 2;
    int new_lag_index = (lag_index >= 0) ? (lag_index + 1) % 2 : -1;
    if (new_lag_index == new_lead_index) new_lag_index = (new_lag_index + 1) % 2;
    int old_lead = cd->lead_ejector;
    cd->lead_ejector = new_lead_index;
    cd->lag_ejector = new_lag_index;
    if (old_lead != cd->lead_ejector) { cmd[old_lead] = 0; cmd[cd->lead_ejector] = 1; }
    cd->rotation_timer = 0.0;
  }
  for (int e = 0; e < 2; e++) {
    if (cmd[e] == 1) { if (!(motive_p < 0.8)) cd->ej_operating_mask |= (1 << e); }  /* start_ejector vacuum_pump.py:277-295 */
    else if (cmd[e] == 0) cd->ej_operating_mask &= ~(1 << e);
  }
  int n_running = ((cd->ej_operating_mask >> 0) & 1) + ((cd->ej_operating_mask >> 1) & 1);
  double total_capacity = 0.0;
  if (diag) { diag[2] = 0.0; diag[3] = 0.0; diag[4] = 0.0;
#pragma unroll
              for (int q = 5; q < 16; q++) diag[q] = 0.0;
              diag[11] = diag[12] = -1.0; }    /* [5 + e] capacity, [7 + e] motive steam flow, [9 + e] steam consumption rate of ejector e; [11 + e] its
                                                * compression ratio of this step (-1 = it did not run: the old one stands), [13 + e] hours it ran
                                                * this step; [15] total air removal  (vacuum_pump.py:470-536, vacuum_system.py:499) */
  for (int e = 0; e < 2; e++) { /* SteamJetEjector.update_state vacuum_pump.py:470-536 */
    int operating = (cd->ej_operating_mask >> e) & 1;
    double capacity = 0.0;
    if (operating) {
      double request = required_capacity / ((n_running > 1) ? n_running : 1);
      double suction = cd->condenser_pressure;
      double overall = cd->ej_nozzle_fouling[e] * cd->ej_diffuser_fouling[e] * cd->ej_nozzle_erosion[e];
      if (diag) { diag[11 + e] = 1.0; diag[13 + e] = dt; }            /* out of its limits: performance.get('compression_ratio', 1.0) */
      if (suction < 0.003 || suction > 0.015) capacity = 0.0;         /* calculate_steam_jet_performance :96-190 */
      else if (motive_p < 0.8) capacity = 0.0;
      else {
        double pressure_capacity_factor = npd_sqrt(motive_p / 1.0);
        double temp_ratio = (motive_steam_temperature + 273.15) / (180.0 + 273.15);
        double temp_capacity_factor = npd_sqrt(npd_sqrt(temp_ratio));   /* ** 0.25 */
        double suction_pressure_ratio = suction / 0.007;
        double suction_capacity_factor = 1.0 / (1.0 + 0.5 * (suction_pressure_ratio - 1.0));
        double available_capacity = (25.0 * pressure_capacity_factor * temp_capacity_factor * suction_capacity_factor * overall);
        capacity = npd_pymax(0.0, npd_pymin(available_capacity, request));
        if (diag && capacity > 0) {   /* the ejector's steam consumption, vacuum_pump.py:150-166 (2.5 kg/kg at design, exponents 0.8 / 1.5) */
          double capacity_factor = capacity / 25.0;
          double rate = (2.5 * npd_exp(0.8 * npd_log(capacity_factor))) * (suction_pressure_ratio * npd_sqrt(suction_pressure_ratio)) * (1.0 / npd_pymax(0.5, overall));
          if (e == 0) { diag[2] = capacity * rate; diag[3] = rate; }
          diag[4] += capacity * rate;
          diag[7 + e] = capacity * rate; diag[9 + e] = rate;
        }
        if (diag) { diag[5 + e] = capacity; diag[11 + e] = 0.101 / suction; }   /* discharge pressure 0.101 MPa, vacuum_pump.py:69, :172 */
      }
      /* update_degradation :246-275 */
      cd->ej_nozzle_fouling[e] = npd_pymax(0.5, cd->ej_nozzle_fouling[e] - 1e-05 * dt);
      cd->ej_diffuser_fouling[e] = npd_pymax(0.6, cd->ej_diffuser_fouling[e] - 2e-05 * dt);
      cd->ej_nozzle_erosion[e] = npd_pymax(0.7, cd->ej_nozzle_erosion[e] - 1e-06 * dt);
    }
    total_capacity += capacity;
  }
  if (diag) diag[15] = total_capacity;
  { /* calculate_air_mass_balance :308-356 */
    double dt_seconds = dt * 3600.0;
    double new_air_mass = cd->air_mass_in_condenser + (cd->current_air_leakage - total_capacity) * dt_seconds;
    new_air_mass = npd_pymax(0.001, new_air_mass);
    double condenser_temp = 39.0 + 273.15;
    double air_density = new_air_mass / 500.0;
    double p_air = (air_density * 287.0 * condenser_temp) / 1e6;
    cd->air_mass_in_condenser = new_air_mass;
    cd->air_partial_pressure = npd_clip(p_air, 0.0001, 0.005);
  }
  double steam_partial_pressure = npd_pymax(0.005, target_pressure - cd->air_partial_pressure);
  cd->condenser_pressure = steam_partial_pressure + cd->air_partial_pressure;
  if (n_running > 0) {
    double total_eff = 0.0;
    for (int e = 0; e < 2; e++)
      if ((cd->ej_operating_mask >> e) & 1) total_eff += cd->ej_nozzle_fouling[e] * cd->ej_diffuser_fouling[e] * cd->ej_nozzle_erosion[e];
    cd->vacuum_system_efficiency = total_eff / n_running;
  } else {
    cd->vacuum_system_efficiency = 0.95;
  }

  /* ---- calculate_enhanced_heat_transfer :564-728 */
  double sat_temp = npd_cond_tsat(steam_pressure);
  double h_g = npd_cond_hg(steam_pressure), h_f = npd_cond_hf(steam_pressure);
  double h_fg = h_g - h_f;
  double h_condensate = 4.18 * sat_temp;
  double q = npd_clip(steam_quality, 0.0, 1.0);
  double h_steam_inlet = h_f + q * h_fg;
  double heat_per_kg = h_steam_inlet - h_condensate;
  if (heat_per_kg <= 0) heat_per_kg = h_fg * q;
  double heat_available_watts = (steam_flow * heat_per_kg) * 1000;
  const double cp_water = 4180.0;
  double temp_rise_estimate = heat_available_watts / (cooling_water_flow * cp_water);
  double cooling_water_temp_out = cooling_water_temp_in + temp_rise_estimate;
  double lmtd = npd_cond_lmtd(sat_temp - cooling_water_temp_in, sat_temp - cooling_water_temp_out);
  double air_concentration = (cd->air_partial_pressure / npd_pymax(0.001, cd->condenser_pressure));
  double air_degradation_factor = 1.0 - 0.5 * air_concentration;
  double h_steam = 12000.0 * air_degradation_factor;
  double flow_factor = npd_powc(cooling_water_flow / 45000.0, 0.8);
  double h_water_base = 5000.0 * flow_factor;
  double h_water = h_water_base * npd_powc(pressure_drop_factor, 0.2);
  double r_steam = 1.0 / h_steam;
  double r_wall = 0.00159 / 385.0;
  double r_water = 1.0 / h_water;
  double overall_htc = 1.0 / (r_steam + cd->total_fouling_resistance + r_wall + r_water);
  if (diag) diag[0] = overall_htc;
  double effective_area = (75000.0 * area_factor);
  double theoretical_heat_transfer = overall_htc * effective_area * lmtd;
  double heat_transfer_rate = npd_pymin(heat_available_watts, theoretical_heat_transfer);
  if (heat_transfer_rate > heat_available_watts) heat_transfer_rate = heat_available_watts;
  double actual_temp_rise = heat_transfer_rate / (cooling_water_flow * cp_water);
  cd->cooling_water_outlet_temp = cooling_water_temp_in + actual_temp_rise;
  cd->heat_rejection_rate = heat_transfer_rate;
  res->heat_rejection_rate = heat_transfer_rate;
  res->condenser_pressure = cd->condenser_pressure;
  res->condensate_temperature = npd_cond_tsat(cd->condenser_pressure);
}

#endif
