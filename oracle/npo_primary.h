/*
 * npo_primary.h -- CPU oracle: primary side of the step.
 * TEST INFRASTRUCTURE ONLY (see npo_common.h).
 *
 * Follows PrimaryReactorPhysics.update_system  systems/primary/__init__.py:178-287
 * in the reference's order: actuators -> heat source -> thermal hydraulics ->
 * steam cycle -> state update -> NaN reset -> scram logic.
 */
#ifndef NPO_PRIMARY_H
#define NPO_PRIMARY_H
#include "npo_common.h"

/* _convert_action_to_control_inputs sim.py:260-288 + _apply_control_actions
 * systems/primary/__init__.py:289-359.  Actions 6,7,11..14 are accepted and do nothing. */
NPO_FN void npo_apply_control_actions(npb_prim_t *s, const npb_params_t *P, int action, double mag, double dt) {
  switch (action) {
    case 0: s->control_rod_position = npo_pymax(0.0, s->control_rod_position - P->max_control_rod_speed * dt * mag); break;
    case 1: s->control_rod_position = npo_pymin(100.0, s->control_rod_position + P->max_control_rod_speed * dt * mag); break;
    case 2: s->coolant_flow_rate = npo_pymin(50000.0, s->coolant_flow_rate + P->max_flow_change_rate * dt * mag); break;
    case 3: s->coolant_flow_rate = npo_pymax(5000.0, s->coolant_flow_rate - P->max_flow_change_rate * dt * mag); break;
    case 4: s->steam_valve_position = npo_pymin(100.0, s->steam_valve_position + P->max_valve_speed * dt * mag); break;
    case 5: s->steam_valve_position = npo_pymax(0.0, s->steam_valve_position - P->max_valve_speed * dt * mag); break;
    case 9: s->boron_concentration = npo_pymax(0.0, s->boron_concentration - 50.0 * dt * mag); break;
    case 10: s->boron_concentration = npo_pymin(3000.0, s->boron_concentration + 50.0 * dt * mag); break;
    default: break;
  }
}

/* ReactivityModel.update_fission_products  reactivity_model.py:313-367 (ReactorConfig :20-61) */
NPO_FN void npo_update_fission_products(npb_prim_t *s, double flux, double dt) {
  const double iodine_yield = 0.064, xenon_yield = 0.061, samarium_yield = 0.0137;
  const double iodine_decay_c = 2.87e-5, xenon_decay_c = 2.09e-5;
  const double sigma_a_xe135 = 2.65e6, sigma_a_sm149 = 4.1e4;
  double iodine = s->iodine_concentration, xenon = s->xenon_concentration, samarium = s->samarium_concentration;
  double fission_rate = flux * 1e-12;
  double iodine_production = iodine_yield * fission_rate;
  double iodine_decay = iodine_decay_c * iodine;
  double new_iodine = iodine + (iodine_production - iodine_decay) * dt;
  double xenon_production = xenon_yield * fission_rate;
  double xenon_from_iodine = iodine_decay_c * iodine;
  double xenon_decay = xenon_decay_c * xenon;
  double xenon_absorption = sigma_a_xe135 * 1e-24 * flux * xenon;
  double dxenon_dt = xenon_production + xenon_from_iodine - xenon_decay - xenon_absorption;
  double new_xenon = xenon + dxenon_dt * dt;
  double samarium_production = samarium_yield * fission_rate;
  double samarium_absorption = sigma_a_sm149 * 1e-24 * flux * samarium;
  double new_samarium = samarium + (samarium_production - samarium_absorption) * dt;
  s->iodine_concentration = npo_pymax(0.0, new_iodine);
  s->xenon_concentration = npo_pymax(0.0, new_xenon);
  s->samarium_concentration = npo_pymax(0.0, new_samarium);
}

/* ReactivityModel.calculate_total_reactivity  reactivity_model.py:77-125, terms :127-311.
 * Python's sum() over the dict adds in insertion order starting from int 0. */
NPO_FN double npo_total_reactivity_pcm(const npb_prim_t *s, double *components) {
  double pos_norm = npo_clip(s->control_rod_position / 100.0, 0.0, 1.0);
  double rods = 3000.0 * (pos_norm - 0.5);
  double boron = -10.0 * s->boron_concentration;
  double doppler = -2.5e-5 * (s->fuel_temperature - 575.0) * 1e5;
  double mod_temp = -3.0e-5 * (s->coolant_temperature - 280.0) * 1e5;
  double mod_void = -1000.0 * s->coolant_void_fraction;
  double pressure = 0.5 * (s->coolant_pressure - 15.5);
  double xenon = (s->xenon_concentration / 1.0e15) * -1800.0;
  double samarium = (s->samarium_concentration / 5.0e14) * -600.0;
  double depletion = 3340.0 + -0.15 * s->fuel_burnup;
  double bp = s->burnable_poison_worth * exp(-0.0002 * s->fuel_burnup);
  double total = 0.0;
  total += rods; total += boron; total += doppler; total += mod_temp; total += mod_void;
  total += pressure; total += xenon; total += samarium; total += depletion; total += bp;
  if (components) {   /* info["reactivity_components"], in the dict's order (NPB_RHO_* of include/npb.h) */
    components[NPB_RHO_CONTROL_RODS] = rods; components[NPB_RHO_BORON] = boron; components[NPB_RHO_DOPPLER] = doppler;
    components[NPB_RHO_MODERATOR_TEMP] = mod_temp; components[NPB_RHO_MODERATOR_VOID] = mod_void; components[NPB_RHO_PRESSURE] = pressure;
    components[NPB_RHO_XENON] = xenon; components[NPB_RHO_SAMARIUM] = samarium; components[NPB_RHO_FUEL_DEPLETION] = depletion;
    components[NPB_RHO_BURNABLE_POISONS] = bp;
  }
  return total;
}

/* ReactorHeatSource.update  heat_sources/reactor_heat_source.py:40-107 with
 * PointKineticsModel  physics/point_kinetics.py:26-131 */
NPO_FN void npo_reactor_heat_source(npb_prim_t *s, const npb_params_t *P, double dt,
                                    double *thermal_power_mw, double *power_percent, double *total_pcm, double *components) {
  const double BETA = 0.0065, LAMBDA_PROMPT = 1e-5;
  const double LAMBDA[6] = {0.077, 0.311, 1.40, 3.87, 1.40, 0.195};
  npo_update_fission_products(s, s->neutron_flux, dt);
  double total = npo_total_reactivity_pcm(s, components);
  double reactivity = total / 100000.0;
  if (s->scram_status) reactivity = -0.5;
  /* solve_point_kinetics */
  double rho = npo_clip(reactivity, -0.9, 0.1);
  if (P->kinetics_rk4_substeps > 0) {
    /* BASELINE config 2's "rk4" mode (no reference counterpart): dn/dt = (rho - beta) / Lambda * n + sum lambda_i C_i,
     * dC_i/dt = beta_i / Lambda * n - lambda_i C_i, fourth-order Runge-Kutta, kinetics_rk4_substeps sub-steps per dt, reactivity
     * held over the step as the reference holds it; no rate clips, the flux kept below the reference's ceiling.
     * The system is linear with constant coefficients over a step, y' = A y, so any Runge-Kutta method advances it by a
     * rational function of hA and only two things matter: the order on the slow (precursor) modes and what the function
     * does to the prompt mode a = (rho - beta) / Lambda, up to -9e4 / s (rho is clipped to [-0.9, 0.1]; a scram is -0.5).
     *   h a >= -2  classical explicit RK4 (R = the degree-4 Taylor polynomial, stable for h |a| < 2.78);
     *   h a <  -2  (an insertion below about -350 pcm at the 2-ms sub-step, every scram) explicit RK4 would amplify the prompt
     *              mode by up to 4e6 per sub-step, so the plant takes the L-stable singly diagonally implicit method of the same
     *              order (five stages, gamma = 1/4: Hairer & Wanner, Solving ODEs II, table IV.6.5), whose stability
     *              function is R(z) = P(z) / (1 - z / 4)^5 with P = the degree-4 truncation of (1 - z / 4)^5 e^z
     *              = 1 - z/4 - z^2/8 + z^3/96 + 7 z^4/768: |R| < 1 on the whole left half plane, R(-inf) = 0.  A is an arrow
     *              matrix (the groups couple only through the flux), so (I - gamma h A) x = r is solved in closed form.
     * Both agree with the matrix exponential of the same system to 1e-12 after a step (tests/test_rk4_cpu.py, from +300 pcm down
     * to the clip at -90 000 pcm). */
    const int ns = P->kinetics_rk4_substeps;
    const double h = dt / ns, a = (rho - BETA) / LAMBDA_PROMPT, b = (BETA / 6) / LAMBDA_PROMPT;
    const double FLUX_CEILING = 1e14;   /* the reference's band (point_kinetics.py:100), applied per sub-step: a prompt-supercritical
                                         * plant saturates there instead of overflowing to inf - inf inside the step */
    double n = s->neutron_flux, C[6];
    for (int i = 0; i < 6; i++) C[i] = s->precursors[i];
    if (h * a < -2.0) {
      const double gh = 0.25 * h;
      double d[6], sld = 0.0;
      for (int i = 0; i < 6; i++) { d[i] = 1.0 / (1.0 + gh * LAMBDA[i]); sld += LAMBDA[i] * d[i]; }
      const double inv_pivot = 1.0 / (1.0 - gh * a - gh * gh * b * sld);   /* Schur complement of the arrow's head */
      const double PC[5] = {1.0, -1.0 / 4.0, -1.0 / 8.0, 1.0 / 96.0, 7.0 / 768.0};
      for (int it = 0; it < ns; it++) {
        /* w = P(hA) y by Horner */
        double wn = PC[4] * n, wc[6];
        for (int i = 0; i < 6; i++) wc[i] = PC[4] * C[i];
        for (int k = 3; k >= 0; k--) {
          double src = 0.0;
          for (int i = 0; i < 6; i++) src += LAMBDA[i] * wc[i];
          const double tn = h * (a * wn + src);
          for (int i = 0; i < 6; i++) wc[i] = PC[k] * C[i] + h * (b * wn - LAMBDA[i] * wc[i]);
          wn = PC[k] * n + tn;
        }
        /* y <- (I - gamma h A)^-5 w */
        for (int st = 0; st < 5; st++) {
          double r = wn;
          for (int i = 0; i < 6; i++) r += gh * (LAMBDA[i] * d[i] * wc[i]);
          wn = r * inv_pivot;
          for (int i = 0; i < 6; i++) wc[i] = (wc[i] + gh * b * wn) * d[i];
        }
        n = npo_pymin(wn, FLUX_CEILING);
        for (int i = 0; i < 6; i++) C[i] = wc[i];
      }
    } else {
      for (int it = 0; it < ns; it++) {
        double kn[4], kc[4][6], yn = n, yc[6];
        for (int i = 0; i < 6; i++) yc[i] = C[i];
        for (int st = 0; st < 4; st++) {
          double src = 0.0;
          for (int i = 0; i < 6; i++) src += LAMBDA[i] * yc[i];
          kn[st] = a * yn + src;
          for (int i = 0; i < 6; i++) kc[st][i] = b * yn - LAMBDA[i] * yc[i];
          const double w = (st == 2) ? h : 0.5 * h;
          if (st < 3) { yn = n + w * kn[st]; for (int i = 0; i < 6; i++) yc[i] = C[i] + w * kc[st][i]; }
        }
        n += h / 6.0 * (kn[0] + 2.0 * kn[1] + 2.0 * kn[2] + kn[3]);
        n = npo_pymin(n, FLUX_CEILING);
        for (int i = 0; i < 6; i++) C[i] += h / 6.0 * (kc[0][i] + 2.0 * kc[1][i] + 2.0 * kc[2][i] + kc[3][i]);
      }
    }
    s->neutron_flux = npo_clip(n, 1e8, 1e14);
    for (int i = 0; i < 6; i++) s->precursors[i] = npo_pymax(C[i], 0.0);
    double pp_rk = s->neutron_flux / 1e13;
    *thermal_power_mw = pp_rk * P->rated_power_mw;
    *power_percent = pp_rk * 100.0;
    s->power_level = *power_percent;
    s->reactivity = reactivity;
    *total_pcm = total;
    return;
  }
  double flux_dot = 0.0, prec_dot[6] = {0, 0, 0, 0, 0, 0};
  if (!(fabs(rho) < 0.01)) {
    double eff = rho; /* second <0.01 branch (:47-50) is unreachable */
    flux_dot = (eff - BETA) / LAMBDA_PROMPT * s->neutron_flux;
    for (int i = 0; i < 6; i++) flux_dot += LAMBDA[i] * s->precursors[i];
    double max_change;
    if (fabs(rho) < 0.0001) max_change = s->neutron_flux * 0.0001;
    else if (fabs(rho) < 0.001) max_change = s->neutron_flux * 0.001;
    else if (fabs(rho) < 0.01) max_change = s->neutron_flux * 0.01;
    else max_change = s->neutron_flux * 0.1;
    flux_dot = npo_clip(flux_dot, -max_change, max_change);
    for (int i = 0; i < 6; i++) {
      double beta_i = BETA / 6;
      prec_dot[i] = beta_i / LAMBDA_PROMPT * s->neutron_flux - LAMBDA[i] * s->precursors[i];
    }
  }
  s->neutron_flux += flux_dot * dt;
  s->neutron_flux = npo_clip(s->neutron_flux, 1e8, 1e14);
  for (int i = 0; i < 6; i++) {
    s->precursors[i] += prec_dot[i] * dt;
    s->precursors[i] = npo_clip(s->precursors[i], 0.0, 1.0);
  }
  double pp = s->neutron_flux / 1e13;
  *thermal_power_mw = pp * P->rated_power_mw;
  *power_percent = pp * 100.0;
  s->power_level = *power_percent;
  s->reactivity = reactivity;
  *total_pcm = total;
}

/* ConstantHeatSource.update  heat_sources/constant_heat_source.py:104-137,169-183.
 * z is the standard-normal sample numpy's RandomState.normal(0, sigma) would have
 * scaled (legacy normal = loc + scale * gauss). */
NPO_FN void npo_constant_heat_source(npb_prim_t *s, const npb_params_t *P, double dt, double z,
                                     double *thermal_power_mw, double *power_percent) {
  double current_power_mw = (s->hs_setpoint_percent / 100.0) * P->rated_power_mw;
  double final_power = current_power_mw;
  if (P->hs_noise_enabled) {
    double noise_std_mw = (P->hs_noise_std_percent / 100.0) * current_power_mw;
    double raw = 0.0 + noise_std_mw * z;
    double alpha = dt / (P->hs_noise_filter_tau + dt);
    s->hs_filtered_noise_mw = alpha * raw + (1.0 - alpha) * s->hs_filtered_noise_mw;
    double noisy = current_power_mw + s->hs_filtered_noise_mw;
    final_power = npo_pymax(0.0, npo_pymin(noisy, P->rated_power_mw));
  }
  *thermal_power_mw = final_power;
  *power_percent = (final_power / P->rated_power_mw) * 100.0;
}

/* ThermalHydraulicsModel.calculate_heat_transfer_coefficient  thermal_hydraulics.py:111-166 */
NPO_FN double npo_core_ua(double coolant_flow_rate) {
  const double fuel_rod_diameter = 0.0095, fuel_rod_length = 3.66;
  const double num_fuel_rods = 50000;
  double heat_transfer_area = NPO_PI * fuel_rod_diameter * fuel_rod_length * num_fuel_rods;
  const double density = 700.0, viscosity = 9.0e-5, thermal_conductivity = 0.55, specific_heat = 5200.0;
  const double flow_area = 10.0;
  double velocity = coolant_flow_rate / (density * flow_area);
  double reynolds = density * velocity * fuel_rod_diameter / viscosity;
  reynolds = npo_pymax(reynolds, 1000.0);
  double prandtl = viscosity * specific_heat / thermal_conductivity;
  double nusselt = 0.023 * pow(reynolds, 0.8) * pow(prandtl, 0.4);
  double h = nusselt * thermal_conductivity / fuel_rod_diameter;
  double overall_ua = h * heat_transfer_area;
  overall_ua = overall_ua * 0.1;
  return npo_clip(overall_ua, 10e6, 50e6);
}

/* PrimaryReactorPhysics.update_system  systems/primary/__init__.py:178-287.
 * Returns scram_activated (True only on the firing step). nan_reset reports :247. */
NPO_FN int npo_primary_update(npb_prim_t *s, const npb_params_t *P, const npo_inputs_t *in, int *nan_reset, double *components) {
  const double dt = P->dt;
  const double FUEL_MASS = 200000.0, FUEL_HEAT_CAPACITY = 1500.0, COOLANT_HEAT_CAPACITY = 5200.0;
  (void)COOLANT_HEAT_CAPACITY;
  npo_apply_control_actions(s, P, in->action, in->magnitude, dt);

  double thermal_power_mw, power_percent, total_pcm = 0.0;
  if (P->heat_source == NPB_HEAT_REACTOR) {
    npo_reactor_heat_source(s, P, dt, &thermal_power_mw, &power_percent, &total_pcm, components);
    s->total_reactivity_pcm = total_pcm;
    s->reactivity = total_pcm / 100000.0; /* __init__.py:220 overwrites heat source's value */
  } else if (P->heat_source == NPB_HEAT_EXTERNAL) {
    /* a HeatSource plugin's result (heat_source_interface.py:23-112): the two columns the caller computed for this step */
    thermal_power_mw = in->noise_z;
    power_percent = isnan(in->power_setpoint) ? thermal_power_mw / P->rated_power_mw * 100.0 : in->power_setpoint;
    s->total_reactivity_pcm = 0.0;
  } else {
    npo_constant_heat_source(s, P, dt, in->noise_z, &thermal_power_mw, &power_percent);
    s->total_reactivity_pcm = 0.0;
  }
  s->thermal_power_mw = thermal_power_mw;
  s->power_level = power_percent;

  /* calculate_thermal_hydraulics  thermal_hydraulics.py:26-109 */
  double thermal_power = thermal_power_mw * 1e6;
  double heat_removal = npo_core_ua(s->coolant_flow_rate) * (s->fuel_temperature - s->coolant_temperature);
  double fuel_temp_dot = (thermal_power - heat_removal) / (FUEL_MASS * FUEL_HEAT_CAPACITY);
  int near_full = fabs(s->power_level - 100.0) < 5.0;
  fuel_temp_dot = near_full ? npo_clip(fuel_temp_dot, -1.0, 1.0) : npo_clip(fuel_temp_dot, -10.0, 10.0);
  double power_fraction = s->power_level / 100.0;
  double target_hot_leg_temp = 293.0 + (34.0 * power_fraction);
  double target_cold_leg_temp = 293.0;
  double target_avg_temp = (target_hot_leg_temp + target_cold_leg_temp) / 2.0;
  double temp_error = target_avg_temp - s->coolant_temperature;
  double coolant_temp_dot = 0.1 * temp_error;
  coolant_temp_dot = near_full ? npo_clip(coolant_temp_dot, -0.5, 0.5) : npo_clip(coolant_temp_dot, -5.0, 5.0);
  double temp_pressure_effect = 0.002 * (s->coolant_temperature - 293.0);
  double pressure_error = s->coolant_pressure - (15.5 + temp_pressure_effect);
  double pressure_dot = npo_clip(-0.01 * pressure_error, -0.05, 0.05);

  /* calculate_steam_cycle  :189-220 */
  double steam_generation = npo_pymin(s->coolant_flow_rate * 0.05, s->steam_valve_position / 100 * 2000);
  double steam_temp_dot = 0.1 * (s->coolant_temperature - s->steam_temperature);
  double steam_pressure_dot = 0.05 * (steam_generation - s->steam_flow_rate);
  double steam_flow_dot = s->steam_valve_position / 100 * 20 - 10;
  double feedwater_flow_dot = steam_generation - s->feedwater_flow_rate;

  /* update_thermal_state :168-187 */
  s->fuel_temperature = npo_clip(s->fuel_temperature + fuel_temp_dot * dt, 200.0, 2000.0);
  s->coolant_temperature = npo_clip(s->coolant_temperature + coolant_temp_dot * dt, 200.0, 400.0);
  s->coolant_pressure = npo_clip(s->coolant_pressure + pressure_dot * dt, 10.0, 20.0);
  /* update_steam_state :222-245 */
  s->steam_temperature = npo_clip(s->steam_temperature + steam_temp_dot * dt, 200.0, 400.0);
  s->steam_pressure = npo_clip(s->steam_pressure + steam_pressure_dot * dt, 1.0, 10.0);
  s->steam_flow_rate = npo_clip(s->steam_flow_rate + steam_flow_dot * dt, 0.0, 3000.0);
  s->feedwater_flow_rate = npo_clip(s->feedwater_flow_rate + feedwater_flow_dot * dt, 0.0, 3000.0);

  /* check_for_nan_values :247-270 */
  *nan_reset = 0;
  if (isnan(s->fuel_temperature) || isnan(s->neutron_flux) || isnan(s->coolant_temperature) ||
      isnan(s->coolant_pressure)) {
    s->neutron_flux = 1e12; s->fuel_temperature = 600.0; s->coolant_temperature = 280.0;
    s->coolant_pressure = 15.5; s->power_level = 100.0;
    *nan_reset = 1;
  }

  /* ScramSystem.check_safety_systems  safety/scram_logic.py:24-61 */
  int cond = (s->fuel_temperature > 1200.0) || (s->coolant_pressure > 17.2) ||
             (s->coolant_flow_rate < 5000.0) || (s->power_level > 118.0);
  if (cond && !s->scram_status) {
    s->scram_status = 1;
    s->control_rod_position = 0.0;
    return 1;
  }
  return 0;
}

/* per-loop primary conditions handed to the secondary side */
typedef struct npo_coupling_t {
  double inlet_temp[NPB_NUM_SG], outlet_temp[NPB_NUM_SG], flow[NPB_NUM_SG];
  double thermal_power[NPB_NUM_SG]; /* MW per loop */
} npo_coupling_t;

/* _calculate_primary_to_secondary_coupling  sim.py:335-427 */
NPO_FN void npo_primary_to_secondary(const npb_prim_t *s, npo_coupling_t *c) {
  double reactor_power_mw = s->power_level / 100.0 * 3000.0;
  double power_fraction = s->power_level / 100.0;
  const double design_flow = 17100.0;
  double flow_fraction = npo_pymax(0.3, power_fraction);
  double total_primary_flow = design_flow * flow_fraction;
  double cold_leg_temp = 293.0 + 2.0 * (power_fraction - 1.0);
  cold_leg_temp = npo_clip(cold_leg_temp, 285.0, 300.0);
  const double cp_primary = 5.2;
  double delta_t_core = (total_primary_flow > 0) ? (reactor_power_mw * 1000.0) / (total_primary_flow * cp_primary) : 0.0;
  double hot_leg_temp = cold_leg_temp + delta_t_core;
  hot_leg_temp = npo_clip(hot_leg_temp, cold_leg_temp + 5.0, 350.0);
  if (power_fraction < 0.1) hot_leg_temp = cold_leg_temp + 5.0;
  if (s->has_heat_removal_factor) {
    double effect = (s->last_heat_removal_factor - 1.0) * 3.0;
    cold_leg_temp += effect;
    cold_leg_temp = npo_clip(cold_leg_temp, 285.0, 300.0);
    hot_leg_temp = cold_leg_temp + delta_t_core;
    hot_leg_temp = npo_clip(hot_leg_temp, cold_leg_temp + 5.0, 350.0);
  }
  double flow_per_loop = total_primary_flow / 3;
  double thermal_power_per_loop = reactor_power_mw / 3;
  for (int i = 0; i < NPB_NUM_SG; i++) {
    double loop_variation = sin(i * 2.0) * 1.0;
    double loop_hot = hot_leg_temp + loop_variation;
    double loop_cold = cold_leg_temp + loop_variation * 0.5;
    loop_hot = npo_clip(loop_hot, loop_cold + 5.0, 350.0);
    loop_cold = npo_clip(loop_cold, 285.0, 300.0);
    c->inlet_temp[i] = loop_hot; c->outlet_temp[i] = loop_cold;
    c->flow[i] = flow_per_loop; c->thermal_power[i] = thermal_power_per_loop;
  }
}

#endif
