/*
 * npd_turbine.h -- device physics: turbine (14-stage expansion chain, rotor dynamics, 4 bearings,
 * metal-temperature tracker, protection trips, bearing lubrication system).
 *
 * Follows EnhancedTurbinePhysics.update_state  turbine/enhanced_physics.py:694-890 as wrapped at
 * construction by integrate_lubrication_with_turbine  turbine/turbine_bearing_lubrication.py:704-797.
 * dt is in HOURS here (secondary/__init__.py:568 passes dt/60).
 */
#ifndef NPD_TURBINE_H
#define NPD_TURBINE_H
#include "npd_common.h"
#include "npd_lube.h"

/* Antoine-form saturation temperature used by the turbine and condenser-side helpers
 * stage_system.py:458-466, enhanced_physics.py:1295-1303 */
NPD_FN double npd_tsat_antoine(double pressure_mpa) {
  if (pressure_mpa <= 0.001) return 10.0;
  const double A = 8.07131, B = 1730.63, C = 233.426;
  double pressure_bar = npd_clip(pressure_mpa * 10.0, 0.01, 100.0);
  double temp_c = B / (A - npd_log_pos(pressure_bar) * 4.34294481903251827651e-01) - C;   /* pressure_bar in [0.01, 100] (or NaN) */
  return npd_clip(temp_c, 10.0, 374.0);
}
/* _saturation_enthalpy_vapor  stage_system.py:468-473 */
NPD_FN double npd_hg_antoine(double pressure_mpa) {
  double temp = npd_tsat_antoine(pressure_mpa);
  double h_f = 4.18 * temp;
  double h_fg = 2257.0 * npd_powc_pos(1.0 - temp / 374.0, 0.38);   /* temp in [10, 51.6]: the Antoine form over [0.01, 100] bar */
  return h_f + h_fg;
}
/* same, from an already known saturation temperature */
NPD_FN double npd_hg_from_tsat(double temp) {
  double h_f = 4.18 * temp;
  double h_fg = 2257.0 * npd_powc_pos(1.0 - temp / 374.0, 0.38);
  return h_f + h_fg;
}
/* TurbineStage._steam_enthalpy  stage_system.py:418-443 */
NPD_FN double npd_stage_steam_enthalpy(double temp_c, double pressure_mpa) {
  pressure_mpa = npd_pymax(0.001, npd_pymin(pressure_mpa, 22.0));
  temp_c = npd_pymax(0.0, npd_pymin(temp_c, 800.0));
  double sat_temp = npd_tsat_antoine(pressure_mpa);
  if (temp_c <= sat_temp) return npd_hg_antoine(pressure_mpa);
  double h_g = npd_hg_antoine(pressure_mpa);
  double superheat = temp_c - sat_temp;
  double cp = (pressure_mpa > 10.0) ? 2.5 : ((pressure_mpa > 1.0) ? 2.2 : 2.0);
  return h_g + cp * superheat;
}
/* _enthalpy_to_temperature  stage_system.py:482-491 */
NPD_FN double npd_stage_enthalpy_to_temperature(double enthalpy, double pressure) {
  double sat_temp = npd_tsat_antoine(pressure);
  double h_g = npd_hg_antoine(pressure);
  if (enthalpy <= h_g) return sat_temp;
  return sat_temp + (enthalpy - h_g) / 2.1;
}

/* default stage table  TurbineStageSystem._create_default_stages  stage_system.py:683-756
 * (np.linspace(6.895, 1.2, 9) and np.linspace(1.15, 0.007, 7): y[i] = i*step + start, y[-1] = stop) */
NPD_FN void npd_stage_design(int k, double *p_in, double *p_out, double *design_flow, int *has_extraction, int *is_lp) {
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

typedef struct npd_stage_out_t {
  double power_output, outlet_pressure, outlet_temperature, outlet_enthalpy, outlet_flow, extraction_flow, loading_factor;
} npd_stage_out_t;

/* TurbineStage.calculate_stage_expansion  stage_system.py:98-292 */
NPD_FN void npd_stage_expansion(int k, double actual_efficiency, double blade_condition_factor, double fouling_factor,
                                double blade_wear_factor, double inlet_pressure, double inlet_temperature, double inlet_flow,
                                double outlet_pressure, double extraction_demand, npd_stage_out_t *o) {
  double d_in, d_out, design_flow; int has_extraction, is_lp;
  npd_stage_design(k, &d_in, &d_out, &design_flow, &has_extraction, &is_lp);
  double design_pressure_ratio = d_out / d_in;
  double load_factor = (design_flow > 0) ? inlet_flow / design_flow : 1.0;
  load_factor = npd_clip(load_factor, 0.3, 1.5);
  double temp_factor = (inlet_temperature + 273.15) / (285.8 + 273.15);
  temp_factor = npd_clip(temp_factor, 0.8, 1.2);
  double load_adjustment = 0.9 + 0.2 * load_factor;
  double temp_adjustment = 0.95 + 0.1 * (temp_factor - 1.0);
  double adjusted_pressure_ratio = design_pressure_ratio * load_adjustment * temp_adjustment;
  adjusted_pressure_ratio = npd_clip(adjusted_pressure_ratio, design_pressure_ratio * 0.85, design_pressure_ratio * 1.15);
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
  double inlet_enthalpy = npd_stage_steam_enthalpy(inlet_temperature, inlet_pressure);
  double extraction_flow = 0.0, extraction_enthalpy = 0.0;
  if (has_extraction && extraction_demand > 0) {
    extraction_flow = npd_clip(extraction_demand, 5.0, npd_pymin(50.0, inlet_flow * 0.3));
    double extraction_pressure = inlet_pressure * 0.7 + outlet_pressure * (1 - 0.7);
    double extraction_temp = npd_tsat_antoine(extraction_pressure);
    extraction_enthalpy = npd_stage_steam_enthalpy(extraction_temp, extraction_pressure);
  }
  double outlet_flow = inlet_flow - extraction_flow;
  double pr = self_outlet_pressure / inlet_pressure;
  double outlet_temp_isentropic = (inlet_temperature + 273.15) * npd_sqrt(npd_sqrt(pr)) - 273.15;
  double outlet_enthalpy_isentropic = npd_stage_steam_enthalpy(outlet_temp_isentropic, self_outlet_pressure);
  double quality_efficiency_factor = 1.0; /* steam_quality is the hard-coded 0.99 (:206) */
  double total_efficiency = (actual_efficiency * blade_condition_factor * fouling_factor * blade_wear_factor * quality_efficiency_factor);
  double isentropic_enthalpy_drop = inlet_enthalpy - outlet_enthalpy_isentropic;
  if (isentropic_enthalpy_drop <= 0) {
    double min_enthalpy_drop = 50.0 * (1.0 - self_outlet_pressure / inlet_pressure);
    isentropic_enthalpy_drop = npd_pymax(min_enthalpy_drop, 10.0);
  }
  double actual_enthalpy_drop = total_efficiency * isentropic_enthalpy_drop;
  if (actual_enthalpy_drop <= 0) actual_enthalpy_drop = npd_pymax(1.0, isentropic_enthalpy_drop * 0.5);
  double outlet_enthalpy = inlet_enthalpy - actual_enthalpy_drop;
  double outlet_temperature = npd_stage_enthalpy_to_temperature(outlet_enthalpy, outlet_pressure);
  double main_power = outlet_flow * actual_enthalpy_drop / 1000.0;
  if (main_power < 0) main_power = 0.0;
  double extraction_power = 0.0;
  if (extraction_flow > 0) extraction_power = extraction_flow * (inlet_enthalpy - extraction_enthalpy) / 1000.0;
  double design_enthalpy_drop = 0.88 * isentropic_enthalpy_drop;
  o->power_output = main_power + extraction_power;
  o->outlet_pressure = self_outlet_pressure; o->outlet_temperature = outlet_temperature;
  o->outlet_enthalpy = outlet_enthalpy; o->outlet_flow = outlet_flow; o->extraction_flow = extraction_flow;
  o->loading_factor = actual_enthalpy_drop / npd_pymax(1.0, design_enthalpy_drop);
}

/* get_dynamic_pressure_ratio closure  stage_system.py:794-868 */
NPD_FN double npd_stage_dynamic_pressure_ratio(int k, double current_pressure, double inlet_flow) {
  double d_in, d_out, design_flow; int has_extraction, is_lp;
  npd_stage_design(k, &d_in, &d_out, &design_flow, &has_extraction, &is_lp);
  double design_pressure_ratio = d_out / d_in;
  double load_factor = (design_flow > 0) ? inlet_flow / design_flow : 1.0;
  load_factor = npd_clip(load_factor, 0.3, 1.5);
  double pressure_factor = (d_in > 0) ? current_pressure / d_in : 1.0;
  pressure_factor = npd_clip(pressure_factor, 0.5, 1.5);
  double load_adjustment = 0.90 + 0.2 * (load_factor - 1.0);
  double pressure_adjustment = 0.95 + 0.1 * (pressure_factor - 1.0);
  double dynamic_ratio = design_pressure_ratio * load_adjustment * pressure_adjustment;
  double min_ratio, max_ratio;
  if (!is_lp) { min_ratio = 0.70; max_ratio = 0.95; } else { min_ratio = 0.50; max_ratio = 0.85; }
  if (is_lp) {
    int remaining_stages = 14 - k - 1;
    if (remaining_stages > 0) {
      /* 0.85 ** remaining_stages (remaining_stages = 1..5 for LP-1..LP-5), correctly rounded */
      const double pow085[6] = {1.0, 0.85, 0.7224999999999999, 0.6141249999999999, 0.5220062499999999, 0.44370531249999995};
      double min_outlet_pressure = 0.007 / pow085[remaining_stages];
      double max_allowable_ratio = min_outlet_pressure / current_pressure;
      min_ratio = npd_pymax(min_ratio, max_allowable_ratio);
    }
  }
  if (k == 13) dynamic_ratio = npd_pymax(0.007 / current_pressure, 0.05);
  else dynamic_ratio = npd_clip(dynamic_ratio, min_ratio, max_ratio);
  return dynamic_ratio;
}

typedef struct npd_stagesys_out_t {
  double total_power, total_extraction, lp6_outlet_enthalpy;
  double hp_power, lp_power, overall_efficiency;   /* enhanced_physics.py:879-880, stage_system.py:983-993 (info only) */
  double max_temp_rate, max_thermal_stress; /* MetalTemperatureTracker reductions */
  double prev_rotor, prev_casing, max_gradient;   /* diagnostics build only: the tracker's largest rotor / casing gradient, enhanced_physics.py:128-135 */
} npd_stagesys_out_t;

/* per-stage column access: the stage / tracker arrays are read from the LDS staging region (the whole
 * tstg section is LDS-DMA'd there while the lubrication step and passes A/B run, npd_stage.h) and each
 * updated value is written straight to its SoA column; every column is read before it is written and
 * never re-read within a step, so the staged copy does not need the update */
#define NPD_TSTG_RD(member, k) NPD_LDS_REAL(0, NPB_F64_SLOT(npb_tstg_t, member) + (k))
#define NPD_TSTG_WR(member, k, v) npd_store_real<SM>(st, NPD_SEC_COL(TSTG, 0) + NPB_F64_SLOT(npb_tstg_t, member) + (k), (v))

/* one stage's share of TurbineStage.update_degradation (stage_system.py:294-339) and of
 * MetalTemperatureTracker.update_temperatures (enhanced_physics.py:73-166, time constant 1 h, ambient 25 C);
 * both only touch stage k's own state, so running them right after stage k's expansion is the
 * reference's result */
template <int SM = 0>
NPD_FN void npd_stage_post(const npd_stage_t &st, const double *stg, int k, double loading_factor,
                           double outlet_temperature, double dt, npd_stagesys_out_t *out) {
  NPD_TSTG_WR(stage_efficiency_degradation, k, NPD_TSTG_RD(stage_efficiency_degradation, k) + 1e-05 * dt);
  NPD_TSTG_WR(stage_deposit_thickness, k, NPD_TSTG_RD(stage_deposit_thickness, k) + 5e-05 * dt);
  double blade_wear = (1e-06 * dt) * npd_sq(loading_factor);
  NPD_TSTG_WR(stage_blade_wear_factor, k, npd_pymax(0.7, NPD_TSTG_RD(stage_blade_wear_factor, k) - blade_wear));
  const double time_constant = 3600.0 / 3600.0, ambient = 25.0;
  if (k < 8) {
    double rt = NPD_TSTG_RD(rotor_temperatures, k);
    double tc = ((outlet_temperature - 50.0) - rt) / time_constant * dt;
    double max_rate = 5.0 * dt;
    tc = npd_clip(tc, -max_rate, max_rate);
    rt += tc;
    NPD_TSTG_WR(rotor_temperatures, k, rt);
    double rate = fabs(tc / dt * 60.0);
    out->max_temp_rate = (k == 0) ? rate : npd_pymax(out->max_temp_rate, rate);
    double stress = (1.2e-05 * (rt - ambient)) * 200000000000.0 * 0.1;
    out->max_thermal_stress = (k == 0) ? stress : npd_pymax(out->max_thermal_stress, stress);
    if (st.diag) {   /* rotor gradient between neighbouring points 1 m apart [C/cm] */
      if (k > 0) { const double g = fabs(rt - out->prev_rotor) / (1.0 * 100); out->max_gradient = (k == 1) ? g : npd_pymax(out->max_gradient, g); }
      out->prev_rotor = rt;
    }
  }
  if (k < 6) {
    double ct = NPD_TSTG_RD(casing_temperatures, k);
    double tc = ((outlet_temperature - 80.0) - ct) / time_constant * dt;
    tc = npd_clip(tc, -3.0 * dt, 3.0 * dt);
    NPD_TSTG_WR(casing_temperatures, k, ct + tc);
    if (st.diag) {   /* casing points 1.5 m apart; max(max(rotor), max(casing)) taken at the end (stage 5 comes after every rotor pair but the last two) */
      if (k > 0) out->max_gradient = npd_pymax(out->max_gradient, fabs((ct + tc) - out->prev_casing) / (1.5 * 100));
      out->prev_casing = ct + tc;
    }
  }
  {
    double bt = NPD_TSTG_RD(blade_temperatures, k);
    double tc = ((outlet_temperature - 20.0) - bt) / (time_constant * 0.5) * dt;
    tc = npd_clip(tc, -10.0 * dt, 10.0 * dt);
    NPD_TSTG_WR(blade_temperatures, k, bt + tc);
  }
}

/* requested outlet pressure of stage k  calculate_stage_by_stage_expansion  stage_system.py:870-895 */
NPD_FN double npd_stage_requested_outlet(int k, double current_pressure, double inlet_flow) {
  const double final_pressure = 0.007;
  double pressure_ratio = npd_stage_dynamic_pressure_ratio(k, current_pressure, inlet_flow);
  double outlet_pressure = current_pressure * pressure_ratio;
  outlet_pressure = npd_pymax(outlet_pressure, final_pressure);
  int remaining_stages = 14 - k - 1;
  if (remaining_stages == 0) outlet_pressure = final_pressure;
  else if (remaining_stages == 1) outlet_pressure = npd_pymax(outlet_pressure, final_pressure / 0.5);
  if (outlet_pressure >= current_pressure) {
    outlet_pressure = current_pressure * 0.95;
    outlet_pressure = npd_pymax(outlet_pressure, final_pressure);
  }
  return outlet_pressure;
}

/* TurbineStageSystem.update_state  stage_system.py:928-1016, reference order, one stage at a time.
 * Exact for every input; used when a lane of the wave leaves the fast path's assumptions. */
template <int SM = 0>
NPD_FN void npd_stage_system_update_seq(const npd_stage_t &st, const double *stg, double inlet_pressure,
                                        double inlet_temperature, double inlet_flow, double load_demand,
                                        double pressure_stability_factor, double dt, npd_stagesys_out_t *out) {
  double current_pressure = inlet_pressure, current_temperature = inlet_temperature, current_flow = inlet_flow;
  double total_power = 0.0, total_extraction = 0.0, hp_power = 0.0, lp_power = 0.0;
  NPD_DMA_WAIT(); /* the staged stage arrays are read from here on */
#pragma unroll 1
  for (int k = 0; k < 14; k++) {
    double extraction_demand = (k == 2) ? 25.0 * load_demand : (k == 3) ? 30.0 * load_demand : (k == 4) ? 20.0 * load_demand
                             : (k == 8) ? 15.0 * load_demand : (k == 9) ? 10.0 * load_demand : 0.0;
    double outlet_pressure = npd_stage_requested_outlet(k, current_pressure, inlet_flow);
    double fouling_factor = 1.0 / (1.0 + NPD_TSTG_RD(stage_deposit_thickness, k) / 0.5);
    double blade_wear_factor = NPD_TSTG_RD(stage_blade_wear_factor, k);
    double blade_condition_factor = npd_pymin(fouling_factor, blade_wear_factor);
    double actual_efficiency = npd_pymax(0.7, 0.88 - NPD_TSTG_RD(stage_efficiency_degradation, k));
    npd_stage_out_t so;
    npd_stage_expansion(k, actual_efficiency, blade_condition_factor, fouling_factor, blade_wear_factor, current_pressure,
                        current_temperature, current_flow, outlet_pressure, extraction_demand, &so);
    total_power += so.power_output; total_extraction += so.extraction_flow;
    if (k < 8) hp_power += so.power_output; else lp_power += so.power_output;
    if (k == 13) out->lp6_outlet_enthalpy = so.outlet_enthalpy;
    NPD_DIAG(st, NPB_DIAG_STAGE_INLET_PRESSURE + k, current_pressure); NPD_DIAG(st, NPB_DIAG_STAGE_INLET_TEMPERATURE + k, current_temperature);
    NPD_DIAG(st, NPB_DIAG_STAGE_OUTLET_PRESSURE + k, so.outlet_pressure); NPD_DIAG(st, NPB_DIAG_STAGE_OUTLET_TEMPERATURE + k, so.outlet_temperature);
    NPD_DIAG(st, NPB_DIAG_STAGE_POWER_OUTPUT + k, so.power_output); NPD_DIAG(st, NPB_DIAG_STAGE_LOADING_FACTOR + k, so.loading_factor);
    NPD_DIAG(st, NPB_DIAG_STAGE_EXTRACTION_FLOW + k, so.extraction_flow);
    npd_stage_post<SM>(st, stg, k, so.loading_factor, so.outlet_temperature, dt, out);
    current_pressure = so.outlet_pressure; current_temperature = so.outlet_temperature; current_flow = so.outlet_flow;
  }
  out->total_power = total_power * pressure_stability_factor;
  out->total_extraction = total_extraction;
  out->hp_power = hp_power; out->lp_power = lp_power;
  out->overall_efficiency = 0.0;
  if (inlet_flow > 0) {
    const double h_in = npd_stage_steam_enthalpy(inlet_temperature, inlet_pressure);
    out->overall_efficiency = (h_in - npd_stage_steam_enthalpy(current_temperature, current_pressure)) / h_in;
  }
}

/* Same result, restructured for instruction-level parallelism (one wave per SIMD has nothing else to
 * hide latency with): the pressure / flow chain does not depend on the temperature chain unless a stage
 * takes the "invalid pressure ratio" branch (stage_system.py:146-155), so
 *   pass A  walks the 14 stages' pressures and flows (no transcendentals),
 *   pass B  evaluates saturation temperature / vapour enthalpy for all 15 + 5 distinct pressures and
 *           the 14 isentropic temperature ratios as independent, interleavable streams,
 *   pass C  walks the temperature / enthalpy chain with plain arithmetic and streams each stage's
 *           degradation and metal-temperature state.
 * Lanes that would take a rare branch make the whole wave use npd_stage_system_update_seq. */
template <int SM = 0>
NPD_FN void npd_stage_system_update(const npd_stage_t &st, const double *stg, double inlet_pressure,
                                    double inlet_temperature, double inlet_flow, double load_demand,
                                    double pressure_stability_factor, double dt, npd_stagesys_out_t *out) {
  /* extraction stages 2, 3, 4, 8, 9 -> compact index 0..4 */
#define NPD_EXT_IDX(k) ((k) == 2 ? 0 : (k) == 3 ? 1 : (k) == 4 ? 2 : (k) == 8 ? 3 : 4)
#define NPD_IS_EXT(k) ((k) == 2 || (k) == 3 || (k) == 4 || (k) == 8 || (k) == 9)
  double p_self[14], flow_out[14], p_ext[5], ext_flow[5];
  bool rare = !(inlet_pressure >= 0.001 && inlet_pressure <= 22.0);
  {
    double cur_p = inlet_pressure, cur_flow = inlet_flow;
#pragma unroll
    for (int k = 0; k < 14; k++) {
      double d_in, d_out, design_flow; int has_extraction, is_lp;
      npd_stage_design(k, &d_in, &d_out, &design_flow, &has_extraction, &is_lp);
      double design_pressure_ratio = d_out / d_in;
      double extraction_demand = (k == 2) ? 25.0 * load_demand : (k == 3) ? 30.0 * load_demand : (k == 4) ? 20.0 * load_demand
                               : (k == 8) ? 15.0 * load_demand : (k == 9) ? 10.0 * load_demand : 0.0;
      double outlet_pressure = npd_stage_requested_outlet(k, cur_p, inlet_flow);
      rare = rare || (outlet_pressure >= cur_p);
      double min_allowed, max_allowed;
      if (k == 13) { min_allowed = 0.002; max_allowed = 0.009; }
      else { min_allowed = cur_p * (design_pressure_ratio * 0.7); max_allowed = cur_p * (design_pressure_ratio * 1.3); }
      double self_out = (outlet_pressure < min_allowed) ? min_allowed : ((outlet_pressure > max_allowed) ? max_allowed : outlet_pressure);
      /* a clamped request makes the stage's own outlet state differ from the one handed to the next stage
       * (stage_system.py:146-155 vs :915): left to the sequential path */
      rare = rare || (self_out != outlet_pressure);
      double ef = 0.0, pe = cur_p;
      if (has_extraction && extraction_demand > 0) {
        ef = npd_clip(extraction_demand, 5.0, npd_pymin(50.0, cur_flow * 0.3));
        pe = cur_p * 0.7 + outlet_pressure * (1 - 0.7);
      }
      if (NPD_IS_EXT(k)) { ext_flow[NPD_EXT_IDX(k)] = ef; p_ext[NPD_EXT_IDX(k)] = pe; }
      p_self[k] = self_out;
      flow_out[k] = cur_flow - ef;
      rare = rare || !(self_out >= 0.001 && self_out <= 22.0) || !(pe >= 0.001 && pe <= 22.0) || !(outlet_pressure >= 0.001);
      cur_p = self_out; cur_flow = flow_out[k];
    }
  }
  if (__builtin_amdgcn_ballot_w64(rare) != 0) { /* wave-uniform: any lane off the fast path */
    npd_stage_system_update_seq<SM>(st, stg, inlet_pressure, inlet_temperature, inlet_flow, load_demand,
                                pressure_stability_factor, dt, out);
    return;
  }
  NPD_STAMP(13);
  /* pass B: independent transcendental streams */
  double sat_self[14], hg_self[14], tratio[14], hg_ext[5];
  double sat_in0 = npd_tsat_antoine(inlet_pressure);
  double hg_in0 = npd_hg_from_tsat(sat_in0);
#pragma unroll
  for (int k = 0; k < 14; k++) {
    sat_self[k] = npd_tsat_antoine(p_self[k]);
    hg_self[k] = npd_hg_from_tsat(sat_self[k]);
    tratio[k] = npd_sqrt(npd_sqrt(p_self[k] / ((k == 0) ? inlet_pressure : p_self[k > 0 ? k - 1 : 0])));
  }
#pragma unroll
  for (int e = 0; e < 5; e++) hg_ext[e] = npd_hg_from_tsat(npd_tsat_antoine(p_ext[e]));
  NPD_STAMP(14);
  NPD_DMA_WAIT(); /* the staged stage arrays are read from here on */
  /* pass C: temperature / enthalpy chain */
  double T_in = inlet_temperature, sat_in = sat_in0, hg_in = hg_in0;
  double total_power = 0.0, total_extraction = 0.0, hp_power = 0.0, lp_power = 0.0, h_in0 = 0.0;
#pragma unroll
  for (int k = 0; k < 14; k++) {
    const double p_in = (k == 0) ? inlet_pressure : p_self[k > 0 ? k - 1 : 0];
    if (k == 1) NPD_STAMP(24);
    if (k == 2) NPD_STAMP(25);
    if (k == 7) NPD_STAMP(26);
    if (k == 13) NPD_STAMP(27);
    double fouling_factor = 1.0 / (1.0 + NPD_TSTG_RD(stage_deposit_thickness, k) / 0.5);
    double blade_wear_factor = NPD_TSTG_RD(stage_blade_wear_factor, k);
    double blade_condition_factor = npd_pymin(fouling_factor, blade_wear_factor);
    double actual_efficiency = npd_pymax(0.7, 0.88 - NPD_TSTG_RD(stage_efficiency_degradation, k));
    double cp_in = (p_in > 10.0) ? 2.5 : ((p_in > 1.0) ? 2.2 : 2.0);
    double T_c = npd_pymax(0.0, npd_pymin(T_in, 800.0));
    double inlet_enthalpy = (T_c <= sat_in) ? hg_in : hg_in + cp_in * (T_c - sat_in);
    double T_isen = (T_in + 273.15) * tratio[k] - 273.15;
    double T_isen_c = npd_pymax(0.0, npd_pymin(T_isen, 800.0));
    double cp_out = (p_self[k] > 10.0) ? 2.5 : ((p_self[k] > 1.0) ? 2.2 : 2.0);
    double h_isen = (T_isen_c <= sat_self[k]) ? hg_self[k] : hg_self[k] + cp_out * (T_isen_c - sat_self[k]);
    double total_efficiency = (actual_efficiency * blade_condition_factor * fouling_factor * blade_wear_factor * 1.0);
    double isentropic_enthalpy_drop = inlet_enthalpy - h_isen;
    if (isentropic_enthalpy_drop <= 0) {
      double min_enthalpy_drop = 50.0 * (1.0 - p_self[k] / p_in);
      isentropic_enthalpy_drop = npd_pymax(min_enthalpy_drop, 10.0);
    }
    double actual_enthalpy_drop = total_efficiency * isentropic_enthalpy_drop;
    if (actual_enthalpy_drop <= 0) actual_enthalpy_drop = npd_pymax(1.0, isentropic_enthalpy_drop * 0.5);
    double outlet_enthalpy = inlet_enthalpy - actual_enthalpy_drop;
    /* requested outlet pressure == the stage's own outlet pressure on this path */
    double T_out = (outlet_enthalpy <= hg_self[k]) ? sat_self[k] : sat_self[k] + (outlet_enthalpy - hg_self[k]) / 2.1;
    double main_power = flow_out[k] * actual_enthalpy_drop / 1000.0;
    if (main_power < 0) main_power = 0.0;
    double extraction_power = 0.0;
    double ef = NPD_IS_EXT(k) ? ext_flow[NPD_EXT_IDX(k)] : 0.0;
    if (ef > 0) extraction_power = ef * (inlet_enthalpy - hg_ext[NPD_EXT_IDX(k)]) / 1000.0;
    double loading_factor = actual_enthalpy_drop / npd_pymax(1.0, 0.88 * isentropic_enthalpy_drop);
    total_power += main_power + extraction_power; total_extraction += ef;
    if (k < 8) hp_power += main_power + extraction_power; else lp_power += main_power + extraction_power;
    if (k == 0) h_in0 = inlet_enthalpy;   /* = _steam_enthalpy(inlet_temperature, inlet_pressure) of stage_system.py:985 */
    if (k == 13) out->lp6_outlet_enthalpy = outlet_enthalpy;
    NPD_DIAG(st, NPB_DIAG_STAGE_INLET_PRESSURE + k, p_in); NPD_DIAG(st, NPB_DIAG_STAGE_INLET_TEMPERATURE + k, T_in);
    NPD_DIAG(st, NPB_DIAG_STAGE_OUTLET_PRESSURE + k, p_self[k]); NPD_DIAG(st, NPB_DIAG_STAGE_OUTLET_TEMPERATURE + k, T_out);
    NPD_DIAG(st, NPB_DIAG_STAGE_POWER_OUTPUT + k, main_power + extraction_power); NPD_DIAG(st, NPB_DIAG_STAGE_LOADING_FACTOR + k, loading_factor);
    NPD_DIAG(st, NPB_DIAG_STAGE_EXTRACTION_FLOW + k, ef);
    npd_stage_post<SM>(st, stg, k, loading_factor, T_out, dt, out);
    T_in = T_out; sat_in = sat_self[k]; hg_in = hg_self[k];
  }
  NPD_STAMP(28);
  out->total_power = total_power * pressure_stability_factor;
  out->total_extraction = total_extraction;
  out->hp_power = hp_power; out->lp_power = lp_power;
  {   /* stage_system.py:983-993: _steam_enthalpy at the last stage's outlet, whose saturation state pass B already has */
    const double T_c = npd_pymax(0.0, npd_pymin(T_in, 800.0));
    const double cp = (p_self[13] > 10.0) ? 2.5 : ((p_self[13] > 1.0) ? 2.2 : 2.0);
    const double h_out = (T_c <= sat_in) ? hg_in : hg_in + cp * (T_c - sat_in);
    out->overall_efficiency = (inlet_flow > 0) ? (h_in0 - h_out) / h_in0 : 0.0;
  }
#undef NPD_EXT_IDX
#undef NPD_IS_EXT
}

/* _calculate_pressure_variation_effects  enhanced_physics.py:1312-1350 */
NPD_FN double npd_pressure_stability_factor(const double *sg_pressures) {
  double avg_pressure = (0.0 + sg_pressures[0] + sg_pressures[1] + sg_pressures[2]) / 3;
  double max_deviation = fabs(sg_pressures[0] - avg_pressure);
  for (int i = 1; i < 3; i++) max_deviation = npd_pymax(max_deviation, fabs(sg_pressures[i] - avg_pressure));
  double variation_factor = max_deviation / 0.1;
  double stability_factor;
  if (max_deviation < 0.02) stability_factor = 1.0;
  else if (max_deviation < 0.05) stability_factor = 1.0 - (max_deviation - 0.02) / 0.03 * 0.05;
  else stability_factor = 0.95 - npd_pymin(variation_factor - 0.5, 0.25);
  return npd_clip(stability_factor, 0.7, 1.0);
}

/* LubricationComponent tables  turbine_bearing_lubrication.py:97-173 */
static __device__ const double NPD_TLUB_BASE[5] = {0.0003, 0.0004, 0.0006, 0.0008, 0.0001};
static __device__ const double NPD_TLUB_LOAD_EXP[5] = {1.8, 1.6, 2.5, 1.4, 1.0};
static __device__ const double NPD_TLUB_SPEED_EXP[5] = {1.5, 1.5, 1.3, 1.1, 0.5};
static __device__ const double NPD_TLUB_CONTAM[5] = {3.0, 2.8, 4.0, 3.5, 1.5};
static __device__ const double NPD_TLUB_OIL_FLOW[4] = {25.0, 30.0, 40.0, 15.0}; /* oil_flow_requirement injected into TB-001..004 */

typedef struct npd_turbine_result_t {
  double electrical_power_net, electrical_power_gross, mechanical_power, effective_steam_flow;
  double condenser_pressure, condenser_temperature, lp6_outlet_enthalpy;
  double hp_power, lp_power, overall_efficiency;
  int trip_active;
} npd_turbine_result_t;

/* ================= lubrication wrapper (pre-step, previous step's bearing state) =============
 * update_with_lubrication  turbine_bearing_lubrication.py:715-784: called with keyword arguments
 * only, so every rotor quantity takes its default and the load factor is turbine.load_demand
 * as left by the PREVIOUS step (:744-746).  Reads rotor_speed, bearing_load, bearing_metal_temp and load_demand
 * as the previous step left them; writes the lub_* members only. */
NPD_FN void npd_turbine_lube(npb_turb_t *t, double dt, double *oil_temps_out = nullptr) {
  double lub_load_factor = t->load_demand;
  double friction_heat[4], b_load_factor[4], b_speed_factor, b_temperature[4];
  b_speed_factor = t->rotor_speed / 3600.0;
  double total_heat_generation = 0.0;
  for (int i = 0; i < 4; i++) { /* collect_bearing_states :800-838, calculate_bearing_friction_heat :841-863 */
    double load_n = t->bearing_load[i] * 1000.0;
    double clearance_m = 0.15 / 1000.0;
    double omega = t->rotor_speed * 2 * NPD_PI / 60.0;
    double friction_power = load_n * 0.001 * omega * clearance_m;
    friction_heat[i] = npd_pymax(0.0, friction_power);
    b_load_factor[i] = t->bearing_load[i] / npd_pymax(500.0, 1.0);
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
  if (oil_temps_out) {   /* state-log diagnostics: calculate_component_oil_temperatures :967-1072, the oil temperature each bearing is handed (:1103-1120) and logs as its own */
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const double offset = (i == 0) ? 2.0 : (i == 2) ? 1.0 : (i == 3) ? -5.0 : 0.0;
      double temp = system_oil_temp + offset;
      if (NPD_TLUB_OIL_FLOW[i] > 0 && friction_heat[i] > 0) {
        const double mass_flow = NPD_TLUB_OIL_FLOW[i] / 60.0 * 0.85;
        const double temp_rise_basic = friction_heat[i] / (mass_flow * 2000.0);
        const double heat_dissipated = 50.0 * 0.5 * npd_pymax(0.0, temp_rise_basic);
        const double net_heat = npd_pymax(0.0, friction_heat[i] - heat_dissipated);
        double temp_rise = (net_heat > 0) ? net_heat / (mass_flow * 2000.0) : 0.0;
        const double max_temp_rise = (i == 0) ? 25.0 : (i == 3) ? 15.0 : 20.0;
        temp_rise = npd_pymin(max_temp_rise, npd_pymax(0.0, temp_rise));
        temp = system_oil_temp + temp_rise;
        temp += offset;
      }
      oil_temps_out[i] = npd_pymax(35.0, npd_pymin(70.0, temp));
    }
  }
  double contamination_input = lub_load_factor * 0.02 + (1.0 - 0.99) * 0.5;
  double moisture_input = (1.0 - 0.99) * 0.01;
  double avg_wear = 0.0;
  for (int i = 0; i < 5; i++) avg_wear += t->lub_wear[i];
  avg_wear = avg_wear / 5;
  npd_oil_t oil = {t->lub_oil_temperature, t->lub_oil_contamination, t->lub_oil_moisture, t->lub_oil_acidity,
                   t->lub_oil_viscosity_change, t->lub_antioxidant_level, t->lub_anti_wear_level,
                   t->lub_corrosion_inhibitor_level, t->lub_effectiveness};
  const npd_oil_limits_t lim = {8.0, 0.3, 0.05, 20.0};
  npd_update_oil_quality(&oil, &lim, avg_wear, system_oil_temp, contamination_input, moisture_input, dt);
  t->lub_oil_temperature = oil.temperature; t->lub_oil_contamination = oil.contamination; t->lub_oil_moisture = oil.moisture;
  t->lub_oil_acidity = oil.acidity; t->lub_oil_viscosity_change = oil.viscosity_change;
  t->lub_antioxidant_level = oil.antioxidant; t->lub_anti_wear_level = oil.anti_wear;
  t->lub_corrosion_inhibitor_level = oil.corrosion_inhibitor; t->lub_effectiveness = oil.effectiveness;
  /* update_component_wear with calculate_component_wear :261-333; map_bearing_to_lubrication_components :931-964 */
  const double l_bspeed = npd_log(b_speed_factor); /* shared base of the three bearing wear-rate powers */
  for (int i = 0; i < 5; i++) {
    double wear_rate;
    if (i == 0) {
      double steam_temp_factor = npd_pymax(1.0, (b_temperature[0] - 70.0) / 20.0);
      double load_factor_adj = b_load_factor[0] * 1.2;
      wear_rate = (NPD_TLUB_BASE[0] * npd_pow_logs(npd_log(load_factor_adj), NPD_TLUB_LOAD_EXP[0], l_bspeed, NPD_TLUB_SPEED_EXP[0]) * steam_temp_factor);
    } else if (i == 1) {
      double moisture_factor = npd_pymax(1.0, (1.0 - 0.99) * 10.0);
      double temp_factor = npd_pymax(1.0, (b_temperature[1] - 60.0) / 25.0);
      wear_rate = (NPD_TLUB_BASE[1] * npd_pow_logs(npd_log(b_load_factor[1]), NPD_TLUB_LOAD_EXP[1], l_bspeed, NPD_TLUB_SPEED_EXP[1]) * moisture_factor * temp_factor);
    } else if (i == 2) {
      double axial_load_factor = b_load_factor[2] * 1.0;
      double temp_factor = npd_pymax(1.0, (b_temperature[2] - 50.0) / 30.0);
      wear_rate = (NPD_TLUB_BASE[2] * npd_pow_logs(npd_log(axial_load_factor), NPD_TLUB_LOAD_EXP[2], l_bspeed, NPD_TLUB_SPEED_EXP[2]) * temp_factor);
    } else if (i == 3) {
      double contamination_factor = 1.0 + t->lub_oil_contamination / 10.0;
      wear_rate = (NPD_TLUB_BASE[3] * 1.0 * contamination_factor);
    } else {
      wear_rate = (NPD_TLUB_BASE[4] * 1.0 * 1.0); /* oil_coolers: no bearing maps to it -> defaults */
    }
    double lubrication_wear_factor = 1.0 + (1.0 - t->lub_effectiveness) * NPD_TLUB_CONTAM[i];
    t->lub_wear[i] += (wear_rate * lubrication_wear_factor) * dt;
  }

}

/* RotorDynamicsModel.update_state  rotor_dynamics.py:956-1070 (rotor, thermal effects, four bearings, vibration
 * response); returns the hottest bearing metal temperature and the total vibration displacement for the protection */
NPD_FN void npd_turbine_rotor(npb_turb_t *t, double stage_power_mw, double steam_temperature, double load_demand, double dt,
                              double *max_bearing_metal_out, double *total_displacement_out, double *torques_out = nullptr) {
  double applied_torque = stage_power_mw * 1e6 / (2 * NPD_PI * 3600 / 60);

  /* ---- RotorDynamicsModel.update_state  rotor_dynamics.py:956-1070 */
  /* calculate_rotor_dynamics :855-911 */
  double dt_seconds = dt * 3600.0;
  double total_friction = 0.0;
  for (int i = 0; i < 4; i++) total_friction += (t->bearing_load[i] * 1000.0 * 0.001 * 0.15 / 1000.0);
  double net_torque = applied_torque - total_friction;
  double angular_acceleration = net_torque / 45000.0;
  double rotor_acceleration = angular_acceleration * 60.0 / (2 * NPD_PI);
  if (torques_out) { torques_out[0] = total_friction; torques_out[1] = net_torque; torques_out[2] = rotor_acceleration; }   /* state-log diagnostics */
  t->rotor_speed += rotor_acceleration * dt_seconds;
  t->rotor_speed = npd_pymax(0.0, npd_pymin(t->rotor_speed, 3780.0));
  if (torques_out) torques_out[7] = (t->rotor_speed > 3780.0 * 0.99) ? 1.0 : 0.0;     /* an overspeed event, :900-902 */
  /* calculate_thermal_effects :913-954 */
  double temp_change = (steam_temperature - t->rotor_temperature) / 2.0 * dt;
  t->rotor_temperature += temp_change;
  double temp_difference = t->rotor_temperature - 25.0;
  t->thermal_expansion = (temp_difference * 1.2e-05 * 12.0 * 1000.0);
  if (t->rotor_speed < 100.0) {
    double thermal_gradient = (dt > 0) ? fabs(temp_change) / dt : 0;
    double bow_increase = thermal_gradient * 0.001 * dt;
    t->thermal_bow = npd_pymin(2.0, t->thermal_bow + bow_increase);
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
    double unbalance_force = npd_sq(t->rotor_speed / 3600.0) * 0.1;
    double total_load = static_load + thrust_load + thermal_load + unbalance_force;
    total_load *= (2.0 - t->bearing_wear_factor[i]);
    t->bearing_load[i] = total_load;
    /* calculate_bearing_temperature :157-270 (oil inlet 40 C, flow injected by the lubrication wrapper) */
    double oil_inlet_temp = npd_pymax(20.0, npd_pymin(150.0, 40.0));
    double bearing_load = npd_pymax(0.0, total_load);
    double rotor_speed = npd_pymax(0.0, t->rotor_speed);
    double bearing_load_n = bearing_load * 1000.0;
    double angular_velocity = rotor_speed * 2 * NPD_PI / 60.0;
    double friction_torque = 0.001 * bearing_load_n * 0.15;
    double friction_power = friction_torque * angular_velocity;
    friction_power = npd_pymin(friction_power, 50000.0);
    if (!isfinite(friction_power) || friction_power < 0) friction_power = 0.0;
    double oil_mass_flow = NPD_TLUB_OIL_FLOW[i] / 60.0 * 850.0 / 1000.0;
    double temp_rise = friction_power / (oil_mass_flow * 2000.0);
    temp_rise = npd_pymin(50.0, npd_pymax(0.0, temp_rise));
    t->bearing_metal_temp[i] = npd_pymax(30.0, npd_pymin(200.0, oil_inlet_temp + temp_rise * 1.5));
    /* update_bearing_wear :272-315 (oil_contamination argument is the constant 5.0) */
    double lf = total_load / 500.0;
    double load_wear_rate = 0.00001 * npd_sq(lf) * dt;
    double contamination_wear_rate = 0.000005 * 5.0 * dt;
    t->bearing_wear_factor[i] = npd_pymax(0.5, t->bearing_wear_factor[i] - (load_wear_rate + contamination_wear_rate));
    if (torques_out) torques_out[3 + i] = (load_wear_rate + contamination_wear_rate) * 0.01;     /* this step's clearance increase [mm], rotor_dynamics.py:298-300 */
    max_bearing_metal = npd_pymax(max_bearing_metal, t->bearing_metal_temp[i]);
  }
  /* VibrationMonitor.calculate_vibration_response :624-704 */
  double avg_stiffness = (0.0 + 1e8 + 1e8 + 1e8 + 1e8) / 4, avg_damping = (0.0 + 1e5 + 1e5 + 1e5 + 1e5) / 4;
  double vib_unbalance_force = npd_sq(t->rotor_speed / 60.0) * 0.1;
  double rotation_frequency = t->rotor_speed / 60.0;
  const double rotor_mass = 15000.0;
  double natural_frequency = npd_sqrt(avg_stiffness / rotor_mass) / (2 * NPD_PI);
  double frequency_ratio = rotation_frequency / natural_frequency;
  double critical_damping = 2 * npd_sqrt(avg_stiffness * rotor_mass);
  double damping_ratio = avg_damping / critical_damping;
  double denominator = npd_sqrt(npd_sq(1 - npd_sq(frequency_ratio)) + npd_sq(2 * damping_ratio * frequency_ratio));
  double unbalance_response = vib_unbalance_force / avg_stiffness / denominator;
  double thermal_response = t->thermal_bow * npd_sq(frequency_ratio) / denominator;
  double displacement_1x = (unbalance_response + thermal_response) * 39.37;
  double displacement_2x = displacement_1x * 0.1, displacement_3x = displacement_1x * 0.05;
  double total_displacement = npd_sqrt(npd_sq(displacement_1x) + npd_sq(displacement_2x) + npd_sq(displacement_3x));
  t->vibration_displacement = total_displacement;

  *max_bearing_metal_out = max_bearing_metal; *total_displacement_out = total_displacement;
}

/* TurbineProtectionSystem.check_trip_conditions  enhanced_physics.py:348-436 and the power it leaves (:760-800) */
NPD_FN void npd_turbine_protect(npb_turb_t *t, double stage_power_mw, double max_stress, double max_bearing_metal, double total_displacement,
                                int sg_system_availability, double condenser_pressure, double dt) {
  double dt_seconds = dt * 3600.0;

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

}

template <int SM = 0>
NPD_FN void npd_turbine_update(npb_turb_t *t, const npd_stage_t &st, double steam_pressure,
                               double steam_temperature, double steam_flow,
                               const double *sg_pressures, int sg_system_availability, double load_demand,
                               double condenser_pressure, double dt, npd_turbine_result_t *res) {
  double oil_temps[4];
  npd_turbine_lube(t, dt, st.diag ? oil_temps : nullptr);
  if (st.diag) {
#pragma unroll
    for (int i = 0; i < 4; i++) NPD_DIAG(st, NPB_DIAG_BEARING_OIL_TEMP + i, oil_temps[i]);
  }

  NPD_STAMP(12);
  /* ================= EnhancedTurbinePhysics.update_state  enhanced_physics.py:694-890 ========= */
  t->load_demand = load_demand;
  double pressure_stability_factor = npd_pressure_stability_factor(sg_pressures);
  npd_stagesys_out_t ss;
  npd_stage_system_update<SM>(st, (const double *)0, steam_pressure, steam_temperature, steam_flow, load_demand,
                          pressure_stability_factor, dt, &ss);
  NPD_STAMP(15);
  double stage_power_mw = ss.total_power;
  double max_bearing_metal, total_displacement;
  double torques[8];
  npd_turbine_rotor(t, stage_power_mw, steam_temperature, load_demand, dt, &max_bearing_metal, &total_displacement, st.diag ? torques : nullptr);
  if (st.diag) {
    NPD_DIAG(st, NPB_DIAG_ROTOR_FRICTION_TORQUE, torques[0]); NPD_DIAG(st, NPB_DIAG_ROTOR_NET_TORQUE, torques[1]); NPD_DIAG(st, NPB_DIAG_ROTOR_ACCELERATION, torques[2]);
    NPD_DIAG(st, NPB_DIAG_STAGE_SYSTEM_TOTAL_POWER, stage_power_mw);     /* stage_system.py:976: the stability-adjusted stage power, before the protection gates it */
    /* accumulators the reference carries and nothing in the physics reads: summed in the caller's buffer, i.e. since the
     * diagnostics were switched on (from a zeroed buffer at construction they are the reference's) */
#pragma unroll
    for (int q = 0; q < 5; q++) st.diag[(size_t)(NPB_DIAG_ROTOR_CLEARANCE_INCREASE + q) * st.diag_pitch] += torques[3 + q];
    /* the stage system's efficiency factor, another carried product nothing reads (stage_system.py:981: multiplied by the pressure
     * stability factor every step and never restored); a stored 0 is a buffer that has not been through a step, i.e. 1.0 */
    double *se_row = &st.diag[(size_t)NPB_DIAG_STAGE_SYSTEM_EFFICIENCY * st.diag_pitch];
    const double se = npd_pymin(1.0, ((*se_row == 0.0) ? 1.0 : *se_row) * pressure_stability_factor);
    *se_row = npd_pymax(se, 2.2250738585072014e-308);
    /* performance factor, enhanced_physics.py:147-156 and :823-828 */
    const double rate_risk = npd_pymin(1.0, ss.max_temp_rate / 10.0), gradient_risk = npd_pymin(1.0, ss.max_gradient / 5.0);
    const double stress_risk = npd_pymin(1.0, ss.max_thermal_stress / 800e6);
    const double thermal_shock_risk = npd_pymax3(rate_risk, gradient_risk, stress_risk);
    const double applied_torque = stage_power_mw * 1e6 / (2 * NPD_PI * 3600 / 60);
    NPD_DIAG(st, NPB_DIAG_TURBINE_PERFORMANCE_FACTOR, se * (1.0 - torques[0] / npd_pymax(1.0, applied_torque) * 0.1) * (1.0 - thermal_shock_risk * 0.1));
  }
  /* MetalTemperatureTracker.update_temperatures ran per stage inside the stage pass (npd_stage_post) */
  npd_turbine_protect(t, stage_power_mw, ss.max_thermal_stress, max_bearing_metal, total_displacement, sg_system_availability, condenser_pressure, dt);
  res->electrical_power_gross = t->total_power_output;
  res->mechanical_power = t->total_power_output / 0.985;
  res->electrical_power_net = t->total_power_output * 0.98;
  res->effective_steam_flow = steam_flow - ss.total_extraction;
  res->condenser_pressure = condenser_pressure;
  res->condenser_temperature = npd_tsat_antoine(condenser_pressure);
  res->lp6_outlet_enthalpy = ss.lp6_outlet_enthalpy;
  res->hp_power = ss.hp_power; res->lp_power = ss.lp_power; res->overall_efficiency = ss.overall_efficiency;
  res->trip_active = t->trip_active;
}

#endif
