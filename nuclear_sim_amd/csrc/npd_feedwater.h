/*
 * npd_feedwater.h -- device physics: feedwater system (4 pumps with lubrication, three-element
 * level control, shared cavitation monitor, protection system).
 *
 * Follows EnhancedFeedwaterPhysics.update_state  feedwater/physics.py:662-863.
 * dt is the simulator's dt, which this subsystem treats as minutes
 * (secondary/__init__.py:490; pump_lubrication.py:1786-1797 divides by 60 for hours).
 */
#ifndef NPD_FEEDWATER_H
#define NPD_FEEDWATER_H
#include "npd_common.h"
#include "npd_lube.h"

enum { NPD_PUMP_RUNNING = 0, NPD_PUMP_STOPPED = 1, NPD_PUMP_STARTING = 2, NPD_PUMP_STOPPING = 3, NPD_PUMP_TRIPPED = 4 };
enum {
  NPD_TRIP_NONE = 0, NPD_TRIP_LOW_FLOW, NPD_TRIP_NPSH, NPD_TRIP_LOW_SUCTION, NPD_TRIP_HIGH_DISCHARGE,
  NPD_TRIP_SG_HIGH_LEVEL, NPD_TRIP_SEVERE_CAVITATION, NPD_TRIP_CAVITATION_DAMAGE, NPD_TRIP_CRITICAL_NPSH,
  NPD_TRIP_LUB_VERY_LOW_OIL = 10, NPD_TRIP_LUB_LOW_OIL, NPD_TRIP_LUB_OVERFILL, NPD_TRIP_LUB_COMPONENT_WEAR,
  NPD_TRIP_LUB_SEAL_LEAKAGE, NPD_TRIP_LUB_COMBINED_WEAR, NPD_TRIP_LUB_PERFORMANCE
};

/* pump constants: FeedwaterPumpConfig as built by EnhancedFeedwaterPhysics.__init__
 * (feedwater/physics.py:112-117: rated_flow = design_flow_per_pump, rated_power = flow * 0.02) */
#define NPD_PUMP_RATED_FLOW 500.0
#define NPD_PUMP_RATED_POWER 10.0
#define NPD_PUMP_RAMP_RATE 15.0
#define NPD_PUMP_STARTUP_TIME 20.0
#define NPD_PUMP_COASTDOWN_TIME 60.0
#define NPD_PUMP_MAX_SPEED 110.0
#define NPD_OIL_RESERVOIR_CAPACITY 150.0

/* LubricationComponent tables  pump_lubrication.py:108-199 (order = dict order) */
typedef struct { double base, load_exp, speed_exp, contam_factor, trip_threshold; } npd_lubcomp_t;
static __device__ const npd_lubcomp_t NPD_PUMP_COMP[6] = {
  /* impeller        */ {0.002, 1.8, 2.0, 1.5, 25.0},
  /* motor_bearings  */ {0.004, 1.8, 2.0, 2.5, 60.0},
  /* pump_bearings   */ {0.006, 2.2, 1.8, 3.0, 50.0},
  /* thrust_bearing  */ {0.008, 2.4, 1.6, 3.5, 40.0},
  /* mechanical_seals*/ {0.01,  2.0, 1.4, 4.0, 50.0},
  /* coupling_system */ {0.0003,1.3, 1.0, 1.5, 35.0},
};

/* property accessors  pump_lubrication.py:225-238 */
NPD_FN double npd_pump_efficiency_factor(const npb_pump_t *p) { return npd_pymax(0.5, 1.0 - (p->efficiency_degradation / 100.0)); }
NPD_FN double npd_pump_flow_factor(const npb_pump_t *p) { return npd_pymax(0.5, 1.0 - (p->flow_degradation / 100.0)); }

NPD_FN double npd_pymax3(double a, double b, double c) { return npd_pymax(npd_pymax(a, b), c); }

/* FeedwaterPumpLubricationSystem._calculate_pump_performance_factors  pump_lubrication.py:1412-1478 */
NPD_FN void npd_pump_performance_factors(npb_pump_t *p, double cavitation_damage) {
  double bearing_efficiency_loss = ((p->wear_motor_bearings / 100.0) * 0.01 + (p->wear_pump_bearings / 100.0) * 0.015 +
                                    (p->wear_thrust_bearing / 100.0) * 0.02);
  double seal_efficiency_loss = (p->wear_mechanical_seals / 100.0) * 0.01;
  double lubrication_efficiency_loss = (1.0 - p->lubrication_effectiveness) * 0.02;
  double cavitation_efficiency_loss = npd_pymin(0.3, cavitation_damage * 0.01);
  double cavitation_flow_loss = cavitation_efficiency_loss * 0.5;
  double impeller_flow_loss = cavitation_damage * 0.02;
  double impeller_efficiency_loss = cavitation_damage * 0.015;
  double total_efficiency_loss = (bearing_efficiency_loss + seal_efficiency_loss + lubrication_efficiency_loss +
                                  cavitation_efficiency_loss + impeller_efficiency_loss);
  double total_flow_loss = (cavitation_flow_loss + impeller_flow_loss + bearing_efficiency_loss * 0.3);
  double total_head_loss = (impeller_flow_loss * 0.8 + cavitation_efficiency_loss * 0.4);
  p->efficiency_degradation = npd_pymin(50.0, total_efficiency_loss * 100.0);
  p->flow_degradation = npd_pymin(50.0, total_flow_loss * 100.0);
  p->head_degradation = npd_pymin(30.0, total_head_loss * 100.0);
  double total_bearing_wear = p->wear_motor_bearings + p->wear_pump_bearings + p->wear_thrust_bearing;
  p->vibration_increase = total_bearing_wear * 0.1 + cavitation_damage * 0.05;
}

/* FeedwaterPumpLubricationSystem._calculate_lubrication_effectiveness  pump_lubrication.py:240-269
 * (used at construction and by maintenance actions) */
NPD_FN void npd_pump_lubrication_effectiveness(npb_pump_t *p) {
  double cf = npd_pymax(0.1, 1.0 - p->oil_contamination / 15.0);
  double af = npd_pymax(0.1, 1.0 - p->oil_acidity / 1.6);
  double mf = npd_pymax(0.1, 1.0 - p->oil_moisture / 0.08);
  double ao = npd_pymax(0.1, p->antioxidant_level / 100.0);
  double aw = npd_pymax(0.1, p->anti_wear_level / 100.0);
  double ci = npd_pymax(0.1, p->corrosion_inhibitor_level / 100.0);
  double e = (cf * 0.25 + ao * 0.20 + aw * 0.20 + ci * 0.15 + af * 0.10 + mf * 0.10);
  p->lubrication_effectiveness = npd_pymax(0.1, npd_pymin(1.0, e));
}

/* FeedwaterPump._calculate_dynamic_npsh_required  pump_system.py:362-420 */
NPD_FN double npd_pump_npsh_required(const npb_pump_t *p) {
  const double base_npsh = 12.0;
  double impeller_wear_penalty = p->wear_impeller * 0.1;
  double cavitation_roughness_penalty = p->cavitation_damage * 0.2;
  double speed_penalty = npd_pymax(0.0, (p->speed_percent - 100.0) * 0.02);
  double flow_ratio = p->flow_rate / NPD_PUMP_RATED_FLOW;
  double flow_penalty = npd_pymax(0.0, (flow_ratio - 1.0) * 1.5);
  double max_bearing_wear = npd_pymax3(p->wear_motor_bearings, p->wear_pump_bearings, p->wear_thrust_bearing);
  double bearing_penalty = max_bearing_wear * 0.05;
  double coupling_penalty = (p->wear_impeller * max_bearing_wear / 10000.0) * 0.3;
  double total = (base_npsh + impeller_wear_penalty + cavitation_roughness_penalty + speed_penalty + flow_penalty +
                  bearing_penalty + coupling_penalty);
  return npd_pymax(base_npsh, total);
}

/* FeedwaterPump.set_flow_demand  pump_system.py:422-447 */
NPD_FN void npd_pump_set_flow_demand(npb_pump_t *p, double flow_demand) {
  p->flow_demand = npd_clip(flow_demand, 0.0, NPD_PUMP_RATED_FLOW * 1.2);
  if (flow_demand > 0) {
    double effective_capacity = NPD_PUMP_RATED_FLOW * npd_pump_flow_factor(p);
    double speed_setpoint;
    if (effective_capacity > 0) speed_setpoint = npd_sqrt(flow_demand / effective_capacity) * 100.0;
    else speed_setpoint = 100.0;
    p->speed_setpoint = npd_clip(speed_setpoint, 0.0, 100.0);
  } else {
    p->speed_setpoint = 0.0;
  }
}

NPD_FN void npd_pump_trip(npb_pump_t *p, int reason) { /* BasePump._trip_pump pump_models.py:261-267 */
  p->status = NPD_PUMP_TRIPPED; p->trip_active = 1; p->trip_reason = reason; p->available = 0;
}

/* FeedwaterPumpLubricationSystem.check_protection_trips  pump_lubrication.py:1536-1580 */
NPD_FN int npd_pump_lub_trip(const npb_pump_t *p) {
  if (p->oil_level < 5.0) return NPD_TRIP_LUB_VERY_LOW_OIL;
  if (p->oil_level < 10.0) return NPD_TRIP_LUB_LOW_OIL;
  if (p->oil_level > 105.0) return NPD_TRIP_LUB_OVERFILL;
  const double wear[6] = {p->wear_impeller, p->wear_motor_bearings, p->wear_pump_bearings, p->wear_thrust_bearing,
                          p->wear_mechanical_seals, p->wear_coupling_system};
  double total_wear = 0.0;
  for (int i = 0; i < 6; i++) if (wear[i] > NPD_PUMP_COMP[i].trip_threshold) return NPD_TRIP_LUB_COMPONENT_WEAR;
  if (p->seal_leakage_rate > 10.0) return NPD_TRIP_LUB_SEAL_LEAKAGE;
  for (int i = 0; i < 6; i++) total_wear += wear[i];
  if (total_wear > 40.0) return NPD_TRIP_LUB_COMBINED_WEAR;
  if ((1.0 - npd_pump_efficiency_factor(p)) * 100.0 > 25.0) return NPD_TRIP_LUB_PERFORMANCE;
  return 0;
}

/* FeedwaterPump._simulate_sensors  pump_system.py:637-744 (with _initial_conditions_applied set for
 * every pump by EnhancedFeedwaterPhysics._apply_initial_conditions, physics.py:353, so suction/discharge
 * pressure and NPSH keep their initial-condition values) */
NPD_FN void npd_pump_sensors(npb_pump_t *p) {
  p->differential_pressure = p->discharge_pressure - p->suction_pressure;
  double load_factor = (NPD_PUMP_RATED_FLOW > 0) ? p->flow_rate / NPD_PUMP_RATED_FLOW : 0.0;
  p->motor_temperature = 60.0 + 20.0 * load_factor;
  double base_vibration = 1.0 + 0.05 * p->speed_percent;
  double v = base_vibration + p->vibration_increase + p->cavitation_intensity * 2.0;
  p->vibration_level = npd_pymax(0.5, npd_pymin(15.0, v));
}

typedef struct npd_pump_sysconds_t {
  double feedwater_temperature, suction_pressure, discharge_pressure, max_sg_level;
} npd_pump_sysconds_t;

/* update_with_lubrication closure  pump_lubrication.py:1659-1852, then
 * FeedwaterPump.update_pump  pump_system.py:449-554 -> BasePump.update_pump  pump_models.py:104-146 */
NPD_FN void npd_pump_update(npb_pump_t *p, const npd_pump_sysconds_t *sc, double dt) {
  /* ---- lubrication pre-step, from the PREVIOUS step's pump state */
  double load_factor = (NPD_PUMP_RATED_FLOW > 0) ? p->flow_rate / NPD_PUMP_RATED_FLOW : 0.0;
  double speed_factor = p->speed_percent / 100.0;
  double electrical_load_factor = (NPD_PUMP_RATED_POWER > 0) ? p->power_consumption / NPD_PUMP_RATED_POWER : 0.0;
  double cav = p->cavitation_intensity;
  double pressure_factor = p->differential_pressure / 7.5;
  double base_temp = 40.0 + load_factor * 10.0;
  double motor_heat = electrical_load_factor * 2.0;
  double feedwater_heat_effect = (sc->feedwater_temperature - 200.0) * 0.01;
  double pressure_ratio = (sc->suction_pressure > 0) ? sc->discharge_pressure / sc->suction_pressure : 16.0;
  double pressure_heat = npd_pymax(0.0, (pressure_ratio - 12.0) * 0.5);
  double cavitation_heat = cav * 3.0;
  double oil_temp = base_temp + motor_heat + feedwater_heat_effect + pressure_heat + cavitation_heat;
  oil_temp = npd_pymax(35.0, npd_pymin(75.0, oil_temp));
  double base_contamination_input = load_factor * 0.002;
  double bearing_wear_contamination = (p->wear_motor_bearings + p->wear_pump_bearings + p->wear_thrust_bearing) * 0.0025;
  double seal_wear_contamination = p->wear_mechanical_seals * 0.004;
  double cavitation_contamination = cav * 0.01;
  double temp_contamination = (oil_temp > 70.0) ? (oil_temp - 70.0) * 0.0025 : 0.0;
  double lubrication_quality_factor = npd_pymax(0.3, p->lubrication_effectiveness);
  double contamination_scaling = 2.0 - (lubrication_quality_factor * 0.7);
  double total_contamination_input = (base_contamination_input + bearing_wear_contamination + seal_wear_contamination +
                                      cavitation_contamination + temp_contamination) * contamination_scaling;
  total_contamination_input = npd_pymin(0.5, npd_pymax(0.0005, total_contamination_input));

  double avg_wear = 0.0;
  avg_wear += p->wear_impeller; avg_wear += p->wear_motor_bearings; avg_wear += p->wear_pump_bearings;
  avg_wear += p->wear_thrust_bearing; avg_wear += p->wear_mechanical_seals; avg_wear += p->wear_coupling_system;
  avg_wear = avg_wear / 6;
  npd_oil_t oil = {p->oil_temperature, p->oil_contamination, p->oil_moisture, p->oil_acidity, p->oil_viscosity_change,
                   p->antioxidant_level, p->anti_wear_level, p->corrosion_inhibitor_level, p->lubrication_effectiveness};
  const npd_oil_limits_t lim = {15.0, 1.6, 0.08, 10.0};
  npd_update_oil_quality(&oil, &lim, avg_wear, oil_temp, total_contamination_input, 0.0001, dt / 60.0);
  p->oil_temperature = oil.temperature; p->oil_contamination = oil.contamination; p->oil_moisture = oil.moisture;
  p->oil_acidity = oil.acidity; p->oil_viscosity_change = oil.viscosity_change; p->antioxidant_level = oil.antioxidant;
  p->anti_wear_level = oil.anti_wear; p->corrosion_inhibitor_level = oil.corrosion_inhibitor;
  p->lubrication_effectiveness = oil.effectiveness;

  /* update_component_wear  lubrication_base.py:354-401 with calculate_component_wear
   * pump_lubrication.py:275-396; wear levels are read live, so later components see earlier updates */
  /* the three distinct bases of the wear-rate powers with fractional exponents; every rate is then one exp of a sum of their logarithms */
  const double l_electrical = npd_log(electrical_load_factor), l_speed = npd_log(speed_factor), l_load = npd_log(load_factor);
#pragma unroll
  for (int i = 0; i < 6; i++) {
    const npd_lubcomp_t *c = &NPD_PUMP_COMP[i];
    double impeller_wear = p->wear_impeller;
    double max_bearing_wear = npd_pymax3(p->wear_motor_bearings, p->wear_pump_bearings, p->wear_thrust_bearing);
    double wear_rate;
    switch (i) {
      case 0: { /* impeller: no entry in component_conditions -> all defaults (load 1, speed 1, 55 C, no cavitation) */
        double temp_factor = npd_pymax(1.0, (55.0 - 80.0) / 40.0);
        double bearing_coupling = 1.0 + (max_bearing_wear / 100.0) * 0.3;
        wear_rate = (c->base * 1.0 * 1.0 * (1.0 + 0.0 * 3.0) * temp_factor * bearing_coupling); /* 1 ** e = 1 exactly */
      } break;
      case 1: {
        double temperature = 60.0 + electrical_load_factor * 25.0;
        double temp_factor = npd_pymax(1.0, (temperature - 60.0) / 25.0);
        double coupling = 1.0 + (impeller_wear / 100.0) * 0.2;
        wear_rate = (c->base * npd_pow_logs(l_electrical, c->load_exp, l_speed, c->speed_exp) * temp_factor * coupling);
      } break;
      case 2: {
        double temperature = 50.0 + load_factor * 30.0;
        double cavitation_factor = 1.0 + cav * 2.0;
        double temp_factor = npd_pymax(1.0, (temperature - 50.0) / 30.0);
        double coupling = 1.0 + (impeller_wear / 100.0) * 0.4;
        wear_rate = (c->base * npd_pow_logs(l_load, c->load_exp, l_speed, c->speed_exp) * cavitation_factor * temp_factor * coupling);
      } break;
      case 3: {
        double coupling = 1.0 + (impeller_wear / 100.0) * 0.25;   /* axial_load_factor = 1.0 * load_factor */
        wear_rate = (c->base * npd_pow_logs(l_load, c->load_exp, l_speed, c->speed_exp) * coupling);
      } break;
      case 4: {
        double cavitation_seal_factor = 1.0 + cav * 5.0;
        double impeller_coupling = 1.0 + (impeller_wear / 100.0) * 0.15;
        double bearing_coupling = 1.0 + (max_bearing_wear / 100.0) * 0.2;
        wear_rate = (c->base * (pressure_factor * pressure_factor) * 1.0 * cavitation_seal_factor * impeller_coupling * bearing_coupling); /* load_exp = 2.0: the square itself */
      } break;
      default: {
        double bearing_coupling = 1.0 + (max_bearing_wear / 100.0) * 0.3;
        wear_rate = (c->base * 1.0 * 1.0 * npd_exp(c->load_exp * l_load) * bearing_coupling);
      } break;
    }
    wear_rate *= 1.0; /* chemistry_wear_factor default */
    double lubrication_wear_factor = 1.0 + (1.0 - p->lubrication_effectiveness) * c->contam_factor;
    double actual_wear_rate = wear_rate * lubrication_wear_factor;
    double inc = actual_wear_rate * (dt / 60.0);
    switch (i) {
      case 0: p->wear_impeller += inc; break;
      case 1: p->wear_motor_bearings += inc; break;
      case 2: p->wear_pump_bearings += inc; break;
      case 3: p->wear_thrust_bearing += inc; break;
      case 4: p->wear_mechanical_seals += inc; break;
      default: p->wear_coupling_system += inc; break;
    }
  }

  /* update_pump_lubrication_effects  pump_lubrication.py:570-623 (dt in minutes) */
  double total_leakage = 0.001 + p->wear_mechanical_seals * 0.2 + cav * 0.1;
  p->seal_leakage_rate = npd_pymin(total_leakage, 0.05);
  if (p->seal_leakage_rate > 0) {
    double oil_lost_liters = p->seal_leakage_rate * dt;
    double oil_loss_percentage = (oil_lost_liters / NPD_OIL_RESERVOIR_CAPACITY) * 100.0;
    p->oil_level = npd_pymax(0.0, p->oil_level - oil_loss_percentage * 0.5);
  }
  p->oil_level = npd_pymin(100.0, npd_pymax(0.0, p->oil_level));
  npd_pump_performance_factors(p, 0.0); /* 'cavitation_damage' is not in pump_conditions -> 0.0 */

  /* ---- BasePump.update_pump */
  /* _update_pump_dynamics  pump_models.py:166-198 */
  if (p->status == NPD_PUMP_RUNNING) {
    double speed_error = p->speed_setpoint - p->speed_percent;
    double max_change = NPD_PUMP_RAMP_RATE * dt;
    if (fabs(speed_error) <= max_change) p->speed_percent = p->speed_setpoint;
    else p->speed_percent += max_change * ((speed_error > 0) - (speed_error < 0));
  } else if (p->status == NPD_PUMP_STARTING) {
    p->speed_percent += (100.0 / NPD_PUMP_STARTUP_TIME) * dt;
    if (p->speed_percent >= p->speed_setpoint * 0.95) { p->status = NPD_PUMP_RUNNING; p->speed_percent = p->speed_setpoint; }
  } else if (p->status == NPD_PUMP_STOPPING) {
    p->speed_percent -= (100.0 / NPD_PUMP_COASTDOWN_TIME) * dt;
    if (p->speed_percent <= 5.0) { p->speed_percent = 0.0; p->status = NPD_PUMP_STOPPED; }
  }
  p->speed_percent = npd_clip(p->speed_percent, 0.0, NPD_PUMP_MAX_SPEED);

  /* _calculate_flow_rate  pump_system.py:172-216 (+ _apply_system_effects :218-237) */
  int active = (p->status == NPD_PUMP_RUNNING || p->status == NPD_PUMP_STARTING);
  if (active) {
    double speed_ratio = p->speed_percent / 100.0;
    if (p->flow_demand > 0) {
      if (speed_ratio > 0.8) p->flow_rate = p->flow_demand;
      else p->flow_rate = npd_pymin(p->flow_demand, NPD_PUMP_RATED_FLOW * speed_ratio);
    } else {
      p->flow_rate = NPD_PUMP_RATED_FLOW * speed_ratio;
    }
    double temp_factor = 1.0 - (sc->feedwater_temperature - 227.0) * 0.0002;
    p->flow_rate *= temp_factor;
    p->flow_rate *= npd_pump_flow_factor(p);
    if (p->status == NPD_PUMP_RUNNING && p->speed_percent < 20.0) p->flow_rate = npd_pymax(p->flow_rate, NPD_PUMP_RATED_FLOW * 0.05);
    p->flow_rate = npd_pymin(p->flow_rate, NPD_PUMP_RATED_FLOW * 1.2);
  } else {
    p->flow_rate = 0.0;
  }
  /* _calculate_power_consumption  pump_system.py:250-275 */
  if (active) {
    double speed_ratio = p->speed_percent / 100.0;
    double flow_ratio = p->flow_rate / NPD_PUMP_RATED_FLOW;
    double head_ratio = npd_sq(speed_ratio);
    double base_power = NPD_PUMP_RATED_POWER * (flow_ratio * head_ratio);
    p->power_consumption = base_power / npd_pump_efficiency_factor(p);
    if (p->status == NPD_PUMP_STARTING) p->power_consumption = npd_pymax(p->power_consumption, NPD_PUMP_RATED_POWER * 0.2);
  } else {
    p->power_consumption = 0.0;
  }
  npd_pump_sensors(p);

  /* _check_protection_systems  pump_models.py:243-259 then pump_system.py:277-333 */
  do {
    if (p->status == NPD_PUMP_STOPPED) { p->trip_active = 0; p->trip_reason = 0; }
    else if (p->status == NPD_PUMP_RUNNING && p->flow_rate < 25.0) { npd_pump_trip(p, NPD_TRIP_LOW_FLOW); }
    if (p->trip_active) break;
    if (p->status == NPD_PUMP_STARTING) break;
    double npsh_required = npd_pump_npsh_required(p);
    if (p->status == NPD_PUMP_RUNNING && p->npsh_available < npsh_required) { npd_pump_trip(p, NPD_TRIP_NPSH); break; }
    if (p->status == NPD_PUMP_RUNNING && p->suction_pressure < 0.2) { npd_pump_trip(p, NPD_TRIP_LOW_SUCTION); break; }
    if (sc->discharge_pressure > 10.0) { npd_pump_trip(p, NPD_TRIP_HIGH_DISCHARGE); break; }
    if (sc->max_sg_level > 16.0) { npd_pump_trip(p, NPD_TRIP_SG_HIGH_LEVEL); break; }
    /* _check_cavitation_trips :335-359 */
    if (p->cavitation_intensity > 0.7) { npd_pump_trip(p, NPD_TRIP_SEVERE_CAVITATION); break; }
    if (p->cavitation_damage > 10.0) { npd_pump_trip(p, NPD_TRIP_CAVITATION_DAMAGE); break; }
    if (p->npsh_available < 8.0 * 0.5) { npd_pump_trip(p, NPD_TRIP_CRITICAL_NPSH); break; }
    int lub = npd_pump_lub_trip(p);
    if (lub) { npd_pump_trip(p, lub); break; }
  } while (0);

  /* ---- FeedwaterPump.update_pump tail: sensors again, cavitation, wear */
  npd_pump_sensors(p);
  /* _simulate_cavitation  pump_system.py:556-601 */
  if (!(p->status == NPD_PUMP_RUNNING || p->status == NPD_PUMP_STARTING)) {
    p->cavitation_intensity = 0.0; p->cavitation_time = 0.0;
  } else {
    double cavitation_threshold = npd_pump_npsh_required(p) + 2.0;
    if (p->npsh_available < cavitation_threshold) {
      double npsh_deficit = cavitation_threshold - p->npsh_available;
      double severity = npd_pymin(1.0, npsh_deficit / cavitation_threshold);
      double flow_factor = npd_sq(p->flow_rate / NPD_PUMP_RATED_FLOW);
      p->cavitation_intensity = severity * flow_factor;
      p->cavitation_time += dt * 60.0;
      p->vibration_level += p->cavitation_intensity * 2.0;
    } else {
      p->cavitation_intensity = 0.0;
      p->cavitation_time = npd_pymax(0.0, p->cavitation_time - dt * 6.0);
    }
    p->cavitation_damage += npd_sq(p->cavitation_intensity) * dt / 60.0;
  }
  /* _simulate_mechanical_wear  :603-617 */
  if (p->status == NPD_PUMP_RUNNING && p->cavitation_intensity > 0.1)
    p->cavitation_damage += npd_sq(p->cavitation_intensity) * dt / 60.0;
}


/* ---- EnhancedFeedwaterPhysics.update_state  feedwater/physics.py:662-863, split for streaming:
 *   npd_fw_level_control  (once)      ThreeElementControl.calculate_flow_demands
 *   npd_fw_pump_step      (per pump)  pump update + that pump's share of the diagnostics and
 *                                     protection passes (their shared state -- one CavitationModel,
 *                                     one NPSHProtection, shared trip timers -- is visited in pump
 *                                     order, exactly as the reference's per-pump loops do, and the
 *                                     passes touch disjoint state, so interleaving them per pump
 *                                     gives the same result)
 *   npd_fw_finish         (once)      system-level protection, availability
 * sg_levels / sg_steam_flows / sg_qualities are the PREVIOUS step's SG conditions. */
typedef struct npd_fw_acc_t {
  double total_flow, total_power, flow_sum;
  double total_cavitation_risk, total_wear_level, total_vibration;
  int running_count, running_mask, trips;
  uint32_t trip_mask;
  int trip_kinds;   /* state-log diagnostics only: bit 0 = a trip that starts emergency feedwater, bit 1 = one that opens the steam dump
                     * (protection_system.py:680-692); dead code in every build that does not write diagnostics */
} npd_fw_acc_t;

/* ThreeElementControl.calculate_flow_demands  level_control.py:157-363
 * (SteamQualityCompensator.calculate_quality_compensation :58-105 first, for all SGs) */
NPD_FN double npd_fw_level_control(npb_fw_t *fw, const double *sg_levels, const double *sg_steam_flows,
                                   const double *sg_qualities, double dt) {
  double quality_corrections[NPB_NUM_SG];
#pragma unroll
  for (int i = 0; i < NPB_NUM_SG; i++) {
    double quality_error = 0.99 - sg_qualities[i];
    if (fabs(quality_error) < 0.005) quality_error = 0.0;
    double proportional = quality_error * 1.0 * sg_steam_flows[i];
    fw->quality_integral_error += quality_error * dt;
    double integral = fw->quality_integral_error * 0.1 * sg_steam_flows[i];
    quality_corrections[i] = npd_clip(proportional + integral, -50.0, 50.0);
  }
  const double design_flow_per_sg = 500.0, target_level = 12.5;
  double total_flow_demand = 0.0;
#pragma unroll
  for (int i = 0; i < NPB_NUM_SG; i++) {
    double level_error = target_level - sg_levels[i];
    double steam_flow = sg_steam_flows[i];
    double proportional_correction = 1.0 * level_error * 0.2;
    fw->level_integral_errors[i] += level_error * dt;
    double integral_correction = fw->level_integral_errors[i] * 0.0005 * 0.5;
    double level_error_rate = (level_error - fw->previous_level_errors[i]) / dt;
    double derivative_correction = level_error_rate * 0.0002 * 0.5;
    double absolute_minimum = design_flow_per_sg * 0.05;
    double feedforward_demand = (steam_flow < absolute_minimum) ? absolute_minimum : steam_flow;
    double level_correction = proportional_correction + integral_correction + derivative_correction;
    double max_level_correction = feedforward_demand * 0.01;
    level_correction = npd_clip(level_correction, -max_level_correction, max_level_correction);
    double flow_error = feedforward_demand - 500.0; /* previous_feedwater_flows is never updated (level_control.py:143) */
    double flow_feedback_correction = flow_error * 0.1;
    double level_contribution = level_correction * 0.4;
    double flow_contribution = flow_feedback_correction * 0.1;
    double total_demand = feedforward_demand + level_contribution + flow_contribution + quality_corrections[i];
    total_demand = npd_clip(total_demand, design_flow_per_sg * 0.05, design_flow_per_sg * 2.0);
    total_flow_demand += total_demand;
    fw->previous_level_errors[i] = level_error;
  }
  return total_flow_demand;
}

/* one pump of FeedwaterPumpSystem.update_system  pump_system.py:1235-1329, followed by this pump's pass
 * through PerformanceDiagnostics.update_diagnostics  performance_monitoring.py:423-542 (shared
 * CavitationModel :113-202) and the per-pump loops of FeedwaterProtectionSystem.check_protection_systems
 * protection_system.py:378-481.  Protection setpoints resolve through getattr fallbacks against
 * FeedwaterProtectionConfig: NPSH low-low / critical and suction-pressure trips all fall back to
 * low_suction_pressure_trip = 0.1; discharge 10.0; vibration 10, bearing 120, motor 130; delays 5/10/30/60 s. */
NPD_FN void npd_fw_pump_step(npb_pump_t *p, npb_fw_t *fw, npd_fw_acc_t *acc, int i, int n_prev_running,
                             double flow_per_pump, const npd_pump_sysconds_t *sc, double dt) {
  if (p->status == NPD_PUMP_RUNNING && n_prev_running > 0) {
    if (acc->running_count < n_prev_running) {
      if (!(flow_per_pump < NPD_PUMP_RATED_FLOW * 0.2)) npd_pump_set_flow_demand(p, flow_per_pump);
    }
  }
  npd_pump_update(p, sc, dt);
  if (p->status == NPD_PUMP_RUNNING) {
    acc->total_flow += p->flow_rate; acc->total_power += p->power_consumption;
    acc->running_count++; acc->running_mask |= 1 << i;
  }
  if (p->trip_active) acc->trip_mask |= 1u << i;
  acc->flow_sum += p->flow_rate;

  /* diagnostics: shared cavitation monitor, dt "hours" = simulator dt */
  {
    double npsh_required = npd_pump_npsh_required(p);
    double cavitation_threshold = npsh_required + 2.0;
    double current_intensity;
    if (p->npsh_available < cavitation_threshold) {
      double npsh_deficit = cavitation_threshold - p->npsh_available;
      double severity = npd_pymin(1.0, npsh_deficit / cavitation_threshold);
      double flow_factor = npd_sq(p->flow_rate / 555.0);
      double speed_factor = npd_powc(p->speed_percent / 100.0, 1.5);
      current_intensity = severity * flow_factor * speed_factor;
      fw->cav_time_in_cavitation += dt;
      if (current_intensity > 0.1) { fw->cav_events_count += 1; if (fw->cav_events_count > 100) fw->cav_events_count = 100; }
    } else {
      current_intensity = 0.0;
    }
    if (current_intensity > 0.1) fw->cav_accumulated_damage += (npd_sq(current_intensity) * 0.01) * dt;
    double intensity_risk = npd_pymin(1.0, current_intensity / 0.5);
    double damage_risk = npd_pymin(1.0, fw->cav_accumulated_damage / 10.0);
    double frequency_risk = npd_pymin(1.0, fw->cav_events_count / 50.0);
    acc->total_cavitation_risk += (intensity_risk * 0.4 + damage_risk * 0.4 + frequency_risk * 0.2);
    double max_bearing = npd_pymax3(p->wear_motor_bearings, p->wear_pump_bearings, p->wear_thrust_bearing);
    acc->total_wear_level += (max_bearing + p->wear_mechanical_seals);
    acc->total_vibration += p->vibration_level;
  }
  /* protection, per-pump loops */
  double dt_seconds = dt * 60.0;
  {
    double npsh = p->npsh_available;
    int critical_active = 0;
    if (npsh < 0.1) {
      fw->npsh_low_low_timer += dt_seconds;
      if (fw->npsh_low_low_timer >= 5.0) fw->npsh_low_low_trip_active = 1;
    } else {
      fw->npsh_low_low_timer = 0.0;
      fw->npsh_low_low_trip_active = 0;
    }
    if (npsh < 0.1) critical_active = 1;
    if (critical_active || fw->npsh_low_low_trip_active) acc->trips++;
    if (critical_active) acc->trip_kinds |= 1;                       /* '<pump>_npsh_critical' */
  }
  if (p->suction_pressure < 0.1) { acc->trips++; acc->trip_kinds |= 1; }   /* '<pump>_suction_pressure_low' */
  if (p->discharge_pressure > 10.0) acc->trips++;
  if (p->vibration_level > 10.0) { fw->timer_vibration += dt_seconds; if (fw->timer_vibration >= 10.0) acc->trips++; }
  else fw->timer_vibration = 0.0;
  double bearing_temp = p->oil_temperature + 5.0; /* pump_system.py:477 */
  if (bearing_temp > 120.0) { fw->timer_bearing_temp += dt_seconds; if (fw->timer_bearing_temp >= 30.0) acc->trips++; }
  else fw->timer_bearing_temp = 0.0;
  if (p->motor_temperature > 130.0) { fw->timer_motor_temp += dt_seconds; if (fw->timer_motor_temp >= 60.0) acc->trips++; }
  else fw->timer_motor_temp = 0.0;
}

/* state-log diagnostics: the alarms FeedwaterProtectionSystem.check_protection_systems collects next to its trips
 * (protection_system.py:399-445; setpoints through the same getattr fallbacks: NPSH 18 m, suction 0.3 MPa, discharge 9 MPa,
 * vibration 5, bearing 80 C, motor 100 C; flow 2 x the low trip and 0.9 x the high trip; SG level 15.5 / 11 m; health 0.5,
 * cavitation risk 0.5, wear 50 %).  Where the reference writes if-trip / elif-alarm the alarm is not raised beside the trip. */
NPD_FN int npd_fw_pump_alarms(const npb_pump_t *p) {
  int n = 0;
  n += p->npsh_available < 18.0;
  n += !(p->suction_pressure < 0.1) && p->suction_pressure < 0.3;
  n += !(p->discharge_pressure > 10.0) && p->discharge_pressure > 9.0;
  n += p->vibration_level > 5.0;
  n += (p->oil_temperature + 5.0) > 80.0;
  n += p->motor_temperature > 100.0;
  return n;
}
NPD_FN int npd_fw_system_alarms(const npb_fw_t *fw, const npd_fw_acc_t *acc, const double *sg_levels) {
  int n = 0;
  n += acc->flow_sum < (0.05 * 1500.0) * 2.0;
  n += acc->flow_sum > (1.3 * 1500.0) * 0.9;
#pragma unroll
  for (int i = 0; i < NPB_NUM_SG; i++) {
    n += !(sg_levels[i] > 16.5) && sg_levels[i] > 15.5;
    n += sg_levels[i] < 11.0;     /* "low critical" below 10 m or "low" below 11 m: one alarm either way */
  }
  const double avg_cavitation_risk = acc->total_cavitation_risk / NPB_NUM_PUMPS, avg_wear_level = acc->total_wear_level / NPB_NUM_PUMPS;
  n += !(fw->overall_health_score < 0.3) && fw->overall_health_score < 0.5;
  n += !(avg_cavitation_risk > 0.8) && avg_cavitation_risk > 0.5;
  n += !(avg_wear_level > 85.0) && avg_wear_level > 50.0;
  return n;
}

typedef struct npd_fw_result_t {
  double total_flow_rate, total_power_consumption;
  int system_availability, num_running_pumps;
  uint32_t pump_trip_mask;
  int active_trips, trip_kinds;   /* state-log diagnostics: len(active_trips) and the kinds of npd_fw_acc_t.trip_kinds */
} npd_fw_result_t;

/* system-level tail: _calculate_health_score performance_monitoring.py:544, flow / SG-level /
 * diagnostic protection protection_system.py:483-678 (flow trips 0.05*1500 and 1.3*1500 with
 * 10 s / 2 s delays, SG level 16.5 m), availability feedwater/physics.py:786-788 */
NPD_FN void npd_fw_finish(npb_fw_t *fw, npd_fw_acc_t *acc, const double *sg_levels, double dt, npd_fw_result_t *res) {
  fw->running_mask = acc->running_mask;
  int pump_system_available = acc->running_count >= 3;
  double avg_cavitation_risk = acc->total_cavitation_risk / NPB_NUM_PUMPS;
  double avg_wear_level = acc->total_wear_level / NPB_NUM_PUMPS;
  double avg_vibration = acc->total_vibration / NPB_NUM_PUMPS;
  double cavitation_health = npd_pymax(0.0, 1.0 - avg_cavitation_risk);
  double wear_health = npd_pymax(0.0, 1.0 - avg_wear_level / 50.0);
  double vibration_health = npd_pymax(0.0, 1.0 - avg_vibration / 10.0);
  fw->overall_health_score = (cavitation_health * 0.3 + wear_health * 0.4 + vibration_health * 0.2 + 1.0 * 0.1);
  double dt_seconds = dt * 60.0;
  int trips = acc->trips;
  {
    double low_flow_trip = 0.05 * 1500.0, high_flow_trip = 1.3 * 1500.0;
    if (acc->flow_sum < low_flow_trip) { fw->timer_low_flow += dt_seconds; if (fw->timer_low_flow >= 10.0) { trips++; acc->trip_kinds |= 1; } }
    else fw->timer_low_flow = 0.0;
    if (acc->flow_sum > high_flow_trip) { fw->timer_high_flow += dt_seconds; if (fw->timer_high_flow >= 2.0) { trips++; acc->trip_kinds |= 2; } }
    else fw->timer_high_flow = 0.0;
  }
#pragma unroll
  for (int i = 0; i < NPB_NUM_SG; i++) if (sg_levels[i] > 16.5) { trips++; acc->trip_kinds |= 2; }
  if (fw->overall_health_score < 0.3) trips++;
  if (avg_cavitation_risk > 0.8) trips++;
  if (avg_wear_level > 85.0) trips++;
  fw->system_trip_active = trips > 0;
  fw->total_flow_rate = acc->total_flow;
  fw->total_power_consumption = acc->total_power;
  fw->system_availability = pump_system_available && !fw->system_trip_active;
  res->total_flow_rate = acc->total_flow; res->total_power_consumption = acc->total_power;
  res->system_availability = fw->system_availability; res->num_running_pumps = acc->running_count;
  res->pump_trip_mask = acc->trip_mask;
  res->active_trips = trips; res->trip_kinds = acc->trip_kinds;
}

#endif
