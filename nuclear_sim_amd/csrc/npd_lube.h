/*
 * npd_lube.h -- device physics: BaseLubricationSystem oil-quality kinetics shared by the
 * feedwater-pump and turbine-bearing lubrication systems.
 * Follows systems/secondary/lubrication_base.py:186-352 (update_oil_quality).
 */
#ifndef NPD_LUBE_H
#define NPD_LUBE_H
#include "npd_common.h"

/* the oil-state scalars update_oil_quality reads/writes, held by VALUE (no pointers into other
 * register structs: address-taken locals would be forced out of registers) */
typedef struct npd_oil_t {
  double temperature, contamination, moisture, acidity, viscosity_change;
  double antioxidant, anti_wear, corrosion_inhibitor, effectiveness;
} npd_oil_t;

typedef struct npd_oil_limits_t {
  double contamination_limit, acidity_limit, moisture_limit, viscosity_change_limit;
} npd_oil_limits_t;

/* avg_wear = mean of the system's component_wear dict (lubrication_base.py:262-264) */
NPD_FN void npd_update_oil_quality(npd_oil_t *o, const npd_oil_limits_t *lim, double avg_wear,
                                   double operating_temperature, double contamination_input,
                                   double moisture_input, double dt) {
  double temp_change = (operating_temperature - o->temperature) / 0.5 * dt;
  double max_temp_change = 10.0 * dt;
  temp_change = npd_pymax(-max_temp_change, npd_pymin(max_temp_change, temp_change));
  o->temperature += temp_change;
  o->temperature = npd_pymax(20.0, npd_pymin(120.0, o->temperature));
  double temp_diff = o->temperature - 60.0;
  temp_diff = npd_pymax(-50.0, npd_pymin(200.0, temp_diff));
  double activation_factor = npd_pymax(0.1, npd_pymin(1.5, 1.0 + temp_diff / 50.0));
  double thermal_degradation_rate = 0.00001 * activation_factor;
  double filter_loading_factor = npd_pymax(0.3, 1.0 - (o->contamination / 50.0));
  double temp_factor = npd_pymax(0.5, 1.0 - (o->temperature - 60.0) / 60.0);
  double effective_filtration_efficiency = 0.60 * filter_loading_factor * temp_factor;
  double contamination_removal_rate = o->contamination * effective_filtration_efficiency * 0.005;
  double base_thermal_contamination = thermal_degradation_rate * 0.75;
  double wear_contamination_factor = 1.0 + (avg_wear / 20.0);
  double thermal_contamination_input = base_thermal_contamination * wear_contamination_factor;
  double contamination_change = contamination_input - contamination_removal_rate + thermal_contamination_input;
  o->contamination += contamination_change * dt;
  o->contamination = npd_pymax(1.0, o->contamination);
  double moisture_change;
  if (o->temperature > 70.0) moisture_change = moisture_input - (o->temperature - 70.0) * 0.001;
  else moisture_change = moisture_input;
  o->moisture += moisture_change * dt;
  o->moisture = npd_pymax(0.001, o->moisture);
  double contamination_factor = 1.0 + o->contamination / 50.0;
  double acidity_increase_rate = thermal_degradation_rate * contamination_factor * 0.1;
  o->acidity += acidity_increase_rate * dt;
  double viscosity_change_rate = thermal_degradation_rate * 0.5 + contamination_change * 0.01;
  o->viscosity_change += viscosity_change_rate * dt;
  double antioxidant_consumption_rate = thermal_degradation_rate * 10.0;
  o->antioxidant = npd_pymax(0.0, o->antioxidant - antioxidant_consumption_rate * dt * 100.0);
  double aw_consumption_rate = (contamination_input * 0.1) * 0.5;
  o->anti_wear = npd_pymax(0.0, o->anti_wear - aw_consumption_rate * dt * 100.0);
  double ci_consumption_rate = o->moisture * 2.0;
  o->corrosion_inhibitor = npd_pymax(0.0, o->corrosion_inhibitor - ci_consumption_rate * dt * 100.0);
  double cf = npd_pymax(0.1, 1.0 - o->contamination / lim->contamination_limit);
  double af = npd_pymax(0.1, 1.0 - o->acidity / lim->acidity_limit);
  double mf = npd_pymax(0.1, 1.0 - o->moisture / lim->moisture_limit);
  double vf = npd_pymax(0.1, 1.0 - fabs(o->viscosity_change) / lim->viscosity_change_limit);
  double antioxidant_factor = o->antioxidant / 100.0;
  double aw_factor = o->anti_wear / 100.0;
  double critical = npd_powc(cf * antioxidant_factor * aw_factor, 1.0 / 3);
  double secondary = (0.0 + af + mf + vf) / 3;
  double eff = critical * 0.7 + secondary * 0.3;
  o->effectiveness = npd_pymax(0.3, npd_pymin(1.0, eff));
}

#endif
