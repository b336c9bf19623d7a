/*
 * npd_chem.h -- device physics: WaterChemistry (one routine, two live instances).
 * Follows systems/secondary/water_chemistry.py:277-438 (update_chemistry and helpers).
 */
#ifndef NPD_CHEM_H
#define NPD_CHEM_H
#include "npd_common.h"

/* _calculate_composite_parameters  water_chemistry.py:277-320 (iron 0.1, alkalinity 120 never change) */
NPD_FN void npd_chem_composites(npb_chem_t *c) {
  double iron_effect = 0.1 * 0.5;
  double chloride_effect = c->chloride / 100.0;
  double ph_effect = fabs(c->ph - 7.0) * 0.2;
  double hardness_effect = npd_pymax(0.0, (c->hardness - 150.0) / 150.0) * 0.3;
  c->water_aggressiveness = npd_clip(1.0 + iron_effect + chloride_effect + ph_effect + hardness_effect, 0.5, 3.0);
  double A = (npd_log10(c->total_dissolved_solids) - 1) / 10;
  double B = -13.12 * log10(25.0 + 273) + 34.55;
  double C = npd_log10(c->hardness) - 0.4;
  double D = log10(120.0);
  double ph_saturation = (9.3 + A + B) - (C + D);
  c->scaling_tendency = c->ph - ph_saturation;
}

/* WaterChemistry.update_chemistry  water_chemistry.py:322-389 with the makeup-water dict both callers
 * pass (tds 300, hardness 100, chloride 30, ph 7.2, dissolved_oxygen 8.0) and blowdown_rate 0.02.
 * Pending pH-control effects of the shared instance are applied by the caller (npd_ph.h). */
NPD_FN void npd_chem_update(npb_chem_t *c, double dt) {
  double dt_hours;
  if (dt > 100) dt_hours = dt / 3600.0;       /* :335-344 unit guess by magnitude */
  else if (dt > 1) dt_hours = dt / 60.0;
  else dt_hours = dt;
  /* _update_from_makeup_water :391-414 */
  double blend_factor = npd_pymin(0.05 * dt_hours * 0.1, 0.5);
  c->ph += (7.2 - c->ph) * blend_factor;
  c->hardness += (100.0 - c->hardness) * blend_factor;
  c->total_dissolved_solids += (300.0 - c->total_dissolved_solids) * blend_factor;
  c->chloride += (30.0 - c->chloride) * blend_factor;
  c->dissolved_oxygen = 8.0 * 0.8;
  /* concentration effects :369-380 */
  double concentration_factor = 1.0 / (0.02 + 0.01);
  concentration_factor = npd_pymin(concentration_factor, 5.0);
  if (concentration_factor > 1.1) {
    double concentration_increase = (concentration_factor - 1.0) * 0.1 * dt_hours;
    c->total_dissolved_solids += concentration_increase * 50.0;
    c->hardness += concentration_increase * 10.0;
    c->chloride += concentration_increase * 5.0;
  }
  /* _update_chemical_treatment :416-438 */
  double dose_rate = 0.5 * dt_hours;
  c->antiscalant_concentration += (5.0 - c->antiscalant_concentration) * dose_rate;
  c->corrosion_inhibitor_level += (10.0 - c->corrosion_inhibitor_level) * dose_rate;
  double chlorine_decay = 0.1 * dt_hours;
  c->chlorine_residual *= npd_exp_bounded(-chlorine_decay);
  c->chlorine_residual += (1.0 - c->chlorine_residual) * dose_rate;
  double chlorine_effectiveness = (c->chlorine_residual > 0.2) ? 1.0 : 0.5;
  double antiscalant_effectiveness = (c->antiscalant_concentration > 2.0) ? 1.0 : 0.7;
  double corrosion_effectiveness = (c->corrosion_inhibitor_level > 5.0) ? 1.0 : 0.8;
  c->treatment_efficiency = (chlorine_effectiveness * antiscalant_effectiveness * corrosion_effectiveness * 0.95);
  npd_chem_composites(c);
}

#endif
