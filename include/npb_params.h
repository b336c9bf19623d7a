/*
 * npb_params.h -- plant constants of the batched stepper (public ABI, part of NpbConfig).
 *
 * These are the values the reference keeps in its config dataclasses AFTER
 * SecondarySystemConfig._synchronize_subsystems() has run
 * (systems/secondary/config.py:222-257), i.e. what the live objects actually
 * hold, not the class defaults (e.g. design_total_steam_flow is 1500, not 1665).
 * P(name, default, "reference attribute it mirrors").
 * All plants of one handle share one parameter block (broadcast from constant
 * memory on the device); per-plant variation goes through state fields.
 */
#ifndef NPB_PARAMS_H
#define NPB_PARAMS_H

#define NPB_PARAM_LIST(P) \
  /* ---- primary (systems/primary/__init__.py:174-176, constant_heat_source.py:29-33) */ \
  P(rated_power_mw,            3000.0,  "primary_physics.rated_power_mw") \
  P(max_control_rod_speed,     5.0,     "primary_physics.max_control_rod_speed") \
  P(max_valve_speed,           10.0,    "primary_physics.max_valve_speed") \
  P(max_flow_change_rate,      1000.0,  "primary_physics.max_flow_change_rate") \
  P(hs_noise_std_percent,      0.1,     "primary_physics.heat_source.noise_std_percent") \
  P(hs_noise_filter_tau,       30.0,    "primary_physics.heat_source.noise_filter_time_constant") \
  /* ---- steam generators (steam_generator/config.py:179-260, live values) */ \
  P(sg_design_total_steam_flow,   1500.0,   "secondary_physics.steam_generator_system.config.design_total_steam_flow") \
  P(sg_design_thermal_power_per_sg, 1000.0e6, "secondary_physics.steam_generator_system.steam_generators[0].config.design_thermal_power_per_sg") \
  P(sg_design_steam_flow_per_sg,  500.0,    "secondary_physics.steam_generator_system.steam_generators[0].config.design_steam_flow_per_sg") \
  P(sg_design_feedwater_flow_per_sg, 500.0, "secondary_physics.steam_generator_system.steam_generators[0].config.design_feedwater_flow_per_sg") \
  P(sg_primary_design_flow,       5700.0,   "secondary_physics.steam_generator_system.steam_generators[0].config.primary_design_flow") \
  P(sg_secondary_design_flow,     500.0,    "secondary_physics.steam_generator_system.steam_generators[0].config.secondary_design_flow") \
  P(sg_heat_transfer_area,        5000.0,   "secondary_physics.steam_generator_system.steam_generators[0].config.heat_transfer_area_per_sg") \
  P(sg_tube_count,                3388.0,   "secondary_physics.steam_generator_system.steam_generators[0].config.tube_count_per_sg") \
  P(sg_tube_inner_diameter,       0.0191,   "secondary_physics.steam_generator_system.steam_generators[0].config.tube_inner_diameter") \
  P(sg_tube_wall_thickness,       0.00109,  "secondary_physics.steam_generator_system.steam_generators[0].config.tube_wall_thickness") \
  P(sg_secondary_water_mass,      68000.0,  "secondary_physics.steam_generator_system.steam_generators[0].config.secondary_water_mass") \
  P(sg_primary_htc,               28000.0,  "secondary_physics.steam_generator_system.steam_generators[0].config.primary_htc") \
  P(sg_secondary_htc,             18000.0,  "secondary_physics.steam_generator_system.steam_generators[0].config.secondary_htc") \
  P(sg_design_pressure_secondary, 6.895,    "secondary_physics.steam_generator_system.steam_generators[0].config.design_pressure_secondary") \
  P(sg_tube_conductivity,         385.0,    "secondary_physics.steam_generator_system.steam_generators[0].config.tube_material_conductivity") \
  /* ---- SG-system-owned WaterChemistry: constructed, never updated (enhanced_physics.py:67-71) */ \
  P(sgchem_iron,                  0.1,      "secondary_physics.steam_generator_system.water_chemistry.iron_concentration") \
  P(sgchem_copper,                0.05,     "secondary_physics.steam_generator_system.water_chemistry.copper_concentration") \
  P(sgchem_silica,                20.0,     "secondary_physics.steam_generator_system.water_chemistry.silica_concentration") \
  P(sgchem_ph,                    9.2,      "secondary_physics.steam_generator_system.water_chemistry.ph") \
  P(sgchem_dissolved_oxygen,      0.005,    "secondary_physics.steam_generator_system.water_chemistry.dissolved_oxygen") \
  /* ---- automatic maintenance of the feedwater pumps (used only when maint_enabled; values of the data-gen \
   *      action-test scenario: auto_maintenance.py:121-160,187-198 aggressive mode = no start delays).  The threshold \
   *      table is separate (include/npb_maint.h, npb_set_maintenance_table); the two oil_level entries here \
   *      override the table's oil_level row (kept from ABI version 1) */ \
  P(maint_check_interval_hours,   0.25,     "maintenance_system.check_interval_hours") \
  P(maint_oil_level_threshold,    58.0,     "state_manager.maintenance_thresholds['FWP-1']['oil_level']['threshold']") \
  P(maint_oil_level_cooldown_hours, 168.0,  "state_manager.maintenance_thresholds['FWP-1']['oil_level']['cooldown_hours']") \
  P(maint_work_order_cooldown,    24.0,     "maintenance_system.work_order_cooldown_hours") \
  P(maint_start_delay_hours,      0.0,      "maintenance_system.high_priority_delay_hours") \
  P(maint_top_off_target,         95.0,     "") \
  P(maint_medium_delay_hours,     0.0,      "maintenance_system.medium_priority_delay_hours") \
  P(maint_low_delay_hours,        0.0,      "maintenance_system.low_priority_delay_hours") \
  P(maint_emergency_delay_hours,  0.0,      "maintenance_system.emergency_delay_hours")

typedef struct npb_params_t {
#define NPB__P(name, dflt, path) double name;
  NPB_PARAM_LIST(NPB__P)
#undef NPB__P
  /* run-mode switches (not reference attributes) */
  double dt;               /* NuclearPlantSimulator(dt=...)  sim.py:33 */
  int heat_source;         /* NPB_HEAT_CONSTANT | NPB_HEAT_REACTOR | NPB_HEAT_EXTERNAL */
  int hs_noise_enabled;    /* ConstantHeatSource(noise_enabled=...) */
  int mode;                /* NPB_MODE_FULL | NPB_MODE_PRIMARY_SG | NPB_MODE_PRIMARY */
  int maint_enabled;       /* 1: run the automatic maintenance of the feedwater pumps after every step (maint.*, mpump.* columns) */
  int info_reactivity_components; /* 1: under NPB_HEAT_REACTOR the step also writes info["reactivity_components"] (npb.h NPB_RHO_*) */
  int kinetics_rk4_substeps;      /* 0: the reference's clipped explicit-Euler point kinetics (the parity path).  n > 0: the
                                   * point-kinetics equations themselves (flux + six precursor groups, no rate clips) advanced by n
                                   * classical RK4 sub-steps per dt inside the step kernel -- BASELINE config 2's "rk4" mode; the
                                   * reference has no such integrator, so this mode is self-consistency-tested only */
} npb_params_t;

/* NPB_HEAT_EXTERNAL: a heat source the caller computes (the reference's HeatSource plugin interface,
 * heat_sources/heat_source_interface.py:23-112, consumed at primary/__init__.py:203-225): the step takes the plugin's result as
 * two per-step input columns -- under this mode npb_step's `noise_z` column IS heat_result['thermal_power_mw'] and its
 * `power_setpoint` column IS heat_result['power_percent'] (NULL / NaN: thermal power / rated power x 100) -- sets
 * total_reactivity_pcm = 0 (a result without 'reactivity_pcm', :218-222) and leaves neutron_flux alone (:214-215). */
enum { NPB_HEAT_CONSTANT = 0, NPB_HEAT_REACTOR = 1, NPB_HEAT_EXTERNAL = 2 };
/* NPB_MODE_PRIMARY: NuclearPlantSimulator(enable_secondary=False), sim.py:155,309,333 -- primary side only, 12 observations */
enum { NPB_MODE_FULL = 0, NPB_MODE_PRIMARY_SG = 1, NPB_MODE_PRIMARY = 2 };

static inline void npb_params_default(npb_params_t *p) {
#define NPB__P(name, dflt, path) p->name = (dflt);
  NPB_PARAM_LIST(NPB__P)
#undef NPB__P
  p->dt = 1.0;
  p->heat_source = NPB_HEAT_CONSTANT;
  p->hs_noise_enabled = 0;
  p->mode = NPB_MODE_FULL;
  p->maint_enabled = 0;
  p->info_reactivity_components = 0;
  p->kinetics_rk4_substeps = 0;
}

#endif /* NPB_PARAMS_H */
