/*
 * npb_fields.h -- state schema of the batched plant stepper (public ABI vocabulary).
 *
 * One table drives everything that has to agree on "what is a plant's state":
 *   - the struct-of-arrays layout in HBM (slot index -> column of N plants),
 *   - the field ids accepted by npb_get_field()/npb_set_field() (include/npb.h),
 *   - the per-subsystem register structs the HIP kernel streams through,
 *   - the CPU oracle's array-of-structs plant record (oracle/),
 *   - the test harness that compares every field with the reference's own
 *     attribute of the same meaning (the quoted path, relative to the
 *     reference's NuclearPlantSimulator object; "" = no single reference leaf).
 *
 * X-macro kinds (each section macro takes F, A, I):
 *   F(name, "path")          one fp64 scalar
 *   A(name, count, "path")   fp64 array; path uses {k} for the element index
 *   I(name, "path")          one int32 (flags, enums, counters)
 * Inside a section all fp64 members precede all int32 members.
 *
 * Output members.  The last NPB_<SEC>_NOUT fp64 members of a section are OUTPUTS of the step: the step kernel
 * overwrites them without ever reading the old value (established on the compiled kernel itself by
 * tools/probe_liveness.py: a marker on every loaded member, deleted by the compiler where the value is unused).
 * They are kept for get_observation(), field reads and the reference's attribute of the same name, but they are
 * not carried state: the arena stores them as float, packed two to an 8-byte column together with the int32
 * members ("narrow" members; 6e-8 relative rounding against the 1e-6 parity budget).  That takes ~270 B per
 * plant out of the working set, which at 65 536 plants is what decides whether a step's data fits the 256 MB
 * Infinity Cache.
 * Path placeholders: {i} = 0-based instance, {j} = 1-based instance, {k} = array index.
 *
 * Only CARRIED state (read before it is overwritten in the next step) and the
 * handful of outputs needed by get_observation() live here; everything else is
 * recomputed inside the step.  The algorithmic-bytes figure used for the
 * roofline is derived from this table (npb_state_bytes()).
 */
#ifndef NPB_FIELDS_H
#define NPB_FIELDS_H

#include <stdint.h>
#include <stddef.h>

#define NPB_NUM_SG 3
#define NPB_NUM_TSP 7
#define NPB_NUM_PUMPS 4
#define NPB_NUM_STAGES 14
#define NPB_NUM_BEARINGS 4
#define NPB_NUM_EJECTORS 2

/* ---- primary side: ReactorState + heat source + simulator-level carried scalars
 * reference: systems/primary/__init__.py:48-106, heat_sources/constant_heat_source.py:47-66,
 *            simulator/core/sim.py:391-399,495 */
/* the NPB_PRIM_NKIN carried members before the outputs (reactivity ... fuel_burnup) belong to the point-kinetics model:
 * only ReactorHeatSource reads or writes them, so under ConstantHeatSource the step kernel neither loads nor stores
 * their columns (they keep their values) */
#define NPB_PRIM_NKIN 12
#define NPB_PRIM_NOUT 3   /* the last 3 fp64 members are outputs of the step (see "output members" above) */
#define NPB_PRIM_FIELDS(F, A, I) \
  F(neutron_flux,          "primary_physics.state.neutron_flux") \
  F(fuel_temperature,      "primary_physics.state.fuel_temperature") \
  F(coolant_temperature,   "primary_physics.state.coolant_temperature") \
  F(coolant_pressure,      "primary_physics.state.coolant_pressure") \
  F(coolant_flow_rate,     "primary_physics.state.coolant_flow_rate") \
  F(coolant_void_fraction, "primary_physics.state.coolant_void_fraction") \
  F(steam_temperature,     "primary_physics.state.steam_temperature") \
  F(steam_pressure,        "primary_physics.state.steam_pressure") \
  F(steam_flow_rate,       "primary_physics.state.steam_flow_rate") \
  F(feedwater_flow_rate,   "primary_physics.state.feedwater_flow_rate") \
  F(control_rod_position,  "primary_physics.state.control_rod_position") \
  F(steam_valve_position,  "primary_physics.state.steam_valve_position") \
  F(boron_concentration,   "primary_physics.state.boron_concentration") \
  F(hs_setpoint_percent,   "primary_physics.heat_source.power_setpoint_percent") \
  F(hs_filtered_noise_mw,  "primary_physics.heat_source.filtered_noise_mw") \
  F(last_heat_removal_factor, "_last_heat_removal_factor") \
  F(sim_time,              "time") \
  F(reactivity,            "primary_physics.state.reactivity") \
  A(precursors, 6,         "primary_physics.state.delayed_neutron_precursors[{k}]") \
  F(xenon_concentration,   "primary_physics.state.xenon_concentration") \
  F(iodine_concentration,  "primary_physics.state.iodine_concentration") \
  F(samarium_concentration,"primary_physics.state.samarium_concentration") \
  F(burnable_poison_worth, "primary_physics.state.burnable_poison_worth") \
  F(fuel_burnup,           "primary_physics.state.fuel_burnup") \
  F(power_level,           "primary_physics.state.power_level") \
  F(thermal_power_mw,      "primary_physics.thermal_power_mw") \
  F(total_reactivity_pcm,  "primary_physics.total_reactivity_pcm") \
  I(scram_status,          "primary_physics.state.scram_status") \
  I(has_heat_removal_factor, "=float(hasattr(root, '_last_heat_removal_factor'))")

/* ---- one U-tube steam generator (x3)
 * reference: steam_generator/steam_generator.py:87-112, tsp_fouling_model.py:126-147,175-190,
 *            tube_interior_fouling.py:67-79, fouling_model_base.py:84-92 */
#define NPB_SG_NOUT 5   /* the last 5 fp64 members are outputs of the step (see "output members" above) */
#define NPB_SG_FIELDS(F, A, I) \
  F(secondary_pressure,   "secondary_physics.steam_generator_system.steam_generators[{i}].secondary_pressure") \
  F(steam_quality,        "secondary_physics.steam_generator_system.steam_generators[{i}].steam_quality") \
  F(water_level,          "secondary_physics.steam_generator_system.steam_generators[{i}].water_level") \
  A(tsp_magnetite, 7,     "secondary_physics.steam_generator_system.steam_generators[{i}].tsp_fouling.deposits.magnetite_thickness[{k}]") \
  A(tsp_copper, 7,        "secondary_physics.steam_generator_system.steam_generators[{i}].tsp_fouling.deposits.copper_thickness[{k}]") \
  A(tsp_silica, 7,        "secondary_physics.steam_generator_system.steam_generators[{i}].tsp_fouling.deposits.silica_thickness[{k}]") \
  A(tsp_biological, 7,    "secondary_physics.steam_generator_system.steam_generators[{i}].tsp_fouling.deposits.biological_thickness[{k}]") \
  F(tsp_fouling_fraction, "secondary_physics.steam_generator_system.steam_generators[{i}].tsp_fouling.fouling_fraction") \
  F(tsp_ht_degradation,   "secondary_physics.steam_generator_system.steam_generators[{i}].tsp_fouling.heat_transfer_degradation") \
  F(tsp_operating_years,  "secondary_physics.steam_generator_system.steam_generators[{i}].tsp_fouling.operating_years") \
  F(scale_thickness,      "secondary_physics.steam_generator_system.steam_generators[{i}].tube_interior_fouling.scale_thickness") \
  F(scale_iron_oxide,     "secondary_physics.steam_generator_system.steam_generators[{i}].tube_interior_fouling.scale_composition['iron_oxide']") \
  F(scale_crud,           "secondary_physics.steam_generator_system.steam_generators[{i}].tube_interior_fouling.scale_composition['crud_deposits']") \
  F(scale_corrosion,      "secondary_physics.steam_generator_system.steam_generators[{i}].tube_interior_fouling.scale_composition['corrosion_products']") \
  F(scale_thermal_resistance, "secondary_physics.steam_generator_system.steam_generators[{i}].tube_interior_fouling.scale_thermal_resistance") \
  F(scale_operating_years,"secondary_physics.steam_generator_system.steam_generators[{i}].tube_interior_fouling.operating_years") \
  F(steam_flow_rate,      "secondary_physics.steam_generator_system.steam_generators[{i}].steam_flow_rate") \
  F(secondary_temperature,"secondary_physics.steam_generator_system.steam_generators[{i}].secondary_temperature") \
  F(heat_transfer_rate,   "secondary_physics.steam_generator_system.steam_generators[{i}].heat_transfer_rate") \
  F(tube_wall_temp,       "secondary_physics.steam_generator_system.steam_generators[{i}].tube_wall_temp") \
  F(tsp_pressure_drop_ratio, "secondary_physics.steam_generator_system.steam_generators[{i}].tsp_fouling.pressure_drop_ratio") \
  I(tsp_shutdown_required,"secondary_physics.steam_generator_system.steam_generators[{i}].tsp_fouling.shutdown_required")

/* ---- one feedwater pump with its lubrication system (x4: 3 running + 1 spare)
 * reference: feedwater/pump_system.py:62-90 (FeedwaterPumpState), primary/coolant/pump_models.py:30-48,
 *            feedwater/pump_lubrication.py:204-215, lubrication_base.py:150-176
 * status codes follow PumpStatus order (pump_models.py:21-27): 0 RUNNING 1 STOPPED 2 STARTING 3 STOPPING 4 TRIPPED */
#define NPB_PUMP_NOUT 6   /* the last 6 fp64 members are outputs of the step (see "output members" above) */
#define NPB_PUMP_FIELDS(F, A, I) \
  F(speed_percent,      "secondary_physics.feedwater_system.pump_system.pumps['FWP-{j}'].state.speed_percent") \
  F(speed_setpoint,     "secondary_physics.feedwater_system.pump_system.pumps['FWP-{j}'].state.speed_setpoint") \
  F(flow_rate,          "secondary_physics.feedwater_system.pump_system.pumps['FWP-{j}'].state.flow_rate") \
  F(power_consumption,  "secondary_physics.feedwater_system.pump_system.pumps['FWP-{j}'].state.power_consumption") \
  F(flow_demand,        "secondary_physics.feedwater_system.pump_system.pumps['FWP-{j}'].flow_demand") \
  F(suction_pressure,   "secondary_physics.feedwater_system.pump_system.pumps['FWP-{j}'].state.suction_pressure") \
  F(discharge_pressure, "secondary_physics.feedwater_system.pump_system.pumps['FWP-{j}'].state.discharge_pressure") \
  F(npsh_available,     "secondary_physics.feedwater_system.pump_system.pumps['FWP-{j}'].state.npsh_available") \
  F(differential_pressure, "secondary_physics.feedwater_system.pump_system.pumps['FWP-{j}'].state.differential_pressure") \
  F(cavitation_intensity, "secondary_physics.feedwater_system.pump_system.pumps['FWP-{j}'].state.cavitation_intensity") \
  F(cavitation_damage,  "secondary_physics.feedwater_system.pump_system.pumps['FWP-{j}'].state.cavitation_damage") \
  F(cavitation_time,    "secondary_physics.feedwater_system.pump_system.pumps['FWP-{j}'].state.cavitation_time") \
  F(oil_level,          "secondary_physics.feedwater_system.pump_system.pumps['FWP-{j}'].lubrication_system.oil_level") \
  F(oil_temperature,    "secondary_physics.feedwater_system.pump_system.pumps['FWP-{j}'].lubrication_system.oil_temperature") \
  F(oil_contamination,  "secondary_physics.feedwater_system.pump_system.pumps['FWP-{j}'].lubrication_system.oil_contamination_level") \
  F(oil_moisture,       "secondary_physics.feedwater_system.pump_system.pumps['FWP-{j}'].lubrication_system.oil_moisture_content") \
  F(oil_acidity,        "secondary_physics.feedwater_system.pump_system.pumps['FWP-{j}'].lubrication_system.oil_acidity_number") \
  F(oil_viscosity_change, "secondary_physics.feedwater_system.pump_system.pumps['FWP-{j}'].lubrication_system.oil_viscosity_change") \
  F(antioxidant_level,  "secondary_physics.feedwater_system.pump_system.pumps['FWP-{j}'].lubrication_system.antioxidant_level") \
  F(anti_wear_level,    "secondary_physics.feedwater_system.pump_system.pumps['FWP-{j}'].lubrication_system.anti_wear_additive_level") \
  F(corrosion_inhibitor_level, "secondary_physics.feedwater_system.pump_system.pumps['FWP-{j}'].lubrication_system.corrosion_inhibitor_level") \
  F(lubrication_effectiveness, "secondary_physics.feedwater_system.pump_system.pumps['FWP-{j}'].lubrication_system.lubrication_effectiveness") \
  F(wear_impeller,      "secondary_physics.feedwater_system.pump_system.pumps['FWP-{j}'].lubrication_system.component_wear['impeller']") \
  F(wear_motor_bearings,"secondary_physics.feedwater_system.pump_system.pumps['FWP-{j}'].lubrication_system.component_wear['motor_bearings']") \
  F(wear_pump_bearings, "secondary_physics.feedwater_system.pump_system.pumps['FWP-{j}'].lubrication_system.component_wear['pump_bearings']") \
  F(wear_thrust_bearing,"secondary_physics.feedwater_system.pump_system.pumps['FWP-{j}'].lubrication_system.component_wear['thrust_bearing']") \
  F(wear_mechanical_seals, "secondary_physics.feedwater_system.pump_system.pumps['FWP-{j}'].lubrication_system.component_wear['mechanical_seals']") \
  F(wear_coupling_system, "secondary_physics.feedwater_system.pump_system.pumps['FWP-{j}'].lubrication_system.component_wear['coupling_system']") \
  F(flow_degradation,   "secondary_physics.feedwater_system.pump_system.pumps['FWP-{j}'].lubrication_system.pump_flow_degradation") \
  F(motor_temperature,  "secondary_physics.feedwater_system.pump_system.pumps['FWP-{j}'].state.motor_temperature") \
  F(vibration_level,    "secondary_physics.feedwater_system.pump_system.pumps['FWP-{j}'].state.vibration_level") \
  F(efficiency_degradation, "secondary_physics.feedwater_system.pump_system.pumps['FWP-{j}'].lubrication_system.pump_efficiency_degradation") \
  F(head_degradation,   "secondary_physics.feedwater_system.pump_system.pumps['FWP-{j}'].lubrication_system.pump_head_degradation") \
  F(vibration_increase, "secondary_physics.feedwater_system.pump_system.pumps['FWP-{j}'].lubrication_system.vibration_increase") \
  F(seal_leakage_rate,  "secondary_physics.feedwater_system.pump_system.pumps['FWP-{j}'].lubrication_system.seal_leakage_rate") \
  I(status,             "secondary_physics.feedwater_system.pump_system.pumps['FWP-{j}'].state.status") \
  I(available,          "secondary_physics.feedwater_system.pump_system.pumps['FWP-{j}'].state.available") \
  I(trip_active,        "secondary_physics.feedwater_system.pump_system.pumps['FWP-{j}'].state.trip_active") \
  I(trip_reason,        "=H.pump_trip_reason(root, {i})")

/* ---- feedwater system level: three-element control, shared cavitation monitor, protection timers
 * reference: feedwater/level_control.py:39-44,137-150, performance_monitoring.py:91-113,
 *            protection_system.py:40-57 and the trip_timers dict, feedwater/physics.py:157-161 */
#define NPB_FW_NOUT 3   /* the last 3 fp64 members are outputs of the step (see "output members" above) */
#define NPB_FW_FIELDS(F, A, I) \
  A(level_integral_errors, 3, "secondary_physics.feedwater_system.level_control.level_integral_errors[{k}]") \
  A(previous_level_errors, 3, "secondary_physics.feedwater_system.level_control.previous_level_errors[{k}]") \
  F(quality_integral_error,   "secondary_physics.feedwater_system.level_control.quality_compensator.quality_integral_error") \
  F(cav_accumulated_damage,   "secondary_physics.feedwater_system.diagnostics.cavitation_model.accumulated_damage") \
  F(cav_time_in_cavitation,   "secondary_physics.feedwater_system.diagnostics.cavitation_model.time_in_cavitation") \
  F(npsh_low_low_timer,       "secondary_physics.feedwater_system.protection_system.npsh_protection.npsh_low_low_timer") \
  F(timer_low_flow,           "secondary_physics.feedwater_system.protection_system.trip_timers['low_flow']") \
  F(timer_high_flow,          "secondary_physics.feedwater_system.protection_system.trip_timers['high_flow']") \
  F(timer_bearing_temp,       "secondary_physics.feedwater_system.protection_system.trip_timers['bearing_temp']") \
  F(timer_motor_temp,         "secondary_physics.feedwater_system.protection_system.trip_timers['motor_temp']") \
  F(timer_vibration,          "secondary_physics.feedwater_system.protection_system.trip_timers['vibration']") \
  F(total_flow_rate,          "secondary_physics.feedwater_system.total_flow_rate") \
  F(total_power_consumption,  "secondary_physics.feedwater_system.total_power_consumption") \
  F(overall_health_score,     "secondary_physics.feedwater_system.diagnostics.overall_health_score") \
  I(system_availability,      "secondary_physics.feedwater_system.system_availability") \
  I(running_mask,             "=sum(1 << (int(p[-1]) - 1) for p in root.secondary_physics.feedwater_system.pump_system.running_pumps)") \
  I(cav_events_count,         "=len(root.secondary_physics.feedwater_system.diagnostics.cavitation_model.cavitation_events)") \
  I(system_trip_active,       "secondary_physics.feedwater_system.protection_system.system_trip_active") \
  I(npsh_low_low_trip_active, "secondary_physics.feedwater_system.protection_system.npsh_protection.npsh_low_low_trip_active")

/* ---- turbine: 14 stages, rotor, 4 bearings, metal-temperature tracker, protection timers,
 * bearing lubrication system
 * reference: turbine/stage_system.py:49-96, rotor_dynamics.py:55-82,803-822,
 *            turbine/enhanced_physics.py:56-71,190-198,531-544, lubrication_base.py:150-176,
 *            turbine_bearing_lubrication.py:97-185
 * Per stage only efficiency_degradation, deposit_thickness and blade_wear_factor are stored:
 * fouling_factor, blade_condition_factor and actual_efficiency are pure functions of them
 * (stage_system.py:294-339) and are re-derived when the stage is loaded. */
#define NPB_TURB_NOUT 4   /* the last 4 fp64 members are outputs of the step (see "output members" above) */
#define NPB_TURB_FIELDS(F, A, I) \
  F(rotor_speed,          "secondary_physics.turbine.rotor_dynamics.rotor_speed") \
  F(rotor_temperature,    "secondary_physics.turbine.rotor_dynamics.rotor_temperature") \
  F(thermal_bow,          "secondary_physics.turbine.rotor_dynamics.thermal_bow") \
  A(bearing_load, 4,        "=list(root.secondary_physics.turbine.rotor_dynamics.bearings.values())[{k}].current_load") \
  A(bearing_metal_temp, 4,  "=list(root.secondary_physics.turbine.rotor_dynamics.bearings.values())[{k}].metal_temperature") \
  A(bearing_wear_factor, 4, "=list(root.secondary_physics.turbine.rotor_dynamics.bearings.values())[{k}].wear_factor") \
  F(timer_overspeed,      "secondary_physics.turbine.protection_system.trip_timers['overspeed']") \
  F(timer_vibration,      "secondary_physics.turbine.protection_system.trip_timers['vibration']") \
  F(timer_bearing_temp,   "secondary_physics.turbine.protection_system.trip_timers['bearing_temp']") \
  F(load_demand,          "secondary_physics.turbine.load_demand") \
  F(lub_oil_temperature,  "secondary_physics.turbine.bearing_lubrication_system.oil_temperature") \
  F(lub_oil_contamination,"secondary_physics.turbine.bearing_lubrication_system.oil_contamination_level") \
  F(lub_oil_moisture,     "secondary_physics.turbine.bearing_lubrication_system.oil_moisture_content") \
  F(lub_oil_acidity,      "secondary_physics.turbine.bearing_lubrication_system.oil_acidity_number") \
  F(lub_oil_viscosity_change, "secondary_physics.turbine.bearing_lubrication_system.oil_viscosity_change") \
  F(lub_antioxidant_level,"secondary_physics.turbine.bearing_lubrication_system.antioxidant_level") \
  F(lub_anti_wear_level,  "secondary_physics.turbine.bearing_lubrication_system.anti_wear_additive_level") \
  F(lub_corrosion_inhibitor_level, "secondary_physics.turbine.bearing_lubrication_system.corrosion_inhibitor_level") \
  A(lub_wear, 5,          "=list(root.secondary_physics.turbine.bearing_lubrication_system.component_wear.values())[{k}]") \
  F(thermal_expansion,    "secondary_physics.turbine.rotor_dynamics.thermal_expansion") \
  F(total_power_output,   "secondary_physics.turbine.total_power_output") \
  F(vibration_displacement, "secondary_physics.turbine.rotor_dynamics.vibration_monitor.displacement_x") \
  F(lub_effectiveness,    "secondary_physics.turbine.bearing_lubrication_system.lubrication_effectiveness") \
  I(trip_active,          "secondary_physics.turbine.protection_system.trip_active") \
  I(trip_latched_mask,    "=H.turbine_trip_mask(root)")

/* ---- turbine, per-stage and metal-temperature arrays: visited one stage at a time, so the kernel
 * streams them straight from / to their SoA columns instead of holding them in registers */
#define NPB_TSTG_NOUT 0
#define NPB_TSTG_FIELDS(F, A, I) \
  A(stage_efficiency_degradation, 14, "=list(root.secondary_physics.turbine.stage_system.stages.values())[{k}].efficiency_degradation") \
  A(stage_deposit_thickness, 14,      "=list(root.secondary_physics.turbine.stage_system.stages.values())[{k}].deposit_thickness") \
  A(stage_blade_wear_factor, 14,      "=list(root.secondary_physics.turbine.stage_system.stages.values())[{k}].blade_wear_factor") \
  A(rotor_temperatures, 8,  "secondary_physics.turbine.thermal_tracker.rotor_temperatures[{k}]") \
  A(casing_temperatures, 6, "secondary_physics.turbine.thermal_tracker.casing_temperatures[{k}]") \
  A(blade_temperatures, 14, "secondary_physics.turbine.thermal_tracker.blade_temperatures[{k}]")

/* ---- WaterChemistry instances (x2): [0] the secondary-level one shared with the feedwater system
 * (secondary/__init__.py:316-321), [1] the condenser-owned one (condenser/physics.py:528-532).
 * The third instance (SG-system-owned) is never updated and lives in npb_params.h.
 * reference: water_chemistry.py:222-275; composite indices are recomputed inside every update. */
#define NPB_CHEM_NOUT 4   /* the last 4 fp64 members are outputs of the step (see "output members" above) */
#define NPB_CHEM_FIELDS(F, A, I) \
  F(ph,                     "=(root.secondary_physics.water_chemistry, root.secondary_physics.condenser.water_chemistry)[{i}].ph") \
  F(hardness,               "=(root.secondary_physics.water_chemistry, root.secondary_physics.condenser.water_chemistry)[{i}].hardness") \
  F(total_dissolved_solids, "=(root.secondary_physics.water_chemistry, root.secondary_physics.condenser.water_chemistry)[{i}].total_dissolved_solids") \
  F(chloride,               "=(root.secondary_physics.water_chemistry, root.secondary_physics.condenser.water_chemistry)[{i}].chloride") \
  F(chlorine_residual,      "=(root.secondary_physics.water_chemistry, root.secondary_physics.condenser.water_chemistry)[{i}].chlorine_residual") \
  F(antiscalant_concentration, "=(root.secondary_physics.water_chemistry, root.secondary_physics.condenser.water_chemistry)[{i}].antiscalant_concentration") \
  F(corrosion_inhibitor_level, "=(root.secondary_physics.water_chemistry, root.secondary_physics.condenser.water_chemistry)[{i}].corrosion_inhibitor_level") \
  F(dissolved_oxygen,       "=(root.secondary_physics.water_chemistry, root.secondary_physics.condenser.water_chemistry)[{i}].dissolved_oxygen") \
  F(treatment_efficiency,   "=(root.secondary_physics.water_chemistry, root.secondary_physics.condenser.water_chemistry)[{i}].treatment_efficiency") \
  F(water_aggressiveness,   "=(root.secondary_physics.water_chemistry, root.secondary_physics.condenser.water_chemistry)[{i}].water_aggressiveness") \
  F(scaling_tendency,       "=(root.secondary_physics.water_chemistry, root.secondary_physics.condenser.water_chemistry)[{i}].scaling_tendency")

/* ---- pH controller of the secondary system + the effects it leaves pending for the next
 * WaterChemistry update.  reference: ph_control_system.py:131-190 (PHControllerState),
 * water_chemistry.py:659-681 (_pending_chemistry_effects).  The controller's sensor noise and random
 * equipment failures use numpy's unseeded GLOBAL RNG (ph_control_system.py:278,288,409-420) and are
 * therefore not reproducible in the reference itself; this model is the deterministic limit
 * (noise 0, no random failures), which is also how the golden vectors were taken. */
#define NPB_PH_NOUT 1   /* the last 1 fp64 members are outputs of the step (see "output members" above) */
#define NPB_PH_FIELDS(F, A, I) \
  F(measured_ph,            "secondary_physics.ph_control_system.controller.state.measured_ph") \
  F(integral_sum,           "secondary_physics.ph_control_system.controller.state.integral_sum") \
  F(previous_error,         "secondary_physics.ph_control_system.controller.state.previous_error") \
  F(ammonia_tank_level,     "secondary_physics.ph_control_system.controller.state.ammonia_tank_level") \
  F(morpholine_tank_level,  "secondary_physics.ph_control_system.controller.state.morpholine_tank_level") \
  F(pending_ammonia_dose,   "=root.secondary_physics.water_chemistry._pending_chemistry_effects['ph_control']['ammonia_dose_rate']") \
  F(pending_morpholine_dose,"=root.secondary_physics.water_chemistry._pending_chemistry_effects['ph_control']['morpholine_dose_rate']") \
  F(controller_output,      "secondary_physics.ph_control_system.controller.state.controller_output") \
  I(controller_enabled,     "secondary_physics.ph_control_system.controller.state.controller_enabled") \
  I(ammonia_supply_available, "secondary_physics.ph_control_system.controller.state.ammonia_supply_available") \
  I(morpholine_supply_available, "secondary_physics.ph_control_system.controller.state.morpholine_supply_available") \
  I(has_pending_effects,    "=float('ph_control' in getattr(root.secondary_physics.water_chemistry, '_pending_chemistry_effects', {}))")

/* ---- condenser: tube degradation, 3-species fouling, vacuum system with 2 steam-jet ejectors
 * reference: condenser/physics.py:55-71,151-165,540-559, vacuum_system.py:40-52,270-300,
 *            vacuum_pump.py:52-92 */
#define NPB_COND_NOUT 4   /* the last 4 fp64 members are outputs of the step (see "output members" above) */
#define NPB_COND_FIELDS(F, A, I) \
  F(cooling_water_outlet_temp,  "secondary_physics.condenser.cooling_water_outlet_temp") \
  F(active_tube_count,          "secondary_physics.condenser.tube_degradation.active_tube_count") \
  F(plugged_tube_count,         "secondary_physics.condenser.tube_degradation.plugged_tube_count") \
  F(average_wall_thickness,     "secondary_physics.condenser.tube_degradation.average_wall_thickness") \
  F(vibration_damage,           "secondary_physics.condenser.tube_degradation.vibration_damage_accumulation") \
  F(corrosion_damage,           "secondary_physics.condenser.tube_degradation.corrosion_damage_accumulation") \
  F(biofouling_thickness,       "secondary_physics.condenser.fouling_model.biofouling_thickness") \
  F(scale_thickness,            "secondary_physics.condenser.fouling_model.scale_thickness") \
  F(corrosion_product_thickness,"secondary_physics.condenser.fouling_model.corrosion_product_thickness") \
  F(fouling_distribution_factor,"secondary_physics.condenser.fouling_model.fouling_distribution_factor") \
  F(time_since_cleaning,        "secondary_physics.condenser.fouling_model.time_since_cleaning") \
  F(condenser_pressure,         "secondary_physics.condenser.vacuum_system.condenser_pressure") \
  F(current_air_leakage,        "secondary_physics.condenser.vacuum_system.current_air_leakage") \
  F(air_mass_in_condenser,      "secondary_physics.condenser.vacuum_system.air_mass_in_condenser") \
  F(rotation_timer,             "secondary_physics.condenser.vacuum_system.control_logic.rotation_timer") \
  A(ej_nozzle_fouling, 2,       "=list(root.secondary_physics.condenser.vacuum_system.ejectors.values())[{k}].nozzle_fouling_factor") \
  A(ej_diffuser_fouling, 2,     "=list(root.secondary_physics.condenser.vacuum_system.ejectors.values())[{k}].diffuser_fouling_factor") \
  A(ej_nozzle_erosion, 2,       "=list(root.secondary_physics.condenser.vacuum_system.ejectors.values())[{k}].nozzle_erosion_factor") \
  F(heat_rejection_rate,        "secondary_physics.condenser.heat_rejection_rate") \
  F(total_fouling_resistance,   "secondary_physics.condenser.fouling_model.total_fouling_resistance") \
  F(air_partial_pressure,       "secondary_physics.condenser.vacuum_system.air_partial_pressure") \
  F(vacuum_system_efficiency,   "secondary_physics.condenser.vacuum_system.system_efficiency") \
  I(ej_operating_mask,          "=sum(int(e.is_operating) << k for k, e in enumerate(root.secondary_physics.condenser.vacuum_system.ejectors.values()))") \
  I(lead_ejector,               "=(-1 if root.secondary_physics.condenser.vacuum_system.control_logic.lead_ejector_id is None else int(root.secondary_physics.condenser.vacuum_system.control_logic.lead_ejector_id[-1]) - 1)") \
  I(lag_ejector,                "=(-1 if root.secondary_physics.condenser.vacuum_system.control_logic.lag_ejector_id is None else int(root.secondary_physics.condenser.vacuum_system.control_logic.lag_ejector_id[-1]) - 1)")

/* ---- secondary-system level carried scalars and the outputs get_observation() reads
 * reference: systems/secondary/__init__.py:300-310,385-398,447-453,921-927 */
#define NPB_SEC_NOUT 9   /* the last 9 fp64 members are outputs of the step (see "output members" above) */
#define NPB_SEC_FIELDS(F, A, I) \
  F(previous_feedwater_temp,  "secondary_physics._previous_feedwater_temp") \
  F(cooling_water_temperature,"secondary_physics.cooling_water_temperature") \
  F(operating_hours,          "secondary_physics.operating_hours") \
  A(prev_sg_levels, 3,        "secondary_physics._previous_sg_conditions['levels'][{k}]") \
  A(prev_sg_steam_flows, 3,   "secondary_physics._previous_sg_conditions['steam_flows'][{k}]") \
  A(prev_sg_qualities, 3,     "secondary_physics._previous_sg_conditions['steam_qualities'][{k}]") \
  F(electrical_power_output,  "secondary_physics.electrical_power_output") \
  F(thermal_efficiency,       "secondary_physics.thermal_efficiency") \
  F(total_steam_flow,         "secondary_physics.total_steam_flow") \
  F(total_heat_transfer,      "secondary_physics.total_heat_transfer") \
  F(total_feedwater_flow,     "secondary_physics.total_feedwater_flow") \
  F(load_demand,              "secondary_physics.load_demand") \
  F(sg_avg_pressure,          "secondary_physics.steam_generator_system.average_steam_pressure") \
  F(sg_avg_temperature,       "secondary_physics.steam_generator_system.average_steam_temperature") \
  F(sg_avg_quality,           "secondary_physics.steam_generator_system.average_steam_quality") \
  I(has_previous_sg_conditions, "=float(hasattr(root.secondary_physics, '_previous_sg_conditions'))") \
  I(sg_system_availability,   "secondary_physics.steam_generator_system.system_availability")

/* ---- automatic maintenance of the feedwater pumps: what a run observes of the reference's control
 * plane -- StateManager threshold scan with per-threshold cooldowns (simulator/state/state_manager.py:1267-1369), the
 * orchestrator's choice of one action per pump and step (systems/maintenance/maintenance_orchestrator.py:81-524),
 * AutoMaintenanceSystem's work-order creation with its duplicate rules (systems/maintenance/auto_maintenance.py:331-456)
 * and the one-execution-per-check queue (:200-236, :468-580) -- restated in nuclear_sim_amd/csrc/npd_maintenance.h.
 * Carried only when params.maint_enabled; the step kernel never touches these columns.
 * mpump (one per pump): array index = catalogued parameter / action of include/npb_maint.h.
 *   last_violation_time[p]  StateManager.threshold_last_violation_times[pump][parameter]; -1 = never
 *   wo_order[a]             0 = the pump has no open work order for action a, n = its open order was the n-th created
 *                           (WorkOrderManager.work_orders is insertion ordered and executed in that order)
 *   wo_planned_start[a]     planned_start_date of that order (minutes)
 *   last_trigger_time[a]    AutoMaintenanceSystem.recent_work_order_triggers[pump:action]; -1 = never
 *   wo_bearing              which bearing the open bearing_replacement order names (work_order.metadata), NPB_BEARING_*
 * "=H.*" paths are evaluated by oracle/ref_harness/leaves.py helpers (dict lookups with defaults). */
#define NPB_MPUMP_NOUT 0
#define NPB_MPUMP_FIELDS(F, A, I) \
  A(last_violation_time, 16,  "=H.last_violation(root, {i}, {k})") \
  A(wo_order, 18,             "=H.open_wo(root, {i}, {k}, 'order')") \
  A(wo_planned_start, 18,     "=H.open_wo(root, {i}, {k}, 'planned_start_date')") \
  A(last_trigger_time, 18,    "=H.last_trigger(root, {i}, {k})") \
  F(wo_bearing,               "=H.open_wo_bearing(root, {i})")

/* maint (one per plant): check cadence and counters; executed[a] = work orders of action a carried out so far, on any
 * pump (the reference keeps them as COMPLETED work orders) */
#define NPB_MAINT_NOUT 0
#define NPB_MAINT_FIELDS(F, A, I) \
  F(last_check_time,          "maintenance_system.last_check_time") \
  A(executed, 18,             "=H.executed(root, {k})") \
  I(work_orders_created,      "maintenance_system.work_orders_created") \
  I(maintenance_actions_performed, "maintenance_system.maintenance_actions_performed")

/* section list: S(member, TYPE, struct_type, count) */
#define NPB_SECTIONS(S) \
  S(prim, PRIM, npb_prim_t, 1) \
  S(sg,   SG,   npb_sg_t,   NPB_NUM_SG) \
  S(pump, PUMP, npb_pump_t, NPB_NUM_PUMPS) \
  S(fw,   FW,   npb_fw_t,   1) \
  S(turb, TURB, npb_turb_t, 1) \
  S(tstg, TSTG, npb_tstg_t, 1) \
  S(chem, CHEM, npb_chem_t, 2) \
  S(ph,   PH,   npb_ph_t,   1) \
  S(cond, COND, npb_cond_t, 1) \
  S(sec,  SEC,  npb_sec_t,  1) \
  S(mpump, MPUMP, npb_mpump_t, NPB_NUM_PUMPS) \
  S(maint, MAINT, npb_maint_t, 1)

/* ------------------------------------------------------------------ structs */
#define NPB__F(name, path)        double name;
#define NPB__A(name, count, path) double name[count];
#define NPB__I(name, path)        int32_t name;
#define NPB__NOF(name, path)
#define NPB__NOA(name, count, path)
#define NPB__NOI(name, path)
#define NPB__CNTF(name, path)        + 1
#define NPB__CNTA(name, count, path) + (count)
#define NPB__CNTI(name, path)        + 1

/* per section: struct npb_<sec>_t (all fp64 members first, then all int32),
 * NPB_<SEC>_NF64 / NPB_<SEC>_NI32 = slots of one instance */
#define NPB__DEFINE(member, T, stype, count) \
  typedef struct stype { \
    NPB_##T##_FIELDS(NPB__F, NPB__A, NPB__NOI) \
    NPB_##T##_FIELDS(NPB__NOF, NPB__NOA, NPB__I) \
  } stype; \
  enum { NPB_##T##_NF64 = (0 NPB_##T##_FIELDS(NPB__CNTF, NPB__CNTA, NPB__NOI)), \
         NPB_##T##_NI32 = (0 NPB_##T##_FIELDS(NPB__NOF, NPB__NOA, NPB__CNTI)), \
         NPB_##T##_NCARRY = NPB_##T##_NF64 - NPB_##T##_NOUT,    /* fp64 members that are carried state */ \
         NPB_##T##_NNARROW = NPB_##T##_NOUT + NPB_##T##_NI32,   /* 4-byte members: outputs (as float), then int32 */ \
         NPB_##T##_NCOL64 = NPB_##T##_NCARRY + (NPB_##T##_NNARROW + 1) / 2, /* 8-byte columns of one instance */ \
         NPB_##T##_NCOL32 = NPB_##T##_NCARRY + NPB_##T##_NNARROW,        /* 4-byte columns (fp32 storage) */ \
         NPB_##T##_COUNT = (count) };
NPB_SECTIONS(NPB__DEFINE)

/* base slot of each section in the global fp64 / int32 column tables
 * (section-major, then instance, then member order) */
enum {
#define NPB__BASEF(member, T, stype, count) \
  NPB_##T##_F64_BASE, NPB_##T##_F64_LAST_ = NPB_##T##_F64_BASE + (count) * NPB_##T##_NF64 - 1,
  NPB_SECTIONS(NPB__BASEF)
  NPB_TOTAL_F64
};
enum {
#define NPB__BASEI(member, T, stype, count) \
  NPB_##T##_I32_BASE, NPB_##T##_I32_LAST_ = NPB_##T##_I32_BASE + (count) * NPB_##T##_NI32 - 1,
  NPB_SECTIONS(NPB__BASEI)
  NPB_TOTAL_I32
};

/* arena columns: section-major, then instance; inside an instance the carried fp64 members in order, then the
 * narrow members (outputs as float, then int32) two per 8-byte column -- or one per 4-byte column under fp32
 * storage, where every column is 4 bytes wide */
enum {
#define NPB__BASEC64(member, T, stype, count) \
  NPB_##T##_COL64_BASE, NPB_##T##_COL64_LAST_ = NPB_##T##_COL64_BASE + (count) * NPB_##T##_NCOL64 - 1,
  NPB_SECTIONS(NPB__BASEC64)
  NPB_TOTAL_COL64
};
enum {
#define NPB__BASEC32(member, T, stype, count) \
  NPB_##T##_COL32_BASE, NPB_##T##_COL32_LAST_ = NPB_##T##_COL32_BASE + (count) * NPB_##T##_NCOL32 - 1,
  NPB_SECTIONS(NPB__BASEC32)
  NPB_TOTAL_COL32
};

/* slot of a member inside its section instance (fp64 members are laid out first) */
#define NPB_F64_SLOT(stype, member) ((int)(offsetof(stype, member) / sizeof(double)))
#define NPB_I32_SLOT(stype, T, member) \
  ((int)((offsetof(stype, member) - (size_t)NPB_##T##_NF64 * sizeof(double)) / sizeof(int32_t)))

#endif /* NPB_FIELDS_H */
