/*
 * npb.h -- C ABI of the MI355X batched plant stepper (libnpb.so).
 *
 * This is the drop-in boundary for the per-timestep physics path of NuclearnAI/nuclear-sim:
 * the reference has no FFI (it is pure Python), so each entry point names the Python interface
 * it stands in for.  A maintainer of the reference binds these with ctypes;
 * this repo's own host mirror is nuclear_sim_amd/env.py.
 *
 * Conventions: plain pointers and sizes only; every function returns 0 on success or a negative
 * NPB_E* code (text via npb_last_error); the caller owns every I/O buffer and passes raw DEVICE
 * pointers (e.g. torch.Tensor.data_ptr()); the library owns only the struct-of-arrays state arena
 * inside the handle; npb_step() allocates nothing and only enqueues work on `stream`
 * (a hipStream_t, NULL = default stream).  One handle per stream; distinct handles are independent.
 */
#ifndef NPB_H
#define NPB_H

#include <stddef.h>
#include <stdint.h>
#include "npb_fields.h"
#include "npb_params.h"
#include "npb_maint.h"

#ifdef __cplusplus
extern "C" {
#endif

#define NPB_VERSION 142 /* 0.1.4.2: NPB_DIAG_DIM 170 (state-log rows of round 4), npb_state_arena_layout, NPB_EINVAL for NPB_HEAT_EXTERNAL without its input column; 0.1.4.1: npb_state_arena_segment (segmented arenas), step-kernel variant 5, NPB_DIAG_DIM 136; 0.1.4: npb_debug_last_step_kernel, npb_info_dim / npb_obs_dim / npb_diag_dim, maintenance catalogs by index; 0.1.3.1: params.kinetics_rk4_substeps; 0.1.3: NPB_MODE_PRIMARY, reactivity components behind the info block (params.info_reactivity_components); 0.1.2: npb_reset_reference, maintenance table (npb_maint.h, mpump.* columns); 0.1.1: one arena of equally wide columns, npb_locate_field, npb_gather_fields, npb_create_storage */
#ifndef NPB_API
#define NPB_API __attribute__((visibility("default")))
#endif

enum { NPB_OK = 0, NPB_EINVAL = -1, NPB_EHIP = -2, NPB_ENOMEM = -3 };
enum { NPB_KIND_F64 = 0, NPB_KIND_I32 = 1 };
enum { NPB_OBS_DIM = 22, NPB_INFO_DIM = 17 };

/* info columns written by npb_step (the scalar keys of step()'s info dict, sim.py:199-250) */
enum {
  NPB_INFO_THERMAL_POWER = 0, NPB_INFO_REACTIVITY_PCM, NPB_INFO_ELECTRICAL_POWER, NPB_INFO_THERMAL_EFFICIENCY,
  NPB_INFO_STEAM_FLOW, NPB_INFO_STEAM_PRESSURE, NPB_INFO_CONDENSER_PRESSURE, NPB_INFO_CONDENSER_HEAT_REJECTION,
  NPB_INFO_TIME, NPB_INFO_FEEDWATER_FLOW,
  /* fp64 inputs of the reference's heat-flow bookkeeping and of the derived keys of its secondary result dict
   * (secondary/__init__.py:679-744, 922-1010; nuclear_sim_amd/env.py secondary_result): total steam-generator heat transfer
   * [W], turbine gross electrical power [MW], feedwater pump power [MW], primary thermal power over the three loops [MW] */
  NPB_INFO_SG_HEAT_TRANSFER, NPB_INFO_TURBINE_POWER, NPB_INFO_FEEDWATER_POWER, NPB_INFO_PRIMARY_THERMAL_POWER,
  /* the three keys of that dict that are left over from inside the turbine step (secondary/__init__.py:955-958): the stage
   * system's cycle efficiency (h_in - h_out) / h_in (stage_system.py:983-993) and the summed outputs of the HP-1..8 and
   * LP-1..6 stages [MW] (enhanced_physics.py:879-880); 0 where the turbine is not stepped (NPB_MODE_PRIMARY_SG) */
  NPB_INFO_TURBINE_EFFICIENCY, NPB_INFO_TURBINE_HP_POWER, NPB_INFO_TURBINE_LP_POWER
};
/* Step-internal diagnostics (optional: npb_set_diagnostics): what the reference's state log holds per turbine stage from inside
 * the expansion (TurbineStage.get_state_dict, stage_system.py:379-393) and nothing later in the step can recover -- fourteen
 * values each, HP-1..8 then LP-1..6, column (NPB_DIAG_* + stage) of a [NPB_DIAG_DIM][pitch] fp64 buffer -- and per steam generator. */
enum {
  NPB_DIAG_STAGE_INLET_PRESSURE = 0, NPB_DIAG_STAGE_INLET_TEMPERATURE = 14, NPB_DIAG_STAGE_OUTLET_PRESSURE = 28,
  NPB_DIAG_STAGE_OUTLET_TEMPERATURE = 42, NPB_DIAG_STAGE_POWER_OUTPUT = 56, NPB_DIAG_STAGE_LOADING_FACTOR = 70,
  /* per steam generator (SteamGenerator.get_state_dict, steam_generator.py:943-985), three values each, SG-0..2 */
  NPB_DIAG_SG_PRIMARY_INLET_TEMP = 84, NPB_DIAG_SG_PRIMARY_OUTLET_TEMP = 87, NPB_DIAG_SG_OVERALL_HTC = 90,
  NPB_DIAG_SG_FEEDWATER_FLOW_RATE = 93,
  /* per feedwater pump, FWP-1..4: the lubrication system's health factor as the step's wear update leaves it
   * (lubrication_base.py:398-399: mean component performance x lubrication effectiveness -- a maintenance action carried out
   * later in the same step does not refresh it), and the two maintenance flags of the pump's state dict
   * (pump_lubrication.py:1636-1641: an action / an oil top-off was carried out on this pump in this step) */
  NPB_DIAG_PUMP_HEALTH_FACTOR = 96, NPB_DIAG_PUMP_MAINTENANCE_OCCURRED = 100, NPB_DIAG_PUMP_OIL_TOP_OFF_OCCURRED = 104,
  /* the steam-generator conditions the feedwater system was given this step -- the copies of the step before, the hard-coded
   * ones at the first step (secondary/__init__.py:447-453) -- as its state dict averages them (feedwater/physics.py:1140-1145) */
  NPB_DIAG_FW_AVG_SG_LEVEL = 108, NPB_DIAG_FW_AVG_SG_PRESSURE = 109, NPB_DIAG_FW_TOTAL_STEAM_FLOW = 110, NPB_DIAG_FW_AVG_STEAM_QUALITY = 111,
  /* the rotor model's torque balance of this step (rotor_dynamics.py:855-911): bearing friction torque [N m] from the bearing loads
   * the step began with, net torque, acceleration [RPM/s] */
  NPB_DIAG_ROTOR_FRICTION_TORQUE = 112, NPB_DIAG_ROTOR_NET_TORQUE = 113, NPB_DIAG_ROTOR_ACCELERATION = 114,
  /* the condenser's step (condenser/physics.py:564-728, 73-145; vacuum_pump.py:96-190): overall heat-transfer coefficient, tube
   * leak rate, the first ejector's motive steam flow and steam consumption rate, the vacuum system's total motive steam */
  NPB_DIAG_COND_OVERALL_HTC = 115, NPB_DIAG_COND_TUBE_LEAK_RATE = 116, NPB_DIAG_COND_SJE1_STEAM_FLOW = 117,
  NPB_DIAG_COND_SJE1_STEAM_CONSUMPTION = 118, NPB_DIAG_COND_VACUUM_STEAM_CONSUMPTION = 119,
  /* per steam generator: the tube-scale formation rate of the step [mm / year] (tube_interior_fouling.py:117-188) */
  NPB_DIAG_SG_SCALE_FORMATION_RATE = 120,
  /* the feedwater system's performance factor (feedwater/physics.py:800-805: mean flow x efficiency factor of the running pumps x
   * water-quality factor, from the shared chemistry between its two updates of the step, x the diagnostics' health score) */
  NPB_DIAG_FW_PERFORMANCE_FACTOR = 123,
  /* accumulators of the rotor model that no physics reads, summed in the caller's buffer from the step the diagnostics were
   * switched on (zero the buffer at construction and they are the reference's): the four bearings' clearance increase [mm]
   * (rotor_dynamics.py:298-300) and the overspeed event count (:900-902) */
  NPB_DIAG_ROTOR_CLEARANCE_INCREASE = 124, NPB_DIAG_ROTOR_OVERSPEED_EVENTS = 128,
  /* the oil temperature the lubrication system hands each turbine bearing, TB-001..004 (turbine_bearing_lubrication.py:967-1072) */
  NPB_DIAG_BEARING_OIL_TEMP = 129,
  /* the stage system's efficiency factor -- a product carried from step to step in the caller's buffer like the accumulators above
   * (a row of zeros reads as 1.0) -- and the turbine's performance factor built on it (stage_system.py:981, enhanced_physics.py:823-828) */
  NPB_DIAG_STAGE_SYSTEM_EFFICIENCY = 133, NPB_DIAG_TURBINE_PERFORMANCE_FACTOR = 134,
  /* number of alarms the feedwater protection system holds after the step (protection_system.py:399-445) */
  NPB_DIAG_FW_ACTIVE_ALARMS = 135,
  /* round 4: what the state log's remaining columns need from inside the step.
   * per feedwater pump, FWP-1..4: the maintenance action carried out on the pump in this step as catalog index + 1 (0 = none;
   * include/npb_maint.h), from which the thirteen <action>_occurred flags of the pump's state dict follow
   * (pump_lubrication.py:642-643, 1636-1641); written by the maintenance rule like NPB_DIAG_PUMP_MAINTENANCE_OCCURRED */
  NPB_DIAG_PUMP_MAINTENANCE_ACTION = 136,
  /* the feedwater protection system's bookkeeping (protection_system.py:447-476, 680-716): trips standing after the step;
   * carried in the caller's buffer from the step diagnostics were switched on: steps on which a trip came up
   * (valid_trip_count), emergency feedwater / steam dump set by such a step's trips and never cleared by the step */
  NPB_DIAG_FW_ACTIVE_TRIPS = 140, NPB_DIAG_FW_VALID_TRIP_COUNT = 141, NPB_DIAG_FW_EMERGENCY_FEEDWATER = 142, NPB_DIAG_FW_STEAM_DUMP = 143,
  /* per turbine stage: the extraction flow the stage took [kg/s] (stage_system.py:186-203) */
  NPB_DIAG_STAGE_EXTRACTION_FLOW = 144,
  /* per steam-jet ejector, SJE-001..002 (vacuum_pump.py:470-536): air-removal capacity, motive steam flow, steam consumption
   * rate of the step; and carried in the caller's buffer: the compression ratio (kept from the ejector's last operating step;
   * the row starts at 1.0) and the hours it has operated; then the vacuum system's total air removal (vacuum_system.py:499) */
  NPB_DIAG_COND_SJE_CAPACITY = 158, NPB_DIAG_COND_SJE_STEAM_FLOW = 160, NPB_DIAG_COND_SJE_STEAM_CONSUMPTION = 162,
  NPB_DIAG_COND_SJE_COMPRESSION_RATIO = 164, NPB_DIAG_COND_SJE_OPERATING_HOURS = 166, NPB_DIAG_COND_AIR_REMOVAL = 168,
  /* the stage system's total power [MW] as its state dict holds it (stage_system.py:976): stability-adjusted, before the turbine's
   * protection and availability factors */
  NPB_DIAG_STAGE_SYSTEM_TOTAL_POWER = 169,
  NPB_DIAG_DIM = 170
};
/* info["reactivity_components"] (sim.py:205; reactivity_model.py:77-125, pcm, the dict's insertion order).  Only the
 * reactor heat source has them, and only a caller that sets params.info_reactivity_components gets them: the info
 * buffer handed to npb_step must then hold a second block behind the first, [n_plants][NPB_INFO_DIM] followed by
 * [n_plants][NPB_INFO_NRHO]. */
enum {
  NPB_RHO_CONTROL_RODS = 0, NPB_RHO_BORON, NPB_RHO_DOPPLER, NPB_RHO_MODERATOR_TEMP, NPB_RHO_MODERATOR_VOID, NPB_RHO_PRESSURE,
  NPB_RHO_XENON, NPB_RHO_SAMARIUM, NPB_RHO_FUEL_DEPLETION, NPB_RHO_BURNABLE_POISONS, NPB_INFO_NRHO
};
/* trip_flags bits */
enum {
  NPB_TRIP_SCRAM = 1u << 0,        /* ReactorState.scram_status latched (scram_logic.py:57) */
  NPB_TRIP_SCRAM_FIRED = 1u << 1,  /* scram fired on this step == step()['done'] (sim.py:256) */
  NPB_TRIP_NAN_RESET = 1u << 2,    /* NaN reset taken (thermal_hydraulics.py:247-270) */
  NPB_TRIP_TURBINE = 1u << 3,      /* TurbineProtectionSystem.trip_active */
  NPB_TRIP_FW_SYSTEM = 1u << 4,    /* FeedwaterProtectionSystem.system_trip_active */
  NPB_TRIP_FW_PUMP0 = 1u << 8      /* bits 8..11: feedwater pump i trip_active */
};

typedef struct NpbHandle NpbHandle;

NPB_API int npb_version(void);
/* schema sizes (must equal NPB_TOTAL_F64 / NPB_TOTAL_I32 the caller was compiled against) */
NPB_API int npb_num_f64(void);
NPB_API int npb_num_i32(void);
/* widths of the row-major output blocks of npb_step the library was built with (NPB_OBS_DIM / NPB_INFO_DIM / NPB_INFO_NRHO /
 * NPB_DIAG_DIM).  A binding that allocates those buffers checks them at load: a caller sized for fewer info columns than the
 * library writes would be overrun (the round-2 host crash). */
NPB_API int npb_obs_dim(void);
NPB_API int npb_info_dim(void);
NPB_API int npb_info_nrho(void);
NPB_API int npb_diag_dim(void);
/* the catalogs of include/npb_maint.h by index (NULL past the end): threshold parameter names as the state log spells them,
 * MaintenanceActionType values */
NPB_API int npb_maint_num_params(void);
NPB_API int npb_maint_num_actions(void);
NPB_API const char *npb_maint_param_name(int k);
NPB_API const char *npb_maint_action_name(int a);
NPB_API int npb_maint_action_has_handler(int a);
/* arena bytes per plant (fp64 storage): 8 * (carried fp64 members + ceil(narrow members / 2) per section instance) */
NPB_API size_t npb_state_bytes(void);
/* algorithmic HBM bytes of one plant-step: carried fp64 members read and written (16 B), int32 members read and
 * written (8 B), output members written as float (4 B) -- all but the maint.* section, which only the
 * maintenance kernel touches -- + per-step inputs (action 4 + magnitude/setpoint/noise/cooling 4*8) + outputs
 * (obs 22*8 + reward 8 + done 1 + trip_flags 4 + info 17*8) */
NPB_API size_t npb_step_bytes_per_plant(void);
/* the same for one handle: with fp32 storage every carried real moves 4 bytes instead of 8, and under
 * ConstantHeatSource the 12 point-kinetics columns of the primary section are not touched at all */
NPB_API size_t npb_handle_step_bytes_per_plant(const NpbHandle *h);

NPB_API void npb_default_params(npb_params_t *p);

/* NuclearPlantSimulator.__init__ (sim.py:30-87) for n_plants plants on HIP device `device`:
 * allocates the SoA arena and fills it with the construction-time state.  One handle carries at most
 * 4 GiB of fp64 state (about one million plants: the step kernel's column offsets are 32-bit);
 * larger batches use several handles (they are independent).  When the bytes a step touches are about the size of the
 * 256 MB Infinity Cache (200-340 MB: 65 536 fp64 plants), where the arena lands in physical memory decides between 0.097
 * and 0.110 ms per step; creation then times the step kernel on up to four candidate arenas and keeps the fastest
 * (~15 ms; the arena ends in the construction-time state all the same; environment NPB_PLACEMENT_PROBE=0 turns it off). */
NPB_API int npb_create(const npb_params_t *params, int n_plants, int device, NpbHandle **out);
/* The same with the element type of the real-valued state columns chosen (BASELINE config 5, "fp32-mixed"):
 * NPB_STORAGE_F32 keeps the carried state as float in HBM and in the LDS staging -- half the traffic of
 * the HBM-bound step kernel -- while every expression is still evaluated in fp64 exactly as in the fp64
 * build; values are rounded to float once per step, when they are stored.  Flags, counters and enums are
 * int32 columns either way and stay bit-exact as long as no rounded value sits within 1e-7 relative of a
 * trip threshold.  Inputs, observations, rewards and info stay fp64; npb_get_field / npb_set_field convert.
 * The reference has no counterpart (it is fp64 throughout); parity for this mode is 1e-4 relative on
 * observations (tests/test_gpu_parity.py). */
enum { NPB_STORAGE_F64 = 0, NPB_STORAGE_F32 = 1 };
NPB_API int npb_create_storage(const npb_params_t *params, int n_plants, int device, int storage, NpbHandle **out);
NPB_API int npb_storage(const NpbHandle *h);
NPB_API int npb_destroy(NpbHandle *h);
NPB_API const char *npb_last_error(const NpbHandle *h); /* h may be NULL: last create error */
NPB_API int npb_num_plants(const NpbHandle *h);
NPB_API int npb_set_params(NpbHandle *h, const npb_params_t *params);
/* thresholds of the automatic maintenance of the feedwater pumps (include/npb_maint.h): what the reference reads from
 * the maintenance_system section of its configuration into StateManager.maintenance_thresholds.  A new handle carries
 * npb_maint_table_default() (the data-gen action-test configuration), whose oil_level row follows the two parameters of ABI
 * version 1 (params.maint_oil_level_threshold / _cooldown_hours); a table set here is taken exactly as given. */
NPB_API int npb_set_maintenance_table(NpbHandle *h, const npb_maint_table_t *table);
NPB_API void npb_default_maintenance_table(npb_maint_table_t *table);
/* AutoMaintenanceSystem.maintenance_actions_performed of every plant as a column the caller owns (device, int32[n_plants]; NULL =
 * none): with params.maint_enabled every npb_step leaves it current -- filled whole after any call that may have changed state,
 * then kept by the maintenance rule kernel for the plants whose count it moves -- so a loop that wants the event counts after
 * each step (the data-gen runner does, maintenance_scenario_runner.py:392-411) needs no npb_get_field launch per step. */
NPB_API int npb_set_maintenance_count_buffer(NpbHandle *h, int32_t *counts);

/* re-initialise plants to the construction-time state; mask (device, uint8[n], NULL = all) selects plants.
 * Stands in for constructing a fresh simulator (the data-gen runner's episode start,
 * maintenance_scenario_runner.py:210-244). */
NPB_API int npb_reset(NpbHandle *h, const uint8_t *mask, void *stream);
/* NuclearPlantSimulator.reset(start_at_steady_state)  sim.py:546-581 with the reference's own semantics: each
 * subsystem's reset() puts part of its state back to literals, re-applies initial conditions for another part and
 * leaves the rest (lubrication systems, metal temperatures, pH controller, tube scale, the previous step's SG
 * conditions) as the history left it; start_at_steady_state != 0 additionally runs initialize_to_steady_state
 * (secondary/__init__.py:1074-1357: one steam-generator update as a side effect, every pump force-set).
 * mask as for npb_reset.  Follow with npb_observe for reset()'s return value. */
NPB_API int npb_reset_reference(NpbHandle *h, const uint8_t *mask, int start_at_steady_state, void *stream);

/* state columns: sim.state.<attr> / component attribute access.  `slot` is the global slot of
 * npb_fields.h; buf holds n_plants elements (double or int32_t) on the device or the host. */
NPB_API int npb_get_field(NpbHandle *h, int kind, int slot, void *buf, int buf_is_device, void *stream);
NPB_API int npb_set_field(NpbHandle *h, int kind, int slot, const void *buf, int buf_is_device, void *stream);
/* Many fields in one launch, every value widened to double: out (device) = [n_fields][n_plants].  This is the sampling
 * step of a columnar state log -- the batched counterpart of StateManager.collect_states (state_manager.py:152-213),
 * which walks every provider's get_state_dict() and appends one pandas row per step.  kinds / slots are host arrays;
 * the request is remembered, so repeating it costs one kernel launch. */
NPB_API int npb_gather_fields(NpbHandle *h, int n_fields, const int *kinds, const int *slots, double *out, void *stream);
/* raw arena (checkpointing, external kernels): one allocation of equally wide columns, column-major with `pitch`
 * plants per column; the members of the schema are mapped onto columns as include/npb_fields.h describes.  A handle of
 * more than 45 056 plants keeps its arena in SEGMENTS (npb_state_arena_segment(h) plants each -- 16 384 --, 0 = not
 * segmented): the allocation is then consecutive [columns][pitch] blocks, pitch = the segment size, block s holding
 * plants s * pitch .. (s + 1) * pitch - 1 -- plant p's element of column c is at (p / pitch) * pitch * columns +
 * c * pitch + p % pitch.  (Measured: the step of 65 536 plants is 5 % faster on such an arena than on one block, that
 * of 262 144 plants 10 %.) */
NPB_API int npb_state_arena(NpbHandle *h, void **arena, size_t *pitch, int *storage);
NPB_API size_t npb_state_arena_segment(const NpbHandle *h);
/* The same with the whole layout in one answer: *segment = plants per segment (0 = one block), *columns = arena columns of the
 * handle's storage type.  npb_state_arena itself REFUSES to hand out the pointer of a segmented arena (NPB_EINVAL, npb_last_error
 * says why): a caller written before segments existed would compute arena + (column * pitch + plant) * width and touch the wrong
 * plants without any error.  A raw-arena dump is a checkpoint only together with (pitch, segment, columns, storage): record them.
 * Asking for the layout alone (arena = NULL) does not invalidate the maintenance cooldown cache; asking for the pointer does. */
NPB_API int npb_state_arena_layout(NpbHandle *h, void **arena, size_t *pitch, size_t *segment, int *columns, int *storage);
/* (With params.maint_enabled the step kernels consult a cache of which maintenance thresholds are inside their cooldown; every
 * entry point that can change state, the table or the clock invalidates it, this one included.  A caller that keeps the pointer
 * and writes the arena between later steps calls npb_state_arena again after each such write.) */
/* where a field lives: arena column, position of a narrow member inside the column (0/1), and how it is stored
 * (0 = carried real of the column width, 1 = output real stored as float, 2 = int32); element address =
 * arena + (column * pitch + plant) * width + sub * 4, width = 8 (NPB_STORAGE_F64) or 4 (NPB_STORAGE_F32) */
NPB_API int npb_locate_field(const NpbHandle *h, int kind, int slot, int *column, int *sub, int *access);

/* NuclearPlantSimulator.step (sim.py:130-258) for every plant.  Input columns (device, n_plants each)
 * may be NULL: action -> NO_ACTION(8), magnitude -> 1.0, power_setpoint -> unchanged
 * (NaN entries also mean unchanged; replaces heat_source.set_power_setpoint), noise_z -> 0
 * (standard-normal sample that ConstantHeatSource would draw, constant_heat_source.py:178),
 * cooling_water_temp -> unchanged.  Output columns (device) may be NULL:
 * obs [n,22] row-major, reward [n], done [n] u8, trip_flags [n] u32, info [n,NPB_INFO_DIM] (+ [n,NPB_INFO_NRHO] behind it with
 * params.info_reactivity_components).  Under NPB_HEAT_EXTERNAL the noise_z / power_setpoint columns carry the caller's heat
 * source (include/npb_params.h).
 * With params.maint_enabled the automatic maintenance of the feedwater pumps -- the handle's threshold table (npb_maint.h), the
 * orchestrator's rule, the work-order queue, the thirteen handlers -- runs after the physics, in the reference's order
 * (sim.py:208-223: AutoMaintenanceSystem.update, then the state manager's threshold scan), inside the same launch (full mode);
 * its state and counters are the maint.* / mpump.* columns of npb_fields.h. */
NPB_API int npb_step(NpbHandle *h, const int32_t *action, const double *magnitude, const double *power_setpoint,
             const double *noise_z, const double *cooling_water_temp, double *obs, double *reward, uint8_t *done,
             uint32_t *trip_flags, double *info, void *stream);

/* Have npb_step write the NPB_DIAG_* columns of every following step into buf ([NPB_DIAG_DIM][pitch] doubles on the handle's
 * device, pitch >= n_plants rounded up to a multiple of 64; NULL = off, the default).  While it is set the step runs the diagnostics build of the one-wave
 * kernel at every batch size (same results, ~1.4x the time at small batches); full mode only. */
NPB_API int npb_set_diagnostics(NpbHandle *h, double *buf, size_t pitch);

/* Which of the two step kernels npb_step launches (same device functions in the same per-plant order: identical int32
 * columns and flags, reals equal to the last bit or two; this is a measurement / A-B aid):
 * 0 = by batch size (default; also the environment variable NPB_STEP_KERNEL at handle creation), 1 = one wavefront per
 * 64 plants with an LDS-DMA staging pipeline, 2 = two wavefronts per 64 plants that own different subsystems (two builds of
 * it: the whole register file up to 32 768 plants, where a SIMD holds one wave anyway, 256 registers above), 3 = the
 * 256-register build at any size, 4 = the one-wavefront kernel with streaming (non-temporal) state stores, which 0 takes
 * above ~90 000 plants of fp64 storage, where nothing a step writes is still cached when the next step reads it, 5 = four
 * wavefronts per 64 plants handing values to each other through progress words in LDS (what 0 takes up to 32 768 plants,
 * where all of its 2 048 wavefronts are resident at once, and again between 45 057 and 114 688 plants, on the segmented
 * arena handles of that size have, npb_state_arena). */
NPB_API int npb_set_step_kernel(NpbHandle *h, int variant);
/* Which kernel the handle's last npb_step actually launched (NPB_KERNEL_NONE before the first step): the selection above is by
 * batch size, mode, storage and override, and a test or a benchmark that means to exercise one kernel asserts it here instead
 * of trusting the selection rule.  npb_step_kernel_name(id) = the kernel's symbol as rocprofv3 lists it. */
enum {
  NPB_KERNEL_NONE = 0,
  NPB_KERNEL_STEP = 1,         /* npb_step_kernel: one wavefront per 64 plants */
  NPB_KERNEL_STEP2_WIDE = 2,   /* npb_step2_wide_kernel: two wavefronts, the whole register file (<= 32 768 plants) */
  NPB_KERNEL_STEP2 = 3,        /* npb_step2_kernel: two wavefronts, 256 registers (two waves per SIMD) */
  NPB_KERNEL_STEP_NT = 4,      /* npb_step_nt_kernel: one wavefront, streaming state stores */
  NPB_KERNEL_STEP_DIAG = 5,    /* npb_step_diag_kernel: one wavefront, step-internal diagnostics written (npb_set_diagnostics) */
  NPB_KERNEL_STEP_PRIMARY = 6, /* npb_step_primary_kernel: NPB_MODE_PRIMARY */
  /* the builds of 1-4 with the automatic maintenance compiled in (params.maint_enabled, full mode): same step, same results */
  NPB_KERNEL_STEP_MAINT = 7, NPB_KERNEL_STEP2_WIDE_MAINT = 8, NPB_KERNEL_STEP2_MAINT = 9, NPB_KERNEL_STEP_NT_MAINT = 10,
  NPB_KERNEL_STEP4 = 11,       /* npb_step4_kernel: four wavefronts per 64 plants, 256 registers (two waves per SIMD at 32 768 plants) */
  NPB_KERNEL_STEP4_MAINT = 12,
  NPB_KERNEL_COUNT_
};
NPB_API int npb_debug_last_step_kernel(const NpbHandle *h);
NPB_API const char *npb_step_kernel_name(int kernel_id);

/* NuclearPlantSimulator.get_observation (sim.py:290-333) */
NPB_API int npb_observe(NpbHandle *h, double *obs, void *stream);

/* Measurement aid (no reference counterpart): streams every state column through the GPU unchanged,
 * 2 * npb_state_bytes() * pitch bytes with the step kernel's access shape; used to calibrate the
 * rocprofv3 FETCH_SIZE / WRITE_SIZE counters (tools/profile_traffic.py). */
NPB_API int npb_debug_touch(NpbHandle *h, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* NPB_H */
