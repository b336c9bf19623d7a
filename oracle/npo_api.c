/*
 * npo_api.c -- C entry points of the CPU oracle (built into oracle/libnpo.so).
 * TEST INFRASTRUCTURE ONLY: loaded by tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg; never by the product path.
 */
#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include "npo_step.h"
#include "npo_init.h"
#include "npo_reset.h"

#define NPO_API __attribute__((visibility("default")))

NPO_API int npo_plant_size(void) { return (int)sizeof(npo_plant_t); }
NPO_API int npo_num_f64(void) { return NPB_TOTAL_F64; }
NPO_API int npo_num_i32(void) { return NPB_TOTAL_I32; }
/* width of the info block npo_step_batch writes per plant: the binding sizes its buffer with it (a binding that assumed
 * fewer columns than the library writes is a host heap overrun, which is what crashed a round-2 test run) */
NPO_API int npo_info_dim(void) { return NPB_INFO_DIM; }
NPO_API int npo_params_size(void) { return (int)sizeof(npb_params_t); }
NPO_API void npo_params_default(npb_params_t *p) { npb_params_default(p); }

/* worker threads of npo_step_batch (bench.py's cpu_baseline times 1 thread and all allowed cores); returns the
 * count in effect (1 without OpenMP) */
NPO_API int npo_set_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
  return omp_get_max_threads();
#else
  (void)n; return 1;
#endif
}

NPO_API void npo_set_maint_table(const npb_maint_table_t *t) {
  if (t) npo_maint_table = *t; else npb_maint_table_default(&npo_maint_table);
  npo_maint_table_set = 1;
  npo_maint_table_custom = t != 0;
}
NPO_API int npo_maint_table_size(void) { return (int)sizeof(npb_maint_table_t); }
NPO_API void npo_default_maint_table(npb_maint_table_t *t) { npb_maint_table_default(t); }

/* plants: n contiguous npo_plant_t records */
NPO_API void npo_init(npo_plant_t *plants, int n, const npb_params_t *P) {
  if (!npo_maint_table_set) npo_set_maint_table(0);
  for (int i = 0; i < n; i++) npo_plant_init(&plants[i], P);
}
NPO_API double npo_get_f64(npo_plant_t *plants, int plant, int slot) { return *npo_f64_slot(&plants[plant], slot); }
NPO_API void npo_set_f64(npo_plant_t *plants, int plant, int slot, double v) { *npo_f64_slot(&plants[plant], slot) = v; }
NPO_API int npo_get_i32(npo_plant_t *plants, int plant, int slot) { return *npo_i32_slot(&plants[plant], slot); }
NPO_API void npo_set_i32(npo_plant_t *plants, int plant, int slot, int v) { *npo_i32_slot(&plants[plant], slot) = v; }
/* gather all slots of one plant (for trajectory comparison) */
NPO_API void npo_get_all(npo_plant_t *plants, int plant, double *f64, int32_t *i32) {
  for (int s = 0; s < NPB_TOTAL_F64; s++) f64[s] = *npo_f64_slot(&plants[plant], s);
  for (int s = 0; s < NPB_TOTAL_I32; s++) i32[s] = *npo_i32_slot(&plants[plant], s);
}
/* checker for the product's fp32-storage mode (include/npb.h, NPB_STORAGE_F32): that mode computes in fp64 and
 * rounds every real-valued state column to float when it is stored, once per step -- which is what this does
 * to the oracle's state between steps */
NPO_API void npo_round_state_f32(npo_plant_t *plants, int n, const uint8_t *keep_f64) {
  for (int i = 0; i < n; i++)
    for (int s = 0; s < NPB_TOTAL_F64; s++) {
      if (keep_f64 && keep_f64[s]) continue;   /* columns the mode keeps in fp64 (slow integrators) */
      double *v = npo_f64_slot(&plants[i], s); *v = (double)(float)*v;
    }
}

/* NuclearPlantSimulator.reset(start_at_steady_state) for n plants (npo_reset.h); mask NULL = all */
NPO_API void npo_reset_batch(npo_plant_t *plants, int n, const npb_params_t *P, const uint8_t *mask, int steady) {
  for (int i = 0; i < n; i++) {
    if (mask && !mask[i]) continue;
    npo_plant_t *pl = &plants[i];
    npo_prim_reset(&pl->prim);
    npo_sec_reset(&pl->sec);
    for (int k = 0; k < NPB_NUM_SG; k++) npo_sg_reset(&pl->sg[k]);
    npo_equilibrium_t eq;
    if (steady) {
      npo_steady_state_equilibrium(pl->sg, &pl->sec, P, P->rated_power_mw, &eq);
      npo_sec_steady_state(&pl->sec, &eq);
    }
    for (int k = 0; k < NPB_NUM_PUMPS; k++) {
      npo_pump_reset(&pl->pump[k], k);
      if (steady) npo_pump_steady_state(&pl->pump[k], k, eq.steam_pressure, eq.feedwater_flow, eq.pumps_needed, eq.pump_speed);
    }
    npo_fw_reset(&pl->fw);
    npo_turb_reset(&pl->turb, &pl->tstg, steady, steady ? eq.load_demand : 0.0, steady ? eq.electrical_power : 0.0);
    for (int k = 0; k < 2; k++) npo_chem_reset(&pl->chem[k]);
    npo_cond_reset(&pl->cond);
  }
}

/* One step for n plants. Per-plant input columns may be NULL (defaults: NO_ACTION,
 * magnitude 1, setpoint/cooling unchanged, z = 0). Outputs may be NULL. */
NPO_API void npo_step_batch(npo_plant_t *plants, int n, const npb_params_t *P,
                            const int32_t *action, const double *magnitude, const double *setpoint,
                            const double *noise_z, const double *cw_temp,
                            double *obs, double *reward, uint8_t *done, uint32_t *trip_flags, double *info) {
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
  for (int i = 0; i < n; i++) {
    npo_inputs_t in;
    in.action = action ? action[i] : 8;
    in.magnitude = magnitude ? magnitude[i] : 1.0;
    in.power_setpoint = setpoint ? setpoint[i] : NAN;
    in.noise_z = noise_z ? noise_z[i] : 0.0;
    in.cooling_water_temp = cw_temp ? cw_temp[i] : NAN;
    npo_outputs_t o;
    npo_step(&plants[i], P, &in, &o);
    if (obs) memcpy(obs + (size_t)i * NPB_OBS_DIM, o.obs, sizeof(o.obs));
    if (reward) reward[i] = o.reward;
    if (done) done[i] = o.done;
    if (trip_flags) trip_flags[i] = o.trip_flags;
    if (info) memcpy(info + (size_t)i * NPB_INFO_DIM, o.info, sizeof(o.info));
    if (info && P->info_reactivity_components && P->heat_source == NPB_HEAT_REACTOR)   /* second block, as npb_step lays it out */
      memcpy(info + (size_t)n * NPB_INFO_DIM + (size_t)i * NPB_INFO_NRHO, o.rho, sizeof(o.rho));
  }
}
NPO_API void npo_observe_batch(npo_plant_t *plants, int n, const npb_params_t *P, double *obs) {
  for (int i = 0; i < n; i++) npo_observation(&plants[i], P->mode, obs + (size_t)i * NPB_OBS_DIM);
}
