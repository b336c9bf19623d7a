/*
 * npd_step2.h -- the fused plant-step kernel, two wavefronts per 64 plants (included by npb_kernels.hip).
 *
 * Why: with one wavefront lane per plant, 65 536 plants are 1 024 wavefronts -- exactly one per SIMD of the chip.  A lone
 * wave issues at most one instruction every four cycles whatever its type (the CU's issue arbiter visits a SIMD every
 * fourth cycle), has nothing to hide a memory or LDS wait behind, and at half that batch half the SIMDs sit idle: the
 * one-wave kernel (npb_step_kernel) measured 53 % of its cycles issuing, and one wave's ~30 k instructions set a floor
 * of ~75 us per step however few plants there are.  The path has parallelism INSIDE a plant --
 * four pumps, three steam generators, the turbine's lubrication step, the chemistry sidecar, the per-stage metal
 * temperature tracker and the condenser do not all depend on each other -- so this kernel gives every group of
 * 64 plants TWO wavefronts (block = 128 threads; lane l of both waves is plant l) that own different subsystems and
 * hand the few coupling scalars to each other through LDS:
 *
 *   wave A: primary -> feedwater control -> pump 0, 1 | pump diagnostics + protection (all 4), system level |
 *           SG 0, SG 1 | turbine: stage pass A, half of pass B, stage chain, rotor / bearings / vibration,
 *           protection | electrical-power gates, feedback, observation, done, trip flags
 *   wave B: turbine lubrication pre-step, chemistry sidecar -> pump 2, 3 | SG 2 | other half of stage pass B |
 *           per-stage degradation + metal temperatures (the 70 stage-array columns) -> condenser | reward, info
 *
 * 2 048 wavefronts at 65 536 plants = two per SIMD (<= 256 registers, 40 KB of LDS per group): the SIMD issues
 * one wave's scalar / memory instructions beside the other's vector instructions and either wave's stalls are
 * covered by the other; at 32 768 plants every SIMD still has a wave.  State is read with plain global loads
 * straight into registers (no LDS staging pipeline: LDS is needed for the exchange, and the second wave hides the
 * latency) and written with the same unchanged-column elision as the one-wave kernel.
 *
 * Exactness.  Every device function is the one the one-wave kernel calls; sums over pumps / steam generators
 * are taken in the reference's order by wave A.  The one place where the reference's sequential loop makes a later
 * pump depend on an earlier one -- FeedwaterPumpSystem.update_system hands out flow demands only while fewer pumps
 * are RUNNING than ran in the previous step (pump_system.py:1262-1283) -- is decided before the pumps are split: if
 * for some lane that gate could close (more pumps RUNNING or STARTING than the previous running list holds), the whole
 * group runs the pumps one after the other, wave A's first, with the real counts.
 */
#ifndef NPD_STEP2_H
#define NPD_STEP2_H

#define NPD2_THREADS 128
#define NPD2_SPLIT 11                         /* stage pass B: wave A evaluates stages 0 .. NPD2_SPLIT-1 (and the inlet), wave B the rest and the five extractions */
#ifndef NPD2_OCCUPANCY
#define NPD2_OCCUPANCY __attribute__((amdgpu_waves_per_eu(2, 2)))
#endif
#define NPD2_SLOTS 80                         /* exchange slots of 64 doubles: 40 KB per group, 4 groups per CU */
#define NPD2_SYNC_() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#ifdef NPB_STAMPS
/* diagnostic build (tools/phase_stamps2.py): lane 0 of each wave stamps before and after barrier j (slots 2j-1, 2j) */
#define NPD2_STAMP(k) do { if (lane == 0 && npb_stamp_buf) npb_stamp_buf[((size_t)blockIdx.x * 2 + wave) * 32 + (k)] = __builtin_readcyclecounter(); } while (0)
#define NPD2_SYNCJ(j) do { NPD2_STAMP(2 * (j) - 1); NPD2_SYNC_(); NPD2_STAMP(2 * (j)); } while (0)
#else
#define NPD2_STAMP(k)
#define NPD2_SYNCJ(j) NPD2_SYNC_()
#endif
/* per-stage hand-over inside the turbine phase without a barrier: wave A raises the flag to k + 1 once stage k's outlet
 * temperature and loading factor are in LDS (LDS operations of a wave complete in order); wave B waits for it */
#define NPD2_FLAG2_SET(v) do { NPD_LDS_DRAIN(); *(volatile int *)&xch[X_FLAG2 * NPB_WAVE] = (v); } while (0)
#define NPD2_FLAG2_WAIT(v) do { while (__builtin_amdgcn_readfirstlane(*(volatile int *)&xch[X_FLAG2 * NPB_WAVE]) < (v)) __builtin_amdgcn_s_sleep(1); } while (0)
#define NPD2_FLAG_SET(v) do { NPD_LDS_DRAIN(); *(volatile int *)&xch[X_FLAG * NPB_WAVE] = (v); } while (0)
#define NPD2_FLAG_WAIT(v) do { while (__builtin_amdgcn_readfirstlane(*(volatile int *)&xch[X_FLAG * NPB_WAVE]) < (v)) __builtin_amdgcn_s_sleep(1); } while (0)
#define XW(slot, v) (xch[(slot) * NPB_WAVE + lane] = (v))
#define XR(slot) (xch[(slot) * NPB_WAVE + lane])

/* exchange slot plan (regions are reused once their readers are past the next barrier) */
enum {
  /* primary / feedwater control -> wave B */
  X_CFLOW0 = 0, X_CFLOW1, X_CFLOW2, X_CIN2, X_COUT2, X_LDF, X_FWTEMP, X_NPREV, X_FPP, X_MAXLVL,
  /* per-pump values for the diagnostics / protection passes: 13 per pump, pumps 0..3 */
  X_PUMP = 10, X_PUMP_N = 13,
  /* steam generators 1, 2 between their two parts: heat transfer, and the fp64 values of the two output members
   * (stored as float) that are still needed: TSP pressure-drop ratio, secondary temperature */
  X_SGCARRY = 62, X_FLAG2 = 68, X_MAINT_TAB = 69,
  X_MAINT_TIME = 67,   /* the plants' clock after this step, wave A -> wave B's pump phase (the steam-generator carry region is not in use yet) */
  X_FWFLOW = 70, X_RUNCOUNT = 71, X_CIN0 = 72, X_COUT0 = 73, X_CIN1 = 79, X_COUT1 = 74,   /* (74: before the turbine phase) */
  /* SG 1, SG 2 -> wave A (reuses the pump region): 6 values each */
  X_SG1 = 10, X_SG2 = 16,
  /* turbine */
  X_PSELF = 0, X_PEXT = 14, X_PIN = 19,             /* pass A results: 14 + 5 + 1 */
  X_SAT = 20, X_HG = 27, X_TRATIO = 34, X_HGEXT = 41, /* wave B's half of pass B: 7 + 7 + 7 + 5 */
  X_TOUT = 46, X_LOADING = 60,                        /* per stage, for the degradation / metal-temperature pass */
  X_MAXSTRESS = 74, X_EFFLOW = 75, X_LP6H = 76, X_CWT = 77, X_FLAG = 78,
  /* tail scalars for reward / info (reuses X_PSELF ...; three more where the feedwater hand-over was) */
  X_TAIL = 0, X_TAIL2 = 70,
  /* transposes */
  X_OBS = 22, X_INFO = 46
};
static_assert(NPD_MH_N <= NPB_WAVE && X_FLAG2 < X_MAINT_TAB && X_MAINT_TAB < X_FWFLOW && X_PUMP + 4 * X_PUMP_N <= X_SGCARRY && X_SGCARRY + 6 <= X_FLAG2 && X_OBS + NPB_OBS_PAD <= X_INFO && X_INFO + NPB_OBS_PAD <= X_TAIL2 && X_TAIL2 + 3 <= X_MAXSTRESS, "exchange slot plan");

/* [64][W] block held one row per lane -> row-major global memory through a transpose buffer of this wave's own */
template <int W>
__device__ __forceinline__ void npd2_store_rows(const double *row, double *__restrict__ out, double *buf, int lane,
                                                size_t block_base, size_t n_valid) {
#pragma unroll
  for (int j = 0; j < W; j++) buf[lane * NPB_OBS_PAD + j] = row[j];
  NPD_LDS_DRAIN();
#pragma unroll
  for (int k = 0; k < W; k++) {
    int idx = k * NPB_WAVE + lane;
    int r = idx / W, c = idx % W;
    if (block_base + r < n_valid) __builtin_nontemporal_store(buf[r * NPB_OBS_PAD + c], &out[block_base * W + idx]);
  }
  NPD_LDS_DRAIN();
}

/* the values of one updated pump that the diagnostics / protection passes and the system sums read */
__device__ __forceinline__ void npd2_publish_pump(double *xch, int lane, int i, const npb_pump_t &p) {
  const int b = X_PUMP + i * X_PUMP_N;
  XW(b + 0, (double)((p.status == NPD_PUMP_RUNNING) | (p.trip_active ? 2 : 0)));
  XW(b + 1, p.flow_rate); XW(b + 2, p.power_consumption); XW(b + 3, npd_pump_npsh_required(&p)); XW(b + 4, p.npsh_available);
  XW(b + 5, p.speed_percent); XW(b + 6, npd_pymax3(p.wear_motor_bearings, p.wear_pump_bearings, p.wear_thrust_bearing));
  XW(b + 7, p.wear_mechanical_seals); XW(b + 8, p.vibration_level); XW(b + 9, p.suction_pressure); XW(b + 10, p.discharge_pressure);
  XW(b + 11, p.oil_temperature); XW(b + 12, p.motor_temperature);
}

/* npd_fw_pump_step's tail for one pump, from the published values: the system sums of FeedwaterPumpSystem.update_system
 * (pump_system.py:1285-1329), this pump's pass through PerformanceDiagnostics.update_diagnostics and through the per-pump
 * loops of FeedwaterProtectionSystem.check_protection_systems (npd_feedwater.h, npd_fw_pump_step) */
__device__ __forceinline__ void npd2_pump_tail(const double *xch, int lane, int i, npb_fw_t *fw, npd_fw_acc_t *acc, double dt) {
  const int b = X_PUMP + i * X_PUMP_N;
  const int code = (int)XR(b + 0);
  const double flow_rate = XR(b + 1), power = XR(b + 2), npsh_required = XR(b + 3), npsh_available = XR(b + 4), speed_percent = XR(b + 5);
  const double max_bearing = XR(b + 6), seal_wear = XR(b + 7), vibration_level = XR(b + 8), suction_pressure = XR(b + 9);
  const double discharge_pressure = XR(b + 10), oil_temperature = XR(b + 11), motor_temperature = XR(b + 12);
  if (code & 1) { acc->total_flow += flow_rate; acc->total_power += power; acc->running_count++; acc->running_mask |= 1 << i; }
  if (code & 2) acc->trip_mask |= 1u << i;
  acc->flow_sum += flow_rate;
  {
    double cavitation_threshold = npsh_required + 2.0;
    double current_intensity;
    if (npsh_available < cavitation_threshold) {
      double npsh_deficit = cavitation_threshold - npsh_available;
      double severity = npd_pymin(1.0, npsh_deficit / cavitation_threshold);
      double flow_factor = npd_sq(flow_rate / 555.0);
      double speed_factor = npd_powc(speed_percent / 100.0, 1.5);
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
    acc->total_wear_level += (max_bearing + seal_wear);
    acc->total_vibration += vibration_level;
  }
  double dt_seconds = dt * 60.0;
  {
    int critical_active = 0;
    if (npsh_available < 0.1) {
      fw->npsh_low_low_timer += dt_seconds;
      if (fw->npsh_low_low_timer >= 5.0) fw->npsh_low_low_trip_active = 1;
    } else {
      fw->npsh_low_low_timer = 0.0;
      fw->npsh_low_low_trip_active = 0;
    }
    if (npsh_available < 0.1) critical_active = 1;
    if (critical_active || fw->npsh_low_low_trip_active) acc->trips++;
  }
  if (suction_pressure < 0.1) acc->trips++;
  if (discharge_pressure > 10.0) acc->trips++;
  if (vibration_level > 10.0) { fw->timer_vibration += dt_seconds; if (fw->timer_vibration >= 10.0) acc->trips++; }
  else fw->timer_vibration = 0.0;
  double bearing_temp = oil_temperature + 5.0;
  if (bearing_temp > 120.0) { fw->timer_bearing_temp += dt_seconds; if (fw->timer_bearing_temp >= 30.0) acc->trips++; }
  else fw->timer_bearing_temp = 0.0;
  if (motor_temperature > 130.0) { fw->timer_motor_temp += dt_seconds; if (fw->timer_motor_temp >= 60.0) acc->trips++; }
  else fw->timer_motor_temp = 0.0;
}

/* one pump of FeedwaterPumpSystem.update_system: the demand hand-out gate, then the pump itself */
__device__ __forceinline__ void npd2_pump(npb_pump_t *p, int gate_open, int n_prev_running, double flow_per_pump,
                                          const npd_pump_sysconds_t *sc, double dt) {
  if (p->status == NPD_PUMP_RUNNING && n_prev_running > 0 && gate_open) {
    if (!(flow_per_pump < NPD_PUMP_RATED_FLOW * 0.2)) npd_pump_set_flow_demand(p, flow_per_pump);
  }
  npd_pump_update(p, sc, dt);
}

/* stage pass A for all 14 stages (pressures, flows; no transcendentals): npd_stage_system_update's first block */
__device__ __forceinline__ bool npd2_stage_pass_a(double inlet_pressure, double inlet_flow, double load_demand,
                                                  double *p_self, double *flow_out, double *p_ext, double *ext_flow) {
#define NPD_EXT_IDX(k) ((k) == 2 ? 0 : (k) == 3 ? 1 : (k) == 4 ? 2 : (k) == 8 ? 3 : 4)
#define NPD_IS_EXT(k) ((k) == 2 || (k) == 3 || (k) == 4 || (k) == 8 || (k) == 9)
  bool rare = !(inlet_pressure >= 0.001 && inlet_pressure <= 22.0);
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
  return rare;
}

/* one stage of the temperature / enthalpy chain (npd_stage_system_update, pass C) without its degradation / metal part */
struct npd2_chain_t { double T_in, sat_in, hg_in, total_power, total_extraction, lp6_outlet_enthalpy, hp_power, lp_power, h_in0; };
__device__ __forceinline__ void npd2_chain_stage(int k, npd2_chain_t &c, double p_in, double p_self_k, double sat_k, double hg_k, double tratio_k,
                                                 double flow_out_k, double ef, double hg_ext_k, double total_efficiency,
                                                 double *T_out_o, double *loading_o) {
  double cp_in = (p_in > 10.0) ? 2.5 : ((p_in > 1.0) ? 2.2 : 2.0);
  double T_c = npd_pymax(0.0, npd_pymin(c.T_in, 800.0));
  double inlet_enthalpy = (T_c <= c.sat_in) ? c.hg_in : c.hg_in + cp_in * (T_c - c.sat_in);
  double T_isen = (c.T_in + 273.15) * tratio_k - 273.15;
  double T_isen_c = npd_pymax(0.0, npd_pymin(T_isen, 800.0));
  double cp_out = (p_self_k > 10.0) ? 2.5 : ((p_self_k > 1.0) ? 2.2 : 2.0);
  double h_isen = (T_isen_c <= sat_k) ? hg_k : hg_k + cp_out * (T_isen_c - sat_k);
  double isentropic_enthalpy_drop = inlet_enthalpy - h_isen;
  if (isentropic_enthalpy_drop <= 0) {
    double min_enthalpy_drop = 50.0 * (1.0 - p_self_k / p_in);
    isentropic_enthalpy_drop = npd_pymax(min_enthalpy_drop, 10.0);
  }
  double actual_enthalpy_drop = total_efficiency * isentropic_enthalpy_drop;
  if (actual_enthalpy_drop <= 0) actual_enthalpy_drop = npd_pymax(1.0, isentropic_enthalpy_drop * 0.5);
  double outlet_enthalpy = inlet_enthalpy - actual_enthalpy_drop;
  double T_out = (outlet_enthalpy <= hg_k) ? sat_k : sat_k + (outlet_enthalpy - hg_k) / 2.1;
  double main_power = flow_out_k * actual_enthalpy_drop / 1000.0;
  if (main_power < 0) main_power = 0.0;
  double extraction_power = 0.0;
  if (ef > 0) extraction_power = ef * (inlet_enthalpy - hg_ext_k) / 1000.0;
  *loading_o = actual_enthalpy_drop / npd_pymax(1.0, 0.88 * isentropic_enthalpy_drop);
  c.total_power += main_power + extraction_power; c.total_extraction += ef;
  if (k < 8) c.hp_power += main_power + extraction_power; else c.lp_power += main_power + extraction_power;
  if (k == 0) c.h_in0 = inlet_enthalpy;
  if (k == 13) c.lp6_outlet_enthalpy = outlet_enthalpy;
  *T_out_o = T_out;
  c.T_in = T_out; c.sat_in = sat_k; c.hg_in = hg_k;
}

/* the sequential stage chain of npd_stage_system_update_seq for one stage, without the degradation / metal part */
__device__ __forceinline__ void npd2_seq_stage(int k, double &cur_p, double &cur_T, double &cur_flow, double inlet_flow, double load_demand,
                                               double total_efficiency, npd2_chain_t &c, double *T_out_o, double *loading_o) {
  double extraction_demand = (k == 2) ? 25.0 * load_demand : (k == 3) ? 30.0 * load_demand : (k == 4) ? 20.0 * load_demand
                           : (k == 8) ? 15.0 * load_demand : (k == 9) ? 10.0 * load_demand : 0.0;
  double outlet_pressure = npd_stage_requested_outlet(k, cur_p, inlet_flow);
  npd_stage_out_t so;   /* the four efficiency factors enter the expansion only as their product (stage_system.py:209-212) */
  npd_stage_expansion(k, total_efficiency, 1.0, 1.0, 1.0, cur_p, cur_T, cur_flow, outlet_pressure, extraction_demand, &so);
  c.total_power += so.power_output; c.total_extraction += so.extraction_flow;
  if (k < 8) c.hp_power += so.power_output; else c.lp_power += so.power_output;
  if (k == 13) c.lp6_outlet_enthalpy = so.outlet_enthalpy;
  *T_out_o = so.outlet_temperature; *loading_o = so.loading_factor;
  cur_p = so.outlet_pressure; cur_T = so.outlet_temperature; cur_flow = so.outlet_flow;
}

/* npd_stage_post (npd_turbine.h) with the stage's old values already in registers; new values go straight to the arena */
#define NPD2_TSTG(member, k) (*NPD_RP(NPD_SEC_COL(TSTG, 0) + NPB_F64_SLOT(npb_tstg_t, member) + (k)))
struct npd2_tstg_old_t { double eff_deg[14], deposit[14], blade_wear[14], rotor_t[8], casing_t[6], blade_t[14]; };
/* stage k's old values as scalars (rotor_t / casing_t are read only where the stage has such a point); stress_out: rotor point
 * k's thermal stress (k < 8) */
/* DEG: this caller also advances the stage's efficiency degradation and deposit thickness (the four-wave kernel's chain wave does
 * that itself, from the copies it has to load anyway: npd_step4.h) */
template <bool DEG = true>
__device__ __forceinline__ void npd2_stage_post_vals(const npd_stage_t &st, int k, double eff_deg, double deposit, double blade_wear_old, double rotor_t,
                                                     double casing_t, double blade_t, double loading_factor, double outlet_temperature, double dt,
                                                     double *stress_out) {
  if constexpr (DEG) {
    NPD2_TSTG(stage_efficiency_degradation, k) = (npd_real_t)(eff_deg + 1e-05 * dt);
    NPD2_TSTG(stage_deposit_thickness, k) = (npd_real_t)(deposit + 5e-05 * dt);
  }
  double blade_wear = (1e-06 * dt) * npd_sq(loading_factor);
  NPD2_TSTG(stage_blade_wear_factor, k) = (npd_real_t)npd_pymax(0.7, blade_wear_old - blade_wear);
  const double time_constant = 3600.0 / 3600.0, ambient = 25.0;
  if (k < 8) {
    double rt = rotor_t;
    double tc = ((outlet_temperature - 50.0) - rt) / time_constant * dt;
    double max_rate = 5.0 * dt;
    tc = npd_clip(tc, -max_rate, max_rate);
    rt += tc;
    NPD2_TSTG(rotor_temperatures, k < 8 ? k : 0) = (npd_real_t)rt;
    *stress_out = (1.2e-05 * (rt - ambient)) * 200000000000.0 * 0.1;
  }
  if (k < 6) {
    double ct = casing_t;
    double tc = ((outlet_temperature - 80.0) - ct) / time_constant * dt;
    tc = npd_clip(tc, -3.0 * dt, 3.0 * dt);
    NPD2_TSTG(casing_temperatures, k < 6 ? k : 0) = (npd_real_t)(ct + tc);
  }
  {
    double bt = blade_t;
    double tc = ((outlet_temperature - 20.0) - bt) / (time_constant * 0.5) * dt;
    tc = npd_clip(tc, -10.0 * dt, 10.0 * dt);
    NPD2_TSTG(blade_temperatures, k) = (npd_real_t)(bt + tc);
  }
}
__device__ __forceinline__ void npd2_stage_post_one(const npd_stage_t &st, const npd2_tstg_old_t &o, int k, double loading_factor,
                                                    double outlet_temperature, double dt, double *stress_out) {
  npd2_stage_post_vals(st, k, o.eff_deg[k], o.deposit[k], o.blade_wear[k], o.rotor_t[k < 8 ? k : 0], o.casing_t[k < 6 ? k : 0], o.blade_t[k],
                       loading_factor, outlet_temperature, dt, stress_out);
}
/* the same, folding MetalTemperatureTracker's max over the rotor points in stage order */
__device__ __forceinline__ void npd2_stage_post(const npd_stage_t &st, const npd2_tstg_old_t &o, int k, double loading_factor,
                                                double outlet_temperature, double dt, double *max_thermal_stress) {
  double stress = 0.0;
  npd2_stage_post_one(st, o, k, loading_factor, outlet_temperature, dt, &stress);
  if (k < 8) *max_thermal_stress = (k == 0) ? stress : npd_pymax(*max_thermal_stress, stress);
}

/* the body of both two-wave kernels (below): same code, compiled once per register budget */
/* MAINT: with the automatic maintenance compiled in (see npd_step1.h, NPD_STEP1_MAINT) */
template <int WHO, bool MAINT>
__device__ __forceinline__ void npd_step2_body(
    const npb_params_t &P, int n_plants, size_t N, npd_real_t *__restrict__ f64,
    const int32_t *__restrict__ action, const double *__restrict__ magnitude, const double *__restrict__ setpoint,
    const double *__restrict__ noise_z, const double *__restrict__ cw_temp, double *__restrict__ obs_out,
    double *__restrict__ reward_out, uint8_t *__restrict__ done_out, uint32_t *__restrict__ trip_out,
    double *__restrict__ info_out, const npd_maint_hot_t &MH, const npd_maint_rule_consts_t *maint_rc, const npd_maint_cache_t &MC) {
  __shared__ __attribute__((aligned(16))) double xch[NPD2_SLOTS * NPB_WAVE];
  const int lane = threadIdx.x & (NPB_WAVE - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   /* wave-uniform role: 0 = A, 1 = B */
  const size_t block_base = (size_t)blockIdx.x * NPB_WAVE;
  NPD_SEGMENT(f64, N, block_base);
  const size_t p = block_base + lane;
  const bool live = p < (size_t)n_plants;
  const double dt = P.dt;
  const bool kinetics = P.heat_source == NPB_HEAT_REACTOR;
  npd_stage_t st;
  {
    st.lds = (char *)xch;
    st.f64b = (npd_gchar_t *)(f64 + block_base);
    st.nr = (uint32_t)(N * NPD_RB);
    st.laner = (uint32_t)lane * NPD_RB;
    st.grp16 = 0;
  }

  /* automatic maintenance on: the folded threshold table (npd_maintenance.h, "the threshold screen inside the step kernels")
   * goes to exchange slot X_MAINT_TAB, which nothing else uses, written by wave A before barrier #1 and read by both waves'
   * pump phases behind it */
  const bool maint = MAINT && P.maint_enabled && maint_rc != nullptr;
  unsigned maint_hit_bits = 0, maint_due_with_orders = 0;     /* wave-uniform: what this wave's part of the screen found */
  double *const maint_tab = xch + X_MAINT_TAB * NPB_WAVE;
  NPD2_STAMP(0);
  if (wave == 0) {
    /* =========================================== wave A =========================================== */
    double maint_entry = 0.0;
    if (maint && lane < NPD_MH_N) maint_entry = MH.tab[lane];
    npd_maint_due_t maint_due = {};
    npd_u32x4 maint_cache01 = {0, 0, 0, 0};       /* {mask, until} of this wave's pumps 0, 1 (npd_maintenance.h) */
    if (maint) { npd_maint_due_load(&maint_due, f64, N, p); maint_cache01 = npd_maint_cache_fetch(MC, p, 0); }
    npd_inputs_t in;
    in.action = (live && action) ? action[p] : 8;
    in.magnitude = (live && magnitude) ? magnitude[p] : 1.0;
    in.power_setpoint = (live && setpoint) ? setpoint[p] : NAN;
    in.noise_z = (live && noise_z) ? noise_z[p] : 0.0;
    in.cooling_water_temp = (live && cw_temp) ? cw_temp[p] : NAN;
    double base_reward, load_demand, cooling_water_temperature, primary_thermal_power = 0.0;
    int scram_fired, nan_reset, scram_status;
    double thermal_power_info, reactivity_info, time_info;
    /* ---- phase 0: primary side + coupling (sim.py:141-161) */
    {
      npb_prim_t s;
      if (kinetics) {
        NPD_ST_LOAD(PRIM, npb_prim_t, s, 0);
      } else {   /* the point-kinetics columns stay where they are under ConstantHeatSource */
        double *d = reinterpret_cast<double *>(&s);
#pragma unroll
        for (int k = 0; k < NPD_PRIM_KIN0; k++) d[k] = (double)*NPD_RP(NPD_SEC_COL(PRIM, 0) + k);
#pragma unroll
        for (int k = NPD_PRIM_KIN0; k < NPB_PRIM_NCARRY; k++) d[k] = 0.0;
#pragma unroll
        for (int j = 0; j < NPB_PRIM_NOUT; j++) d[NPB_PRIM_NCARRY + j] = (double)*NPD_NP(const float, NPD_SEC_COL(PRIM, 0) + NPB_PRIM_NCARRY + j / NPD_NPC, j % NPD_NPC);
        int32_t *q = reinterpret_cast<int32_t *>(d + NPB_PRIM_NF64);
#pragma unroll
        for (int k = 0; k < NPB_PRIM_NI32; k++) q[k] = *NPD_NP(const int32_t, NPD_SEC_COL(PRIM, 0) + NPB_PRIM_NCARRY + (NPB_PRIM_NOUT + k) / NPD_NPC, (NPB_PRIM_NOUT + k) % NPD_NPC);
      }
      const npb_prim_t s_old = s;
      if (P.heat_source != NPB_HEAT_EXTERNAL && !isnan(in.power_setpoint)) s.hs_setpoint_percent = npd_clip(in.power_setpoint, 0.0, 150.0);
      double rho[NPB_INFO_NRHO];
      scram_fired = npd_primary_update(&s, &P, &in, &nan_reset, rho);
      npd_store_reactivity_components(P, rho, info_out, n_plants, p);
      npd_coupling_t c;
      npd_primary_to_secondary(&s, &c);
      /* the per-loop conditions go to LDS: wave B's steam generators read them there, and so does this wave later
       * (holding them in registers across the pump phase costs more than three LDS reads) */
      XW(X_CFLOW0, c.flow[0]); XW(X_CFLOW1, c.flow[1]); XW(X_CFLOW2, c.flow[2]);
      XW(X_CIN0, c.inlet_temp[0]); XW(X_COUT0, c.outlet_temp[0]); XW(X_CIN1, c.inlet_temp[1]); XW(X_COUT1, c.outlet_temp[1]);
      XW(X_CIN2, c.inlet_temp[2]); XW(X_COUT2, c.outlet_temp[2]);
#pragma unroll
      for (int i = 0; i < NPB_NUM_SG; i++) primary_thermal_power += c.thermal_power[i];
      s.sim_time += dt;
      load_demand = s.power_level;
      scram_status = s.scram_status;
      double power_reward = -fabs(s.power_level - 100) / 100;
      double temp_penalty = 0, pressure_penalty = 0;
      if (s.fuel_temperature > 800) temp_penalty = -(s.fuel_temperature - 800) / 100;
      if (s.coolant_pressure > 16) pressure_penalty = -(s.coolant_pressure - 16);
      double scram_penalty = s.scram_status ? -100 : 0;
      base_reward = power_reward + temp_penalty + pressure_penalty + scram_penalty;
      thermal_power_info = s.thermal_power_mw; reactivity_info = s.total_reactivity_pcm; time_info = s.sim_time;
      s.has_heat_removal_factor = 1;
      NPD_ST_STORE_ELIDE_PRIM(s, s_old);
      if (maint) {   /* sim.py:208-216 as far as no work order is involved; t = the clock after this step */
        XW(X_MAINT_TIME, s.sim_time);
        const bool work = npd_maint_due_decide(&maint_due, s.sim_time, MH.tab[2 * NPB_MAINT_NPARAM + 1]);
        maint_due_with_orders = __builtin_amdgcn_ballot_w64(work) != 0 ? 1u : 0u;
      }
    }
    /* ---- secondary prelude (secondary/__init__.py:371-453) */
    cooling_water_temperature = (double)NPD_ST_F64(SEC, npb_sec_t, cooling_water_temperature, 0, 0);
    const double cw_old = cooling_water_temperature;
    const double prev_feedwater_temp = (double)NPD_ST_F64(SEC, npb_sec_t, previous_feedwater_temp, 0, 0);
    const double operating_hours = (double)NPD_ST_F64(SEC, npb_sec_t, operating_hours, 0, 0);
    const int has_prev = *NPD_NP(const int32_t, NPD_SEC_COL(SEC, 0) + NPB_SEC_NCARRY + (NPB_SEC_NOUT + NPB_I32_SLOT(npb_sec_t, SEC, has_previous_sg_conditions)) / NPD_NPC,
                                 (NPB_SEC_NOUT + NPB_I32_SLOT(npb_sec_t, SEC, has_previous_sg_conditions)) % NPD_NPC);
    double prev_levels[NPB_NUM_SG], prev_flows[NPB_NUM_SG], prev_quals[NPB_NUM_SG];
#pragma unroll
    for (int i = 0; i < NPB_NUM_SG; i++) {
      prev_levels[i] = (double)NPD_ST_F64(SEC, npb_sec_t, prev_sg_levels, 0, i);
      prev_flows[i] = (double)NPD_ST_F64(SEC, npb_sec_t, prev_sg_steam_flows, 0, i);
      prev_quals[i] = (double)NPD_ST_F64(SEC, npb_sec_t, prev_sg_qualities, 0, i);
    }
    if (!isnan(in.cooling_water_temp)) cooling_water_temperature = in.cooling_water_temp;
    const double actual_feedwater_temp = (0.1 * (40.0 + 187.0) + (1 - 0.1) * prev_feedwater_temp);
    double load_demand_fraction = npd_pymin(1.0, primary_thermal_power / 3000.0);
    load_demand_fraction = npd_pymax(load_demand_fraction, 0.2);
    if (!has_prev) {
#pragma unroll
      for (int i = 0; i < NPB_NUM_SG; i++) { prev_levels[i] = 12.5; prev_flows[i] = 555.0 * load_demand_fraction; prev_quals[i] = 0.99; }
    }
    /* ---- feedwater control, and what wave B needs to start its pumps and its steam generator */
    npb_fw_t fw;
    NPD_ST_LOAD(FW, npb_fw_t, fw, 0);
    const npb_fw_t fw_old = fw;
    const double total_flow_demand = npd_fw_level_control(&fw, prev_levels, prev_flows, prev_quals, dt);
    int n_prev_running = 0;
#pragma unroll
    for (int i = 0; i < NPB_NUM_PUMPS; i++) n_prev_running += (fw.running_mask >> i) & 1;
    const double flow_per_pump = (n_prev_running > 0) ? total_flow_demand / n_prev_running : 0.0;
    npd_pump_sysconds_t sc;
    sc.feedwater_temperature = 40.0; sc.suction_pressure = 0.5; sc.discharge_pressure = 7.4;
    sc.max_sg_level = npd_pymax3(prev_levels[0], prev_levels[1], prev_levels[2]);
    XW(X_LDF, load_demand_fraction); XW(X_FWTEMP, actual_feedwater_temp);
    XW(X_NPREV, (double)n_prev_running); XW(X_FPP, flow_per_pump); XW(X_MAXLVL, sc.max_sg_level);
    /* could the demand gate close for a later pump?  (both waves evaluate this on the same data) */
    int may_run = 0;
#pragma unroll
    for (int i = 0; i < NPB_NUM_PUMPS; i++) {
      const int stt = *NPD_NP(const int32_t, NPD_SEC_COL(PUMP, i) + NPB_PUMP_NCARRY + (NPB_PUMP_NOUT + NPB_I32_SLOT(npb_pump_t, PUMP, status)) / NPD_NPC,
                              (NPB_PUMP_NOUT + NPB_I32_SLOT(npb_pump_t, PUMP, status)) % NPD_NPC);
      may_run += (stt == NPD_PUMP_RUNNING || stt == NPD_PUMP_STARTING);
    }
    const bool serial_pumps = __builtin_amdgcn_ballot_w64(n_prev_running > 0 && may_run > n_prev_running) != 0;
    if (maint && lane < NPD_MH_N) maint_tab[lane] = maint_entry;
    NPD2_SYNCJ(1);                                                                                     /* #1 */
    if (maint) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); npd_maint_cache_landed(maint_cache01); }
    /* ---- pumps 0 and 1 */
    int running_count = 0;
#pragma unroll 1
    for (int i = 0; i < 2; i++) {
      npb_pump_t pm;
      NPD_ST_LOAD(PUMP, npb_pump_t, pm, i);
      const npb_pump_t pm_old = pm;
      const uint32_t cooling_mask = i == 0 ? maint_cache01.x : maint_cache01.z;     /* this (plant, pump)'s entry of the cooldown cache */
      const float cooling_until = __uint_as_float(i == 0 ? maint_cache01.y : maint_cache01.w);
      npd2_pump(&pm, running_count < n_prev_running, n_prev_running, flow_per_pump, &sc, dt);
      running_count += pm.status == NPD_PUMP_RUNNING;
      npd2_publish_pump(xch, lane, i, pm);
      if (maint) {   /* anything new at this pump, for any plant of the group?  (npd_maintenance.h) */
        if (__builtin_amdgcn_ballot_w64(npd_maint_pump_hit(&pm, maint_tab, cooling_mask, cooling_until, time_info)) != 0) maint_hit_bits |= 1u << i;
      }
      NPD_ST_STORE_ELIDE(PUMP, npb_pump_t, pm, pm_old, i);
    }
    if (serial_pumps) { XW(X_RUNCOUNT, (double)running_count); NPD2_SYNC_(); }                       /* #1b */
    NPD2_SYNCJ(2);                                                                                     /* #2 */
    /* ---- diagnostics + protection passes over the four pumps, system level (feedwater/physics.py:720-863) */
    npd_fw_acc_t acc;
    acc.total_flow = acc.total_power = acc.flow_sum = 0.0;
    acc.total_cavitation_risk = acc.total_wear_level = acc.total_vibration = 0.0;
    acc.running_count = acc.running_mask = acc.trips = 0; acc.trip_mask = 0; acc.trip_kinds = 0;
#pragma unroll
    for (int i = 0; i < NPB_NUM_PUMPS; i++) npd2_pump_tail(xch, lane, i, &fw, &acc, dt);
    npd_fw_result_t fwr;
    npd_fw_finish(&fw, &acc, prev_levels, dt, &fwr);
    const double fw_total_flow = fwr.total_flow_rate, fw_total_power = fwr.total_power_consumption;
    const int fw_available = fwr.system_availability;
    uint32_t trip_flags = (fwr.pump_trip_mask << 8) | (fw.system_trip_active ? NPB_TRIP_FW_SYSTEM : 0);
    /* wave B's steam generators wait for this in their part 2 (a flag, not a barrier: this wave goes on to SG 0 without
     * waiting for wave B's part 1); raising it also tells wave B that the pump region has been read */
    XW(X_FWFLOW, fw_total_flow);
    NPD2_FLAG2_SET(1);
    NPD_ST_STORE_ELIDE(FW, npb_fw_t, fw, fw_old, 0);
    /* ---- steam generator 0 (enhanced_physics.py:433-547); wave B runs SG 1 and SG 2 */
    double sg_total_thermal = 0.0, sg_total_steam = 0.0, sg_ap = 0.0, sg_at = 0.0, sg_aq = 0.0;
    double sg_pressures[NPB_NUM_SG];
    int sg_effective = 0;
    {
      const double actual_total_steam_flow = P.sg_design_total_steam_flow * load_demand_fraction;
      const double c_flow = XR(X_CFLOW0), c_inlet = XR(X_CIN0), c_outlet = XR(X_COUT0);
      double total_primary_flow = 0.0;
      total_primary_flow += XR(X_CFLOW0); total_primary_flow += XR(X_CFLOW1); total_primary_flow += XR(X_CFLOW2);
      double demand = (total_primary_flow > 0) ? actual_total_steam_flow * (c_flow / total_primary_flow) : actual_total_steam_flow / NPB_NUM_SG;
      npb_sg_t g;
      NPD_ST_LOAD(SG, npb_sg_t, g, 0);
      const npb_sg_t g_old = g;
      npd_sg_result_t r;
      r.heat_transfer_rate = 0.0; r.steam_flow_rate = 0.0; r.thermal_efficiency = 0.0;
      npd_sg_update(&g, &P, c_inlet, c_outlet, c_flow, demand, fw_total_flow / NPB_NUM_SG, actual_feedwater_temp, dt * 60, &r);
      sg_total_thermal += r.heat_transfer_rate; sg_total_steam += r.steam_flow_rate;
      sg_ap += g.secondary_pressure; sg_at += g.secondary_temperature; sg_aq += g.steam_quality;
      sg_pressures[0] = g.secondary_pressure;
      if (r.thermal_efficiency > 0.1) sg_effective++;
      NPD_ST_STORE_ELIDE(SG, npb_sg_t, g, g_old, 0);
      NPD_ST_F64_ELIDE(SEC, npb_sec_t, prev_sg_levels, 0, 0, g.water_level, prev_levels[0]);
      NPD_ST_F64(SEC, npb_sec_t, prev_sg_steam_flows, 0, 0) = (npd_real_t)r.steam_flow_rate;
      NPD_ST_F64(SEC, npb_sec_t, prev_sg_qualities, 0, 0) = (npd_real_t)g.steam_quality;
    }
    /* while wave B is on its second steam generator: the turbine section and the 14 stages' efficiency products
     * (TurbineStage state as the previous step left it, stage_system.py:128-133, 294-339) */
    const double tdt = dt / 60.0;
    npb_turb_t t;
    NPD_ST_LOAD(TURB, npb_turb_t, t, 0);          /* wave B owns the lub_* members; they are neither used nor stored here */
    const npb_turb_t t_old = t;
    double stage_eff[14];
#pragma unroll
    for (int k = 0; k < 14; k++) {
      double fouling_factor = 1.0 / (1.0 + (double)NPD2_TSTG(stage_deposit_thickness, k) / 0.5);
      double blade_wear_factor = (double)NPD2_TSTG(stage_blade_wear_factor, k);
      double blade_condition_factor = npd_pymin(fouling_factor, blade_wear_factor);
      double actual_efficiency = npd_pymax(0.7, 0.88 - (double)NPD2_TSTG(stage_efficiency_degradation, k));
      stage_eff[k] = (actual_efficiency * blade_condition_factor * fouling_factor * blade_wear_factor * 1.0);
    }
    NPD2_SYNCJ(5);                                                                                     /* #4 */
#pragma unroll
    for (int i = 1; i < NPB_NUM_SG; i++) {
      const int b = i == 1 ? X_SG1 : X_SG2;
      sg_total_thermal += XR(b + 0); sg_total_steam += XR(b + 1);
      sg_ap += XR(b + 2); sg_at += XR(b + 3); sg_aq += XR(b + 4);
      sg_pressures[i] = XR(b + 2);
      if (XR(b + 5) != 0.0) sg_effective++;
    }
    const double sg_avg_pressure = sg_ap / NPB_NUM_SG, sg_avg_temperature = sg_at / NPB_NUM_SG, sg_avg_quality = sg_aq / NPB_NUM_SG;
    const int sg_system_availability = sg_effective >= (NPB_NUM_SG - 1);
    /* ---- turbine (dt in hours, load demand in PERCENT, secondary/__init__.py:564-569) */
    t.load_demand = load_demand;
    const double pressure_stability_factor = npd_pressure_stability_factor(sg_pressures);
    double p_self[14], flow_out[14], p_ext[5], ext_flow[5];
    const bool rare = npd2_stage_pass_a(sg_avg_pressure, sg_total_steam, load_demand, p_self, flow_out, p_ext, ext_flow);
    const bool seq = __builtin_amdgcn_ballot_w64(rare) != 0;
#pragma unroll
    for (int k = 0; k < 14; k++) XW(X_PSELF + k, p_self[k]);
#pragma unroll
    for (int e = 0; e < 5; e++) XW(X_PEXT + e, p_ext[e]);
    XW(X_PIN, seq ? NAN : sg_avg_pressure);         /* NaN tells wave B that the group takes the sequential chain */
    XW(X_CWT, cooling_water_temperature);
    NPD2_SYNCJ(7);                                                                                     /* #5 */
    npd2_chain_t ch;
    ch.T_in = sg_avg_temperature; ch.total_power = 0.0; ch.total_extraction = 0.0; ch.lp6_outlet_enthalpy = 0.0;
    ch.hp_power = 0.0; ch.lp_power = 0.0; ch.h_in0 = 0.0;
    double turbine_efficiency = 0.0;   /* stage_system.py:983-993 (info only) */
    if (!seq) {
      /* pass B, stages 0 .. NPD2_SPLIT-1 and the inlet (wave B does the rest and the five extraction pressures) */
      double sat_a[NPD2_SPLIT], hg_a[NPD2_SPLIT], tr_a[NPD2_SPLIT];
      ch.sat_in = npd_tsat_antoine(sg_avg_pressure);
      ch.hg_in = npd_hg_from_tsat(ch.sat_in);
#pragma unroll
      for (int k = 0; k < NPD2_SPLIT; k++) {
        sat_a[k] = npd_tsat_antoine(p_self[k]);
        hg_a[k] = npd_hg_from_tsat(sat_a[k]);
        tr_a[k] = npd_sqrt(npd_sqrt(p_self[k] / ((k == 0) ? sg_avg_pressure : p_self[k > 0 ? k - 1 : 0])));
      }
      NPD2_SYNCJ(8);                                                                                   /* #6 */
#pragma unroll
      for (int k = 0; k < 14; k++) {
        const double p_in = (k == 0) ? sg_avg_pressure : p_self[k > 0 ? k - 1 : 0];
        const double sat_k = k < NPD2_SPLIT ? sat_a[k < NPD2_SPLIT ? k : 0] : XR(X_SAT + (k < NPD2_SPLIT ? 0 : k - NPD2_SPLIT));
        const double hg_k = k < NPD2_SPLIT ? hg_a[k < NPD2_SPLIT ? k : 0] : XR(X_HG + (k < NPD2_SPLIT ? 0 : k - NPD2_SPLIT));
        const double tr_k = k < NPD2_SPLIT ? tr_a[k < NPD2_SPLIT ? k : 0] : XR(X_TRATIO + (k < NPD2_SPLIT ? 0 : k - NPD2_SPLIT));
        const double ef = NPD_IS_EXT(k) ? ext_flow[NPD_EXT_IDX(k)] : 0.0;
        const double hgx = NPD_IS_EXT(k) ? XR(X_HGEXT + NPD_EXT_IDX(k)) : 0.0;
        double T_out, loading;
        npd2_chain_stage(k, ch, p_in, p_self[k], sat_k, hg_k, tr_k, flow_out[k], ef, hgx, stage_eff[k], &T_out, &loading);
        XW(X_TOUT + k, T_out); XW(X_LOADING + k, loading);
        NPD2_FLAG_SET(k + 1);
      }
      {   /* _steam_enthalpy at the last stage's outlet, whose saturation state pass B has */
        const double T_c = npd_pymax(0.0, npd_pymin(ch.T_in, 800.0));
        const double cp = (p_self[13] > 10.0) ? 2.5 : ((p_self[13] > 1.0) ? 2.2 : 2.0);
        const double h_out = (T_c <= ch.sat_in) ? ch.hg_in : ch.hg_in + cp * (T_c - ch.sat_in);
        if (sg_total_steam > 0) turbine_efficiency = (ch.h_in0 - h_out) / ch.h_in0;
      }
    } else {
      NPD2_SYNCJ(8);                                                                                   /* #6 */
      double cur_p = sg_avg_pressure, cur_T = sg_avg_temperature, cur_flow = sg_total_steam;
#pragma unroll
      for (int k = 0; k < 14; k++) {
        double T_out, loading;
        npd2_seq_stage(k, cur_p, cur_T, cur_flow, sg_total_steam, load_demand, stage_eff[k], ch, &T_out, &loading);
        XW(X_TOUT + k, T_out); XW(X_LOADING + k, loading);
        NPD2_FLAG_SET(k + 1);
      }
      if (sg_total_steam > 0) {
        const double h_in = npd_stage_steam_enthalpy(sg_avg_temperature, sg_avg_pressure);
        turbine_efficiency = (h_in - npd_stage_steam_enthalpy(cur_T, cur_p)) / h_in;
      }
    }
    const double stage_power_mw = ch.total_power * pressure_stability_factor;
    XW(X_EFFLOW, sg_total_steam - ch.total_extraction); XW(X_LP6H, ch.lp6_outlet_enthalpy);
    double max_bearing_metal, total_displacement;
    npd_turbine_rotor(&t, stage_power_mw, sg_avg_temperature, load_demand, tdt, &max_bearing_metal, &total_displacement);
    NPD2_SYNCJ(9);                                                                                     /* #7: wave B has finished the stage arrays */
    const double max_stress = XR(X_MAXSTRESS);
    npd_turbine_protect(&t, stage_power_mw, max_stress, max_bearing_metal, total_displacement, sg_system_availability, 0.007, tdt);
    {   /* store the turbine section but for wave B's lub_* members */
      npb_turb_t ts = t;
      constexpr int L0 = NPB_F64_SLOT(npb_turb_t, lub_oil_temperature), L1 = NPB_F64_SLOT(npb_turb_t, thermal_expansion);
      const double *d = reinterpret_cast<const double *>(&ts), *od = reinterpret_cast<const double *>(&t_old);
#pragma unroll
      for (int k = 0; k < NPB_TURB_NCARRY; k++) {
        if (k >= L0 && k < L1) continue;
        if ((NPD_ELIDE_TURB_F >> k) & 1) {
          if (__builtin_amdgcn_ballot_w64(npd_real_bits(d[k]) != npd_real_bits(od[k])) != 0) *NPD_RP(NPD_SEC_COL(TURB, 0) + k) = (npd_real_t)d[k];
        } else {
          *NPD_RP(NPD_SEC_COL(TURB, 0) + k) = (npd_real_t)d[k];
        }
      }
      /* narrow members: thermal_expansion, total_power_output, vibration_displacement as float + the two int32; the
       * fourth output, lub_effectiveness, is wave B's: it is stored by wave B into its own 4 bytes */
      static_assert(NPB_TURB_NOUT == 4 && NPB_TURB_NI32 == 2, "turbine narrow layout");
      constexpr int NC = NPB_TURB_NCARRY;
      *NPD_NP(float, NPD_SEC_COL(TURB, 0) + NC + 0 / NPD_NPC, 0 % NPD_NPC) = (float)t.thermal_expansion;
      *NPD_NP(float, NPD_SEC_COL(TURB, 0) + NC + 1 / NPD_NPC, 1 % NPD_NPC) = (float)t.total_power_output;
      *NPD_NP(float, NPD_SEC_COL(TURB, 0) + NC + 2 / NPD_NPC, 2 % NPD_NPC) = (float)t.vibration_displacement;
      *NPD_NP(int32_t, NPD_SEC_COL(TURB, 0) + NC + 4 / NPD_NPC, 4 % NPD_NPC) = t.trip_active;
      *NPD_NP(int32_t, NPD_SEC_COL(TURB, 0) + NC + 5 / NPD_NPC, 5 % NPD_NPC) = t.trip_latched_mask;
    }
    /* ---- electrical-power gates (secondary/__init__.py:750-932) */
    const double turbine_electrical_power = t.total_power_output * 0.98;
    const double total_system_heat_rejection = (primary_thermal_power - turbine_electrical_power) * 1e6;
    double power_reduction_factor = 1.0;
    if (fw_total_flow < 300.0) power_reduction_factor = 0.0;
    if (power_reduction_factor > 0.0) {
      if (sg_total_steam < (300.0 * 0.5)) power_reduction_factor *= 0.1;
      if (sg_avg_pressure < (1.0 * 0.5)) power_reduction_factor *= 0.1;
      if (primary_thermal_power > (primary_thermal_power * 1.1)) power_reduction_factor = 0.0;
    }
    const double electrical_power = turbine_electrical_power * power_reduction_factor;
    const double thermal_efficiency = (primary_thermal_power > 0) ? electrical_power / primary_thermal_power : 0.0;
    if (t.trip_active) trip_flags |= NPB_TRIP_TURBINE;
    /* what wave B needs for reward and info */
    XW(X_TAIL + 0, base_reward); XW(X_TAIL + 1, electrical_power); XW(X_TAIL + 2, thermal_efficiency); XW(X_TAIL + 3, load_demand);
    XW(X_TAIL + 4, sg_avg_pressure); XW(X_TAIL + 5, sg_total_steam); XW(X_TAIL + 6, fw_total_flow); XW(X_TAIL + 7, total_system_heat_rejection);
    XW(X_TAIL + 8, thermal_power_info); XW(X_TAIL + 9, reactivity_info); XW(X_TAIL + 10, time_info);
    XW(X_TAIL + 11, sg_total_thermal); XW(X_TAIL + 12, sg_avg_temperature); XW(X_TAIL + 13, sg_avg_quality);
    XW(X_TAIL + 14, (double)(sg_system_availability | (fw_available << 1))); XW(X_TAIL + 15, prev_feedwater_temp); XW(X_TAIL + 16, cw_old);
    XW(X_TAIL + 17, operating_hours); XW(X_TAIL + 18, t.total_power_output); XW(X_TAIL + 19, fw_total_power); XW(X_TAIL + 20, primary_thermal_power);
    XW(X_TAIL2 + 0, turbine_efficiency); XW(X_TAIL2 + 1, ch.hp_power); XW(X_TAIL2 + 2, ch.lp_power);
    NPD2_SYNCJ(11);                                                                                     /* #9 */
    /* ---- observation, done, trip flags: the primary part (sim.py:290-333) from the primary section as stored in phase 0
     * (carried members: the stored value is the value; power_level, an output member, was kept) */
    double obs[NPB_OBS_DIM];
    obs[0] = (double)NPD_ST_F64(PRIM, npb_prim_t, neutron_flux, 0, 0) / 1e12;
    obs[1] = (double)NPD_ST_F64(PRIM, npb_prim_t, fuel_temperature, 0, 0) / 1000;
    obs[2] = (double)NPD_ST_F64(PRIM, npb_prim_t, coolant_temperature, 0, 0) / 300;
    obs[3] = (double)NPD_ST_F64(PRIM, npb_prim_t, coolant_pressure, 0, 0) / 20;
    obs[4] = (double)NPD_ST_F64(PRIM, npb_prim_t, coolant_flow_rate, 0, 0) / 50000;
    obs[5] = (double)NPD_ST_F64(PRIM, npb_prim_t, steam_temperature, 0, 0) / 300;
    obs[6] = (double)NPD_ST_F64(PRIM, npb_prim_t, steam_pressure, 0, 0) / 10;
    obs[8] = (double)NPD_ST_F64(PRIM, npb_prim_t, control_rod_position, 0, 0) / 100;
    obs[9] = (double)NPD_ST_F64(PRIM, npb_prim_t, steam_valve_position, 0, 0) / 100;
    obs[10] = load_demand / 100;                 /* load_demand IS state.power_level (sim.py:161) */
    obs[11] = (double)(scram_status != 0);
    obs[7] = sg_total_steam / 3000;
    obs[12] = electrical_power / 1100; obs[13] = thermal_efficiency / 0.35; obs[14] = sg_total_steam / 1665;
    obs[15] = load_demand / 100; obs[16] = 227.0 / 250; obs[17] = cooling_water_temperature / 35;
    obs[18] = fw_total_flow / 1665; obs[19] = fw_total_power / 40; obs[20] = (double)fw_available; obs[21] = fw_total_flow / 1665;
    if (scram_status) trip_flags |= NPB_TRIP_SCRAM;
    if (scram_fired) trip_flags |= NPB_TRIP_SCRAM_FIRED;
    if (nan_reset) trip_flags |= NPB_TRIP_NAN_RESET;
    if (live) {
      if (done_out) __builtin_nontemporal_store((uint8_t)scram_fired, &done_out[p]);
      if (trip_out) __builtin_nontemporal_store(trip_flags, &trip_out[p]);
    }
    if (obs_out) npd2_store_rows<NPB_OBS_DIM>(obs, obs_out, xch + X_OBS * NPB_WAVE, lane, block_base, (size_t)n_plants);
  } else {
    /* =========================================== wave B =========================================== */
    npd_u32x4 maint_cache23 = {0, 0, 0, 0};       /* {mask, until} of this wave's pumps 2, 3 (npd_maintenance.h) */
    if (maint) maint_cache23 = npd_maint_cache_fetch(MC, p, 1);
    const double tdt = dt / 60.0;
    if (lane == 0) { *(volatile int *)&xch[X_FLAG * NPB_WAVE] = 0; *(volatile int *)&xch[X_FLAG2 * NPB_WAVE] = 0; }
    /* ---- turbine lubrication pre-step: reads the previous step's rotor / bearing members, owns the lub_* ones */
    {
      npb_turb_t t;
      NPD_ST_LOAD(TURB, npb_turb_t, t, 0);
      const npb_turb_t t_old = t;
      npd_turbine_lube(&t, tdt);
      constexpr int L0 = NPB_F64_SLOT(npb_turb_t, lub_oil_temperature), L1 = NPB_F64_SLOT(npb_turb_t, thermal_expansion);
      static_assert(L1 - L0 == 13, "lub_* carried members are contiguous");
      const double *d = reinterpret_cast<const double *>(&t), *od = reinterpret_cast<const double *>(&t_old);
#pragma unroll
      for (int k = L0; k < L1; k++) {
        if ((NPD_ELIDE_TURB_F >> k) & 1) {
          if (__builtin_amdgcn_ballot_w64(npd_real_bits(d[k]) != npd_real_bits(od[k])) != 0) *NPD_RP(NPD_SEC_COL(TURB, 0) + k) = (npd_real_t)d[k];
        } else {
          *NPD_RP(NPD_SEC_COL(TURB, 0) + k) = (npd_real_t)d[k];
        }
      }
      *NPD_NP(float, NPD_SEC_COL(TURB, 0) + NPB_TURB_NCARRY + 3 / NPD_NPC, 3 % NPD_NPC) = (float)t.lub_effectiveness;
    }
    /* ---- chemistry sidecar: shared WaterChemistry + pH controller (secondary/__init__.py:634-665) */
    {
      npb_chem_t ch0; npb_ph_t ph;
      NPD_ST_LOAD(CHEM, npb_chem_t, ch0, 0);
      NPD_ST_LOAD(PH, npb_ph_t, ph, 0);
      const npb_chem_t ch0_old = ch0; const npb_ph_t ph_old = ph;
      npd_chemistry_sidecar(&ch0, &ph, dt);
      NPD_ST_STORE_ELIDE(CHEM, npb_chem_t, ch0, ch0_old, 0);
      NPD_ST_STORE_ELIDE(PH, npb_ph_t, ph, ph_old, 0);
    }
    int fw_mask = *NPD_NP(const int32_t, NPD_SEC_COL(FW, 0) + NPB_FW_NCARRY + (NPB_FW_NOUT + NPB_I32_SLOT(npb_fw_t, FW, running_mask)) / NPD_NPC,
                          (NPB_FW_NOUT + NPB_I32_SLOT(npb_fw_t, FW, running_mask)) % NPD_NPC);
    int n_prev_b = 0, may_run = 0;
#pragma unroll
    for (int i = 0; i < NPB_NUM_PUMPS; i++) {
      n_prev_b += (fw_mask >> i) & 1;
      const int stt = *NPD_NP(const int32_t, NPD_SEC_COL(PUMP, i) + NPB_PUMP_NCARRY + (NPB_PUMP_NOUT + NPB_I32_SLOT(npb_pump_t, PUMP, status)) / NPD_NPC,
                              (NPB_PUMP_NOUT + NPB_I32_SLOT(npb_pump_t, PUMP, status)) % NPD_NPC);
      may_run += (stt == NPD_PUMP_RUNNING || stt == NPD_PUMP_STARTING);
    }
    const bool serial_pumps = __builtin_amdgcn_ballot_w64(n_prev_b > 0 && may_run > n_prev_b) != 0;
    NPD2_SYNCJ(1);                                                                                     /* #1 */
    const int n_prev_running = (int)XR(X_NPREV);
    const double flow_per_pump = XR(X_FPP);
    npd_pump_sysconds_t sc;
    sc.feedwater_temperature = 40.0; sc.suction_pressure = 0.5; sc.discharge_pressure = 7.4; sc.max_sg_level = XR(X_MAXLVL);
    int running_count = 0;
    const double maint_time_b = maint ? XR(X_MAINT_TIME) : 0.0;
    if (maint) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); npd_maint_cache_landed(maint_cache23); }
    if (serial_pumps) { NPD2_SYNC_(); running_count = (int)XR(X_RUNCOUNT); }                         /* #1b */
#pragma unroll 1
    for (int i = 2; i < NPB_NUM_PUMPS; i++) {
      npb_pump_t pm;
      NPD_ST_LOAD(PUMP, npb_pump_t, pm, i);
      const npb_pump_t pm_old = pm;
      const uint32_t cooling_mask = i == 2 ? maint_cache23.x : maint_cache23.z;
      const float cooling_until = __uint_as_float(i == 2 ? maint_cache23.y : maint_cache23.w);
      /* parallel mode: the gate cannot close (serial_pumps is false for every lane), so its outcome needs no count */
      npd2_pump(&pm, serial_pumps ? (running_count < n_prev_running) : 1, n_prev_running, flow_per_pump, &sc, dt);
      running_count += pm.status == NPD_PUMP_RUNNING;
      npd2_publish_pump(xch, lane, i, pm);
      if (maint) {
        if (__builtin_amdgcn_ballot_w64(npd_maint_pump_hit(&pm, maint_tab, cooling_mask, cooling_until, maint_time_b)) != 0) maint_hit_bits |= 1u << i;
      }
      NPD_ST_STORE_ELIDE(PUMP, npb_pump_t, pm, pm_old, i);
    }
    NPD2_SYNCJ(2);                                                                                     /* #2 */
    const double load_demand_fraction = XR(X_LDF), actual_feedwater_temp = XR(X_FWTEMP);
    /* ---- steam generators 1 and 2, part 1 (heat transfer, fouling): needs no feedwater flow, so it runs while wave A
     * walks the four pumps' diagnostics / protection passes and the system level */
#pragma unroll 1
    for (int i = 1; i < NPB_NUM_SG; i++) {
      const double c_flow = XR(i == 1 ? X_CFLOW1 : X_CFLOW2), c_in = XR(i == 1 ? X_CIN1 : X_CIN2), c_out = XR(i == 1 ? X_COUT1 : X_COUT2);
      npb_sg_t g;
      NPD_ST_LOAD(SG, npb_sg_t, g, i);
      const npb_sg_t g_old = g;
      const double heat_transfer = npd_sg_part1(&g, &P, c_in, c_out, c_flow, dt * 60);
      XW(X_SGCARRY + 3 * (i - 1), heat_transfer); XW(X_SGCARRY + 3 * (i - 1) + 1, g.tsp_pressure_drop_ratio);
      XW(X_SGCARRY + 3 * (i - 1) + 2, g.secondary_temperature);
      NPD_ST_STORE_ELIDE(SG, npb_sg_t, g, g_old, i);
    }
    NPD2_FLAG2_WAIT(1);
    const double fw_total_flow = XR(X_FWFLOW);
    /* ---- part 2 (flow restrictions, secondary-side dynamics) */
    {
      const double actual_total_steam_flow = P.sg_design_total_steam_flow * load_demand_fraction;
      double total_primary_flow = 0.0;
      total_primary_flow += XR(X_CFLOW0); total_primary_flow += XR(X_CFLOW1); total_primary_flow += XR(X_CFLOW2);
#pragma unroll 1
      for (int i = 1; i < NPB_NUM_SG; i++) {
        const double c_flow = XR(i == 1 ? X_CFLOW1 : X_CFLOW2);
        double demand = (total_primary_flow > 0) ? actual_total_steam_flow * (c_flow / total_primary_flow) : actual_total_steam_flow / NPB_NUM_SG;
        npb_sg_t g;
        NPD_ST_LOAD(SG, npb_sg_t, g, i);             /* as part 1 left it (its output members rounded to float ...) */
        const npb_sg_t g_old = g;
        g.tsp_pressure_drop_ratio = XR(X_SGCARRY + 3 * (i - 1) + 1);   /* ... but for the one part 2 reads ... */
        g.secondary_temperature = XR(X_SGCARRY + 3 * (i - 1) + 2);     /* ... and the one the system average takes */
        const double level_old = (double)NPD_ST_F64(SEC, npb_sec_t, prev_sg_levels, 0, i);
        npd_sg_result_t r;
        r.heat_transfer_rate = 0.0; r.steam_flow_rate = 0.0; r.thermal_efficiency = 0.0;
        npd_sg_part2(&g, &P, XR(X_SGCARRY + 3 * (i - 1)), demand, fw_total_flow / NPB_NUM_SG, actual_feedwater_temp, dt * 60, &r);
        const int b = i == 1 ? X_SG1 : X_SG2;
        XW(b + 0, r.heat_transfer_rate); XW(b + 1, r.steam_flow_rate); XW(b + 2, g.secondary_pressure);
        XW(b + 3, g.secondary_temperature); XW(b + 4, g.steam_quality); XW(b + 5, r.thermal_efficiency > 0.1 ? 1.0 : 0.0);
        NPD_ST_STORE_ELIDE(SG, npb_sg_t, g, g_old, i);
        NPD_ST_F64_ELIDE(SEC, npb_sec_t, prev_sg_levels, 0, i, g.water_level, level_old);
        NPD_ST_F64(SEC, npb_sec_t, prev_sg_steam_flows, 0, i) = (npd_real_t)r.steam_flow_rate;
        NPD_ST_F64(SEC, npb_sec_t, prev_sg_qualities, 0, i) = (npd_real_t)g.steam_quality;
      }
    }
    NPD2_SYNCJ(5);                                                                                     /* #4 */
    /* the 70 stage-array columns go to registers while wave A runs stage pass A (they are updated stage by stage
     * behind wave A's chain, below) */
    npd2_tstg_old_t old;
#pragma unroll
    for (int k = 0; k < 14; k++) {
      old.eff_deg[k] = (double)NPD2_TSTG(stage_efficiency_degradation, k); old.deposit[k] = (double)NPD2_TSTG(stage_deposit_thickness, k);
      old.blade_wear[k] = (double)NPD2_TSTG(stage_blade_wear_factor, k); old.blade_t[k] = (double)NPD2_TSTG(blade_temperatures, k);
    }
#pragma unroll
    for (int k = 0; k < 8; k++) old.rotor_t[k] = (double)NPD2_TSTG(rotor_temperatures, k);
#pragma unroll
    for (int k = 0; k < 6; k++) old.casing_t[k] = (double)NPD2_TSTG(casing_temperatures, k);
    NPD2_SYNCJ(7);                                                                                     /* #5 */
    const double p_in0 = XR(X_PIN);
    const bool seq = __builtin_amdgcn_ballot_w64(isnan(p_in0)) != 0;
    if (!seq) {
      /* pass B, stages NPD2_SPLIT .. 13 and the five extraction pressures */
#pragma unroll
      for (int k = NPD2_SPLIT; k < 14; k++) {
        const double pk = XR(X_PSELF + k), pkm = XR(X_PSELF + k - 1);
        const double sat = npd_tsat_antoine(pk);
        XW(X_SAT + k - NPD2_SPLIT, sat); XW(X_HG + k - NPD2_SPLIT, npd_hg_from_tsat(sat)); XW(X_TRATIO + k - NPD2_SPLIT, npd_sqrt(npd_sqrt(pk / pkm)));
      }
#pragma unroll
      for (int e = 0; e < 5; e++) XW(X_HGEXT + e, npd_hg_from_tsat(npd_tsat_antoine(XR(X_PEXT + e))));
    }
    NPD2_SYNCJ(8);                                                                                     /* #6 */
    /* ---- per-stage degradation and metal temperatures (stage_system.py:294-339, enhanced_physics.py:73-166), each
     * stage as soon as wave A's chain has passed it */
    double max_stress = 0.0;
#pragma unroll
    for (int k = 0; k < 14; k++) {
      NPD2_FLAG_WAIT(k + 1);
      npd2_stage_post(st, old, k, XR(X_LOADING + k), XR(X_TOUT + k), tdt, &max_stress);
    }
    XW(X_MAXSTRESS, max_stress);
    npb_cond_t cd; npb_chem_t chc;
    NPD_ST_LOAD(COND, npb_cond_t, cd, 0);
    NPD_ST_LOAD(CHEM, npb_chem_t, chc, 1);
    const npb_cond_t cd_old = cd; const npb_chem_t chc_old = chc;
    NPD2_SYNCJ(9);                                                                                     /* #7 */
    const double effective_steam_flow = XR(X_EFFLOW), lp6_outlet_enthalpy = XR(X_LP6H), cooling_water_temperature = XR(X_CWT);
    /* ---- condenser (secondary/__init__.py:591-621) */
    double lp_exhaust_quality = 0.90;
    {
      double h_f = npd_cond_hf(0.007), h_g = npd_cond_hg(0.007);
      double h_fg = h_g - h_f;
      if (h_fg > 0) {
        lp_exhaust_quality = (lp6_outlet_enthalpy - h_f) / h_fg;
        lp_exhaust_quality = npd_pymax(0.0, npd_pymin(1.0, lp_exhaust_quality));
      }
    }
    npd_condenser_result_t cr;
    npd_condenser_update(&cd, &chc, 0.007, effective_steam_flow, lp_exhaust_quality, 45000.0, cooling_water_temperature, 1.2, 185.0, tdt, &cr);
    NPD_ST_STORE_ELIDE(COND, npb_cond_t, cd, cd_old, 0);
    NPD_ST_STORE_ELIDE(CHEM, npb_chem_t, chc, chc_old, 1);
    const double condenser_pressure = cr.condenser_pressure;
    NPD2_SYNCJ(11);                                                                                     /* #9 */
    /* ---- reward (sim.py:521-542) and info (sim.py:199-250) */
    const double base_reward = XR(X_TAIL + 0), electrical_power = XR(X_TAIL + 1), thermal_efficiency = XR(X_TAIL + 2), load_demand = XR(X_TAIL + 3);
    const double sg_avg_pressure = XR(X_TAIL + 4), sg_total_steam = XR(X_TAIL + 5), fw_total_flow_t = XR(X_TAIL + 6), heat_rejection = XR(X_TAIL + 7);
    double efficiency_reward = (thermal_efficiency - 0.30) * 10;
    double target_electrical_power = load_demand / 100.0 * 1100.0;
    double electrical_reward = -fabs(electrical_power - target_electrical_power) / 100;
    double steam_pressure_penalty = 0;
    if (sg_avg_pressure < 5.0 || sg_avg_pressure > 8.0) steam_pressure_penalty = -fabs(sg_avg_pressure - 6.895) * 5;
    double condenser_penalty = 0;
    if (condenser_pressure > 0.01) condenser_penalty = -(condenser_pressure - 0.007) * 100;
    double secondary_reward = efficiency_reward + electrical_reward + steam_pressure_penalty + condenser_penalty;
    double reward = base_reward + secondary_reward * 0.5;
    if (live && reward_out) __builtin_nontemporal_store(reward, &reward_out[p]);
    /* ---- secondary-level state write-back, feedback into the primary state (sim.py:429-498) */
    {
      const int avail = (int)XR(X_TAIL + 14);
      NPD_ST_F64_ELIDE(SEC, npb_sec_t, previous_feedwater_temp, 0, 0, actual_feedwater_temp, XR(X_TAIL + 15));
      NPD_ST_F64_ELIDE(SEC, npb_sec_t, cooling_water_temperature, 0, 0, cooling_water_temperature, XR(X_TAIL + 16));
      NPD_ST_F64(SEC, npb_sec_t, operating_hours, 0, 0) = (npd_real_t)(XR(X_TAIL + 17) + dt / 3600.0);
      npb_sec_t so;
      so.electrical_power_output = electrical_power; so.thermal_efficiency = thermal_efficiency;
      so.total_steam_flow = sg_total_steam; so.total_heat_transfer = XR(X_TAIL + 11); so.total_feedwater_flow = fw_total_flow_t;
      so.load_demand = load_demand; so.sg_avg_pressure = sg_avg_pressure; so.sg_avg_temperature = XR(X_TAIL + 12);
      so.sg_avg_quality = XR(X_TAIL + 13); so.has_previous_sg_conditions = 1; so.sg_system_availability = avail & 1;
      NPD_ST_STORE_NARROW(SEC, npb_sec_t, so, 0);
      double heat_removal_factor = sg_total_steam / 1665.0;
      if (!(avail & 2)) heat_removal_factor *= 0.5;
      NPD_ST_F64(PRIM, npb_prim_t, steam_flow_rate, 0, 0) = (npd_real_t)sg_total_steam;
      NPD_ST_F64(PRIM, npb_prim_t, last_heat_removal_factor, 0, 0) = (npd_real_t)heat_removal_factor;
    }
    if (info_out) {
      double info[NPB_INFO_DIM];
      info[NPB_INFO_THERMAL_POWER] = XR(X_TAIL + 8); info[NPB_INFO_REACTIVITY_PCM] = XR(X_TAIL + 9); info[NPB_INFO_TIME] = XR(X_TAIL + 10);
      info[NPB_INFO_ELECTRICAL_POWER] = isfinite(electrical_power) ? electrical_power : 0.0;
      info[NPB_INFO_THERMAL_EFFICIENCY] = npd_pymax(0.0, npd_pymin(isfinite(thermal_efficiency) ? thermal_efficiency : 0.0, 0.35));
      info[NPB_INFO_STEAM_FLOW] = isfinite(sg_total_steam) ? sg_total_steam : 1665.0;
      info[NPB_INFO_STEAM_PRESSURE] = isfinite(sg_avg_pressure) ? sg_avg_pressure : 6.895;
      info[NPB_INFO_CONDENSER_PRESSURE] = isfinite(condenser_pressure) ? condenser_pressure : 0.007;
      info[NPB_INFO_CONDENSER_HEAT_REJECTION] = isfinite(heat_rejection) ? heat_rejection : 0.0;
      info[NPB_INFO_FEEDWATER_FLOW] = fw_total_flow_t;
      info[NPB_INFO_SG_HEAT_TRANSFER] = XR(X_TAIL + 11); info[NPB_INFO_TURBINE_POWER] = XR(X_TAIL + 18);
      info[NPB_INFO_FEEDWATER_POWER] = XR(X_TAIL + 19); info[NPB_INFO_PRIMARY_THERMAL_POWER] = XR(X_TAIL + 20);
      info[NPB_INFO_TURBINE_EFFICIENCY] = XR(X_TAIL2 + 0); info[NPB_INFO_TURBINE_HP_POWER] = XR(X_TAIL2 + 1); info[NPB_INFO_TURBINE_LP_POWER] = XR(X_TAIL2 + 2);
      npd2_store_rows<NPB_INFO_DIM>(info, info_out, xch + X_INFO * NPB_WAVE, lane, block_base, (size_t)n_plants);
    }
  }
  NPD2_STAMP(31);
  /* ================= automatic maintenance (sim.py:208-223), for a group whose screen found something: rarely.  Wave B hands
   * its two pumps' verdict over (a word of the table's slot, past the table), both waves' state stores are in memory behind
   * the barrier, and wave A runs the rule for the 64 plants. */
  if (maint) {
    volatile unsigned *handover = (volatile unsigned *)(maint_tab + 48);
    if (wave == 1 && lane == 0) *handover = maint_hit_bits;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (wave == 0) {
      const unsigned bits = maint_hit_bits | (unsigned)__builtin_amdgcn_readfirstlane((int)*handover);
      if (bits | maint_due_with_orders) npd_maint_rule_for_wave<WHO>(maint_rc, MC, f64, N, p, bits, maint_due_with_orders);
    }
  }
#undef NPD_EXT_IDX
#undef NPD_IS_EXT
}

#define NPD2_KERNEL_ARGS \
    npb_params_t P, int n_plants, size_t N, npd_real_t *__restrict__ f64, const int32_t *__restrict__ action, \
    const double *__restrict__ magnitude, const double *__restrict__ setpoint, const double *__restrict__ noise_z, \
    const double *__restrict__ cw_temp, double *__restrict__ obs_out, double *__restrict__ reward_out, uint8_t *__restrict__ done_out, \
    uint32_t *__restrict__ trip_out, double *__restrict__ info_out, npd_maint_hot_t MH, const npd_maint_rule_consts_t *maint_rc, npd_maint_cache_t MC
#define NPD2_KERNEL_PASS P, n_plants, N, f64, action, magnitude, setpoint, noise_z, cw_temp, obs_out, reward_out, done_out, trip_out, info_out, MH, maint_rc, MC
/* two waves per SIMD (256 registers each, part of the state spilled): for batches between one and two waves per SIMD */
__global__ __launch_bounds__(NPD2_THREADS) NPD2_OCCUPANCY void npb_step2_kernel(NPD2_KERNEL_ARGS) { npd_step2_body<4, false>(NPD2_KERNEL_PASS); }
__global__ __launch_bounds__(NPD2_THREADS) NPD2_OCCUPANCY void npb_step2_maint_kernel(NPD2_KERNEL_ARGS) { npd_step2_body<4, true>(NPD2_KERNEL_PASS); }
/* one wave per SIMD and the whole register file: up to 32 768 plants (1 024 waves) nothing is gained by leaving room for a
 * second wave, and the spill code goes away */
__global__ __launch_bounds__(NPD2_THREADS) __attribute__((amdgpu_waves_per_eu(1, 1))) void npb_step2_wide_kernel(NPD2_KERNEL_ARGS) { npd_step2_body<5, false>(NPD2_KERNEL_PASS); }
__global__ __launch_bounds__(NPD2_THREADS) __attribute__((amdgpu_waves_per_eu(1, 1))) void npb_step2_wide_maint_kernel(NPD2_KERNEL_ARGS) { npd_step2_body<5, true>(NPD2_KERNEL_PASS); }

#endif
