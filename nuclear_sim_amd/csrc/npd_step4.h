/*
 * npd_step4.h -- the fused plant-step kernel, FOUR wavefronts per 64 plants (included by npb_kernels.hip behind npd_step2.h,
 * whose helpers it shares).
 *
 * Why: what bounds a step at and below 32 768 plants is not bandwidth but the length of one wave's instruction stream (a lone
 * wave of the one-wave kernel needs ~75 us however empty the chip is; the two-wave kernel's longer wave ~55 us, DESIGN.md
 * section 3).  The plant has more parallelism than two waves use: the four pumps are independent of each other once the level
 * control has handed out the demand, the three steam generators once the primary side is known, the twenty saturation states of
 * the turbine's pass B, the per-stage degradation / metal-temperature updates.  Here a group of 64 plants has four waves
 * (block = 256 threads, lane l of every wave is plant l), each <= 256 registers so that two groups share a CU's SIMDs at
 * 32 768 plants (2 048 waves, two per SIMD: one wave's scalar and memory instructions issue beside the other's vector ones):
 *
 *   segment        wave 0                    wave 1                    wave 2                    wave 3
 *   1              primary side, coupling    turbine lubrication       chemistry sidecar         secondary prelude, level control
 *   2              pump 0                    pump 1                    pump 2                    pump 3
 *   3              steam generator 0         steam generator 1         steam generator 2         pump tails, system level, load turbine
 *   4              stage arrays 0,3,6,..     stage arrays 1,4,7,..     stage arrays 2,5,8,..     SG sums, stage pass A
 *   5  (pass B)    stages 4..8               stages 9..13              the five extractions      inlet, stages 0..3
 *   6              stage post 0,3,6,..       stage post 1,4,7,..       stage post 2,5,8,..       stage chain, rotor
 *   7                                        condenser                                           protection, power gates
 *   8              observation, flags        reward, write-back        info
 *
 * Exactness: every device function is the one the other kernels call, sums over pumps / steam generators / stages are taken in
 * the reference's order by one wave from the values the others publish, and the one sequential dependence between pumps (the
 * demand gate of FeedwaterPumpSystem.update_system, npd_step2.h) falls back to running the pumps one after the other.
 */
#ifndef NPD_STEP4_H
#define NPD_STEP4_H

#define NPD4_THREADS 256
#define NPD4_SLOTS 96                         /* exchange slots of 64 doubles: 48 KB per group */
#ifdef NPB_STAMPS
#define NPD4_STAMP(k) do { if (lane == 0 && npb_stamp_buf) npb_stamp_buf[((size_t)blockIdx.x * 4 + wave) * 32 + (k)] = __builtin_readcyclecounter(); } while (0)
#define NPD4_SYNCJ(j) do { NPD4_STAMP(2 * (j) - 1); NPD2_SYNC_(); NPD4_STAMP(2 * (j)); } while (0)
#else
#define NPD4_STAMP(k)
#define NPD4_SYNCJ(j) NPD2_SYNC_()
#endif
/* three progress words in one slot: the stage chain's (wave 3 -> the stage-post waves), the feedwater flow's (wave 3 -> the
 * steam generators' part 2) and the primary side's (wave 0 -> wave 3, only when a plant has no previous SG conditions) */
#define NPD4_FLAGP(n) ((volatile int *)&xch[Y_FLAGS * NPB_WAVE + 2 * (n)])
#define NPD4_FLAG_SET(n, v) do { NPD_LDS_DRAIN(); *NPD4_FLAGP(n) = (v); } while (0)
#define NPD4_FLAG_WAIT(n, v) do { while (__builtin_amdgcn_readfirstlane(*NPD4_FLAGP(n)) < (v)) __builtin_amdgcn_s_sleep(1); } while (0)

enum {
  /* until the steam generators are done */
  Y_CFLOW = 0, Y_CIN = 3, Y_COUT = 6, Y_LDF = 9, Y_FWTEMP = 10, Y_NPREV = 11, Y_FPP = 12, Y_MAXLVL = 13, Y_RUNCOUNT = 15, Y_FWFLOW = 16,
  Y_PUMP = 17,                                /* 4 x X_PUMP_N */
  Y_SG = 17,                                  /* 3 x 6 results, over the pump region once wave 3 has read it (flag 1) */
  /* turbine */
  Y_PSELF = 0, Y_PEXT = 14, Y_PIN = 19,
  Y_SAT = 20, Y_HG = 30, Y_TRATIO = 40, Y_HGEXT = 50,   /* stages 4 .. 13 (wave 3 keeps its own: inlet, 0 .. 3) and the extractions */
  Y_TOUT = 0, Y_LOADING = 55, Y_STRESS = 69, Y_EFFLOW = 77, Y_LP6H = 78, Y_CWT = 79, Y_CONDP = 80,
  /* the whole step */
  Y_PRIM = 81,                                /* base reward, load demand, thermal power, reactivity, primary thermal power, scram bits */
  Y_TIME = 93, Y_FLAGS = 94, Y_MAINT_TAB = 95,
  /* tail */
  Y_TAIL = 0, Y_OBS = 24, Y_INFO = 47
};
static_assert(Y_PUMP + 4 * X_PUMP_N <= Y_PRIM && Y_SG + 18 <= Y_PUMP + 4 * X_PUMP_N && Y_HGEXT + 5 <= Y_LOADING && Y_LOADING + 14 <= Y_STRESS &&
              Y_STRESS + 8 <= Y_EFFLOW && Y_CONDP < Y_PRIM && Y_PRIM + 6 <= Y_TIME && Y_OBS + NPB_OBS_PAD <= Y_INFO && Y_INFO + NPB_OBS_PAD <= Y_PRIM &&
              Y_MAINT_TAB < NPD4_SLOTS && NPD_MH_N + 8 <= NPB_WAVE, "exchange slot plan");

/* the stage arrays of the stages k = R, R + 3, R + 6 ... (at most five) into registers / their post-pass behind the chain's flag.
 * The arrays are indexed by the stage's position j in the wave's list, so that the three waves that share this code path keep them
 * in the same registers (a barrier is a point where the compiler must assume any of them can be the wave running) */
struct npd4_old_t { double eff_deg[5], deposit[5], blade_wear[5], blade_t[5], rotor_t[5], casing_t[5]; };
template <int R>
__device__ __forceinline__ void npd4_stage_preload(const npd_stage_t &st, npd4_old_t &old) {
#pragma unroll
  for (int j = 0; j < 5; j++) {
    const int k = R + 3 * j;
    if (k >= 14) { old.eff_deg[j] = old.deposit[j] = old.blade_wear[j] = old.blade_t[j] = 0.0; }
    else {
      old.eff_deg[j] = (double)NPD2_TSTG(stage_efficiency_degradation, k < 14 ? k : 0); old.deposit[j] = (double)NPD2_TSTG(stage_deposit_thickness, k < 14 ? k : 0);
      old.blade_wear[j] = (double)NPD2_TSTG(stage_blade_wear_factor, k < 14 ? k : 0); old.blade_t[j] = (double)NPD2_TSTG(blade_temperatures, k < 14 ? k : 0);
    }
    old.rotor_t[j] = (k < 8) ? (double)NPD2_TSTG(rotor_temperatures, k < 8 ? k : 0) : 0.0;
    old.casing_t[j] = (k < 6) ? (double)NPD2_TSTG(casing_temperatures, k < 6 ? k : 0) : 0.0;
  }
}
template <int R>
__device__ __forceinline__ void npd4_stage_post(const npd_stage_t &st, const npd4_old_t &old, double *xch, int lane, double tdt) {
#pragma unroll
  for (int j = 0; j < 5; j++) {
    const int k = R + 3 * j;
    if (k >= 14) continue;
    NPD4_FLAG_WAIT(0, k + 1);
    double stress = 0.0;
    npd2_stage_post_vals(st, k, old.eff_deg[j], old.deposit[j], old.blade_wear[j], old.rotor_t[j], old.casing_t[j], old.blade_t[j],
                         XR(Y_LOADING + (k < 14 ? k : 0)), XR(Y_TOUT + (k < 14 ? k : 0)), tdt, &stress);
    if (k < 8) XW(Y_STRESS + (k < 8 ? k : 0), stress);
  }
}

/* segment 2, the same for every wave: pump `wave` (serial: one after the other with the real counts, four barriers) */
#define NPD4_PUMP_SEGMENT() \
  { \
    const int i = wave; \
    const int n_prev_running = (int)XR(Y_NPREV); \
    const double flow_per_pump = XR(Y_FPP); \
    npd_pump_sysconds_t sc; \
    sc.feedwater_temperature = 40.0; sc.suction_pressure = 0.5; sc.discharge_pressure = 7.4; sc.max_sg_level = XR(Y_MAXLVL); \
    const double maint_time = maint ? XR(Y_TIME) : 0.0; \
    if (maint) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); npd_maint_cache_landed(maint_cache); } \
    const uint32_t cooling_mask = (wave & 1) ? maint_cache.z : maint_cache.x; \
    const float cooling_until = __uint_as_float((wave & 1) ? maint_cache.w : maint_cache.y); \
    npb_pump_t pm; \
    NPD_ST_LOAD(PUMP, npb_pump_t, pm, i); \
    const npb_pump_t pm_old = pm; \
    if (!serial_pumps) { \
      npd2_pump(&pm, 1, n_prev_running, flow_per_pump, &sc, dt);        /* the gate cannot close: its outcome needs no count */ \
    } else { \
      _Pragma("unroll 1") \
      for (int turn = 0; turn < NPB_NUM_PUMPS; turn++) { \
        if (turn == i) { \
          const int running_count = (i == 0) ? 0 : (int)XR(Y_RUNCOUNT); \
          npd2_pump(&pm, running_count < n_prev_running, n_prev_running, flow_per_pump, &sc, dt); \
          XW(Y_RUNCOUNT, (double)(running_count + (pm.status == NPD_PUMP_RUNNING))); \
        } \
        NPD2_SYNC_(); \
      } \
    } \
    {   /* npd2_publish_pump, into this kernel's region */ \
      const int b = Y_PUMP + i * X_PUMP_N; \
      XW(b + 0, (double)((pm.status == NPD_PUMP_RUNNING) | (pm.trip_active ? 2 : 0))); \
      XW(b + 1, pm.flow_rate); XW(b + 2, pm.power_consumption); XW(b + 3, npd_pump_npsh_required(&pm)); XW(b + 4, pm.npsh_available); \
      XW(b + 5, pm.speed_percent); XW(b + 6, npd_pymax3(pm.wear_motor_bearings, pm.wear_pump_bearings, pm.wear_thrust_bearing)); \
      XW(b + 7, pm.wear_mechanical_seals); XW(b + 8, pm.vibration_level); XW(b + 9, pm.suction_pressure); XW(b + 10, pm.discharge_pressure); \
      XW(b + 11, pm.oil_temperature); XW(b + 12, pm.motor_temperature); \
    } \
    if (maint) {   /* anything new at this pump, for any plant of the group?  (npd_maintenance.h) */ \
      if (__builtin_amdgcn_ballot_w64(npd_maint_pump_hit(&pm, maint_tab, cooling_mask, cooling_until, maint_time)) != 0) maint_hit_bits |= 1u << i; \
    } \
    NPD_ST_STORE_ELIDE(PUMP, npb_pump_t, pm, pm_old, i); \
  }
#define NPD4_GATE_VERDICT() \
  { \
    int n_prev = 0, may_run = 0; \
    _Pragma("unroll") \
    for (int i = 0; i < NPB_NUM_PUMPS; i++) { \
      n_prev += (gate_fw_mask >> i) & 1; \
      may_run += (gate_status[i] == NPD_PUMP_RUNNING || gate_status[i] == NPD_PUMP_STARTING); \
    } \
    serial_pumps = __builtin_amdgcn_ballot_w64(n_prev > 0 && may_run > n_prev) != 0; \
  }

template <int WHO, bool MAINT>
__device__ __forceinline__ void npd_step4_body(
    const npb_params_t &P, int n_plants, size_t N, npd_real_t *__restrict__ f64,
    const int32_t *__restrict__ action, const double *__restrict__ magnitude, const double *__restrict__ setpoint,
    const double *__restrict__ noise_z, const double *__restrict__ cw_temp, double *__restrict__ obs_out,
    double *__restrict__ reward_out, uint8_t *__restrict__ done_out, uint32_t *__restrict__ trip_out,
    double *__restrict__ info_out, const npd_maint_hot_t &MH, const npd_maint_rule_consts_t *maint_rc, const npd_maint_cache_t &MC) {
  __shared__ __attribute__((aligned(16))) double xch[NPD4_SLOTS * NPB_WAVE];
  const int lane = threadIdx.x & (NPB_WAVE - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   /* wave-uniform role 0 .. 3 */
  const size_t block_base = (size_t)blockIdx.x * NPB_WAVE;
  const size_t p = block_base + lane;
  const bool live = p < (size_t)n_plants;
  const double dt = P.dt, tdt = dt / 60.0;
  const bool kinetics = P.heat_source == NPB_HEAT_REACTOR;
  npd_stage_t st;
  {
    st.lds = (char *)xch;
    st.f64b = (npd_gchar_t *)(f64 + block_base);
    st.nr = (uint32_t)(N * NPD_RB);
    st.laner = (uint32_t)lane * NPD_RB;
    st.grp16 = 0;
    st.diag = nullptr; st.diag_pitch = 0;
  }
  const bool maint = MAINT && P.maint_enabled && maint_rc != nullptr;
  unsigned maint_hit_bits = 0, maint_due_with_orders = 0;
  double *const maint_tab = xch + Y_MAINT_TAB * NPB_WAVE;
  NPD4_STAMP(0);
  /* this wave's pump's entry of the cooldown cache: {mask, until} of pumps 2h, 2h + 1 come as one 16-byte load */
  npd_u32x4 maint_cache = {0, 0, 0, 0};
  if (maint) maint_cache = npd_maint_cache_fetch(MC, p, wave >> 1);
  /* could the demand gate close for a later pump?  (every wave evaluates this on the same data, npd_step2.h; the loads go out
   * here, the verdict is formed at the end of segment 1, before any wave stores a pump or the feedwater section) */
  const int gate_fw_mask = *NPD_NP(const int32_t, NPD_SEC_COL(FW, 0) + NPB_FW_NCARRY + (NPB_FW_NOUT + NPB_I32_SLOT(npb_fw_t, FW, running_mask)) / NPD_NPC,
                                   (NPB_FW_NOUT + NPB_I32_SLOT(npb_fw_t, FW, running_mask)) % NPD_NPC);
  int gate_status[NPB_NUM_PUMPS];
#pragma unroll
  for (int i = 0; i < NPB_NUM_PUMPS; i++)
    gate_status[i] = *NPD_NP(const int32_t, NPD_SEC_COL(PUMP, i) + NPB_PUMP_NCARRY + (NPB_PUMP_NOUT + NPB_I32_SLOT(npb_pump_t, PUMP, status)) / NPD_NPC,
                             (NPB_PUMP_NOUT + NPB_I32_SLOT(npb_pump_t, PUMP, status)) % NPD_NPC);
  if (wave == 3 && lane == 0) { *NPD4_FLAGP(0) = 0; *NPD4_FLAGP(1) = 0; *NPD4_FLAGP(2) = 0; }
  NPD2_SYNC_();                                                                                          /* flags are down */

  bool serial_pumps;
  if (wave == 3) {
    /* ================================ wave 3: feedwater system level, then the turbine ================================ */
    npb_fw_t fw; npb_fw_t fw_old;
    double prev_levels[NPB_NUM_SG];
    double prev_feedwater_temp = 0.0, cw_old = 0.0, operating_hours = 0.0, cooling_water_temperature = 0.0, actual_feedwater_temp = 0.0;
    /* ---- secondary prelude (secondary/__init__.py:371-453), feedwater level control */
    double maint_entry = 0.0;
    if (maint && lane < NPD_MH_N) maint_entry = MH.tab[lane];
    cooling_water_temperature = (double)NPD_ST_F64(SEC, npb_sec_t, cooling_water_temperature, 0, 0);
    cw_old = cooling_water_temperature;
    prev_feedwater_temp = (double)NPD_ST_F64(SEC, npb_sec_t, previous_feedwater_temp, 0, 0);
    operating_hours = (double)NPD_ST_F64(SEC, npb_sec_t, operating_hours, 0, 0);
    const int has_prev = *NPD_NP(const int32_t, NPD_SEC_COL(SEC, 0) + NPB_SEC_NCARRY + (NPB_SEC_NOUT + NPB_I32_SLOT(npb_sec_t, SEC, has_previous_sg_conditions)) / NPD_NPC,
                                 (NPB_SEC_NOUT + NPB_I32_SLOT(npb_sec_t, SEC, has_previous_sg_conditions)) % NPD_NPC);
    double prev_flows[NPB_NUM_SG], prev_quals[NPB_NUM_SG];
#pragma unroll
    for (int i = 0; i < NPB_NUM_SG; i++) {
      prev_levels[i] = (double)NPD_ST_F64(SEC, npb_sec_t, prev_sg_levels, 0, i);
      prev_flows[i] = (double)NPD_ST_F64(SEC, npb_sec_t, prev_sg_steam_flows, 0, i);
      prev_quals[i] = (double)NPD_ST_F64(SEC, npb_sec_t, prev_sg_qualities, 0, i);
    }
    const double cw_in = (live && cw_temp) ? cw_temp[p] : NAN;
    if (!isnan(cw_in)) cooling_water_temperature = cw_in;
    actual_feedwater_temp = (0.1 * (40.0 + 187.0) + (1 - 0.1) * prev_feedwater_temp);
    NPD_ST_LOAD(FW, npb_fw_t, fw, 0);
    fw_old = fw;
    if (__builtin_amdgcn_ballot_w64(!has_prev) != 0) {   /* a plant's first step: its previous conditions come from the primary side's load */
      NPD4_FLAG_WAIT(2, 1);
      const double load_demand_fraction = XR(Y_LDF);
      if (!has_prev) {
#pragma unroll
        for (int i = 0; i < NPB_NUM_SG; i++) { prev_levels[i] = 12.5; prev_flows[i] = 555.0 * load_demand_fraction; prev_quals[i] = 0.99; }
      }
    }
    const double total_flow_demand = npd_fw_level_control(&fw, prev_levels, prev_flows, prev_quals, dt);
    int n_prev_running = 0;
#pragma unroll
    for (int i = 0; i < NPB_NUM_PUMPS; i++) n_prev_running += (fw.running_mask >> i) & 1;
    const double flow_per_pump = (n_prev_running > 0) ? total_flow_demand / n_prev_running : 0.0;
    XW(Y_FWTEMP, actual_feedwater_temp); XW(Y_NPREV, (double)n_prev_running); XW(Y_FPP, flow_per_pump);
    XW(Y_MAXLVL, npd_pymax3(prev_levels[0], prev_levels[1], prev_levels[2]));
    if (maint && lane < NPD_MH_N) maint_tab[lane] = maint_entry;
    NPD4_GATE_VERDICT();
    NPD4_SYNCJ(1);                                                                                        /* #1 */
    NPD4_PUMP_SEGMENT();
    NPD4_SYNCJ(2);                                                                                        /* #2 */
    npb_turb_t t; npb_turb_t t_old;
    double stage_eff[14];
    double fw_total_flow = 0.0, fw_total_power = 0.0;
    int fw_available = 0; uint32_t trip_flags = 0;
    /* ---- diagnostics + protection passes over the four pumps, system level (feedwater/physics.py:720-863) */
    npd_fw_acc_t acc;
    acc.total_flow = acc.total_power = acc.flow_sum = 0.0;
    acc.total_cavitation_risk = acc.total_wear_level = acc.total_vibration = 0.0;
    acc.running_count = acc.running_mask = acc.trips = 0; acc.trip_mask = 0;
#pragma unroll
    for (int i = 0; i < NPB_NUM_PUMPS; i++) npd2_pump_tail(xch + (Y_PUMP - X_PUMP) * NPB_WAVE, lane, i, &fw, &acc, dt);
    npd_fw_result_t fwr;
    npd_fw_finish(&fw, &acc, prev_levels, dt, &fwr);
    fw_total_flow = fwr.total_flow_rate; fw_total_power = fwr.total_power_consumption;
    fw_available = fwr.system_availability;
    trip_flags = (fwr.pump_trip_mask << 8) | (fw.system_trip_active ? NPB_TRIP_FW_SYSTEM : 0);
    /* the steam generators wait for this in their part 2; raising the flag also tells them that the pump region has been read */
    XW(Y_FWFLOW, fw_total_flow);
    NPD4_FLAG_SET(1, 1);
    NPD_ST_STORE_ELIDE(FW, npb_fw_t, fw, fw_old, 0);
    /* while the steam generators run: the turbine section and the 14 stages' efficiency products (TurbineStage state as the
     * previous step left it, stage_system.py:128-133, 294-339) */
    NPD_ST_LOAD(TURB, npb_turb_t, t, 0);          /* wave 1 owns the lub_* members; they are neither used nor stored here */
    t_old = t;
#pragma unroll
    for (int k = 0; k < 14; k++) {
      double fouling_factor = 1.0 / (1.0 + (double)NPD2_TSTG(stage_deposit_thickness, k) / 0.5);
      double blade_wear_factor = (double)NPD2_TSTG(stage_blade_wear_factor, k);
      double blade_condition_factor = npd_pymin(fouling_factor, blade_wear_factor);
      double actual_efficiency = npd_pymax(0.7, 0.88 - (double)NPD2_TSTG(stage_efficiency_degradation, k));
      stage_eff[k] = (actual_efficiency * blade_condition_factor * fouling_factor * blade_wear_factor * 1.0);
    }
    NPD4_SYNCJ(3);                                                                                        /* #3 */
    double sg_total_thermal = 0.0, sg_total_steam = 0.0, sg_avg_pressure = 0.0, sg_avg_temperature = 0.0, sg_avg_quality = 0.0;
    int sg_system_availability = 0;
    double pressure_stability_factor = 1.0, load_demand = 0.0;
    double p_self[14], flow_out[14], ext_flow[5];
    bool seq = false;
    double sg_ap = 0.0, sg_at = 0.0, sg_aq = 0.0, sg_pressures[NPB_NUM_SG];
    int sg_effective = 0;
#pragma unroll
    for (int i = 0; i < NPB_NUM_SG; i++) {
      const int b = Y_SG + 6 * i;
      sg_total_thermal += XR(b + 0); sg_total_steam += XR(b + 1);
      sg_ap += XR(b + 2); sg_at += XR(b + 3); sg_aq += XR(b + 4);
      sg_pressures[i] = XR(b + 2);
      if (XR(b + 5) != 0.0) sg_effective++;
    }
    sg_avg_pressure = sg_ap / NPB_NUM_SG; sg_avg_temperature = sg_at / NPB_NUM_SG; sg_avg_quality = sg_aq / NPB_NUM_SG;
    sg_system_availability = sg_effective >= (NPB_NUM_SG - 1);
    /* ---- turbine (dt in hours, load demand in PERCENT, secondary/__init__.py:564-569) */
    load_demand = XR(Y_PRIM + 1);
    t.load_demand = load_demand;
    pressure_stability_factor = npd_pressure_stability_factor(sg_pressures);
    double p_ext[5];
    const bool rare = npd2_stage_pass_a(sg_avg_pressure, sg_total_steam, load_demand, p_self, flow_out, p_ext, ext_flow);
    seq = __builtin_amdgcn_ballot_w64(rare) != 0;
    NPD_LDS_DRAIN();                                /* the steam generators' results have been read: their slots are written below */
#pragma unroll
    for (int k = 0; k < 14; k++) XW(Y_PSELF + k, p_self[k]);
#pragma unroll
    for (int e = 0; e < 5; e++) XW(Y_PEXT + e, p_ext[e]);
    XW(Y_PIN, seq ? NAN : sg_avg_pressure);         /* NaN tells the others that the group takes the sequential chain */
    XW(Y_CWT, cooling_water_temperature);
    NPD4_SYNCJ(4);                                                                                        /* #4 */
    /* pass B: twenty saturation states over the four waves; this one's are the inlet and stages 0 .. 3 */
    double sat_a[4], hg_a[4], tr_a[4], sat_in0 = 0.0, hg_in0 = 0.0;
    if (!seq) {
      sat_in0 = npd_tsat_antoine(sg_avg_pressure);
      hg_in0 = npd_hg_from_tsat(sat_in0);
#pragma unroll
      for (int k = 0; k < 4; k++) {
        sat_a[k] = npd_tsat_antoine(p_self[k]);
        hg_a[k] = npd_hg_from_tsat(sat_a[k]);
        tr_a[k] = npd_sqrt(npd_sqrt(p_self[k] / ((k == 0) ? sg_avg_pressure : p_self[k > 0 ? k - 1 : 0])));
      }
    }
    NPD4_SYNCJ(5);                                                                                        /* #5 */
    double stage_power_mw = 0.0, turbine_efficiency = 0.0, hp_power = 0.0, lp_power = 0.0, max_bearing_metal = 0.0, total_displacement = 0.0;
    npd2_chain_t ch;
    ch.T_in = sg_avg_temperature; ch.total_power = 0.0; ch.total_extraction = 0.0; ch.lp6_outlet_enthalpy = 0.0;
    ch.hp_power = 0.0; ch.lp_power = 0.0; ch.h_in0 = 0.0;
#define NPD_EXT_IDX(k) ((k) == 2 ? 0 : (k) == 3 ? 1 : (k) == 4 ? 2 : (k) == 8 ? 3 : 4)
#define NPD_IS_EXT(k) ((k) == 2 || (k) == 3 || (k) == 4 || (k) == 8 || (k) == 9)
    if (!seq) {
      ch.sat_in = sat_in0; ch.hg_in = hg_in0;
#pragma unroll
      for (int k = 0; k < 14; k++) {
        const double p_in = (k == 0) ? sg_avg_pressure : p_self[k > 0 ? k - 1 : 0];
        const double sat_k = k < 4 ? sat_a[k < 4 ? k : 0] : XR(Y_SAT + (k < 4 ? 0 : k - 4));
        const double hg_k = k < 4 ? hg_a[k < 4 ? k : 0] : XR(Y_HG + (k < 4 ? 0 : k - 4));
        const double tr_k = k < 4 ? tr_a[k < 4 ? k : 0] : XR(Y_TRATIO + (k < 4 ? 0 : k - 4));
        const double ef = NPD_IS_EXT(k) ? ext_flow[NPD_EXT_IDX(k)] : 0.0;
        const double hgx = NPD_IS_EXT(k) ? XR(Y_HGEXT + NPD_EXT_IDX(k)) : 0.0;
        double T_out, loading;
        npd2_chain_stage(k, ch, p_in, p_self[k], sat_k, hg_k, tr_k, flow_out[k], ef, hgx, stage_eff[k], &T_out, &loading);
        XW(Y_TOUT + k, T_out); XW(Y_LOADING + k, loading);
        NPD4_FLAG_SET(0, k + 1);
      }
      {   /* _steam_enthalpy at the last stage's outlet, whose saturation state pass B has */
        const double T_c = npd_pymax(0.0, npd_pymin(ch.T_in, 800.0));
        const double cp = (p_self[13] > 10.0) ? 2.5 : ((p_self[13] > 1.0) ? 2.2 : 2.0);
        const double h_out = (T_c <= ch.sat_in) ? ch.hg_in : ch.hg_in + cp * (T_c - ch.sat_in);
        if (sg_total_steam > 0) turbine_efficiency = (ch.h_in0 - h_out) / ch.h_in0;
      }
    } else {
      double cur_p = sg_avg_pressure, cur_T = sg_avg_temperature, cur_flow = sg_total_steam;
#pragma unroll
      for (int k = 0; k < 14; k++) {
        double T_out, loading;
        npd2_seq_stage(k, cur_p, cur_T, cur_flow, sg_total_steam, load_demand, stage_eff[k], ch, &T_out, &loading);
        XW(Y_TOUT + k, T_out); XW(Y_LOADING + k, loading);
        NPD4_FLAG_SET(0, k + 1);
      }
      if (sg_total_steam > 0) {
        const double h_in = npd_stage_steam_enthalpy(sg_avg_temperature, sg_avg_pressure);
        turbine_efficiency = (h_in - npd_stage_steam_enthalpy(cur_T, cur_p)) / h_in;
      }
    }
#undef NPD_EXT_IDX
#undef NPD_IS_EXT
    stage_power_mw = ch.total_power * pressure_stability_factor;
    hp_power = ch.hp_power; lp_power = ch.lp_power;
    XW(Y_EFFLOW, sg_total_steam - ch.total_extraction); XW(Y_LP6H, ch.lp6_outlet_enthalpy);
    npd_turbine_rotor(&t, stage_power_mw, sg_avg_temperature, load_demand, tdt, &max_bearing_metal, &total_displacement);
    NPD4_SYNCJ(6);                                                                                        /* #6: the stage arrays are done */
    double max_stress = 0.0;      /* MetalTemperatureTracker's max over the rotor points, in their order */
#pragma unroll
    for (int k = 0; k < 8; k++) max_stress = (k == 0) ? XR(Y_STRESS) : npd_pymax(max_stress, XR(Y_STRESS + k));
    npd_turbine_protect(&t, stage_power_mw, max_stress, max_bearing_metal, total_displacement, sg_system_availability, 0.007, tdt);
    {   /* store the turbine section but for wave 1's lub_* members */
      constexpr int L0 = NPB_F64_SLOT(npb_turb_t, lub_oil_temperature), L1 = NPB_F64_SLOT(npb_turb_t, thermal_expansion);
      const double *d = reinterpret_cast<const double *>(&t), *od = reinterpret_cast<const double *>(&t_old);
#pragma unroll
      for (int k = 0; k < NPB_TURB_NCARRY; k++) {
        if (k >= L0 && k < L1) continue;
        if ((NPD_ELIDE_TURB_F >> k) & 1) {
          if (__builtin_amdgcn_ballot_w64(npd_real_bits(d[k]) != npd_real_bits(od[k])) != 0) *NPD_RP(NPD_SEC_COL(TURB, 0) + k) = (npd_real_t)d[k];
        } else {
          *NPD_RP(NPD_SEC_COL(TURB, 0) + k) = (npd_real_t)d[k];
        }
      }
      static_assert(NPB_TURB_NOUT == 4 && NPB_TURB_NI32 == 2, "turbine narrow layout");
      constexpr int NC = NPB_TURB_NCARRY;
      *NPD_NP(float, NPD_SEC_COL(TURB, 0) + NC + 0 / NPD_NPC, 0 % NPD_NPC) = (float)t.thermal_expansion;
      *NPD_NP(float, NPD_SEC_COL(TURB, 0) + NC + 1 / NPD_NPC, 1 % NPD_NPC) = (float)t.total_power_output;
      *NPD_NP(float, NPD_SEC_COL(TURB, 0) + NC + 2 / NPD_NPC, 2 % NPD_NPC) = (float)t.vibration_displacement;
      *NPD_NP(int32_t, NPD_SEC_COL(TURB, 0) + NC + 4 / NPD_NPC, 4 % NPD_NPC) = t.trip_active;
      *NPD_NP(int32_t, NPD_SEC_COL(TURB, 0) + NC + 5 / NPD_NPC, 5 % NPD_NPC) = t.trip_latched_mask;
    }
    /* ---- electrical-power gates (secondary/__init__.py:750-932) */
    const double primary_thermal_power = XR(Y_PRIM + 4);
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
    /* the stage-post waves are past their reads of Y_TOUT (barrier #6): the tail goes there */
    XW(Y_TAIL + 0, electrical_power); XW(Y_TAIL + 1, thermal_efficiency); XW(Y_TAIL + 2, sg_avg_pressure); XW(Y_TAIL + 3, sg_total_steam);
    XW(Y_TAIL + 4, fw_total_flow); XW(Y_TAIL + 5, total_system_heat_rejection); XW(Y_TAIL + 6, sg_total_thermal); XW(Y_TAIL + 7, sg_avg_temperature);
    XW(Y_TAIL + 8, sg_avg_quality); XW(Y_TAIL + 9, (double)(sg_system_availability | (fw_available << 1))); XW(Y_TAIL + 10, prev_feedwater_temp);
    XW(Y_TAIL + 11, cw_old); XW(Y_TAIL + 12, operating_hours); XW(Y_TAIL + 13, t.total_power_output); XW(Y_TAIL + 14, fw_total_power);
    XW(Y_TAIL + 15, turbine_efficiency); XW(Y_TAIL + 16, hp_power); XW(Y_TAIL + 17, lp_power); XW(Y_TAIL + 18, (double)trip_flags);
    XW(Y_TAIL + 19, actual_feedwater_temp); XW(Y_TAIL + 20, (double)fw_available);
    NPD4_SYNCJ(7);                                                                                        /* #7 */
  } else {
    /* ================================ waves 0 .. 2 ================================ */
    int scram_bits = 0;                               /* wave 0: scram_status | scram_fired << 1 | nan_reset << 2 */
    if (wave == 0) {
      npd_maint_due_t maint_due = {};
      if (maint) npd_maint_due_load(&maint_due, f64, N, p);
      npd_inputs_t in;
      in.action = (live && action) ? action[p] : 8;
      in.magnitude = (live && magnitude) ? magnitude[p] : 1.0;
      in.power_setpoint = (live && setpoint) ? setpoint[p] : NAN;
      in.noise_z = (live && noise_z) ? noise_z[p] : 0.0;
      in.cooling_water_temp = NAN;      /* (wave 3 reads the cooling-water input) */
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
      int nan_reset;
      const int scram_fired = npd_primary_update(&s, &P, &in, &nan_reset, rho);
      npd_store_reactivity_components(P, rho, info_out, n_plants, p);
      npd_coupling_t c;
      npd_primary_to_secondary(&s, &c);
      double primary_thermal_power = 0.0;
  #pragma unroll
      for (int i = 0; i < NPB_NUM_SG; i++) {
        XW(Y_CFLOW + i, c.flow[i]); XW(Y_CIN + i, c.inlet_temp[i]); XW(Y_COUT + i, c.outlet_temp[i]);
        primary_thermal_power += c.thermal_power[i];
      }
      double load_demand_fraction = npd_pymin(1.0, primary_thermal_power / 3000.0);
      load_demand_fraction = npd_pymax(load_demand_fraction, 0.2);
      XW(Y_LDF, load_demand_fraction);
      s.sim_time += dt;
      const double power_reward = -fabs(s.power_level - 100) / 100;
      double temp_penalty = 0, pressure_penalty = 0;
      if (s.fuel_temperature > 800) temp_penalty = -(s.fuel_temperature - 800) / 100;
      if (s.coolant_pressure > 16) pressure_penalty = -(s.coolant_pressure - 16);
      const double scram_penalty = s.scram_status ? -100 : 0;
      scram_bits = (s.scram_status != 0) | (scram_fired ? 2 : 0) | (nan_reset ? 4 : 0);
      XW(Y_PRIM + 0, power_reward + temp_penalty + pressure_penalty + scram_penalty); XW(Y_PRIM + 1, s.power_level);
      XW(Y_PRIM + 2, s.thermal_power_mw); XW(Y_PRIM + 3, s.total_reactivity_pcm); XW(Y_PRIM + 4, primary_thermal_power);
      XW(Y_PRIM + 5, (double)scram_bits); XW(Y_TIME, s.sim_time);
      NPD4_FLAG_SET(2, 1);
      s.has_heat_removal_factor = 1;
      NPD_ST_STORE_ELIDE_PRIM(s, s_old);
      if (maint) {   /* sim.py:208-216 as far as no work order is involved; t = the clock after this step */
        const bool work = npd_maint_due_decide(&maint_due, s.sim_time, MH.tab[2 * NPB_MAINT_NPARAM + 1]);
        maint_due_with_orders = __builtin_amdgcn_ballot_w64(work) != 0 ? 1u : 0u;
      }
    } else if (wave == 1) {
      /* ---- turbine lubrication pre-step: reads the previous step's rotor / bearing members, owns the lub_* ones */
      npb_turb_t t;
      NPD_ST_LOAD(TURB, npb_turb_t, t, 0);
      const npb_turb_t t_old = t;
      npd_turbine_lube(&t, tdt);
      constexpr int L0 = NPB_F64_SLOT(npb_turb_t, lub_oil_temperature), L1 = NPB_F64_SLOT(npb_turb_t, thermal_expansion);
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
    } else {
      /* ---- chemistry sidecar: shared WaterChemistry + pH controller (secondary/__init__.py:634-665) */
      npb_chem_t ch0; npb_ph_t ph;
      NPD_ST_LOAD(CHEM, npb_chem_t, ch0, 0);
      NPD_ST_LOAD(PH, npb_ph_t, ph, 0);
      const npb_chem_t ch0_old = ch0; const npb_ph_t ph_old = ph;
      npd_chemistry_sidecar(&ch0, &ph, dt);
      NPD_ST_STORE_ELIDE(CHEM, npb_chem_t, ch0, ch0_old, 0);
      NPD_ST_STORE_ELIDE(PH, npb_ph_t, ph, ph_old, 0);
    }
    NPD4_GATE_VERDICT();
    NPD4_SYNCJ(1);                                                                                        /* #1 */
    NPD4_PUMP_SEGMENT();
    NPD4_SYNCJ(2);                                                                                        /* #2 */
    {
      /* ---- steam generator `wave` (enhanced_physics.py:433-547): part 1 needs no feedwater flow, part 2 waits for wave 3's */
      const int i = wave;
      const double c_flow = XR(Y_CFLOW + i), c_in = XR(Y_CIN + i), c_out = XR(Y_COUT + i);
      const double load_demand_fraction = XR(Y_LDF), feedwater_temp = XR(Y_FWTEMP);
      double total_primary_flow = 0.0;
      total_primary_flow += XR(Y_CFLOW + 0); total_primary_flow += XR(Y_CFLOW + 1); total_primary_flow += XR(Y_CFLOW + 2);
      const double actual_total_steam_flow = P.sg_design_total_steam_flow * load_demand_fraction;
      const double demand = (total_primary_flow > 0) ? actual_total_steam_flow * (c_flow / total_primary_flow) : actual_total_steam_flow / NPB_NUM_SG;
      npb_sg_t g;
      NPD_ST_LOAD(SG, npb_sg_t, g, i);
      const npb_sg_t g_old = g;
      const double level_old = (double)NPD_ST_F64(SEC, npb_sec_t, prev_sg_levels, 0, i);
      const double heat_transfer = npd_sg_part1(&g, &P, c_in, c_out, c_flow, dt * 60);
      NPD4_FLAG_WAIT(1, 1);
      const double fw_flow = XR(Y_FWFLOW);
      npd_sg_result_t r;
      r.heat_transfer_rate = 0.0; r.steam_flow_rate = 0.0; r.thermal_efficiency = 0.0;
      npd_sg_part2(&g, &P, heat_transfer, demand, fw_flow / NPB_NUM_SG, feedwater_temp, dt * 60, &r);
      const int b = Y_SG + 6 * i;
      XW(b + 0, r.heat_transfer_rate); XW(b + 1, r.steam_flow_rate); XW(b + 2, g.secondary_pressure);
      XW(b + 3, g.secondary_temperature); XW(b + 4, g.steam_quality); XW(b + 5, r.thermal_efficiency > 0.1 ? 1.0 : 0.0);
      NPD_ST_STORE_ELIDE(SG, npb_sg_t, g, g_old, i);
      NPD_ST_F64_ELIDE(SEC, npb_sec_t, prev_sg_levels, 0, i, g.water_level, level_old);
      NPD_ST_F64(SEC, npb_sec_t, prev_sg_steam_flows, 0, i) = (npd_real_t)r.steam_flow_rate;
      NPD_ST_F64(SEC, npb_sec_t, prev_sg_qualities, 0, i) = (npd_real_t)g.steam_quality;
    }
    NPD4_SYNCJ(3);                                                                                        /* #3 */
    npd4_old_t old;
    if (wave == 0) npd4_stage_preload<0>(st, old);
    else if (wave == 1) npd4_stage_preload<1>(st, old);
    else npd4_stage_preload<2>(st, old);
    NPD4_SYNCJ(4);                                                                                        /* #4 */
    {
      const double p_in0 = XR(Y_PIN);
      const bool seq = __builtin_amdgcn_ballot_w64(isnan(p_in0)) != 0;
      if (!seq) {
        if (wave == 2) {
  #pragma unroll
          for (int e = 0; e < 5; e++) XW(Y_HGEXT + e, npd_hg_from_tsat(npd_tsat_antoine(XR(Y_PEXT + e))));
        } else {
          const int k0 = (wave == 0) ? 4 : 9;           /* stages k0 .. k0 + 4 */
  #pragma unroll
          for (int j = 0; j < 5; j++) {
            const double pk = XR(Y_PSELF + k0 + j), pkm = XR(Y_PSELF + k0 + j - 1);
            const double sat = npd_tsat_antoine(pk);
            XW(Y_SAT + k0 + j - 4, sat); XW(Y_HG + k0 + j - 4, npd_hg_from_tsat(sat)); XW(Y_TRATIO + k0 + j - 4, npd_sqrt(npd_sqrt(pk / pkm)));
          }
        }
      }
    }
    NPD4_SYNCJ(5);                                                                                        /* #5 */
    npb_cond_t cd; npb_cond_t cd_old; npb_chem_t chc; npb_chem_t chc_old;   /* wave 1 */
    if (wave == 0) npd4_stage_post<0>(st, old, xch, lane, tdt);
    else if (wave == 1) {
      npd4_stage_post<1>(st, old, xch, lane, tdt);
      NPD_ST_LOAD(COND, npb_cond_t, cd, 0);
      NPD_ST_LOAD(CHEM, npb_chem_t, chc, 1);
      cd_old = cd; chc_old = chc;
    } else npd4_stage_post<2>(st, old, xch, lane, tdt);
    NPD4_SYNCJ(6);                                                                                        /* #6 */
    if (wave == 1) {
      /* ---- condenser (secondary/__init__.py:591-621) */
      const double effective_steam_flow = XR(Y_EFFLOW), lp6_outlet_enthalpy = XR(Y_LP6H), cwt = XR(Y_CWT);
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
      npd_condenser_update(&cd, &chc, 0.007, effective_steam_flow, lp_exhaust_quality, 45000.0, cwt, 1.2, 185.0, tdt, &cr);
      NPD_ST_STORE_ELIDE(COND, npb_cond_t, cd, cd_old, 0);
      NPD_ST_STORE_ELIDE(CHEM, npb_chem_t, chc, chc_old, 1);
      XW(Y_CONDP, cr.condenser_pressure);
    }
    NPD4_SYNCJ(7);                                                                                        /* #7 */
    if (wave == 0) {
      /* ---- observation, done, trip flags: the primary part (sim.py:290-333) from the primary section as stored in segment 1
       * (carried members: the stored value is the value; power_level, an output member, was published in fp64) */
      const double ld = XR(Y_PRIM + 1), sg_total_steam_t = XR(Y_TAIL + 3), fw_flow_t = XR(Y_TAIL + 4);
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
      obs[10] = ld / 100;                          /* load_demand IS state.power_level (sim.py:161) */
      obs[11] = (double)(scram_bits & 1);
      obs[7] = sg_total_steam_t / 3000;
      obs[12] = XR(Y_TAIL + 0) / 1100; obs[13] = XR(Y_TAIL + 1) / 0.35; obs[14] = sg_total_steam_t / 1665;
      obs[15] = ld / 100; obs[16] = 227.0 / 250; obs[17] = XR(Y_CWT) / 35;
      obs[18] = fw_flow_t / 1665; obs[19] = XR(Y_TAIL + 14) / 40; obs[20] = XR(Y_TAIL + 20); obs[21] = fw_flow_t / 1665;
      uint32_t flags = (uint32_t)XR(Y_TAIL + 18);
      if (scram_bits & 1) flags |= NPB_TRIP_SCRAM;
      if (scram_bits & 2) flags |= NPB_TRIP_SCRAM_FIRED;
      if (scram_bits & 4) flags |= NPB_TRIP_NAN_RESET;
      if (live) {
        if (done_out) __builtin_nontemporal_store((uint8_t)((scram_bits >> 1) & 1), &done_out[p]);
        if (trip_out) __builtin_nontemporal_store(flags, &trip_out[p]);
      }
      if (obs_out) npd2_store_rows<NPB_OBS_DIM>(obs, obs_out, xch + Y_OBS * NPB_WAVE, lane, block_base, (size_t)n_plants);
    } else if (wave == 1) {
      /* ---- reward (sim.py:521-542), secondary-level state write-back, feedback into the primary state (sim.py:429-498) */
      const double condenser_pressure = XR(Y_CONDP);
      const double base_reward = XR(Y_PRIM + 0), ld = XR(Y_PRIM + 1);
      const double electrical_power = XR(Y_TAIL + 0), thermal_efficiency = XR(Y_TAIL + 1), sg_avg_pressure_t = XR(Y_TAIL + 2), sg_total_steam_t = XR(Y_TAIL + 3);
      double efficiency_reward = (thermal_efficiency - 0.30) * 10;
      double target_electrical_power = ld / 100.0 * 1100.0;
      double electrical_reward = -fabs(electrical_power - target_electrical_power) / 100;
      double steam_pressure_penalty = 0;
      if (sg_avg_pressure_t < 5.0 || sg_avg_pressure_t > 8.0) steam_pressure_penalty = -fabs(sg_avg_pressure_t - 6.895) * 5;
      double condenser_penalty = 0;
      if (condenser_pressure > 0.01) condenser_penalty = -(condenser_pressure - 0.007) * 100;
      double secondary_reward = efficiency_reward + electrical_reward + steam_pressure_penalty + condenser_penalty;
      double reward = base_reward + secondary_reward * 0.5;
      if (live && reward_out) __builtin_nontemporal_store(reward, &reward_out[p]);
      const int avail = (int)XR(Y_TAIL + 9);
      NPD_ST_F64_ELIDE(SEC, npb_sec_t, previous_feedwater_temp, 0, 0, XR(Y_TAIL + 19), XR(Y_TAIL + 10));
      NPD_ST_F64_ELIDE(SEC, npb_sec_t, cooling_water_temperature, 0, 0, XR(Y_CWT), XR(Y_TAIL + 11));
      NPD_ST_F64(SEC, npb_sec_t, operating_hours, 0, 0) = (npd_real_t)(XR(Y_TAIL + 12) + dt / 3600.0);
      npb_sec_t so;
      so.electrical_power_output = electrical_power; so.thermal_efficiency = thermal_efficiency;
      so.total_steam_flow = sg_total_steam_t; so.total_heat_transfer = XR(Y_TAIL + 6); so.total_feedwater_flow = XR(Y_TAIL + 4);
      so.load_demand = ld; so.sg_avg_pressure = sg_avg_pressure_t; so.sg_avg_temperature = XR(Y_TAIL + 7);
      so.sg_avg_quality = XR(Y_TAIL + 8); so.has_previous_sg_conditions = 1; so.sg_system_availability = avail & 1;
      NPD_ST_STORE_NARROW(SEC, npb_sec_t, so, 0);
      double heat_removal_factor = sg_total_steam_t / 1665.0;
      if (!(avail & 2)) heat_removal_factor *= 0.5;
      NPD_ST_F64(PRIM, npb_prim_t, steam_flow_rate, 0, 0) = (npd_real_t)sg_total_steam_t;
      NPD_ST_F64(PRIM, npb_prim_t, last_heat_removal_factor, 0, 0) = (npd_real_t)heat_removal_factor;
    } else {
      if (info_out) {   /* info (sim.py:199-250) */
        const double condenser_pressure = XR(Y_CONDP), electrical_power = XR(Y_TAIL + 0), thermal_efficiency = XR(Y_TAIL + 1);
        const double sg_avg_pressure_t = XR(Y_TAIL + 2), sg_total_steam_t = XR(Y_TAIL + 3), heat_rejection = XR(Y_TAIL + 5);
        double info[NPB_INFO_DIM];
        info[NPB_INFO_THERMAL_POWER] = XR(Y_PRIM + 2); info[NPB_INFO_REACTIVITY_PCM] = XR(Y_PRIM + 3); info[NPB_INFO_TIME] = XR(Y_TIME);
        info[NPB_INFO_ELECTRICAL_POWER] = isfinite(electrical_power) ? electrical_power : 0.0;
        info[NPB_INFO_THERMAL_EFFICIENCY] = npd_pymax(0.0, npd_pymin(isfinite(thermal_efficiency) ? thermal_efficiency : 0.0, 0.35));
        info[NPB_INFO_STEAM_FLOW] = isfinite(sg_total_steam_t) ? sg_total_steam_t : 1665.0;
        info[NPB_INFO_STEAM_PRESSURE] = isfinite(sg_avg_pressure_t) ? sg_avg_pressure_t : 6.895;
        info[NPB_INFO_CONDENSER_PRESSURE] = isfinite(condenser_pressure) ? condenser_pressure : 0.007;
        info[NPB_INFO_CONDENSER_HEAT_REJECTION] = isfinite(heat_rejection) ? heat_rejection : 0.0;
        info[NPB_INFO_FEEDWATER_FLOW] = XR(Y_TAIL + 4);
        info[NPB_INFO_SG_HEAT_TRANSFER] = XR(Y_TAIL + 6); info[NPB_INFO_TURBINE_POWER] = XR(Y_TAIL + 13);
        info[NPB_INFO_FEEDWATER_POWER] = XR(Y_TAIL + 14); info[NPB_INFO_PRIMARY_THERMAL_POWER] = XR(Y_PRIM + 4);
        info[NPB_INFO_TURBINE_EFFICIENCY] = XR(Y_TAIL + 15); info[NPB_INFO_TURBINE_HP_POWER] = XR(Y_TAIL + 16); info[NPB_INFO_TURBINE_LP_POWER] = XR(Y_TAIL + 17);
        npd2_store_rows<NPB_INFO_DIM>(info, info_out, xch + Y_INFO * NPB_WAVE, lane, block_base, (size_t)n_plants);
      }
    }
  }
  NPD4_STAMP(31);
  /* ================= automatic maintenance (sim.py:208-223), for a group whose screen found something: rarely.  Each wave hands
   * its pump's verdict over (words of the table's slot, past the table), every wave's state stores are in memory behind the
   * barrier, and wave 0 runs the rule for the 64 plants. */
  if (maint) {
    volatile unsigned *handover = (volatile unsigned *)(maint_tab + 48);
    if (wave != 0 && lane == 0) handover[2 * wave] = maint_hit_bits;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (wave == 0) {
      unsigned bits = maint_hit_bits;
#pragma unroll
      for (int w = 1; w < 4; w++) bits |= (unsigned)__builtin_amdgcn_readfirstlane((int)handover[2 * w]);
      if (bits | maint_due_with_orders) npd_maint_rule_for_wave<WHO>(maint_rc, MC, f64, N, p, bits, maint_due_with_orders);
    }
  }
}

/* two waves per SIMD (<= 256 registers each): 2 048 waves = 32 768 plants resident at once */
__global__ __launch_bounds__(NPD4_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void npb_step4_kernel(NPD2_KERNEL_ARGS) { npd_step4_body<6, false>(NPD2_KERNEL_PASS); }
__global__ __launch_bounds__(NPD4_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void npb_step4_maint_kernel(NPD2_KERNEL_ARGS) { npd_step4_body<6, true>(NPD2_KERNEL_PASS); }

#endif
