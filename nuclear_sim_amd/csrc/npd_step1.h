/* The one-wave step kernel: one step of the 64 plants of this wave.  Included several times by npb_kernels.hip: as npb_step_kernel
 * (and npb_step_nt_kernel, the streaming-store build), each also with the automatic maintenance compiled in (NPD_STEP1_MAINT:
 * npb_step_maint_kernel, npb_step_nt_maint_kernel), and as
 * its diagnostics build npb_step_diag_kernel (NPD_STEP1_DIAG: two more arguments, and the NPB_DIAG_* columns are written as the
 * values come up -- in the plain build st.diag stays the NULL npd_stage_init left and every such store folds away). */
__global__ __launch_bounds__(NPB_WAVE) void NPD_STEP1_KERNEL(
    npb_params_t P, int n_plants, size_t N, npd_real_t *__restrict__ f64,
    const int32_t *__restrict__ action, const double *__restrict__ magnitude, const double *__restrict__ setpoint,
    const double *__restrict__ noise_z, const double *__restrict__ cw_temp, double *__restrict__ obs_out,
    double *__restrict__ reward_out, uint8_t *__restrict__ done_out, uint32_t *__restrict__ trip_out,
    double *__restrict__ info_out, npd_maint_hot_t MH, const npd_maint_rule_consts_t *maint_rc, npd_maint_cache_t MC
#ifdef NPD_STEP1_DIAG
    , double *diag, size_t diag_pitch
#endif
    ) {
  /* staging region of the LDS-DMA pipeline (npd_stage.h); the obs / info transposes at the very end reuse it */
  __shared__ __attribute__((aligned(16))) double lds[NPB_STAGE_BYTES / 8];
  static_assert(NPB_STAGE_SLOTS >= NPB_OBS_PAD, "the transposes alias the staging region");
  const size_t block_base = (size_t)blockIdx.x * NPB_WAVE;
  NPD_SEGMENT(f64, N, block_base);
  const size_t p = block_base + threadIdx.x; /* always < N (arena is padded to a multiple of 64) */
  const bool live = p < (size_t)n_plants;
  const double dt = P.dt;
  const bool full = P.mode == NPB_MODE_FULL;
  npd_stage_t st;
  npd_stage_init(st, lds, f64, N, block_base);
#ifdef NPD_STEP1_DIAG
  st.diag = diag + p; st.diag_pitch = diag_pitch;
#endif

  /* per-step inputs first (plain loads), then the first staged group: primary + secondary-level scalars */
  npd_inputs_t in;
  in.action = (live && action) ? action[p] : 8;
  in.magnitude = (live && magnitude) ? magnitude[p] : 1.0;
  in.power_setpoint = (live && setpoint) ? setpoint[p] : NAN;
  in.noise_z = (live && noise_z) ? noise_z[p] : 0.0;
  in.cooling_water_temp = (live && cw_temp) ? cw_temp[p] : NAN;
  /* automatic maintenance on (npd_maintenance.h, "the threshold screen inside the step kernels"): lane l fetches entry l of the
   * folded threshold table from the kernel-argument segment; it goes to the last 512 B of the staging region -- no staged
   * section reaches that far -- once the first staged group has landed, and the pump phase reads it from there */
  /* NPD_STEP1_MAINT: this build carries the maintenance path (npb_step launches it when params.maint_enabled); the plain builds
   * fold all of it away -- compiled in, even switched off it cost the step 1-3 % (a call makes the kernel a non-leaf: stack
   * set-up, fewer scalar registers; profiles/r3_ab_r2_vs_maintenance_capable_kernels.txt) */
  const bool maint = NPD_STEP1_MAINT && P.maint_enabled && maint_rc != nullptr && full;
  unsigned maint_hit_bits = 0, maint_due_with_orders = 0;     /* wave-uniform: what the screen found (npd_maintenance.h) */
  double *const maint_tab = lds + (NPB_STAGE_BYTES - 512) / 8;
  static_assert(NPD_MH_N <= 64 && (size_t)NPD_SLOTS(TSTG) * NPD_SLOTB <= NPB_STAGE_BYTES - 512, "room for the threshold table behind the largest staged group");
  double maint_entry = 0.0, maint_time = 0.0;
  if (maint && threadIdx.x < NPD_MH_N) maint_entry = MH.tab[threadIdx.x];
  npd_maint_due_t maint_due = {};
  npd_u32x4 maint_cache01 = {0, 0, 0, 0}, maint_cache23 = {0, 0, 0, 0};      /* {mask, until} of pumps 0,1 | 2,3 (npd_maintenance.h) */
  if (maint) {
    npd_maint_due_load(&maint_due, f64, N, p);
    maint_cache01 = npd_maint_cache_fetch(MC, p, 0); maint_cache23 = npd_maint_cache_fetch(MC, p, 1);
  }
  /* under ConstantHeatSource the point-kinetics columns of the primary section stay where they are: neither staged
   * nor stored (their register copies are then never used either) */
  const bool kinetics = P.heat_source == NPB_HEAT_REACTOR;
  if (kinetics) {
    NPD_DMA(PRIM, 0, NPD_LS_PRIM);
  } else {
    npd_dma<NPD_PRIM_KIN0>(st, NPD_SEC_COL(PRIM, 0), NPD_LS_PRIM);
    npd_dma<NPD_NCOL(PRIM) - NPB_PRIM_NCARRY>(st, NPD_SEC_COL(PRIM, 0) + NPB_PRIM_NCARRY, NPD_LS_PRIM + NPB_PRIM_NCARRY);
  }
  NPD_DMA(SEC, 0, NPD_LS_SEC);

  double obs[NPB_OBS_DIM];
  double info[NPB_INFO_DIM];
  double base_reward, load_demand, cooling_water_temperature;
  int scram_fired, nan_reset, scram_status;
  npd_coupling_t c;

  NPD_STAMP(0);
  NPD_WAIT_ACC_INIT();
  npb_prim_t s;
  double prev_feedwater_temp, operating_hours;
  double prev_levels[NPB_NUM_SG], prev_flows[NPB_NUM_SG], prev_quals[NPB_NUM_SG];
  int has_prev;
  NPD_DMA_WAIT();
  NPD_CONSUME(PRIM, npb_prim_t, s, NPD_LS_PRIM);
  const npb_prim_t s_old = s; /* for the unchanged-column elision; only the members named in its mask stay live */
  cooling_water_temperature = NPD_STAGED_F64(SEC, npb_sec_t, cooling_water_temperature, 0, NPD_LS_SEC);
  prev_feedwater_temp = NPD_STAGED_F64(SEC, npb_sec_t, previous_feedwater_temp, 0, NPD_LS_SEC);
  operating_hours = NPD_STAGED_F64(SEC, npb_sec_t, operating_hours, 0, NPD_LS_SEC);
  has_prev = NPD_STAGED_I32(SEC, npb_sec_t, has_previous_sg_conditions, NPD_LS_SEC);
  /* staged values of the secondary-level columns that are usually rewritten unchanged (store elision) */
  const double cw_old = cooling_water_temperature;
  const double elec_old = NPD_STAGED_F64(SEC, npb_sec_t, electrical_power_output, 0, NPD_LS_SEC);
  const double eff_old = NPD_STAGED_F64(SEC, npb_sec_t, thermal_efficiency, 0, NPD_LS_SEC);
  const int sgavail_old = NPD_STAGED_I32(SEC, npb_sec_t, sg_system_availability, NPD_LS_SEC);
  double pl_old[NPB_NUM_SG];
#pragma unroll
  for (int i = 0; i < NPB_NUM_SG; i++) {
    prev_levels[i] = NPD_STAGED_F64(SEC, npb_sec_t, prev_sg_levels, i, NPD_LS_SEC);
    pl_old[i] = prev_levels[i];
    prev_flows[i] = NPD_STAGED_F64(SEC, npb_sec_t, prev_sg_steam_flows, i, NPD_LS_SEC);
    prev_quals[i] = NPD_STAGED_F64(SEC, npb_sec_t, prev_sg_qualities, i, NPD_LS_SEC);
  }
  if (maint && threadIdx.x < NPD_MH_N) maint_tab[threadIdx.x] = maint_entry;
  if (maint) { npd_maint_cache_landed(maint_cache01); npd_maint_cache_landed(maint_cache23); }    /* behind the NPD_DMA_WAIT above */
  NPD_LDS_DRAIN();
  if (full) { NPD_DMA(FW, 0, NPD_LS_FW); NPD_DMA(PUMP, 0, NPD_LS_PUMP0); }
  else NPD_DMA(SG, 0, 0);

  /* ================= phase 0: primary side + coupling (sim.py:141-161) ================= */
  {
    if (P.heat_source != NPB_HEAT_EXTERNAL && !isnan(in.power_setpoint)) s.hs_setpoint_percent = npd_clip(in.power_setpoint, 0.0, 150.0);
    double rho[NPB_INFO_NRHO];
    scram_fired = npd_primary_update(&s, &P, &in, &nan_reset, rho);
    npd_store_reactivity_components(P, rho, info_out, n_plants, p);
    npd_primary_to_secondary(&s, &c);
    s.sim_time += dt;
    load_demand = s.power_level; /* sim.py:161: the caller's load_demand is overwritten */
    if (maint) {   /* sim.py:208-216 as far as no work order is involved; t = the clock after this step */
      maint_time = s.sim_time;
      const bool work = npd_maint_due_decide(&maint_due, s.sim_time, maint_tab[2 * NPB_MAINT_NPARAM + 1]);
      maint_due_with_orders = __builtin_amdgcn_ballot_w64(work) != 0 ? 1u : 0u;
    }
    scram_status = s.scram_status;
    npd_obs_primary(s, obs); /* obs[7] is patched after the secondary side has produced the steam flow */
    /* base reward, sim.py:503-519 */
    double power_reward = -fabs(s.power_level - 100) / 100;
    double temp_penalty = 0, pressure_penalty = 0;
    if (s.fuel_temperature > 800) temp_penalty = -(s.fuel_temperature - 800) / 100;
    if (s.coolant_pressure > 16) pressure_penalty = -(s.coolant_pressure - 16);
    double scram_penalty = s.scram_status ? -100 : 0;
    base_reward = power_reward + temp_penalty + pressure_penalty + scram_penalty;
    info[NPB_INFO_THERMAL_POWER] = s.thermal_power_mw;
    info[NPB_INFO_REACTIVITY_PCM] = s.total_reactivity_pcm;
    info[NPB_INFO_TIME] = s.sim_time;
  }

  /* ================= secondary prelude (secondary/__init__.py:371-453) ================= */
  if (!isnan(in.cooling_water_temp)) cooling_water_temperature = in.cooling_water_temp; /* sim.py:138-139 */
  double actual_feedwater_temp;
  {
    double estimated_feedwater_temp = 40.0 + 187.0;
    double alpha = 0.1;
    actual_feedwater_temp = (alpha * estimated_feedwater_temp + (1 - alpha) * prev_feedwater_temp);
  }
  double primary_thermal_power = 0.0;
#pragma unroll
  for (int i = 0; i < NPB_NUM_SG; i++) primary_thermal_power += c.thermal_power[i];
  double load_demand_fraction = npd_pymin(1.0, primary_thermal_power / 3000.0);
  load_demand_fraction = npd_pymax(load_demand_fraction, 0.2);
  if (!has_prev) { /* :447-453 hard-coded first-step values, not the SG initial conditions */
#pragma unroll
    for (int i = 0; i < NPB_NUM_SG; i++) { prev_levels[i] = 12.5; prev_flows[i] = 555.0 * load_demand_fraction; prev_quals[i] = 0.99; }
  }

#ifdef NPD_STEP1_DIAG
  NPD_DIAG(st, NPB_DIAG_FW_AVG_SG_LEVEL, (0.0 + prev_levels[0] + prev_levels[1] + prev_levels[2]) / 3);
  NPD_DIAG(st, NPB_DIAG_FW_TOTAL_STEAM_FLOW, 0.0 + prev_flows[0] + prev_flows[1] + prev_flows[2]);
  NPD_DIAG(st, NPB_DIAG_FW_AVG_STEAM_QUALITY, (0.0 + prev_quals[0] + prev_quals[1] + prev_quals[2]) / 3);
  double diag_prev_pressure_sum = 0.0, diag_perf_sum = 0.0, diag_health = 1.0;
  int diag_perf_n = 0, diag_alarms = 0;
#endif
  double fw_total_flow = 0.0, fw_total_power = 0.0;
  int fw_available = 1;
  uint32_t trip_flags = 0;
  npb_sg_t g, g_old;

  NPD_STAMP(1);
  if (full) {
    /* ================= phase 1: feedwater system (physics.py:662-863) ================= */
    npb_fw_t fw;
    npb_pump_t pm;
    /* boundary: fw + pump 0 are staged; store the primary section, stage pump 1 */
    NPD_DMA_WAIT();
    s.has_heat_removal_factor = 1; /* consumed by phase 0; the feedback below always leaves a factor behind (sim.py:495) */
    NPD_ST_STORE_ELIDE_PRIM(s, s_old);
    NPD_CONSUME(FW, npb_fw_t, fw, NPD_LS_FW);
    NPD_CONSUME(PUMP, npb_pump_t, pm, NPD_LS_PUMP0);
    const npb_fw_t fw_old = fw;
    npb_pump_t pm_old = pm;
    NPD_LDS_DRAIN();
    NPD_DMA(PUMP, 1, 0);
    double total_flow_demand = npd_fw_level_control(&fw, prev_levels, prev_flows, prev_quals, dt);
    int n_prev_running = 0;
#pragma unroll
    for (int i = 0; i < NPB_NUM_PUMPS; i++) n_prev_running += (fw.running_mask >> i) & 1;
    double flow_per_pump = (n_prev_running > 0) ? total_flow_demand / n_prev_running : 0.0;
    npd_pump_sysconds_t sc;
    sc.feedwater_temperature = 40.0; sc.suction_pressure = 0.5; sc.discharge_pressure = 7.4; /* :457-464 */
    sc.max_sg_level = npd_pymax3(prev_levels[0], prev_levels[1], prev_levels[2]);
    npd_fw_acc_t acc;
    acc.total_flow = acc.total_power = acc.flow_sum = 0.0;
    acc.total_cavitation_risk = acc.total_wear_level = acc.total_vibration = 0.0;
    acc.running_count = acc.running_mask = acc.trips = 0; acc.trip_mask = 0; acc.trip_kinds = 0;
#pragma unroll 1
    for (int i = 0; i < NPB_NUM_PUMPS; i++) {
      NPD_STAMP(2 + i);
      /* this (plant, pump)'s entry of the cooldown cache (rolled loop: picked by selects, as the per-SG values are) */
      const uint32_t cooling_mask = i == 0 ? maint_cache01.x : (i == 1 ? maint_cache01.z : (i == 2 ? maint_cache23.x : maint_cache23.z));
      const float cooling_until = __uint_as_float(i == 0 ? maint_cache01.y : (i == 1 ? maint_cache01.w : (i == 2 ? maint_cache23.y : maint_cache23.w)));
      npd_fw_pump_step(&pm, &fw, &acc, i, n_prev_running, flow_per_pump, &sc, dt);
#ifdef NPD_STEP1_DIAG
      {   /* FeedwaterPumpLubricationSystem.system_health_factor (lubrication_base.py:380-399, component constants pump_lubrication.py:110-195) */
        const double wpf[6] = {0.025, 0.015, 0.02, 0.025, 0.03, 0.01}, lpf[6] = {0.0, 0.05, 0.1, 0.15, 0.2, 0.05};
        const double wear[6] = {pm.wear_impeller, pm.wear_motor_bearings, pm.wear_pump_bearings, pm.wear_thrust_bearing, pm.wear_mechanical_seals, pm.wear_coupling_system};
        double sum = 0.0;
#pragma unroll
        for (int q = 0; q < 6; q++) sum += npd_pymax(0.1, 1.0 - (wear[q] * wpf[q] + (1.0 - pm.lubrication_effectiveness) * lpf[q]));
        NPD_DIAG(st, NPB_DIAG_PUMP_HEALTH_FACTOR + i, sum / 6 * pm.lubrication_effectiveness);
        if (pm.status == NPD_PUMP_RUNNING) {   /* pump_system.py:1304-1316 */
          diag_perf_sum += npd_pymax(0.5, 1.0 - pm.flow_degradation / 100.0) * npd_pymax(0.5, 1.0 - pm.efficiency_degradation / 100.0);
          diag_perf_n += 1;
        }
        diag_alarms += npd_fw_pump_alarms(&pm);
        NPD_DIAG(st, NPB_DIAG_PUMP_MAINTENANCE_OCCURRED + i, 0.0); NPD_DIAG(st, NPB_DIAG_PUMP_OIL_TOP_OFF_OCCURRED + i, 0.0);   /* the rule sets them */
        NPD_DIAG(st, NPB_DIAG_PUMP_MAINTENANCE_ACTION + i, 0.0);
      }
#endif
      if (maint) {   /* anything new at this pump -- a threshold crossed outside its cooldown, a cooldown run out -- for any plant of the wave?  (npd_maintenance.h) */
        if (__builtin_amdgcn_ballot_w64(npd_maint_pump_hit(&pm, maint_tab, cooling_mask, cooling_until, maint_time)) != 0) maint_hit_bits |= 1u << i;
      }
      /* boundary: pump i -> HBM, pump i+1 (staged during this pump's arithmetic) -> the same registers,
       * then stage pump i+2, or SG 0 once the last pump is on its way */
      NPD_DMA_WAIT();
      NPD_ST_STORE_ELIDE(PUMP, npb_pump_t, pm, pm_old, i);
      if (i + 1 < NPB_NUM_PUMPS) { NPD_CONSUME(PUMP, npb_pump_t, pm, 0); pm_old = pm; }
      NPD_LDS_DRAIN();
      if (i + 2 < NPB_NUM_PUMPS) NPD_DMA(PUMP, i + 2, 0);
      else if (i + 2 == NPB_NUM_PUMPS) NPD_DMA(SG, 0, 0);
    }
    NPD_STAMP(6);
    npd_fw_result_t fwr;
#ifdef NPD_STEP1_DIAG
    const int diag_trip_before = fw.system_trip_active;
#endif
    npd_fw_finish(&fw, &acc, prev_levels, dt, &fwr);
#ifdef NPD_STEP1_DIAG
    /* FeedwaterProtectionSystem's bookkeeping beside system_trip_active (protection_system.py:447-476, 680-716): the number of
     * trips standing, and -- on the step a trip comes up -- the count of such steps and the two emergency actions that stay set
     * until someone resets the protection system: carried in the caller's buffer from the step diagnostics were switched on */
    NPD_DIAG(st, NPB_DIAG_FW_ACTIVE_TRIPS, (double)fwr.active_trips);
    if (st.diag && fw.system_trip_active && !diag_trip_before) {
      st.diag[(size_t)NPB_DIAG_FW_VALID_TRIP_COUNT * st.diag_pitch] += 1.0;
      if (fwr.trip_kinds & 1) st.diag[(size_t)NPB_DIAG_FW_EMERGENCY_FEEDWATER * st.diag_pitch] = 1.0;
      if (fwr.trip_kinds & 2) st.diag[(size_t)NPB_DIAG_FW_STEAM_DUMP * st.diag_pitch] = 1.0;
    }
    diag_health = fw.overall_health_score;
    NPD_DIAG(st, NPB_DIAG_FW_ACTIVE_ALARMS, (double)(diag_alarms + npd_fw_system_alarms(&fw, &acc, prev_levels)));
#endif
    fw_total_flow = fwr.total_flow_rate; fw_total_power = fwr.total_power_consumption;
    fw_available = fwr.system_availability;
    trip_flags |= (fwr.pump_trip_mask << 8) | (fw.system_trip_active ? NPB_TRIP_FW_SYSTEM : 0);
    /* boundary: SG 0 -> registers (its DMA ran during pump 3), fw -> HBM, stage SG 1 */
    NPD_DMA_WAIT();
    NPD_ST_STORE_ELIDE(FW, npb_fw_t, fw, fw_old, 0);
    NPD_CONSUME(SG, npb_sg_t, g, 0);
    g_old = g;
    NPD_LDS_DRAIN();
    NPD_DMA(SG, 1, 0);
  } else {
    /* config-2 mode: no feedwater system; boundary straight to SG 0 */
    NPD_DMA_WAIT();
    s.has_heat_removal_factor = 1; /* consumed by phase 0; the feedback below always leaves a factor behind (sim.py:495) */
    NPD_ST_STORE_ELIDE_PRIM(s, s_old);
    NPD_CONSUME(SG, npb_sg_t, g, 0);
    g_old = g;
    NPD_LDS_DRAIN();
    NPD_DMA(SG, 1, 0);
  }

  NPD_STAMP(7);
  /* ================= phase 2: steam generators (enhanced_physics.py:433-547) ================= */
  double sg_total_thermal = 0.0, sg_total_steam = 0.0, sg_ap = 0.0, sg_at = 0.0, sg_aq = 0.0;
  double sg_pressures[NPB_NUM_SG];
  int sg_effective = 0;
  {
    double actual_total_steam_flow = P.sg_design_total_steam_flow * load_demand_fraction;
    double total_primary_flow = 0.0;
#pragma unroll
    for (int i = 0; i < NPB_NUM_SG; i++) total_primary_flow += c.flow[i];
#pragma unroll 1
    for (int i = 0; i < NPB_NUM_SG; i++) {
      /* the loop is rolled (code size), so per-SG values are picked by selects: a dynamically indexed
       * private array would live in scratch, and every scratch load drains vmcnt, i.e. waits for the
       * whole staged prefetch and the previous stores */
      const double c_flow = npd_sel3(i, c.flow[0], c.flow[1], c.flow[2]);
      const double c_inlet = npd_sel3(i, c.inlet_temp[0], c.inlet_temp[1], c.inlet_temp[2]);
      const double c_outlet = npd_sel3(i, c.outlet_temp[0], c.outlet_temp[1], c.outlet_temp[2]);
      double demand = (total_primary_flow > 0) ? actual_total_steam_flow * (c_flow / total_primary_flow)
                                               : actual_total_steam_flow / NPB_NUM_SG;
      /* full mode: equal split of the actual feedwater flow (:500-506 key mismatch); config-2 mode:
       * "perfect mass balance" fallback (enhanced_physics.py:495-497) */
      double fwflow = full ? fw_total_flow / NPB_NUM_SG : demand;
      NPD_STAMP(8 + i);
      npd_sg_result_t r;
      r.heat_transfer_rate = 0.0; r.steam_flow_rate = 0.0; r.thermal_efficiency = 0.0;
#ifdef NPD_STEP1_DIAG
      /* SteamGenerator.get_state_dict's step-internal values (steam_generator.py:943-985): the primary temperatures this step was
       * given (:754-755), the overall heat-transfer coefficient from the pressure BEFORE the update (npd_sg_part1's own
       * expression, :170-215 -> :289), and below the feedwater flow the fouled TSPs let through (:761 with :516-547) */
      diag_prev_pressure_sum += has_prev ? g.secondary_pressure : 6.895;      /* secondary/__init__.py:447-453 */
      if (i == NPB_NUM_SG - 1) NPD_DIAG(st, NPB_DIAG_FW_AVG_SG_PRESSURE, diag_prev_pressure_sum / 3);
      NPD_DIAG(st, NPB_DIAG_SG_SCALE_FORMATION_RATE + i,
               npd_scale_formation_rate(&g, (c_inlet + c_outlet) / 2.0, c_flow / (1000.0 * (P.sg_tube_count * (NPD_PI * npd_sq(P.sg_tube_inner_diameter / 2.0))))));
      NPD_DIAG(st, NPB_DIAG_SG_PRIMARY_INLET_TEMP + i, c_inlet); NPD_DIAG(st, NPB_DIAG_SG_PRIMARY_OUTLET_TEMP + i, c_outlet);
      {
        double flow_factor = npd_powc(c_flow / P.sg_primary_design_flow, 0.8);
        double h_primary = P.sg_primary_htc * flow_factor;
        double pressure_factor = npd_powc(g.secondary_pressure / P.sg_design_pressure_secondary, 0.15);
        double h_secondary = P.sg_secondary_htc * pressure_factor;
        double r_primary = 1.0 / h_primary;
        double r_wall = P.sg_tube_wall_thickness / P.sg_tube_conductivity;
        double r_secondary = 1.0 / h_secondary;
        NPD_DIAG(st, NPB_DIAG_SG_OVERALL_HTC + i, 1.0 / (r_primary + r_wall + r_secondary));
      }
#endif
      npd_sg_update(&g, &P, c_inlet, c_outlet, c_flow, demand, fwflow, actual_feedwater_temp, dt * 60, &r);
#ifdef NPD_STEP1_DIAG
      NPD_DIAG(st, NPB_DIAG_SG_FEEDWATER_FLOW_RATE + i, npd_pymin(fwflow, P.sg_design_feedwater_flow_per_sg * (1.0 / npd_sqrt(g.tsp_pressure_drop_ratio))));
#endif
      sg_total_thermal += r.heat_transfer_rate; sg_total_steam += r.steam_flow_rate;
      sg_ap += g.secondary_pressure; sg_at += g.secondary_temperature; sg_aq += g.steam_quality;
      if (i == 0) sg_pressures[0] = g.secondary_pressure;
      else if (i == 1) sg_pressures[1] = g.secondary_pressure;
      else sg_pressures[2] = g.secondary_pressure;
      if (r.thermal_efficiency > 0.1) sg_effective++;
      /* boundary: SG i -> HBM, SG i+1 -> the same registers, stage SG i+2 / the turbine scalars */
      NPD_DMA_WAIT();
      NPD_ST_STORE_ELIDE(SG, npb_sg_t, g, g_old, i);
      NPD_ST_F64_ELIDE(SEC, npb_sec_t, prev_sg_levels, 0, i, g.water_level, npd_sel3(i, pl_old[0], pl_old[1], pl_old[2]));
      NPD_ST_F64(SEC, npb_sec_t, prev_sg_steam_flows, 0, i) = r.steam_flow_rate;
      NPD_ST_F64(SEC, npb_sec_t, prev_sg_qualities, 0, i) = g.steam_quality;
      if (i + 1 < NPB_NUM_SG) { NPD_CONSUME(SG, npb_sg_t, g, 0); g_old = g; }
      NPD_LDS_DRAIN();
      if (i + 2 < NPB_NUM_SG) NPD_DMA(SG, i + 2, 0);
      else if (i + 2 == NPB_NUM_SG && full) NPD_DMA(TURB, 0, 0);
    }
  }
  const double sg_avg_pressure = sg_ap / NPB_NUM_SG, sg_avg_temperature = sg_at / NPB_NUM_SG, sg_avg_quality = sg_aq / NPB_NUM_SG;
  const int sg_system_availability = sg_effective >= (NPB_NUM_SG - 1);

  double electrical_power = 0.0, thermal_efficiency = 0.0, condenser_pressure = 0.007;
  double total_system_heat_rejection = 0.0, turbine_gross_power = 0.0;
  double turbine_efficiency = 0.0, turbine_hp_power = 0.0, turbine_lp_power = 0.0;
  if (full) {
    NPD_STAMP(11);
    /* ================= phase 3: turbine (dt in hours, load demand in PERCENT, :564-569) ========== */
    npd_turbine_result_t tr;
    npb_turb_t t;
    /* boundary: turbine scalars (staged during SG 2) -> registers; stage the 70 stage-array columns, which
     * land while the lubrication step and stage passes A / B run */
    NPD_DMA_WAIT();
    NPD_CONSUME(TURB, npb_turb_t, t, 0);
    const npb_turb_t t_old = t;
    NPD_LDS_DRAIN();
    NPD_DMA(TSTG, 0, 0);
    /* the condenser, both WaterChemistry instances and the pH controller go to REGISTERS now, by plain loads
     * issued behind the stage-array DMA: vector memory completes in issue order, so they have long landed
     * when pass C's 70 stores are still draining, and the condenser needs neither an LDS-DMA issued into a
     * full store queue after pass C (measured: 17k cycles of blocked issue) nor a wait for it */
    npb_cond_t cd; npb_chem_t ch; npb_chem_t ch0; npb_ph_t ph;
    NPD_ST_LOAD(COND, npb_cond_t, cd, 0);
    NPD_ST_LOAD(CHEM, npb_chem_t, ch, 1);
    NPD_ST_LOAD(CHEM, npb_chem_t, ch0, 0);
    NPD_ST_LOAD(PH, npb_ph_t, ph, 0);
    const npb_cond_t cd_old = cd; const npb_chem_t ch_old = ch, ch0_old = ch0; const npb_ph_t ph_old = ph;
    npd_turbine_update<NPD_SM>(&t, st, sg_avg_pressure, sg_avg_temperature, sg_total_steam, sg_pressures, sg_system_availability,
                       load_demand, 0.007, dt / 60.0, &tr);
    NPD_STAMP(18);
    /* ================= phase 4: condenser (:591-621) ================= */
    double lp_exhaust_quality = 0.90;
    {
      double h_f = npd_cond_hf(tr.condenser_pressure), h_g = npd_cond_hg(tr.condenser_pressure);
      double h_fg = h_g - h_f;
      if (h_fg > 0) {
        lp_exhaust_quality = (tr.lp6_outlet_enthalpy - h_f) / h_fg;
        lp_exhaust_quality = npd_pymax(0.0, npd_pymin(1.0, lp_exhaust_quality));
      }
    }
    npd_condenser_result_t cr;
    {
      NPD_ST_STORE_ELIDE(TURB, npb_turb_t, t, t_old, 0);
#ifdef NPD_STEP1_DIAG
      double cond_diag[16];
      npd_condenser_update(&cd, &ch, tr.condenser_pressure, tr.effective_steam_flow, lp_exhaust_quality, 45000.0,
                           cooling_water_temperature, 1.2, 185.0, dt / 60.0, &cr, cond_diag);
#pragma unroll
      for (int q = 0; q < 5; q++) NPD_DIAG(st, NPB_DIAG_COND_OVERALL_HTC + q, cond_diag[q]);
      if (st.diag) {
#pragma unroll
        for (int e = 0; e < 2; e++) {
          NPD_DIAG(st, NPB_DIAG_COND_SJE_CAPACITY + e, cond_diag[5 + e]); NPD_DIAG(st, NPB_DIAG_COND_SJE_STEAM_FLOW + e, cond_diag[7 + e]);
          NPD_DIAG(st, NPB_DIAG_COND_SJE_STEAM_CONSUMPTION + e, cond_diag[9 + e]);
          if (cond_diag[11 + e] >= 0.0) NPD_DIAG(st, NPB_DIAG_COND_SJE_COMPRESSION_RATIO + e, cond_diag[11 + e]);   /* carried while the ejector rests */
          st.diag[(size_t)(NPB_DIAG_COND_SJE_OPERATING_HOURS + e) * st.diag_pitch] += cond_diag[13 + e];              /* vacuum_pump.py:269-270 */
        }
        NPD_DIAG(st, NPB_DIAG_COND_AIR_REMOVAL, cond_diag[15]);
      }
#else
      npd_condenser_update(&cd, &ch, tr.condenser_pressure, tr.effective_steam_flow, lp_exhaust_quality, 45000.0,
                           cooling_water_temperature, 1.2, 185.0, dt / 60.0, &cr);
#endif
      NPD_ST_STORE_ELIDE(COND, npb_cond_t, cd, cd_old, 0);
      NPD_ST_STORE_ELIDE(CHEM, npb_chem_t, ch, ch_old, 1);
    }
    condenser_pressure = cr.condenser_pressure;
    NPD_STAMP(19);
    /* ================= chemistry sidecar: shared WaterChemistry + pH controller (:634-665) ========= */
#ifdef NPD_STEP1_DIAG
    double diag_aggressiveness;
    npd_chemistry_sidecar(&ch0, &ph, dt, &diag_aggressiveness);
    NPD_DIAG(st, NPB_DIAG_FW_PERFORMANCE_FACTOR, (diag_perf_n > 0 ? diag_perf_sum / diag_perf_n : 0.0) * (1.0 - diag_aggressiveness * 0.1) * diag_health);
#else
    npd_chemistry_sidecar(&ch0, &ph, dt);
#endif
    NPD_ST_STORE_ELIDE(CHEM, npb_chem_t, ch0, ch0_old, 0);
    NPD_ST_STORE_ELIDE(PH, npb_ph_t, ph, ph_old, 0);
    NPD_STAMP(20);
    /* ================= electrical-power gates (:750-932) ================= */
    double turbine_electrical_power = tr.electrical_power_net;
    turbine_gross_power = tr.electrical_power_gross;
    turbine_efficiency = tr.overall_efficiency; turbine_hp_power = tr.hp_power; turbine_lp_power = tr.lp_power;
    total_system_heat_rejection = (primary_thermal_power - turbine_electrical_power) * 1e6;
    double power_reduction_factor = 1.0;
    if (fw_total_flow < 300.0) power_reduction_factor = 0.0;
    if (power_reduction_factor > 0.0) {
      if (sg_total_steam < (300.0 * 0.5)) power_reduction_factor *= 0.1;
      if (sg_avg_pressure < (1.0 * 0.5)) power_reduction_factor *= 0.1;
      if (primary_thermal_power > (primary_thermal_power * 1.1)) power_reduction_factor = 0.0;
    }
    electrical_power = turbine_electrical_power * power_reduction_factor;
    thermal_efficiency = (primary_thermal_power > 0) ? electrical_power / primary_thermal_power : 0.0;
    if (tr.trip_active) trip_flags |= NPB_TRIP_TURBINE;
  } else {
    fw_total_flow = sg_total_steam; /* config-2 mode: feedwater == steam demand */
  }

  /* ================= secondary-level state write-back ================= */
  NPD_ST_F64_ELIDE(SEC, npb_sec_t, previous_feedwater_temp, 0, 0, actual_feedwater_temp, prev_feedwater_temp);
  NPD_ST_F64_ELIDE(SEC, npb_sec_t, cooling_water_temperature, 0, 0, cooling_water_temperature, cw_old);
  NPD_ST_F64(SEC, npb_sec_t, operating_hours, 0, 0) = operating_hours + dt / 3600.0;
  { /* the section's outputs and flags: narrow members, stored as whole columns */
    npb_sec_t so;
    so.electrical_power_output = electrical_power; so.thermal_efficiency = thermal_efficiency;
    so.total_steam_flow = sg_total_steam; so.total_heat_transfer = sg_total_thermal; so.total_feedwater_flow = fw_total_flow;
    so.load_demand = load_demand; so.sg_avg_pressure = sg_avg_pressure; so.sg_avg_temperature = sg_avg_temperature;
    so.sg_avg_quality = sg_avg_quality; so.has_previous_sg_conditions = 1; so.sg_system_availability = sg_system_availability;
    NPD_ST_STORE_NARROW(SEC, npb_sec_t, so, 0);
  }

  /* ================= _apply_secondary_to_primary_feedback  sim.py:429-498 ================= */
  double heat_removal_factor = sg_total_steam / 1665.0;
  if (!fw_available) heat_removal_factor *= 0.5;
  NPD_ST_F64(PRIM, npb_prim_t, steam_flow_rate, 0, 0) = sg_total_steam;
  NPD_ST_F64(PRIM, npb_prim_t, last_heat_removal_factor, 0, 0) = heat_removal_factor;

  /* ================= observation / reward / done / flags / info ================= */
  obs[7] = sg_total_steam / 3000;
  obs[12] = electrical_power / 1100;
  obs[13] = thermal_efficiency / 0.35;
  obs[14] = sg_total_steam / 1665;
  obs[15] = load_demand / 100;
  obs[16] = 227.0 / 250;
  obs[17] = cooling_water_temperature / 35;
  obs[18] = fw_total_flow / 1665;
  obs[19] = fw_total_power / 40;
  obs[20] = (double)fw_available;
  obs[21] = fw_total_flow / 1665;

  /* calculate_reward  sim.py:521-542 */
  double efficiency_reward = (thermal_efficiency - 0.30) * 10;
  double target_electrical_power = load_demand / 100.0 * 1100.0;
  double electrical_reward = -fabs(electrical_power - target_electrical_power) / 100;
  double steam_pressure_penalty = 0;
  if (sg_avg_pressure < 5.0 || sg_avg_pressure > 8.0) steam_pressure_penalty = -fabs(sg_avg_pressure - 6.895) * 5;
  double condenser_penalty = 0;
  if (condenser_pressure > 0.01) condenser_penalty = -(condenser_pressure - 0.007) * 100;
  double secondary_reward = efficiency_reward + electrical_reward + steam_pressure_penalty + condenser_penalty;
  double reward = base_reward + secondary_reward * 0.5;

  if (scram_status) trip_flags |= NPB_TRIP_SCRAM;
  if (scram_fired) trip_flags |= NPB_TRIP_SCRAM_FIRED;
  if (nan_reset) trip_flags |= NPB_TRIP_NAN_RESET;

  if (live) {
    /* per-step outputs are streamed with the non-temporal bit: nobody on the GPU reads them back before the next
     * step, and without it their 17 MB per step push arena lines out of the Infinity Cache, which the arena of
     * 65 536 plants only just fits (tools/membench/ntstore.hip) */
    if (reward_out) __builtin_nontemporal_store(reward, &reward_out[p]);
    if (done_out) __builtin_nontemporal_store((uint8_t)scram_fired, &done_out[p]);
    if (trip_out) __builtin_nontemporal_store(trip_flags, &trip_out[p]);
  }
  NPD_STAMP(21);
  if (obs_out) npd_store_rows<NPB_OBS_DIM>(obs, obs_out, lds, block_base, (size_t)n_plants);
  if (info_out) {
    /* info  sim.py:199-250 with the non-finite substitutions of :231-240 */
    info[NPB_INFO_ELECTRICAL_POWER] = isfinite(electrical_power) ? electrical_power : 0.0;
    info[NPB_INFO_THERMAL_EFFICIENCY] = npd_pymax(0.0, npd_pymin(isfinite(thermal_efficiency) ? thermal_efficiency : 0.0, 0.35));
    info[NPB_INFO_STEAM_FLOW] = isfinite(sg_total_steam) ? sg_total_steam : 1665.0;
    info[NPB_INFO_STEAM_PRESSURE] = isfinite(sg_avg_pressure) ? sg_avg_pressure : 6.895;
    info[NPB_INFO_CONDENSER_PRESSURE] = isfinite(condenser_pressure) ? condenser_pressure : 0.007;
    info[NPB_INFO_CONDENSER_HEAT_REJECTION] = isfinite(total_system_heat_rejection) ? total_system_heat_rejection : 0.0;
    info[NPB_INFO_FEEDWATER_FLOW] = fw_total_flow;
    info[NPB_INFO_SG_HEAT_TRANSFER] = sg_total_thermal; info[NPB_INFO_TURBINE_POWER] = turbine_gross_power;
    info[NPB_INFO_FEEDWATER_POWER] = fw_total_power; info[NPB_INFO_PRIMARY_THERMAL_POWER] = primary_thermal_power;
    info[NPB_INFO_TURBINE_EFFICIENCY] = turbine_efficiency;
    info[NPB_INFO_TURBINE_HP_POWER] = turbine_hp_power; info[NPB_INFO_TURBINE_LP_POWER] = turbine_lp_power;
    npd_store_rows<NPB_INFO_DIM>(info, info_out, lds, block_base, (size_t)n_plants);
  }
  NPD_STAMP(22);
  NPD_WAIT_ACC_STORE();
  /* ================= automatic maintenance (sim.py:208-223), for a wave whose screen found something: rarely ================= */
  if (maint && (maint_hit_bits | maint_due_with_orders)) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      /* this wave's own state stores are in memory before the rule reads them back */
    npd_maint_rule_for_wave<NPD_STEP1_WHO>(maint_rc, MC, f64, N, p, maint_hit_bits, maint_due_with_orders);
  }
}
