/*
 * npd_step4.h -- the fused plant-step kernel, FOUR wavefronts per 64 plants (included by npb_kernels.hip behind npd_step2.h,
 * whose helpers it shares).
 *
 * Why: what bounds a step at and below 32 768 plants is not bandwidth but the length of one wave's instruction stream (a lone
 * wave of the one-wave kernel needs ~75 us however empty the chip is; the two-wave kernel's longer wave ~55 us, DESIGN.md
 * section 3).  The plant has more parallelism than two waves use: the four pumps are independent of each other once the level
 * control has handed out the demand, the three steam generators once the primary side is known, the twenty saturation states of
 * the turbine's pass B, the per-stage degradation / metal-temperature updates, and the chemistry sidecar and the lubrication
 * pre-step depend on nothing at all.  Here a group of 64 plants has four waves (block = 256 threads, lane l of every wave is
 * plant l), each <= 256 registers so that two groups share a CU's SIMDs at 32 768 plants (2 048 waves, two per SIMD: one wave's
 * scalar and memory instructions issue beside the other's vector ones).  The waves do not meet at barriers: every hand-over is a
 * progress word in LDS that the producer raises behind its data (a wave's LDS operations complete in order) and the consumer polls,
 * so each wave runs as far ahead as its inputs allow:
 *
 *   wave 3  prelude, level control -> pump 0 -> [pumps 1-3] pump tails, system level -> turbine section -> [SG 0-2] SG sums,
 *           stage pass A (publishing each stage's pressure as it is known), its verdict -> pass B units 12-14 -> stage post 2,5,8,11
 *           -> [chain] rotor -> [stage post] protection, power gates, tail scalars -> turbine section to the arena
 *   wave 0  primary side -> [level control] pump 3 -> SG 0 part 1, [feedwater flow] part 2 -> pass B units 0,2,..,10 as pass A
 *           reaches them -> [verdict] stage post 0,3,6,.. behind the chain -> [tail] observation, done, trip flags
 *   wave 1  chemistry sidecar -> [level control] pump 1 -> [primary] SG 1 -> stage efficiencies -> the stage chain, each stage as
 *           soon as its pass B unit is there (i.e. while pass A still runs) -> [verdict] -> turbine lubrication pre-step ->
 *           [tail, condenser] reward, write-back
 *   wave 2  [level control] pump 2 -> [primary] SG 2 -> units 1,3,..,11 -> [verdict] stage post 1,4,7,.. -> [chain] condenser
 *
 * ([..] = what the wave waits for.)  While a wave is on the group's critical path -- the level control, the pump tails, pass A, the
 * chain, rotor to tail, the condenser, the observation -- it raises its priority (s_setprio).  The same kernel serves batches of
 * 45 057 .. 114 688 plants, whose groups no longer fit at once, on the segmented arena such handles have (npb_kernels.hip).
 * Exactness: every device function is the one the other kernels call, sums over pumps / steam
 * generators / stages are taken in the reference's order by one wave from the values the others publish, and the one sequential
 * dependence between pumps (the demand gate of FeedwaterPumpSystem.update_system, npd_step2.h) makes each pump wait for the one
 * before it and take the real count.
 */
#ifndef NPD_STEP4_H
#define NPD_STEP4_H

#define NPD4_THREADS 256
/* a section back to the arena.  The unchanged-column elision of the other kernels (npb_kernels.hip); -DNPD4_NO_ELIDE:
 * plain stores, which spare the registers of the old copies (measured: 51 -> 43 spilled registers, +1.2 % kernel time at 32 768
 * plants, nothing at 8 192: profiles/r3_step4_ab_elide.txt) */
#ifndef NPD4_NO_ELIDE
#define NPD4_ST_STORE(T, stype, s, old, inst) NPD_ST_STORE_ELIDE(T, stype, s, old, inst)
#else
#define NPD4_ST_STORE(T, stype, s, old, inst) NPD_ST_STORE(T, stype, s, inst)
#endif
#define NPD4_SLOTS 142                        /* exchange slots of 64 doubles: 71 KB per group, two groups per CU */
#ifdef NPB_STAMPS
#define NPD4_STAMP(k) do { if (lane == 0 && npb_stamp_buf) npb_stamp_buf[((size_t)blockIdx.x * 4 + wave) * 32 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define NPD4_STAMP(k)
#endif
/* progress words: one 16-byte cell each in slot Y_FLAGS; raised behind the data they announce, polled by the consumer */
enum { FL_CHAIN = 0, FL_FWFLOW = 1, FL_PRIM = 2, FL_FWCTL = 3, FL_PUMP = 4 /* +pump */, FL_SG = 8 /* +sg */, FL_PASSA = 11, FL_UNIT = 12 /* +wave */,
       FL_POST = 15 /* +R */, FL_VERDICT = 18, FL_TAIL = 19, FL_CONDP = 20, FL_LUBE = 21, FL_CHAINDONE = 22, FL_COUNT = 23 };
#define NPD4_FLAGP(n) ((volatile int *)&xch[Y_FLAGS * NPB_WAVE + 2 * (n)])
/* (a wave's LDS operations execute in issue order, so the word needs no wait behind the data, only the compiler's respect) */
#define NPD4_FLAG_SET(n, v) do { asm volatile("" ::: "memory"); *NPD4_FLAGP(n) = (v); asm volatile("" ::: "memory"); } while (0)
#ifndef NPD4_POLL_SLEEP
#define NPD4_POLL_SLEEP 1                     /* s_sleep argument between two polls of a progress word (x 64 clocks) */
#endif
#define NPD4_FLAG_WAIT(n, v) do { while (__builtin_amdgcn_readfirstlane(*NPD4_FLAGP(n)) < (v)) __builtin_amdgcn_s_sleep(NPD4_POLL_SLEEP); asm volatile("" ::: "memory"); } while (0)
/* pass B unit u (0 = the turbine inlet, k + 1 = stage k): up to 11 the even units are wave 0's and the odd ones wave 2's, the last
 * three wave 3's own once its pass A is through; each wave works through its units in rising order and counts them */
#define NPD4_UNIT_WAIT(u) do { if ((u) < 12) NPD4_FLAG_WAIT(FL_UNIT + (u) % 2, (u) / 2 + 1); else NPD4_FLAG_WAIT(FL_UNIT + 2, (u) - 11); } while (0)

enum {
  Y_EXTF = 0, Y_TIN = 5, Y_STEAM = 6, Y_CHRES = 7,   /* pass A / the chain's results, once the steam generators are done with this region */
  Y_CFLOW = 0, Y_CIN = 3, Y_COUT = 6, Y_LDF = 9, Y_FWTEMP = 10, Y_NPREV = 11, Y_FPP = 12, Y_MAXLVL = 13, Y_RUNCOUNT = 14, Y_FWFLOW = 15,
  Y_PRIM = 16,                                /* base reward, load demand, thermal power, reactivity, primary thermal power, scram bits */
  Y_TIME = 22, Y_FLAGS = 23, Y_MAINT_TAB = 24,
  Y_PUMP = 25,                                /* 4 x X_PUMP_N, until wave 3 has walked the pumps' tails */
  Y_SAT = 25, Y_HG = 40, Y_TRATIO = 55, Y_HGEXT = 69,   /* pass B: 15 + 15 + 14 + 5, once pass A runs (the pumps' values have been read by then) */
  Y_OBS = 25, Y_INFO = 48,                    /* the two transposes, after the chain */
  Y_SG = 77,                                  /* 3 x 6 results */
  Y_TOUT = 77,                                /* per stage, written by the chain (the SG results were summed before pass A) */
  Y_PSELF = 95, Y_PEXT = 109, Y_PIN = 114,
  Y_TAIL = 95,                                /* 21 tail scalars, after the chain (every pass B unit has been consumed) */
  Y_LOADING = 116, Y_STRESS = 130, Y_CWT = 140, Y_CONDP = 141
};
static_assert(Y_PUMP + 4 * X_PUMP_N <= Y_SG && Y_HGEXT + 5 <= Y_SG && Y_INFO + NPB_OBS_PAD <= Y_SG && Y_OBS + NPB_OBS_PAD <= Y_INFO && Y_SG + 18 <= Y_PSELF &&
              Y_TOUT + 14 <= Y_PSELF && Y_TAIL + 21 <= Y_LOADING && Y_PIN < Y_LOADING && Y_LOADING + 14 <= Y_STRESS && Y_STRESS + 8 <= Y_CWT && Y_CHRES + 7 <= Y_FWFLOW &&
              Y_CONDP < NPD4_SLOTS && 2 * FL_COUNT <= NPB_WAVE && NPD_MH_N + 8 <= NPB_WAVE, "exchange slot plan");

/* carried members [K0, K1) of the turbine section back to the arena, with the other kernels' unchanged-column elision: the section
 * has two owners here (the lubrication pre-step's lub_* members / everything else) and the rotor's members are final long before the
 * protection system's */
template <int K0, int K1>
__device__ __forceinline__ void npd4_store_turb_range(const npd_stage_t &st, const npb_turb_t &t, const npb_turb_t &t_old) {
  const double *d = reinterpret_cast<const double *>(&t), *od = reinterpret_cast<const double *>(&t_old);
#pragma unroll
  for (int k = K0; k < K1; k++) {
    if ((NPD_ELIDE_TURB_F >> k) & 1) {
      if (__builtin_amdgcn_ballot_w64(npd_real_bits(d[k]) != npd_real_bits(od[k])) != 0) *NPD_RP(NPD_SEC_COL(TURB, 0) + k) = (npd_real_t)d[k];
    } else {
      *NPD_RP(NPD_SEC_COL(TURB, 0) + k) = (npd_real_t)d[k];
    }
  }
}

/* the stage arrays of the stages k = R, R + 3, R + 6 ... (at most five) into registers / their post-pass behind the chain's flag.
 * The arrays are indexed by the stage's position j in the wave's list, so that the three waves that share this code path keep them
 * in the same registers */
/* NPD4_CHAIN_STORES_DEG = 1 (round 4, measured and NOT kept): the chain's wave, which has to load every stage's efficiency degradation and
 * deposit thickness for the stage efficiencies anyway, also stores their advanced values (old + rate x dt), so that the post-pass waves do not
 * read those 28 columns a second time.  65 536 plants 0.0850 -> 0.0887 ms, 32 768 plants 0.0426 -> 0.0450: 28 stores and 28 adds more on the
 * wave whose chain is the group's critical path cost 4-6 %, and the second reads were hitting the L2 anyway (FETCH_SIZE 255.5 -> 253.1 MB per
 * launch for 14.7 MB fewer bytes requested): profiles/r4_ab_chain_stores_deg_rejected.txt */
#ifndef NPD4_CHAIN_STORES_DEG
#define NPD4_CHAIN_STORES_DEG 0
#endif
#ifndef NPD4_UNITS_FIRST
#define NPD4_UNITS_FIRST 1   /* wave 0: its pass B units before the preload of its stage arrays, which only its post-pass needs -- wave 0's steam generator is
                              * the last to finish, so its first unit is what the chain's first stage waits for (round 4: 32 768 plants 0.04277 -> 0.04245 ms,
                              * three alternating runs each, profiles/r4_ab_units_first.txt) */
#endif
struct npd4_old_t { double eff_deg[5], deposit[5], blade_wear[5], blade_t[5], rotor_t[5], casing_t[5]; };
template <int R>
__device__ __forceinline__ void npd4_stage_preload(const npd_stage_t &st, npd4_old_t &old) {
#pragma unroll
  for (int j = 0; j < 5; j++) {
    const int k = R + 3 * j;
    if (k >= 14) { old.eff_deg[j] = old.deposit[j] = old.blade_wear[j] = old.blade_t[j] = 0.0; }
    else {
      if (NPD4_CHAIN_STORES_DEG) { old.eff_deg[j] = old.deposit[j] = 0.0; }
      else { old.eff_deg[j] = (double)NPD2_TSTG(stage_efficiency_degradation, k < 14 ? k : 0); old.deposit[j] = (double)NPD2_TSTG(stage_deposit_thickness, k < 14 ? k : 0); }
      old.blade_wear[j] = (double)NPD2_TSTG(stage_blade_wear_factor, k < 14 ? k : 0); old.blade_t[j] = (double)NPD2_TSTG(blade_temperatures, k < 14 ? k : 0);
    }
    old.rotor_t[j] = (k < 8) ? (double)NPD2_TSTG(rotor_temperatures, k < 8 ? k : 0) : 0.0;
    old.casing_t[j] = (k < 6) ? (double)NPD2_TSTG(casing_temperatures, k < 6 ? k : 0) : 0.0;
  }
}
template <int R>
__device__ __forceinline__ void npd4_stage_post(const npd_stage_t &st, const npd4_old_t &old, double *xch, int lane, double tdt) {
  /* the chain runs ahead of pass A's verdict on the fast path; the arena is only written once the verdict is in (a group that
   * takes the sequential chain gets its stages a second time, counted from 100) */
  NPD4_FLAG_WAIT(FL_VERDICT, 1);
  const int base = (__builtin_amdgcn_readfirstlane(*NPD4_FLAGP(FL_VERDICT)) == 2) ? 100 : 0;
#pragma unroll
  for (int j = 0; j < 5; j++) {
    const int k = R + 3 * j;
    if (k >= 14) continue;
    NPD4_FLAG_WAIT(FL_CHAIN, base + k + 1);
    double stress = 0.0;
    npd2_stage_post_vals<!NPD4_CHAIN_STORES_DEG>(st, k, old.eff_deg[j], old.deposit[j], old.blade_wear[j], old.rotor_t[j], old.casing_t[j], old.blade_t[j],
                                                 XR(Y_LOADING + (k < 14 ? k : 0)), XR(Y_TOUT + (k < 14 ? k : 0)), tdt, &stress);
    if (k < 8) XW(Y_STRESS + (k < 8 ? k : 0), stress);
  }
  NPD4_FLAG_SET(FL_POST + R, 1);
}
/* this wave's pass B units u = R, R + 3, .. (npd_stage_system_update's pass B, npd_turbine.h): each as soon as wave 3's pass A has
 * published the pressures it needs (progress word FL_PASSA: 1 = the inlet pressure, k + 2 = stage k's outlet and, where the stage
 * has one, its extraction pressure) */
template <int U0, int STEP, int COUNT, int WHOSE>
__device__ __forceinline__ void npd4_pass_b_units(double *xch, int lane) {
#define NPD_EXT_IDX(k) ((k) == 2 ? 0 : (k) == 3 ? 1 : (k) == 4 ? 2 : (k) == 8 ? 3 : 4)
#define NPD_IS_EXT(k) ((k) == 2 || (k) == 3 || (k) == 4 || (k) == 8 || (k) == 9)
#pragma unroll
  for (int j = 0; j < COUNT; j++) {
    const int u = U0 + STEP * j;
    if (u == 0) {
      NPD4_FLAG_WAIT(FL_PASSA, 1);
      const double sat = npd_tsat_antoine(XR(Y_PIN));
      XW(Y_SAT + 0, sat); XW(Y_HG + 0, npd_hg_from_tsat(sat));
    } else {
      const int k = u - 1;
      NPD4_FLAG_WAIT(FL_PASSA, k + 2);
      const double pk = XR(Y_PSELF + (k >= 0 ? k : 0)), pkm = (k == 0) ? XR(Y_PIN) : XR(Y_PSELF + (k > 0 ? k - 1 : 0));
      const double sat = npd_tsat_antoine(pk);
      XW(Y_SAT + u, sat); XW(Y_HG + u, npd_hg_from_tsat(sat)); XW(Y_TRATIO + (k >= 0 ? k : 0), npd_sqrt(npd_sqrt(pk / pkm)));
      if (NPD_IS_EXT(k)) XW(Y_HGEXT + NPD_EXT_IDX(k), npd_hg_from_tsat(npd_tsat_antoine(XR(Y_PEXT + NPD_EXT_IDX(k)))));
    }
    NPD4_FLAG_SET(FL_UNIT + WHOSE, j + 1);
  }
#undef NPD_EXT_IDX
#undef NPD_IS_EXT
}

/* pump i of FeedwaterPumpSystem.update_system, by whichever wave has it: waits for the level control's hand-out (and, with the
 * automatic maintenance, the plants' clock is the caller's maint_clock); if the demand gate could close (npd_step2.h), for the pump
 * before it and its count */
#define NPD4_PUMP(i_, BEFORE_THE_STORE) \
  { \
    const int i = (i_); \
    npb_pump_t pm; \
    NPD_ST_LOAD(PUMP, npb_pump_t, pm, i);      /* on its way while this wave polls */ \
    NPD4_FLAG_WAIT(FL_FWCTL, 1); \
    const int n_prev_running = (int)XR(Y_NPREV); \
    const double flow_per_pump = XR(Y_FPP); \
    npd_pump_sysconds_t sc; \
    sc.feedwater_temperature = 40.0; sc.suction_pressure = 0.5; sc.discharge_pressure = 7.4; sc.max_sg_level = XR(Y_MAXLVL); \
    const double maint_time = maint_clock; \
    if (maint) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); npd_maint_cache_landed(maint_cache); } \
    const uint32_t cooling_mask = (i & 1) ? maint_cache.z : maint_cache.x; \
    const float cooling_until = __uint_as_float((i & 1) ? maint_cache.w : maint_cache.y); \
    const npb_pump_t pm_old = pm; \
    if (!serial_pumps) { \
      npd2_pump(&pm, 1, n_prev_running, flow_per_pump, &sc, dt);        /* the gate cannot close: its outcome needs no count */ \
    } else { \
      int running_count = 0; \
      if (i > 0) { NPD4_FLAG_WAIT(FL_PUMP + (i > 0 ? i - 1 : 0), 1); running_count = (int)XR(Y_RUNCOUNT); } \
      npd2_pump(&pm, running_count < n_prev_running, n_prev_running, flow_per_pump, &sc, dt); \
      XW(Y_RUNCOUNT, (double)(running_count + (pm.status == NPD_PUMP_RUNNING))); \
    } \
    {   /* npd2_publish_pump, into this kernel's region */ \
      const int b = Y_PUMP + i * X_PUMP_N; \
      XW(b + 0, (double)((pm.status == NPD_PUMP_RUNNING) | (pm.trip_active ? 2 : 0))); \
      XW(b + 1, pm.flow_rate); XW(b + 2, pm.power_consumption); XW(b + 3, npd_pump_npsh_required(&pm)); XW(b + 4, pm.npsh_available); \
      XW(b + 5, pm.speed_percent); XW(b + 6, npd_pymax3(pm.wear_motor_bearings, pm.wear_pump_bearings, pm.wear_thrust_bearing)); \
      XW(b + 7, pm.wear_mechanical_seals); XW(b + 8, pm.vibration_level); XW(b + 9, pm.suction_pressure); XW(b + 10, pm.discharge_pressure); \
      XW(b + 11, pm.oil_temperature); XW(b + 12, pm.motor_temperature); \
    } \
    NPD4_FLAG_SET(FL_PUMP + i, 1); \
    if (maint) {   /* anything new at this pump, for any plant of the group?  (npd_maintenance.h) */ \
      if (__builtin_amdgcn_ballot_w64(npd_maint_pump_hit(&pm, maint_tab, cooling_mask, cooling_until, maint_time)) != 0) maint_hit_bits |= 1u << i; \
    } \
    BEFORE_THE_STORE; \
    NPD4_ST_STORE(PUMP, npb_pump_t, pm, pm_old, i); \
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
  const int my_pump = (wave == 3) ? 0 : (wave == 0 ? 3 : wave);         /* the primary side's wave takes the spare pump, normally the cheap one */
  const size_t block_base = (size_t)blockIdx.x * NPB_WAVE;
  if (block_base >= (size_t)n_plants) return;       /* a group of padding only: the whole group leaves */
  NPD_SEGMENT(f64, N, block_base);
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
  if (maint) maint_cache = npd_maint_cache_fetch(MC, p, my_pump >> 1);
  /* could the demand gate close for a later pump?  (every wave evaluates this on the same data, npd_step2.h, read before the
   * barrier below, i.e. before any wave stores a pump or the feedwater section) */
  /* the plants' clock after this step, for the maintenance screen of this wave's pump: the primary side only adds dt to it
   * (npd_primary_update never writes sim_time), so every wave can form it from the column as the previous step left it -- loaded
   * here, ahead of the barrier below, i.e. before wave 0 stores the primary section -- instead of waiting for wave 0 */
  double maint_clock = 0.0;
  if (maint) maint_clock = (double)NPD_ST_F64(PRIM, npb_prim_t, sim_time, 0, 0);
  bool serial_pumps;
  {
    const int fw_mask = *NPD_NP(const int32_t, NPD_SEC_COL(FW, 0) + NPB_FW_NCARRY + (NPB_FW_NOUT + NPB_I32_SLOT(npb_fw_t, FW, running_mask)) / NPD_NPC,
                                (NPB_FW_NOUT + NPB_I32_SLOT(npb_fw_t, FW, running_mask)) % NPD_NPC);
    int n_prev = 0, may_run = 0;
#pragma unroll
    for (int i = 0; i < NPB_NUM_PUMPS; i++) {
      n_prev += (fw_mask >> i) & 1;
      const int stt = *NPD_NP(const int32_t, NPD_SEC_COL(PUMP, i) + NPB_PUMP_NCARRY + (NPB_PUMP_NOUT + NPB_I32_SLOT(npb_pump_t, PUMP, status)) / NPD_NPC,
                              (NPB_PUMP_NOUT + NPB_I32_SLOT(npb_pump_t, PUMP, status)) % NPD_NPC);
      may_run += (stt == NPD_PUMP_RUNNING || stt == NPD_PUMP_STARTING);
    }
    if (wave == 3 && lane < FL_COUNT) *NPD4_FLAGP(lane) = 0;
    serial_pumps = __builtin_amdgcn_ballot_w64(n_prev > 0 && may_run > n_prev) != 0;
    if (maint) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); maint_clock += dt; }
  }
  NPD2_SYNC_();                                                                       /* the only barrier before the end: the progress words are down */

  if (wave == 3) {
    /* ================================ wave 3: feedwater system level, then the turbine ================================ */
    npb_fw_t fw; npb_fw_t fw_old;
    double prev_levels[NPB_NUM_SG];
    double prev_feedwater_temp = 0.0, cw_old = 0.0, operating_hours = 0.0, cooling_water_temperature = 0.0, actual_feedwater_temp = 0.0;
    __builtin_amdgcn_s_setprio(3);                 /* the level control gates all four pumps */
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
      NPD4_FLAG_WAIT(FL_PRIM, 1);
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
    NPD4_FLAG_SET(FL_FWCTL, 1);
    __builtin_amdgcn_s_setprio(0);
    NPD4_STAMP(1);
    NPD4_PUMP(0, (void)0);
    NPD4_STAMP(2);
    NPD4_FLAG_WAIT(FL_PUMP + 1, 1); NPD4_FLAG_WAIT(FL_PUMP + 2, 1); NPD4_FLAG_WAIT(FL_PUMP + 3, 1);
    NPD4_STAMP(3);
    __builtin_amdgcn_s_setprio(3);                 /* the feedwater flow gates the steam generators' second halves */
    npb_turb_t t; npb_turb_t t_old;
    double fw_total_flow = 0.0, fw_total_power = 0.0;
    int fw_available = 0; uint32_t trip_flags = 0;
    /* ---- diagnostics + protection passes over the four pumps, system level (feedwater/physics.py:720-863) */
    npd_fw_acc_t acc;
    acc.total_flow = acc.total_power = acc.flow_sum = 0.0;
    acc.total_cavitation_risk = acc.total_wear_level = acc.total_vibration = 0.0;
    acc.running_count = acc.running_mask = acc.trips = 0; acc.trip_mask = 0; acc.trip_kinds = 0;
#pragma unroll
    for (int i = 0; i < NPB_NUM_PUMPS; i++) npd2_pump_tail(xch + (Y_PUMP - X_PUMP) * NPB_WAVE, lane, i, &fw, &acc, dt);
    npd_fw_result_t fwr;
    npd_fw_finish(&fw, &acc, prev_levels, dt, &fwr);
    fw_total_flow = fwr.total_flow_rate; fw_total_power = fwr.total_power_consumption;
    fw_available = fwr.system_availability;
    trip_flags = (fwr.pump_trip_mask << 8) | (fw.system_trip_active ? NPB_TRIP_FW_SYSTEM : 0);
    /* the steam generators wait for this in their part 2 */
    XW(Y_FWFLOW, fw_total_flow);
    NPD4_FLAG_SET(FL_FWFLOW, 1);
    __builtin_amdgcn_s_setprio(0);
    NPD4_STAMP(4);
    /* while the steam generators run: the turbine section (asked for before the feedwater section goes back) */
    NPD_ST_LOAD(TURB, npb_turb_t, t, 0);          /* wave 1 owns the lub_* members; they are neither used nor stored here */
    NPD4_ST_STORE(FW, npb_fw_t, fw, fw_old, 0);
    t_old = t;
    npd4_old_t old;
    npd4_stage_preload<2>(st, old);               /* this wave's share of the stage post-pass: stages 2, 5, 8, 11 */
    NPD4_STAMP(5);
    NPD4_FLAG_WAIT(FL_SG + 0, 1); NPD4_FLAG_WAIT(FL_SG + 1, 1); NPD4_FLAG_WAIT(FL_SG + 2, 1);
    NPD4_STAMP(6);
    __builtin_amdgcn_s_setprio(3);                 /* pass A gates everything the other three waves do next */
    double sg_total_thermal = 0.0, sg_total_steam = 0.0, sg_avg_pressure = 0.0, sg_avg_temperature = 0.0, sg_avg_quality = 0.0;
    int sg_system_availability = 0;
    double pressure_stability_factor = 1.0, load_demand = 0.0;
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
    XW(Y_CWT, cooling_water_temperature);
    XW(Y_PIN, sg_avg_pressure); XW(Y_TIN, sg_avg_temperature); XW(Y_STEAM, sg_total_steam);
    NPD4_FLAG_SET(FL_PASSA, 1);
    {   /* stage pass A (npd2_stage_pass_a, npd_step2.h: pressures and flows, no transcendentals), each stage's outlet pressure
         * published as it is known so that the other waves' pass B runs behind this loop instead of behind its end */
#define NPD_EXT_IDX(k) ((k) == 2 ? 0 : (k) == 3 ? 1 : (k) == 4 ? 2 : (k) == 8 ? 3 : 4)
#define NPD_IS_EXT(k) ((k) == 2 || (k) == 3 || (k) == 4 || (k) == 8 || (k) == 9)
      const double inlet_pressure = sg_avg_pressure, inlet_flow = sg_total_steam;
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
        if (NPD_IS_EXT(k)) { XW(Y_EXTF + NPD_EXT_IDX(k), ef); XW(Y_PEXT + NPD_EXT_IDX(k), pe); }
        const double flow_out_k = cur_flow - ef;
        rare = rare || !(self_out >= 0.001 && self_out <= 22.0) || !(pe >= 0.001 && pe <= 22.0) || !(outlet_pressure >= 0.001);
        cur_p = self_out; cur_flow = flow_out_k;
        XW(Y_PSELF + k, self_out);
        NPD4_FLAG_SET(FL_PASSA, k + 2);
      }
      seq = __builtin_amdgcn_ballot_w64(rare) != 0;   /* any lane off the fast path: the group takes the sequential chain (pass B and wave 1's chain so far are then not used) */
      NPD4_FLAG_SET(FL_VERDICT, seq ? 2 : 1);
#undef NPD_EXT_IDX
#undef NPD_IS_EXT
    }
    __builtin_amdgcn_s_setprio(0);
    NPD4_STAMP(7);
    npd4_pass_b_units<12, 1, 3, 2>(xch, lane);     /* the last three saturation states, which the other two waves would reach last */
    npd4_stage_post<2>(st, old, xch, lane, tdt);
    NPD4_STAMP(8);
    NPD4_FLAG_WAIT(FL_CHAINDONE, 1);               /* wave 1's chain: total power, extraction, efficiency ... */
    __builtin_amdgcn_s_setprio(3);                 /* rotor -> protection -> tail: the critical path again */
    const double stage_power_mw = XR(Y_CHRES + 0) * pressure_stability_factor;
    const double turbine_efficiency = XR(Y_CHRES + 5), hp_power = XR(Y_CHRES + 3), lp_power = XR(Y_CHRES + 4);
    double max_bearing_metal = 0.0, total_displacement = 0.0;
    npd_turbine_rotor(&t, stage_power_mw, sg_avg_temperature, load_demand, tdt, &max_bearing_metal, &total_displacement);
    NPD4_STAMP(9);
    NPD4_FLAG_WAIT(FL_POST + 0, 1); NPD4_FLAG_WAIT(FL_POST + 1, 1);
    NPD4_STAMP(10);
    double max_stress = 0.0;      /* MetalTemperatureTracker's max over the rotor points, in their order */
#pragma unroll
    for (int k = 0; k < 8; k++) max_stress = (k == 0) ? XR(Y_STRESS) : npd_pymax(max_stress, XR(Y_STRESS + k));
    npd_turbine_protect(&t, stage_power_mw, max_stress, max_bearing_metal, total_displacement, sg_system_availability, 0.007, tdt);
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
    XW(Y_TAIL + 0, electrical_power); XW(Y_TAIL + 1, thermal_efficiency); XW(Y_TAIL + 2, sg_avg_pressure); XW(Y_TAIL + 3, sg_total_steam);
    XW(Y_TAIL + 4, fw_total_flow); XW(Y_TAIL + 5, total_system_heat_rejection); XW(Y_TAIL + 6, sg_total_thermal); XW(Y_TAIL + 7, sg_avg_temperature);
    XW(Y_TAIL + 8, sg_avg_quality); XW(Y_TAIL + 9, (double)(sg_system_availability | (fw_available << 1))); XW(Y_TAIL + 10, prev_feedwater_temp);
    XW(Y_TAIL + 11, cw_old); XW(Y_TAIL + 12, operating_hours); XW(Y_TAIL + 13, t.total_power_output); XW(Y_TAIL + 14, fw_total_power);
    XW(Y_TAIL + 15, turbine_efficiency); XW(Y_TAIL + 16, hp_power); XW(Y_TAIL + 17, lp_power); XW(Y_TAIL + 18, (double)trip_flags);
    XW(Y_TAIL + 19, actual_feedwater_temp); XW(Y_TAIL + 20, (double)fw_available);
    NPD4_FLAG_SET(FL_TAIL, 1);
    __builtin_amdgcn_s_setprio(0);
    NPD4_STAMP(11);
    {   /* the turbine section but for the lubrication's members, behind the tail (nobody waits for it) and behind wave 2's load of
         * the PREVIOUS step's rotor state for the lubrication pre-step (update_with_lubrication reads it before the rotor moves) */
      constexpr int L0 = NPB_F64_SLOT(npb_turb_t, lub_oil_temperature);
      static_assert(NPB_F64_SLOT(npb_turb_t, load_demand) + 1 == L0, "turbine section layout");
      NPD4_FLAG_WAIT(FL_LUBE, 1);
      npd4_store_turb_range<0, L0>(st, t, t_old);
      static_assert(NPB_TURB_NOUT == 4 && NPB_TURB_NI32 == 2, "turbine narrow layout");
      constexpr int NC = NPB_TURB_NCARRY;
      *NPD_NP(float, NPD_SEC_COL(TURB, 0) + NC + 0 / NPD_NPC, 0 % NPD_NPC) = (float)t.thermal_expansion;
      *NPD_NP(float, NPD_SEC_COL(TURB, 0) + NC + 1 / NPD_NPC, 1 % NPD_NPC) = (float)t.total_power_output;
      *NPD_NP(float, NPD_SEC_COL(TURB, 0) + NC + 2 / NPD_NPC, 2 % NPD_NPC) = (float)t.vibration_displacement;
      *NPD_NP(int32_t, NPD_SEC_COL(TURB, 0) + NC + 4 / NPD_NPC, 4 % NPD_NPC) = t.trip_active;
      *NPD_NP(int32_t, NPD_SEC_COL(TURB, 0) + NC + 5 / NPD_NPC, 5 % NPD_NPC) = t.trip_latched_mask;
    }
    NPD4_FLAG_WAIT(FL_CONDP, 1);                   /* wave 2's condenser */
    NPD4_STAMP(12);
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
    NPD4_STAMP(13);
  } else {
    /* ================================ waves 0 .. 2 ================================ */
    int scram_bits = 0;                               /* wave 0: scram_status | scram_fired << 1 | nan_reset << 2 */
    if (wave == 1) {   /* while the level control runs */
      /* ---- chemistry sidecar: shared WaterChemistry + pH controller (secondary/__init__.py:634-665) */
      npb_chem_t ch0; npb_ph_t ph;
      NPD_ST_LOAD(CHEM, npb_chem_t, ch0, 0);
      NPD_ST_LOAD(PH, npb_ph_t, ph, 0);
      const npb_chem_t ch0_old = ch0; const npb_ph_t ph_old = ph;
      npd_chemistry_sidecar(&ch0, &ph, dt);
      NPD4_ST_STORE(CHEM, npb_chem_t, ch0, ch0_old, 0);
      NPD4_ST_STORE(PH, npb_ph_t, ph, ph_old, 0);
    }
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
      NPD4_FLAG_SET(FL_PRIM, 1);
      s.has_heat_removal_factor = 1;
      NPD_ST_STORE_ELIDE_PRIM(s, s_old);
      if (maint) {   /* sim.py:208-216 as far as no work order is involved; t = the clock after this step */
        const bool work = npd_maint_due_decide(&maint_due, s.sim_time, MH.tab[2 * NPB_MAINT_NPARAM + 1]);
        maint_due_with_orders = __builtin_amdgcn_ballot_w64(work) != 0 ? 1u : 0u;
      }
    }
    NPD4_STAMP(1);
    npb_sg_t g;                                       /* this wave's steam generator: its section is asked for before the pump's goes back */
    NPD4_PUMP(my_pump, NPD_ST_LOAD(SG, npb_sg_t, g, wave));
    NPD4_STAMP(2);
    {
      const int i = wave;
      NPD4_FLAG_WAIT(FL_PRIM, 1);
      NPD4_STAMP(3);
      /* ---- steam generator i (enhanced_physics.py:433-547): part 1 needs the primary side only, part 2 waits for wave 3's feedwater flow */
      const double c_flow = XR(Y_CFLOW + i), c_in = XR(Y_CIN + i), c_out = XR(Y_COUT + i);
      const double load_demand_fraction = XR(Y_LDF), feedwater_temp = XR(Y_FWTEMP);
      double total_primary_flow = 0.0;
      total_primary_flow += XR(Y_CFLOW + 0); total_primary_flow += XR(Y_CFLOW + 1); total_primary_flow += XR(Y_CFLOW + 2);
      const double actual_total_steam_flow = P.sg_design_total_steam_flow * load_demand_fraction;
      const double demand = (total_primary_flow > 0) ? actual_total_steam_flow * (c_flow / total_primary_flow) : actual_total_steam_flow / NPB_NUM_SG;
      const npb_sg_t g_old = g;
      const double level_old = (double)NPD_ST_F64(SEC, npb_sec_t, prev_sg_levels, 0, i);
      const double heat_transfer = npd_sg_part1(&g, &P, c_in, c_out, c_flow, dt * 60);
      NPD4_STAMP(4);
      NPD4_FLAG_WAIT(FL_FWFLOW, 1);
      NPD4_STAMP(5);
      const double fw_flow = XR(Y_FWFLOW);
      npd_sg_result_t r;
      r.heat_transfer_rate = 0.0; r.steam_flow_rate = 0.0; r.thermal_efficiency = 0.0;
      npd_sg_part2(&g, &P, heat_transfer, demand, fw_flow / NPB_NUM_SG, feedwater_temp, dt * 60, &r);
      const int b = Y_SG + 6 * i;
      XW(b + 0, r.heat_transfer_rate); XW(b + 1, r.steam_flow_rate); XW(b + 2, g.secondary_pressure);
      XW(b + 3, g.secondary_temperature); XW(b + 4, g.steam_quality); XW(b + 5, r.thermal_efficiency > 0.1 ? 1.0 : 0.0);
      NPD4_ST_STORE(SG, npb_sg_t, g, g_old, i);
      NPD_ST_F64_ELIDE(SEC, npb_sec_t, prev_sg_levels, 0, i, g.water_level, level_old);
      NPD_ST_F64(SEC, npb_sec_t, prev_sg_steam_flows, 0, i) = (npd_real_t)r.steam_flow_rate;
      NPD_ST_F64(SEC, npb_sec_t, prev_sg_qualities, 0, i) = (npd_real_t)g.steam_quality;
      NPD4_FLAG_SET(FL_SG + i, 1);
    }
    NPD4_STAMP(6);
    npd4_old_t old;
    npb_cond_t cd, cd_old; npb_chem_t chc, chc_old;   /* wave 2: the condenser's sections, loaded ahead of its stage post-pass */
    if (wave == 1) {
      /* ---- the stage chain (npd_stage_system_update's pass C, npd_turbine.h), behind the other waves' pass B units, i.e. while
       * wave 3's pass A is still walking the later stages.  The 14 stages' efficiency products first (TurbineStage state as the
       * previous step left it, stage_system.py:128-133, 294-339: loaded before any stage post-pass writes) */
      double stage_eff[14];
#pragma unroll
      for (int k = 0; k < 14; k++) {
        const double deposit = (double)NPD2_TSTG(stage_deposit_thickness, k), eff_deg = (double)NPD2_TSTG(stage_efficiency_degradation, k);
        double fouling_factor = 1.0 / (1.0 + deposit / 0.5);
        double blade_wear_factor = (double)NPD2_TSTG(stage_blade_wear_factor, k);
        double blade_condition_factor = npd_pymin(fouling_factor, blade_wear_factor);
        double actual_efficiency = npd_pymax(0.7, 0.88 - eff_deg);
        stage_eff[k] = (actual_efficiency * blade_condition_factor * fouling_factor * blade_wear_factor * 1.0);
        if (NPD4_CHAIN_STORES_DEG) {   /* the stage post-pass's two rate updates (npd2_stage_post_vals), here where the old values are in registers */
          NPD2_TSTG(stage_efficiency_degradation, k) = (npd_real_t)(eff_deg + 1e-05 * tdt);
          NPD2_TSTG(stage_deposit_thickness, k) = (npd_real_t)(deposit + 5e-05 * tdt);
        }
      }
      NPD4_STAMP(7);
      __builtin_amdgcn_s_setprio(3);               /* the chain is the group's critical path from here to its last stage */
      NPD4_FLAG_WAIT(FL_PASSA, 1);
      const double sg_avg_pressure = XR(Y_PIN), sg_avg_temperature = XR(Y_TIN), sg_total_steam = XR(Y_STEAM), load_demand = XR(Y_PRIM + 1);
      npd2_chain_t ch;
      ch.T_in = sg_avg_temperature; ch.total_power = 0.0; ch.total_extraction = 0.0; ch.lp6_outlet_enthalpy = 0.0;
      ch.hp_power = 0.0; ch.lp_power = 0.0; ch.h_in0 = 0.0;
      double turbine_efficiency = 0.0;   /* stage_system.py:983-993 (info only) */
#define NPD_EXT_IDX(k) ((k) == 2 ? 0 : (k) == 3 ? 1 : (k) == 4 ? 2 : (k) == 8 ? 3 : 4)
#define NPD_IS_EXT(k) ((k) == 2 || (k) == 3 || (k) == 4 || (k) == 8 || (k) == 9)
      {   /* the fast path, ahead of pass A's verdict (a group that turns out to need the sequential chain runs it below) */
        NPD4_UNIT_WAIT(0);                           /* the inlet's saturation state */
        ch.sat_in = XR(Y_SAT + 0); ch.hg_in = XR(Y_HG + 0);
        double cur_flow = sg_total_steam;
#pragma unroll
        for (int k = 0; k < 14; k++) {
          NPD4_UNIT_WAIT(k + 1);                     /* stage k's, and its extraction's if it has one (pass A has then published stage k) */
          const double p_in = (k == 0) ? sg_avg_pressure : XR(Y_PSELF + (k > 0 ? k - 1 : 0)), p_self_k = XR(Y_PSELF + k);
          const double sat_k = XR(Y_SAT + k + 1), hg_k = XR(Y_HG + k + 1), tr_k = XR(Y_TRATIO + k);
          const double ef = NPD_IS_EXT(k) ? XR(Y_EXTF + NPD_EXT_IDX(k)) : 0.0;
          const double hgx = NPD_IS_EXT(k) ? XR(Y_HGEXT + NPD_EXT_IDX(k)) : 0.0;
          const double flow_out_k = cur_flow - ef;   /* pass A's own recurrence (npd2_stage_pass_a) */
          cur_flow = flow_out_k;
          double T_out, loading;
          npd2_chain_stage(k, ch, p_in, p_self_k, sat_k, hg_k, tr_k, flow_out_k, ef, hgx, stage_eff[k], &T_out, &loading);
          XW(Y_TOUT + k, T_out); XW(Y_LOADING + k, loading);
          NPD4_FLAG_SET(FL_CHAIN, k + 1);
        }
        {   /* _steam_enthalpy at the last stage's outlet, whose saturation state pass B has */
          const double p13 = XR(Y_PSELF + 13);
          const double T_c = npd_pymax(0.0, npd_pymin(ch.T_in, 800.0));
          const double cp = (p13 > 10.0) ? 2.5 : ((p13 > 1.0) ? 2.2 : 2.0);
          const double h_out = (T_c <= ch.sat_in) ? ch.hg_in : ch.hg_in + cp * (T_c - ch.sat_in);
          if (sg_total_steam > 0) turbine_efficiency = (ch.h_in0 - h_out) / ch.h_in0;
        }
      }
      NPD4_FLAG_WAIT(FL_VERDICT, 1);
      if (__builtin_amdgcn_readfirstlane(*NPD4_FLAGP(FL_VERDICT)) == 2) {   /* some lane left the fast path's assumptions: the reference's own order, stage by stage */
        ch.T_in = sg_avg_temperature; ch.total_power = 0.0; ch.total_extraction = 0.0; ch.lp6_outlet_enthalpy = 0.0;
        ch.hp_power = 0.0; ch.lp_power = 0.0; ch.h_in0 = 0.0;
        turbine_efficiency = 0.0;
        double cur_p = sg_avg_pressure, cur_T = sg_avg_temperature, cur_flow = sg_total_steam;
#pragma unroll
        for (int k = 0; k < 14; k++) {
          double T_out, loading;
          npd2_seq_stage(k, cur_p, cur_T, cur_flow, sg_total_steam, load_demand, stage_eff[k], ch, &T_out, &loading);
          XW(Y_TOUT + k, T_out); XW(Y_LOADING + k, loading);
          NPD4_FLAG_SET(FL_CHAIN, 100 + k + 1);
        }
        if (sg_total_steam > 0) {
          const double h_in = npd_stage_steam_enthalpy(sg_avg_temperature, sg_avg_pressure);
          turbine_efficiency = (h_in - npd_stage_steam_enthalpy(cur_T, cur_p)) / h_in;
        }
      }
#undef NPD_EXT_IDX
#undef NPD_IS_EXT
      XW(Y_CHRES + 0, ch.total_power); XW(Y_CHRES + 1, ch.total_extraction); XW(Y_CHRES + 2, ch.lp6_outlet_enthalpy);
      XW(Y_CHRES + 3, ch.hp_power); XW(Y_CHRES + 4, ch.lp_power); XW(Y_CHRES + 5, turbine_efficiency);
      XW(Y_CHRES + 6, sg_total_steam - ch.total_extraction);     /* the effective steam flow the condenser sees */
      NPD4_FLAG_SET(FL_CHAINDONE, 1);
      __builtin_amdgcn_s_setprio(0);
      NPD4_STAMP(8);
    } else if (wave == 0) {
#if NPD4_UNITS_FIRST
      npd4_pass_b_units<0, 2, 6, 0>(xch, lane); NPD4_STAMP(7); npd4_stage_preload<0>(st, old); NPD4_STAMP(8); npd4_stage_post<0>(st, old, xch, lane, tdt);
#else
      npd4_stage_preload<0>(st, old); NPD4_STAMP(7); npd4_pass_b_units<0, 2, 6, 0>(xch, lane); NPD4_STAMP(8); npd4_stage_post<0>(st, old, xch, lane, tdt);
#endif
    } else {
      npd4_stage_preload<1>(st, old); NPD4_STAMP(7); npd4_pass_b_units<1, 2, 6, 1>(xch, lane); NPD4_STAMP(8);
      NPD_ST_LOAD(COND, npb_cond_t, cd, 0);        /* for the condenser, which this wave runs as soon as the chain is through */
      NPD_ST_LOAD(CHEM, npb_chem_t, chc, 1);
      cd_old = cd; chc_old = chc;
      npd4_stage_post<1>(st, old, xch, lane, tdt);
    }
    NPD4_STAMP(9);
    if (wave == 0) {
      /* ---- observation, done, trip flags: the primary part (sim.py:290-333) from the primary section as stored by this wave
       * at the start (carried members: the stored value is the value; power_level, an output member, was published in fp64);
       * loaded before the wait for the tail */
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
      obs[11] = (double)(scram_bits & 1);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      NPD4_FLAG_WAIT(FL_TAIL, 1);
      __builtin_amdgcn_s_setprio(2);
      NPD4_STAMP(10);
      const double ld = XR(Y_PRIM + 1), sg_total_steam_t = XR(Y_TAIL + 3), fw_flow_t = XR(Y_TAIL + 4);
      obs[10] = ld / 100;                          /* load_demand IS state.power_level (sim.py:161) */
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
      NPD4_STAMP(11);
    } else if (wave == 1) {
      /* ---- turbine lubrication pre-step: reads the PREVIOUS step's rotor / bearing members (wave 3 holds its store of them back
       * until they have landed here), owns the lub_* ones */
      {
        npb_turb_t t;
        NPD_ST_LOAD(TURB, npb_turb_t, t, 0);
        const npb_turb_t t_old = t;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        NPD4_FLAG_SET(FL_LUBE, 1);
        npd_turbine_lube(&t, tdt);
        constexpr int L0 = NPB_F64_SLOT(npb_turb_t, lub_oil_temperature), L1 = NPB_F64_SLOT(npb_turb_t, thermal_expansion);
        static_assert(L1 == NPB_TURB_NCARRY, "lub_* carried members close the section");
        npd4_store_turb_range<L0, L1>(st, t, t_old);
        *NPD_NP(float, NPD_SEC_COL(TURB, 0) + NPB_TURB_NCARRY + 3 / NPD_NPC, 3 % NPD_NPC) = (float)t.lub_effectiveness;
      }
      NPD4_STAMP(10);
      NPD4_STAMP(11);
      NPD4_FLAG_WAIT(FL_TAIL, 1); NPD4_FLAG_WAIT(FL_CONDP, 1);
      NPD4_STAMP(12);
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
      NPD4_STAMP(13);
    } else {
      NPD4_FLAG_WAIT(FL_CHAINDONE, 1);
      __builtin_amdgcn_s_setprio(2);               /* condenser -> info is what ends the step */
      /* ---- condenser (secondary/__init__.py:591-621) */
      const double effective_steam_flow = XR(Y_CHRES + 6), lp6_outlet_enthalpy = XR(Y_CHRES + 2), cwt = XR(Y_CWT);
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
      XW(Y_CONDP, cr.condenser_pressure);
      NPD4_FLAG_SET(FL_CONDP, 1);
      NPD4_ST_STORE(COND, npb_cond_t, cd, cd_old, 0);
      NPD4_ST_STORE(CHEM, npb_chem_t, chc, chc_old, 1);
      NPD4_STAMP(10);
      NPD4_STAMP(12);
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
