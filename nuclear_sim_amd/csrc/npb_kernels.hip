/*
 * npb_kernels.hip -- the fused plant-step kernel for gfx950 (MI355X) and its small companions.
 *
 * One wavefront lane = one plant.  State lives in HBM as a struct-of-arrays: column `slot` of the
 * fp64 table is f64[slot * Npad + plant], so every load/store instruction of a wave touches 64
 * consecutive doubles (512 B, fully coalesced).  A plant's ~530 carried scalars do not fit a lane's
 * register file at once, so the step STREAMS the plant subsystem by subsystem in the reference's own
 * order (NuclearPlantSimulator.step, simulator/core/sim.py:130-258; SecondaryReactorPhysics.update_system,
 * systems/secondary/__init__.py:340-1021):
 *
 *   primary -> coupling -> feedwater (4 pumps, one at a time) -> 3 steam generators (one at a time)
 *   -> turbine -> condenser -> electrical-power gates -> feedback -> observation / reward / done
 *
 * Each phase's section struct reaches the registers through the LDS staging pipeline of npd_stage.h (LDS-DMA
 * one phase ahead, because one wave per SIMD has nothing else to hide HBM latency behind), is updated in
 * registers and stored back at the next phase boundary; only the ~30 coupling scalars stay live between
 * phases.  No MFMA (there is no dense contraction on this path), no inter-lane communication except the
 * turbine stage pass's ballot and the LDS transposes that turn the wave's 64 x 22 observation block into
 * coalesced row-major stores.  Plants are independent, so blocks never share data and the block -> XCD
 * placement cannot matter.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "npd_common.h"
/* diagnostic build only (-DNPB_STAMPS, tools/phase_stamps.py): lane 0 of every wave records s_memtime
 * at phase boundaries so the kernel's time can be attributed to phases on the GPU */
#ifdef NPB_STAMPS
#ifdef NPB_BUILD_F32
#define npb_stamp_buf npb32_stamp_buf
#define npb_debug_set_stamp_buffer npb32_debug_set_stamp_buffer
#endif
__device__ unsigned long long *npb_stamp_buf;
#define NPD_STAMP(k) do { if (threadIdx.x == 0 && npb_stamp_buf) npb_stamp_buf[(size_t)blockIdx.x * 32 + (k)] = __builtin_readcyclecounter(); } while (0)
extern "C" __attribute__((visibility("default"))) int npb_debug_set_stamp_buffer(unsigned long long *dev) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(npb_stamp_buf), &dev, sizeof(dev));
}
#define NPD_WAIT_ACC_STORE() do { if (threadIdx.x == 0 && npb_stamp_buf) npb_stamp_buf[(size_t)blockIdx.x * 32 + 23] = npd_wait_acc_s; } while (0)
#else
#define NPD_STAMP(k)
#define NPD_WAIT_ACC_STORE()
#endif
#include "npd_stage.h"
#include "npd_primary.h"
#include "npd_sg.h"
#include "npd_feedwater.h"
#include "npd_turbine.h"
#include "npd_condenser.h"
#include "npd_ph.h"
#include "npd_maintenance.h"
#include "npd_init.h"
#include "npd_reset.h"
#include "npb_kernels.h"

#ifndef NPD_OUT_STORE
#define NPD_OUT_STORE 1   /* how a narrow column of output members only is stored (npd_st_store_elide): 1 = plain store, never fetched (round 4: 65 536 plants 0.0884 -> 0.0838 ms, calibrated reads 269.8 -> 255.7 MB per launch, profiles/r4_out_store.txt) */
#endif
#define NPB_OBS_PAD 23 /* LDS row stride in doubles: 22 + 1 keeps the transpose at <= 2-way bank conflicts */

/* ---- segmented arena.  A handle of more than 45 056 plants keeps its arena in SEGMENTS of 16 384 plants: segment s
 * is the whole [column][plant] block of plants s * seg .. (s + 1) * seg - 1, so a launch over one segment sweeps one dense range of
 * memory (two handles of 32 768 plants step 4-6 % faster than one of 65 536 laid out column by column over all plants,
 * profiles/r3_shared_launches.txt; what makes the difference is not the launch per handle but the layout: one launch of the four-wave kernel
 * over a segmented arena is as fast, profiles/r3_segment_size.txt).  Every kernel takes
 * the arena pointer and "N", the column pitch in plants, whose upper 32 bits carry the segment size (0 = one segment); with
 *   address(column, plant p) = arena + s * seg * columns + column * seg + (p - s * seg) = [arena + s * seg * (columns - 1)] + column * seg + p
 * a kernel only has to move its base pointer once, by its plant's segment, and use the segment size as its pitch. */
#ifdef NPB_BUILD_F32
#define NPD_ARENA_COLS NPB_TOTAL_COL32
#else
#define NPD_ARENA_COLS NPB_TOTAL_COL64
#endif
#define NPD_SEGMENT(arena, N, p) do { const size_t seg__ = (size_t)(N) >> 32; (N) = (size_t)(N) & 0xffffffffu; \
    if (seg__) { (arena) += ((size_t)(p) / seg__) * seg__ * (size_t)(NPD_ARENA_COLS - 1); (N) = seg__; } } while (0)
/* launcher side: the pitch and the segment size out of a packed N */
#define NPD_NPAD(npad) ((size_t)(npad) & 0xffffffffu)
#define NPD_SEG_OF(npad) ((size_t)(npad) >> 32)

/* ---- section <-> arena movers outside the step kernel (init, observe, maintenance): plain global accesses.
 * A section struct is NF doubles (the last NO of them outputs) followed by NI int32s (include/npb_fields.h); in the
 * arena the NF - NO carried doubles take one column each and the narrow members share columns (npd_stage.h). */
__device__ __forceinline__ char *npd_gaddr(npd_real_t *arena, size_t N, size_t p, int col) {
  return (char *)(arena + (size_t)col * N + p);
}
template <int NF, int NO, int NI, typename S>
__device__ __forceinline__ void npd_load(S &s, const npd_real_t *__restrict__ arena, size_t N, size_t p, int col0) {
  double *d = reinterpret_cast<double *>(&s);
  constexpr int NC = NF - NO;
  npd_real_t *a = const_cast<npd_real_t *>(arena);
#pragma unroll
  for (int k = 0; k < NC; k++) d[k] = (double)*(const npd_real_t *)npd_gaddr(a, N, p, col0 + k);
#pragma unroll
  for (int j = 0; j < NO; j++) d[NC + j] = (double)*(const float *)(npd_gaddr(a, N, p, col0 + NC + j / NPD_NPC) + (j % NPD_NPC) * 4);
  int32_t *q = reinterpret_cast<int32_t *>(d + NF);
#pragma unroll
  for (int k = 0; k < NI; k++) q[k] = *(const int32_t *)(npd_gaddr(a, N, p, col0 + NC + (NO + k) / NPD_NPC) + ((NO + k) % NPD_NPC) * 4);
}
template <int NF, int NO, int NI, typename S>
__device__ __forceinline__ void npd_store(const S &s, npd_real_t *__restrict__ arena, size_t N, size_t p, int col0) {
  const double *d = reinterpret_cast<const double *>(&s);
  constexpr int NC = NF - NO;
#pragma unroll
  for (int k = 0; k < NC; k++) *(npd_real_t *)npd_gaddr(arena, N, p, col0 + k) = (npd_real_t)d[k];
#pragma unroll
  for (int j = 0; j < NO; j++) *(float *)(npd_gaddr(arena, N, p, col0 + NC + j / NPD_NPC) + (j % NPD_NPC) * 4) = (float)d[NC + j];
  const int32_t *q = reinterpret_cast<const int32_t *>(d + NF);
#pragma unroll
  for (int k = 0; k < NI; k++) *(int32_t *)(npd_gaddr(arena, N, p, col0 + NC + (NO + k) / NPD_NPC) + ((NO + k) % NPD_NPC) * 4) = q[k];
  if ((NO + NI) % NPD_NPC) *(int32_t *)(npd_gaddr(arena, N, p, col0 + NC + (NO + NI) / NPD_NPC) + 4) = 0; /* unused half of the last column */
}
#define NPD_LOAD(T, stype, s, inst) npd_load<NPB_##T##_NF64, NPB_##T##_NOUT, NPB_##T##_NI32, stype>(s, f64, N, p, NPD_SEC_COL(T, inst))
#define NPD_STORE(T, stype, s, inst) npd_store<NPB_##T##_NF64, NPB_##T##_NOUT, NPB_##T##_NI32, stype>(s, f64, N, p, NPD_SEC_COL(T, inst))
/* single member reads: fp64 member (carried or output) / int32 member */
template <int NC> __device__ __forceinline__ double npd_gread_real(const npd_real_t *arena, size_t N, size_t p, int col0, int idx) {
  npd_real_t *a = const_cast<npd_real_t *>(arena);
  if (idx < NC) return (double)*(const npd_real_t *)npd_gaddr(a, N, p, col0 + idx);
  const int j = idx - NC;
  return (double)*(const float *)(npd_gaddr(a, N, p, col0 + NC + j / NPD_NPC) + (j % NPD_NPC) * 4);
}
#define NPD_F64_COL(T, stype, member, inst) npd_gread_real<NPB_##T##_NCARRY>(f64, N, p, NPD_SEC_COL(T, inst), NPB_F64_SLOT(stype, member))
#define NPD_F64_COLK(T, stype, member, inst, k) npd_gread_real<NPB_##T##_NCARRY>(f64, N, p, NPD_SEC_COL(T, inst), NPB_F64_SLOT(stype, member) + (k))
#define NPD_I32_COL(T, stype, member, inst) \
  (*(const int32_t *)(npd_gaddr(const_cast<npd_real_t *>(f64), N, p, NPD_SEC_COL(T, inst) + NPB_##T##_NCARRY + (NPB_##T##_NOUT + NPB_I32_SLOT(stype, T, member)) / NPD_NPC) + \
                      ((NPB_##T##_NOUT + NPB_I32_SLOT(stype, T, member)) % NPD_NPC) * 4))

/* ---- step kernel: section stores through the pinned 32-bit-offset addressing of npd_stage.h.
 * bits of narrow member j of a section struct: outputs as float, then the int32 members */
template <int NF, int NO, int NI, typename S>
__device__ __forceinline__ uint32_t npd_narrow_bits(const S &s, int j) {
  const double *d = reinterpret_cast<const double *>(&s);
  const int32_t *q = reinterpret_cast<const int32_t *>(d + NF);
  if (j < NO) return __float_as_uint((float)d[NF - NO + j]);
  if (j < NO + NI) return (uint32_t)q[j - NO];
  return 0u;
}
/* store the narrow column c of a section instance (col = its arena column) from NPD_NPC 32-bit words */
#define NPD_STORE_REAL(col, v) npd_store_real<SM>(st, (col), (v))
template <int SM = 0>
__device__ __forceinline__ void npd_st_store_narrow(const npd_stage_t &st, int col, uint32_t w0, uint32_t w1) {
#ifdef NPB_BUILD_F32
  npd_gstore<SM == 2>(NPD_NP(uint32_t, col, 0), w0);
#else
  typedef uint32_t npd_u32x2 __attribute__((ext_vector_type(2)));
  npd_u32x2 v; v.x = w0; v.y = w1;
  if constexpr (SM >= 1) npd_store8<SM == 2>(st, (uint32_t)col, v);
  else npd_gstore<false>(NPD_RPO(npd_u32x2, col, st.laner), v);
#endif
}
template <int NF, int NO, int NI, typename S, int SM = 0>
__device__ __forceinline__ void npd_st_store(const S &s, const npd_stage_t &st, int col0) {
  const double *d = reinterpret_cast<const double *>(&s);
  constexpr int NC = NF - NO, NNC = (NO + NI + NPD_NPC - 1) / NPD_NPC;
#pragma unroll
  for (int k = 0; k < NC; k++) NPD_STORE_REAL(col0 + k, d[k]);
#pragma unroll
  for (int c = 0; c < NNC; c++)
    npd_st_store_narrow<SM>(st, col0 + NC + c, npd_narrow_bits<NF, NO, NI>(s, c * NPD_NPC), npd_narrow_bits<NF, NO, NI>(s, c * NPD_NPC + 1));
}
template <int NF, int NO, int NI, int SID, typename S>
__device__ __forceinline__ void npd_st_load(S &s, const npd_stage_t &st, int col0) {
  double *d = reinterpret_cast<double *>(&s);
  constexpr int NC = NF - NO;
#pragma unroll
  for (int k = 0; k < NC; k++) d[k] = (double)*NPD_RP(col0 + k);
#pragma unroll
  for (int j = 0; j < NO; j++) d[NC + j] = (double)*NPD_NP(const float, col0 + NC + j / NPD_NPC, j % NPD_NPC);
  int32_t *q = reinterpret_cast<int32_t *>(d + NF);
#pragma unroll
  for (int k = 0; k < NI; k++) q[k] = *NPD_NP(const int32_t, col0 + NC + (NO + k) / NPD_NPC, (NO + k) % NPD_NPC);
  NPD_PROBE_STRUCT(SID, NF, NI, d, q);
}
#define NPD_ST_LOAD(T, stype, s, inst) \
  npd_st_load<NPB_##T##_NF64, NPB_##T##_NOUT, NPB_##T##_NI32, NPB_##T##_F64_BASE, stype>(s, st, NPD_SEC_COL(T, inst))
#define NPD_ST_STORE(T, stype, s, inst) \
  npd_st_store<NPB_##T##_NF64, NPB_##T##_NOUT, NPB_##T##_NI32, stype, NPD_SM>(s, st, NPD_SEC_COL(T, inst))
/* only the narrow columns of a section (its outputs and int32 members) */
template <int NF, int NO, int NI, typename S, int SM = 0>
__device__ __forceinline__ void npd_st_store_narrow_cols(const S &s, const npd_stage_t &st, int col0) {
  constexpr int NC = NF - NO, NNC = (NO + NI + NPD_NPC - 1) / NPD_NPC;
#pragma unroll
  for (int c = 0; c < NNC; c++)
    npd_st_store_narrow<SM>(st, col0 + NC + c, npd_narrow_bits<NF, NO, NI>(s, c * NPD_NPC), npd_narrow_bits<NF, NO, NI>(s, c * NPD_NPC + 1));
}
#define NPD_ST_STORE_NARROW(T, stype, s, inst) \
  npd_st_store_narrow_cols<NPB_##T##_NF64, NPB_##T##_NOUT, NPB_##T##_NI32, stype, NPD_SM>(s, st, NPD_SEC_COL(T, inst))
/* ---- unchanged-column elision.  Measured on the bench workload (and on a reactor-heat-source batch): about a
 * quarter of the carried columns keep their exact bits over a step for every plant of a wave -- flags, status
 * codes, protection timers at rest, pump pressures and cavitation state in normal operation, the spare pump,
 * the kinetics block under the constant heat source.  A global store occupies the lone wave of a SIMD for its
 * transfer time, a bitwise compare + wave ballot costs three instructions, so the
 * members named in the masks below are stored only if some lane's bits changed.  Which members are listed is a
 * performance choice only: the compare is on the bit patterns, so the arena always ends up with exactly the
 * bits a plain store would have written.  "old" is the copy of the section as it was staged in. */
/* bit pattern of a value as it is stored (fp32 storage: after rounding to fp32) */
__device__ __forceinline__ long long npd_real_bits(double v) {
#ifdef NPB_BUILD_F32
  return (long long)__float_as_int((float)v);
#else
  return __double_as_longlong(v);
#endif
}
/* SKIP0 .. SKIP1: carried members that were not loaded this step and must not be stored (wave-uniform `skip`) */
template <int NF, int NO, int NI, int SKIP0 = 0, int SKIP1 = 0, typename S, int SM = 0>
__device__ __forceinline__ void npd_st_store_elide(const S &s, const S &old, const npd_stage_t &st, int col0, uint64_t fmask, bool skip = false) {
  const double *d = reinterpret_cast<const double *>(&s), *od = reinterpret_cast<const double *>(&old);
  constexpr int NC = NF - NO, NNC = (NO + NI + NPD_NPC - 1) / NPD_NPC;
#ifdef NPB_PROBE
  fmask = 0; /* the liveness probe looks at the physics only, not at the elision's old copies */
#endif
  if (SKIP1 > SKIP0 && !skip) {
#pragma unroll
    for (int k = SKIP0; k < SKIP1; k++) {
      if ((fmask >> k) & 1) {
        if (__builtin_amdgcn_ballot_w64(npd_real_bits(d[k]) != npd_real_bits(od[k])) != 0) NPD_STORE_REAL(col0 + k, d[k]);
      } else {
        NPD_STORE_REAL(col0 + k, d[k]);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < NC; k++) {
    if (k >= SKIP0 && k < SKIP1) continue;
    if ((fmask >> k) & 1) {
      if (__builtin_amdgcn_ballot_w64(npd_real_bits(d[k]) != npd_real_bits(od[k])) != 0) NPD_STORE_REAL(col0 + k, d[k]);
    } else {
      NPD_STORE_REAL(col0 + k, d[k]);
    }
  }
  /* narrow columns (outputs as float, flags, status codes, counters): compared -- except the columns that hold nothing but
   * OUTPUT members (the first NO / NPD_NPC of them): the step never reads an output, so comparing its old bits is the only
   * reason the column would be fetched at all (round 3's counters: reads 1.16x algorithmic, 284 B per plant of it these).
   * NPD_OUT_STORE: 0 = compare like the rest (round 3), 1 = plain store, no compare, no load, 2 = the same, non-temporal */
#pragma unroll
  for (int c = 0; c < NNC; c++) {
    const uint32_t w0 = npd_narrow_bits<NF, NO, NI>(s, c * NPD_NPC), w1 = npd_narrow_bits<NF, NO, NI>(s, c * NPD_NPC + 1);
#ifdef NPB_PROBE
    npd_st_store_narrow<SM>(st, col0 + NC + c, w0, w1);
#else
    if (NPD_OUT_STORE != 0 && (c + 1) * NPD_NPC <= NO) {
      npd_st_store_narrow<(NPD_OUT_STORE == 2 ? 2 : SM)>(st, col0 + NC + c, w0, w1);
      continue;
    }
    const uint32_t o0 = npd_narrow_bits<NF, NO, NI>(old, c * NPD_NPC), o1 = npd_narrow_bits<NF, NO, NI>(old, c * NPD_NPC + 1);
    if (__builtin_amdgcn_ballot_w64(NPD_NPC == 2 ? ((w0 != o0) | (w1 != o1)) : (w0 != o0)) != 0) npd_st_store_narrow<SM>(st, col0 + NC + c, w0, w1);
#endif
  }
}
#define NPD_ST_STORE_ELIDE(T, stype, s, old, inst) \
  npd_st_store_elide<NPB_##T##_NF64, NPB_##T##_NOUT, NPB_##T##_NI32, 0, 0, stype, NPD_SM>(s, old, st, NPD_SEC_COL(T, inst), NPD_ELIDE_##T##_F)
/* the primary section: the point-kinetics members move only under ReactorHeatSource (npb_fields.h, NPB_PRIM_NKIN) */
#define NPD_PRIM_KIN0 (NPB_PRIM_NCARRY - NPB_PRIM_NKIN)
#define NPD_ST_STORE_ELIDE_PRIM(s, old) \
  npd_st_store_elide<NPB_PRIM_NF64, NPB_PRIM_NOUT, NPB_PRIM_NI32, NPD_PRIM_KIN0, NPB_PRIM_NCARRY, npb_prim_t, NPD_SM>( \
      s, old, st, NPD_SEC_COL(PRIM, 0), NPD_ELIDE_PRIM_F, !kinetics)
#define NPD_FB(stype, m) (1ull << NPB_F64_SLOT(stype, m))
#define NPD_FBN(stype, m, n) ((((1ull << (n)) - 1)) << NPB_F64_SLOT(stype, m))
static constexpr uint64_t NPD_ELIDE_PRIM_F =
    NPD_FB(npb_prim_t, neutron_flux) | NPD_FB(npb_prim_t, reactivity) | NPD_FBN(npb_prim_t, precursors, 6) |
    NPD_FB(npb_prim_t, coolant_flow_rate) | NPD_FB(npb_prim_t, coolant_void_fraction) | NPD_FB(npb_prim_t, steam_pressure) |
    NPD_FB(npb_prim_t, feedwater_flow_rate) | NPD_FB(npb_prim_t, control_rod_position) | NPD_FB(npb_prim_t, steam_valve_position) |
    NPD_FB(npb_prim_t, boron_concentration) | NPD_FB(npb_prim_t, xenon_concentration) | NPD_FB(npb_prim_t, iodine_concentration) |
    NPD_FB(npb_prim_t, samarium_concentration) | NPD_FB(npb_prim_t, burnable_poison_worth) | NPD_FB(npb_prim_t, fuel_burnup) |
    NPD_FB(npb_prim_t, total_reactivity_pcm) | NPD_FB(npb_prim_t, hs_filtered_noise_mw);
static constexpr uint64_t NPD_ELIDE_SG_F = NPD_FB(npb_sg_t, water_level);
static constexpr uint64_t NPD_ELIDE_PUMP_F =
    NPD_FB(npb_pump_t, suction_pressure) | NPD_FB(npb_pump_t, discharge_pressure) | NPD_FB(npb_pump_t, npsh_available) |
    NPD_FB(npb_pump_t, differential_pressure) | NPD_FB(npb_pump_t, cavitation_intensity) | NPD_FB(npb_pump_t, cavitation_damage) |
    NPD_FB(npb_pump_t, cavitation_time) | NPD_FB(npb_pump_t, head_degradation) | NPD_FB(npb_pump_t, seal_leakage_rate) |
    /* the spare pump */ NPD_FB(npb_pump_t, speed_percent) | NPD_FB(npb_pump_t, speed_setpoint) | NPD_FB(npb_pump_t, flow_rate) |
    NPD_FB(npb_pump_t, power_consumption) | NPD_FB(npb_pump_t, flow_demand) | NPD_FB(npb_pump_t, motor_temperature) |
    NPD_FB(npb_pump_t, vibration_level) | NPD_FB(npb_pump_t, wear_motor_bearings) | NPD_FB(npb_pump_t, wear_pump_bearings) |
    NPD_FB(npb_pump_t, wear_thrust_bearing) | NPD_FB(npb_pump_t, wear_coupling_system) | NPD_FB(npb_pump_t, flow_degradation) |
    NPD_FB(npb_pump_t, vibration_increase);
static constexpr uint64_t NPD_ELIDE_FW_F =
    NPD_FBN(npb_fw_t, previous_level_errors, 3) | NPD_FB(npb_fw_t, cav_accumulated_damage) | NPD_FB(npb_fw_t, cav_time_in_cavitation) |
    NPD_FB(npb_fw_t, npsh_low_low_timer) | NPD_FB(npb_fw_t, timer_low_flow) | NPD_FB(npb_fw_t, timer_high_flow) |
    NPD_FB(npb_fw_t, timer_bearing_temp) | NPD_FB(npb_fw_t, timer_motor_temp) | NPD_FB(npb_fw_t, timer_vibration) |
    NPD_FB(npb_fw_t, quality_integral_error);
static constexpr uint64_t NPD_ELIDE_TURB_F =
    NPD_FB(npb_turb_t, rotor_speed) | NPD_FB(npb_turb_t, thermal_bow) | NPD_FBN(npb_turb_t, bearing_metal_temp, 4) |
    NPD_FBN(npb_turb_t, bearing_wear_factor, 4) | NPD_FB(npb_turb_t, timer_overspeed) | NPD_FB(npb_turb_t, timer_vibration) |
    NPD_FB(npb_turb_t, timer_bearing_temp) | NPD_FB(npb_turb_t, total_power_output) | NPD_FB(npb_turb_t, vibration_displacement);
static constexpr uint64_t NPD_ELIDE_CHEM_F =
    NPD_FB(npb_chem_t, dissolved_oxygen) | NPD_FB(npb_chem_t, corrosion_inhibitor_level) | NPD_FB(npb_chem_t, treatment_efficiency) |
    NPD_FB(npb_chem_t, chlorine_residual) | NPD_FB(npb_chem_t, antiscalant_concentration) | NPD_FB(npb_chem_t, water_aggressiveness);
static constexpr uint64_t NPD_ELIDE_PH_F = NPD_FB(npb_ph_t, morpholine_tank_level) | NPD_FB(npb_ph_t, pending_morpholine_dose) |
                                           NPD_FB(npb_ph_t, integral_sum);
static constexpr uint64_t NPD_ELIDE_COND_F =
    NPD_FB(npb_cond_t, vibration_damage) | NPD_FB(npb_cond_t, condenser_pressure) | NPD_FB(npb_cond_t, air_partial_pressure) |
    NPD_FB(npb_cond_t, air_mass_in_condenser) | /* the idle ejector */ NPD_FBN(npb_cond_t, ej_nozzle_fouling, 2) |
    NPD_FBN(npb_cond_t, ej_diffuser_fouling, 2) | NPD_FBN(npb_cond_t, ej_nozzle_erosion, 2);

/* single carried column, same rule (output / int32 members are stored as whole narrow columns, see the tail) */
#define NPD_ST_F64_ELIDE(T, stype, member, inst, k, newv, oldv) do { \
    const double nv__ = (newv); \
    if (__builtin_amdgcn_ballot_w64(npd_real_bits(nv__) != npd_real_bits(oldv)) != 0) NPD_ST_F64(T, stype, member, inst, k) = (npd_real_t)nv__; } while (0)
#define NPD_ST_F64(T, stype, member, inst, k) \
  (*NPD_RP(NPD_SEC_COL(T, inst) + npd_carried_slot<NPB_##T##_NCARRY>(NPB_F64_SLOT(stype, member) + (k))))
template <int NC> __device__ __forceinline__ constexpr int npd_carried_slot(int idx) { return idx; }

/* wave-cooperative store of a [64][W] block held one row per lane into row-major global memory (non-temporal: see
 * the reward / done / flags stores) */
template <int W>
__device__ __forceinline__ void npd_store_rows(const double *row, double *__restrict__ out, double *lds,
                                               size_t block_base, size_t n_valid) {
  const int lane = threadIdx.x;
#pragma unroll
  for (int j = 0; j < W; j++) lds[lane * NPB_OBS_PAD + j] = row[j];
  NPD_LDS_DRAIN(); /* the block is one wave: LDS ordering inside a wave needs no barrier (and no vmcnt drain) */
#pragma unroll
  for (int k = 0; k < W; k++) {
    int idx = k * NPB_WAVE + lane;
    int r = idx / W, c = idx % W;
    if (block_base + r < n_valid) __builtin_nontemporal_store(lds[r * NPB_OBS_PAD + c], &out[block_base * W + idx]);
  }
  NPD_LDS_DRAIN();
}

__device__ __forceinline__ double npd_sel3(int i, double a0, double a1, double a2) { return (i == 0) ? a0 : ((i == 1) ? a1 : a2); }

/* get_observation  sim.py:290-333 (primary part) */
__device__ __forceinline__ void npd_obs_primary(const npb_prim_t &s, double *obs) {
  obs[0] = s.neutron_flux / 1e12;
  obs[1] = s.fuel_temperature / 1000;
  obs[2] = s.coolant_temperature / 300;
  obs[3] = s.coolant_pressure / 20;
  obs[4] = s.coolant_flow_rate / 50000;
  obs[5] = s.steam_temperature / 300;
  obs[6] = s.steam_pressure / 10;
  obs[7] = s.steam_flow_rate / 3000;
  obs[8] = s.control_rod_position / 100;
  obs[9] = s.steam_valve_position / 100;
  obs[10] = s.power_level / 100;
  obs[11] = (double)(s.scram_status != 0);
}

/* info["reactivity_components"] (sim.py:205): the second block of the info buffer, only for a caller that asked
 * (params.info_reactivity_components) under the reactor heat source -- include/npb.h NPB_RHO_* */
__device__ __forceinline__ void npd_store_reactivity_components(const npb_params_t &P, const double *rho, double *__restrict__ info_out,
                                                                int n_plants, size_t p) {
  if (P.info_reactivity_components && P.heat_source == NPB_HEAT_REACTOR && info_out && p < (size_t)n_plants) {
    double *out = info_out + (size_t)n_plants * NPB_INFO_DIM + p * NPB_INFO_NRHO;
#pragma unroll
    for (int k = 0; k < NPB_INFO_NRHO; k++) out[k] = rho[k];
  }
}

/* NuclearPlantSimulator(enable_secondary=False).step  sim.py:141-151,186-206,253-258: the primary side alone (no coupling,
 * no secondary update, no feedback), twelve observations (the other ten columns of the obs row are written as 0), the
 * base reward, and NaN in the info columns whose keys the reference's dict then lacks.  A small kernel of its own: nothing
 * here is worth the staging pipeline. */
__global__ __launch_bounds__(NPB_WAVE) void npb_step_primary_kernel(
    npb_params_t P, int n_plants, size_t N, npd_real_t *__restrict__ f64,
    const int32_t *__restrict__ action, const double *__restrict__ magnitude, const double *__restrict__ setpoint,
    const double *__restrict__ noise_z, double *__restrict__ obs_out, double *__restrict__ reward_out,
    uint8_t *__restrict__ done_out, uint32_t *__restrict__ trip_out, double *__restrict__ info_out) {
  __shared__ double lds[NPB_WAVE * NPB_OBS_PAD];
  const size_t block_base = (size_t)blockIdx.x * NPB_WAVE;
  NPD_SEGMENT(f64, N, block_base);
  const size_t p = block_base + threadIdx.x;
  const bool live = p < (size_t)n_plants;
  npd_inputs_t in;
  in.action = (live && action) ? action[p] : 8;
  in.magnitude = (live && magnitude) ? magnitude[p] : 1.0;
  in.power_setpoint = (live && setpoint) ? setpoint[p] : NAN;
  in.noise_z = (live && noise_z) ? noise_z[p] : 0.0;
  in.cooling_water_temp = NAN;
  npb_prim_t s;
  NPD_LOAD(PRIM, npb_prim_t, s, 0);
  if (P.heat_source != NPB_HEAT_EXTERNAL && !isnan(in.power_setpoint)) s.hs_setpoint_percent = npd_clip(in.power_setpoint, 0.0, 150.0);
  int nan_reset;
  double rho[NPB_INFO_NRHO];
  const int scram_fired = npd_primary_update(&s, &P, &in, &nan_reset, rho);
  npd_store_reactivity_components(P, rho, info_out, n_plants, p);
  s.sim_time += P.dt;
  NPD_STORE(PRIM, npb_prim_t, s, 0);
  double obs[NPB_OBS_DIM], info[NPB_INFO_DIM];
  npd_obs_primary(s, obs);
#pragma unroll
  for (int k = 12; k < NPB_OBS_DIM; k++) obs[k] = 0.0;
  /* calculate_reward(None)  sim.py:503-519 */
  double power_reward = -fabs(s.power_level - 100) / 100;
  double temp_penalty = 0, pressure_penalty = 0;
  if (s.fuel_temperature > 800) temp_penalty = -(s.fuel_temperature - 800) / 100;
  if (s.coolant_pressure > 16) pressure_penalty = -(s.coolant_pressure - 16);
  double scram_penalty = s.scram_status ? -100 : 0;
  if (live) {
    if (reward_out) reward_out[p] = power_reward + temp_penalty + pressure_penalty + scram_penalty;
    if (done_out) done_out[p] = (uint8_t)scram_fired;
    if (trip_out) trip_out[p] = (s.scram_status ? NPB_TRIP_SCRAM : 0u) | (scram_fired ? NPB_TRIP_SCRAM_FIRED : 0u) | (nan_reset ? NPB_TRIP_NAN_RESET : 0u);
  }
  if (obs_out) npd_store_rows<NPB_OBS_DIM>(obs, obs_out, lds, block_base, (size_t)n_plants);
  if (info_out) {
#pragma unroll
    for (int k = 0; k < NPB_INFO_DIM; k++) info[k] = NAN;
    info[NPB_INFO_THERMAL_POWER] = s.thermal_power_mw;
    info[NPB_INFO_REACTIVITY_PCM] = s.total_reactivity_pcm;
    info[NPB_INFO_TIME] = s.sim_time;
    npd_store_rows<NPB_INFO_DIM>(info, info_out, lds, block_base, (size_t)n_plants);
  }
}

/* automatic maintenance after a step (params.maint_enabled): AutoMaintenanceSystem.update, then the state manager's
 * threshold scan with work-order creation (npd_maintenance.h), for the four feedwater pumps -- inside the step kernels:
 *   the screen   what nearly every step of nearly every plant ends with is "nothing new".  The pump phase answers "is any
 *     threshold of this pump crossed outside its cooldown" from the registers it has just updated, the primary phase moves
 *     last_check_time where a check fell due with no order open (npd_maintenance.h).
 *   the rule     a wave that did find something calls npd_maint_rule_for_wave before it ends: first a proper look -- the
 *     rows' real comparisons on the stored state, then that pump's 16 last-violation stamps: a crossed threshold inside
 *     its cooldown is no work -- and only then the full rule for its 64 plants: work orders, the orchestrator, the
 *     thirteen handlers.  A real function call (noinline), so that its registers and its scratch are its own: rare, so it is
 *     written for clarity, not for registers, and the step kernels' own allocation does not see it.  Its constants -- the
 *     parameters, the table -- come from device memory (npd_maint_rule_consts_t, uploaded by npb_step when they change).
 * No second launch: a separate rule kernel cost 4-5 us per step just to find nothing flagged (its code and arguments are
 * cold behind the step kernel's 500 MB sweep), more than the whole screen.  npb_maint_kernel below is the same rule for the
 * modes whose step kernels do not step the pumps. */
#define NPD_MP_COL(inst, member, k) (NPD_SEC_COL(MPUMP, inst) + NPB_F64_SLOT(npb_mpump_t, member) + (k))
#define NPD_MP_LOAD(inst, member, count) do { _Pragma("unroll") for (int q__ = 0; q__ < (count); q__++) \
    mp.member[q__] = (double)*(const npd_real_t *)npd_gaddr(f64, N, p, NPD_MP_COL(inst, member, q__)); } while (0)
#define NPD_MP_STORE(inst, member, count) do { _Pragma("unroll") for (int q__ = 0; q__ < (count); q__++) \
    *(npd_real_t *)npd_gaddr(f64, N, p, NPD_MP_COL(inst, member, q__)) = (npd_real_t)mp.member[q__]; } while (0)
/* the proper look's view of the table: scan membership folded into the comparison masks on the host side */
struct npd_maint_screen_t {
  double threshold[NPB_MAINT_NPARAM];
  double cooldown_minutes[NPB_MAINT_NPARAM];
  uint32_t want_gt, want_lt, want_eq, want_near, want_far;    /* bit q: row q fires on value > / < / == threshold, |value - threshold| < / >= 0.001 */
};
struct npd_maint_rule_consts_t { npb_params_t P; npb_maint_table_t T; npd_maint_screen_t S; };
/* does pump k of this lane's plant have a crossed threshold outside its cooldown?  (StateManager._check_maintenance_thresholds up
 * to the point where a violation is recorded, state_manager.py:1307-1369) */
__device__ __forceinline__ bool npd_maint_second_look(const npd_maint_screen_t &S, const npd_real_t *f64c, size_t N, size_t p, int k, double t) {
  npd_real_t *f64 = const_cast<npd_real_t *>(f64c);
  npb_pump_t pm;      /* only the members npd_maint_values reads are loaded */
#define NPD_PM(member) pm.member = NPD_F64_COL(PUMP, npb_pump_t, member, k)
  NPD_PM(oil_level); NPD_PM(oil_contamination); NPD_PM(lubrication_effectiveness); NPD_PM(wear_impeller); NPD_PM(cavitation_damage);
  NPD_PM(cavitation_intensity); NPD_PM(npsh_available); NPD_PM(wear_motor_bearings); NPD_PM(wear_pump_bearings); NPD_PM(wear_thrust_bearing);
  NPD_PM(wear_mechanical_seals); NPD_PM(vibration_level); NPD_PM(oil_temperature); NPD_PM(motor_temperature); NPD_PM(seal_leakage_rate);
#undef NPD_PM
  double values[NPB_MAINT_NPARAM];
  npd_maint_values(&pm, values);
  uint32_t hits = 0;
#pragma unroll
  for (int q = 0; q < NPB_MAINT_NPARAM; q++) {
    const double v = values[q], thr = S.threshold[q];
    const bool near_eq = fabs(v - thr) < 0.001;                                  /* _check_threshold_condition */
    const bool hit = ((((S.want_gt >> q) & 1u) != 0) & (v > thr)) | ((((S.want_lt >> q) & 1u) != 0) & (v < thr)) |
                     ((((S.want_eq >> q) & 1u) != 0) & (v == thr)) | ((((S.want_near >> q) & 1u) != 0) & near_eq) |
                     ((((S.want_far >> q) & 1u) != 0) & !near_eq);
    hits |= (uint32_t)hit << q;
  }
  bool work = false;
  if (__any(hits != 0)) {
#pragma unroll
    for (int q = 0; q < NPB_MAINT_NPARAM; q++) {
      const double lv = (double)*(const npd_real_t *)npd_gaddr(f64, N, p, NPD_MP_COL(k, last_violation_time, q));
      const bool cooling = (lv >= 0.0) & (t - lv < S.cooldown_minutes[q]);        /* _is_threshold_in_cooldown */
      work |= (((hits >> q) & 1u) != 0) & !cooling;
    }
  }
  return work;
}
/* the cooldown cache of the step kernels' screen (npd_maintenance.h) for the four pumps of this lane's plant, from the stamps as
 * they are now: whenever the rule has looked at a wave */
__device__ __forceinline__ void npd_maint_refresh_cache(const npd_maint_screen_t &S, npd_u32x4 *cache_entry, const npd_real_t *f64c, size_t N, size_t p, double t) {
  npd_real_t *f64 = const_cast<npd_real_t *>(f64c);
  const uint32_t scan_mask = S.want_gt | S.want_lt | S.want_eq | S.want_near | S.want_far;
#pragma unroll 1
  for (int k = 0; k < NPB_NUM_PUMPS; k++) {
    double lv[NPB_MAINT_NPARAM];
#pragma unroll
    for (int q = 0; q < NPB_MAINT_NPARAM; q++) lv[q] = (double)*(const npd_real_t *)npd_gaddr(f64, N, p, NPD_MP_COL(k, last_violation_time, q));
    uint32_t mask; float until;
    npd_maint_cache_entry(lv, S.cooldown_minutes, scan_mask, t, &mask, &until);
    uint32_t *e = (uint32_t *)(cache_entry + p * 2) + 2 * k;
    e[0] = mask; e[1] = __float_as_uint(until);
  }
}
/* one wave = the 64 plants p - lane .. p - lane + 63, whose step (all its stores) is complete.  hit_bits: bit k = the screen
 * flagged pump k for some lane; due_with_orders: some lane's check falls on open orders (wave-uniform both) */
/* WHO: one instantiation per calling kernel, so that each inherits its caller's register budget (the build of the two-wave
 * kernel that shares a SIMD between two waves must not be dragged to one wave per SIMD by a callee with the whole file) */
template <int WHO>
__device__ __attribute__((noinline)) void npd_maint_rule_for_wave(const npd_maint_rule_consts_t *RC, npd_maint_cache_t MC, npd_real_t *f64, size_t N, size_t p,
                                                                 unsigned hit_bits, unsigned due_with_orders) {
  const npb_params_t &P = RC->P; const npb_maint_table_t &T = RC->T; const npd_maint_screen_t &S = RC->S;
  const double t = NPD_F64_COL(PRIM, npb_prim_t, sim_time, 0);
  /* look properly, pump by pump, before anything heavy is fetched: the screen's bit k says "something may be new at pump k for some
   * plant of the wave"; the second look says whether the scan below would record a violation there (npd_maint_scan_pump returns
   * without touching anything when no threshold is crossed outside its cooldown), so a pump without one is not scanned at all --
   * the pump and mpump sections are ~140 columns per pump, the second look 15 */
  unsigned scan_bits = 0;
#pragma unroll 1
  for (int k = 0; k < NPB_NUM_PUMPS; k++) {
    if (((hit_bits >> k) & 1u) && __any(npd_maint_second_look(S, f64, N, p, k, t))) scan_bits |= 1u << k;
  }
  if (!due_with_orders && !scan_bits) { npd_maint_refresh_cache(S, MC.entry, f64, N, p, t); return; }
  npb_maint_t m;
  NPD_LOAD(MAINT, npb_maint_t, m, 0);
  int dirty = 0, executed = -1;      /* executed: the pump this lane's plant has just maintained (its readings have moved: scanned in any case) */
  /* ---- AutoMaintenanceSystem.update: one due order, the earliest created, is carried out */
  if (npd_maint_check_due(&m, &P, t)) {
    dirty = 1;
    if (m.work_orders_created > m.maintenance_actions_performed) {      /* some order is open */
      double best = 0.0; int pick = -1, pick_action = -1;
#pragma unroll 1
      for (int k = 0; k < NPB_NUM_PUMPS; k++) {
        npb_mpump_t mp;
        NPD_MP_LOAD(k, wo_order, NPB_MAINT_NACT); NPD_MP_LOAD(k, wo_planned_start, NPB_MAINT_NACT);
        int a; const double o = npd_maint_first_due(&mp, t, &a);
        if (o > 0.0 && (best == 0.0 || o < best)) { best = o; pick = k; pick_action = a; }
      }
      if (pick >= 0) {
        npb_mpump_t mp;
        NPD_MP_LOAD(pick, wo_order, NPB_MAINT_NACT); NPD_MP_LOAD(pick, wo_planned_start, NPB_MAINT_NACT);
        mp.wo_bearing = (double)*(const npd_real_t *)npd_gaddr(f64, N, p, NPD_MP_COL(pick, wo_bearing, 0));
        const int bearing = npd_maint_close_order(&mp, &m, pick_action);
        NPD_MP_STORE(pick, wo_order, NPB_MAINT_NACT); NPD_MP_STORE(pick, wo_planned_start, NPB_MAINT_NACT);
        *(npd_real_t *)npd_gaddr(f64, N, p, NPD_MP_COL(pick, wo_bearing, 0)) = (npd_real_t)mp.wo_bearing;
        npb_pump_t pm;
        NPD_LOAD(PUMP, npb_pump_t, pm, pick);
        npd_maint_execute(&pm, &P, pick_action, bearing);
        NPD_STORE(PUMP, npb_pump_t, pm, pick);
        executed = pick;
        if (MC.diag && ((NPD_MA_HANDLER_MASK >> pick_action) & 1u)) {    /* pump_lubrication.py:642-643, 1636-1637: the flags of this step's state-log row (action types the dispatcher knows) */
          MC.diag[(size_t)(NPB_DIAG_PUMP_MAINTENANCE_OCCURRED + pick) * MC.diag_pitch + p] = 1.0;
          if (pick_action == NPB_MA_OIL_TOP_OFF) MC.diag[(size_t)(NPB_DIAG_PUMP_OIL_TOP_OFF_OCCURRED + pick) * MC.diag_pitch + p] = 1.0;
          MC.diag[(size_t)(NPB_DIAG_PUMP_MAINTENANCE_ACTION + pick) * MC.diag_pitch + p] = (double)(pick_action + 1);
        }
      }
    }
  }
  /* ---- StateManager.collect_states: threshold scan, one orchestrated event per pump (after the work above, as the
   * reference orders it) */
#pragma unroll 1
  for (int k = 0; k < NPB_NUM_PUMPS; k++) {
    if (!((scan_bits >> k) & 1u) && !__any(executed == k)) continue;
    npb_pump_t pm;
    NPD_LOAD(PUMP, npb_pump_t, pm, k);
    npb_mpump_t mp;
    NPD_LOAD(MPUMP, npb_mpump_t, mp, k);
    if (npd_maint_scan_pump(&mp, &m, &P, &T, &pm, t)) {
      dirty = 1;
      NPD_STORE(MPUMP, npb_mpump_t, mp, k);
    }
  }
  if (dirty) {
    NPD_STORE(MAINT, npb_maint_t, m, 0);
    if (MC.counts && p < (size_t)MC.n_plants) MC.counts[p] = m.maintenance_actions_performed;
  }
  npd_maint_refresh_cache(S, MC.entry, f64, N, p, t);
}
/* the same rule as a launch of its own, every wave looked at in full: for the modes whose step kernels do not step the pumps
 * (primary-only, primary + steam generators), where nothing has screened anything */
__global__ __launch_bounds__(NPB_WAVE) void npb_maint_kernel(const npd_maint_rule_consts_t *RC, npd_maint_cache_t MC, size_t N, npd_real_t *__restrict__ f64) {
  NPD_SEGMENT(f64, N, (size_t)blockIdx.x * NPB_WAVE);
  npd_maint_rule_for_wave<0>(RC, MC, f64, N, (size_t)blockIdx.x * NPB_WAVE + threadIdx.x, 0xFu, 1u);
}

#define NPD_SM 1          /* store mode of the kernels below (npd_store_real): the one-wave kernel's own 8-byte stores ... */
#define NPD_STEP1_KERNEL npb_step_kernel
#define NPD_STEP1_MAINT 0
#define NPD_STEP1_WHO 1
#include "npd_step1.h"
#undef NPD_STEP1_KERNEL
#undef NPD_STEP1_MAINT
/* the same with the automatic maintenance compiled in (what npb_step launches when params.maint_enabled) */
#define NPD_STEP1_KERNEL npb_step_maint_kernel
#define NPD_STEP1_MAINT 1
#include "npd_step1.h"
#undef NPD_STEP1_KERNEL
/* the same step with the step-internal diagnostics written (npb_set_diagnostics): for state logging, not for throughput */
#define NPD_STEP1_KERNEL npb_step_diag_kernel
#undef NPD_STEP1_WHO
#define NPD_STEP1_WHO 2
#define NPD_STEP1_DIAG
#include "npd_step1.h"
#undef NPD_STEP1_DIAG
#undef NPD_STEP1_KERNEL
#undef NPD_STEP1_MAINT
/* ... with the non-temporal bit in this one, for batches whose sweep is far past the 256 MB Infinity Cache: nothing a step writes is still
 * cached when the next step reads it, and stores that do not allocate leave the caches to the loads (131 072 plants: 0.218 ->
 * 0.196 ms; at 65 536, where a tenth of the arena still hits, they cost 4 %: profiles/r2_ab_streaming_state_stores.txt) */
#undef NPD_SM
#define NPD_SM 2
#define NPD_STEP1_KERNEL npb_step_nt_kernel
#define NPD_STEP1_MAINT 0
#undef NPD_STEP1_WHO
#define NPD_STEP1_WHO 3
#include "npd_step1.h"
#undef NPD_STEP1_KERNEL
#undef NPD_STEP1_MAINT
#define NPD_STEP1_KERNEL npb_step_nt_maint_kernel
#define NPD_STEP1_MAINT 1
#include "npd_step1.h"
#undef NPD_STEP1_KERNEL
#undef NPD_STEP1_MAINT
#undef NPD_SM
#define NPD_SM 0          /* ... and the two-wave kernels leave theirs to the compiler */

#include "npd_step2.h"
#include "npd_step4.h"

/* get_observation() without stepping (after reset / set_field): sim.py:290-333 */
__global__ __launch_bounds__(NPB_WAVE) void npb_observe_kernel(int mode, int n_plants, size_t N, const npd_real_t *__restrict__ f64c,
                                                               double *__restrict__ obs_out) {
  __shared__ double lds[NPB_WAVE * NPB_OBS_PAD];
  npd_real_t *f64 = const_cast<npd_real_t *>(f64c);
  const size_t block_base = (size_t)blockIdx.x * NPB_WAVE;
  NPD_SEGMENT(f64, N, block_base);
  const size_t p = block_base + threadIdx.x;
  double obs[NPB_OBS_DIM];
  npb_prim_t s;
  NPD_LOAD(PRIM, npb_prim_t, s, 0);
  npd_obs_primary(s, obs);
  if (mode == NPB_MODE_PRIMARY) {   /* sim.py:333: twelve values */
#pragma unroll
    for (int k = 12; k < NPB_OBS_DIM; k++) obs[k] = 0.0;
    npd_store_rows<NPB_OBS_DIM>(obs, obs_out, lds, block_base, (size_t)n_plants);
    return;
  }
  obs[12] = NPD_F64_COL(SEC, npb_sec_t, electrical_power_output, 0) / 1100;
  obs[13] = NPD_F64_COL(SEC, npb_sec_t, thermal_efficiency, 0) / 0.35;
  obs[14] = NPD_F64_COL(SEC, npb_sec_t, total_steam_flow, 0) / 1665;
  obs[15] = NPD_F64_COL(SEC, npb_sec_t, load_demand, 0) / 100;
  obs[16] = 227.0 / 250;
  obs[17] = NPD_F64_COL(SEC, npb_sec_t, cooling_water_temperature, 0) / 35;
  double fwf, fwp; int fwa;
  if (mode == NPB_MODE_PRIMARY_SG) { fwf = NPD_F64_COL(SEC, npb_sec_t, total_feedwater_flow, 0); fwp = 0.0; fwa = 1; }
  else {
    fwf = NPD_F64_COL(FW, npb_fw_t, total_flow_rate, 0);
    fwp = NPD_F64_COL(FW, npb_fw_t, total_power_consumption, 0);
    fwa = NPD_I32_COL(FW, npb_fw_t, system_availability, 0) != 0;
  }
  obs[18] = fwf / 1665;
  obs[19] = fwp / 40;
  obs[20] = (double)fwa;
  obs[21] = fwf / 1665;
  npd_store_rows<NPB_OBS_DIM>(obs, obs_out, lds, block_base, (size_t)n_plants);
}

/* construction-time state for every plant selected by mask (NULL = all): the state the reference's
 * constructors leave behind with the default SecondarySystemConfig (npd_init.h) */
__global__ __launch_bounds__(NPB_WAVE) void npb_init_kernel(npb_params_t P, size_t N, npd_real_t *__restrict__ f64,
                                                            const uint8_t *__restrict__ mask,
                                                            int n_plants) {
  const size_t p = (size_t)blockIdx.x * NPB_WAVE + threadIdx.x;
  NPD_SEGMENT(f64, N, p);
  if (mask && p < (size_t)n_plants && !mask[p]) return;
  if (mask && p >= (size_t)n_plants) return;
  { npb_prim_t s; npd_prim_init(&s); NPD_STORE(PRIM, npb_prim_t, s, 0); }
#pragma unroll 1
  for (int i = 0; i < NPB_NUM_SG; i++) { npb_sg_t g; npd_sg_init(&g); NPD_STORE(SG, npb_sg_t, g, i); }
#pragma unroll 1
  for (int i = 0; i < NPB_NUM_PUMPS; i++) { npb_pump_t pm; npd_pump_init(&pm, i); NPD_STORE(PUMP, npb_pump_t, pm, i); }
  { npb_fw_t fw; npd_fw_init(&fw); NPD_STORE(FW, npb_fw_t, fw, 0); }
  { npb_turb_t t; npb_tstg_t g; npd_turb_init(&t, &g); NPD_STORE(TURB, npb_turb_t, t, 0); NPD_STORE(TSTG, npb_tstg_t, g, 0); }
#pragma unroll 1
  for (int i = 0; i < 2; i++) { npb_chem_t ch; npd_chem_init(&ch, i); NPD_STORE(CHEM, npb_chem_t, ch, i); }
  { npb_ph_t ph; npd_ph_init(&ph); NPD_STORE(PH, npb_ph_t, ph, 0); }
  { npb_cond_t cd; npd_cond_init(&cd); NPD_STORE(COND, npb_cond_t, cd, 0); }
  { npb_sec_t sec; npd_sec_init(&sec); NPD_STORE(SEC, npb_sec_t, sec, 0); }
  { npb_maint_t m; npd_maint_init(&m); NPD_STORE(MAINT, npb_maint_t, m, 0); }
#pragma unroll 1
  for (int i = 0; i < NPB_NUM_PUMPS; i++) { npb_mpump_t mp; npd_mpump_init(&mp); NPD_STORE(MPUMP, npb_mpump_t, mp, i); }
  (void)P;
}

/* NuclearPlantSimulator.reset(start_at_steady_state)  sim.py:546-581 for every plant selected by mask (NULL = all):
 * the reference's own reset semantics (npd_reset.h), which keep part of the plant's history -- unlike
 * npb_init_kernel, which stands in for constructing a new simulator.  The maint.* section is left alone (the
 * maintenance system is not reset; only the state manager's log is cleared, sim.py:573-574). */
__global__ __launch_bounds__(NPB_WAVE) void npb_reset_kernel(npb_params_t P, size_t N, npd_real_t *__restrict__ f64,
                                                             const uint8_t *__restrict__ mask, int n_plants, int steady) {
  const size_t p = (size_t)blockIdx.x * NPB_WAVE + threadIdx.x;
  NPD_SEGMENT(f64, N, p);
  if (mask && (p >= (size_t)n_plants || !mask[p])) return;
  { npb_prim_t s; NPD_LOAD(PRIM, npb_prim_t, s, 0); npd_prim_reset(&s); NPD_STORE(PRIM, npb_prim_t, s, 0); }
  npb_sec_t sec;
  NPD_LOAD(SEC, npb_sec_t, sec, 0);
  npd_sec_reset(&sec);
  npb_sg_t sg[NPB_NUM_SG];
  for (int i = 0; i < NPB_NUM_SG; i++) { NPD_LOAD(SG, npb_sg_t, sg[i], i); npd_sg_reset(&sg[i]); }
  npd_equilibrium_t eq;
  if (steady) {
    /* sim.py:558-563: primary_physics.thermal_power_mw was zeroed by reset_system, so the rated power is used */
    npd_steady_state_equilibrium(sg, &sec, &P, P.rated_power_mw, &eq);
    npd_sec_steady_state(&sec, &eq);
  }
  for (int i = 0; i < NPB_NUM_SG; i++) NPD_STORE(SG, npb_sg_t, sg[i], i);
  NPD_STORE(SEC, npb_sec_t, sec, 0);
#pragma unroll 1
  for (int i = 0; i < NPB_NUM_PUMPS; i++) {
    npb_pump_t pm;
    NPD_LOAD(PUMP, npb_pump_t, pm, i);
    npd_pump_reset(&pm, i);
    if (steady) npd_pump_steady_state(&pm, i, eq.steam_pressure, eq.feedwater_flow, eq.pumps_needed, eq.pump_speed);
    NPD_STORE(PUMP, npb_pump_t, pm, i);
  }
  { npb_fw_t fw; npd_fw_reset(&fw); NPD_STORE(FW, npb_fw_t, fw, 0); }
  {
    npb_turb_t t; npb_tstg_t g;
    NPD_LOAD(TURB, npb_turb_t, t, 0); NPD_LOAD(TSTG, npb_tstg_t, g, 0);
    npd_turb_reset(&t, &g, steady, steady ? eq.load_demand : 0.0, steady ? eq.electrical_power : 0.0);
    NPD_STORE(TURB, npb_turb_t, t, 0); NPD_STORE(TSTG, npb_tstg_t, g, 0);
  }
#pragma unroll 1
  for (int i = 0; i < 2; i++) { npb_chem_t ch; npd_chem_reset(&ch); NPD_STORE(CHEM, npb_chem_t, ch, i); }
  { npb_cond_t cd; NPD_LOAD(COND, npb_cond_t, cd, 0); npd_cond_reset(&cd); NPD_STORE(COND, npb_cond_t, cd, 0); }
  /* the pH controller and its pending doses are not reset (secondary/__init__.py:1041-1072 never touches them) */
}

#ifndef NPB_BUILD_F32
/* calibration aid for the HBM traffic counters: reads every arena column and writes it back unchanged,
 * with exactly the access shape of the step kernel (8 B per lane, one 512-B line per wave and column),
 * so that FETCH_SIZE / WRITE_SIZE can be scaled against a known byte count (2 * state_bytes * pitch) */
__global__ __launch_bounds__(NPB_WAVE) void npb_touch_kernel(size_t N, double *__restrict__ f64) {
  const size_t p = (size_t)blockIdx.x * NPB_WAVE + threadIdx.x;
  { const size_t seg = N >> 32; N &= 0xffffffffu; if (seg) { f64 += (p / seg) * seg * (size_t)(NPB_TOTAL_COL64 - 1); N = seg; } }
#pragma unroll 8
  for (int k = 0; k < NPB_TOTAL_COL64; k++) { double v = f64[(size_t)k * N + p]; f64[(size_t)k * N + p] = v + 0.0; }
}
extern "C" void npb_launch_touch(size_t npad_seg, double *f64, hipStream_t stream) {
  const size_t npad = NPD_NPAD(npad_seg);
  dim3 grid((unsigned)(npad / NPB_WAVE)), block(NPB_WAVE);
  hipLaunchKernelGGL(npb_touch_kernel, grid, block, 0, stream, npad_seg, f64);
}
#endif

/* ---- host-side launchers (called from npb_api.hip); one set per storage type */
#ifdef NPB_BUILD_F32
#define NPB_LAUNCHER(name) npb32_launch_##name
#else
#define NPB_LAUNCHER(name) npb_launch_##name
#endif

/* field access of the C ABI: one member of every plant <-> a contiguous buffer of double (real members) or
 * int32 (int members).  where = arena column, sub = narrow position inside the column, kind: 0 carried real,
 * 1 output real (stored as float), 2 int32 */
__global__ void npb_field_get_kernel(const npd_real_t *__restrict__ arena, size_t N, int col, int sub, int kind, void *__restrict__ out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  NPD_SEGMENT(arena, N, i);
  const char *e = (const char *)(arena + (size_t)col * N + i);
  if (kind == 0) ((double *)out)[i] = (double)*(const npd_real_t *)e;
  else if (kind == 1) ((double *)out)[i] = (double)*(const float *)(e + sub * 4);
  else ((int32_t *)out)[i] = *(const int32_t *)(e + sub * 4);
}
__global__ void npb_field_set_kernel(npd_real_t *__restrict__ arena, size_t N, int col, int sub, int kind, const void *__restrict__ in, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  NPD_SEGMENT(arena, N, i);
  char *e = (char *)(arena + (size_t)col * N + i);
  if (kind == 0) *(npd_real_t *)e = (npd_real_t)((const double *)in)[i];
  else if (kind == 1) *(float *)(e + sub * 4) = (float)((const double *)in)[i];
  else *(int32_t *)(e + sub * 4) = ((const int32_t *)in)[i];
}
/* many members at once, every value widened to double: out[f * n + plant]; plan[f] = {column, sub, kind} (the state
 * log's sampling step, nuclear_sim_amd/statelog.py) */
__global__ void npb_gather_kernel(const npd_real_t *__restrict__ arena, size_t N, const int *__restrict__ plan, double *__restrict__ out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, f = blockIdx.y;
  if (i >= n) return;
  const int col = plan[3 * f], sub = plan[3 * f + 1], kind = plan[3 * f + 2];
  NPD_SEGMENT(arena, N, i);
  const char *e = (const char *)(arena + (size_t)col * N + i);
  double v;
  if (kind == 0) v = (double)*(const npd_real_t *)e;
  else if (kind == 1) v = (double)*(const float *)(e + sub * 4);
  else v = (double)*(const int32_t *)(e + sub * 4);
  out[(size_t)f * n + i] = v;
}
extern "C" void NPB_LAUNCHER(gather)(const void *arena, size_t npad, const int *plan_dev, int n_fields, double *out, int n, hipStream_t stream) {
  hipLaunchKernelGGL(npb_gather_kernel, dim3((n + 255) / 256, n_fields), dim3(256), 0, stream, (const npd_real_t *)arena, npad, plan_dev, out, n);
}
extern "C" void NPB_LAUNCHER(field_get)(const void *arena, size_t npad, int col, int sub, int kind, void *out, int n, hipStream_t stream) {
  hipLaunchKernelGGL(npb_field_get_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, (const npd_real_t *)arena, npad, col, sub, kind, out, n);
}
extern "C" void NPB_LAUNCHER(field_set)(void *arena, size_t npad, int col, int sub, int kind, const void *in, int n, hipStream_t stream) {
  hipLaunchKernelGGL(npb_field_set_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, (npd_real_t *)arena, npad, col, sub, kind, in, n);
}
/* fp64-storage plants above which the sweep of a step (4 221 B per plant) is so far past the 256 MB Infinity Cache that streaming
 * state stores win (measured: even at 81 920, -5 % at 98 304, -10 % at 131 072); fp32 storage moves half the bytes */
#define NPB_NT_STORE_ABOVE ((size_t)90112)
/* plants (of either storage type) from which on npb_create segments the arena ("segmented arena" above), and between which npb_step gives
 * the batch to the four-wave kernel although its groups no longer fit at once.  Measured
 * (profiles/r3_segment_sweep.txt, r3_segment_size.txt): 65 536 plants 0.0906 ms against the one-wave kernel's 0.0956 on a one-block arena
 * (the four-wave kernel there: 0.0975); 49 152 0.0784 against the two-wave kernel's 0.0792, 40 960 level; 81 920 0.122 against 0.166;
 * 106 496 0.169 against the streaming build's 0.175, 131 072 0.202 against 0.193; the streaming build itself gains 2 % at 131 072 plants
 * and 10 % at 262 144 from the segments (profiles/r3_segment_size.txt) */
#define NPB_SEGMENTED_FROM ((size_t)45056)
#define NPB_SEGMENTED_UP_TO ((size_t)114688)
/* the table as the step kernels' pump phase evaluates it (npd_maintenance.h): a strict comparison as the sign of fma(value, sgn, c);
 * any other comparison kind in the scan makes every (wave, pump) "look properly" */
static void npd_maint_fold_table(const npb_params_t *P, const npb_maint_table_t *T, npd_maint_hot_t *H) {
  bool always = false;
  for (int q = 0; q < NPB_MAINT_NPARAM; q++) {
    H->tab[q] = 0.0; H->tab[NPB_MAINT_NPARAM + q] = -1.0;
    if (!T || T->rank[q] < 0) continue;
    if (T->comparison[q] == NPB_CMP_GREATER_THAN) { H->tab[q] = 1.0; H->tab[NPB_MAINT_NPARAM + q] = -T->threshold[q]; }
    else if (T->comparison[q] == NPB_CMP_LESS_THAN) { H->tab[q] = -1.0; H->tab[NPB_MAINT_NPARAM + q] = T->threshold[q]; }
    else always = true;
    if (!(T->threshold[q] == T->threshold[q]) || T->threshold[q] - T->threshold[q] != 0.0) always = true;   /* a NaN or infinite threshold: leave it to the real comparison */
  }
  H->tab[2 * NPB_MAINT_NPARAM] = always ? 1.0 : 0.0;
  H->tab[2 * NPB_MAINT_NPARAM + 1] = P->maint_check_interval_hours * 60;
}
/* the handle's maintenance side buffer: [rule constants, 256-byte slot][cache entries: npad x 4 pumps x {u32, float}] */
#define NPD_MAINT_CONSTS_BYTES ((sizeof(npd_maint_rule_consts_t) + 255) / 256 * 256)
static npd_maint_cache_t npd_maint_cache_of(void *maint_side, int32_t *counts, int n_plants, double *diag = nullptr, size_t diag_pitch = 0) {
  npd_maint_cache_t C;
  C.counts = counts; C.n_plants = n_plants; C.diag = diag; C.diag_pitch = diag_pitch;
  C.entry = maint_side ? (npd_u32x4 *)((char *)maint_side + NPD_MAINT_CONSTS_BYTES) : nullptr;
  return C;
}
/* maint_table / maint_side (the handle's maintenance side buffer, rule constants uploaded): NULL unless the automatic maintenance is on (npb_step) */
extern "C" int NPB_LAUNCHER(step)(const npb_params_t *P, int n_plants, size_t npad_seg, void *arena,
                                const int32_t *action, const double *magnitude, const double *setpoint,
                                const double *noise_z, const double *cw_temp, double *obs, double *reward, uint8_t *done,
                                uint32_t *trip_flags, double *info, int variant, double *diag, size_t diag_pitch,
                                const npb_maint_table_t *maint_table, void *maint_side, int32_t *maint_counts, hipStream_t stream) {
  const size_t npad = NPD_NPAD(npad_seg), seg = NPD_SEG_OF(npad_seg);   /* column pitch in plants / plants per arena segment (0: one segment) */
  npd_maint_hot_t MH;
  npd_maint_fold_table(P, maint_side ? maint_table : nullptr, &MH);
  const npd_maint_cache_t MC = npd_maint_cache_of(maint_side, maint_counts, n_plants, diag, diag_pitch);
  const npd_maint_rule_consts_t *maint_rc = (const npd_maint_rule_consts_t *)maint_side;     /* NULL = off */
  dim3 grid((unsigned)(npad / NPB_WAVE)), block(NPB_WAVE);
  if (diag && P->mode == NPB_MODE_FULL) {   /* npb_set_diagnostics: the diagnostics build of the one-wave kernel at any size */
    hipLaunchKernelGGL(npb_step_diag_kernel, grid, block, 0, stream, *P, n_plants, npad_seg, (npd_real_t *)arena, action, magnitude, setpoint,
                       noise_z, cw_temp, obs, reward, done, trip_flags, info, MH, maint_rc, MC, diag, diag_pitch);
    return NPB_KERNEL_STEP_DIAG;
  }
  /* two kernels, one result (the same device functions in the same order per plant; tests/test_gpu_parity.py,
   * test_the_two_step_kernels_agree).  The more waves share a plant, the shorter the critical path of a step and the more of
   * the chip a small batch fills: four waves (npd_step4.h) while they are all resident, two (npd_step2.h) up to ~57 k plants;
   * once the one-wave kernel has a wave for every SIMD its LDS-DMA pipeline wins (measured crossovers).
   * variant: 0 = by batch size, 1 = one wave per 64 plants, 2 = two waves, 3 = their two-per-SIMD build, 4 = one wave with
   * streaming state stores (what 0 picks once the sweep is far past the Infinity Cache), 5 = four waves (npd_step4.h: what 0
   * picks while all its waves are resident at once, up to 32 768 plants, and again on the segmented arenas of 45 057 .. 114 688
   * plants).  The primary + steam-generator
   * mode always takes a one-wave kernel.  The return value names the kernel that was launched (npb_debug_last_step_kernel). */
  if (P->mode == NPB_MODE_PRIMARY) {
    hipLaunchKernelGGL(npb_step_primary_kernel, grid, block, 0, stream, *P, n_plants, npad_seg, (npd_real_t *)arena, action, magnitude, setpoint,
                       noise_z, obs, reward, done, trip_flags, info);
    return NPB_KERNEL_STEP_PRIMARY;
  }
  if (variant == 0) variant = npad <= 32768 ? 5 : (npad <= NPB_SEGMENTED_FROM ? 2 : (npad <= NPB_SEGMENTED_UP_TO ? 5 : (npad * sizeof(npd_real_t) > NPB_NT_STORE_ABOVE * 8 ? 4 : 1)));
  const bool with_maint = maint_rc != nullptr;     /* the builds with the automatic maintenance compiled in */
  if (variant == 4) {
    hipLaunchKernelGGL(with_maint ? npb_step_nt_maint_kernel : npb_step_nt_kernel, grid, block, 0, stream, *P, n_plants, npad_seg, (npd_real_t *)arena, action, magnitude, setpoint,
                       noise_z, cw_temp, obs, reward, done, trip_flags, info, MH, maint_rc, MC);
    return with_maint ? NPB_KERNEL_STEP_NT_MAINT : NPB_KERNEL_STEP_NT;
  }
  if (variant == 5 && P->mode == NPB_MODE_FULL) {     /* four waves per 64 plants (npd_step4.h) */
    hipLaunchKernelGGL(with_maint ? npb_step4_maint_kernel : npb_step4_kernel, grid, dim3(NPD4_THREADS), 0, stream, *P, n_plants, npad_seg, (npd_real_t *)arena, action, magnitude,
                       setpoint, noise_z, cw_temp, obs, reward, done, trip_flags, info, MH, maint_rc, MC);
    return with_maint ? NPB_KERNEL_STEP4_MAINT : NPB_KERNEL_STEP4;
  }
  const bool two_wave = (variant == 2 || variant == 3) && P->mode == NPB_MODE_FULL;
  const bool wide = two_wave && variant == 2 && npad <= 32768;   /* the whole register file while one wave per SIMD is all there is; variant 3 = never */
  if (wide) {
    hipLaunchKernelGGL(with_maint ? npb_step2_wide_maint_kernel : npb_step2_wide_kernel, grid, dim3(NPD2_THREADS), 0, stream, *P, n_plants, npad_seg, (npd_real_t *)arena, action, magnitude,
                       setpoint, noise_z, cw_temp, obs, reward, done, trip_flags, info, MH, maint_rc, MC);
    return with_maint ? NPB_KERNEL_STEP2_WIDE_MAINT : NPB_KERNEL_STEP2_WIDE;
  }
  if (two_wave) {
    hipLaunchKernelGGL(with_maint ? npb_step2_maint_kernel : npb_step2_kernel, grid, dim3(NPD2_THREADS), 0, stream, *P, n_plants, npad_seg, (npd_real_t *)arena, action, magnitude,
                       setpoint, noise_z, cw_temp, obs, reward, done, trip_flags, info, MH, maint_rc, MC);
    return with_maint ? NPB_KERNEL_STEP2_MAINT : NPB_KERNEL_STEP2;
  }
  hipLaunchKernelGGL(with_maint ? npb_step_maint_kernel : npb_step_kernel, grid, block, 0, stream, *P, n_plants, npad_seg, (npd_real_t *)arena, action, magnitude, setpoint,
                     noise_z, cw_temp, obs, reward, done, trip_flags, info, MH, maint_rc, MC);
  return with_maint ? NPB_KERNEL_STEP_MAINT : NPB_KERNEL_STEP;
}
/* the rule as a launch of its own (modes that do not step the pumps) */
extern "C" void NPB_LAUNCHER(maint)(size_t npad, void *arena, void *maint_side, int32_t *counts, int n_plants, hipStream_t stream) {
  hipLaunchKernelGGL(npb_maint_kernel, dim3((unsigned)(NPD_NPAD(npad) / NPB_WAVE)), dim3(NPB_WAVE), 0, stream, (const npd_maint_rule_consts_t *)maint_side,
                     npd_maint_cache_of(maint_side, counts, n_plants), npad, (npd_real_t *)arena);
}
/* the rule's constants as the device reads them: host_out = NPB_LAUNCHER(maint_consts_bytes)() bytes */
extern "C" void NPB_LAUNCHER(maint_consts)(const npb_params_t *P, const npb_maint_table_t *T, void *host_out) {
  npd_maint_rule_consts_t *RC = (npd_maint_rule_consts_t *)host_out;
  memset(RC, 0, sizeof(*RC));
  RC->P = *P; RC->T = *T;
  npd_maint_screen_t &S = RC->S;
  for (int q = 0; q < NPB_MAINT_NPARAM; q++) {
    S.threshold[q] = T->threshold[q];
    S.cooldown_minutes[q] = T->cooldown_hours[q] * 60;
    if (T->rank[q] < 0) continue;             /* not in the scan: no mask bit, never fires */
    const int c = T->comparison[q];
    const uint32_t bit = 1u << q;
    if (c == NPB_CMP_GREATER_THAN || c == NPB_CMP_GREATER_EQUAL) S.want_gt |= bit;
    if (c == NPB_CMP_LESS_THAN || c == NPB_CMP_LESS_EQUAL) S.want_lt |= bit;
    if (c == NPB_CMP_GREATER_EQUAL || c == NPB_CMP_LESS_EQUAL) S.want_eq |= bit;
    if (c == NPB_CMP_EQUALS) S.want_near |= bit;
    if (c != NPB_CMP_GREATER_THAN && c != NPB_CMP_GREATER_EQUAL && c != NPB_CMP_LESS_THAN && c != NPB_CMP_LESS_EQUAL && c != NPB_CMP_EQUALS) S.want_far |= bit;
  }
}
extern "C" size_t NPB_LAUNCHER(maint_consts_bytes)(void) { return sizeof(npd_maint_rule_consts_t); }
/* rule constants + cooldown cache (npd_maint_cache_of); a zeroed cache = "nothing known: look" */
extern "C" size_t NPB_LAUNCHER(maint_side_bytes)(size_t npad) { return NPD_MAINT_CONSTS_BYTES + (size_t)NPB_NUM_PUMPS * NPD_NPAD(npad) * (sizeof(float) + sizeof(uint32_t)); }
extern "C" size_t NPB_LAUNCHER(maint_cache_offset)(void) { return NPD_MAINT_CONSTS_BYTES; }
extern "C" void NPB_LAUNCHER(observe)(int mode, int n_plants, size_t npad, const void *arena, double *obs, hipStream_t stream) {
  dim3 grid((unsigned)(NPD_NPAD(npad) / NPB_WAVE)), block(NPB_WAVE);
  hipLaunchKernelGGL(npb_observe_kernel, grid, block, 0, stream, mode, n_plants, npad, (const npd_real_t *)arena, obs);
}
extern "C" void NPB_LAUNCHER(reset)(const npb_params_t *P, int n_plants, size_t npad, void *arena, const uint8_t *mask, int steady, hipStream_t stream) {
  dim3 grid((unsigned)(NPD_NPAD(npad) / NPB_WAVE)), block(NPB_WAVE);
  hipLaunchKernelGGL(npb_reset_kernel, grid, block, 0, stream, *P, npad, (npd_real_t *)arena, mask, n_plants, steady);
}
extern "C" void NPB_LAUNCHER(init)(const npb_params_t *P, int n_plants, size_t npad, void *arena, const uint8_t *mask, hipStream_t stream) {
  dim3 grid((unsigned)(NPD_NPAD(npad) / NPB_WAVE)), block(NPB_WAVE);
  hipLaunchKernelGGL(npb_init_kernel, grid, block, 0, stream, *P, npad, (npd_real_t *)arena, mask, n_plants);
}
