/*
 * npd_common.h -- shared helpers of the HIP plant stepper's device physics (product code).
 *
 * The npd_*.h headers hold the per-subsystem update rules of the per-timestep physics path
 * (reference: NuclearPlantSimulator.step, simulator/core/sim.py:130-258) as __device__
 * functions over small per-subsystem register structs; the fused kernel in npb_kernels.hip
 * streams a plant's state through them subsystem by subsystem (load SoA columns -> update ->
 * store).  The update rules reproduce the reference's clipped-Euler / first-order-lag /
 * branchy arithmetic literally, including its unit quirks.
 *
 * Helper semantics mirror the numpy / Python builtins the reference calls on scalars:
 *   np.clip(x, lo, hi) -> npd_clip (propagates NaN); max(a, b) / min(a, b) keep the FIRST
 *   argument unless the second compares strictly greater / smaller.
 */
#ifndef NPD_COMMON_H
#define NPD_COMMON_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/npb.h"

#define NPD_FN __device__ __forceinline__

#define NPD_PI 3.141592653589793

/* ---- exp / log for this kernel.  The device library's fp64 exp / log / log10 cost ~50 / ~105 / ~115
 * instructions; a step evaluates a few hundred of them and, with one wave per SIMD, every instruction is
 * paid in full.  These versions cost ~25 / ~40 and stay within 4e-16 relative of libm (CPU mirror checked on
 * 4e6 points over 1e-13..1e13, tools/fastmath_check.c), far inside the 1e-6 parity budget.
 * Domain handling kept: log(0) = -inf, log(x<0) = NaN, log(inf) = inf, exp(-inf) = 0, exp(inf) = inf, NaN in ->
 * NaN out. */
NPD_FN double npd_rcp(double y) { /* 1 / y to working precision: hardware seed + two Newton steps */
  double r = __builtin_amdgcn_rcp(y);
  r = __builtin_fma(__builtin_fma(-y, r, 1.0), r, r);
  r = __builtin_fma(__builtin_fma(-y, r, 1.0), r, r);
  return r;
}
NPD_FN double npd_log(double x) {
  double m = __builtin_amdgcn_frexp_mant(x); /* [0.5, 1) */
  int e = __builtin_amdgcn_frexp_exp(x);
  const bool low = m < 0.70710678118654752440;
  m = low ? m * 2.0 : m;
  e = low ? e - 1 : e;
  const double f = m - 1.0, g = 2.0 + f;
  const double r = npd_rcp(g);
  double s = f * r;
  s = __builtin_fma(__builtin_fma(-g, s, f), r, s);
  const double z = s * s;
  double p = 2.0 / 21.0; /* 2 atanh(s) = 2s + 2s^3/3 + ..., |s| <= 0.1716 */
  p = __builtin_fma(p, z, 2.0 / 19.0); p = __builtin_fma(p, z, 2.0 / 17.0); p = __builtin_fma(p, z, 2.0 / 15.0);
  p = __builtin_fma(p, z, 2.0 / 13.0); p = __builtin_fma(p, z, 2.0 / 11.0); p = __builtin_fma(p, z, 2.0 / 9.0);
  p = __builtin_fma(p, z, 2.0 / 7.0); p = __builtin_fma(p, z, 2.0 / 5.0); p = __builtin_fma(p, z, 2.0 / 3.0);
  const double lm = __builtin_fma(s * z, p, 2.0 * s);
  const double de = (double)e;
  double res = __builtin_fma(de, 6.93147180369123816490e-01, __builtin_fma(de, 1.90821492927058770002e-10, lm));
  res = (x == 0.0) ? -INFINITY : res;
  res = (x < 0.0) ? NAN : res;
  res = (x == INFINITY) ? INFINITY : res;
  return res;
}
NPD_FN double npd_exp(double x) {
  double xc = (x < -800.0) ? -800.0 : x; /* NaN falls through both comparisons */
  xc = (xc > 800.0) ? 800.0 : xc;
  const double n = __builtin_rint(xc * 1.44269504088896338700e+00);
  double r = __builtin_fma(-n, 6.93147180369123816490e-01, xc);
  r = __builtin_fma(-n, 1.90821492927058770002e-10, r);
  double p = 1.0 / 6227020800.0; /* Taylor to r^13, |r| <= 0.3466 */
  p = __builtin_fma(p, r, 1.0 / 479001600.0); p = __builtin_fma(p, r, 1.0 / 39916800.0); p = __builtin_fma(p, r, 1.0 / 3628800.0);
  p = __builtin_fma(p, r, 1.0 / 362880.0); p = __builtin_fma(p, r, 1.0 / 40320.0); p = __builtin_fma(p, r, 1.0 / 5040.0);
  p = __builtin_fma(p, r, 1.0 / 720.0); p = __builtin_fma(p, r, 1.0 / 120.0); p = __builtin_fma(p, r, 1.0 / 24.0);
  p = __builtin_fma(p, r, 1.0 / 6.0); p = __builtin_fma(p, r, 0.5); p = __builtin_fma(p, r, 1.0); p = __builtin_fma(p, r, 1.0);
  return __builtin_amdgcn_ldexp(p, (int)n); /* overflow -> inf, underflow -> 0 */
}
NPD_FN double npd_log10(double x) { return npd_log(x) * 4.34294481903251827651e-01; }

/* x^c for x >= 0: exp(c * log(x)); npd_powc(0, c) = 0 for c > 0 is preserved (log(0) = -inf, exp(-inf) = 0) */
NPD_FN double npd_powc(double x, double c) { return npd_exp(c * npd_log(x)); }

/* a^p * b^q as ONE exponential of p*log(a) + q*log(b), with the logarithms supplied by the caller: the wear-rate
 * formulas raise the same few factors to several exponents, and a product of powers needs one exp, not one per
 * factor (a log + exp pair is ~120 instructions for a lone wave).  Zero factors still give 0 (log 0 = -inf, exp -inf
 * = 0), NaN and negative factors NaN; the result differs from the product of separately rounded powers in the last
 * bits only. */
NPD_FN double npd_pow_logs(double la, double p, double lb, double q) { return npd_exp(p * la + q * lb); }

NPD_FN double npd_sq(double x) { return x * x; }
/* np.clip / Python max / min on doubles.  A compare + select on fp64 costs a lone wave ~20 cycles (v_cmp_f64 to VCC,
 * the VCC hazard nop, one v_cndmask_b32 per register half) against ~5 for v_max_f64 / v_min_f64, and the path has
 * several hundred of them per plant-step (tools/membench/clipcost.hip, cmpsel.hip).  The hardware min / max return
 * the non-NaN operand, while np.clip and Python's max(a, b) / min(a, b) give NaN when x / a is NaN; the
 * fma(x, 0.0, r) restores exactly that (0 * NaN = NaN, 0 * finite = +-0 and r + +-0 = r).  The one difference left:
 * an INFINITE x / a comes out as NaN instead of the bound / itself -- nothing on the path produces infinities from
 * finite state, and NaN is the value the reference's own check_for_nan_values looks for. */
NPD_FN double npd_clip(double x, double lo, double hi) { /* np.minimum(np.maximum(x, lo), hi): NaN propagates, lo > hi gives hi */
  return __builtin_fma(x, 0.0, __builtin_fmin(__builtin_fmax(x, lo), hi));
}
NPD_FN double npd_pymax(double a, double b) { return __builtin_fma(a, 0.0, __builtin_fmax(a, b)); } /* (b > a) ? b : a */
NPD_FN double npd_pymin(double a, double b) { return __builtin_fma(a, 0.0, __builtin_fmin(a, b)); } /* (b < a) ? b : a */

/* per-step inputs of one plant (what step() receives, sim.py:130-133, plus the
 * pre-drawn standard-normal sample that replaces ConstantHeatSource's MT19937 draw) */
typedef struct npd_inputs_t {
  int32_t action;          /* ControlAction value 0..14 (primary/__init__.py:28-45); 8 = NO_ACTION */
  double magnitude;        /* step(magnitude=...) */
  double power_setpoint;   /* heat_source.set_power_setpoint(x) before the step; NaN = leave unchanged */
  double noise_z;          /* standard normal sample for constant_heat_source.py:178 */
  double cooling_water_temp; /* step(cooling_water_temp=...); NaN = leave unchanged */
} npd_inputs_t;

typedef struct npd_outputs_t {
  double obs[NPB_OBS_DIM];
  double reward;
  double info[NPB_INFO_DIM];
  uint32_t trip_flags;
  uint8_t done;
} npd_outputs_t;

#endif
