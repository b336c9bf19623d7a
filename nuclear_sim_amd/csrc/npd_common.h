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
 * paid in full.  These versions cost ~21 / ~34 (plus their literals) and stay within 4e-16 relative of libm (CPU mirror checked on
 * 4e6 points over 1e-13..1e13, tools/fastmath_check.c), far inside the 1e-6 parity budget.
 * Domain handling kept: log(0) = -inf, log(x<0) = NaN, log(inf) = inf, exp(-inf) = 0, exp(inf) = inf, NaN in ->
 * NaN out. */
NPD_FN double npd_rcp(double y) { /* 1 / y to working precision: hardware seed + two Newton steps */
  double r = __builtin_amdgcn_rcp(y);
  r = __builtin_fma(__builtin_fma(-y, r, 1.0), r, r);
  r = __builtin_fma(__builtin_fma(-y, r, 1.0), r, r);
  return r;
}
/* the logarithm of a positive finite number (a NaN comes out as NaN by itself; anything else is the caller's to exclude) */
NPD_FN double npd_log_pos(double x) {
  double m = __builtin_amdgcn_frexp_mant(x); /* [0.5, 1) */
  int e = __builtin_amdgcn_frexp_exp(x);
  /* into [sqrt(1/2), sqrt(2)) by integer arithmetic on the high word: no compare, no select (each costs a lone wave ~20 cycles) */
  const uint32_t hm = (uint32_t)__double2hiint(m);
  const uint32_t low = (hm - 0x3fe6a09eu) >> 31;
  m = __hiloint2double((int)(hm + (low << 20)), __double2loint(m));
  e -= (int)low;
  const double f = m - 1.0, g = m + 1.0;
  double r = __builtin_amdgcn_rcp(g);                     /* seed + one Newton step; the quotient below is corrected once more */
  r = __builtin_fma(__builtin_fma(-g, r, 1.0), r, r);
  double s = f * r;
  s = __builtin_fma(__builtin_fma(-g, s, f), r, s);
  const double z = s * s;
  double p = 0.14616878919029820754;  /* near-minimax for (2 atanh(s) - 2s) / s^3 in z = s^2, |s| <= 0.1716 (mpmath chebyfit, 3e-16) */
  p = __builtin_fma(p, z, 0.15331686868638428253); p = __builtin_fma(p, z, 0.18182890170313970214); p = __builtin_fma(p, z, 0.22222211120449298486);
  p = __builtin_fma(p, z, 0.28571428626063380364); p = __builtin_fma(p, z, 0.39999999999899310681); p = __builtin_fma(p, z, 0.66666666666666696929);
  const double lm = __builtin_fma(s * z, p, 2.0 * s);
  const double de = (double)e;
  return __builtin_fma(de, 6.93147180369123816490e-01, __builtin_fma(de, 1.90821492927058770002e-10, lm));
}
NPD_FN double npd_log(double x) {
  const double res = npd_log_pos(x);
  /* everything but a positive finite number (zero, negative, +inf, NaN): one class test, and the value the float unit's own
   * logarithm gives for it: log(+-0) = -inf, log(x < 0) = NaN, log(inf) = inf, NaN -> NaN */
  const double special = (double)__builtin_amdgcn_logf((float)x);
  return __builtin_amdgcn_class(x, 0x27f) ? special : res;
}
/* exp of an argument the caller knows to be of moderate size (|x| < 700) or NaN: no clamp, and a NaN propagates by itself */
NPD_FN double npd_exp_bounded(double x) {
  const double n = __builtin_rint(x * 1.44269504088896338700e+00);
  double r = __builtin_fma(-n, 6.93147180369123816490e-01, x);
  r = __builtin_fma(-n, 1.90821492927058770002e-10, r);
  double q = 2.5100385495510319077e-8;   /* near-minimax for (exp(r) - 1 - r) / r^2, |r| <= ln2 / 2 (mpmath chebyfit, 1e-16) */
  q = __builtin_fma(q, r, 2.762008844540974816e-7); q = __builtin_fma(q, r, 2.7557268459997064772e-6); q = __builtin_fma(q, r, 0.000024801521295954375131);
  q = __builtin_fma(q, r, 0.00019841269863053616878); q = __builtin_fma(q, r, 0.0013888888917213716901); q = __builtin_fma(q, r, 0.0083333333333300618325);
  q = __builtin_fma(q, r, 0.041666666666624127873); q = __builtin_fma(q, r, 0.16666666666666667453); q = __builtin_fma(q, r, 0.50000000000000010221);
  const double p = __builtin_fma(__builtin_fma(q, r, 1.0), r, 1.0);
  return __builtin_amdgcn_ldexp(p, (int)n);
}
NPD_FN double npd_exp(double x) {
  const double xc = __builtin_fmin(__builtin_fmax(x, -800.0), 800.0);   /* v_max / v_min: a NaN is put back at the end */
  const double n = __builtin_rint(xc * 1.44269504088896338700e+00);
  double r = __builtin_fma(-n, 6.93147180369123816490e-01, xc);
  r = __builtin_fma(-n, 1.90821492927058770002e-10, r);
  double q = 2.5100385495510319077e-8;   /* near-minimax for (exp(r) - 1 - r) / r^2, |r| <= ln2 / 2 (mpmath chebyfit, 1e-16) */
  q = __builtin_fma(q, r, 2.762008844540974816e-7); q = __builtin_fma(q, r, 2.7557268459997064772e-6); q = __builtin_fma(q, r, 0.000024801521295954375131);
  q = __builtin_fma(q, r, 0.00019841269863053616878); q = __builtin_fma(q, r, 0.0013888888917213716901); q = __builtin_fma(q, r, 0.0083333333333300618325);
  q = __builtin_fma(q, r, 0.041666666666624127873); q = __builtin_fma(q, r, 0.16666666666666667453); q = __builtin_fma(q, r, 0.50000000000000010221);
  const double p = __builtin_fma(__builtin_fma(q, r, 1.0), r, 1.0);
  const double res = __builtin_amdgcn_ldexp(p, (int)n); /* overflow -> inf, underflow -> 0 */
  /* NaN in -> NaN out: only the high word needs the select */
  return __hiloint2double(__builtin_isnan(x) ? 0x7ff80000 : __double2hiint(res), __double2loint(res));
}
/* sqrt for the magnitudes of this path (exact zeros, and 1e-200 .. 1e200 otherwise): hardware rsq seed, one coupled
 * Newton step and one residual correction (the device library's version spends 8 more instructions on rescaling
 * operands outside that range); sqrt(+-0) = +-0, sqrt(inf) = inf, negative and NaN operands give NaN */
NPD_FN double npd_sqrt(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = 0.5 * y;
  const double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g); h = __builtin_fma(h, r, h);
  g = __builtin_fma(__builtin_fma(-g, g, x), h, g);
  return __builtin_amdgcn_class(x, 0x260) ? x : g;
}
NPD_FN double npd_log10(double x) { return npd_log(x) * 4.34294481903251827651e-01; }

/* x^c for x >= 0: exp(c * log(x)); npd_powc(0, c) = 0 for c > 0 is preserved (log(0) = -inf, exp(-inf) = 0) */
NPD_FN double npd_powc(double x, double c) { return npd_exp(c * npd_log(x)); }
/* the same for a base that is positive, finite and within a few orders of magnitude of 1 (or NaN) */
NPD_FN double npd_powc_pos(double x, double c) { return npd_exp_bounded(c * npd_log_pos(x)); }

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
