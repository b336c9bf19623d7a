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

/* x^c for a compile-time exponent and x >= 0: exp(c * log(x)) costs about half of the generic npd_powc()
 * and differs from it by a few ulp, far inside the 1e-6 parity budget (npd_powc(0, c) = 0 is preserved:
 * log(0) = -inf, exp(-inf) = 0). */
NPD_FN double npd_powc(double x, double c) { return exp(c * log(x)); }

NPD_FN double npd_sq(double x) { return x * x; }
NPD_FN double npd_clip(double x, double lo, double hi) { /* np.minimum(np.maximum(x, lo), hi): NaN propagates, lo > hi gives hi */
  double t = (x < lo) ? lo : x;
  return (t > hi) ? hi : t;
}
NPD_FN double npd_pymax(double a, double b) { return (b > a) ? b : a; }
NPD_FN double npd_pymin(double a, double b) { return (b < a) ? b : a; }

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
