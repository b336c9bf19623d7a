/*
 * npo_common.h -- helpers shared by the CPU oracle's subsystem restatements.
 *
 * TEST INFRASTRUCTURE ONLY.  oracle/ is a plain-C, scalar, fp64, one-plant-per-call
 * restatement of the reference's per-timestep physics path
 * (NuclearPlantSimulator.step, simulator/core/sim.py:130-258).  It exists to check
 * the HIP stepper; nothing in the product path may include, link or call it.
 * It is pinned against golden vectors generated from the importable Python
 * reference (tests/golden/, generator: oracle/ref_harness/make_golden.py).
 *
 * The helpers reproduce the *semantics* of the numpy / builtin calls the
 * reference uses on Python scalars:
 *   np.clip(x, lo, hi)  -> npo_clip   (propagates NaN like numpy)
 *   max(a, b), min(a, b) (Python builtins: keep the FIRST argument unless the
 *                         second compares strictly greater / smaller)
 */
#ifndef NPO_COMMON_H
#define NPO_COMMON_H

#include <math.h>
#include <string.h>
#include <stdint.h>
#include "../include/npb.h"

#ifndef NPO_FN
#define NPO_FN static inline
#endif

#define NPO_PI 3.141592653589793

NPO_FN double npo_clip(double x, double lo, double hi) { /* np.minimum(np.maximum(x, lo), hi): NaN propagates, lo > hi gives hi */
  double t = (x < lo) ? lo : x;
  return (t > hi) ? hi : t;
}
NPO_FN double npo_pymax(double a, double b) { return (b > a) ? b : a; }
NPO_FN double npo_pymin(double a, double b) { return (b < a) ? b : a; }

/* per-step inputs of one plant (what step() receives, sim.py:130-133, plus the
 * pre-drawn standard-normal sample that replaces ConstantHeatSource's MT19937 draw) */
typedef struct npo_inputs_t {
  int32_t action;          /* ControlAction value 0..14 (primary/__init__.py:28-45); 8 = NO_ACTION */
  double magnitude;        /* step(magnitude=...) */
  double power_setpoint;   /* heat_source.set_power_setpoint(x) before the step; NaN = leave unchanged */
  double noise_z;          /* standard normal sample for constant_heat_source.py:178 */
  double cooling_water_temp; /* step(cooling_water_temp=...); NaN = leave unchanged */
} npo_inputs_t;

typedef struct npo_outputs_t {
  double obs[NPB_OBS_DIM];
  double reward;
  double info[NPB_INFO_DIM];
  double rho[NPB_INFO_NRHO];   /* reactivity components (reactor heat source only; NaN otherwise) */
  uint32_t trip_flags;
  uint8_t done;
} npo_outputs_t;

#endif
