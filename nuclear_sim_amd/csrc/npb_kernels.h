/* npb_kernels.h -- host-callable launchers of the device kernels (internal to libnpb.so).
 * npb_kernels.hip is compiled twice: fp64 storage (npb_launch_*) and fp32 storage (npb32_launch_*, -DNPB_BUILD_F32);
 * the arena pointer is void* here and typed inside each translation unit. */
#ifndef NPB_KERNELS_H
#define NPB_KERNELS_H
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/npb_params.h"
#ifdef __cplusplus
extern "C" {
#endif
#define NPB__DECL(prefix) \
  void prefix##step(const npb_params_t *P, int n_plants, size_t npad, void *f64, int32_t *i32, const int32_t *action, \
                    const double *magnitude, const double *setpoint, const double *noise_z, const double *cw_temp, \
                    double *obs, double *reward, uint8_t *done, uint32_t *trip_flags, double *info, hipStream_t stream); \
  void prefix##maint(const npb_params_t *P, size_t npad, void *f64, int32_t *i32, hipStream_t stream); \
  void prefix##observe(int mode, int n_plants, size_t npad, const void *f64, const int32_t *i32, double *obs, hipStream_t stream); \
  void prefix##init(const npb_params_t *P, int n_plants, size_t npad, void *f64, int32_t *i32, const uint8_t *mask, hipStream_t stream);
NPB__DECL(npb_launch_)
NPB__DECL(npb32_launch_)
#undef NPB__DECL
void npb_launch_touch(size_t npad, double *f64, int32_t *i32, hipStream_t stream);
void npb32_launch_col_to_f64(const float *col, double *out, int n, hipStream_t stream);
void npb32_launch_col_from_f64(float *col, const double *in, int n, hipStream_t stream);
#ifdef __cplusplus
}
#endif
#endif
