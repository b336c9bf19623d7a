/* npb_kernels.h -- host-callable launchers of the device kernels (internal to libnpb.so). */
#ifndef NPB_KERNELS_H
#define NPB_KERNELS_H
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/npb_params.h"
#ifdef __cplusplus
extern "C" {
#endif
void npb_launch_step(const npb_params_t *P, int n_plants, size_t npad, double *f64, int32_t *i32, const int32_t *action,
                     const double *magnitude, const double *setpoint, const double *noise_z, const double *cw_temp,
                     double *obs, double *reward, uint8_t *done, uint32_t *trip_flags, double *info, hipStream_t stream);
void npb_launch_maint(const npb_params_t *P, size_t npad, double *f64, int32_t *i32, hipStream_t stream);
void npb_launch_observe(int mode, int n_plants, size_t npad, const double *f64, const int32_t *i32, double *obs,
                        hipStream_t stream);
void npb_launch_init(const npb_params_t *P, int n_plants, size_t npad, double *f64, int32_t *i32, const uint8_t *mask,
                     hipStream_t stream);
void npb_launch_touch(size_t npad, double *f64, int32_t *i32, hipStream_t stream);
#ifdef __cplusplus
}
#endif
#endif
