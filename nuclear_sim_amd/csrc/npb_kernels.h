/* npb_kernels.h -- host-callable launchers of the device kernels (internal to libnpb.so).
 * npb_kernels.hip is compiled twice: 8-byte arena columns (npb_launch_*) and 4-byte columns for fp32 storage
 * (npb32_launch_*, -DNPB_BUILD_F32); the arena pointer is void* here and typed inside each translation unit. */
#ifndef NPB_KERNELS_H
#define NPB_KERNELS_H
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/npb_params.h"
#include "../../include/npb_maint.h"
#include "../../include/npb.h"
#ifdef __cplusplus
extern "C" {
#endif
#define NPB__DECL(prefix) \
  int prefix##step(const npb_params_t *P, int n_plants, size_t npad, void *arena, const int32_t *action, \
                    const double *magnitude, const double *setpoint, const double *noise_z, const double *cw_temp, \
                    double *obs, double *reward, uint8_t *done, uint32_t *trip_flags, double *info, int variant, double *diag, size_t diag_pitch, \
                    const npb_maint_table_t *maint_table, void *maint_side, int32_t *maint_counts, hipStream_t stream); \
  void prefix##maint(size_t npad, void *arena, void *maint_side, int32_t *counts, int n_plants, hipStream_t stream); \
  void prefix##maint_consts(const npb_params_t *P, const npb_maint_table_t *T, void *host_out); \
  size_t prefix##maint_consts_bytes(void); \
  size_t prefix##maint_side_bytes(size_t npad); \
  size_t prefix##maint_cache_offset(void); \
  void prefix##observe(int mode, int n_plants, size_t npad, const void *arena, double *obs, hipStream_t stream); \
  void prefix##init(const npb_params_t *P, int n_plants, size_t npad, void *arena, const uint8_t *mask, hipStream_t stream); \
  void prefix##reset(const npb_params_t *P, int n_plants, size_t npad, void *arena, const uint8_t *mask, int steady, hipStream_t stream); \
  /* kind: 0 carried real, 1 output real (float), 2 int32; buffers: double for reals, int32 for ints */ \
  void prefix##field_get(const void *arena, size_t npad, int col, int sub, int kind, void *out, int n, hipStream_t stream); \
  void prefix##field_set(void *arena, size_t npad, int col, int sub, int kind, const void *in, int n, hipStream_t stream); \
  void prefix##gather(const void *arena, size_t npad, const int *plan_dev, int n_fields, double *out, int n, hipStream_t stream);
NPB__DECL(npb_launch_)
NPB__DECL(npb32_launch_)
#undef NPB__DECL
void npb_launch_touch(size_t npad, double *arena, hipStream_t stream);
#ifdef __cplusplus
}
#endif
#endif
