/*
 * npb_api.hip -- the C ABI of libnpb.so (include/npb.h): handle, SoA arena, field access, step launch.
 * Host code only; the kernels live in npb_kernels.hip.
 */
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include "../../include/npb.h"
#include "npb_kernels.h"

struct NpbHandle {
  npb_params_t params;
  int n_plants;
  int device;
  size_t pitch;        /* n_plants rounded up to a multiple of the wave size */
  int storage;         /* NPB_STORAGE_F64 | NPB_STORAGE_F32: element type of the real-valued columns */
  size_t real_bytes;   /* 8 | 4 */
  void *f64;           /* [NPB_TOTAL_F64][pitch] of double (or float) */
  int32_t *i32;        /* [NPB_TOTAL_I32][pitch] */
  double *convert;     /* fp32 storage only: one fp64 column used by get/set_field with host buffers */
  std::string error;
};

static thread_local std::string g_create_error;

static int fail(NpbHandle *h, int code, const char *what, hipError_t e = hipSuccess) {
  std::string msg = what;
  if (e != hipSuccess) { msg += ": "; msg += hipGetErrorString(e); }
  if (h) h->error = msg; else g_create_error = msg;
  return code;
}
#define NPB_HIP(h, call) do { hipError_t e__ = (call); if (e__ != hipSuccess) return fail(h, NPB_EHIP, #call, e__); } while (0)

extern "C" {

int npb_version(void) { return NPB_VERSION; }
int npb_num_f64(void) { return NPB_TOTAL_F64; }
int npb_num_i32(void) { return NPB_TOTAL_I32; }
size_t npb_state_bytes(void) { return (size_t)NPB_TOTAL_F64 * 8 + (size_t)NPB_TOTAL_I32 * 4; }
size_t npb_step_bytes_per_plant(void) {
  const size_t maint = (size_t)NPB_MAINT_NF64 * 8 + (size_t)NPB_MAINT_NI32 * 4; /* not touched by the step kernel */
  return 2 * (npb_state_bytes() - maint) + (4 + 4 * 8) + (NPB_OBS_DIM * 8 + 8 + 1 + 4 + NPB_INFO_DIM * 8);
}
size_t npb_handle_step_bytes_per_plant(const NpbHandle *h) {
  if (!h || h->storage == NPB_STORAGE_F64) return npb_step_bytes_per_plant();
  return npb_step_bytes_per_plant() - 2 * (size_t)(NPB_TOTAL_F64 - NPB_MAINT_NF64) * 4;
}
void npb_default_params(npb_params_t *p) { npb_params_default(p); }

const char *npb_last_error(const NpbHandle *h) { return h ? h->error.c_str() : g_create_error.c_str(); }
int npb_num_plants(const NpbHandle *h) { return h ? h->n_plants : 0; }

int npb_create(const npb_params_t *params, int n_plants, int device, NpbHandle **out) {
  return npb_create_storage(params, n_plants, device, NPB_STORAGE_F64, out);
}

int npb_storage(const NpbHandle *h) { return h ? h->storage : -1; }

int npb_create_storage(const npb_params_t *params, int n_plants, int device, int storage, NpbHandle **out) {
  if (!out || n_plants <= 0) return fail(nullptr, NPB_EINVAL, "npb_create: bad arguments");
  *out = nullptr;
  if (storage != NPB_STORAGE_F64 && storage != NPB_STORAGE_F32) return fail(nullptr, NPB_EINVAL, "npb_create: storage must be NPB_STORAGE_F64 or NPB_STORAGE_F32");
  const size_t real_bytes = storage == NPB_STORAGE_F32 ? sizeof(float) : sizeof(double);
  /* the step kernel addresses a column as (one 64-bit base) + (32-bit byte offset), nuclear_sim_amd/csrc/npd_stage.h */
  if ((((size_t)n_plants + 63) / 64 * 64) * NPB_TOTAL_F64 * real_bytes >= ((size_t)1 << 32))
    return fail(nullptr, NPB_EINVAL, "npb_create: more than 4 GiB of real-valued state per handle (about one million fp64 plants); use several handles");
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0) return fail(nullptr, NPB_EHIP, "npb_create: no HIP device", e);
  if (device < 0 || device >= ndev) return fail(nullptr, NPB_EINVAL, "npb_create: device index out of range");
  NPB_HIP(nullptr, hipSetDevice(device));
  NpbHandle *h = new NpbHandle();
  if (params) h->params = *params; else npb_params_default(&h->params);
  h->n_plants = n_plants; h->device = device;
  h->pitch = ((size_t)n_plants + 63) / 64 * 64;
  h->storage = storage; h->real_bytes = real_bytes;
  h->f64 = nullptr; h->i32 = nullptr; h->convert = nullptr;
  e = hipMalloc(&h->f64, (size_t)NPB_TOTAL_F64 * h->pitch * real_bytes);
  if (e == hipSuccess) e = hipMalloc((void **)&h->i32, (size_t)NPB_TOTAL_I32 * h->pitch * sizeof(int32_t));
  if (e == hipSuccess && storage == NPB_STORAGE_F32) e = hipMalloc((void **)&h->convert, h->pitch * sizeof(double));
  if (e != hipSuccess) {
    if (h->f64) (void)hipFree(h->f64);
    if (h->i32) (void)hipFree(h->i32);
    delete h;
    return fail(nullptr, NPB_ENOMEM, "npb_create: hipMalloc of the state arena failed", e);
  }
  (storage == NPB_STORAGE_F32 ? npb32_launch_init : npb_launch_init)(&h->params, n_plants, h->pitch, h->f64, h->i32, nullptr, nullptr);
  e = hipDeviceSynchronize();
  if (e != hipSuccess) {
    (void)hipFree(h->f64); (void)hipFree(h->i32); if (h->convert) (void)hipFree(h->convert);
    delete h; return fail(nullptr, NPB_EHIP, "npb_create: init kernel failed", e);
  }
  *out = h;
  return NPB_OK;
}

int npb_destroy(NpbHandle *h) {
  if (!h) return NPB_OK;
  (void)hipSetDevice(h->device);
  (void)hipFree(h->f64); (void)hipFree(h->i32);
  if (h->convert) (void)hipFree(h->convert);
  delete h;
  return NPB_OK;
}

int npb_set_params(NpbHandle *h, const npb_params_t *params) {
  if (!h || !params) return NPB_EINVAL;
  h->params = *params;
  return NPB_OK;
}

int npb_reset(NpbHandle *h, const uint8_t *mask, void *stream) {
  if (!h) return NPB_EINVAL;
  NPB_HIP(h, hipSetDevice(h->device));
  (h->storage == NPB_STORAGE_F32 ? npb32_launch_init : npb_launch_init)(&h->params, h->n_plants, h->pitch, h->f64, h->i32, mask, (hipStream_t)stream);
  NPB_HIP(h, hipGetLastError());
  return NPB_OK;
}

static int field_ptr(NpbHandle *h, int kind, int slot, void **col, size_t *bytes) {
  if (kind == NPB_KIND_F64) {
    if (slot < 0 || slot >= NPB_TOTAL_F64) return fail(h, NPB_EINVAL, "field slot out of range");
    *col = (char *)h->f64 + (size_t)slot * h->pitch * h->real_bytes; *bytes = (size_t)h->n_plants * sizeof(double);
  } else if (kind == NPB_KIND_I32) {
    if (slot < 0 || slot >= NPB_TOTAL_I32) return fail(h, NPB_EINVAL, "field slot out of range");
    *col = h->i32 + (size_t)slot * h->pitch; *bytes = (size_t)h->n_plants * sizeof(int32_t);
  } else {
    return fail(h, NPB_EINVAL, "field kind must be NPB_KIND_F64 or NPB_KIND_I32");
  }
  return NPB_OK;
}

int npb_get_field(NpbHandle *h, int kind, int slot, void *buf, int buf_is_device, void *stream) {
  if (!h || !buf) return NPB_EINVAL;
  void *col; size_t bytes;
  int rc = field_ptr(h, kind, slot, &col, &bytes);
  if (rc) return rc;
  NPB_HIP(h, hipSetDevice(h->device));
  if (kind == NPB_KIND_F64 && h->storage == NPB_STORAGE_F32) { /* the ABI speaks fp64: widen the column on the device */
    double *wide = buf_is_device ? (double *)buf : h->convert;
    npb32_launch_col_to_f64((const float *)col, wide, h->n_plants, (hipStream_t)stream);
    NPB_HIP(h, hipGetLastError());
    if (buf_is_device) return NPB_OK;
    col = wide;
  }
  NPB_HIP(h, hipMemcpyAsync(buf, col, bytes, buf_is_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, (hipStream_t)stream));
  if (!buf_is_device) NPB_HIP(h, hipStreamSynchronize((hipStream_t)stream));
  return NPB_OK;
}

int npb_set_field(NpbHandle *h, int kind, int slot, const void *buf, int buf_is_device, void *stream) {
  if (!h || !buf) return NPB_EINVAL;
  void *col; size_t bytes;
  int rc = field_ptr(h, kind, slot, &col, &bytes);
  if (rc) return rc;
  NPB_HIP(h, hipSetDevice(h->device));
  if (kind == NPB_KIND_F64 && h->storage == NPB_STORAGE_F32) { /* round the fp64 values to the stored type on the device */
    const double *wide = (const double *)buf;
    if (!buf_is_device) {
      NPB_HIP(h, hipMemcpyAsync(h->convert, buf, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
      wide = h->convert;
    }
    npb32_launch_col_from_f64((float *)col, wide, h->n_plants, (hipStream_t)stream);
    NPB_HIP(h, hipGetLastError());
    if (!buf_is_device) NPB_HIP(h, hipStreamSynchronize((hipStream_t)stream));
    return NPB_OK;
  }
  NPB_HIP(h, hipMemcpyAsync(col, buf, bytes, buf_is_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, (hipStream_t)stream));
  if (!buf_is_device) NPB_HIP(h, hipStreamSynchronize((hipStream_t)stream));
  return NPB_OK;
}

int npb_state_arena(NpbHandle *h, double **f64, int32_t **i32, size_t *pitch) {
  if (!h) return NPB_EINVAL;
  if (h->storage != NPB_STORAGE_F64) return fail(h, NPB_EINVAL, "npb_state_arena: fp32-storage handle; use npb_state_arena_raw");
  if (f64) *f64 = (double *)h->f64;
  if (i32) *i32 = h->i32;
  if (pitch) *pitch = h->pitch;
  return NPB_OK;
}

int npb_state_arena_raw(NpbHandle *h, void **real, int32_t **i32, size_t *pitch, int *storage) {
  if (!h) return NPB_EINVAL;
  if (real) *real = h->f64;
  if (i32) *i32 = h->i32;
  if (pitch) *pitch = h->pitch;
  if (storage) *storage = h->storage;
  return NPB_OK;
}

int npb_step(NpbHandle *h, const int32_t *action, const double *magnitude, const double *power_setpoint,
             const double *noise_z, const double *cooling_water_temp, double *obs, double *reward, uint8_t *done,
             uint32_t *trip_flags, double *info, void *stream) {
  if (!h) return NPB_EINVAL;
  const bool narrow = h->storage == NPB_STORAGE_F32;
  (narrow ? npb32_launch_step : npb_launch_step)(&h->params, h->n_plants, h->pitch, h->f64, h->i32, action, magnitude, power_setpoint,
                                                 noise_z, cooling_water_temp, obs, reward, done, trip_flags, info, (hipStream_t)stream);
  if (h->params.maint_enabled) (narrow ? npb32_launch_maint : npb_launch_maint)(&h->params, h->pitch, h->f64, h->i32, (hipStream_t)stream);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(h, NPB_EHIP, "npb_step: kernel launch failed", e);
  return NPB_OK;
}

int npb_debug_touch(NpbHandle *h, void *stream) {
  if (!h) return NPB_EINVAL;
  if (h->storage != NPB_STORAGE_F64) return fail(h, NPB_EINVAL, "npb_debug_touch: fp64-storage handles only");
  npb_launch_touch(h->pitch, (double *)h->f64, h->i32, (hipStream_t)stream);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(h, NPB_EHIP, "npb_debug_touch: kernel launch failed", e);
  return NPB_OK;
}

int npb_observe(NpbHandle *h, double *obs, void *stream) {
  if (!h || !obs) return NPB_EINVAL;
  (h->storage == NPB_STORAGE_F32 ? npb32_launch_observe : npb_launch_observe)(h->params.mode, h->n_plants, h->pitch, h->f64, h->i32, obs, (hipStream_t)stream);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(h, NPB_EHIP, "npb_observe: kernel launch failed", e);
  return NPB_OK;
}

} /* extern "C" */
