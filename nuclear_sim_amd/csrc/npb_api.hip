/*
 * npb_api.hip -- the C ABI of libnpb.so (include/npb.h): handle, SoA arena, field access, step launch.
 * Host code only; the kernels live in npb_kernels.hip.
 */
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include "../../include/npb.h"
#include "npb_kernels.h"

struct NpbHandle {
  npb_params_t params;
  int n_plants;
  int device;
  size_t pitch;        /* n_plants rounded up to a multiple of the wave size */
  size_t seg;          /* plants per arena segment (npb_kernels.hip, "segmented arena"), 0 = the arena is one [column][pitch] block */
  int storage;         /* NPB_STORAGE_F64 | NPB_STORAGE_F32: element type of the real-valued columns */
  size_t real_bytes;   /* 8 | 4 */
  void *f64;           /* the arena: [NPB_TOTAL_COL64][pitch] 8-byte columns, or [NPB_TOTAL_COL32][pitch] 4-byte ones */
  double *convert;     /* one staging column (pitch doubles) used by get/set_field with host buffers */
  void *maint_side;      /* automatic maintenance: the rule's constants as the device reads them + the screen's cooldown cache (npb_kernels.hip; behind the staging column) */
  std::vector<char> maint_consts_host;
  double *diag; size_t diag_pitch; /* npb_set_diagnostics: the caller's [NPB_DIAG_DIM][diag_pitch] buffer, or NULL */
  int32_t *maint_counts;           /* npb_set_maintenance_count_buffer: the caller's [n_plants] int32 column, or NULL */
  bool maint_cache_stale;          /* the cooldown cache of the step kernels' maintenance screen must be zeroed before the next step */
  int last_kernel;                 /* NPB_KERNEL_*: what the last npb_step launched */
  int step_kernel;                 /* 0 = chosen by batch size, 1 = one-wave kernel, 2 = two-wave kernel, 3 = its two-waves-per-SIMD build, 4 = one-wave with streaming stores, 5 = four-wave kernel (npb_set_step_kernel) */
  npb_maint_table_t maint_table;   /* thresholds of the automatic maintenance (include/npb_maint.h) */
  bool maint_table_custom;         /* set through npb_set_maintenance_table: the table is then taken as it is */
  int *plan_dev;       /* npb_gather_fields: {column, sub, kind} per requested field, and the request it was built for */
  std::vector<int> plan_key;
  std::string error;
};

static thread_local std::string g_create_error;

static int fail(NpbHandle *h, int code, const char *what, hipError_t e = hipSuccess) {
  std::string msg = what;
  if (e != hipSuccess) { msg += ": "; msg += hipGetErrorString(e); }
  if (h) h->error = msg; else g_create_error = msg;
  return code;
}
/* every entry point that launches makes the handle's device current for the duration of the call and puts the
 * caller's device back afterwards: a process may hold handles on several GPUs, and a launch on whichever device
 * happened to be current would run against another device's arena.  hipGetDevice is a thread-local read;
 * hipSetDevice is paid only when the caller was elsewhere. */
struct DeviceGuard {
  int prev = -1; bool switched = false; hipError_t err = hipSuccess;
  explicit DeviceGuard(int device) {
    err = hipGetDevice(&prev);
    if (err == hipSuccess && prev != device) { err = hipSetDevice(device); switched = err == hipSuccess; }
  }
  ~DeviceGuard() { if (switched) (void)hipSetDevice(prev); }
};
#define NPB_USE_DEVICE(h) DeviceGuard guard__((h)->device); if (guard__.err != hipSuccess) return fail(h, NPB_EHIP, "hipSetDevice", guard__.err)
#define NPB_HIP(h, call) do { hipError_t e__ = (call); if (e__ != hipSuccess) return fail(h, NPB_EHIP, #call, e__); } while (0)

/* where the members of the schema live in the arena (include/npb_fields.h: carried fp64 members one per column,
 * then the narrow members -- outputs as float, int32 -- two per 8-byte column or one per 4-byte column) */
struct SectionInfo { int f64_base, nf64, nout, i32_base, ni32, count, col64_base, ncol64, col32_base, ncol32; };
static const SectionInfo g_sections[] = {
#define NPB__INFO(member, T, stype, count) \
  {NPB_##T##_F64_BASE, NPB_##T##_NF64, NPB_##T##_NOUT, NPB_##T##_I32_BASE, NPB_##T##_NI32, (count), \
   NPB_##T##_COL64_BASE, NPB_##T##_NCOL64, NPB_##T##_COL32_BASE, NPB_##T##_NCOL32},
  NPB_SECTIONS(NPB__INFO)
#undef NPB__INFO
};
/* kind: NPB_KIND_F64 / NPB_KIND_I32 and the global slot -> arena column, narrow position, access kind of the field kernels */
static bool locate(int storage, int kind, int slot, int *col, int *sub, int *akind) {
  const int npc = storage == NPB_STORAGE_F32 ? 1 : 2;
  for (const SectionInfo &s : g_sections) {
    const int base = kind == NPB_KIND_F64 ? s.f64_base : s.i32_base, per = kind == NPB_KIND_F64 ? s.nf64 : s.ni32;
    if (per == 0 || slot < base || slot >= base + per * s.count) continue;
    const int inst = (slot - base) / per, k = (slot - base) % per, ncarry = s.nf64 - s.nout;
    const int col0 = (storage == NPB_STORAGE_F32 ? s.col32_base + inst * s.ncol32 : s.col64_base + inst * s.ncol64);
    if (kind == NPB_KIND_F64 && k < ncarry) { *col = col0 + k; *sub = 0; *akind = 0; return true; }
    const int j = kind == NPB_KIND_F64 ? k - ncarry : s.nout + k;
    *col = col0 + ncarry + j / npc; *sub = j % npc; *akind = kind == NPB_KIND_F64 ? 1 : 2;
    return true;
  }
  return false;
}
static size_t arena_columns(int storage) { return storage == NPB_STORAGE_F32 ? (size_t)NPB_TOTAL_COL32 : (size_t)NPB_TOTAL_COL64; }
/* plants the arena has room for: whole segments when it is segmented */
static size_t arena_plants(const NpbHandle *h) { return h->seg ? (h->pitch + h->seg - 1) / h->seg * h->seg : h->pitch; }
/* what the launchers take as the column pitch: the pitch with the segment size in the upper half (npb_kernels.hip, NPD_SEGMENT) */
#define NPB_N(h) ((size_t)(h)->pitch | ((size_t)(h)->seg << 32))

/* Where an arena lands in physical memory changes the step kernel's time when the bytes a step touches are about the
 * size of the 256 MB Infinity Cache (65 536 fp64 plants: 276 MB): handles created one after another in one process run
 * at 0.097 ms or at 0.110 ms per step, reproducibly per handle for its whole life, and nothing cheaper than the step
 * kernel itself predicts which (tools/launch_series.py).  So in that zone npb_create allocates up to
 * four candidate arenas, times a dozen launches of the step kernel on each (construction state, no inputs, no
 * outputs), keeps the fastest and frees the rest; the kept arena is initialised again by the caller below.  About 15 ms,
 * once per handle.  NPB_PLACEMENT_PROBE=0 turns it off. */
static void probe_placement(NpbHandle *h, size_t step_columns) {
  const char *env = getenv("NPB_PLACEMENT_PROBE");
  if (env && atoi(env) == 0) return;
  const double touched_mb = (double)step_columns * h->real_bytes * h->pitch / 1.0e6;
  if (touched_mb < 200.0 || touched_mb > 340.0 || h->params.mode == NPB_MODE_PRIMARY) return;
  const bool narrow = h->storage == NPB_STORAGE_F32;
  const size_t bytes = arena_columns(h->storage) * arena_plants(h) * h->real_bytes;
  const int max_candidates = 4, launches = 12;
  void *cand[max_candidates] = {h->f64, nullptr, nullptr, nullptr};
  float ms[max_candidates] = {0, 0, 0, 0};
  hipEvent_t a, b;
  if (hipEventCreate(&a) != hipSuccess) return;
  if (hipEventCreate(&b) != hipSuccess) { (void)hipEventDestroy(a); return; }
  /* the clocks first: after an idle period the step kernel needs ~170 launches to reach its steady time (bench.py,
   * "preconditioning"), and the first candidate would otherwise be timed on the ramp -- a 5-7 % bias against it, half of
   * the effect being selected on.  Untimed launches on candidate 0 until ~20 ms have passed. */
  (narrow ? npb32_launch_init : npb_launch_init)(&h->params, h->n_plants, NPB_N(h), cand[0], nullptr, nullptr);
  for (int k = 0; k < 200; k++)
    (void)(narrow ? npb32_launch_step : npb_launch_step)(&h->params, h->n_plants, NPB_N(h), cand[0], nullptr, nullptr, nullptr, nullptr, nullptr,
                                                         nullptr, nullptr, nullptr, nullptr, nullptr, h->step_kernel, nullptr, 0, nullptr, nullptr, nullptr, nullptr);
  if (hipDeviceSynchronize() != hipSuccess) { (void)hipGetLastError(); (void)hipEventDestroy(a); (void)hipEventDestroy(b); return; }
  int n = 0;
  for (; n < max_candidates; n++) {
    if (n > 0 && hipMalloc(&cand[n], bytes) != hipSuccess) { cand[n] = nullptr; (void)hipGetLastError(); break; }
    (narrow ? npb32_launch_init : npb_launch_init)(&h->params, h->n_plants, NPB_N(h), cand[n], nullptr, nullptr);
    float best = 1e30f;
    for (int k = 0; k < launches; k++) {
      (void)hipEventRecord(a, nullptr);
      (void)(narrow ? npb32_launch_step : npb_launch_step)(&h->params, h->n_plants, NPB_N(h), cand[n], nullptr, nullptr, nullptr, nullptr, nullptr,
                                                           nullptr, nullptr, nullptr, nullptr, nullptr, h->step_kernel, nullptr, 0, nullptr, nullptr, nullptr, nullptr);
      (void)hipEventRecord(b, nullptr);
      if (hipEventSynchronize(b) != hipSuccess) { best = 1e30f; break; }
      float t = 0;
      if (hipEventElapsedTime(&t, a, b) != hipSuccess) { best = 1e30f; break; }
      if (k >= 4 && t < best) best = t;     /* the first launches warm the caches */
    }
    ms[n] = best;
  }
  /* nothing of a candidate may still be in flight when it is freed (an event wait that failed above leaves that unknown) */
  (void)hipDeviceSynchronize();
  int keep = 0;
  for (int i = 1; i < n; i++) if (ms[i] < ms[keep]) keep = i;
  for (int i = 0; i < n; i++) if (i != keep && cand[i]) (void)hipFree(cand[i]);
  h->f64 = cand[keep];
  (void)hipEventDestroy(a); (void)hipEventDestroy(b);
}

extern "C" {

int npb_version(void) { return NPB_VERSION; }
int npb_num_f64(void) { return NPB_TOTAL_F64; }
int npb_num_i32(void) { return NPB_TOTAL_I32; }
int npb_obs_dim(void) { return NPB_OBS_DIM; }
int npb_info_dim(void) { return NPB_INFO_DIM; }
int npb_info_nrho(void) { return NPB_INFO_NRHO; }
int npb_diag_dim(void) { return NPB_DIAG_DIM; }
static const char *const g_maint_params[] = {
#define NPB__X(id, name) name,
  NPB_MAINT_PARAMS(NPB__X)
#undef NPB__X
};
static const struct { const char *name; int handler; } g_maint_actions[] = {
#define NPB__X(id, name, handler) {name, handler},
  NPB_MAINT_ACTIONS(NPB__X)
#undef NPB__X
};
int npb_maint_num_params(void) { return NPB_MAINT_NPARAM; }
int npb_maint_num_actions(void) { return NPB_MAINT_NACT; }
const char *npb_maint_param_name(int k) { return k >= 0 && k < NPB_MAINT_NPARAM ? g_maint_params[k] : nullptr; }
const char *npb_maint_action_name(int a) { return a >= 0 && a < NPB_MAINT_NACT ? g_maint_actions[a].name : nullptr; }
int npb_maint_action_has_handler(int a) { return a >= 0 && a < NPB_MAINT_NACT ? g_maint_actions[a].handler : 0; }
static_assert(sizeof(g_maint_params) / sizeof(g_maint_params[0]) == NPB_MAINT_NPARAM, "parameter catalog");
static_assert(sizeof(g_maint_actions) / sizeof(g_maint_actions[0]) == NPB_MAINT_NACT, "action catalog");
size_t npb_state_bytes(void) { return (size_t)NPB_TOTAL_COL64 * 8; }
/* carried fp64 members are read and written, int32 members too, output members are only written (as float);
 * the maint.* section belongs to the maintenance kernel */
static size_t step_bytes(size_t real_bytes, bool kinetics = true) {
  size_t carried = 0, outputs = 0, ints = 0;
  for (const SectionInfo &s : g_sections) {
    if (s.f64_base == NPB_MAINT_F64_BASE || s.f64_base == NPB_MPUMP_F64_BASE) continue;
    carried += (size_t)(s.nf64 - s.nout) * s.count; outputs += (size_t)s.nout * s.count; ints += (size_t)s.ni32 * s.count;
  }
  if (!kinetics) carried -= NPB_PRIM_NKIN; /* ConstantHeatSource: the point-kinetics columns are not touched */
  return 2 * carried * real_bytes + 2 * ints * 4 + outputs * 4 + (4 + 4 * 8) + (NPB_OBS_DIM * 8 + 8 + 1 + 4 + NPB_INFO_DIM * 8);
}
size_t npb_step_bytes_per_plant(void) { return step_bytes(8); }
size_t npb_handle_step_bytes_per_plant(const NpbHandle *h) {
  return step_bytes(h && h->storage == NPB_STORAGE_F32 ? 4 : 8, !h || h->params.heat_source == NPB_HEAT_REACTOR);
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
  /* the step kernel addresses a column as (one 64-bit base) + (32-bit byte offset), nuclear_sim_amd/csrc/npd_stage.h;
   * the maintenance sections behind its columns are addressed with 64-bit arithmetic by their own kernel */
  const size_t step_columns = storage == NPB_STORAGE_F32 ? (size_t)NPB_MPUMP_COL32_BASE : (size_t)NPB_MPUMP_COL64_BASE;
  if ((((size_t)n_plants + 63) / 64 * 64) * step_columns * real_bytes >= ((size_t)1 << 32))
    return fail(nullptr, NPB_EINVAL, "npb_create: more than 4 GiB of real-valued state per handle (about one million fp64 plants); use several handles");
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0) return fail(nullptr, NPB_EHIP, "npb_create: no HIP device", e);
  if (device < 0 || device >= ndev) return fail(nullptr, NPB_EINVAL, "npb_create: device index out of range");
  int caller_device = -1;
  (void)hipGetDevice(&caller_device);
  NPB_HIP(nullptr, hipSetDevice(device));
  NpbHandle *h = new NpbHandle();
  if (params) h->params = *params; else npb_params_default(&h->params);
  npb_maint_table_default(&h->maint_table);
  h->maint_table_custom = false;
  { const char *e = getenv("NPB_STEP_KERNEL"); h->step_kernel = e ? atoi(e) : 0; if (h->step_kernel < 0 || h->step_kernel > 5) h->step_kernel = 0; }
  h->n_plants = n_plants; h->device = device;
  h->pitch = ((size_t)n_plants + 63) / 64 * 64;
  h->storage = storage; h->real_bytes = real_bytes;
  h->f64 = nullptr; h->convert = nullptr; h->plan_dev = nullptr;
  /* batches past the two-wave kernel's range keep their arena in segments of 16 384 plants (npb_kernels.hip, "segmented arena") */
  {   /* NPB_ARENA_SEGMENT=0 turns it off, =<plants> (a multiple of 64) forces that segment size at any batch size: A/B aids */
    const char *e2 = getenv("NPB_ARENA_SEGMENT");
    h->seg = h->pitch > 45056 ? 16384 : 0;     /* from NPB_SEGMENTED_FROM of npb_kernels.hip on: the four-wave kernel's second range and everything above it */
    if (e2) { const long v = atol(e2); h->seg = (v > 0 && v % 64 == 0 && (size_t)v < h->pitch) ? (size_t)v : 0; }
  }
  e = hipMalloc(&h->f64, arena_columns(storage) * arena_plants(h) * real_bytes);
  if (e == hipSuccess) probe_placement(h, step_columns);
  if (e == hipSuccess) e = hipMalloc((void **)&h->convert, h->pitch * sizeof(double) + npb_launch_maint_side_bytes(h->pitch));
  if (e != hipSuccess) {
    if (h->f64) (void)hipFree(h->f64);
    delete h;
    if (caller_device >= 0) (void)hipSetDevice(caller_device);
    return fail(nullptr, NPB_ENOMEM, "npb_create: hipMalloc of the state arena failed", e);
  }
  h->maint_side = (void *)(h->convert + h->pitch);
  h->maint_cache_stale = true;
  (storage == NPB_STORAGE_F32 ? npb32_launch_init : npb_launch_init)(&h->params, n_plants, NPB_N(h), h->f64, nullptr, nullptr);
  e = hipDeviceSynchronize();
  if (caller_device >= 0 && caller_device != device) (void)hipSetDevice(caller_device); /* the caller's current device is left as it was */
  if (e != hipSuccess) {
    (void)hipSetDevice(device);
    (void)hipFree(h->f64); if (h->convert) (void)hipFree(h->convert);
    if (caller_device >= 0) (void)hipSetDevice(caller_device);
    delete h; return fail(nullptr, NPB_EHIP, "npb_create: init kernel failed", e);
  }
  *out = h;
  return NPB_OK;
}

int npb_destroy(NpbHandle *h) {
  if (!h) return NPB_OK;
  DeviceGuard guard__(h->device);
  (void)hipFree(h->f64);
  if (h->convert) (void)hipFree(h->convert);
  if (h->plan_dev) (void)hipFree(h->plan_dev);
  delete h;
  return NPB_OK;
}

int npb_set_params(NpbHandle *h, const npb_params_t *params) {
  if (!h || !params) return NPB_EINVAL;
  h->params = *params;
  h->maint_cache_stale = true;
  return NPB_OK;
}

int npb_set_step_kernel(NpbHandle *h, int variant) {
  if (!h || variant < 0 || variant > 5) return NPB_EINVAL;
  h->step_kernel = variant;
  return NPB_OK;
}

int npb_debug_last_step_kernel(const NpbHandle *h) { return h ? h->last_kernel : NPB_KERNEL_NONE; }
const char *npb_step_kernel_name(int id) {
  static const char *const names[NPB_KERNEL_COUNT_] = {"", "npb_step_kernel", "npb_step2_wide_kernel", "npb_step2_kernel", "npb_step_nt_kernel",
                                                       "npb_step_diag_kernel", "npb_step_primary_kernel", "npb_step_maint_kernel", "npb_step2_wide_maint_kernel",
                                                       "npb_step2_maint_kernel", "npb_step_nt_maint_kernel", "npb_step4_kernel", "npb_step4_maint_kernel"};
  return id >= 0 && id < NPB_KERNEL_COUNT_ ? names[id] : nullptr;
}

int npb_set_diagnostics(NpbHandle *h, double *buf, size_t pitch) {
  if (!h) return NPB_EINVAL;
  if (buf && h->params.mode != NPB_MODE_FULL) return fail(h, NPB_EINVAL, "npb_set_diagnostics: full mode only");
  if (buf && pitch < h->pitch) return fail(h, NPB_EINVAL, "npb_set_diagnostics: pitch must be at least n_plants rounded up to 64");
  h->diag = buf; h->diag_pitch = buf ? pitch : 0;
  return NPB_OK;
}

int npb_set_maintenance_table(NpbHandle *h, const npb_maint_table_t *table) {
  if (!h || !table) return NPB_EINVAL;
  for (int k = 0; k < NPB_MAINT_NPARAM; k++) {
    if (table->rank[k] < 0) continue;
    if (table->action[k] < 0 || table->action[k] >= NPB_MAINT_NACT || table->comparison[k] < 0 || table->comparison[k] > NPB_CMP_NOT_EQUALS ||
        table->priority[k] < NPB_PRIO_LOW || table->priority[k] > NPB_PRIO_EMERGENCY || table->bearing[k] < 0 || table->bearing[k] > NPB_BEARING_THRUST)
      return fail(h, NPB_EINVAL, "npb_set_maintenance_table: action / comparison / priority / bearing code out of range");
  }
  h->maint_table = *table;
  h->maint_table_custom = true;
  h->maint_cache_stale = true;
  return NPB_OK;
}
int npb_set_maintenance_count_buffer(NpbHandle *h, int32_t *counts) {
  if (!h) return NPB_EINVAL;
  h->maint_counts = counts;
  h->maint_cache_stale = true;     /* filled whole before the next step */
  return NPB_OK;
}
void npb_default_maintenance_table(npb_maint_table_t *table) { if (table) npb_maint_table_default(table); }

int npb_reset(NpbHandle *h, const uint8_t *mask, void *stream) {
  if (!h) return NPB_EINVAL;
  NPB_USE_DEVICE(h);
  h->maint_cache_stale = true;
  (h->storage == NPB_STORAGE_F32 ? npb32_launch_init : npb_launch_init)(&h->params, h->n_plants, NPB_N(h), h->f64, mask, (hipStream_t)stream);
  NPB_HIP(h, hipGetLastError());
  return NPB_OK;
}

int npb_reset_reference(NpbHandle *h, const uint8_t *mask, int start_at_steady_state, void *stream) {
  if (!h) return NPB_EINVAL;
  NPB_USE_DEVICE(h);
  h->maint_cache_stale = true;
  (h->storage == NPB_STORAGE_F32 ? npb32_launch_reset : npb_launch_reset)(&h->params, h->n_plants, NPB_N(h), h->f64, mask, start_at_steady_state != 0, (hipStream_t)stream);
  NPB_HIP(h, hipGetLastError());
  return NPB_OK;
}

/* one member of every plant <-> a contiguous buffer: a small gather / scatter kernel (members share columns and
 * outputs are stored as float, so this is never a plain copy); host buffers go through the staging column */
static int field_args(NpbHandle *h, int kind, int slot, int *col, int *sub, int *akind, size_t *bytes) {
  if (kind != NPB_KIND_F64 && kind != NPB_KIND_I32) return fail(h, NPB_EINVAL, "field kind must be NPB_KIND_F64 or NPB_KIND_I32");
  if (!locate(h->storage, kind, slot, col, sub, akind)) return fail(h, NPB_EINVAL, "field slot out of range");
  *bytes = (size_t)h->n_plants * (kind == NPB_KIND_F64 ? sizeof(double) : sizeof(int32_t));
  return NPB_OK;
}

int npb_get_field(NpbHandle *h, int kind, int slot, void *buf, int buf_is_device, void *stream) {
  if (!h || !buf) return NPB_EINVAL;
  int col, sub, akind; size_t bytes;
  int rc = field_args(h, kind, slot, &col, &sub, &akind, &bytes);
  if (rc) return rc;
  NPB_USE_DEVICE(h);
  void *dst = buf_is_device ? buf : (void *)h->convert;
  (h->storage == NPB_STORAGE_F32 ? npb32_launch_field_get : npb_launch_field_get)(h->f64, NPB_N(h), col, sub, akind, dst, h->n_plants, (hipStream_t)stream);
  NPB_HIP(h, hipGetLastError());
  if (!buf_is_device) {
    NPB_HIP(h, hipMemcpyAsync(buf, dst, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    NPB_HIP(h, hipStreamSynchronize((hipStream_t)stream));
  }
  return NPB_OK;
}

int npb_set_field(NpbHandle *h, int kind, int slot, const void *buf, int buf_is_device, void *stream) {
  if (!h || !buf) return NPB_EINVAL;
  int col, sub, akind; size_t bytes;
  int rc = field_args(h, kind, slot, &col, &sub, &akind, &bytes);
  if (rc) return rc;
  NPB_USE_DEVICE(h);
  h->maint_cache_stale = true;      /* a stamp, a pump member or the clock may just have been written */
  const void *src = buf;
  if (!buf_is_device) {
    NPB_HIP(h, hipMemcpyAsync(h->convert, buf, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    src = h->convert;
  }
  (h->storage == NPB_STORAGE_F32 ? npb32_launch_field_set : npb_launch_field_set)(h->f64, NPB_N(h), col, sub, akind, src, h->n_plants, (hipStream_t)stream);
  NPB_HIP(h, hipGetLastError());
  if (!buf_is_device) NPB_HIP(h, hipStreamSynchronize((hipStream_t)stream));
  return NPB_OK;
}

int npb_gather_fields(NpbHandle *h, int n_fields, const int *kinds, const int *slots, double *out, void *stream) {
  if (!h || !kinds || !slots || !out || n_fields <= 0 || n_fields > NPB_TOTAL_F64 + NPB_TOTAL_I32) return NPB_EINVAL;
  NPB_USE_DEVICE(h);
  std::vector<int> key(2 * (size_t)n_fields);
  for (int f = 0; f < n_fields; f++) { key[2 * f] = kinds[f]; key[2 * f + 1] = slots[f]; }
  if (key != h->plan_key) { /* a log asks for the same members every time: build and upload the plan once */
    std::vector<int> plan(3 * (size_t)n_fields);
    for (int f = 0; f < n_fields; f++)
      if ((kinds[f] != NPB_KIND_F64 && kinds[f] != NPB_KIND_I32) || !locate(h->storage, kinds[f], slots[f], &plan[3 * f], &plan[3 * f + 1], &plan[3 * f + 2]))
        return fail(h, NPB_EINVAL, "npb_gather_fields: bad field kind or slot");
    if (!h->plan_dev) NPB_HIP(h, hipMalloc((void **)&h->plan_dev, sizeof(int) * 3 * (NPB_TOTAL_F64 + NPB_TOTAL_I32)));
    NPB_HIP(h, hipMemcpyAsync(h->plan_dev, plan.data(), sizeof(int) * plan.size(), hipMemcpyHostToDevice, (hipStream_t)stream));
    NPB_HIP(h, hipStreamSynchronize((hipStream_t)stream));
    h->plan_key = key;
  }
  (h->storage == NPB_STORAGE_F32 ? npb32_launch_gather : npb_launch_gather)(h->f64, NPB_N(h), h->plan_dev, n_fields, out, h->n_plants, (hipStream_t)stream);
  NPB_HIP(h, hipGetLastError());
  return NPB_OK;
}

int npb_state_arena(NpbHandle *h, void **arena, size_t *pitch, int *storage) {
  if (!h) return NPB_EINVAL;
  if (h->seg && arena) {      /* column * pitch + plant is NOT where a plant's element is on this arena: say so instead of handing out a pointer */
    return fail(h, NPB_EINVAL, "npb_state_arena: this handle's arena is segmented (npb_state_arena_segment plants per segment): column * pitch + plant is "
                               "not where a plant's element is; use npb_state_arena_layout, which reports the segment size with the pointer");
  }
  if (arena) { *arena = h->f64; h->maint_cache_stale = true; }      /* the caller may write the arena through this pointer (before the next step) */
  if (pitch) *pitch = h->seg ? h->seg : h->pitch;
  if (storage) *storage = h->storage;
  return NPB_OK;
}

int npb_state_arena_layout(NpbHandle *h, void **arena, size_t *pitch, size_t *segment, int *columns, int *storage) {
  if (!h) return NPB_EINVAL;
  if (arena) { *arena = h->f64; h->maint_cache_stale = true; }      /* a query of the layout alone (arena = NULL) leaves the maintenance cache alone */
  if (pitch) *pitch = h->seg ? h->seg : h->pitch;
  if (segment) *segment = h->seg;
  if (columns) *columns = h->storage == NPB_STORAGE_F32 ? (int)NPB_TOTAL_COL32 : (int)NPB_TOTAL_COL64;
  if (storage) *storage = h->storage;
  return NPB_OK;
}

size_t npb_state_arena_segment(const NpbHandle *h) { return h ? h->seg : 0; }

int npb_locate_field(const NpbHandle *h, int kind, int slot, int *column, int *sub, int *access) {
  if (!h) return NPB_EINVAL;
  int c, s2, a;
  if ((kind != NPB_KIND_F64 && kind != NPB_KIND_I32) || !locate(h->storage, kind, slot, &c, &s2, &a)) return NPB_EINVAL;
  if (column) *column = c;
  if (sub) *sub = s2;
  if (access) *access = a;
  return NPB_OK;
}

int npb_step(NpbHandle *h, const int32_t *action, const double *magnitude, const double *power_setpoint,
             const double *noise_z, const double *cooling_water_temp, double *obs, double *reward, uint8_t *done,
             uint32_t *trip_flags, double *info, void *stream) {
  if (!h) return NPB_EINVAL;
  if (h->params.heat_source == NPB_HEAT_EXTERNAL && !noise_z)      /* a NULL column would read as 0 MW thermal, silently */
    return fail(h, NPB_EINVAL, "npb_step: params.heat_source is NPB_HEAT_EXTERNAL, whose thermal power arrives in the noise_z column (include/npb_params.h): it must not be NULL");
  NPB_USE_DEVICE(h);
  const bool narrow = h->storage == NPB_STORAGE_F32;
  npb_maint_table_t table;
  const bool maint = h->params.maint_enabled != 0;
  if (maint) {
    table = h->maint_table;
    if (!h->maint_table_custom) {   /* with the default table the two oil_level params of ABI version 1 still set their row */
      table.threshold[NPB_MP_OIL_LEVEL] = h->params.maint_oil_level_threshold;
      table.cooldown_hours[NPB_MP_OIL_LEVEL] = h->params.maint_oil_level_cooldown_hours;
    }
    if (h->maint_cache_stale) {
      /* parameters, table, state or clock may have changed since the last step: the rule's constants go to the device anew and
       * the screen's cooldown cache is zeroed (= nothing known: every wave is looked at once and its entries rebuilt) */
      h->maint_consts_host.resize(npb_launch_maint_consts_bytes());
      npb_launch_maint_consts(&h->params, &table, h->maint_consts_host.data());
      NPB_HIP(h, hipMemcpyAsync(h->maint_side, h->maint_consts_host.data(), h->maint_consts_host.size(), hipMemcpyHostToDevice, (hipStream_t)stream));
      NPB_HIP(h, hipStreamSynchronize((hipStream_t)stream));      /* the host copy may change again before an asynchronous copy would read it */
      NPB_HIP(h, hipMemsetAsync((char *)h->maint_side + npb_launch_maint_cache_offset(), 0, npb_launch_maint_side_bytes(h->pitch) - npb_launch_maint_cache_offset(),
                                (hipStream_t)stream));
      if (h->maint_counts) {       /* the caller's event-count column: whole once, then kept by the rule for the plants whose count it moves */
        int col, sub, akind;
        if (locate(h->storage, NPB_KIND_I32, NPB_MAINT_I32_BASE + NPB_I32_SLOT(npb_maint_t, MAINT, maintenance_actions_performed), &col, &sub, &akind))
          (narrow ? npb32_launch_field_get : npb_launch_field_get)(h->f64, NPB_N(h), col, sub, akind, h->maint_counts, h->n_plants, (hipStream_t)stream);
      }
      h->maint_cache_stale = false;
    }
  }
  h->last_kernel = (narrow ? npb32_launch_step : npb_launch_step)(&h->params, h->n_plants, NPB_N(h), h->f64, action, magnitude, power_setpoint,
                                                 noise_z, cooling_water_temp, obs, reward, done, trip_flags, info, h->step_kernel, h->diag, h->diag_pitch,
                                                 maint ? &table : nullptr, maint ? h->maint_side : nullptr, maint ? h->maint_counts : nullptr, (hipStream_t)stream);
  if (maint && h->params.mode != NPB_MODE_FULL)   /* a full-mode step kernel has run the rule itself, for the waves whose pump phase found something */
    (narrow ? npb32_launch_maint : npb_launch_maint)(NPB_N(h), h->f64, h->maint_side, h->maint_counts, h->n_plants, (hipStream_t)stream);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(h, NPB_EHIP, "npb_step: kernel launch failed", e);
  return NPB_OK;
}

int npb_debug_touch(NpbHandle *h, void *stream) {
  if (!h) return NPB_EINVAL;
  if (h->storage != NPB_STORAGE_F64) return fail(h, NPB_EINVAL, "npb_debug_touch: fp64-storage handles only");
  NPB_USE_DEVICE(h);
  npb_launch_touch(NPB_N(h), (double *)h->f64, (hipStream_t)stream);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(h, NPB_EHIP, "npb_debug_touch: kernel launch failed", e);
  return NPB_OK;
}

int npb_observe(NpbHandle *h, double *obs, void *stream) {
  if (!h || !obs) return NPB_EINVAL;
  NPB_USE_DEVICE(h);
  (h->storage == NPB_STORAGE_F32 ? npb32_launch_observe : npb_launch_observe)(h->params.mode, h->n_plants, NPB_N(h), h->f64, obs, (hipStream_t)stream);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(h, NPB_EHIP, "npb_observe: kernel launch failed", e);
  return NPB_OK;
}

} /* extern "C" */
