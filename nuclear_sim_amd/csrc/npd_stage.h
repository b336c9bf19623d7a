/*
 * npd_stage.h -- LDS staging pipeline of the fused step kernel (product code).
 *
 * Why: at BASELINE size (65 536 plants = 1 024 waves) every SIMD holds exactly one wave, so nothing
 * covers a wave's HBM latency but the wave itself, and all waves run the same phase at the same time
 * (measured with tools/phase_stamps.py: every "cold" section load cost 7-9 us, 41 % of the kernel was
 * SQ_WAIT_ANY).  A plant's state does not fit in registers, so the next section cannot be prefetched
 * into VGPRs either.  Instead each section is copied global -> LDS by the LDS-DMA path
 * (global_load_lds_dwordx4 / _dword: no VGPR destination, asynchronous) one phase ahead:
 *
 *   boundary between phase j and j+1:
 *     s_waitcnt vmcnt(0)         DMA(j+1) was issued a whole compute phase ago -> normally free
 *     ds_read   section j+1      LDS -> registers (conflict-free: lane l reads word l of each column)
 *     s_waitcnt lgkmcnt(0)
 *     store     section j        deferred to here so that the NEXT boundary's vmcnt(0) does not have
 *                                to wait for freshly issued stores
 *     issue DMA(j+2)             into the same LDS region, lands while phase j+1 computes
 *
 * One staging region of 36 KB per wave (4 waves per CU = 144 KB of the CU's 160 KB), addressed in column slots
 * of 64 lanes x the arena's column width.
 *
 * Arena columns (include/npb_fields.h): a section instance is NCARRY carried reals, one per column, followed by its
 * "narrow" 4-byte members -- the step's output reals as float, then the int32 members -- packed NPD_NPC to a column
 * (two per 8-byte column; one per column under fp32 storage, where every column is 4 bytes wide).
 *
 * hipcc (ROCm 7.2) does not insert the vmcnt wait between an LDS-DMA and a later ds_read of the same
 * bytes, so the waits here are explicit and carry a "memory" clobber so nothing moves across them.
 */
#ifndef NPD_STAGE_H
#define NPD_STAGE_H
#include "npd_common.h"

#define NPB_WAVE 64

/* Storage type of the fp64 state columns in HBM.  The arithmetic is fp64 either way; a build with
 * -DNPB_BUILD_F32 keeps the arena in fp32 (BASELINE config 5, "fp32 mixed precision"): half the bytes, state
 * rounded to fp32 at every store.  npb_kernels.hip is compiled once per storage type (Makefile). */
#ifdef NPB_BUILD_F32
typedef float npd_real_t;
#else
typedef double npd_real_t;
#endif
#define NPD_RB ((int)sizeof(npd_real_t))            /* bytes of a stored real */
#define NPD_SLOTB (NPB_WAVE * NPD_RB)               /* bytes of one staged real column (one wave) */
#define NPD_GROUP (16 / NPD_RB)                     /* real columns one dwordx4 LDS-DMA moves: 2 (fp64) or 4 (fp32) */
#define NPB_STAGE_BYTES 36864                       /* staging region per wave; 4 waves per CU = 144 KB of 160 */
#define NPB_STAGE_SLOTS (NPB_STAGE_BYTES / NPD_SLOTB)

#define NPD_NPC (NPD_RB / 4)                        /* narrow (4-byte) members per arena column */
#ifdef NPB_BUILD_F32
#define NPD_COL_BASE(T) NPB_##T##_COL32_BASE
#define NPD_NCOL(T) NPB_##T##_NCOL32
#else
#define NPD_COL_BASE(T) NPB_##T##_COL64_BASE
#define NPD_NCOL(T) NPB_##T##_NCOL64
#endif
#define NPD_SEC_COL(T, inst) (NPD_COL_BASE(T) + (inst) * NPD_NCOL(T))   /* first arena column of a section instance */

typedef __attribute__((address_space(1))) const void npd_gptr_t;
typedef __attribute__((address_space(3))) void npd_lptr_t;

/* Arena addresses are (one wave-uniform 64-bit base for the whole kernel) + (32-bit byte offset = column *
 * pitch + this lane's part): the SGPR-base form of global_load_lds / global_store with a 32-bit VGPR offset.
 * One s_mul + one v_add per access instead of 64-bit per-lane arithmetic and an address register pair per
 * column.  The offsets are computed by a volatile asm where they are used: as plain C the optimiser hoists
 * these one-instruction values out of the phases and keeps hundreds of them live.  32-bit offsets bound an
 * arena at 4 GiB; npb_create enforces it (about one million plants per handle). */
typedef __attribute__((address_space(1))) char npd_gchar_t;
typedef struct npd_stage_t {
  char *lds;              /* staging region base (wave-uniform) */
  npd_gchar_t *f64b;      /* the arena, offset to this wave's first plant (wave-uniform) */
  uint32_t nr;            /* column pitch in bytes */
  uint32_t laner;         /* lane * column width: this lane's plant within a column */
  uint32_t grp16;         /* grouped LDS-DMA: 64 / NPD_GROUP consecutive lanes carry one column, 16 B each:
                           * (lane / lanes_per_col) * nr + (lane % lanes_per_col) * 16 */
  double *diag;           /* this lane's element of diagnostics column 0 (include/npb.h NPB_DIAG_*), or NULL: only the
                           * diagnostics build of the step kernel sets it, everywhere else the stores below fold away */
  size_t diag_pitch;
} npd_stage_t;
/* a state store, with the non-temporal bit in the streaming build of the one-wave kernel (NPD_SM, npb_kernels.hip) -- a template
 * argument, not a run-time flag: two stores that differ only in that bit get merged into a plain one before inlining could fold */
template <bool NT, typename PTR, typename V>
__device__ __forceinline__ void npd_gstore(PTR ptr, V v) {
  if constexpr (NT) __builtin_nontemporal_store(v, ptr);
  else *ptr = v;
}
#define NPD_DIAG(st, col, v) do { if ((st).diag) (st).diag[(size_t)(col) * (st).diag_pitch] = (v); } while (0)

__device__ __forceinline__ void npd_stage_init(npd_stage_t &st, void *lds, npd_real_t *arena, size_t N, size_t block_base) {
  const uint32_t lane = threadIdx.x, lpc = NPB_WAVE / NPD_GROUP;
  st.lds = (char *)lds;
  st.f64b = (npd_gchar_t *)(arena + block_base);
  st.nr = (uint32_t)(N * NPD_RB);
  st.laner = lane * NPD_RB;
  st.grp16 = (lane / lpc) * st.nr + (lane % lpc) * 16u;
  st.diag = nullptr; st.diag_pitch = 0;
}
__device__ __forceinline__ uint32_t npd_voff(uint32_t col, uint32_t pitch, uint32_t lane_off) {
  uint32_t v, t;
  asm volatile("s_mul_i32 %1, %2, %3\n\tv_add_u32 %0, %1, %4" : "=v"(v), "=&s"(t) : "s"(col), "s"(pitch), "v"(lane_off));
  return v;
}
/* this lane's element of arena column `col`; NPD_RPO: column base + explicit byte offset; NPD_NP: the narrow
 * member `sub` (0 .. NPD_NPC-1) of this lane's element */
#define NPD_RPO(type, col, off) ((__attribute__((address_space(1))) type *)(st.f64b + npd_voff((uint32_t)(col), st.nr, (off))))
#define NPD_RP(col) NPD_RPO(npd_real_t, col, st.laner)
#define NPD_NP(type, col, sub) NPD_RPO(type, col, st.laner + (uint32_t)(sub) * 4u)

/* an 8-byte state store to this lane's element of arena column `col`, written out in the SGPR-base form the addressing above is
 * made for (measured -1 % against the compiler's own selection at 65 536 plants: profiles/r2_ab_state_store_form.txt); NT: with
 * the non-temporal bit (the streaming build) */
template <bool NT, typename V>
__device__ __forceinline__ void npd_store8(const npd_stage_t &st, uint32_t col, V v) {
  static_assert(sizeof(V) == 8, "8-byte stores only");
  const uint32_t vo = npd_voff(col, st.nr, st.laner);
  if constexpr (NT) asm volatile("global_store_dwordx2 %0, %1, %2 nt" :: "v"(vo), "v"(v), "s"(st.f64b) : "memory");
  else asm volatile("global_store_dwordx2 %0, %1, %2" :: "v"(vo), "v"(v), "s"(st.f64b) : "memory");
}

/* a carried real of this lane to arena column col.  SM, the store mode of the kernel being compiled (NPD_SM below): 0 = left to the
 * compiler (the two-wave kernels, fp32 storage), 1 = npd_store8 (the one-wave kernel: -1.2 % at 65 536 plants, where the two-wave
 * builds lose 0.3-0.9 %: profiles/r2_ab_state_store_form.txt), 2 = npd_store8 with the non-temporal bit (its streaming build) */
template <int SM>
__device__ __forceinline__ void npd_store_real(const npd_stage_t &st, int col, double v) {
#ifndef NPB_BUILD_F32
  if constexpr (SM >= 1) { npd_store8<SM == 2>(st, (uint32_t)col, v); return; }
#endif
  npd_gstore<SM == 2>(NPD_RP(col), (npd_real_t)v);
}

#ifdef NPB_STAMPS
/* diagnostic build: ticks this wave spent inside the staging pipeline's waits (lane 0 keeps the sum in LDS) */
__shared__ unsigned long long npd_wait_acc_s;
#define NPD_WAIT_ACC_INIT() do { if (threadIdx.x == 0) npd_wait_acc_s = 0; } while (0)
#define NPD_DMA_WAIT() do { unsigned long long w0__ = __builtin_readcyclecounter(); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); \
                            if (threadIdx.x == 0) npd_wait_acc_s += __builtin_readcyclecounter() - w0__; } while (0)
#else
#define NPD_WAIT_ACC_INIT()
#define NPD_DMA_WAIT() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#endif
#define NPD_LDS_DRAIN() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

/* staging slots a section occupies = its arena columns */
#define NPD_SLOTS(T) NPD_NCOL(T)

/* issue the LDS-DMA of NCOL arena columns starting at column col0 into staging slots ls ...: one dwordx4 instruction
 * moves NPD_GROUP columns (the LDS image, base + lane * 16, is those columns back to back); left-over columns go as
 * 256-B dword pieces. */
template <int NCOL>
__device__ __forceinline__ void npd_dma(const npd_stage_t &st, int col0, int ls) {
  char *l = st.lds + ls * NPD_SLOTB;
#pragma unroll
  for (int c = 0; c + NPD_GROUP <= NCOL; c += NPD_GROUP)
    __builtin_amdgcn_global_load_lds((npd_gptr_t *)NPD_RPO(char, col0 + c, st.grp16), (npd_lptr_t *)(l + c * NPD_SLOTB), 16, 0, 0);
#pragma unroll
  for (int c = NCOL / NPD_GROUP * NPD_GROUP; c < NCOL; c++)
#pragma unroll
    for (int piece = 0; piece < NPD_RB / 4; piece++)
      __builtin_amdgcn_global_load_lds((npd_gptr_t *)NPD_RPO(char, col0 + c, threadIdx.x * 4u + piece * 256),
                                       (npd_lptr_t *)(l + c * NPD_SLOTB + piece * 256), 4, 0, 0);
}
#define NPD_DMA(T, inst, ls) npd_dma<NPD_NCOL(T)>(st, NPD_SEC_COL(T, inst), ls)

/* staged carried real k / narrow member j (raw 32 bits) of the section image at slot ls, for this lane */
#define NPD_LDS_REAL(ls, k) ((double)((const npd_real_t *)(st.lds + (ls) * NPD_SLOTB))[(k) * NPB_WAVE + threadIdx.x])
#define NPD_LDS_NARROW(ls, nc, j) \
  (*(const uint32_t *)(st.lds + ((ls) + (nc) + (j) / NPD_NPC) * NPD_SLOTB + threadIdx.x * NPD_RB + ((j) % NPD_NPC) * 4))

#ifdef NPB_PROBE
/* liveness probe (diagnostic build, tools/probe_liveness.py): every loaded member passes through a NON-volatile
 * asm that carries a marker; the compiler deletes the asm when nothing uses its result, so the markers left in the
 * ISA are exactly the members the step reads.  Marker: PROBE <first fp64 slot of the section type> <member index> */
template <int SID, int K> __device__ __forceinline__ double npd_probe_f(double v) { double r; asm("v_mov_b64 %0, %1 ; PROBE_F %2 %3" : "=v"(r) : "v"(v), "n"(SID), "n"(K)); return r; }
template <int SID, int K> __device__ __forceinline__ int32_t npd_probe_i(int32_t v) { int32_t r; asm("v_mov_b32 %0, %1 ; PROBE_I %2 %3" : "=v"(r) : "v"(v), "n"(SID), "n"(K)); return r; }
template <int SID, int NF, int NI, int K> struct npd_probe_all {
  static __device__ __forceinline__ void run(double *d, int32_t *q) {
    if constexpr (K < NF) d[K] = npd_probe_f<SID, K>(d[K]);
    else q[K - NF] = npd_probe_i<SID, K - NF>(q[K - NF]);
    if constexpr (K + 1 < NF + NI) npd_probe_all<SID, NF, NI, K + 1>::run(d, q);
  }
};
#define NPD_PROBE_STRUCT(SID, NF, NI, d, q) npd_probe_all<SID, NF, NI, 0>::run(d, q)
#else
#define NPD_PROBE_STRUCT(SID, NF, NI, d, q)
#endif

/* staging slot ls -> register struct (NF fp64 members, the last NO of them outputs kept as float; then NI int32) */
template <int NF, int NO, int NI, int SID, typename S>
__device__ __forceinline__ void npd_consume(S &s, const npd_stage_t &st, int ls) {
  double *d = reinterpret_cast<double *>(&s);
  constexpr int NC = NF - NO;
#pragma unroll
  for (int k = 0; k < NC; k++) d[k] = NPD_LDS_REAL(ls, k);
#pragma unroll
  for (int j = 0; j < NO; j++) d[NC + j] = (double)__uint_as_float(NPD_LDS_NARROW(ls, NC, j));
  int32_t *q = reinterpret_cast<int32_t *>(d + NF);
#pragma unroll
  for (int k = 0; k < NI; k++) q[k] = (int32_t)NPD_LDS_NARROW(ls, NC, NO + k);
  NPD_PROBE_STRUCT(SID, NF, NI, d, q);
}
#define NPD_CONSUME(T, stype, s, ls) npd_consume<NPB_##T##_NF64, NPB_##T##_NOUT, NPB_##T##_NI32, NPB_##T##_F64_BASE, stype>(s, st, ls)

/* single staged members: fp64 member at struct index idx (carried or output), int32 member islot */
template <int NC> __device__ __forceinline__ double npd_staged_real(const npd_stage_t &st, int ls, int idx) {
  return idx < NC ? NPD_LDS_REAL(ls, idx) : (double)__uint_as_float(NPD_LDS_NARROW(ls, NC, idx - NC));
}
#define NPD_STAGED_F64(T, stype, member, k, ls) npd_staged_real<NPB_##T##_NCARRY>(st, ls, NPB_F64_SLOT(stype, member) + (k))
#define NPD_STAGED_I32(T, stype, member, ls) ((int32_t)NPD_LDS_NARROW(ls, NPB_##T##_NCARRY, NPB_##T##_NOUT + NPB_I32_SLOT(stype, T, member)))

/* fixed slot plan (see the kernel): groups that are staged together */
#define NPD_LS_PRIM 0
#define NPD_LS_SEC (NPD_LS_PRIM + NPD_SLOTS(PRIM))
#define NPD_LS_FW 0
#define NPD_LS_PUMP0 (NPD_LS_FW + NPD_SLOTS(FW))

static_assert(NPD_LS_SEC + NPD_SLOTS(SEC) <= NPB_STAGE_SLOTS, "prim + sec must fit the staging region");
static_assert(NPD_LS_PUMP0 + NPD_SLOTS(PUMP) <= NPB_STAGE_SLOTS, "fw + pump must fit the staging region");
static_assert(NPD_SLOTS(SG) <= NPB_STAGE_SLOTS && NPD_SLOTS(TURB) <= NPB_STAGE_SLOTS, "sg / turb must fit");
static_assert(NPD_SLOTS(TSTG) <= NPB_STAGE_SLOTS, "turbine stage arrays must fit the staging region");

#endif
