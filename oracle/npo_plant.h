/*
 * npo_plant.h -- CPU oracle: one plant's state record (array-of-structs view of the
 * schema in include/npb_fields.h).  TEST INFRASTRUCTURE ONLY (see npo_common.h).
 */
#ifndef NPO_PLANT_H
#define NPO_PLANT_H
#include "npo_common.h"

/* member names must match the first column of NPB_SECTIONS */
typedef struct npo_plant_t {
  npb_prim_t prim;
  npb_sg_t sg[NPB_NUM_SG];
  npb_pump_t pump[NPB_NUM_PUMPS];
  npb_fw_t fw;
  npb_turb_t turb;
  npb_tstg_t tstg;
  npb_chem_t chem[2];
  npb_ph_t ph;
  npb_cond_t cond;
  npb_sec_t sec;
  npb_mpump_t mpump[NPB_NUM_PUMPS];
  npb_maint_t maint;
} npo_plant_t;

/* generic slot access (global fp64 / int32 slot numbering of npb_fields.h) */
NPO_FN double *npo_f64_slot(npo_plant_t *pl, int slot) {
#define NPO__S(member, T, stype, count) \
  if (slot >= NPB_##T##_F64_BASE && slot < NPB_##T##_F64_BASE + (count) * NPB_##T##_NF64) { \
    int rel = slot - NPB_##T##_F64_BASE; \
    return (double *)((char *)&pl->member + (size_t)(rel / NPB_##T##_NF64) * sizeof(stype)) + rel % NPB_##T##_NF64; \
  }
  NPB_SECTIONS(NPO__S)
#undef NPO__S
  return 0;
}
NPO_FN int32_t *npo_i32_slot(npo_plant_t *pl, int slot) {
#define NPO__S(member, T, stype, count) \
  if (NPB_##T##_NI32 > 0 && slot >= NPB_##T##_I32_BASE && slot < NPB_##T##_I32_BASE + (count) * NPB_##T##_NI32) { \
    int rel = slot - NPB_##T##_I32_BASE; \
    return (int32_t *)((char *)&pl->member + (size_t)(rel / (NPB_##T##_NI32 ? NPB_##T##_NI32 : 1)) * sizeof(stype) + \
                       (size_t)NPB_##T##_NF64 * sizeof(double)) + rel % (NPB_##T##_NI32 ? NPB_##T##_NI32 : 1); \
  }
  NPB_SECTIONS(NPO__S)
#undef NPO__S
  return 0;
}

#endif
