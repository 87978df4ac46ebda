// kvz_inter_host.hpp -- host side of the inter CTU pass (kvz_inter_ctu.hpp): the per-picture model (lambda, quantisation scalars, the B slice's context
// initialisation, kvz_init_contexts context.c:202-305 with row 0 of the tables :36-193).
#pragma once
#include <math.h>
#include <string.h>

#include "kvz_inter_ctu.hpp"
#include "kvz_tables.hpp"

namespace kvz {

inline int inter_ctx_state(int qp, int init_value)  // context.c:202-213 kvz_ctx_init
{
  const int slope = (init_value >> 4) * 5 - 45, offset = ((init_value & 15) << 3) - 16;
  int st = ((slope * qp) >> 4) + offset;
  st = st < 1 ? 1 : (st > 126 ? 126 : st);
  return st >= 64 ? ((st - 64) << 1) + 1 : (63 - st) << 1;
}

// Row 0 (B slices) of the residual coder's initialisation tables (context.c:111-193: INIT_SIG_CG_FLAG, INIT_SIG_FLAG, INIT_LAST, INIT_ONE_FLAG, INIT_ABS_FLAG), entry
// KVZ_HIP_CX_x - KVZ_HIP_CX_SIG_CG for context KVZ_HIP_CX_x: the layout both the entropy coder's context set and the search contexts (ICtx from IX_RES on) keep
inline void b_slice_residual_init_values(uint8_t v[KVZ_HIP_CX_ABS_CHROMA + 2 - KVZ_HIP_CX_SIG_CG])
{
  static const uint8_t sig_cg[4] = { 121, 140, 61, 154 };
  static const uint8_t sig[42] = { 170, 154, 139, 153, 139, 123, 123, 63, 124, 166, 183, 140, 136, 153, 154, 166, 183, 140, 136, 153, 154, 166, 183, 140, 136, 153, 154,
                                   170, 153, 138, 138, 122, 121, 122, 121, 167, 151, 183, 140, 151, 183, 140 };
  static const uint8_t last[30] = { 125, 110, 124, 110, 95, 94, 125, 111, 111, 79, 125, 126, 111, 111, 79, 108, 123, 93, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154 };
  static const uint8_t one[24] = { 154, 196, 167, 167, 154, 152, 167, 182, 182, 134, 149, 136, 153, 121, 136, 122, 169, 208, 166, 167, 154, 152, 167, 182 };
  static const uint8_t absf[6] = { 107, 167, 91, 107, 107, 167 };
  const int b = KVZ_HIP_CX_SIG_CG;
  for (int i = 0; i < 4; i++) v[KVZ_HIP_CX_SIG_CG - b + i] = sig_cg[i];
  for (int i = 0; i < 27; i++) v[KVZ_HIP_CX_SIG_LUMA - b + i] = sig[i];
  for (int i = 0; i < 15; i++) {
    v[KVZ_HIP_CX_SIG_CHROMA - b + i] = sig[27 + i];
    v[KVZ_HIP_CX_LAST_Y_LUMA - b + i] = v[KVZ_HIP_CX_LAST_X_LUMA - b + i] = last[i];
    v[KVZ_HIP_CX_LAST_Y_CHROMA - b + i] = v[KVZ_HIP_CX_LAST_X_CHROMA - b + i] = last[15 + i];
  }
  for (int i = 0; i < 16; i++) v[KVZ_HIP_CX_ONE_LUMA - b + i] = one[i];
  for (int i = 0; i < 8; i++) v[KVZ_HIP_CX_ONE_CHROMA - b + i] = one[16 + i];
  for (int i = 0; i < 4; i++) v[KVZ_HIP_CX_ABS_LUMA - b + i] = absf[i];
  for (int i = 0; i < 2; i++) v[KVZ_HIP_CX_ABS_CHROMA - b + i] = absf[4 + i];
}

inline void inter_model_init(InterModel *m, int qp, int poc, uint64_t coeff_weights, const float fbits[128], int mv_constraint, int sao, int deblock, int fme_level,
                             int pu_depth_inter_max, int no_wpp, int fast_residual_cost, int pic_w = 0, int pic_h = 0, int ref_w = 0, int ref_h = 0, int tile_x = 0, int tile_y = 0, int no_tmvp = 0)
{
  memset(m, 0, sizeof *m);
  m->qp = qp; m->poc = poc;
  m->lambda = 0.57 * pow(2.0, (qp - 12) / 3.0);  // rate_control.c:678-691
  m->lambda_sqrt = sqrt(m->lambda);
  m->coeff_weights = coeff_weights;
  m->coeff_cabac = !(qp < fast_residual_cost && qp < 50);  // rdo.c:311-340: cfg.fast_residual_cost_limit (28 `ultrafast` .. `veryfast`, 0 `faster`), MAX_FAST_COEFF_COST_QP
  m->ref_w = ref_w > 0 ? ref_w : pic_w; m->ref_h = ref_h > 0 ? ref_h : pic_h; m->tile_x = ref_w > 0 ? tile_x : 0; m->tile_y = ref_h > 0 ? tile_y : 0;
  m->no_tmvp = no_tmvp;
  m->mv_constraint = mv_constraint; m->sao = sao; m->deblock = deblock; m->fme_level = fme_level; m->pu_depth_inter_max = pu_depth_inter_max; m->no_wpp = no_wpp;
  uint8_t init[IX_COUNT];
  memset(init, 154, sizeof init);
  const uint8_t split[3] = { 107, 139, 126 }, skip[3] = { 197, 185, 201 }, inter_dir[5] = { 95, 79, 63, 31, 31 };
  for (int i = 0; i < 3; i++) { init[IX_SPLIT + i] = split[i]; init[IX_SKIP + i] = skip[i]; }
  init[IX_MERGE_FLAG] = 154; init[IX_MERGE_IDX] = 137; init[IX_PRED_MODE] = 134; init[IX_PART] = 154; init[IX_INTRA] = 183; init[IX_CHROMA] = 152;
  init[IX_CBF_LUMA] = 153; init[IX_CBF_LUMA + 1] = 111; init[IX_CBF_CHROMA] = 149; init[IX_CBF_CHROMA + 1] = 92;
  init[IX_MVD] = 169; init[IX_MVD + 1] = 198; init[IX_MVP_IDX] = 168;
  for (int i = 0; i < 5; i++) init[IX_INTER_DIR + i] = inter_dir[i];
  init[IX_ROOT_CBF] = 79;
  b_slice_residual_init_values(init + IX_RES);
  for (int i = 0; i < IX_COUNT; i++) m->ctx_init[i] = (uint8_t)inter_ctx_state(qp, init[i]);
  for (int l2 = 2; l2 <= 5; l2++)
    for (int c = 0; c < 2; c++) {
      m->qf[c][l2 - 2] = quant_scalars(qp, 8, 0 /* B slice: rounding 85 */, 0, 1 << l2, c ? 2 : 0);
      m->qi[c][l2 - 2] = quant_scalars(qp, 8, 0, 0, 1 << l2, c ? 2 : 0);
    }
  memcpy(m->fbits, fbits, sizeof m->fbits);
}

}  // namespace kvz
