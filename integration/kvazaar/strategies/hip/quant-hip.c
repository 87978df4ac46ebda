/* strategies/hip/quant-hip.c -- strategies-quant.h:78-84.  encoder_state_t / cu_info_t are reduced to kvz_hip_quant_params
 * on the host; the device never sees a kvazaar struct. */
#include "strategies/hip/hip-common.h"

#include <stdlib.h>
#include <string.h>

#include "cabac.h"
#include "cu.h"
#include "encoder.h"
#include "encoderstate.h"
#include "rdo.h"
#include "scalinglist.h"
#include "strategies/generic/quant-generic.h"
#include "strategies/strategies-quant.h"
#include "strategyselector.h"
#include "tables.h"
#include "transform.h"

static void fill_params(const encoder_state_t *state, int width, int8_t type, int8_t block_type, kvz_hip_quant_params *p)
{
  const encoder_control_t *enc = state->encoder_control;
  const uint32_t log2_tr_size = kvz_g_convert_to_bit[width] + 2;
  const int32_t qp_scaled = kvz_get_scaled_qp(type, state->qp, (enc->bitdepth - 8) * 6);
  const int32_t list = (block_type == CU_INTRA ? 0 : 3) + (int8_t)("\0\3\1\2"[type]);
  p->qp = state->qp;
  p->bitdepth = enc->bitdepth;
  p->slice_is_intra = state->frame->slicetype == KVZ_SLICE_I;
  p->signhide = enc->cfg.signhide_enable;
  p->scaling_list = enc->scaling_list.enable;
  p->cu_is_intra = block_type == CU_INTRA;
  /* flat lists hold kvz_g_quant_scales[qp%6] in every entry (scalinglist.c:327-330): only real lists are shipped */
  p->quant_coeff = enc->scaling_list.enable ? enc->scaling_list.quant_coeff[log2_tr_size - 2][list][qp_scaled % 6] : NULL;
  p->dequant_coeff = enc->scaling_list.enable ? enc->scaling_list.de_quant_coeff[log2_tr_size - 2][list][qp_scaled % 6] : NULL;
}

static void quant_hip(const encoder_state_t *const state, coeff_t *coef, coeff_t *q_coef, int32_t width, int32_t height, int8_t type,
                      int8_t scan_idx, int8_t block_type)
{
  kvz_hip_quant_params p;
  fill_params(state, width, type, block_type, &p);
  kvz_hip_quant(&p, coef, q_coef, width, height, type, scan_idx, block_type);
}

static void dequant_hip(const encoder_state_t *const state, coeff_t *q_coef, coeff_t *coef, int32_t width, int32_t height, int8_t type,
                        int8_t block_type)
{
  kvz_hip_quant_params p;
  fill_params(state, width, type, block_type, &p);
  kvz_hip_dequant(&p, q_coef, coef, width, height, type, block_type);
}

/* The context states kvz_rdoq prices with (state->cabac, rdo.c:664-730) in the order of kvz_hip_types.h's KVZ_HIP_CX_* indices */
static void gather_rdoq_contexts(const encoder_state_t *state, uint8_t ctx[160])
{
  const cabac_data_t *cb = &state->cabac;
  memset(ctx, 0, 160);
#define PUT(at, src, n) for (int i_ = 0; i_ < (n); i_++) ctx[(at) + i_] = (src)[i_].uc_state
  PUT(KVZ_HIP_CX_CBF_LUMA, cb->ctx.qt_cbf_model_luma, 2);
  PUT(KVZ_HIP_CX_CBF_CHROMA, cb->ctx.qt_cbf_model_chroma, 2);
  PUT(KVZ_HIP_CX_CBF_CHROMA_DEEP, cb->ctx.qt_cbf_model_chroma + 2, 2);  /* tr_depth 2 / 3: the chroma blocks of an NxN CU (quant-generic.c:237-238), --tr-depth-intra */
  PUT(KVZ_HIP_CX_SIG_CG, cb->ctx.cu_sig_coeff_group_model, 4);
  PUT(KVZ_HIP_CX_SIG_LUMA, cb->ctx.cu_sig_model_luma, 27);
  PUT(KVZ_HIP_CX_SIG_CHROMA, cb->ctx.cu_sig_model_chroma, 15);
  PUT(KVZ_HIP_CX_LAST_Y_LUMA, cb->ctx.cu_ctx_last_y_luma, 15);
  PUT(KVZ_HIP_CX_LAST_Y_CHROMA, cb->ctx.cu_ctx_last_y_chroma, 15);
  PUT(KVZ_HIP_CX_LAST_X_LUMA, cb->ctx.cu_ctx_last_x_luma, 15);
  PUT(KVZ_HIP_CX_LAST_X_CHROMA, cb->ctx.cu_ctx_last_x_chroma, 15);
  PUT(KVZ_HIP_CX_ONE_LUMA, cb->ctx.cu_one_model_luma, 16);
  PUT(KVZ_HIP_CX_ONE_CHROMA, cb->ctx.cu_one_model_chroma, 8);
  PUT(KVZ_HIP_CX_ABS_LUMA, cb->ctx.cu_abs_model_luma, 4);
  PUT(KVZ_HIP_CX_ABS_CHROMA, cb->ctx.cu_abs_model_chroma, 2);
#undef PUT
}

/* quant_residual_func.  rdoq off: one fused device call.  rdoq on, intra block with flat lists / no sign hiding / no transform skip (what the all-intra presets
 * with --rdoq produce): one fused device call as well -- kvz_rdoq is host code in the reference (rdo.c:661, called directly by quant-generic.c:234-244), the library
 * carries it for these blocks (kvz_hip_quantize_residual_rdoq).  Everything else with rdoq on (inter blocks, scaling lists, sign hiding, lossless) goes to the
 * reference's own kvz_quantize_residual_generic, whose transform / quant / dequant steps come back here through the strategy pointers. */
static int quantize_residual_hip(encoder_state_t *const state, const cu_info_t *const cur_cu, const int width, const color_t color,
                                 const coeff_scan_order_t scan_order, const int use_trskip, const int in_stride, const int out_stride,
                                 const kvz_pixel *const ref_in, const kvz_pixel *const pred_in, kvz_pixel *rec_out, coeff_t *coeff_out,
                                 bool early_skip)
{
  const encoder_control_t *enc = state->encoder_control;
  kvz_hip_quant_params p;
  if (!(enc->cfg.rdoq_enable && (width > 4 || !enc->cfg.rdoq_skip)) && !enc->cfg.lossless) {
    fill_params(state, width, color == COLOR_Y ? 0 : 2, cur_cu->type, &p);
    if (enc->scaling_list.enable) {
      /* the inverse pass of a V block uses list type 3 (quant-generic.c:263) */
      const int8_t dq_type = color == COLOR_Y ? 0 : (color == COLOR_U ? 2 : 3);
      const int32_t qp_scaled = kvz_get_scaled_qp(dq_type, state->qp, (enc->bitdepth - 8) * 6);
      const int32_t list = (cur_cu->type == CU_INTRA ? 0 : 3) + (int8_t)("\0\3\1\2"[dq_type]);
      p.dequant_coeff = enc->scaling_list.de_quant_coeff[kvz_g_convert_to_bit[width]][list][qp_scaled % 6];
    }
    return kvz_hip_quantize_residual(&p, width, color, scan_order, use_trskip, in_stride, out_stride, ref_in, pred_in, rec_out,
                                     coeff_out, early_skip);
  }
  static int host_rdoq = -1;  /* KVZ_HIP_RDOQ_HOST=1: always the reference's chain with its host kvz_rdoq (A/B) */
  if (host_rdoq < 0) { const char *e = getenv("KVZ_HIP_RDOQ_HOST"); host_rdoq = e && e[0] == '1'; }
  if (!host_rdoq && cur_cu->type == CU_INTRA && !enc->scaling_list.enable && !enc->cfg.signhide_enable && !enc->cfg.lossless && !use_trskip && enc->bitdepth == 8) {
    uint8_t ctx[160];
    /* kvz_rdoq's tr_depth argument (quant-generic.c:237-238) */
    const int tr_depth = cur_cu->tr_depth - cur_cu->depth + (cur_cu->part_size == SIZE_NxN ? 1 : 0);
    gather_rdoq_contexts(state, ctx);
    fill_params(state, width, color == COLOR_Y ? 0 : 2, cur_cu->type, &p);
    return kvz_hip_quantize_residual_rdoq(&p, state->lambda, ctx, tr_depth, width, color, scan_order, in_stride, out_stride, ref_in, pred_in, rec_out, coeff_out, early_skip);
  }
  return kvz_quantize_residual_generic(state, cur_cu, width, color, scan_order, use_trskip, in_stride, out_stride, ref_in, pred_in, rec_out, coeff_out, early_skip);
}

static void find_last_scanpos_hip(coeff_t *coef, coeff_t *dest_coeff, int8_t type, int32_t q_bits, const coeff_t *quant_coeff,
                                  struct kvz_sh_rates_t *sh_rates, const uint32_t cg_size, uint16_t *ctx_set, const uint32_t *scan,
                                  int32_t *cg_last_scanpos, int32_t *last_scanpos, uint32_t cg_num, int32_t *cg_scanpos, int32_t width,
                                  int8_t scan_mode)
{
  kvz_hip_find_last_scanpos(coef, dest_coeff, type, q_bits, quant_coeff, sh_rates->sig_coeff_inc, cg_size, ctx_set, scan, cg_last_scanpos,
                            last_scanpos, cg_num, cg_scanpos, width, scan_mode);
}

int kvz_strategy_register_quant_hip(void *opaque, uint8_t bitdepth)
{
  bool success = true;
  if (!kvz_hip_strategy_usable(bitdepth)) return 1;
  success &= kvz_strategyselector_register(opaque, "quant", "hip", KVZ_HIP_PRIORITY, (void *)&quant_hip);
  success &= kvz_strategyselector_register(opaque, "quantize_residual", "hip", KVZ_HIP_PRIORITY, (void *)&quantize_residual_hip);
  success &= kvz_strategyselector_register(opaque, "dequant", "hip", KVZ_HIP_PRIORITY, (void *)&dequant_hip);
  success &= kvz_strategyselector_register(opaque, "coeff_abs_sum", "hip", KVZ_HIP_PRIORITY, (void *)&kvz_hip_coeff_abs_sum);
  success &= kvz_strategyselector_register(opaque, "fast_coeff_cost", "hip", KVZ_HIP_PRIORITY, (void *)&kvz_hip_fast_coeff_cost);
  success &= kvz_strategyselector_register(opaque, "find_last_scanpos", "hip", KVZ_HIP_PRIORITY, (void *)&find_last_scanpos_hip);
  return success;
}
