/*
 * kvz_hip_types.h -- plain-old-data types shared by the C-ABI (include/kvz_hip.h), the
 * kvazaar-side registration shim (integration/) and the test oracle (oracle/).
 *
 * Nothing here depends on kvazaar's headers: the reference's host structs
 * (encoder_state_t, encoder_control_t, cu_info_t, lcu_t, sao_info_t, kvz_epol_args) never
 * cross the boundary.  The registration shim reads the handful of scalars a kernel needs on
 * the host side and passes them in these PODs (SURVEY.md 8b "Reference host structs in the ABI").
 */
#ifndef KVZ_HIP_TYPES_H_
#define KVZ_HIP_TYPES_H_

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef uint8_t kvz_hip_pixel;  /* kvz_pixel at KVZ_BIT_DEPTH 8 (kvazaar.h:90-98) */
typedef int16_t kvz_hip_coeff;  /* coeff_t (global.h:115) */

/* Transform selector for kvz_hip_transform*(): forward/inverse DCT 4..32 and the 4x4 DST that
 * intra luma 4x4 uses (strategies-dct.h:48-60, strategies-dct.c:78-116). */
enum kvz_hip_transform_kind {
  KVZ_HIP_DCT_4 = 0, KVZ_HIP_DCT_8, KVZ_HIP_DCT_16, KVZ_HIP_DCT_32, KVZ_HIP_DST_4,
  KVZ_HIP_IDCT_4, KVZ_HIP_IDCT_8, KVZ_HIP_IDCT_16, KVZ_HIP_IDCT_32, KVZ_HIP_IDST_4,
  KVZ_HIP_TRANSFORM_KINDS
};

/* Scalars that kvz_quant / kvz_dequant / kvz_quantize_residual read through
 * state->encoder_control, state->frame and state (quant-generic.c:50-81, 298-340). */
typedef struct kvz_hip_quant_params {
  int32_t qp;              /* state->qp */
  int32_t bitdepth;        /* encoder->bitdepth (8) */
  int32_t slice_is_intra;  /* state->frame->slicetype == KVZ_SLICE_I -> rounding 171 else 85 */
  int32_t signhide;        /* encoder->cfg.signhide_enable */
  int32_t scaling_list;    /* encoder->scaling_list.enable: 0 = flat (default), 1 = per-coefficient lists */
  int32_t cu_is_intra;     /* cur_cu->type == CU_INTRA (selects scaling list + DST for 4x4 luma) */
  /* Per-coefficient forward scales: encoder->scaling_list.quant_coeff[log2-2][list][qp%6], width*height
   * entries (always present in the reference, flat lists hold kvz_g_quant_scales[qp%6]). NULL = flat. */
  const int16_t *quant_coeff;
  /* Per-coefficient inverse scales, only read when scaling_list != 0. */
  const int16_t *dequant_coeff;
} kvz_hip_quant_params;

/* sao_info_t fields sao_reconstruct_color reads (sao.h:55-63, sao-generic.c:84-124). */
typedef struct kvz_hip_sao_params {
  int32_t type;              /* 0 none, 1 band, 2 edge (sao_type) */
  int32_t eo_class;          /* 0..3 */
  int32_t band_position[2];
  int32_t offsets[10];       /* NUM_SAO_EDGE_CATEGORIES * 2 */
  int32_t bitdepth;
} kvz_hip_sao_params;

/* kvz_epol_args without the three output pointers (strategies-ipol.h:68-93). */
typedef struct kvz_hip_epol_params {
  int32_t src_w, src_h, src_s;
  int32_t blk_x, blk_y, blk_w, blk_h;
  int32_t pad_l, pad_r, pad_t, pad_b, pad_b_simd;
} kvz_hip_epol_params;

#ifdef __cplusplus
}
#endif
#endif
