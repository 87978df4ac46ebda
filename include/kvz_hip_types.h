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

/* Bit-cost model of the batched all-intra CTU pass (kvz_hip_batch.h).  kvazaar prices syntax elements with
 * CTX_ENTROPY_FBITS(ctx, val) = kvz_f_entropy_bits[ctx->uc_state ^ val] (cabac.h:131) on a copy of the row's CABAC contexts that
 * adapts while it searches (search.c:1211, encode_coding_tree.c:948).  With adaptive != 0 (what kvz_hip_intra_cost_model_init
 * sets) the pass does the same: the ten contexts the all-intra ultrafast search touches are updated by the mock encode of every
 * CU the search evaluates, with the save / restore points of search_cu (search.c:655-1060), then by the real syntax of the
 * finished CTU in coding order (encode_coding_tree.c:745), and handed from CTU to CTU along a row and from the second CTU of a
 * row to the first of the row below (WPP, encoderstate.c:763-771) -- the reconstruction is then kvazaar's own, picture for
 * picture (tests/test_encoder_parity.py), at every QP: below fast_residual_cost_limit (cfg.c: 28 in `ultrafast`) with the fast
 * coefficient cost, from it on with the residual coder in counting mode (coeff_cabac).  adaptive == 0 keeps every context at its
 * slice-start state: CTUs then depend on each other through pixels and CU info only; a valid encode, not kvazaar's. */
/* Index of each context in kvz_hip_intra_cost_model::ctx_init (cabac.h:63-100): CU / transform-tree syntax, then the residual-coding
 * contexts, which only matter when coefficients are priced with the CABAC model (coeff_cabac). */
enum {
  KVZ_HIP_CX_SPLIT = 0,          /* split_flag_model[0..2] */
  KVZ_HIP_CX_PART = 3,           /* part_size_model[0] */
  KVZ_HIP_CX_INTRA = 4,          /* intra_mode_model */
  KVZ_HIP_CX_CHROMA = 5,         /* chroma_pred_model[0] */
  KVZ_HIP_CX_CBF_LUMA = 6,       /* qt_cbf_model_luma[0..1] */
  KVZ_HIP_CX_CBF_CHROMA = 8,     /* qt_cbf_model_chroma[0..1]; [2..3] are KVZ_HIP_CX_CBF_CHROMA_DEEP below */
  KVZ_HIP_CX_SIG_CG = 10,        /* cu_sig_coeff_group_model[0..3] */
  KVZ_HIP_CX_SIG_LUMA = 14,      /* cu_sig_model_luma[0..26] */
  KVZ_HIP_CX_SIG_CHROMA = 41,    /* cu_sig_model_chroma[0..14] */
  KVZ_HIP_CX_LAST_Y_LUMA = 56,   /* cu_ctx_last_y_luma[0..14] */
  KVZ_HIP_CX_LAST_Y_CHROMA = 71, /* cu_ctx_last_y_chroma[0..14] */
  KVZ_HIP_CX_LAST_X_LUMA = 86,   /* cu_ctx_last_x_luma[0..14] */
  KVZ_HIP_CX_LAST_X_CHROMA = 101,/* cu_ctx_last_x_chroma[0..14] */
  KVZ_HIP_CX_ONE_LUMA = 116,     /* cu_one_model_luma[0..15] */
  KVZ_HIP_CX_ONE_CHROMA = 132,   /* cu_one_model_chroma[0..7] */
  KVZ_HIP_CX_ABS_LUMA = 140,     /* cu_abs_model_luma[0..3] */
  KVZ_HIP_CX_ABS_CHROMA = 144,   /* cu_abs_model_chroma[0..1] */
  /* qt_cbf_model_chroma[2..3] (cabac.h:75): nothing in these configurations ever CODES a chroma coded-block flag at transform depth 2 or 3, but kvz_rdoq PRICES
   * the flag of a chroma block on qt_cbf_model_chroma[tr_depth] (rdo.c:907-915), and the blocks of an NxN CU arrive with tr_depth 2 (quant-generic.c:237-238) */
  KVZ_HIP_CX_CBF_CHROMA_DEEP = 146,
  KVZ_HIP_CX_COUNT = 148,        /* contexts of the CTU pass */
  /* the two contexts of the SAO syntax (encoderstate.c:467-552), used by the SAO parameter decision only */
  KVZ_HIP_CX_SAO_MERGE = 148,    /* sao_merge_flag_model */
  KVZ_HIP_CX_SAO_TYPE = 149      /* sao_type_idx_model */
};

typedef struct kvz_hip_intra_cost_model {
  /* sizeof(kvz_hip_intra_cost_model) of the headers the CALLER was compiled against (kvz_hip_intra_cost_model_init sets it).  The struct grows at the end between
   * versions: an entry point that is handed a size it does not know returns -1 instead of reading whatever lies behind a shorter struct. */
  uint32_t struct_size;
  double   lambda;            /* state->lambda: 0.57 * 2^((qp-12)/3) at constant QP (rate_control.c:678-691) */
  double   lambda_sqrt;       /* state->lambda_sqrt */
  /* fbits[val] of each context at its slice-start state (= entropy_fbits[ctx_init[i] ^ val]; informational, the pass prices
   * from ctx_init / entropy_fbits): */
  float    split_flag[3][2];  /* ctx.split_flag_model[0..2]      (search.c:952-956, encode_coding_tree.c:985-997) */
  float    part_size[2];      /* ctx.part_size_model[0]          (encode_coding_tree.c:695-703) */
  float    intra_mode[2];     /* ctx.intra_mode_model            (search_intra.c:641-676) */
  float    chroma_mode[2];    /* ctx.chroma_pred_model[0]        (search_intra.c:679-690) */
  float    cbf_luma[2][2];    /* ctx.qt_cbf_model_luma[0..1]     (search.c:489-497) */
  float    cbf_chroma[2][2];  /* ctx.qt_cbf_model_chroma[0..1]   (search.c:463-470) */
  uint64_t coeff_weights;     /* kvz_fast_coeff_get_weights(state): 4 x Q8.8 (fast_coeff_cost.c:84-88) */
  int32_t  qp;                /* state->qp (constant over the frame) */
  int32_t  adaptive;          /* see above */
  /* != 0: coefficients are priced by running the residual coder in counting mode (get_coeff_cabac_cost, rdo.c:220-263) instead of
   * the fast estimate (kvz_fast_coeff_cost); what kvazaar does for QP >= fast_residual_cost_limit (28 in `ultrafast`), rdo.c:311-340 */
  int32_t  coeff_cabac;
  /* != 0: no wavefront parallel processing -- one coder runs through the picture in raster order, so the first CTU of a row takes its
   * contexts from the last CTU of the row above instead of from its second one.  kvazaar's default is WPP on (0 here), except that it
   * switches WPP off when tiles are requested (cfg.c:925-978).  Rows then form one serial chain per picture: the batch needs as many
   * pictures (or tiles) in flight as the device has workgroup slots. */
  int32_t  no_wpp;
  /* != 0: CUs of 32x32 are searched like the 16x16 and 8x8 ones (kvazaar's --pu-depth-intra 1-3, preset `fast`): rough search + reconstruction of the
   * 32x32 CU first, then -- unless it has no coefficients (cu-split-termination zero, search.c:975-984) -- its four 16x16 children with early termination.
   * 0: --pu-depth-intra 2-3 (ultrafast ... faster): 32x32 CUs only arise by merging four 16x16 CUs under the top-left one's mode (search.c:996-1044). */
  int32_t  search_32x32;
  /* != 0: every transform block is quantised by rate-distortion optimised quantisation (kvazaar's --rdoq with --rdoq-skip 0 and --signhide 0, preset `medium`;
   * kvz_rdoq, rdo.c:661-1000, on the contexts of the row's real coder as they stand when the CTU's search begins: state->cabac) instead of kvz_quant.  Needs
   * coeff_cabac (the presets that switch RDOQ on have --fast-residual-cost 0) and search_32x32 is independent of it. */
  int32_t  rdoq;
  /* != 0: 8x8 CUs are also tried as four 4x4 prediction units (part_size NxN: kvazaar's --pu-depth-intra ..-4, preset `medium`) -- one more level of search_cu's
   * recursion (search.c:691, 794, 970-974): each PU with its own rough search, DST, reconstruction and most probable modes at 4x4 granularity, chroma once per CU
   * under the first PU's mode.  Results: kvz_hip_batch_download_partitions (cu_mode keeps the first PU's mode). */
  int32_t  search_nxn;
  uint8_t  ctx_init[160];     /* uc_state at slice start (kvz_init_contexts, context.c:202-305) of the KVZ_HIP_CX_* contexts; the rest unused */
  float    entropy_fbits[128];/* kvz_f_entropy_bits (rdo.c:69-83) */
} kvz_hip_intra_cost_model;

/* Per-CTU result record of the batched pass: what kvazaar keeps in cu_array / lcu_t for the CTU. */
#define KVZ_HIP_CTU_COEFFS 6144 /* 64*64 Y + 32*32 U + 32*32 V coefficients, each plane in lcu_t z-order (cu.h:385-421) */

#ifdef __cplusplus
}
#endif
#endif
