/*
 * ref_shim.c -- TEST INFRASTRUCTURE.  Flat C wrappers around the REFERENCE's own strategy function
 * pointers (libkvazaar_ref.so built from /root/reference by oracle/Makefile), so that tests can call
 * the reference with the same plain-pointer signatures as oracle/kvz_oracle.h and include/kvz_hip.h.
 *
 * This file is OUR code but it includes the reference's headers, so it only compiles where
 * /root/reference exists (this container); the product of the build, oracle/_ref/libkvz_refshim.so,
 * travels to the GPU box like every other file of oracle/_ref/.
 *
 * kvz_ref_select(0) binds the generic strategies (cpuid off, strategyselector.c:456),
 * kvz_ref_select(1) the best available ones (AVX2 on this host).
 */
#include <string.h>
#include <stdlib.h>

#include "global.h"
#include "encoder.h"
#include "encoderstate.h"
#include "cu.h"
#include "sao.h"
#include "scalinglist.h"
#include "strategyselector.h"
#include "fast_coeff_cost.h"
#include "rdo.h"
#include "tables.h"
#include "transform.h"
#include "image.h"

#include "../include/kvz_hip_types.h"

static encoder_control_t g_ctrl;
static encoder_state_t g_state;
static encoder_state_config_frame_t g_frame;
static int g_init = 0;

int kvz_ref_select(int cpuid)
{
  if (!g_init) {
    memset(&g_ctrl, 0, sizeof(g_ctrl));
    memset(&g_state, 0, sizeof(g_state));
    memset(&g_frame, 0, sizeof(g_frame));
    g_ctrl.bitdepth = KVZ_BIT_DEPTH;
    kvz_scalinglist_init(&g_ctrl.scaling_list);
    kvz_scalinglist_process(&g_ctrl.scaling_list, KVZ_BIT_DEPTH);
    kvz_fast_coeff_use_default_table(&g_ctrl.fast_coeff_table);
    g_state.encoder_control = &g_ctrl;
    g_state.frame = &g_frame;
    g_init = 1;
  }
  return kvz_strategyselector_init(cpuid, KVZ_BIT_DEPTH, 0);
}

static void set_state(const kvz_hip_quant_params *p)
{
  g_state.qp = (int8_t)p->qp;
  g_frame.slicetype = p->slice_is_intra ? KVZ_SLICE_I : KVZ_SLICE_P;
  g_ctrl.cfg.signhide_enable = p->signhide;
  g_ctrl.cfg.rdoq_enable = 0;
}

/* exported reference tables (for pinning the oracle's generated tables) */
const int16_t *kvz_ref_dct_matrix(int n)
{
  extern const int16_t kvz_g_dct_4[4][4], kvz_g_dct_8[8][8], kvz_g_dct_16[16][16], kvz_g_dct_32[32][32];
  return n == 4 ? &kvz_g_dct_4[0][0] : n == 8 ? &kvz_g_dct_8[0][0] : n == 16 ? &kvz_g_dct_16[0][0] : &kvz_g_dct_32[0][0];
}
const int16_t *kvz_ref_dst_matrix(void) { extern const int16_t kvz_g_dst_4[4][4]; return &kvz_g_dst_4[0][0]; }
const uint32_t *kvz_ref_scan_table(int scan_idx, int log2_size) { return kvz_g_sig_last_scan[scan_idx][log2_size - 1]; }
uint64_t kvz_ref_fast_coeff_weights(int qp) { return g_ctrl.fast_coeff_table.wts_by_qp[qp]; }
const int16_t *kvz_ref_quant_coeff(int log2_size, int list, int qp_rem) { return g_ctrl.scaling_list.quant_coeff[log2_size - 2][list][qp_rem]; }
const int16_t *kvz_ref_dequant_coeff(int log2_size, int list, int qp_rem) { return g_ctrl.scaling_list.de_quant_coeff[log2_size - 2][list][qp_rem]; }
/* The scaling lists of the encoder control every wrapper below runs on (encoder.c:257-311): mode 0 = --scaling-list off (flat), 1 = --scaling-list default
 * (enable + use_default_list), 2 = custom lists -- coeff[size][list][64] (16 used at size 0) and dc[size][list] as kvz_scalinglist_parse would leave them, stored
 * as coeff_t directly (scalinglist.c:202 writes the parsed values through an int32_t pointer into coeff_t storage, so a cqm FILE never reaches
 * kvz_scalinglist_process intact; the processing itself -- scalinglist.c:289-425 -- is what is pinned here).  Lists are re-processed on every call. */
void kvz_ref_set_scaling_list(int mode, const int16_t *coeff, const int32_t *dc)
{
  scaling_list_t *sl = &g_ctrl.scaling_list;
  sl->enable = mode != 0;
  sl->use_default_list = mode == 1;
  for (int size = 0; size < 4; size++) {
    for (int list = 0; list < (size == 3 ? 2 : 6); list++) {
      coeff_t *dst = (coeff_t *)sl->scaling_list_coeff[size][list];
      const int n = size == 0 ? 16 : 64;
      for (int i = 0; i < n; i++) dst[i] = (mode == 2) ? coeff[(size * 6 + list) * 64 + i] : 0;
      sl->scaling_list_dc[size][list] = (mode == 2) ? dc[size * 6 + list] : 0;
    }
  }
  kvz_scalinglist_process(sl, KVZ_BIT_DEPTH);
}
float kvz_ref_entropy_fbits(int i) { extern const float kvz_f_entropy_bits[128]; return kvz_f_entropy_bits[i]; }

/* ---- picture ---- */
unsigned kvz_ref_reg_sad(const uint8_t *d1, const uint8_t *d2, int w, int h, unsigned s1, unsigned s2) { return kvz_reg_sad(d1, d2, w, h, s1, s2); }
unsigned kvz_ref_sad_nxn(int n, const uint8_t *b1, const uint8_t *b2) { return kvz_pixels_get_sad_func(n)(b1, b2); }
unsigned kvz_ref_satd_nxn(int n, const uint8_t *b1, const uint8_t *b2) { return kvz_pixels_get_satd_func(n)(b1, b2); }
void kvz_ref_sad_nxn_dual(int n, const uint8_t *preds, const uint8_t *orig, unsigned num_modes, unsigned *costs)
{ kvz_pixels_get_sad_dual_func(n)((const pred_buffer)preds, orig, num_modes, costs); }
void kvz_ref_satd_nxn_dual(int n, const uint8_t *preds, const uint8_t *orig, unsigned num_modes, unsigned *costs)
{ kvz_pixels_get_satd_dual_func(n)((const pred_buffer)preds, orig, num_modes, costs); }
unsigned kvz_ref_satd_any_size(int w, int h, const uint8_t *b1, int s1, const uint8_t *b2, int s2) { return kvz_satd_any_size(w, h, b1, s1, b2, s2); }
void kvz_ref_satd_any_size_quad(int w, int h, const uint8_t *const *preds, int stride, const uint8_t *orig, int orig_stride,
                                unsigned num_modes, unsigned *costs, int8_t *valid)
{ kvz_satd_any_size_quad(w, h, (const kvz_pixel **)preds, stride, orig, orig_stride, num_modes, costs, valid); }
unsigned kvz_ref_pixels_calc_ssd(const uint8_t *ref, const uint8_t *rec, int rs, int cs, int width) { return kvz_pixels_calc_ssd(ref, rec, rs, cs, width); }
uint32_t kvz_ref_ver_sad(const uint8_t *pic, const uint8_t *ref, int32_t bw, int32_t bh, uint32_t ps) { return kvz_ver_sad(pic, ref, bw, bh, ps); }
uint32_t kvz_ref_hor_sad(const uint8_t *pic, const uint8_t *ref, int32_t w, int32_t h, uint32_t ps, uint32_t rs, uint32_t left, uint32_t right)
{ return kvz_hor_sad(pic, ref, w, h, ps, rs, left, right); }
double kvz_ref_pixel_var(const uint8_t *buf, uint32_t len) { return kvz_pixel_var(buf, len); }
/* optimized SAD for a width, or UINT32_MAX when the strategy has none (generic returns NULL) */
uint32_t kvz_ref_optimized_sad(int width, const uint8_t *pic, const uint8_t *ref, int32_t height, uint32_t s1, uint32_t s2)
{
  optimized_sad_func_ptr_t f = kvz_get_optimized_sad(width);
  return f ? f(pic, ref, height, s1, s2) : 0xffffffffu;
}

/* One plane of bipred_average through the real lcu_t-based entry point: the plane is placed at the
 * luma slot of a scratch lcu_t/yuv_t and the result copied back. */
void kvz_ref_bipred_average_plane(uint8_t *dst, unsigned dst_stride, const uint8_t *px0, const int16_t *im0,
                                  const uint8_t *px1, const int16_t *im1, unsigned w, unsigned h)
{
  static lcu_t lcu;
  yuv_t p0, p1; yuv_im_t i0, i1;
  memset(&p0, 0, sizeof p0); memset(&p1, 0, sizeof p1); memset(&i0, 0, sizeof i0); memset(&i1, 0, sizeof i1);
  p0.y = (kvz_pixel *)px0; p1.y = (kvz_pixel *)px1; i0.y = (kvz_pixel_im *)im0; i1.y = (kvz_pixel_im *)im1;
  kvz_bipred_average(&lcu, &p0, &p1, &i0, &i1, 0, 0, w, h, im0 ? 1 : 0, im1 ? 1 : 0, true, false);
  for (unsigned y = 0; y < h; y++) memcpy(dst + y * dst_stride, lcu.rec.y + y * LCU_WIDTH, w);
}

/* image.c:407 kvz_image_calc_sad on ad-hoc kvz_picture views (only y/width/height/stride are read) */
unsigned kvz_ref_image_calc_sad(const uint8_t *pic, int pic_stride, const uint8_t *ref, int ref_w, int ref_h,
                                int ref_stride, int pic_x, int pic_y, int ref_x, int ref_y, int bw, int bh)
{
  kvz_picture p, r;
  memset(&p, 0, sizeof p); memset(&r, 0, sizeof r);
  p.y = (kvz_pixel *)pic; p.stride = pic_stride; p.width = 1 << 20; p.height = 1 << 20;
  r.y = (kvz_pixel *)ref; r.stride = ref_stride; r.width = ref_w; r.height = ref_h;
  return kvz_image_calc_sad(&p, &r, pic_x, pic_y, ref_x, ref_y, bw, bh, kvz_get_optimized_sad(bw));
}

/* ---- dct ---- */
void kvz_ref_transform(int kind, int8_t bitdepth, const int16_t *in, int16_t *out)
{
  dct_func *f[KVZ_HIP_TRANSFORM_KINDS] = { kvz_dct_4x4, kvz_dct_8x8, kvz_dct_16x16, kvz_dct_32x32, kvz_fast_forward_dst_4x4,
                                           kvz_idct_4x4, kvz_idct_8x8, kvz_idct_16x16, kvz_idct_32x32, kvz_fast_inverse_dst_4x4 };
  f[kind](bitdepth, in, out);
}

/* ---- quant ---- */
void kvz_ref_quant(const kvz_hip_quant_params *p, const int16_t *coef, int16_t *q_coef, int32_t width, int32_t height,
                   int8_t type, int8_t scan_idx, int8_t block_type)
{ set_state(p); kvz_quant(&g_state, (coeff_t *)coef, q_coef, width, height, type, scan_idx, block_type); }
void kvz_ref_dequant(const kvz_hip_quant_params *p, const int16_t *q_coef, int16_t *coef, int32_t width, int32_t height,
                     int8_t type, int8_t block_type)
{ set_state(p); kvz_dequant(&g_state, (coeff_t *)q_coef, coef, width, height, type, block_type); }
int kvz_ref_quantize_residual(const kvz_hip_quant_params *p, int width, int color, int scan_order, int use_trskip,
                              int in_stride, int out_stride, const uint8_t *ref_in, const uint8_t *pred_in,
                              uint8_t *rec_out, int16_t *coeff_out, int early_skip)
{
  cu_info_t cu;
  memset(&cu, 0, sizeof cu);
  cu.type = p->cu_is_intra ? CU_INTRA : CU_INTER;
  cu.part_size = SIZE_2Nx2N;
  set_state(p);
  return kvz_quantize_residual(&g_state, &cu, width, (color_t)color, (coeff_scan_order_t)scan_order, use_trskip,
                               in_stride, out_stride, ref_in, pred_in, rec_out, coeff_out, early_skip);
}
/* kvz_rdoq (rdo.c:661) on an intra block with the given context states (KVZ_HIP_CX_* order) in state->cabac */
void kvz_ref_rdoq(int qp, double lambda, const uint8_t *ctx, const float *entropy_fbits_unused, const int16_t *coef, int16_t *dest, int width, int type, int scan_mode,
                  int tr_depth)
{
  (void)entropy_fbits_unused;
  cabac_data_t *cb = &g_state.cabac;
  g_state.qp = (int8_t)qp;
  g_state.lambda = lambda;
  g_ctrl.cfg.signhide_enable = 0;
  g_frame.slicetype = KVZ_SLICE_I;
#define SETCTX(dst, from, n) for (int i_ = 0; i_ < (n); i_++) (dst)[i_].uc_state = ctx[(from) + i_]
  SETCTX(cb->ctx.qt_cbf_model_luma, KVZ_HIP_CX_CBF_LUMA, 2);
  SETCTX(cb->ctx.qt_cbf_model_chroma, KVZ_HIP_CX_CBF_CHROMA, 2);
  SETCTX(&cb->ctx.qt_cbf_model_chroma[2], KVZ_HIP_CX_CBF_CHROMA_DEEP, 2);  /* read by the blocks of an NxN CU (tr_depth 2) */
  SETCTX(cb->ctx.cu_sig_coeff_group_model, KVZ_HIP_CX_SIG_CG, 4);
  SETCTX(cb->ctx.cu_sig_model_luma, KVZ_HIP_CX_SIG_LUMA, 27);
  SETCTX(cb->ctx.cu_sig_model_chroma, KVZ_HIP_CX_SIG_CHROMA, 15);
  SETCTX(cb->ctx.cu_ctx_last_y_luma, KVZ_HIP_CX_LAST_Y_LUMA, 15);
  SETCTX(cb->ctx.cu_ctx_last_y_chroma, KVZ_HIP_CX_LAST_Y_CHROMA, 15);
  SETCTX(cb->ctx.cu_ctx_last_x_luma, KVZ_HIP_CX_LAST_X_LUMA, 15);
  SETCTX(cb->ctx.cu_ctx_last_x_chroma, KVZ_HIP_CX_LAST_X_CHROMA, 15);
  SETCTX(cb->ctx.cu_one_model_luma, KVZ_HIP_CX_ONE_LUMA, 16);
  SETCTX(cb->ctx.cu_one_model_chroma, KVZ_HIP_CX_ONE_CHROMA, 8);
  SETCTX(cb->ctx.cu_abs_model_luma, KVZ_HIP_CX_ABS_LUMA, 4);
  SETCTX(cb->ctx.cu_abs_model_chroma, KVZ_HIP_CX_ABS_CHROMA, 2);
#undef SETCTX
  kvz_rdoq(&g_state, (coeff_t *)coef, dest, width, width, (int8_t)type, (int8_t)scan_mode, CU_INTRA, (int8_t)tr_depth);
}
uint32_t kvz_ref_coeff_abs_sum(const int16_t *coeffs, size_t length) { return kvz_coeff_abs_sum(coeffs, length); }
#include "strategies/strategies-nal.h"
void kvz_ref_plane_md5(const uint8_t *data, int height, int width, int stride, uint8_t *out16)
{
  unsigned char out[SEI_HASH_MAX_LENGTH];
  kvz_array_md5(data, height, width, stride, out, 8);
  memcpy(out16, out, 16);
}

uint32_t kvz_ref_plane_checksum(const uint8_t *data, int height, int width, int stride)
{
  unsigned char out[SEI_HASH_MAX_LENGTH] = { 0 };
  kvz_array_checksum(data, height, width, stride, out, KVZ_BIT_DEPTH);
  return ((uint32_t)out[0] << 24) | ((uint32_t)out[1] << 16) | ((uint32_t)out[2] << 8) | out[3];
}
double kvz_ref_fast_coeff_cost(const int16_t *coeff, int32_t width, uint64_t weights) { return kvz_fast_coeff_cost(coeff, width, weights); }
void kvz_ref_find_last_scanpos(const int16_t *coef, int16_t *dest_coeff, int8_t type, int32_t q_bits, const int16_t *quant_coeff,
                               int32_t *sig_coeff_inc_out, uint32_t cg_size, uint16_t *ctx_set, const uint32_t *scan,
                               int32_t *cg_last_scanpos, int32_t *last_scanpos, uint32_t cg_num, int32_t *cg_scanpos,
                               int32_t width, int8_t scan_mode)
{
  static struct kvz_sh_rates_t rates;
  memset(&rates, 0x55, sizeof rates);
  kvz_find_last_scanpos((coeff_t *)coef, dest_coeff, type, q_bits, quant_coeff, &rates, cg_size, ctx_set, scan, cg_last_scanpos,
                        last_scanpos, cg_num, cg_scanpos, width, scan_mode);
  for (int i = 0; i < width * width; i++) if (rates.sig_coeff_inc[i] != 0x55555555) sig_coeff_inc_out[i] = rates.sig_coeff_inc[i];
}
int32_t kvz_ref_get_scaled_qp(int8_t type, int8_t qp, int8_t qp_offset) { return kvz_get_scaled_qp(type, qp, qp_offset); }

/* ---- intra ---- */
void kvz_ref_angular_pred(int log2_width, int mode, const uint8_t *above, const uint8_t *left, uint8_t *dst) { kvz_angular_pred(log2_width, mode, above, left, dst); }
void kvz_ref_intra_pred_planar(int log2_width, const uint8_t *top, const uint8_t *left, uint8_t *dst) { kvz_intra_pred_planar(log2_width, top, left, dst); }
void kvz_ref_intra_pred_filtered_dc(int log2_width, const uint8_t *top, const uint8_t *left, uint8_t *dst) { kvz_intra_pred_filtered_dc(log2_width, top, left, dst); }

/* ---- ipol ---- */
void kvz_ref_sample_quarterpel_luma(const uint8_t *src, int16_t ss, int w, int h, uint8_t *dst, int16_t ds, int8_t hf, int8_t vf, const int16_t mv[2])
{ kvz_sample_quarterpel_luma(&g_ctrl, (kvz_pixel *)src, ss, w, h, dst, ds, hf, vf, mv); }
void kvz_ref_sample_quarterpel_luma_hi(const uint8_t *src, int16_t ss, int w, int h, int16_t *dst, int16_t ds, int8_t hf, int8_t vf, const int16_t mv[2])
{ kvz_sample_quarterpel_luma_hi(&g_ctrl, (kvz_pixel *)src, ss, w, h, dst, ds, hf, vf, mv); }
void kvz_ref_sample_octpel_chroma(const uint8_t *src, int16_t ss, int w, int h, uint8_t *dst, int16_t ds, int8_t hf, int8_t vf, const int16_t mv[2])
{ kvz_sample_octpel_chroma(&g_ctrl, (kvz_pixel *)src, ss, w, h, dst, ds, hf, vf, mv); }
void kvz_ref_sample_octpel_chroma_hi(const uint8_t *src, int16_t ss, int w, int h, int16_t *dst, int16_t ds, int8_t hf, int8_t vf, const int16_t mv[2])
{ kvz_sample_octpel_chroma_hi(&g_ctrl, (kvz_pixel *)src, ss, w, h, dst, ds, hf, vf, mv); }

#define IPOL_WRAP(name) \
  void kvz_ref_##name(const uint8_t *src, int16_t ss, int w, int h, uint8_t *filtered, int16_t *hor_intermediate, int8_t fme_level, \
                      int16_t *hor_first_cols, int8_t ox, int8_t oy) \
  { kvz_##name(&g_ctrl, (kvz_pixel *)src, ss, w, h, (kvz_pixel(*)[LCU_LUMA_SIZE])filtered, \
               (int16_t(*)[KVZ_IPOL_MAX_IM_SIZE_LUMA_SIMD])hor_intermediate, fme_level, \
               (int16_t(*)[KVZ_EXT_BLOCK_W_LUMA + 1])hor_first_cols, ox, oy); }
IPOL_WRAP(filter_hpel_blocks_hor_ver_luma)
IPOL_WRAP(filter_hpel_blocks_diag_luma)
IPOL_WRAP(filter_qpel_blocks_hor_ver_luma)
IPOL_WRAP(filter_qpel_blocks_diag_luma)

int kvz_ref_get_extended_block(const kvz_hip_epol_params *a, const uint8_t *src, uint8_t *buf)
{
  kvz_pixel *ext = NULL, *ext_origin = NULL; int ext_s = 0;
  kvz_epol_args args = { .src = (kvz_pixel *)src, .src_w = a->src_w, .src_h = a->src_h, .src_s = a->src_s,
                         .blk_x = a->blk_x, .blk_y = a->blk_y, .blk_w = a->blk_w, .blk_h = a->blk_h,
                         .pad_l = a->pad_l, .pad_r = a->pad_r, .pad_t = a->pad_t, .pad_b = a->pad_b, .pad_b_simd = a->pad_b_simd,
                         .buf = buf, .ext = &ext, .ext_origin = &ext_origin, .ext_s = &ext_s };
  kvz_get_extended_block(&args);
  return ext == buf;
}

/* ---- sao ---- */
int kvz_ref_sao_edge_ddistortion(int bitdepth, const uint8_t *orig, const uint8_t *rec, int bw, int bh, int eo_class, const int offsets[5])
{ (void)bitdepth; return kvz_sao_edge_ddistortion(&g_ctrl, orig, rec, bw, bh, eo_class, (int *)offsets); }
void kvz_ref_calc_sao_edge_dir(int bitdepth, const uint8_t *orig, const uint8_t *rec, int eo_class, int bw, int bh, int cat_sum_cnt[10])
{ (void)bitdepth; kvz_calc_sao_edge_dir(&g_ctrl, orig, rec, eo_class, bw, bh, (int(*)[NUM_SAO_EDGE_CATEGORIES])cat_sum_cnt); }
void kvz_ref_sao_reconstruct_color(const kvz_hip_sao_params *s, const uint8_t *rec, uint8_t *new_rec, int stride, int new_stride, int bw, int bh, int color)
{
  sao_info_t sao;
  memset(&sao, 0, sizeof sao);
  sao.type = (sao_type)s->type; sao.eo_class = (sao_eo_class)s->eo_class;
  sao.band_position[0] = s->band_position[0]; sao.band_position[1] = s->band_position[1];
  for (int i = 0; i < 10; i++) sao.offsets[i] = s->offsets[i];
  kvz_sao_reconstruct_color(&g_ctrl, rec, new_rec, &sao, stride, new_stride, bw, bh, (color_t)color);
}
int kvz_ref_sao_band_ddistortion(int bitdepth, const uint8_t *orig, const uint8_t *rec, int bw, int bh, int band_pos, const int sao_bands[4])
{ (void)bitdepth; return kvz_sao_band_ddistortion(&g_state, orig, rec, bw, bh, band_pos, sao_bands); }

/* ---- deblocking: the reference's kvz_filter_deblock_lcu (filter.c:783) over every LCU of an all-intra, constant-QP picture.
 * Planes are tight (stride = width) and filtered in place; cu_depth holds the CU depth per 8x8 unit. ---- */
#include "filter.h"
#include "videoframe.h"
void kvz_ref_deblock_frame(int width, int height, int qp, int beta_offset_div2, int tc_offset_div2, uint8_t *y, uint8_t *u, uint8_t *v,
                           const uint8_t *cu_depth)
{
  encoder_state_config_tile_t tile;
  videoframe_t frame;
  memset(&tile, 0, sizeof tile);
  memset(&frame, 0, sizeof frame);
  frame.width = width; frame.height = height;
  frame.width_in_lcu = (width + LCU_WIDTH - 1) / LCU_WIDTH; frame.height_in_lcu = (height + LCU_WIDTH - 1) / LCU_WIDTH;
  frame.rec = kvz_image_alloc(KVZ_CSP_420, width, height);
  frame.cu_array = kvz_cu_array_alloc(width, height);
  for (int r = 0; r < height; r++) memcpy(frame.rec->y + r * frame.rec->stride, y + r * width, width);
  for (int r = 0; r < height / 2; r++) {
    memcpy(frame.rec->u + r * (frame.rec->stride / 2), u + r * (width / 2), width / 2);
    memcpy(frame.rec->v + r * (frame.rec->stride / 2), v + r * (width / 2), width / 2);
  }
  for (int py = 0; py < height; py += 4)
    for (int px = 0; px < width; px += 4) {
      cu_info_t *cu = kvz_cu_array_at(frame.cu_array, px, py);
      const int d = cu_depth[(py >> 3) * (width >> 3) + (px >> 3)];
      memset(cu, 0, sizeof *cu);
      cu->type = CU_INTRA; cu->depth = d; cu->tr_depth = d ? d : 1; cu->part_size = SIZE_2Nx2N; cu->qp = (int8_t)qp;
    }
  tile.frame = &frame;
  g_state.tile = &tile;
  g_state.qp = (int8_t)qp;
  g_frame.max_qp_delta_depth = -1;
  g_frame.slicetype = KVZ_SLICE_I;
  g_frame.QP = (int8_t)qp;
  g_ctrl.cfg.deblock_beta = beta_offset_div2; g_ctrl.cfg.deblock_tc = tc_offset_div2;
  g_ctrl.cfg.lossless = 0;
  g_ctrl.chroma_format = KVZ_CSP_420;
  for (int ly = 0; ly < height; ly += LCU_WIDTH)
    for (int lx = 0; lx < width; lx += LCU_WIDTH) kvz_filter_deblock_lcu(&g_state, lx, ly);
  for (int r = 0; r < height; r++) memcpy(y + r * width, frame.rec->y + r * frame.rec->stride, width);
  for (int r = 0; r < height / 2; r++) {
    memcpy(u + r * (width / 2), frame.rec->u + r * (frame.rec->stride / 2), width / 2);
    memcpy(v + r * (width / 2), frame.rec->v + r * (frame.rec->stride / 2), width / 2);
  }
  kvz_image_free(frame.rec);
  kvz_cu_array_free(&frame.cu_array);
  g_state.tile = NULL;
}

/* ... and for pictures with inter CUs: the cu_array is filled from one record per 4x4 unit (include/kvz_hip_dev.h kvz_hip_cu_dbk) */
#include "../include/kvz_hip_dev.h"
void kvz_ref_deblock_frame_inter(int width, int height, int qp, int beta_offset_div2, int tc_offset_div2, uint8_t *y, uint8_t *u, uint8_t *v,
                                 const kvz_hip_cu_dbk *info, int slice_is_b)
{
  encoder_state_config_tile_t tile;
  videoframe_t frame;
  memset(&tile, 0, sizeof tile);
  memset(&frame, 0, sizeof frame);
  frame.width = width; frame.height = height;
  frame.width_in_lcu = (width + LCU_WIDTH - 1) / LCU_WIDTH; frame.height_in_lcu = (height + LCU_WIDTH - 1) / LCU_WIDTH;
  frame.rec = kvz_image_alloc(KVZ_CSP_420, width, height);
  frame.cu_array = kvz_cu_array_alloc(width, height);
  for (int r = 0; r < height; r++) memcpy(frame.rec->y + r * frame.rec->stride, y + r * width, width);
  for (int r = 0; r < height / 2; r++) {
    memcpy(frame.rec->u + r * (frame.rec->stride / 2), u + r * (width / 2), width / 2);
    memcpy(frame.rec->v + r * (frame.rec->stride / 2), v + r * (width / 2), width / 2);
  }
  for (int py = 0; py < height; py += 4)
    for (int px = 0; px < width; px += 4) {
      cu_info_t *cu = kvz_cu_array_at(frame.cu_array, px, py);
      const kvz_hip_cu_dbk *d = &info[(size_t)(py >> 2) * (width >> 2) + (px >> 2)];
      memset(cu, 0, sizeof *cu);
      cu->type = d->type; cu->depth = d->depth; cu->tr_depth = d->tr_depth; cu->part_size = d->part_size; cu->qp = (int8_t)qp;
      if (d->cbf_y) cbf_set(&cu->cbf, d->tr_depth, COLOR_Y);
      if (d->type == CU_INTER) {
        cu->inter.mv_dir = d->mv_dir;
        for (int l = 0; l < 2; l++) {
          cu->inter.mv[l][0] = d->mv[l][0]; cu->inter.mv[l][1] = d->mv[l][1]; cu->inter.mv_ref[l] = d->mv_ref[l];
          if (d->mv_dir & (1 << l)) g_frame.ref_LX[l][d->mv_ref[l]] = (uint8_t)d->ref_id[l];  /* the caller keeps (list, index) -> picture consistent */
        }
      }
    }
  tile.frame = &frame;
  g_state.tile = &tile;
  g_state.qp = (int8_t)qp;
  g_frame.max_qp_delta_depth = -1;
  g_frame.slicetype = slice_is_b ? KVZ_SLICE_B : KVZ_SLICE_P;
  g_frame.QP = (int8_t)qp;
  g_ctrl.cfg.deblock_beta = beta_offset_div2; g_ctrl.cfg.deblock_tc = tc_offset_div2;
  g_ctrl.cfg.lossless = 0;
  g_ctrl.chroma_format = KVZ_CSP_420;
  for (int ly = 0; ly < height; ly += LCU_WIDTH)
    for (int lx = 0; lx < width; lx += LCU_WIDTH) kvz_filter_deblock_lcu(&g_state, lx, ly);
  for (int r = 0; r < height; r++) memcpy(y + r * width, frame.rec->y + r * frame.rec->stride, width);
  for (int r = 0; r < height / 2; r++) {
    memcpy(u + r * (width / 2), frame.rec->u + r * (frame.rec->stride / 2), width / 2);
    memcpy(v + r * (width / 2), frame.rec->v + r * (frame.rec->stride / 2), width / 2);
  }
  kvz_image_free(frame.rec);
  kvz_cu_array_free(&frame.cu_array);
  g_state.tile = NULL;
}

/* ---- SAO applied to a whole picture with the reference's own border logic: kvz_sao_reconstruct (sao.c:302-361) per CTU and
 * plane, input = a separate copy of the deblocked picture (neighbours are pre-SAO samples), output = frame->rec ---- */
void kvz_ref_sao_frame(int width, int height, const uint8_t *in, uint8_t *out, const kvz_hip_sao_params *luma, const kvz_hip_sao_params *chroma)
{
  encoder_state_config_tile_t tile;
  videoframe_t frame;
  memset(&tile, 0, sizeof tile);
  memset(&frame, 0, sizeof frame);
  frame.width = width; frame.height = height;
  frame.rec = kvz_image_alloc(KVZ_CSP_420, width, height);
  kvz_picture *src = kvz_image_alloc(KVZ_CSP_420, width, height);
  for (int color = 0; color < 3; color++) {
    const int sh = color ? 1 : 0, fw = width >> sh, fh = height >> sh;
    const size_t plane = color == 0 ? 0 : (color == 1 ? (size_t)width * height : (size_t)width * height * 5 / 4);
    for (int r = 0; r < fh; r++) {
      memcpy(src->data[color] + r * (src->stride >> sh), in + plane + (size_t)r * fw, fw);
      memcpy(frame.rec->data[color] + r * (frame.rec->stride >> sh), in + plane + (size_t)r * fw, fw);
    }
  }
  tile.frame = &frame;
  g_state.tile = &tile;
  const int wc = (width + 63) / 64, hc = (height + 63) / 64;
  for (int color = 0; color < 3; color++) {
    const int sh = color ? 1 : 0, fw = width >> sh, fh = height >> sh, lcu = 64 >> sh;
    for (int cy = 0; cy < hc; cy++)
      for (int cx = 0; cx < wc; cx++) {
        const kvz_hip_sao_params *s = color ? &chroma[cy * wc + cx] : &luma[cy * wc + cx];
        sao_info_t sao;
        memset(&sao, 0, sizeof sao);
        sao.type = (sao_type)s->type; sao.eo_class = (sao_eo_class)s->eo_class;
        sao.band_position[0] = s->band_position[0]; sao.band_position[1] = s->band_position[1];
        for (int i = 0; i < 10; i++) sao.offsets[i] = s->offsets[i];
        const int x = cx * lcu, y = cy * lcu, w = fw - x < lcu ? fw - x : lcu, h = fh - y < lcu ? fh - y : lcu;
        kvz_sao_reconstruct(&g_state, src->data[color] + y * (src->stride >> sh) + x, src->stride >> sh, x, y, w, h, &sao, (color_t)color);
      }
  }
  for (int color = 0; color < 3; color++) {
    const int sh = color ? 1 : 0, fw = width >> sh, fh = height >> sh;
    const size_t plane = color == 0 ? 0 : (color == 1 ? (size_t)width * height : (size_t)width * height * 5 / 4);
    for (int r = 0; r < fh; r++) memcpy(out + plane + (size_t)r * fw, frame.rec->data[color] + r * (frame.rec->stride >> sh), fw);
  }
  kvz_image_free(frame.rec);
  kvz_image_free(src);
  g_state.tile = NULL;
}

/* ---- kvz_encode_coeff_nxn in counting mode (what get_coeff_cabac_cost runs, rdo.c:220-263) on caller-supplied context states:
 * ctx[] = the residual-coding contexts in the KVZ_HIP_CX_* order from KVZ_HIP_CX_SIG_CG on (uc_state each), updated in place when
 * `update` is set.  Returns the bits. ---- */
#include "cabac.h"
#include "context.h"
double kvz_ref_coeff_cabac_bits(const int16_t *coeff, int width, int type, int scan_mode, int update, uint8_t *ctx)
{
  cabac_data_t cabac;
  memset(&cabac, 0, sizeof cabac);
  cabac.only_count = 1;
  cabac.update = update ? 1 : 0;
  cabac.range = 510; cabac.bits_left = 23;
  uint8_t *p = ctx;
#define KVZ_REF_LOAD(arr, n) for (int i = 0; i < (n); i++) cabac.ctx.arr[i].uc_state = *p++;
  KVZ_REF_LOAD(cu_sig_coeff_group_model, 4) KVZ_REF_LOAD(cu_sig_model_luma, 27) KVZ_REF_LOAD(cu_sig_model_chroma, 15)
  KVZ_REF_LOAD(cu_ctx_last_y_luma, 15) KVZ_REF_LOAD(cu_ctx_last_y_chroma, 15) KVZ_REF_LOAD(cu_ctx_last_x_luma, 15) KVZ_REF_LOAD(cu_ctx_last_x_chroma, 15)
  KVZ_REF_LOAD(cu_one_model_luma, 16) KVZ_REF_LOAD(cu_one_model_chroma, 8) KVZ_REF_LOAD(cu_abs_model_luma, 4) KVZ_REF_LOAD(cu_abs_model_chroma, 2)
  g_ctrl.cfg.signhide_enable = 0; g_ctrl.cfg.trskip_enable = 0; g_ctrl.cfg.lossless = 0; g_ctrl.cfg.crypto_features = 0;
  double bits = 0;
  kvz_encode_coeff_nxn(&g_state, &cabac, coeff, (uint8_t)width, (uint8_t)type, (int8_t)scan_mode, 0, &bits);
  p = ctx;
#define KVZ_REF_STORE(arr, n) for (int i = 0; i < (n); i++) *p++ = cabac.ctx.arr[i].uc_state;
  KVZ_REF_STORE(cu_sig_coeff_group_model, 4) KVZ_REF_STORE(cu_sig_model_luma, 27) KVZ_REF_STORE(cu_sig_model_chroma, 15)
  KVZ_REF_STORE(cu_ctx_last_y_luma, 15) KVZ_REF_STORE(cu_ctx_last_y_chroma, 15) KVZ_REF_STORE(cu_ctx_last_x_luma, 15) KVZ_REF_STORE(cu_ctx_last_x_chroma, 15)
  KVZ_REF_STORE(cu_one_model_luma, 16) KVZ_REF_STORE(cu_one_model_chroma, 8) KVZ_REF_STORE(cu_abs_model_luma, 4) KVZ_REF_STORE(cu_abs_model_chroma, 2)
  return bits;
}

/* kvz_encode_coeff_nxn (strategies-encode.h:49-65) in its REAL mode on a block, from the given context states (KVZ_HIP_CX_* order), then kvz_cabac_finish, a stop bit and
 * the alignment (what ends a substream, encoderstate.c:726-732): the bytes the reference leaves in the stream.  Returns their number. */
#include "bitstream.h"
#include "strategies/strategies-encode.h"
int kvz_ref_encode_coeff_nxn_bytes(const uint8_t *ctx, const int16_t *coeff, int width, int type, int scan_mode, uint8_t *out, int capacity)
{
  cabac_data_t *cb = &g_state.cabac;
  bitstream_t stream;
  kvz_bitstream_init(&stream);
  g_ctrl.cfg.signhide_enable = 0; g_ctrl.cfg.trskip_enable = 0; g_ctrl.cfg.crypto_features = 0; g_ctrl.cfg.lossless = 0;
  g_frame.slicetype = KVZ_SLICE_I;
#define SETCTX(dst, from, n) for (int i_ = 0; i_ < (n); i_++) (dst)[i_].uc_state = ctx[(from) + i_]
  SETCTX(cb->ctx.cu_sig_coeff_group_model, KVZ_HIP_CX_SIG_CG, 4);
  SETCTX(cb->ctx.cu_sig_model_luma, KVZ_HIP_CX_SIG_LUMA, 27);
  SETCTX(cb->ctx.cu_sig_model_chroma, KVZ_HIP_CX_SIG_CHROMA, 15);
  SETCTX(cb->ctx.cu_ctx_last_y_luma, KVZ_HIP_CX_LAST_Y_LUMA, 15);
  SETCTX(cb->ctx.cu_ctx_last_y_chroma, KVZ_HIP_CX_LAST_Y_CHROMA, 15);
  SETCTX(cb->ctx.cu_ctx_last_x_luma, KVZ_HIP_CX_LAST_X_LUMA, 15);
  SETCTX(cb->ctx.cu_ctx_last_x_chroma, KVZ_HIP_CX_LAST_X_CHROMA, 15);
  SETCTX(cb->ctx.cu_one_model_luma, KVZ_HIP_CX_ONE_LUMA, 16);
  SETCTX(cb->ctx.cu_one_model_chroma, KVZ_HIP_CX_ONE_CHROMA, 8);
  SETCTX(cb->ctx.cu_abs_model_luma, KVZ_HIP_CX_ABS_LUMA, 4);
  SETCTX(cb->ctx.cu_abs_model_chroma, KVZ_HIP_CX_ABS_CHROMA, 2);
#undef SETCTX
  kvz_cabac_start(cb);
  cb->stream = &stream;
  cb->only_count = 0;
  cb->update = 1;  /* as encoder_state_worker_encode_lcu_bitstream sets it (encoderstate.c:686): CABAC_FBITS_UPDATE only codes a bin when it is on */
  kvz_encode_coeff_nxn(&g_state, cb, (const coeff_t *)coeff, (uint8_t)width, (uint8_t)type, (int8_t)scan_mode, 0, NULL);
  kvz_cabac_finish(cb);
  kvz_bitstream_put(&stream, 1, 1);
  kvz_bitstream_align_zero(&stream);
  int n = 0;
  for (kvz_data_chunk *c = stream.first; c; c = c->next)
    for (uint32_t i = 0; i < c->len; i++) { if (n < capacity) out[n] = c->data[i]; n++; }
  kvz_bitstream_finalize(&stream);
  return n;
}
