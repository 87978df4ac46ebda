/* strategies/hip/ipol-hip.c -- strategies-ipol.h:141-150.  The encoder_control_t argument is unused at 8 bit. */
#include "strategies/hip/hip-common.h"

#include "encoder.h"
#include "strategies/strategies-ipol.h"
#include "strategyselector.h"

#define SAMPLE(name, dst_t)                                                                                                          \
  static void name##_hip(const encoder_control_t *const encoder, kvz_pixel *src, int16_t src_stride, int width, int height,          \
                         dst_t *dst, int16_t dst_stride, int8_t hor_flag, int8_t ver_flag, const int16_t mv[2])                      \
  { (void)encoder; kvz_hip_##name(src, src_stride, width, height, dst, dst_stride, hor_flag, ver_flag, mv); }
SAMPLE(sample_quarterpel_luma, kvz_pixel)
SAMPLE(sample_octpel_chroma, kvz_pixel)
SAMPLE(sample_quarterpel_luma_hi, int16_t)
SAMPLE(sample_octpel_chroma_hi, int16_t)

#define BLOCKS(name)                                                                                                                 \
  static void name##_hip(const encoder_control_t *encoder, kvz_pixel *src, int16_t src_stride, int width, int height,                \
                         kvz_pixel filtered[4][LCU_LUMA_SIZE], int16_t hor_intermediate[5][KVZ_IPOL_MAX_IM_SIZE_LUMA_SIMD],           \
                         int8_t fme_level, int16_t hor_first_cols[5][KVZ_EXT_BLOCK_W_LUMA + 1], int8_t off_x, int8_t off_y)          \
  { (void)encoder; kvz_hip_##name(src, src_stride, width, height, &filtered[0][0], &hor_intermediate[0][0], fme_level,               \
                                  &hor_first_cols[0][0], off_x, off_y); }
BLOCKS(filter_hpel_blocks_hor_ver_luma)
BLOCKS(filter_hpel_blocks_diag_luma)
BLOCKS(filter_qpel_blocks_hor_ver_luma)
BLOCKS(filter_qpel_blocks_diag_luma)

/* epol_func: when the window is inside the frame the reference returns pointers into the frame (ipol-generic.c:807-812) --
 * pure pointer arithmetic, done here; otherwise the edge-replicated copy is produced on the device. */
static void get_extended_block_hip(kvz_epol_args *args)
{
  kvz_hip_epol_params p = { args->src_w, args->src_h, args->src_s, args->blk_x, args->blk_y, args->blk_w, args->blk_h,
                            args->pad_l, args->pad_r, args->pad_t, args->pad_b, args->pad_b_simd };
  if (kvz_hip_get_extended_block(&p, args->src, args->buf)) {
    *args->ext = args->buf;
    *args->ext_s = args->pad_l + args->blk_w + args->pad_r;
    *args->ext_origin = args->buf + args->pad_t * (*args->ext_s) + args->pad_l;
  } else {
    *args->ext = args->src + (args->blk_y - args->pad_t) * args->src_s + (args->blk_x - args->pad_l);
    *args->ext_origin = args->src + args->blk_y * args->src_s + args->blk_x;
    *args->ext_s = args->src_s;
  }
}

int kvz_strategy_register_ipol_hip(void *opaque, uint8_t bitdepth)
{
  bool success = true;
  if (!kvz_hip_strategy_usable(bitdepth)) return 1;
#define REG(type) success &= kvz_strategyselector_register(opaque, #type, "hip", KVZ_HIP_PRIORITY, (void *)&type##_hip)
  REG(filter_hpel_blocks_hor_ver_luma); REG(filter_hpel_blocks_diag_luma);
  REG(filter_qpel_blocks_hor_ver_luma); REG(filter_qpel_blocks_diag_luma);
  REG(sample_quarterpel_luma); REG(sample_octpel_chroma); REG(sample_quarterpel_luma_hi); REG(sample_octpel_chroma_hi);
  REG(get_extended_block);
#undef REG
  return success;
}
