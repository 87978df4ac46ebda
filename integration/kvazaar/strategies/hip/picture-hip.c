/* strategies/hip/picture-hip.c -- registers the hip versions of strategies-picture.h:198-227 */
#include "strategies/hip/hip-common.h"

#include <stdlib.h>

#include "cu.h"
#include "strategies/strategies-picture.h"
#include "strategyselector.h"

int kvz_hip_strategy_usable(uint8_t bitdepth)
{
  static int usable = -1;
  if (bitdepth != 8) return 0;
  if (usable < 0) {
    const char *off = getenv("KVZ_HIP_DISABLE");
    usable = !(off && off[0] == '1') && kvz_hip_device_count() > 0;
  }
  return usable;
}

int kvz_hip_strategy_priority(void)
{
  static int priority = -1;
  if (priority < 0) {
    const char *on = getenv("KVZ_HIP_DROPIN");
    priority = (on && on[0] == '1') ? 50 : 0;
  }
  return priority;
}

/* inter_recon_bipred_func (strategies-picture.h:136-148): pick the planes out of lcu_t / yuv_t / yuv_im_t exactly as
 * bipred_average_generic does (picture-generic.c:616-668) and hand each plane to the device. */
static void bipred_average_hip(lcu_t *const lcu, const yuv_t *const px_L0, const yuv_t *const px_L1,
                               const yuv_im_t *const im_L0, const yuv_im_t *const im_L1, const unsigned pu_x,
                               const unsigned pu_y, const unsigned pu_w, const unsigned pu_h, const unsigned im_flags_L0,
                               const unsigned im_flags_L1, const bool predict_luma, const bool predict_chroma)
{
  if (predict_luma) {
    unsigned off = SUB_SCU(pu_y) * LCU_WIDTH + SUB_SCU(pu_x);
    kvz_hip_bipred_average_plane(lcu->rec.y + off, LCU_WIDTH, (im_flags_L0 & 1) ? NULL : px_L0->y, (im_flags_L0 & 1) ? im_L0->y : NULL,
                                 (im_flags_L1 & 1) ? NULL : px_L1->y, (im_flags_L1 & 1) ? im_L1->y : NULL, pu_w, pu_h);
  }
  if (predict_chroma) {
    unsigned off = SUB_SCU(pu_y) / 2 * LCU_WIDTH_C + SUB_SCU(pu_x) / 2;
    kvz_hip_bipred_average_plane(lcu->rec.u + off, LCU_WIDTH_C, (im_flags_L0 & 2) ? NULL : px_L0->u, (im_flags_L0 & 2) ? im_L0->u : NULL,
                                 (im_flags_L1 & 2) ? NULL : px_L1->u, (im_flags_L1 & 2) ? im_L1->u : NULL, pu_w / 2, pu_h / 2);
    kvz_hip_bipred_average_plane(lcu->rec.v + off, LCU_WIDTH_C, (im_flags_L0 & 2) ? NULL : px_L0->v, (im_flags_L0 & 2) ? im_L0->v : NULL,
                                 (im_flags_L1 & 2) ? NULL : px_L1->v, (im_flags_L1 & 2) ? im_L1->v : NULL, pu_w / 2, pu_h / 2);
  }
}

static optimized_sad_func_ptr_t get_optimized_sad_hip(int32_t width) { return (optimized_sad_func_ptr_t)kvz_hip_get_optimized_sad(width); }

int kvz_strategy_register_picture_hip(void *opaque, uint8_t bitdepth)
{
  bool success = true;
  if (!kvz_hip_strategy_usable(bitdepth)) return 1;
#define REG(type, fn) success &= kvz_strategyselector_register(opaque, type, "hip", KVZ_HIP_PRIORITY, (void *)(fn))
  REG("reg_sad", &kvz_hip_reg_sad);
  REG("sad_4x4", &kvz_hip_sad_4x4);   REG("sad_8x8", &kvz_hip_sad_8x8);   REG("sad_16x16", &kvz_hip_sad_16x16);
  REG("sad_32x32", &kvz_hip_sad_32x32); REG("sad_64x64", &kvz_hip_sad_64x64);
  REG("satd_4x4", &kvz_hip_satd_4x4); REG("satd_8x8", &kvz_hip_satd_8x8); REG("satd_16x16", &kvz_hip_satd_16x16);
  REG("satd_32x32", &kvz_hip_satd_32x32); REG("satd_64x64", &kvz_hip_satd_64x64);
  REG("sad_4x4_dual", &kvz_hip_sad_4x4_dual);   REG("sad_8x8_dual", &kvz_hip_sad_8x8_dual);   REG("sad_16x16_dual", &kvz_hip_sad_16x16_dual);
  REG("sad_32x32_dual", &kvz_hip_sad_32x32_dual); REG("sad_64x64_dual", &kvz_hip_sad_64x64_dual);
  REG("satd_4x4_dual", &kvz_hip_satd_4x4_dual); REG("satd_8x8_dual", &kvz_hip_satd_8x8_dual); REG("satd_16x16_dual", &kvz_hip_satd_16x16_dual);
  REG("satd_32x32_dual", &kvz_hip_satd_32x32_dual); REG("satd_64x64_dual", &kvz_hip_satd_64x64_dual);
  REG("satd_any_size", &kvz_hip_satd_any_size);
  REG("satd_any_size_quad", &kvz_hip_satd_any_size_quad);
  REG("pixels_calc_ssd", &kvz_hip_pixels_calc_ssd);
  REG("bipred_average", &bipred_average_hip);
  REG("get_optimized_sad", &get_optimized_sad_hip);
  REG("ver_sad", &kvz_hip_ver_sad);
  REG("hor_sad", &kvz_hip_hor_sad);
  REG("pixel_var", &kvz_hip_pixel_var);
#undef REG
  return success;
}
