/* strategies/hip/sao-hip.c -- strategies-sao.h:79-83 */
#include "strategies/hip/hip-common.h"

#include "encoder.h"
#include "encoderstate.h"
#include "sao.h"
#include "strategies/strategies-sao.h"
#include "strategyselector.h"

static int sao_edge_ddistortion_hip(const encoder_control_t *const encoder, const kvz_pixel *orig_data, const kvz_pixel *rec_data,
                                    int block_width, int block_height, int eo_class, int offsets[NUM_SAO_EDGE_CATEGORIES])
{ return kvz_hip_sao_edge_ddistortion(encoder->bitdepth, orig_data, rec_data, block_width, block_height, eo_class, offsets); }

static void calc_sao_edge_dir_hip(const encoder_control_t *const encoder, const kvz_pixel *orig_data, const kvz_pixel *rec_data, int eo_class,
                                  int block_width, int block_height, int cat_sum_cnt[2][NUM_SAO_EDGE_CATEGORIES])
{ kvz_hip_calc_sao_edge_dir(encoder->bitdepth, orig_data, rec_data, eo_class, block_width, block_height, &cat_sum_cnt[0][0]); }

static void sao_reconstruct_color_hip(const encoder_control_t *const encoder, const kvz_pixel *rec_data, kvz_pixel *new_rec_data,
                                      const sao_info_t *sao, int stride, int new_stride, int block_width, int block_height, color_t color_i)
{
  kvz_hip_sao_params p;
  p.type = sao->type; p.eo_class = sao->eo_class; p.bitdepth = encoder->bitdepth;
  p.band_position[0] = sao->band_position[0]; p.band_position[1] = sao->band_position[1];
  for (int i = 0; i < NUM_SAO_EDGE_CATEGORIES * 2; i++) p.offsets[i] = sao->offsets[i];
  kvz_hip_sao_reconstruct_color(&p, rec_data, new_rec_data, stride, new_stride, block_width, block_height, color_i);
}

static int sao_band_ddistortion_hip(const encoder_state_t *const state, const kvz_pixel *orig_data, const kvz_pixel *rec_data,
                                    int block_width, int block_height, int band_pos, const int sao_bands[4])
{ return kvz_hip_sao_band_ddistortion(state->encoder_control->bitdepth, orig_data, rec_data, block_width, block_height, band_pos, sao_bands); }

int kvz_strategy_register_sao_hip(void *opaque, uint8_t bitdepth)
{
  bool success = true;
  if (!kvz_hip_strategy_usable(bitdepth)) return 1;
  success &= kvz_strategyselector_register(opaque, "sao_edge_ddistortion", "hip", KVZ_HIP_PRIORITY, (void *)&sao_edge_ddistortion_hip);
  success &= kvz_strategyselector_register(opaque, "calc_sao_edge_dir", "hip", KVZ_HIP_PRIORITY, (void *)&calc_sao_edge_dir_hip);
  success &= kvz_strategyselector_register(opaque, "sao_reconstruct_color", "hip", KVZ_HIP_PRIORITY, (void *)&sao_reconstruct_color_hip);
  success &= kvz_strategyselector_register(opaque, "sao_band_ddistortion", "hip", KVZ_HIP_PRIORITY, (void *)&sao_band_ddistortion_hip);
  return success;
}
