/* strategies/hip/dct-hip.c -- strategies-dct.h:69-82; every dct_func is struct-free, registered as is */
#include "strategies/hip/hip-common.h"
#include "strategies/strategies-dct.h"
#include "strategyselector.h"

int kvz_strategy_register_dct_hip(void *opaque, uint8_t bitdepth)
{
  bool success = true;
  if (!kvz_hip_strategy_usable(bitdepth)) return 1;
#define REG(type, fn) success &= kvz_strategyselector_register(opaque, type, "hip", KVZ_HIP_PRIORITY, (void *)(fn))
  REG("fast_forward_dst_4x4", &kvz_hip_fast_forward_dst_4x4);
  REG("dct_4x4", &kvz_hip_dct_4x4); REG("dct_8x8", &kvz_hip_dct_8x8); REG("dct_16x16", &kvz_hip_dct_16x16); REG("dct_32x32", &kvz_hip_dct_32x32);
  REG("fast_inverse_dst_4x4", &kvz_hip_fast_inverse_dst_4x4);
  REG("idct_4x4", &kvz_hip_idct_4x4); REG("idct_8x8", &kvz_hip_idct_8x8); REG("idct_16x16", &kvz_hip_idct_16x16); REG("idct_32x32", &kvz_hip_idct_32x32);
#undef REG
  return success;
}
