/* strategies/hip/nal-hip.c -- strategies-nal.h:63-65: the picture-hash SEI's per-plane checksum and MD5 (--hash checksum / md5). */
#include "strategies/hip/hip-common.h"

#include "nal.h"
#include "strategies/strategies-nal.h"
#include "strategyselector.h"

static void array_checksum_hip(const kvz_pixel *data, const int height, const int width, const int stride,
                               unsigned char checksum_out[SEI_HASH_MAX_LENGTH], const uint8_t bitdepth)
{
  kvz_hip_array_checksum(data, height, width, stride, checksum_out, bitdepth);
}

static void array_md5_hip(const kvz_pixel *data, const int height, const int width, const int stride,
                          unsigned char checksum_out[SEI_HASH_MAX_LENGTH], const uint8_t bitdepth)
{
  kvz_hip_array_md5(data, height, width, stride, checksum_out, bitdepth);
}

int kvz_strategy_register_nal_hip(void *opaque, uint8_t bitdepth)
{
  if (!kvz_hip_strategy_usable(bitdepth)) return 1;
  int ok = kvz_strategyselector_register(opaque, "array_checksum", "hip", KVZ_HIP_PRIORITY, (void *)&array_checksum_hip);
  ok &= kvz_strategyselector_register(opaque, "array_md5", "hip", KVZ_HIP_PRIORITY, (void *)&array_md5_hip);
  return ok;
}
