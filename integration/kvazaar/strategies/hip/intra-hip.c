/* strategies/hip/intra-hip.c -- strategies-intra.h:72-75 */
#include "strategies/hip/hip-common.h"
#include "strategies/strategies-intra.h"
#include "strategyselector.h"

int kvz_strategy_register_intra_hip(void *opaque, uint8_t bitdepth)
{
  bool success = true;
  if (!kvz_hip_strategy_usable(bitdepth)) return 1;
  success &= kvz_strategyselector_register(opaque, "angular_pred", "hip", KVZ_HIP_PRIORITY, (void *)&kvz_hip_angular_pred);
  success &= kvz_strategyselector_register(opaque, "intra_pred_planar", "hip", KVZ_HIP_PRIORITY, (void *)&kvz_hip_intra_pred_planar);
  success &= kvz_strategyselector_register(opaque, "intra_pred_filtered_dc", "hip", KVZ_HIP_PRIORITY, (void *)&kvz_hip_intra_pred_filtered_dc);
  return success;
}
