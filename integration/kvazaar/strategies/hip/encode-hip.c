/* strategies/hip/encode-hip.c -- strategies-encode.h:49-65: kvz_encode_coeff_nxn in its real (bit-producing) mode.
 *
 * The block's residual syntax -- last position, coded_sub_block flags, significance / greater-1 / greater-2 flags with their context selection, signs, Golomb-Rice
 * remainders -- is walked on the device and comes back as bin records (kvz_hip_coeff_nxn_bins, kvazaar_amd/csrc/kvz_entropy.hpp); the arithmetic coder state lives in
 * the cabac_data_t the caller handed over, so the records are fed to kvazaar's own kvz_cabac_encode_bin / kvz_cabac_encode_bins_ep here.  The counting mode
 * (cabac->only_count: rdo.c:220-263 prices coefficients with it thousands of times per CTU) and the options the device walk does not model go to the generic
 * function.  The whole-picture form of the same coder, where nothing but finished substreams leaves the device, is kvz_hip_batch_entropy_code (kvz_hip_batch.h). */
#include "strategies/hip/hip-common.h"

#include "cabac.h"
#include "encoderstate.h"
#include "strategies/generic/encode_coding_tree-generic.h"
#include "strategies/strategies-encode.h"
#include "strategyselector.h"

#include "kvz_hip_types.h"

static cabac_ctx_t *context_of(cabac_data_t *c, int idx)  /* kvz_hip_types.h KVZ_HIP_CX_* -> cabac.h:63-100 */
{
  if (idx >= KVZ_HIP_CX_ABS_CHROMA) return &c->ctx.cu_abs_model_chroma[idx - KVZ_HIP_CX_ABS_CHROMA];
  if (idx >= KVZ_HIP_CX_ABS_LUMA) return &c->ctx.cu_abs_model_luma[idx - KVZ_HIP_CX_ABS_LUMA];
  if (idx >= KVZ_HIP_CX_ONE_CHROMA) return &c->ctx.cu_one_model_chroma[idx - KVZ_HIP_CX_ONE_CHROMA];
  if (idx >= KVZ_HIP_CX_ONE_LUMA) return &c->ctx.cu_one_model_luma[idx - KVZ_HIP_CX_ONE_LUMA];
  if (idx >= KVZ_HIP_CX_LAST_X_CHROMA) return &c->ctx.cu_ctx_last_x_chroma[idx - KVZ_HIP_CX_LAST_X_CHROMA];
  if (idx >= KVZ_HIP_CX_LAST_X_LUMA) return &c->ctx.cu_ctx_last_x_luma[idx - KVZ_HIP_CX_LAST_X_LUMA];
  if (idx >= KVZ_HIP_CX_LAST_Y_CHROMA) return &c->ctx.cu_ctx_last_y_chroma[idx - KVZ_HIP_CX_LAST_Y_CHROMA];
  if (idx >= KVZ_HIP_CX_LAST_Y_LUMA) return &c->ctx.cu_ctx_last_y_luma[idx - KVZ_HIP_CX_LAST_Y_LUMA];
  if (idx >= KVZ_HIP_CX_SIG_CHROMA) return &c->ctx.cu_sig_model_chroma[idx - KVZ_HIP_CX_SIG_CHROMA];
  if (idx >= KVZ_HIP_CX_SIG_LUMA) return &c->ctx.cu_sig_model_luma[idx - KVZ_HIP_CX_SIG_LUMA];
  return &c->ctx.cu_sig_coeff_group_model[idx - KVZ_HIP_CX_SIG_CG];
}

static void encode_coeff_nxn_hip(encoder_state_t *const state, cabac_data_t *const cabac, const coeff_t *coeff, uint8_t width, uint8_t type, int8_t scan_mode,
                                 int8_t tr_skip, double *bits_out)
{
  const kvz_config *cfg = &state->encoder_control->cfg;
  if (cabac->only_count || cfg->signhide_enable || cfg->trskip_enable || cfg->crypto_features || cfg->lossless) {
    kvz_encode_coeff_nxn_generic(state, cabac, coeff, width, type, scan_mode, tr_skip, bits_out);
    return;
  }
  enum { CAPACITY = 8192 };  /* a 32x32 block: 64 groups x (16 + 1 + 8 + 1 + 1 + 48) records at the very worst */
  static __thread uint32_t records[CAPACITY];
  /* the buffer the call moves both ways is sized by the block: per 4x4 group at most 59 records (16 + 1 + 8 + 1 + 1 + 32), the last position 22: 6 per coefficient covers it */
  const int cap = width * width * 6 < CAPACITY ? width * width * 6 : CAPACITY;
  const int n = kvz_hip_coeff_nxn_bins(coeff, width, type, scan_mode, records, cap);
  if (n > cap) { kvz_encode_coeff_nxn_generic(state, cabac, coeff, width, type, scan_mode, tr_skip, bits_out); return; }
  for (int i = 0; i < n; i++) {
    const uint32_t r = records[i];
    if ((r >> 30) == 0) {
      if (cabac->update) {  /* CABAC_FBITS_UPDATE (cabac.h:133-139): a context-coded bin is only coded while the coder's contexts adapt */
        cabac->cur_ctx = context_of(cabac, (int)(r & 0xff));
        kvz_cabac_encode_bin(cabac, (r >> 8) & 1);
      }
    } else {
      kvz_cabac_encode_bins_ep(cabac, r & 0xffff, (int)((r >> 16) & 0x3f));
    }
  }
}

int kvz_strategy_register_encode_hip(void *opaque, uint8_t bitdepth)
{
  if (!kvz_hip_strategy_usable(bitdepth)) return 1;
  return kvz_strategyselector_register(opaque, "encode_coeff_nxn", "hip", KVZ_HIP_PRIORITY, (void *)&encode_coeff_nxn_hip);
}
