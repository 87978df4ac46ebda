/*
 * strategies/hip/hip-common.h -- kvazaar-side registration shim for the MI355X `hip` strategy.
 *
 * This directory is what a kvazaar maintainer adds under src/strategies/hip/ (see INTEGRATION.md): plain C that
 * includes kvazaar's own headers, reads the few scalars a kernel needs out of kvazaar's host structs
 * (encoder_state_t, encoder_control_t, cu_info_t, lcu_t, sao_info_t, kvz_epol_args) and calls the struct-free C ABI
 * of libkvz_hip.so (include/kvz_hip.h).  It compiles only against a kvazaar source tree; in this repository
 * oracle/Makefile builds it against /root/reference into oracle/_ref/ (kvazaar_hip, libkvazaar_hip.so).
 *
 * Selection (strategyselector.c:285-306: the highest priority wins, a tie goes to the LATER registration).  A strategy call served per call by the device costs a
 * PCIe round trip (~11 us: two orders of magnitude more than the function it replaces), so the per-call strategies are OPT-IN: by default they register with
 * priority 0 and BEFORE the generic ones, i.e. they lose every selection -- generic's 0 registered later, AVX2's 40 -- and only
 * KVAZAAR_OVERRIDE_<type>=hip (strategyselector.c:291-294) picks one of them.  KVZ_HIP_DROPIN=1 registers them with priority 50 > AVX2's 40: the whole per-call
 * drop-in, what the parity tests run.  The throughput path is the batched pass (search_lcu_hip.c, KVZ_HIP_BATCH_SEARCH=1), which does not go through the pointers.
 * Like the AVX2 strategies, nothing is registered unless bitdepth == 8 (quant-avx2.c:939-945), and nothing is registered when no HIP device is usable.
 */
#ifndef STRATEGIES_HIP_COMMON_H_
#define STRATEGIES_HIP_COMMON_H_

#include "global.h" // IWYU pragma: keep
#include "kvz_hip.h"

/* 50 with KVZ_HIP_DROPIN=1, else 0 (see above) */
int kvz_hip_strategy_priority(void);
#define KVZ_HIP_PRIORITY kvz_hip_strategy_priority()

int kvz_strategy_register_picture_hip(void *opaque, uint8_t bitdepth);
int kvz_strategy_register_dct_hip(void *opaque, uint8_t bitdepth);
int kvz_strategy_register_quant_hip(void *opaque, uint8_t bitdepth);
int kvz_strategy_register_intra_hip(void *opaque, uint8_t bitdepth);
int kvz_strategy_register_ipol_hip(void *opaque, uint8_t bitdepth);
int kvz_strategy_register_sao_hip(void *opaque, uint8_t bitdepth);
int kvz_strategy_register_nal_hip(void *opaque, uint8_t bitdepth);
int kvz_strategy_register_encode_hip(void *opaque, uint8_t bitdepth);

/* 1 when the strategy should register (8-bit build, a usable device, not disabled by KVZ_HIP_DISABLE=1) */
int kvz_hip_strategy_usable(uint8_t bitdepth);

#endif
