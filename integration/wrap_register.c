/*
 * wrap_register.c -- build-time splice used ONLY by this repository's oracle/Makefile.
 *
 * A kvazaar maintainer adds one line per strategies-[x].c.  This repository must not copy or
 * modify reference sources, so the integrated test binary (oracle/_ref/kvazaar_hip) gets the same effect at link time:
 * `-Wl,--wrap=kvz_strategy_register_X_generic` routes each family's generic registration call through the
 * functions below, which register the hip strategy and then generic (the order INTEGRATION.md section 1 prescribes: with the default priority 0 the later
 * registration, generic, wins the tie and the hip entries are only reachable through KVAZAAR_OVERRIDE_<type>=hip; KVZ_HIP_DROPIN=1 lifts them to 50).
 */
#include <stdint.h>

#define WRAP(family)                                                                      \
  int __real_kvz_strategy_register_##family##_generic(void *opaque, uint8_t bitdepth);    \
  int kvz_strategy_register_##family##_hip(void *opaque, uint8_t bitdepth);               \
  int __wrap_kvz_strategy_register_##family##_generic(void *opaque, uint8_t bitdepth)     \
  {                                                                                       \
    int ok = kvz_strategy_register_##family##_hip(opaque, bitdepth);                      \
    return ok & __real_kvz_strategy_register_##family##_generic(opaque, bitdepth);        \
  }
WRAP(picture)
WRAP(dct)
WRAP(quant)
WRAP(intra)
WRAP(ipol)
WRAP(sao)
WRAP(nal)
WRAP(encode)
