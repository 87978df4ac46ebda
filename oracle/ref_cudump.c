/*
 * ref_cudump.c -- TEST INFRASTRUCTURE.  LD_PRELOAD interposer for the reference CLI (oracle/_ref/kvazaar_ref): wraps
 * kvz_encode_coding_tree (encode_coding_tree.c:745), which the encoder calls once per LCU with depth 0 after the search, and appends
 * the CU quadtree the search left in the frame's cu_array -- "<x> <y> <depth> <intra mode> <cbf>" per 8x8 cell -- to the file named
 * by KVZ_CUDUMP.  tests/golden/make_golden.py uses it to record the reference encoder's CU depths and intra modes next to its
 * reconstruction.  Compiled against the reference's headers where they lie; never part of the product.
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>

#include "global.h"
#include "cu.h"
#include "encoderstate.h"
#include "videoframe.h"

void kvz_encode_coding_tree(encoder_state_t *const state, uint16_t x, uint16_t y, uint8_t depth)
{
  static void (*real)(encoder_state_t *const, uint16_t, uint16_t, uint8_t);
  if (!real) real = (void (*)(encoder_state_t *const, uint16_t, uint16_t, uint8_t))dlsym(RTLD_NEXT, "kvz_encode_coding_tree");
  const char *path = getenv("KVZ_CUDUMP");
  if (depth == 0 && path) {
    FILE *f = fopen(path, "a");
    const videoframe_t *frame = state->tile->frame;
    for (int yy = 0; yy < LCU_WIDTH; yy += 8)
      for (int xx = 0; xx < LCU_WIDTH; xx += 8) {
        if (x + xx >= frame->width || y + yy >= frame->height) continue;
        const cu_info_t *c = kvz_cu_array_at_const(frame->cu_array, x + xx, y + yy);
        fprintf(f, "%d %d %d %d %d\n", x + xx, y + yy, c->depth, c->intra.mode, c->cbf);
      }
    fclose(f);
  }
  real(state, x, y, depth);
}
