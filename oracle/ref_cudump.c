/*
 * ref_cudump.c -- TEST INFRASTRUCTURE.  LD_PRELOAD interposer for the reference CLI (oracle/_ref/kvazaar_ref): wraps
 * kvz_encode_coding_tree (encode_coding_tree.c:745), which the encoder calls once per LCU with depth 0 after the search, and appends
 * the CU quadtree the search left in the frame's cu_array -- "<x> <y> <depth> <intra mode> <cbf>" per 8x8 cell -- to the file named
 * by KVZ_CUDUMP.  tests/golden/make_golden.py uses it to record the reference encoder's CU depths and intra modes next to its
 * reconstruction.  With KVZ_CUDUMP_INTER set the record is one line per 4x4 cell with everything the inter search decided (cu.h:130-170):
 * "<x> <y> <type> <depth> <part_size> <tr_depth> <skipped> <merged> <merge_idx> <cbf> <intra mode> <mv_dir> <mv L0 x y> <mv L1 x y> <ref L0 L1>
 * <mvp idx L0 L1>" -- motion fields of a list that mv_dir does not use are written as 0 / 255 (the encoder leaves them undefined); x, y are FRAME positions (the
 * tile's offset added: with --tiles the encoder's coordinates are tile-local).
 * Compiled against the reference's headers where they lie; never part of the product.
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
  if (depth == 0 && path && getenv("KVZ_CUDUMP_INTER")) {
    FILE *f = fopen(path, "a");
    const videoframe_t *frame = state->tile->frame;
    for (int yy = 0; yy < LCU_WIDTH; yy += 4)
      for (int xx = 0; xx < LCU_WIDTH; xx += 4) {
        if (x + xx >= frame->width || y + yy >= frame->height) continue;
        const cu_info_t *c = kvz_cu_array_at_const(frame->cu_array, x + xx, y + yy);
        const int inter = c->type == CU_INTER, l0 = inter && (c->inter.mv_dir & 1), l1 = inter && (c->inter.mv_dir & 2);
        fprintf(f, "%d %d %d %d %d %d %d %d %d %d %d %d %d %d %d %d %d %d %d %d\n", state->tile->offset_x + x + xx, state->tile->offset_y + y + yy, c->type, c->depth, c->part_size, c->tr_depth,
                inter ? c->skipped : 0, inter ? c->merged : 0, inter ? c->merge_idx : 0, c->cbf, inter ? 0 : c->intra.mode, inter ? c->inter.mv_dir : 0,
                l0 ? c->inter.mv[0][0] : 0, l0 ? c->inter.mv[0][1] : 0, l1 ? c->inter.mv[1][0] : 0, l1 ? c->inter.mv[1][1] : 0,
                l0 ? c->inter.mv_ref[0] : 255, l1 ? c->inter.mv_ref[1] : 255, l0 && !c->merged && !c->skipped ? c->inter.mv_cand0 : 0, l1 && !c->merged && !c->skipped ? c->inter.mv_cand1 : 0);
      }
    fclose(f);
  } else if (depth == 0 && path) {
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
