/*
 * kvz_oracle_sao.c -- TEST INFRASTRUCTURE (see kvz_oracle.h): CPU restatement of kvazaar's SAO parameter decision, LCU by LCU in
 * the encoder's own order, for the all-intra constant-QP configuration of the batched pass.
 *
 * Reference: sao.c:671 kvz_sao_search_lcu and everything below it (sao_search_luma / _chroma :588-668, sao_search_best_mode :491-586,
 * sao_search_edge_sao :364-441, sao_search_band_sao :443-478, calc_sao_band_offsets :214-268, calc_sao_bands :275-296, the mode-bit
 * functions :52-177), the syntax that moves the two SAO contexts (encoderstate.c:467-552 encode_sao) and the place in the per-LCU flow
 * (encoderstate.c:636-690): search -> deblock THIS LCU -> SAO search on the picture as it is at that moment -> code the LCU.  The
 * picture the statistics are taken on is therefore only partly deblocked (the LCU's right and bottom edges and the rightmost 4 samples of
 * its horizontal edges come later, filter.c:759-790); this file reproduces that by literally running kvz_oracle_deblock_lcu in LCU
 * order.  Bit costs: CABAC_FBITS_UPDATE on state->search_cabac with update == 0 (sao.c:55-72, cabac.h:133-139) -- a copy of the
 * row's contexts taken at the start of the LCU (search.c:1211) -- so the two contexts only move through the real syntax of
 * finished LCUs, in coding order, with the WPP hand-off of encoderstate.c:763-771.
 */
#include <limits.h>
#include <stdlib.h>
#include <string.h>

#include "kvz_oracle.h"

#define SAO_ABS_OFFSET_MAX 7 /* sao.h: (1 << (MIN(KVZ_BIT_DEPTH, 10) - 5)) - 1 */

typedef struct {
  int type, eo_class, band_position[2], offsets[10], ddistortion;
} sao_t;  /* sao_info_t (sao.h:55-63) without the merge flags */

typedef struct {
  const kvz_hip_intra_cost_model *m;
  uint8_t ctx_merge, ctx_type;  /* uc_state of sao_merge_flag_model / sao_type_idx_model */
} sao_cabac_t;

static float fbits(const sao_cabac_t *c, uint8_t state, int bin) { return c->m->entropy_fbits[state ^ bin]; }

/* sao.c:52-72 */
static double mode_bits_none(const sao_cabac_t *c, const sao_t *top, const sao_t *left)
{
  double bits = 0.0;
  if (left) bits += fbits(c, c->ctx_merge, 0);
  if (top) bits += fbits(c, c->ctx_merge, 0);
  bits += fbits(c, c->ctx_type, 0);
  return bits;
}
/* sao.c:74-87 */
static double mode_bits_merge(const sao_cabac_t *c, int merge_cand)
{
  double bits = 0.0;
  bits += fbits(c, c->ctx_merge, merge_cand == 1);
  if (merge_cand == 1) return bits;
  bits += fbits(c, c->ctx_merge, merge_cand == 2);
  return bits;
}
/* sao.c:90-129 */
static double mode_bits_edge(const sao_cabac_t *c, const int offsets[10], const sao_t *top, const sao_t *left, unsigned buf_cnt)
{
  double bits = 0.0;
  if (left) bits += fbits(c, c->ctx_merge, 0);
  if (top) bits += fbits(c, c->ctx_merge, 0);
  bits += fbits(c, c->ctx_type, 1);
  bits += 1.0;
  for (unsigned b = 0; b < buf_cnt; b++)
    for (int cat = 1; cat <= 4; cat++) {
      const int a = abs(offsets[cat + 5 * b]);
      bits += (a == 0 || a == SAO_ABS_OFFSET_MAX) ? a + 1 : a + 2;
    }
  bits += 2.0;
  return bits;
}
/* sao.c:132-177 */
static double mode_bits_band(const sao_cabac_t *c, const int offsets[10], const sao_t *top, const sao_t *left, unsigned buf_cnt)
{
  double bits = 0.0;
  if (left) bits += fbits(c, c->ctx_merge, 0);
  if (top) bits += fbits(c, c->ctx_merge, 0);
  bits += fbits(c, c->ctx_type, 1);
  bits += 1.0;
  for (unsigned b = 0; b < buf_cnt; b++)
    for (int i = 0; i < 4; i++) {
      const int a = abs(offsets[i + 1 + b * 5]);
      if (a == 0) bits += a + 1;
      else if (a == SAO_ABS_OFFSET_MAX) bits += a + 1 + 1;
      else bits += a + 2 + 1;
    }
  bits += 5.0 * buf_cnt;
  return bits;
}

/* sao.c:214-268 */
static int calc_band_offsets(int bands[2][32], int offsets[4], int *band_position)
{
  int dist[32], temp_offsets[32], best_dist, best_pos = 0;
  for (int band = 0; band < 32; band++) {
    best_dist = INT_MAX;
    int offset = 0;
    if (bands[1][band] != 0) {
      offset = (bands[0][band] + (bands[1][band] >> 1)) / bands[1][band];
      offset = offset < -SAO_ABS_OFFSET_MAX ? -SAO_ABS_OFFSET_MAX : (offset > SAO_ABS_OFFSET_MAX ? SAO_ABS_OFFSET_MAX : offset);
    }
    dist[band] = offset == 0 ? 0 : INT_MAX;
    temp_offsets[band] = 0;
    while (offset != 0) {
      const int temp = bands[1][band] * offset * offset - 2 * offset * bands[0][band];
      if (temp < best_dist) {  /* best_dist is never lowered inside the loop (sao.c:244-248): the LAST improving offset over INT_MAX wins */
        dist[band] = temp;
        temp_offsets[band] = offset;
      }
      offset += offset > 0 ? -1 : 1;
    }
  }
  best_dist = INT_MAX;
  for (int band = 0; band < 28; band++) {
    /* the reference sums four ints, any of which may be INT_MAX: signed overflow in C; two's-complement wrap is what the compiled reference does */
    const int temp = (int)((unsigned)dist[band] + (unsigned)dist[band + 1] + (unsigned)dist[band + 2] + (unsigned)dist[band + 3]);
    if (temp < best_dist) { best_dist = temp; best_pos = band; }
  }
  memcpy(offsets, &temp_offsets[best_pos], 4 * sizeof(int));
  *band_position = best_pos;
  return best_dist;
}

/* sao.c:364-441 */
static void search_edge(const sao_cabac_t *c, const uint8_t *data[], const uint8_t *rec[], int bw, int bh, unsigned buf_cnt, sao_t *out, const sao_t *top,
                        const sao_t *left)
{
  out->type = 2;
  out->ddistortion = INT_MAX;
  for (int ec = 0; ec < 4; ec++) {
    int edge_offset[10], sum_dd = 0;
    for (unsigned i = 0; i < buf_cnt; i++) {
      int csc[10];
      memset(csc, 0, sizeof csc);
      kvz_oracle_calc_sao_edge_dir(8, data[i], rec[i], ec, bw, bh, csc);
      for (int cat = 1; cat <= 4; cat++) {
        const int cat_sum = csc[cat], cat_cnt = csc[5 + cat];
        int offset = 0;
        if (cat_cnt != 0) {
          offset = (cat_sum + (cat_cnt >> 1)) / cat_cnt;
          offset = offset < -SAO_ABS_OFFSET_MAX ? -SAO_ABS_OFFSET_MAX : (offset > SAO_ABS_OFFSET_MAX ? SAO_ABS_OFFSET_MAX : offset);
        }
        if (cat <= 2 && offset < 0) offset = 0;
        if (cat >= 3 && offset > 0) offset = 0;
        edge_offset[cat + 5 * i] = offset;
        sum_dd += cat_cnt * offset * offset - 2 * offset * cat_sum;
      }
    }
    {
      /* edge_offset[0] / [5] are still unset here in the reference too; mode_bits_edge only reads categories 1..4 */
      const float mode_bits = (float)mode_bits_edge(c, edge_offset, top, left, buf_cnt);
      sum_dd += (int)((double)mode_bits * c->m->lambda + 0.5);
    }
    edge_offset[0] = 0;
    edge_offset[5] = 0;
    if (sum_dd < out->ddistortion) {
      out->eo_class = ec;
      out->ddistortion = sum_dd;
      memcpy(out->offsets, edge_offset, sizeof edge_offset);
    }
  }
}

/* sao.c:443-478 (with calc_sao_bands :275-296) */
static void search_band(const sao_cabac_t *c, const uint8_t *data[], const uint8_t *rec[], int bw, int bh, unsigned buf_cnt, sao_t *out, const sao_t *top,
                        const sao_t *left)
{
  out->type = 1;
  out->ddistortion = INT_MAX;
  int temp_offsets[10], dd = 0;
  memset(temp_offsets, 0, sizeof temp_offsets);  /* [0] and [5] are uninitialised stack in the reference; sao_search_best_mode presets the copies' [0] / [5] to 0 and the memcpy below overwrites them with whatever the stack held -- they are never read for band SAO (offset ids 1..4 / 6..9) */
  for (unsigned i = 0; i < buf_cnt; i++) {
    int bands[2][32];
    memset(bands, 0, sizeof bands);
    for (int p = 0; p < bw * bh; p++) {
      const int idx = rec[i][p] >> 3;
      bands[0][idx] += data[i][p] - rec[i][p];
      bands[1][idx]++;
    }
    dd += calc_band_offsets(bands, &temp_offsets[1 + 5 * i], &out->band_position[i]);
  }
  const float rate = (float)mode_bits_band(c, temp_offsets, top, left, buf_cnt);
  dd += (int)((double)rate * c->m->lambda + 0.5);
  if (dd < out->ddistortion) {
    out->type = 1;
    out->ddistortion = dd;
    memcpy(out->offsets, temp_offsets, sizeof(int) * buf_cnt * 5);
  }
}

/* sao.c:491-586, cfg.sao_type == 3 (edge and band) */
static void search_best_mode(const sao_cabac_t *c, const uint8_t *data[], const uint8_t *rec[], int bw, int bh, unsigned buf_cnt, sao_t *out, const sao_t *top,
                             const sao_t *left, int32_t merge_cost[3])
{
  sao_t edge, band;
  memset(&edge, 0, sizeof edge);
  memset(&band, 0, sizeof band);
  search_edge(c, data, rec, bw, bh, buf_cnt, &edge, top, left);
  {
    const float mode_bits = (float)mode_bits_edge(c, edge.offsets, top, left, buf_cnt);
    int dd = (int)(mode_bits * c->m->lambda + 0.5);
    for (unsigned i = 0; i < buf_cnt; i++) dd += kvz_oracle_sao_edge_ddistortion(8, data[i], rec[i], bw, bh, edge.eo_class, &edge.offsets[5 * i]);
    edge.ddistortion = dd;
  }
  search_band(c, data, rec, bw, bh, buf_cnt, &band, top, left);
  {
    const float mode_bits = (float)mode_bits_band(c, band.offsets, top, left, buf_cnt);
    int dd = (int)(mode_bits * c->m->lambda + 0.5);
    for (unsigned i = 0; i < buf_cnt; i++) dd += kvz_oracle_sao_band_ddistortion(8, data[i], rec[i], bw, bh, band.band_position[i], &band.offsets[1 + 5 * i]);
    band.ddistortion = dd;
  }
  if (edge.ddistortion <= band.ddistortion) { *out = edge; merge_cost[0] = edge.ddistortion; }
  else { *out = band; merge_cost[0] = band.ddistortion; }
  {
    const float none_bits = (float)mode_bits_none(c, top, left);
    const int cost_of_nothing = (int)(none_bits * c->m->lambda + 0.5);
    if (out->ddistortion >= cost_of_nothing) { out->type = 0; merge_cost[0] = cost_of_nothing; }
  }
  if (top || left) {
    const sao_t *cands[2] = { left, top };
    for (int i = 0; i < 2; i++) {
      const sao_t *mc = cands[i];
      if (!mc) continue;
      const float mode_bits = (float)mode_bits_merge(c, i + 1);
      int dd = (int)(mode_bits * c->m->lambda + 0.5);
      if (mc->type == 2) for (unsigned b = 0; b < buf_cnt; b++) dd += kvz_oracle_sao_edge_ddistortion(8, data[b], rec[b], bw, bh, mc->eo_class, &mc->offsets[5 * b]);
      else if (mc->type == 1) for (unsigned b = 0; b < buf_cnt; b++) dd += kvz_oracle_sao_band_ddistortion(8, data[b], rec[b], bw, bh, mc->band_position[b], &mc->offsets[1 + 5 * b]);
      merge_cost[i + 1] = dd;
    }
  }
}

static void blit(const uint8_t *src, uint8_t *dst, int w, int h, int stride) { for (int y = 0; y < h; y++) memcpy(dst + y * w, src + (size_t)y * stride, w); }

/* H.265 9.3.4.3 state transition on a coded bin (encoderstate.c:467-552 only moves these two contexts) */
static void code_bin(uint8_t *st, int bin)
{
  const uint8_t *mps = kvz_oracle_next_state_table(0), *lps = kvz_oracle_next_state_table(1);
  *st = (bin != (*st & 1)) ? lps[*st] : mps[*st];
}

static void export_params(const sao_t *s, kvz_hip_sao_params *o, int planes)
{
  memset(o, 0, sizeof *o);
  o->bitdepth = 8;
  if (s->type == 0) return;  /* a NONE record keeps the searched fields in the reference (sao.c:545-548 only resets the type); nothing ever reads them */
  o->type = s->type; o->eo_class = s->eo_class; o->band_position[0] = s->band_position[0]; o->band_position[1] = s->band_position[1];
  memcpy(o->offsets, s->offsets, sizeof o->offsets);
  if (planes == 1) { o->band_position[1] = 0; memset(&o->offsets[5], 0, 5 * sizeof(int)); }  /* luma records only use the first half (the second is stack garbage in the reference) */
  if (s->type == 2) o->band_position[0] = o->band_position[1] = 0; else o->eo_class = 0;      /* fields of the other SAO type: never read */
  o->bitdepth = 8;
}

static void sao_search_frame(const kvz_hip_intra_cost_model *m, int width, int height, const uint8_t *src, uint8_t *rec, const uint8_t *cu_depth,
                             const kvz_hip_cu_dbk *info, int slice_is_b, int deblock, int beta_offset_div2, int tc_offset_div2, kvz_hip_sao_params *luma_out,
                             kvz_hip_sao_params *chroma_out, uint8_t *merge_out);
void kvz_oracle_sao_search_frame(const kvz_hip_intra_cost_model *m, int width, int height, const uint8_t *src, uint8_t *rec, const uint8_t *cu_depth,
                                 int deblock, int beta_offset_div2, int tc_offset_div2, kvz_hip_sao_params *luma_out, kvz_hip_sao_params *chroma_out, uint8_t *merge_out)
{
  sao_search_frame(m, width, height, src, rec, cu_depth, NULL, 0, deblock, beta_offset_div2, tc_offset_div2, luma_out, chroma_out, merge_out);
}
/* the same on a picture of a sequence with inter prediction: the deblocking step takes its edges and strengths from `info` (one record per 4x4 unit) */
void kvz_oracle_sao_search_frame_inter(const kvz_hip_intra_cost_model *m, int width, int height, const uint8_t *src, uint8_t *rec, const kvz_hip_cu_dbk *info,
                                       int slice_is_b, int deblock, int beta_offset_div2, int tc_offset_div2, kvz_hip_sao_params *luma_out,
                                       kvz_hip_sao_params *chroma_out, uint8_t *merge_out)
{
  sao_search_frame(m, width, height, src, rec, NULL, info, slice_is_b, deblock, beta_offset_div2, tc_offset_div2, luma_out, chroma_out, merge_out);
}
static void sao_search_frame(const kvz_hip_intra_cost_model *m, int width, int height, const uint8_t *src, uint8_t *rec, const uint8_t *cu_depth,
                             const kvz_hip_cu_dbk *info, int slice_is_b, int deblock, int beta_offset_div2, int tc_offset_div2, kvz_hip_sao_params *luma_out,
                             kvz_hip_sao_params *chroma_out, uint8_t *merge_out)
{
  const int wl = (width + 63) / 64, hl = (height + 63) / 64, cw = width / 2, ch = height / 2;
  const size_t ys = (size_t)width * height, cs = ys / 4;
  uint8_t *ry = rec, *ru = rec + ys, *rv = rec + ys + cs;
  sao_t *luma = calloc((size_t)wl * hl, sizeof *luma), *chroma = calloc((size_t)wl * hl, sizeof *chroma);
  sao_cabac_t row_start = { m, m->ctx_init[KVZ_HIP_CX_SAO_MERGE], m->ctx_init[KVZ_HIP_CX_SAO_TYPE] }, cab = row_start, next_row = row_start;
  for (int ly = 0; ly < hl; ly++) {
    /* WPP: a row starts from the contexts after the SECOND LCU of the row above (encoderstate.c:763-771; from the first when the picture is one
     * LCU wide: lcu->index == 1 never happens and the row keeps its slice-start state); without WPP the coder simply runs on */
    if (!m->no_wpp) cab = ly == 0 ? row_start : next_row;
    for (int lx = 0; lx < wl; lx++) {
      if (deblock && info) kvz_oracle_deblock_lcu_inter(width, height, m->qp, beta_offset_div2, tc_offset_div2, ry, ru, rv, info, slice_is_b, lx * 64, ly * 64);
      else if (deblock) kvz_oracle_deblock_lcu(width, height, m->qp, beta_offset_div2, tc_offset_div2, ry, ru, rv, cu_depth, lx * 64, ly * 64);
      sao_t *sl = &luma[ly * wl + lx], *sc = &chroma[ly * wl + lx];
      const sao_t *top_l = ly ? &luma[(ly - 1) * wl + lx] : NULL, *left_l = lx ? &luma[ly * wl + lx - 1] : NULL;
      const sao_t *top_c = ly ? &chroma[(ly - 1) * wl + lx] : NULL, *left_c = lx ? &chroma[ly * wl + lx - 1] : NULL;
      int32_t mc_l[3] = { INT_MAX, 0, 0 }, mc_c[3] = { INT_MAX, 0, 0 };
      {  /* sao.c:632-668 sao_search_luma */
        uint8_t orig[4096], rc[4096];
        const int bw = lx * 64 + 64 >= width ? width - lx * 64 : 64, bh = ly * 64 + 64 >= height ? height - ly * 64 : 64;
        blit(src + (size_t)ly * 64 * width + lx * 64, orig, bw, bh, width);
        blit(ry + (size_t)ly * 64 * width + lx * 64, rc, bw, bh, width);
        const uint8_t *ol[1] = { orig }, *rl[1] = { rc };
        search_best_mode(&cab, ol, rl, bw, bh, 1, sl, top_l, left_l, mc_l);
      }
      {  /* sao.c:588-630 sao_search_chroma */
        uint8_t orig[2][1024], rc[2][1024];
        const int bw = lx * 32 + 32 >= cw ? (width - lx * 64) / 2 : 32, bh = ly * 32 + 32 >= ch ? (height - ly * 64) / 2 : 32;
        for (int p = 0; p < 2; p++) {
          blit(src + ys + p * cs + (size_t)ly * 32 * cw + lx * 32, orig[p], bw, bh, cw);
          blit((p ? rv : ru) + (size_t)ly * 32 * cw + lx * 32, rc[p], bw, bh, cw);
        }
        const uint8_t *ol[2] = { orig[0], orig[1] }, *rl[2] = { rc[0], rc[1] };
        search_best_mode(&cab, ol, rl, bw, bh, 2, sc, top_c, left_c, mc_c);
      }
      int merge = 0;  /* sao.c:712-735 */
      if (top_l && (int)((unsigned)mc_l[2] + (unsigned)mc_c[2]) <= (int)((unsigned)mc_l[0] + (unsigned)mc_c[0])) { *sl = *top_l; *sc = *top_c; merge = 2; }
      if (left_l && (int)((unsigned)mc_l[1] + (unsigned)mc_c[1]) <= (int)((unsigned)mc_l[0] + (unsigned)mc_c[0]))
        if (merge != 2 || (int)((unsigned)mc_l[1] + (unsigned)mc_c[1]) < (int)((unsigned)mc_l[2] + (unsigned)mc_c[2])) { *sl = *left_l; *sc = *left_c; merge = 1; }
      merge_out[ly * wl + lx] = (uint8_t)merge;
      /* the LCU's SAO syntax on the row's coder (encoderstate.c:519-552): merge flags, then -- unless merged -- the type bin of luma and
       * of chroma (U codes it for U and V); everything else is bypass-coded */
      if (lx > 0) code_bin(&cab.ctx_merge, merge == 1);
      if (ly > 0 && merge != 1) code_bin(&cab.ctx_merge, merge == 2);
      if (!merge) { code_bin(&cab.ctx_type, sl->type != 0); code_bin(&cab.ctx_type, sc->type != 0); }
      if (lx == 1 || (wl == 1 && 0)) next_row = cab;
    }
  }
  for (int i = 0; i < wl * hl; i++) { export_params(&luma[i], &luma_out[i], 1); export_params(&chroma[i], &chroma_out[i], 2); }
  free(luma);
  free(chroma);
}
