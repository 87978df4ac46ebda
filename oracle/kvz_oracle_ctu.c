/*
 * kvz_oracle_ctu.c -- TEST INFRASTRUCTURE (see kvz_oracle.h).  CPU restatement of the all-intra CTU search +
 * reconstruction that the batched device pass (include/kvz_hip_batch.h, kvz_hip_intra_frames) implements.
 *
 * It restates kvazaar v2.3.2's per-CTU flow for an I slice under the `ultrafast` preset
 *   (cfg.c:485-512: rd=0, pu-depth-intra 2-3, rdoq 0, signhide 0, transform-skip 0, fast-residual-cost > QP,
 *    cu-split-termination zero, combine_intra_cus on (cfg.c:187), tr_depth_intra 0, 4:2:0, 8 bit)
 * function by function, recursion and work-tree copies included:
 *   search_cu                      search.c:646-1063     -> search_cu()
 *   kvz_search_cu_intra            search_intra.c:812    -> search_cu_intra()
 *   search_intra_rough             search_intra.c:391    -> rough_search()
 *   kvz_luma_mode_bits             search_intra.c:641    -> luma_mode_bits()
 *   kvz_intra_get_dir_luma_predictor  intra.c:84         -> mpm_candidates()
 *   kvz_intra_build_reference_any  intra.c:305           -> build_reference()
 *   kvz_intra_predict              intra.c:252           -> intra_predict()
 *   kvz_intra_recon_cu             intra.c:623           -> recon_cu()
 *   kvz_quantize_lcu_residual      transform.c:439       -> (inside recon_cu: leaf TUs only, tr_depth_intra = 0)
 *   kvz_mock_encode_coding_unit    encode_coding_tree.c:948 + encode_intra_coding_unit :467 -> cu_bits()
 *   calc_mode_bits                 search.c:517          -> calc_mode_bits()
 *   cu_rd_cost_tr_split_accurate   search.c:425          -> rd_cost()
 * The pixel/coefficient kernels are the pinned kvz_oracle_* functions of kvz_oracle.c.
 *
 * CABAC contexts (kvz_hip_intra_cost_model::adaptive, the default): the ten contexts this configuration prices syntax with
 * follow kvazaar's life cycle -- search_cabac = copy of the row's state->cabac per CTU (search.c:1211), updates only around
 * the mock encode and cu_rd_cost_tr_split_accurate of an evaluated CU (search.c:895-940), the pre / post / temp copies of
 * search_cu (search.c:655, 956-959, 1005-1041, 1051), the real syntax of each finished CTU on the row's state
 * (code_coding_tree below), WPP seeding of the next row from a row's second CTU (encoderstate.c:763-771).  The mock encode's
 * left-neighbour lookup at the LCU's left edge is followed as well (intra_mode_syntax_bits).  PINNED END TO END: the pass
 * (+ kvz_oracle_deblock_frame) reproduces the reconstruction `kvazaar --preset ultrafast -p 1 --debug` writes, picture for
 * picture, for QP 12..45 (tests/test_encoder_parity.py, tests/golden/encoder_recon.json).  From QP 28 on `ultrafast` prices
 * coefficients by running the residual coder in counting mode (fast_residual_cost_limit, rdo.c:311-340): encode_coeff_nxn() below,
 * pinned against the reference's kvz_encode_coeff_nxn on random blocks and context states (tests/test_oracle_vs_ref.py).
 * adaptive == 0 freezes every context at its slice-start state (CTUs then interact through pixels and CU info only).
 *
 * The other presets are the same flow with more switched on (kvz_hip_intra_cost_model): coeff_cabac at every QP (`faster`), search_32x32 (`fast`: pu-depth-intra
 * 1-3), rdoq (`medium`: kvz_rdoq in every quantisation, kvz_oracle_rdoq.c) and search_nxn (`medium`: pu-depth-intra 1-4 -- depth 4 of search_cu, the four 4x4 PUs
 * of an 8x8 CU: search.c:691, 794, 906-913, 970-974; intra.c:573-577 and transform.c:306-328 for the 4x4 chroma blocks done with the first PU;
 * quant-generic.c:237-238 for kvz_rdoq's tr_depth; encode_coding_tree.c:148-163, 205-225, 505-560 for the coded syntax).  Each is pinned the same way: the
 * reconstruction of `kvazaar --preset <p> -p 1 --debug` before the loop filters, after deblocking and after SAO (tests/test_encoder_parity.py).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "kvz_oracle.h"

struct inter_seq;

/* context.c:202-213 kvz_ctx_init: init value + QP -> uc_state */
static int ctx_state(int qp, int init_value)
{
  int slope = (init_value >> 4) * 5 - 45, offset = ((init_value & 15) << 3) - 16;
  int st = ((slope * qp) >> 4) + offset;
  st = st < 1 ? 1 : st > 126 ? 126 : st;
  return st >= 64 ? ((st - 64) << 1) + 1 : (63 - st) << 1;
}

/* The contexts the all-intra ultrafast search prices syntax with (cabac.h:63-100), as uc_state = (state << 1) | mps: ten for the
 * CU / transform-tree syntax, and -- only touched when coefficients are priced with the CABAC model (coeff_cabac, QP >= 28) --
 * the residual-coding contexts in the order of cabac.h:78-88 */
enum { CX_SPLIT = KVZ_HIP_CX_SPLIT /* ..2 */, CX_PART = KVZ_HIP_CX_PART, CX_INTRA = KVZ_HIP_CX_INTRA, CX_CHROMA = KVZ_HIP_CX_CHROMA,
       CX_CBF_LUMA = KVZ_HIP_CX_CBF_LUMA /* ..7 */, CX_CBF_CHROMA = KVZ_HIP_CX_CBF_CHROMA /* ..9 */,
       CX_SIG_CG = KVZ_HIP_CX_SIG_CG /* 4 */, CX_SIG_LUMA = KVZ_HIP_CX_SIG_LUMA /* 27 */, CX_SIG_CHROMA = KVZ_HIP_CX_SIG_CHROMA /* 15 */,
       CX_LAST_Y_LUMA = KVZ_HIP_CX_LAST_Y_LUMA /* 15 */, CX_LAST_Y_CHROMA = KVZ_HIP_CX_LAST_Y_CHROMA /* 15 */,
       CX_LAST_X_LUMA = KVZ_HIP_CX_LAST_X_LUMA /* 15 */, CX_LAST_X_CHROMA = KVZ_HIP_CX_LAST_X_CHROMA /* 15 */,
       CX_ONE_LUMA = KVZ_HIP_CX_ONE_LUMA /* 16 */, CX_ONE_CHROMA = KVZ_HIP_CX_ONE_CHROMA /* 8 */, CX_ABS_LUMA = KVZ_HIP_CX_ABS_LUMA /* 4 */,
       CX_ABS_CHROMA = KVZ_HIP_CX_ABS_CHROMA /* 2 */, CX_COUNT = KVZ_HIP_CX_COUNT };
/* the contexts only P / B slices use (kvz_oracle_inter.inc; cabac.h:63-77) live behind the device-visible ones (and the two SAO contexts at 148 / 149) */
enum { CXB_SKIP = 150 /* ..152 */, CXB_MERGE_FLAG = 153, CXB_MERGE_IDX = 154, CXB_PRED_MODE = 155, CXB_MVD = 156 /* ..157 */, CXB_MVP_IDX = 158 /* ..159 */,
       CXB_INTER_DIR = 160 /* ..164 */, CXB_ROOT_CBF = 165, CXB_TRANS_SUBDIV = 166 /* ..168 */, CXB_REF_PIC = 169 /* ..170 */, CX_ALL = 172 };
typedef struct { uint8_t s[CX_ALL]; } ctxs_t;

/* H.265 Table 9-41 state transitions in kvazaar's packing (cabac.c:40-62 kvz_g_auc_next_state_mps / _lps) */
static uint8_t g_next_mps[128], g_next_lps[128];
static void build_transitions(void)
{
  static const uint8_t trans_lps[64] = { 0, 0, 1, 2, 2, 4, 4, 5, 6, 7, 8, 9, 9, 11, 11, 12, 13, 13, 15, 15, 16, 16, 18, 18, 19, 19, 21, 21, 22, 22, 23, 24,
                                          24, 25, 26, 26, 27, 27, 28, 29, 29, 30, 30, 30, 31, 32, 32, 33, 33, 33, 34, 34, 35, 35, 35, 36, 36, 36, 37, 37, 37, 38, 38, 63 };
  for (int u = 0; u < 128; u++) {
    const int st = u >> 1, mps = u & 1;
    g_next_mps[u] = (uint8_t)(st >= 62 ? u : ((st + 1) << 1) | mps);
    g_next_lps[u] = (uint8_t)(st == 0 ? (0 << 1) | (mps ^ 1) : (trans_lps[st] << 1) | mps);
  }
  g_next_lps[126] = 127; g_next_lps[127] = 126;  /* the terminating state, never reached by these contexts */
}
const uint8_t *kvz_oracle_next_state_table(int lps) { build_transitions(); return lps ? g_next_lps : g_next_mps; }

#define ORC_CLIP(lo, hi, v) ((v) < (lo) ? (lo) : ((v) > (hi) ? (hi) : (v)))
#define LCU 64
#define NLEVELS 5
#define MAX_COST 1.7e+308 /* global.h:293 MAX_DOUBLE */

/* kvz_oracle_cu (kvz_oracle.h): type 0 not set, 1 intra, 2 inter (cu.h:72-77).  Motion fields of a list mv_dir does not use are kept at 0 / 255: the reference
 * leaves them undefined and clears them whenever a neighbour is taken as a candidate (inter.c:669-677 inter_clear_cu_unused); nothing else reads them */
typedef kvz_oracle_cu cu_t;

typedef struct {
  uint8_t rec[3][LCU * LCU];
  int16_t coeff[3][LCU * LCU];
  cu_t cu[16 * 16]; /* one per 4x4 SCU */
} level_t;

typedef struct {
  const kvz_hip_intra_cost_model *m;
  int W, H;                  /* luma frame size */
  const uint8_t *src[3];     /* source planes, stride W (luma) / W/2 */
  uint8_t *frec[3];          /* frame reconstruction */
  uint8_t *fdepth, *fmode;   /* frame CU info per 8x8 (stride W/8) */
  uint8_t *fmode4, *fnxn;    /* search_nxn: luma mode per 4x4 PU (stride W/4) and "part_size == NxN" per 8x8 CU (stride W/8); owned by kvz_oracle_intra_frame */
  int cx, cy;                /* luma origin of the current CTU */
  uint8_t org[3][LCU * LCU]; /* lcu->ref, zero outside the picture (search.c:1084 FILL) */
  level_t lv[NLEVELS];
  uint8_t tbl_top[16][16], tbl_left[16][16];
  ctxs_t cab;                /* state->search_cabac's contexts (adaptive mode) */
  ctxs_t coder;              /* state->cabac's contexts while this CTU is searched: what kvz_rdoq prices on (rdo.c:665) */
  /* sequences with inter pictures (kvz_oracle_inter.inc): the frame's cu_array per 4x4 unit (neighbours outside the CTU come from it when it is set),
   * the slice type and what the inter search reads of the reference picture */
  cu_t *fcu;
  int slice_b;
  const struct inter_seq *in;
} ctu_t;

/* CABAC_FBITS_UPDATE (cabac.h:133-139) on context idx of t->cab: the price of `bin`, then -- if `update` -- the state change
 * kvz_cabac_encode_bin applies (cabac.c:104-132).  Frozen mode prices from the model's float tables instead. */
static double ctx_price(ctu_t *t, int idx, int bin, int update, float frozen_price)
{
  if (!t->m->adaptive) return frozen_price;
  uint8_t *st = &t->cab.s[idx];
  const double bits = t->m->entropy_fbits[*st ^ bin];
  if (update) *st = (bin != (*st & 1)) ? g_next_lps[*st] : g_next_mps[*st];
  return bits;
}
static void ctx_code(ctxs_t *c, int idx, int bin) { uint8_t *st = &c->s[idx]; *st = (bin != (*st & 1)) ? g_next_lps[*st] : g_next_mps[*st]; }

/* ---- residual coding: kvz_encode_coeff_nxn (strategies/generic/encode_coding_tree-generic.c:40-283) in counting mode ---- */
static const uint8_t g_group_idx[32] = { 0, 1, 2, 3, 4, 4, 5, 5, 6, 6, 6, 6, 7, 7, 7, 7, 8, 8, 8, 8, 8, 8, 8, 8, 9, 9, 9, 9, 9, 9, 9, 9 }; /* encoderstate.h:397 */

/* CABAC_FBITS_UPDATE (cabac.h:133-139) inside the residual coder: the price of `bin` on context idx of *c, then -- only while the
 * coder's contexts have `update` on (they are a copy of the search's, flag included) -- the state change */
static double bin_cost(const kvz_hip_intra_cost_model *m, ctxs_t *c, int update, int idx, int bin)
{
  const double bits = m->entropy_fbits[c->s[idx] ^ bin];
  if (update) ctx_code(c, idx, bin);
  return bits;
}
/* encode_coding_tree.c:63-115 kvz_encode_last_significant_xy */
static double last_xy_cost(const kvz_hip_intra_cost_model *m, ctxs_t *c, int update, int lastpos_x, int lastpos_y, int width, int type, int scan)
{
  int index = 0;
  while ((4 << index) < width) index++;
  const int ctx_offset = type ? 0 : (index * 3 + (index + 1) / 4), shift = type ? index : (index + 3) / 4;
  const int base_x = type ? CX_LAST_X_CHROMA : CX_LAST_X_LUMA, base_y = type ? CX_LAST_Y_CHROMA : CX_LAST_Y_LUMA;
  double bits = 0;
  if (scan == 2) { const int tmp = lastpos_x; lastpos_x = lastpos_y; lastpos_y = tmp; }
  const int gx = g_group_idx[lastpos_x], gy = g_group_idx[lastpos_y];
  for (int i = 0; i < gx; i++) bits += bin_cost(m, c, update, base_x + ctx_offset + (i >> shift), 1);
  if (gx < g_group_idx[width - 1]) bits += bin_cost(m, c, update, base_x + ctx_offset + (gx >> shift), 0);
  for (int i = 0; i < gy; i++) bits += bin_cost(m, c, update, base_y + ctx_offset + (i >> shift), 1);
  if (gy < g_group_idx[width - 1]) bits += bin_cost(m, c, update, base_y + ctx_offset + (gy >> shift), 0);
  if (gx > 3) bits += (gx - 2) / 2;  /* suffixes: bypass bins */
  if (gy > 3) bits += (gy - 2) / 2;
  return bits;
}
/* context.c:366-399 kvz_context_get_sig_ctx_inc */
static int sig_ctx_inc(int pattern_sig_ctx, int scan_idx, int pos_x, int pos_y, int log2_size, int type)
{
  static const int ctx_ind_map[16] = { 0, 1, 4, 5, 2, 3, 4, 5, 6, 6, 8, 8, 7, 7, 8, 8 };
  if (pos_x + pos_y == 0) return 0;
  if (log2_size == 2) return ctx_ind_map[4 * pos_y + pos_x];
  const int offset = log2_size == 3 ? (scan_idx == 0 ? 9 : 15) : (type == 0 ? 21 : 12);
  const int xs = pos_x & 3, ys = pos_y & 3;
  int cnt;
  if (pattern_sig_ctx == 0) cnt = xs + ys <= 2 ? (xs + ys == 0 ? 2 : 1) : 0;
  else if (pattern_sig_ctx == 1) cnt = ys <= 1 ? (ys == 0 ? 2 : 1) : 0;
  else if (pattern_sig_ctx == 2) cnt = xs <= 1 ? (xs == 0 ? 2 : 1) : 0;
  else cnt = 2;
  return ((type == 0 && ((pos_x >> 2) + (pos_y >> 2)) > 0) ? 3 : 0) + offset + cnt;
}
/* cabac.c:275-301 kvz_cabac_write_coeff_remain: number of bypass bins */
static int coeff_remain_bits(int symbol, int r_param)
{
  if (symbol < (3 << r_param)) return (symbol >> r_param) + 1 + r_param;
  int length = r_param, code = symbol - (3 << r_param);
  while (code >= (1 << length)) { code -= 1 << length; length++; }
  return 3 + length + 1 - r_param + length;
}
/* Bits of one transform block's residual syntax; with `update`, *c moves as the coder's contexts do.  sign data hiding, transform skip and
 * encryption are off in this configuration (cfg.c:485-512).  type: 0 luma, 2 chroma; scan_mode: 0 diagonal, 1 horizontal, 2 vertical */
static double encode_coeff_nxn(const kvz_hip_intra_cost_model *m, ctxs_t *c, int update, const int16_t *coeff, int width, int type, int scan_mode)
{
  int log2_size = 2;
  while ((1 << log2_size) < width) log2_size++;
  const int num_blk_side = width >> 2;
  const uint32_t *scan = kvz_oracle_scan_table(scan_mode, log2_size);
  /* g_sig_last_scan_cg (tables.h:84-89): the order of the 4x4 groups.  16x16 and 32x32 blocks are always scanned diagonally; the
   * 8x8 grid of a 32x32 block in plain up-right diagonal order (the 8x8 coefficient scan is hierarchical, not this) */
  static const uint32_t one_cg[1] = { 0 };
  uint32_t diag8[64];
  const uint32_t *scan_cg = log2_size == 2 ? one_cg : kvz_oracle_scan_table(log2_size == 3 ? scan_mode : 0, log2_size - 2);
  if (log2_size == 5) {
    int n = 0;
    for (int d = 0; d < 15; d++) for (int x = 0; x <= d; x++) if (x < 8 && d - x < 8) diag8[n++] = (uint32_t)((d - x) * 8 + x);
    scan_cg = diag8;
  }
  uint8_t sig_cg[64] = { 0 };
  double bits = 0;
  for (int cy = 0; cy < num_blk_side; cy++)
    for (int cx = 0; cx < num_blk_side; cx++)
      for (int i = 0; i < 16; i++) if (coeff[(cy * 4 + (i >> 2)) * width + cx * 4 + (i & 3)]) { sig_cg[cy * num_blk_side + cx] = 1; break; }
  int scan_cg_last = num_blk_side * num_blk_side - 1;
  while (!sig_cg[scan_cg[scan_cg_last]]) scan_cg_last--;
  int scan_pos_last = scan_cg_last * 16 + 15;
  while (!coeff[scan[scan_pos_last]]) scan_pos_last--;
  const int pos_last = (int)scan[scan_pos_last];
  bits += last_xy_cost(m, c, update, pos_last & (width - 1), pos_last >> log2_size, width, type, scan_mode);
  const int base_sig = type == 0 ? CX_SIG_LUMA : CX_SIG_CHROMA;
  int scan_pos_sig = scan_pos_last, c1 = 1;
  for (int i = scan_cg_last; i >= 0; i--) {
    const int sub_pos = i << 4, cg_blk_pos = (int)scan_cg[i], cg_pos_y = cg_blk_pos / num_blk_side, cg_pos_x = cg_blk_pos - cg_pos_y * num_blk_side;
    int abs_coeff[16], num_non_zero = 0, go_rice = 0;
    if (scan_pos_sig == scan_pos_last) { abs_coeff[0] = abs(coeff[pos_last]); num_non_zero = 1; scan_pos_sig--; }
    const int right = cg_pos_x < num_blk_side - 1 && sig_cg[cg_pos_y * num_blk_side + cg_pos_x + 1];
    const int lower = cg_pos_y < num_blk_side - 1 && sig_cg[(cg_pos_y + 1) * num_blk_side + cg_pos_x];
    if (i == scan_cg_last || i == 0) sig_cg[cg_blk_pos] = 1;
    else bits += bin_cost(m, c, update, CX_SIG_CG + type + (right || lower), sig_cg[cg_blk_pos]);  /* context.c:315-327 */
    if (sig_cg[cg_blk_pos]) {
      const int pattern = width == 4 ? -1 : right + (lower << 1);  /* context.c:339-351 */
      for (; scan_pos_sig >= sub_pos; scan_pos_sig--) {
        const int blk_pos = (int)scan[scan_pos_sig], pos_y = blk_pos >> log2_size, pos_x = blk_pos - (pos_y << log2_size), sig = coeff[blk_pos] != 0;
        if (scan_pos_sig > sub_pos || i == 0 || num_non_zero) bits += bin_cost(m, c, update, base_sig + sig_ctx_inc(pattern, scan_mode, pos_x, pos_y, log2_size, type), sig);
        if (sig) abs_coeff[num_non_zero++] = abs(coeff[blk_pos]);
      }
    } else scan_pos_sig = sub_pos - 1;
    if (num_non_zero > 0) {
      int ctx_set = (i > 0 && type == 0) ? 2 : 0;
      if (c1 == 0) ctx_set++;
      c1 = 1;
      const int base_one = (type == 0 ? CX_ONE_LUMA : CX_ONE_CHROMA) + 4 * ctx_set, num_c1 = num_non_zero < 8 ? num_non_zero : 8;
      int first_c2 = -1;
      for (int idx = 0; idx < num_c1; idx++) {
        const int symbol = abs_coeff[idx] > 1;
        bits += bin_cost(m, c, update, base_one + c1, symbol);
        if (symbol) { c1 = 0; if (first_c2 == -1) first_c2 = idx; }
        else if (c1 < 3 && c1 > 0) c1++;
      }
      if (c1 == 0 && first_c2 != -1) bits += bin_cost(m, c, update, (type == 0 ? CX_ABS_LUMA : CX_ABS_CHROMA) + ctx_set, abs_coeff[first_c2] > 2);
      bits += num_non_zero;  /* signs */
      if (c1 == 0 || num_non_zero > 8) {
        int first_coeff2 = 1;
        for (int idx = 0; idx < num_non_zero; idx++) {
          const int base_level = idx < 8 ? 2 + first_coeff2 : 1;
          if (abs_coeff[idx] >= base_level) {
            bits += coeff_remain_bits(abs_coeff[idx] - base_level, go_rice);
            if (abs_coeff[idx] > 3 * (1 << go_rice)) go_rice = go_rice + 1 < 4 ? go_rice + 1 : 4;
          }
          if (abs_coeff[idx] >= 2) first_coeff2 = 0;
        }
      }
    }
  }
  return bits;
}

/* ---- cbf bit helpers (cu.h:510-569: 5 depth bits per plane, is_set tests levels >= depth) ---- */
static const uint16_t cbf_masks[5] = { 0x1f, 0x0f, 0x07, 0x03, 0x1 };
static int cbf_is_set(uint16_t cbf, int depth, int plane) { return (cbf & (cbf_masks[depth] << (5 * plane))) != 0; }
static int cbf_is_set_any(uint16_t cbf, int depth) { return cbf_is_set(cbf, depth, 0) || cbf_is_set(cbf, depth, 1) || cbf_is_set(cbf, depth, 2); }
static void cbf_set(uint16_t *cbf, int depth, int plane) { *cbf |= (0x10 >> depth) << (5 * plane); }
static void cbf_clear(uint16_t *cbf, int depth, int plane) { *cbf &= ~(cbf_masks[depth] << (5 * plane)); }

static unsigned zorder(int x, int y) /* cu.h:385-421 with width 64: Morton index of the 4x4 block times 16 */
{
  unsigned r = 0;
  for (int b = 0; b < 4; b++) r |= (((x >> (2 + b)) & 1) << (2 * b)) | (((y >> (2 + b)) & 1) << (2 * b + 1));
  return r * 16;
}

/* intra.c:47-82 num_ref_pixels_{top,left}: how many reference pixels to the right / below are already coded for
 * the 4x4 unit at (c, r) -- a pure function of z-order (tools/generate_ref_pixel_tables.py), regenerated here. */
static int zidx(int c, int r) { int v = 0; for (int b = 0; b < 4; b++) v |= ((c >> b) & 1) << (2 * b) | ((r >> b) & 1) << (2 * b + 1); return v; }
static void build_avail_tables(ctu_t *t)
{
  for (int r = 0; r < 16; r++)
    for (int c = 0; c < 16; c++) {
      int n = 0;
      if (r == 0) t->tbl_top[r][c] = 64;
      else { for (int cc = c; cc < 16 && zidx(cc, r - 1) < zidx(c, r); cc++) n++; t->tbl_top[r][c] = (uint8_t)(4 * n); }
      n = 0;
      if (c == 0) t->tbl_left[r][c] = (uint8_t)(64 - 4 * r);
      else { for (int rr = r; rr < 16 && zidx(c - 1, rr) < zidx(c, r); rr++) n++; t->tbl_left[r][c] = (uint8_t)(4 * n); }
    }
}

static cu_t *cu_at(level_t *lv, int xl, int yl) { return &lv->cu[(yl >> 2) * 16 + (xl >> 2)]; }

/* CU info of a neighbour at luma frame position (fx, fy): inside the current CTU from the work-tree level, otherwise
 * from the frame arrays (init_lcu_t copies those into the lcu_t border, search.c:1088-1120).  NULL = not available. */
static int neighbour_cu(ctu_t *t, level_t *lv, int fx, int fy, cu_t *out)
{
  if (fx < 0 || fy < 0 || fx >= t->W || fy >= t->H) return 0;
  if (fx >= t->cx && fx < t->cx + LCU && fy >= t->cy && fy < t->cy + LCU) { *out = *cu_at(lv, fx - t->cx, fy - t->cy); return 1; }
  if (t->fcu) { *out = t->fcu[(fy >> 2) * (t->W >> 2) + (fx >> 2)]; return 1; }
  const int i = (fy >> 3) * (t->W >> 3) + (fx >> 3);
  memset(out, 0, sizeof *out);
  out->type = 1; out->depth = t->fdepth[i]; out->mode = t->fmode4[(fy >> 2) * (t->W >> 2) + (fx >> 2)]; out->tr_depth = out->depth; out->cbf = 0;
  return 1;
}

/* reconstructed pixel of plane c at plane coordinates (px, py): current CTU -> level buffer, else frame */
static uint8_t rec_px(ctu_t *t, level_t *lv, int c, int px, int py)
{
  const int sh = c ? 1 : 0, w = LCU >> sh, ox = t->cx >> sh, oy = t->cy >> sh;
  if (px >= ox && px < ox + w && py >= oy && py < oy + w) return lv->rec[c][(py - oy) * w + (px - ox)];
  return t->frec[c][py * (t->W >> sh) + px];
}

/* intra.c:305-425 kvz_intra_build_reference_any (the _inner variant :427-543 is the same function away from the
 * picture edge when the picture size is a multiple of 8).  refs = [2w+1], index 0 = top-left corner. */
static void build_reference(ctu_t *t, level_t *lv, int log2w, int c, int lx, int ly /* luma frame coords */, uint8_t *top, uint8_t *left)
{
  const int sh = c ? 1 : 0, w = 1 << log2w;
  const int px = lx >> sh, py = ly >> sh; /* plane coords */
  const int llx = lx & 63, lly = ly & 63;
  if (lx > 0) {
    int avail = t->tbl_left[lly / 4][llx / 4] >> sh;
    if (avail > 2 * w) avail = 2 * w;
    if (avail > ((t->H - ly) >> sh)) avail = (t->H - ly) >> sh;
    for (int i = 0; i < avail; i++) left[i + 1] = rec_px(t, lv, c, px - 1, py + i);
    const uint8_t nearest = left[avail];
    for (int i = avail; i < 2 * w; i++) left[i + 1] = nearest;
  } else {
    const uint8_t nearest = ly > 0 ? rec_px(t, lv, c, px, py - 1) : 128;
    for (int i = 0; i < 2 * w; i++) left[i + 1] = nearest;
  }
  if (lx > 0 && ly > 0) left[0] = top[0] = rec_px(t, lv, c, px - 1, py - 1);
  else left[0] = top[0] = left[1];
  if (ly > 0) {
    int avail = t->tbl_top[lly / 4][llx / 4] >> sh;
    if (avail > 2 * w) avail = 2 * w;
    if (avail > ((t->W - lx) >> sh)) avail = (t->W - lx) >> sh;
    for (int i = 0; i < avail; i++) top[i + 1] = rec_px(t, lv, c, px + i, py - 1);
    const uint8_t nearest = rec_px(t, lv, c, px + avail - 1, py - 1);
    for (int i = avail; i < 2 * w; i++) top[i + 1] = nearest;
  } else {
    const uint8_t nearest = lx > 0 ? rec_px(t, lv, c, px - 1, py) : 128;
    for (int i = 0; i < 2 * w; i++) top[i + 1] = nearest;
  }
}

/* intra.c:176-204 intra_filter_reference ([1 2 1] smoothing) */
static void filter_reference(int log2w, const uint8_t *top, const uint8_t *left, uint8_t *ftop, uint8_t *fleft)
{
  const int n = 2 * (1 << log2w) + 1;
  fleft[0] = ftop[0] = (uint8_t)((left[1] + 2 * left[0] + top[1] + 2) / 4);
  for (int i = 1; i < n - 1; i++) {
    fleft[i] = (uint8_t)((left[i - 1] + 2 * left[i] + left[i + 1] + 2) / 4);
    ftop[i] = (uint8_t)((top[i - 1] + 2 * top[i] + top[i + 1] + 2) / 4);
  }
  fleft[n - 1] = left[n - 1];
  ftop[n - 1] = top[n - 1];
}

/* intra.c:252-301 kvz_intra_predict (filter_boundary = luma; lossless off) */
static void intra_predict(int log2w, int mode, int c, const uint8_t *top, const uint8_t *left, uint8_t *dst)
{
  const int w = 1 << log2w;
  uint8_t ftop[2 * 32 + 1], fleft[2 * 32 + 1];
  const uint8_t *ut = top, *ul = left;
  int use_filtered = 0;
  if (c != 0 || mode == 1 || w == 4) use_filtered = 0;
  else if (mode == 0) use_filtered = 1;
  else {
    static const int thres[5] = { 0, 7, 1, 0, 0 };
    const int d26 = abs(mode - 26), d10 = abs(mode - 10);
    if ((d26 < d10 ? d26 : d10) > thres[log2w - 2]) use_filtered = 1;
  }
  if (use_filtered) { filter_reference(log2w, top, left, ftop, fleft); ut = ftop; ul = fleft; }
  if (mode == 0) kvz_oracle_intra_pred_planar(log2w, ut, ul, dst);
  else if (mode == 1) {
    if (c == 0 && w < 32) kvz_oracle_intra_pred_filtered_dc(log2w, ut, ul, dst);
    else { /* intra.c:229-249 intra_pred_dc */
      int sum = 0;
      for (int i = 0; i < w; i++) sum += ut[i + 1] + ul[i + 1];
      memset(dst, (uint8_t)((sum + w) >> (log2w + 1)), w * w);
    }
  } else {
    kvz_oracle_angular_pred(log2w, mode, ut, ul, dst);
    if (c == 0 && w < 32 && (mode == 10 || mode == 26)) { /* intra.c:207-219 intra_post_process_angular */
      const uint8_t *ref = mode == 10 ? ut : ul;
      const int stride = mode == 10 ? 1 : w;
      for (int i = 0; i < w; i++) {
        int v = dst[i * stride] + ((ref[i + 1] - ref[0]) >> 1);
        dst[i * stride] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
      }
    }
  }
}

/* intra.c:84-126 kvz_intra_get_dir_luma_predictor */
static void mpm_candidates(int y, const cu_t *left, const cu_t *above, int8_t preds[3])
{
  int l = 1, a = 1;
  if (left && left->type == 1) l = left->mode;
  if (above && above->type == 1 && y % LCU != 0) a = above->mode;
  if (l == a) {
    if (l > 1) { preds[0] = (int8_t)l; preds[1] = (int8_t)(((l + 29) % 32) + 2); preds[2] = (int8_t)(((l - 1) % 32) + 2); }
    else { preds[0] = 0; preds[1] = 1; preds[2] = 26; }
  } else {
    preds[0] = (int8_t)l; preds[1] = (int8_t)a;
    if (l && a) preds[2] = 0; else preds[2] = (l + a) < 2 ? 26 : 1;
  }
}

/* search_intra.c:641-676 kvz_luma_mode_bits (only_count path) */
static double luma_mode_bits(ctu_t *t, int mode, const int8_t preds[3], int update)
{
  double bits = 0;
  int in = 0;
  for (int i = 0; i < 3; i++) if (mode == preds[i]) in = 1;
  bits += ctx_price(t, CX_INTRA, in, update, t->m->intra_mode[in]);
  if (in) bits += (mode == preds[0]) ? 1 : 2; else bits += 5;
  return bits;
}

/* search_intra.c:391-530 search_intra_rough (trskip off).  Returns the number of modes tried. */
static int rough_search(ctu_t *t, int log2w, const uint8_t *orig /* contiguous */, const uint8_t *top, const uint8_t *left,
                        const int8_t preds[3], int8_t modes[35], double costs[35])
{
  const int w = 1 << log2w;
  static const int8_t offsets[4] = { 2, 4, 8, 8 };
  uint8_t pr[2 * 1024];
  unsigned sc[2];
  int n = 0, offset = offsets[log2w - 2];
  int32_t min_cost = 0x7fffffff, max_cost = -0x7fffffff - 1;
  for (int mode = 2; mode <= 34; mode += 2 * offset) {
    for (int i = 0; i < 2; i++) if (mode + i * offset <= 34) intra_predict(log2w, mode + i * offset, 0, top, left, pr + 1024 * i);
    kvz_oracle_satd_nxn_dual(w, pr, orig, 2, sc);
    for (int i = 0; i < 2; i++) if (mode + i * offset <= 34) {
      costs[n] = (double)sc[i]; modes[n] = (int8_t)(mode + i * offset);
      if (costs[n] < min_cost) min_cost = (int32_t)costs[n];
      if (costs[n] > max_cost) max_cost = (int32_t)costs[n];
      n++;
    }
  }
  int bi = 0;
  for (int i = 1; i < n; i++) if (costs[i] < costs[bi]) bi = i;
  int8_t best_mode = modes[bi];
  double best_cost = min_cost;
  if (min_cost != max_cost) {
    while (offset > 1) {
      offset >>= 1;
      const int8_t tm[2] = { (int8_t)(best_mode - offset), (int8_t)(best_mode + offset) };
      int in_range = 0;
      for (int i = 0; i < 2; i++) in_range |= (tm[i] >= 2 && tm[i] <= 34);
      if (!in_range) continue;
      for (int i = 0; i < 2; i++) if (tm[i] >= 2 && tm[i] <= 34) intra_predict(log2w, tm[i], 0, top, left, pr + 1024 * i);
      kvz_oracle_satd_nxn_dual(w, pr, orig, 2, sc);
      for (int i = 0; i < 2; i++) if (tm[i] >= 2 && tm[i] <= 34) {
        costs[n] = (double)sc[i]; modes[n] = tm[i];
        if (costs[n] < best_cost) { best_cost = costs[n]; best_mode = modes[n]; }
        n++;
      }
    }
  }
  const int8_t add_modes[5] = { preds[0], preds[1], preds[2], 0, 1 };
  for (int p = 0; p < 5; p++) {
    int has = 0;
    for (int i = 0; i < n; i++) if (modes[i] == add_modes[p]) { has = 1; break; }
    if (!has) {
      intra_predict(log2w, add_modes[p], 0, top, left, pr);
      costs[n] = (double)kvz_oracle_satd_nxn(w, pr, orig);
      modes[n] = add_modes[p];
      n++;
    }
  }
  for (int i = 0; i < n; i++) costs[i] += t->m->lambda_sqrt * luma_mode_bits(t, modes[i], preds, 0);  /* search_cabac.update == 0 here */
  return n;
}

/* search_intra.c:812-900 kvz_search_cu_intra (rd=0: rough search only) */
static int search_cu_intra_cost(ctu_t *t, level_t *lv, int x, int y, int depth, double *cost_out)
{
  const int log2w = 6 - depth, w = 1 << log2w, xl = x - t->cx, yl = y - t->cy;
  cu_t lc, ac, *left = NULL, *above = NULL;
  if (x >= 4 && neighbour_cu(t, lv, x - 1, y, &lc)) left = &lc;
  if (y >= 4 && yl > 0 && neighbour_cu(t, lv, x, y - 1, &ac)) above = &ac;
  int8_t preds[3], modes[35];
  double costs[35];
  mpm_candidates(y, left, above, preds);
  uint8_t top[2 * 32 + 1], lft[2 * 32 + 1], orig[32 * 32];
  build_reference(t, lv, log2w, 0, x, y, top, lft);
  for (int r = 0; r < w; r++) memcpy(orig + r * w, &t->org[0][(yl + r) * LCU + xl], w);
  const int n = rough_search(t, log2w, orig, top, lft, preds, modes, costs);
  int bi = 0;
  for (int i = 1; i < n; i++) if (costs[i] < costs[bi]) bi = i;
  if (cost_out) *cost_out = costs[bi];
  return modes[bi];
}
static int search_cu_intra(ctu_t *t, level_t *lv, int x, int y, int depth) { return search_cu_intra_cost(t, lv, x, y, depth, NULL); }

/* kvz_get_scan_order for an intra block (encoderstate.c:1761-1775); the chroma mode is the luma mode here */
static int tu_scan_order(int mode, int depth)
{
  if (depth >= 3) {
    if (mode >= 6 && mode <= 14) return 2;
    if (mode >= 22 && mode <= 30) return 1;
  }
  return 0;
}

/* kvz_quantize_residual's rdoq leg (quant-generic.c:198-292 with cfg.rdoq_enable, rdoq_skip 0): transform, kvz_rdoq on state->cabac's contexts, then as usual */
static int quantize_residual_rdoq(const ctu_t *t, int width, int color, int scan_order, int tr_depth, int stride, const uint8_t *ref_in, uint8_t *rec, int16_t *coeff_out)
{
  int16_t residual[32 * 32], coeff[32 * 32];
  kvz_hip_quant_params p;
  memset(&p, 0, sizeof p);
  p.qp = t->m->qp; p.bitdepth = 8; p.slice_is_intra = !t->slice_b; p.cu_is_intra = 1;
  for (int y = 0; y < width; y++)
    for (int x = 0; x < width; x++) residual[x + y * width] = (int16_t)(ref_in[x + y * stride] - rec[x + y * stride]);
  const int idx = width == 4 ? (color == 0 ? 4 : 0) : width == 8 ? 1 : width == 16 ? 2 : 3;
  kvz_oracle_transform(idx, 8, residual, coeff);
  kvz_oracle_rdoq(t->m->qp, t->m->lambda, t->coder.s, t->m->entropy_fbits, coeff, coeff_out, width, color == 0 ? 0 : 2, scan_order, tr_depth);
  int has_coeffs = 0;
  for (int i = 0; i < width * width; i++) if (coeff_out[i] != 0) { has_coeffs = 1; break; }
  if (has_coeffs) {
    kvz_oracle_dequant(&p, coeff_out, coeff, width, width, color == 0 ? 0 : (color == 1 ? 2 : 3), 1);
    kvz_oracle_transform(KVZ_HIP_IDCT_4 + idx, 8, coeff, residual);
    for (int y = 0; y < width; y++)
      for (int x = 0; x < width; x++) rec[x + y * stride] = (uint8_t)ORC_CLIP(0, 255, (int16_t)(residual[x + y * width] + rec[x + y * stride]));
  }
  return has_coeffs;
}

/* One leaf TU: intra_recon_tb_leaf (intra.c:561-608) + quantize_tr_residual (transform.c:294-412).  Returns has_coeffs.
 * depth: the transform unit's depth; tr_rel: cu->tr_depth - cu->depth (1 for the 32x32 units of a 64x64 CU) */
static int recon_tu(ctu_t *t, level_t *lv, int c, int x, int y /* luma frame coords */, int log2w, int mode, int depth, int tr_rel)
{
  const int sh = c ? 1 : 0, w = 1 << log2w, lw = LCU >> sh;
  const int xl = (x - t->cx) >> sh, yl = (y - t->cy) >> sh;
  uint8_t top[2 * 32 + 1], lft[2 * 32 + 1], pred[32 * 32];
  build_reference(t, lv, log2w, c, x, y, top, lft);
  intra_predict(log2w, mode, c, top, lft, pred);
  uint8_t *rec = &lv->rec[c][yl * lw + xl];
  for (int r = 0; r < w; r++) memcpy(rec + r * lw, pred + r * w, w);
  int16_t *coeff = &lv->coeff[c][zorder(xl, yl)];
  if (t->m->rdoq) return quantize_residual_rdoq(t, w, c, tu_scan_order(mode, depth), tr_rel, lw, &t->org[c][yl * lw + xl], rec, coeff);
  kvz_hip_quant_params p;
  memset(&p, 0, sizeof p);
  p.qp = t->m->qp; p.bitdepth = 8; p.slice_is_intra = !t->slice_b; p.cu_is_intra = 1;
  return kvz_oracle_quantize_residual(&p, w, c, 0, 0, lw, lw, &t->org[c][yl * lw + xl], rec, rec, coeff, 0);
}

/* intra.c:623-717 kvz_intra_recon_cu: depth 0 splits into four 32x32 transform units (tr_depth = 1), every other
 * depth is one luma TU (+ chroma TUs of half the size, 4x4 for 8x8 CUs; transform.c:322-328). */
static void recon_cu_rel(ctu_t *t, level_t *lv, int x, int y, int depth, int mode, int do_luma, int do_chroma, int tr_rel)
{
  const int w = LCU >> depth;
  cu_t *cu = cu_at(lv, x - t->cx, y - t->cy);
  if (do_luma) cbf_clear(&cu->cbf, depth, 0);
  if (do_chroma) { cbf_clear(&cu->cbf, depth, 1); cbf_clear(&cu->cbf, depth, 2); }
  if (depth == 0) {
    const int o = w / 2;
    recon_cu_rel(t, lv, x, y, 1, mode, do_luma, do_chroma, 1);  /* cu->tr_depth - cu->depth = 1 for these units */
    recon_cu_rel(t, lv, x + o, y, 1, mode, do_luma, do_chroma, 1);
    recon_cu_rel(t, lv, x, y + o, 1, mode, do_luma, do_chroma, 1);
    recon_cu_rel(t, lv, x + o, y + o, 1, mode, do_luma, do_chroma, 1);
    const uint16_t ch[3] = { cu_at(lv, x - t->cx + o, y - t->cy)->cbf, cu_at(lv, x - t->cx, y - t->cy + o)->cbf, cu_at(lv, x - t->cx + o, y - t->cy + o)->cbf };
    for (int c = 0; c < 3; c++) {
      if ((c == 0 && !do_luma) || (c > 0 && !do_chroma)) continue;
      if (cbf_is_set(ch[0], 1, c) || cbf_is_set(ch[1], 1, c) || cbf_is_set(ch[2], 1, c)) cbf_set(&cu->cbf, 0, c); /* cbf_set_conditionally */
    }
    return;
  }
  const int log2w = 6 - depth;
  if (do_luma) {
    cbf_clear(&cu->cbf, depth, 0);
    if (recon_tu(t, lv, 0, x, y, log2w, mode, depth, tr_rel)) cbf_set(&cu->cbf, depth, 0);
  }
  if (do_chroma && x % 8 == 0 && y % 8 == 0) {
    const int cl2 = depth >= 3 ? 2 : log2w - 1; /* transform.c:326-327; depth 4 (NxN): the 4x4 chroma blocks of the 8x8 CU, done with its first PU (transform.c:306-312) */
    for (int c = 1; c <= 2; c++) {
      cbf_clear(&cu->cbf, depth, c);
      if (recon_tu(t, lv, c, x, y, cl2, mode, depth, tr_rel)) cbf_set(&cu->cbf, depth, c);
    }
  }
}
/* tr_rel: what kvz_rdoq gets as tr_depth = cu->tr_depth - cu->depth, plus one for an NxN CU (quant-generic.c:237-238): 2 for the blocks of its four PUs
 * (depth 4: cu->depth stays 3, search.c:691) */
static void recon_cu(ctu_t *t, level_t *lv, int x, int y, int depth, int mode, int do_luma, int do_chroma) { recon_cu_rel(t, lv, x, y, depth, mode, do_luma, do_chroma, depth == 4 ? 2 : 0); }

/* Test hook: the residual coder's bit count on caller-supplied states of the residual contexts (KVZ_HIP_CX_SIG_CG .. KVZ_HIP_CX_CBF_CHROMA_DEEP - 1,
 * updated in place when `update`), against the reference's kvz_encode_coeff_nxn (tests/test_oracle_vs_ref.py) */
double kvz_oracle_coeff_cabac_bits(const float entropy_fbits[128], const int16_t *coeff, int width, int type, int scan_mode, int update, uint8_t *ctx)
{
  kvz_hip_intra_cost_model m;
  ctxs_t c;
  build_transitions();
  memset(&m, 0, sizeof m);
  memset(&c, 0, sizeof c);
  memcpy(m.entropy_fbits, entropy_fbits, sizeof m.entropy_fbits);
  memcpy(&c.s[CX_SIG_CG], ctx, KVZ_HIP_CX_CBF_CHROMA_DEEP - CX_SIG_CG);
  const double bits = encode_coeff_nxn(&m, &c, update, coeff, width, type, scan_mode);
  memcpy(ctx, &c.s[CX_SIG_CG], KVZ_HIP_CX_CBF_CHROMA_DEEP - CX_SIG_CG);
  return bits;
}

/* encoderstate.c:1761-1775 kvz_get_scan_order for an intra CU (the chroma mode is the luma mode here) */
static int scan_order(int intra_mode, int depth)
{
  if (depth >= 3) {
    if (intra_mode >= 6 && intra_mode <= 14) return 2;   /* SCAN_VER */
    if (intra_mode >= 22 && intra_mode <= 30) return 1;  /* SCAN_HOR */
  }
  return 0;
}
/* rdo.c:311-340 kvz_get_coeff_cost: the fast estimate below fast_residual_cost_limit, else get_coeff_cabac_cost (rdo.c:220-263): the
 * residual coder runs in counting mode on a copy of the search contexts -- `update` flag included, so the states move (and
 * are copied back) only while the search has updates switched on; otherwise every bin is priced at the entry state */
static double coeff_cost(ctu_t *t, const int16_t *coeff, int width, int type, int scan_mode, int update)
{
  if (!t->m->coeff_cabac) return kvz_oracle_fast_coeff_cost(coeff, width, t->m->coeff_weights);
  int found = 0;
  for (int i = 0; i < width * width; i++) if (coeff[i]) { found = 1; break; }
  if (!found) return 0;
  return encode_coeff_nxn(t->m, &t->cab, update && t->m->adaptive, coeff, width, type, scan_mode);
}

/* search.c:425-541 cu_rd_cost_tr_split_accurate (intra CU, fast coefficient cost rdo.c:311-326) */
static double rd_cost(ctu_t *t, level_t *lv, int xl, int yl, int depth, const cu_t *pred_cu, int update)
{
  const kvz_hip_intra_cost_model *m = t->m;
  const int width = LCU >> depth;
  const cu_t *tr_cu = cu_at(lv, xl, yl);
  double coeff_bits = 0, tr_tree_bits = 0;
  const int tr_depth = tr_cu->tr_depth - depth;
  const int cb_u = cbf_is_set(tr_cu->cbf, depth, 1), cb_v = cbf_is_set(tr_cu->cbf, depth, 2);
  /* transform_tree split flag: never coded with tr_depth_intra = 0 (search.c:451-464) */
  if (tr_cu->depth == depth || cbf_is_set(tr_cu->cbf, depth - 1, 1)) tr_tree_bits += ctx_price(t, CX_CBF_CHROMA + depth - tr_cu->depth, cb_u, update, m->cbf_chroma[depth - tr_cu->depth][cb_u]);
  if (tr_cu->depth == depth || cbf_is_set(tr_cu->cbf, depth - 1, 2)) tr_tree_bits += ctx_price(t, CX_CBF_CHROMA + depth - tr_cu->depth, cb_v, update, m->cbf_chroma[depth - tr_cu->depth][cb_v]);
  if (tr_depth > 0) {
    const int o = LCU >> (depth + 1);
    double sum = 0;
    sum += rd_cost(t, lv, xl, yl, depth + 1, pred_cu, update);
    sum += rd_cost(t, lv, xl + o, yl, depth + 1, pred_cu, update);
    sum += rd_cost(t, lv, xl, yl + o, depth + 1, pred_cu, update);
    sum += rd_cost(t, lv, xl + o, yl + o, depth + 1, pred_cu, update);
    return sum + tr_tree_bits * m->lambda;
  }
  const int cb_y = cbf_is_set(tr_cu->cbf, depth, 0);
  const int is_tr_split = depth - tr_cu->depth;
  tr_tree_bits += ctx_price(t, CX_CBF_LUMA + !is_tr_split, cb_y, update, m->cbf_luma[!is_tr_split][cb_y]);
  unsigned luma_ssd = kvz_oracle_pixels_calc_ssd(&t->org[0][yl * LCU + xl], &lv->rec[0][yl * LCU + xl], LCU, LCU, width);
  if (cb_y) coeff_bits += coeff_cost(t, &lv->coeff[0][zorder(xl, yl)], width, 0, scan_order(pred_cu->mode, depth), update);
  unsigned chroma_ssd = 0;
  if (xl % 8 == 0 && yl % 8 == 0) {
    const int cw = depth <= 3 ? LCU >> (depth + 1) : LCU >> depth, i = (yl / 2) * 32 + xl / 2;
    chroma_ssd = kvz_oracle_pixels_calc_ssd(&t->org[1][i], &lv->rec[1][i], 32, 32, cw) + kvz_oracle_pixels_calc_ssd(&t->org[2][i], &lv->rec[2][i], 32, 32, cw);
    const unsigned zi = zorder(xl / 2, yl / 2);
    if (cb_u) coeff_bits += coeff_cost(t, &lv->coeff[1][zi], cw, 2, scan_order(pred_cu->mode, depth), update);
    if (cb_v) coeff_bits += coeff_cost(t, &lv->coeff[2][zi], cw, 2, scan_order(pred_cu->mode, depth), update);
  }
  const double bits = tr_tree_bits + coeff_bits;
  return luma_ssd * 0.8 + chroma_ssd * 1.5 + bits * m->lambda; /* KVZ_LUMA_MULT / KVZ_CHROMA_MULT search.h:50-56 */
}

/* search.c:629-635 get_ctx_cu_split_model == the split_model of encode_coding_tree.c:985-994 */
static int split_model(ctu_t *t, level_t *lv, int x, int y, int depth)
{
  cu_t n;
  int model = 0;
  if (x > 0 && neighbour_cu(t, lv, x - 1, y, &n) && n.depth > depth) model++;
  if (y > 0 && neighbour_cu(t, lv, x, y - 1, &n) && n.depth > depth) model++;
  return model;
}

/* intra-mode part of the CU syntax: prev_intra_luma_pred_flag + mpm_idx / rem_intra_luma_pred_mode + chroma mode
 * (encode_coding_tree.c:467-652; chroma mode == luma mode -> one context bin "0") */
static double intra_mode_syntax_bits(ctu_t *t, level_t *lv, int x, int y, int mode, int with_chroma, int update, int mock)
{
  cu_t lc, ac, *left = NULL, *above = NULL;
  /* The mock encode looks its left neighbour up at LCU-local column SUB_SCU(x - 1) (encode_coding_tree.c:516), which for a CU on
   * the LCU's left edge is column 63 of the work tree -- a cell the z-order search has not reached yet (type CU_NOTSET), so the left
   * candidate falls back to DC there.  calc_mode_bits (search.c:566) and the real encode (lcu == NULL) use the true neighbour. */
  if (x > 0 && !(mock && x % LCU == 0) && neighbour_cu(t, lv, x - 1, y, &lc)) left = &lc;
  if (y % LCU > 0 && y > 0 && neighbour_cu(t, lv, x, y - 1, &ac)) above = &ac;
  int8_t preds[3];
  mpm_candidates(y, left, above, preds);
  double bits = luma_mode_bits(t, mode, preds, update);
  if (with_chroma) bits += ctx_price(t, CX_CHROMA, 0, update, t->m->chroma_mode[0]);
  return bits;
}

/* encode_coding_tree.c:948-1049 kvz_mock_encode_coding_unit for an intra 2Nx2N CU in an I slice */
static double cu_bits(ctu_t *t, level_t *lv, int x, int y, int depth, int mode, int update)
{
  double bits = 0;
  const int w = LCU >> depth;
  if (depth != 3 && !(t->W < x + w || t->H < y + w)) { const int sm = split_model(t, lv, x, y, depth); bits += ctx_price(t, CX_SPLIT + sm, 0, update, t->m->split_flag[sm][0]); }
  if (depth == 3) bits += ctx_price(t, CX_PART, 1, update, t->m->part_size[1]);
  /* encode_intra_coding_unit adds the flag first and the bypass bins after it; same sum order as luma_mode_bits */
  bits += intra_mode_syntax_bits(t, lv, x, y, mode, 1, update, 1);
  return bits;
}

static void fill_cu(level_t *lv, int xl, int yl, int w, const cu_t *src) /* search.c:137-159 lcu_fill_cu_info (+ tr_depth) */
{
  for (int yy = yl; yy < yl + w; yy += 4)
    for (int xx = xl; xx < xl + w; xx += 4) { cu_t *c = cu_at(lv, xx, yy); c->type = src->type; c->depth = src->depth; c->mode = src->mode; c->tr_depth = src->tr_depth; }
}

static void copy_region(ctu_t *t, level_t *from, level_t *to, int xl, int yl, int w, int coeffs) /* search.c:55-100 copy_cu_{info,pixels,coeffs} */
{
  (void)t;
  for (int yy = yl; yy < yl + w; yy += 4) for (int xx = xl; xx < xl + w; xx += 4) *cu_at(to, xx, yy) = *cu_at(from, xx, yy);
  for (int r = 0; r < w; r++) memcpy(&to->rec[0][(yl + r) * LCU + xl], &from->rec[0][(yl + r) * LCU + xl], w);
  for (int c = 1; c <= 2; c++) for (int r = 0; r < w / 2; r++) memcpy(&to->rec[c][(yl / 2 + r) * 32 + xl / 2], &from->rec[c][(yl / 2 + r) * 32 + xl / 2], w / 2);
  if (coeffs) {
    memcpy(&to->coeff[0][zorder(xl, yl)], &from->coeff[0][zorder(xl, yl)], w * w * sizeof(int16_t));
    for (int c = 1; c <= 2; c++) memcpy(&to->coeff[c][zorder(xl / 2, yl / 2)], &from->coeff[c][zorder(xl / 2, yl / 2)], (w / 2) * (w / 2) * sizeof(int16_t));
  }
}

/* search.c:646-1063 search_cu, I slice, pu_depth_intra = [2,3] */
static double search_cu(ctu_t *t, int x, int y, int depth)
{
  const kvz_hip_intra_cost_model *m = t->m;
  const int w = LCU >> depth, xl = x - t->cx, yl = y - t->cy;
  level_t *lv = &t->lv[depth];
  double cost = MAX_COST;
  const ctxs_t pre_search = t->cab;  /* search.c:655-656 */
  if (x >= t->W || y >= t->H) return 0;
  cu_t *cur = cu_at(lv, xl, yl);
  cur->depth = (uint8_t)(depth > 3 ? 3 : depth);
  cur->tr_depth = (uint8_t)(depth > 0 ? depth : 1);
  cur->type = 0;
  const int inside = x + w <= t->W && y + w <= t->H;
  const int max_depth = m->search_nxn ? 4 : 3;  /* pu_depth_intra.max: 4 = the NxN partition of an 8x8 CU, one more level of this recursion (search.c:691, 794) */
  if (inside && depth >= (m->search_32x32 ? 1 : 2) && depth <= max_depth) {  /* pu_depth_intra.min .. max (search.c:794) */
    const int mode = search_cu_intra(t, lv, x, y, depth);
    cur->type = 1; cur->mode = (uint8_t)mode;
    fill_cu(lv, xl, yl, w, cur);
    recon_cu(t, lv, x, y, depth, mode, 1, 0);
    if (x % 8 == 0 && y % 8 == 0) recon_cu(t, lv, x, y, depth, mode, 0, 1);
  }
  if (cur->type == 1) {
    /* search.c:895-940: cabac->update = 1 around the mock encode (a PU of an NxN CU: calc_mode_bits instead, search.c:906-913) ... */
    const double bits = depth == 4 ? intra_mode_syntax_bits(t, lv, x, y, cur->mode, x % 8 == 0 && y % 8 == 0, 1, 0) : cu_bits(t, lv, x, y, depth, cur->mode, 1);
    cost = bits * m->lambda;
    cost += rd_cost(t, lv, xl, yl, depth, cur, 1);                  /* ... and the transform-tree flags */
  }
  const int can_split = cur->type == 0 || depth < max_depth;
  if (can_split) {
    const int half = w / 2;
    double split_cost = 0.0;
    const int cbf = cbf_is_set_any(cur->cbf, depth);
    ctxs_t post_search = t->cab;  /* search.c:956-959: the split alternative starts again from the state at entry */
    t->cab = pre_search;
    double split_bits = 0;
    if (depth < 3) { const int sm = split_model(t, lv, x, y, depth); split_bits += ctx_price(t, CX_SPLIT + sm, 1, 1, m->split_flag[sm][1]); }
    if (cur->type == 1 && depth == 3) split_bits += ctx_price(t, CX_PART, 0, 1, m->part_size[0]);  /* search.c:970-974: part_size NxN */
    split_cost += split_bits * m->lambda;
    if (cur->type == 0 || cbf) {
      if (split_cost < cost) split_cost += search_cu(t, x, y, depth + 1);
      if (split_cost < cost) split_cost += search_cu(t, x + half, y, depth + 1);
      if (split_cost < cost) split_cost += search_cu(t, x, y + half, depth + 1);
      if (split_cost < cost) split_cost += search_cu(t, x + half, y + half, depth + 1);
    } else {
      split_cost = 2147483647; /* INT_MAX */
    }
    /* search.c:996-1044 combine_intra_cus: try the top-left child's mode for the whole CU */
    if (cur->type == 0 && depth < 4 && inside) {
      const cu_t *d1 = cu_at(&t->lv[depth + 1], xl, yl);
      if (d1->type == 1 && d1->depth == depth + 1) {
        /* search.c:1005-1041: priced from the state at entry; pre_search_cabac carries update == 0 (it was copied while the
         * caller had it switched off), so nothing here changes a context */
        const ctxs_t temp = t->cab;
        t->cab = pre_search;
        cost = 0;
        double bits = 0;
        if (depth < 3) { const int sm = split_model(t, lv, x, y, depth); bits += ctx_price(t, CX_SPLIT + sm, 0, 0, m->split_flag[sm][0]); }
        cur->mode = d1->mode; cur->type = 1;
        fill_cu(lv, xl, yl, w, cur);
        recon_cu(t, lv, x, y, depth, cur->mode, 1, 1);
        const double mode_bits = intra_mode_syntax_bits(t, lv, x, y, cur->mode, 1, 0, 0) /* calc_mode_bits search.c:557-581 */ + bits;
        cost += mode_bits * m->lambda;
        cost += rd_cost(t, lv, xl, yl, depth, cur, 0);
        post_search = t->cab;
        t->cab = temp;
      }
    }
    if (split_cost < cost) {
      cost = split_cost;
      copy_region(t, &t->lv[depth + 1], lv, xl, yl, w, 1); /* work_tree_copy_up */
    } else if (depth > 0) {
      t->cab = post_search;  /* search.c:1051 */
      for (int i = depth + 1; i < NLEVELS; i++) copy_region(t, lv, &t->lv[i], xl, yl, w, 0); /* work_tree_copy_down */
    }
  } else if (depth >= 0 && depth < 4) {
    for (int i = depth + 1; i < NLEVELS; i++) copy_region(t, lv, &t->lv[i], xl, yl, w, 0);
  }
  return cost;
}

/* kvz_search_lcu (search.c:1209-1249): init the work tree for the CTU at (cx, cy), search, copy the result out. */
static double encode_ctu(ctu_t *t, int cx, int cy, int16_t *coeff_out)
{
  t->cx = cx; t->cy = cy;
  memset(t->lv, 0, sizeof t->lv);
  memset(t->org, 0, sizeof t->org);
  const int xmax = (cx + LCU < t->W ? LCU : t->W - cx), ymax = (cy + LCU < t->H ? LCU : t->H - cy);
  for (int r = 0; r < ymax; r++) memcpy(&t->org[0][r * LCU], &t->src[0][(cy + r) * t->W + cx], xmax);
  for (int c = 1; c <= 2; c++)
    for (int r = 0; r < ymax / 2; r++) memcpy(&t->org[c][r * 32], &t->src[c][(cy / 2 + r) * (t->W / 2) + cx / 2], xmax / 2);
  const double cost = search_cu(t, cx, cy, 0);
  level_t *l0 = &t->lv[0];
  for (int r = 0; r < ymax; r++) memcpy(&t->frec[0][(cy + r) * t->W + cx], &l0->rec[0][r * LCU], xmax);
  for (int c = 1; c <= 2; c++)
    for (int r = 0; r < ymax / 2; r++) memcpy(&t->frec[c][(cy / 2 + r) * (t->W / 2) + cx / 2], &l0->rec[c][r * 32], xmax / 2);
  for (int yy = 0; yy < ymax; yy += 8)
    for (int xx = 0; xx < xmax; xx += 8) {
      const int i = ((cy + yy) >> 3) * (t->W >> 3) + ((cx + xx) >> 3);
      t->fdepth[i] = cu_at(l0, xx, yy)->depth;
      t->fmode[i] = cu_at(l0, xx, yy)->mode;
      t->fnxn[i] = cu_at(l0, xx, yy)->tr_depth == 4;
      for (int j = 0; j < 4; j++) t->fmode4[((cy + yy) / 4 + (j >> 1)) * (t->W >> 2) + (cx + xx) / 4 + (j & 1)] = cu_at(l0, xx + 4 * (j & 1), yy + 4 * (j >> 1))->mode;
    }
  memcpy(coeff_out, l0->coeff[0], 4096 * sizeof(int16_t));
  memcpy(coeff_out + 4096, l0->coeff[1], 1024 * sizeof(int16_t));
  memcpy(coeff_out + 5120, l0->coeff[2], 1024 * sizeof(int16_t));
  return cost;
}

/* The syntax kvazaar writes for a finished CTU (kvz_encode_coding_tree, encode_coding_tree.c:745-940, with
 * encode_intra_coding_unit :467-652 and encode_transform_coeff :193-309), reduced to the bins that touch the ten contexts:
 * this is how state->cabac moves from one CTU to the next.  Reads the frame-level CU arrays (neighbours may lie in other
 * CTUs) and, for the coded block flags, level 0 of the CTU just searched. */
static void code_transform_tree(ctu_t *t, ctxs_t *c, int xl, int yl, int depth, int tr_depth, int parent_u, int parent_v)
{
  const cu_t *cu = cu_at(&t->lv[0], xl, yl), *cu8 = cu_at(&t->lv[0], xl & ~7, yl & ~7);  /* cur_pu / cur_cu of encode_transform_coeff (encode_coding_tree.c:205-210) */
  const int split = cu8->tr_depth > depth;
  const int cb_y = cbf_is_set(cu->cbf, depth, 0), cb_u = cbf_is_set(cu8->cbf, depth, 1), cb_v = cbf_is_set(cu8->cbf, depth, 2);
  /* split_transform_flag is never coded: tr_depth_intra = 0, and the 64x64 split is inferred (encode_coding_tree.c:236-243) */
  if (depth < 4) {
    if (tr_depth == 0 || parent_u) ctx_code(c, CX_CBF_CHROMA + tr_depth, cb_u);
    if (tr_depth == 0 || parent_v) ctx_code(c, CX_CBF_CHROMA + tr_depth, cb_v);
  }
  if (split) {
    const int o = LCU >> (depth + 1);
    code_transform_tree(t, c, xl, yl, depth + 1, tr_depth + 1, cb_u, cb_v);
    code_transform_tree(t, c, xl + o, yl, depth + 1, tr_depth + 1, cb_u, cb_v);
    code_transform_tree(t, c, xl, yl + o, depth + 1, tr_depth + 1, cb_u, cb_v);
    code_transform_tree(t, c, xl + o, yl + o, depth + 1, tr_depth + 1, cb_u, cb_v);
    return;
  }
  ctx_code(c, CX_CBF_LUMA + !tr_depth, cb_y);  /* always present for intra (encode_coding_tree.c:276-279) */
  if (t->m->coeff_cabac) {  /* encode_transform_unit (encode_coding_tree.c:117-190): the residual moves its own contexts */
    const int w = LCU >> depth, cw = depth == 4 ? w : LCU >> (depth + 1);
    if (cb_y) encode_coeff_nxn(t->m, c, 1, &t->lv[0].coeff[0][zorder(xl, yl)], w, 0, scan_order(cu->mode, depth));
    /* 4x4 luma blocks: the 4x4 chroma blocks of the 8x8 CU follow the last of them, under the first PU's mode (encode_coding_tree.c:148-163) */
    if (depth == 4 && (xl % 8 == 0 || yl % 8 == 0)) return;
    const int cscan = scan_order(cu8->mode, depth), cxl = (xl & ~7) / 2, cyl = (yl & ~7) / 2;
    if (cb_u) encode_coeff_nxn(t->m, c, 1, &t->lv[0].coeff[1][zorder(cxl, cyl)], cw, 2, cscan);
    if (cb_v) encode_coeff_nxn(t->m, c, 1, &t->lv[0].coeff[2][zorder(cxl, cyl)], cw, 2, cscan);
  }
}
static void code_coding_tree(ctu_t *t, ctxs_t *c, int x, int y, int depth)
{
  const int w = LCU >> depth, half = w / 2, w8 = t->W >> 3;
  const int cur_depth = t->fdepth[(y >> 3) * w8 + (x >> 3)];
  const int split_flag = cur_depth > depth;  /* GET_SPLITDATA */
  const int border_x = t->W < x + w, border_y = t->H < y + w, border = border_x || border_y;
  const int border_split_x = t->W >= x + 8 + half, border_split_y = t->H >= y + 8 + half;
  if (depth != 3) {
    if (!border) {
      int sm = 0;
      if (x > 0 && t->fdepth[(y >> 3) * w8 + ((x - 1) >> 3)] > depth) sm++;
      if (y > 0 && t->fdepth[((y - 1) >> 3) * w8 + (x >> 3)] > depth) sm++;
      ctx_code(c, CX_SPLIT + sm, split_flag);
    }
    if (split_flag || border) {
      code_coding_tree(t, c, x, y, depth + 1);
      if (!border_x || border_split_x) code_coding_tree(t, c, x + half, y, depth + 1);
      if (!border_y || border_split_y) code_coding_tree(t, c, x, y + half, depth + 1);
      if (!border || (border_split_x && border_split_y)) code_coding_tree(t, c, x + half, y + half, depth + 1);
      return;
    }
  }
  const int nxn = depth == 3 && t->fnxn[(y >> 3) * w8 + (x >> 3)];
  if (depth == 3) ctx_code(c, CX_PART, !nxn);  /* part_mode at the minimum CU size: 2Nx2N = 1, NxN = 0 */
  {
    /* encode_intra_coding_unit (encode_coding_tree.c:467-652): the prev_intra_luma_pred_flags of all PUs first, each against the most probable modes
     * at its own position (4x4-granular CU info); mpm_idx / rem_intra_luma_pred_mode are bypass bins; chroma mode derived from the first PU's */
    const int w4 = t->W >> 2;
    for (int j = 0; j < (nxn ? 4 : 1); j++) {
      const int px = x + 4 * (j & 1), py = y + 4 * (j >> 1);
      cu_t lc = { 1, 0, 0, 0, 0 }, ac = { 1, 0, 0, 0, 0 }, *left = NULL, *above = NULL;
      const int mode = t->fmode4[(py >> 2) * w4 + (px >> 2)];
      if (px > 0) { lc.mode = t->fmode4[(py >> 2) * w4 + ((px - 1) >> 2)]; left = &lc; }
      if (py % LCU > 0 && py > 0) { ac.mode = t->fmode4[((py - 1) >> 2) * w4 + (px >> 2)]; above = &ac; }
      int8_t preds[3];
      mpm_candidates(py, left, above, preds);
      ctx_code(c, CX_INTRA, mode == preds[0] || mode == preds[1] || mode == preds[2]);
    }
    ctx_code(c, CX_CHROMA, 0);
  }
  code_transform_tree(t, c, x - t->cx, y - t->cy, depth, 0, 0, 0);
}

/* One frame, CTUs in raster order.  Planes are tightly packed (stride = width, chroma = width/2); width and height
 * must be multiples of 8 (kvazaar pads its input to that, encoder.c).  Outputs: rec planes, KVZ_HIP_CTU_COEFFS
 * coefficients per CTU (raster CTU order), CU depth and luma mode per 8x8 block (raster, stride width/8), and the
 * RD cost of every CTU. */
/* ... + (model.search_nxn) cu_part: 1 per 8x8 block that is an NxN CU; cu_mode4: luma mode per 4x4 block (stride width/4); cu_mode then holds the first PU's.
 * Both may be NULL. */
void kvz_oracle_intra_frame_nxn(const kvz_hip_intra_cost_model *m, int width, int height, const uint8_t *src_y, const uint8_t *src_u,
                                const uint8_t *src_v, uint8_t *rec_y, uint8_t *rec_u, uint8_t *rec_v, int16_t *coeff, uint8_t *cu_depth,
                                uint8_t *cu_mode, double *ctu_cost, uint8_t *cu_part, uint8_t *cu_mode4)
{
  ctu_t *t = (ctu_t *)calloc(1, sizeof(ctu_t));
  t->m = m; t->W = width; t->H = height;
  t->src[0] = src_y; t->src[1] = src_u; t->src[2] = src_v;
  t->frec[0] = rec_y; t->frec[1] = rec_u; t->frec[2] = rec_v;
  t->fdepth = cu_depth; t->fmode = cu_mode;
  t->fmode4 = (uint8_t *)calloc((size_t)(width / 4) * (height / 4), 1);
  t->fnxn = (uint8_t *)calloc((size_t)(width / 8) * (height / 8), 1);
  build_avail_tables(t);
  build_transitions();
  const int wc = (width + 63) / 64, hc = (height + 63) / 64;
  /* one context set per CTU row, as kvazaar's wavefront rows have one encoder state each: all start at the slice-init state
   * (encoderstate.c:1218), a row's state after its second CTU seeds the row below (encoderstate.c:763-771) */
  ctxs_t *rows = (ctxs_t *)calloc((size_t)hc, sizeof(ctxs_t));
  for (int r = 0; r < hc; r++) for (int i = 0; i < CX_COUNT; i++) rows[r].s[i] = m->ctx_init[i];
  for (int cy = 0; cy < hc; cy++)
    for (int cx = 0; cx < wc; cx++) {
      /* with WPP every LCU row has its own coder; without it (kvazaar switches it off when tiles are used, cfg.c:925-978) one coder runs
       * through the picture in raster order, so a row starts from where the previous one ended */
      ctxs_t *row = m->no_wpp ? &rows[0] : &rows[cy];
      t->cab = *row;  /* kvz_search_lcu: search_cabac = state->cabac (search.c:1211) */
      t->coder = *row;
      ctu_cost[cy * wc + cx] = encode_ctu(t, cx * 64, cy * 64, coeff + (size_t)(cy * wc + cx) * KVZ_HIP_CTU_COEFFS);
      if (m->adaptive) {
        code_coding_tree(t, row, cx * 64, cy * 64, 0);
        if (!m->no_wpp && cx == 1 && cy + 1 < hc) rows[cy + 1] = rows[cy];
      }
    }
  if (cu_part) memcpy(cu_part, t->fnxn, (size_t)(width / 8) * (height / 8));
  if (cu_mode4) memcpy(cu_mode4, t->fmode4, (size_t)(width / 4) * (height / 4));
  free(t->fmode4); free(t->fnxn);
  free(rows);
  free(t);
}
void kvz_oracle_intra_frame(const kvz_hip_intra_cost_model *m, int width, int height, const uint8_t *src_y, const uint8_t *src_u,
                            const uint8_t *src_v, uint8_t *rec_y, uint8_t *rec_u, uint8_t *rec_v, int16_t *coeff, uint8_t *cu_depth,
                            uint8_t *cu_mode, double *ctu_cost)
{
  kvz_oracle_intra_frame_nxn(m, width, height, src_y, src_u, src_v, rec_y, rec_u, rec_v, coeff, cu_depth, cu_mode, ctu_cost, NULL, NULL);
}

/* The cost model for QP `qp` of an I slice (kvz_hip_intra_cost_model): context init values of the
 * HEVC spec as kvazaar tabulates them (context.c:96-134, I-slice row), kvz_ctx_init (context.c:202-213) and the
 * entropy table passed in by the caller (kvz_f_entropy_bits, rdo.c:83) -- tests take it from the reference build. */
void kvz_oracle_intra_cost_model(int qp, const float entropy_fbits[128], uint64_t coeff_weights, kvz_hip_intra_cost_model *m)
{
  static const uint8_t init_split[3] = { 139, 141, 157 }, init_part = 184, init_intra = 184, init_chroma = 63;
  static const uint8_t init_cbf_luma[2] = { 111, 141 }, init_cbf_chroma[2] = { 94, 138 };
  memset(m, 0, sizeof *m);
  m->struct_size = (uint32_t)sizeof *m;
  m->qp = qp;
  m->lambda = 0.57 * pow(2.0, (qp - 12) / 3.0);
  m->lambda_sqrt = sqrt(m->lambda);
  m->coeff_weights = coeff_weights;
  {
    /* I-slice rows of context.c:142-193 (HEVC spec tables 9-4 ff.): coded_sub_block_flag, sig_coeff_flag (27 luma + 15 chroma),
     * last_sig_coeff prefix (15 luma + 15 chroma, of which 3 are used; x and y share the values), greater1 (16 + 8), greater2 (4 + 2) */
    static const uint8_t init_sig_cg[4] = { 91, 171, 134, 141 };
    static const uint8_t init_sig[42] = { 111, 111, 125, 110, 110, 94, 124, 108, 124, 107, 125, 141, 179, 153, 125, 107, 125, 141, 179, 153, 125, 107, 125, 141, 179, 153, 125,
                                          140, 139, 182, 182, 152, 136, 152, 136, 153, 136, 139, 111, 136, 139, 111 };
    static const uint8_t init_last[30] = { 110, 110, 124, 125, 140, 153, 125, 127, 140, 109, 111, 143, 127, 111, 79, 108, 123, 63,
                                           154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154 };  /* CNU = 154 */
    static const uint8_t init_one[24] = { 140, 92, 137, 138, 140, 152, 138, 139, 153, 74, 149, 92, 139, 107, 122, 152, 140, 179, 166, 182, 140, 227, 122, 197 };
    static const uint8_t init_abs[6] = { 138, 153, 136, 167, 152, 152 };
    uint8_t inits[CX_COUNT] = { init_split[0], init_split[1], init_split[2], init_part, init_intra, init_chroma, init_cbf_luma[0], init_cbf_luma[1],
                                init_cbf_chroma[0], init_cbf_chroma[1] };
    for (int i = 0; i < 4; i++) inits[CX_SIG_CG + i] = init_sig_cg[i];
    for (int i = 0; i < 27; i++) inits[CX_SIG_LUMA + i] = init_sig[i];
    for (int i = 0; i < 15; i++) {
      inits[CX_SIG_CHROMA + i] = init_sig[27 + i];
      inits[CX_LAST_Y_LUMA + i] = inits[CX_LAST_X_LUMA + i] = init_last[i];
      inits[CX_LAST_Y_CHROMA + i] = inits[CX_LAST_X_CHROMA + i] = init_last[15 + i];
    }
    for (int i = 0; i < 16; i++) inits[CX_ONE_LUMA + i] = init_one[i];
    for (int i = 0; i < 8; i++) inits[CX_ONE_CHROMA + i] = init_one[16 + i];
    for (int i = 0; i < 4; i++) inits[CX_ABS_LUMA + i] = init_abs[i];
    for (int i = 0; i < 2; i++) inits[CX_ABS_CHROMA + i] = init_abs[4 + i];
    inits[KVZ_HIP_CX_CBF_CHROMA_DEEP] = 182; inits[KVZ_HIP_CX_CBF_CHROMA_DEEP + 1] = 154;  /* INIT_QT_CBF[2][6..7], context.c:130-134: qt_cbf_model_chroma[2..3] */
    for (int i = 0; i < CX_COUNT; i++) m->ctx_init[i] = (uint8_t)ctx_state(qp, inits[i]);
    m->ctx_init[KVZ_HIP_CX_SAO_MERGE] = (uint8_t)ctx_state(qp, 153);  /* context.c:38-39, I slice (index 2) */
    m->ctx_init[KVZ_HIP_CX_SAO_TYPE] = (uint8_t)ctx_state(qp, 200);
    m->coeff_cabac = qp >= 28;  /* `ultrafast`: fast-residual-cost 28 (cfg.c:485-512); MAX_FAST_COEFF_COST_QP = 50 lies above */
    memcpy(m->entropy_fbits, entropy_fbits, sizeof m->entropy_fbits);
    m->adaptive = 1;  /* kvazaar's behaviour; 0 freezes every context at its slice-start state */
  }
#define CTX_STATE(init) ctx_state(qp, init)
#define FILL2(dst, init) do { int s_ = CTX_STATE(init); (dst)[0] = entropy_fbits[s_ ^ 0]; (dst)[1] = entropy_fbits[s_ ^ 1]; } while (0)
  for (int i = 0; i < 3; i++) FILL2(m->split_flag[i], init_split[i]);
  FILL2(m->part_size, init_part);
  FILL2(m->intra_mode, init_intra);
  FILL2(m->chroma_mode, init_chroma);
  for (int i = 0; i < 2; i++) { FILL2(m->cbf_luma[i], init_cbf_luma[i]); FILL2(m->cbf_chroma[i], init_cbf_chroma[i]); }
}

#include "kvz_oracle_inter.inc"
#include "kvz_oracle_entropy.inc"
