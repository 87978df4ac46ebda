/*
 * kvz_oracle_rdoq.c -- TEST INFRASTRUCTURE (see kvz_oracle.h): CPU restatement of kvazaar's rate-distortion optimised quantisation
 * (kvz_rdoq, rdo.c:661-1000, with kvz_get_ic_rate :345-392, kvz_get_coded_level :413-459, get_rate_last :465-478, calc_last_bits :480-509,
 * find_last_scanpos quant-generic.c:379-399, the context derivations context.c:315-399) for intra blocks, flat scaling lists, sign hiding
 * off, 8 bit.  Pinned against the compiled reference's kvz_rdoq on random blocks x context states by tests/test_rdoq.py.
 */
#include <stdlib.h>
#include <string.h>

#include "kvz_oracle.h"

#define ORC_MIN(a, b) ((a) < (b) ? (a) : (b))

typedef struct { const uint8_t *ctx; const uint32_t *bits; double lambda; } rdoq_ctx;
static int32_t price(const rdoq_ctx *c, int idx, int bin) { return (int32_t)c->bits[c->ctx[idx] ^ bin]; }

// rdo.c:345-392 kvz_get_ic_rate
static int32_t rdoq_ic_rate(const rdoq_ctx *c, uint32_t abs_level, int ctx_one, int ctx_abs, int go_rice, uint32_t c1_idx, uint32_t c2_idx, int type)
{
  int32_t rate = 1 << 15;
  const uint32_t base_level = c1_idx < 8 ? (2 + (c2_idx < 1)) : 1;  // C1FLAG_NUMBER 8, C2FLAG_NUMBER 1
  const int one0 = (type == 0 ? KVZ_HIP_CX_ONE_LUMA : KVZ_HIP_CX_ONE_CHROMA) + ctx_one, abs0 = (type == 0 ? KVZ_HIP_CX_ABS_LUMA : KVZ_HIP_CX_ABS_CHROMA) + ctx_abs;
  if (abs_level >= base_level) {
    int32_t symbol = (int32_t)(abs_level - base_level), length;
    if (symbol < (3 << go_rice)) {
      length = symbol >> go_rice;
      rate += (length + 1 + go_rice) * (1 << 15);
    } else {
      length = go_rice;
      symbol = symbol - (3 << go_rice);
      while (symbol >= (1 << length)) symbol -= (1 << (length++));
      rate += (3 + length + 1 - go_rice + length) * (1 << 15);
    }
    if (c1_idx < 8) {
      rate += price(c, one0, 1);
      if (c2_idx < 1) rate += price(c, abs0, 1);
    }
  } else if (abs_level == 1) {
    rate += price(c, one0, 0);
  } else if (abs_level == 2) {
    rate += price(c, one0, 1);
    rate += price(c, abs0, 0);
  }
  return rate;
}

// rdo.c:413-459 kvz_get_coded_level
static uint32_t rdoq_coded_level(const rdoq_ctx *c, double *coded_cost, double *coded_cost0, double *coded_cost_sig, int32_t level_double, uint32_t max_abs_level, int ctx_sig, int ctx_one,
                             int ctx_abs, int go_rice, uint32_t c1_idx, uint32_t c2_idx, int32_t q_bits, double temp, int last, int type)
{
  double cur_cost_sig = 0;
  uint32_t best_abs_level = 0;
  const int sig0 = (type ? KVZ_HIP_CX_SIG_CHROMA : KVZ_HIP_CX_SIG_LUMA) + ctx_sig;
  if (!last && max_abs_level < 3) {
    *coded_cost_sig = c->lambda * price(c, sig0, 0);
    *coded_cost = *coded_cost0 + *coded_cost_sig;
    if (max_abs_level == 0) return best_abs_level;
  } else {
    *coded_cost = 1.7e+308;  // MAX_DOUBLE (global.h)
  }
  if (!last) cur_cost_sig = c->lambda * price(c, sig0, 1);
  const int32_t min_abs_level = max_abs_level > 1 ? (int32_t)max_abs_level - 1 : 1;
  for (int32_t abs_level = (int32_t)max_abs_level; abs_level >= min_abs_level; abs_level--) {
    const double err = (double)(level_double - (abs_level * (1 << q_bits)));
    double cur_cost = err * err * temp + c->lambda * rdoq_ic_rate(c, (uint32_t)abs_level, ctx_one, ctx_abs, go_rice, c1_idx, c2_idx, type);
    cur_cost += cur_cost_sig;
    if (cur_cost < *coded_cost) {
      best_abs_level = (uint32_t)abs_level;
      *coded_cost = cur_cost;
      *coded_cost_sig = cur_cost_sig;
    }
  }
  return best_abs_level;
}

static int rdoq_group_idx(int pos)  // g_group_idx (rdo.c:60): index of the last-position prefix group
{
  return pos < 4 ? pos : (pos < 6 ? 4 : (pos < 8 ? 5 : (pos < 12 ? 6 : (pos < 16 ? 7 : (pos < 24 ? 8 : 9)))));
}

// quant tables of the flat lists: kvz_g_quant_scales (scalinglist.c:78)
static int rdoq_quant_scale(int qp_rem)
{
  return qp_rem == 0 ? 26214 : (qp_rem == 1 ? 23302 : (qp_rem == 2 ? 20560 : (qp_rem == 3 ? 18396 : (qp_rem == 4 ? 16384 : 14564))));
}

// chroma QP of a luma QP (kvz_get_scaled_qp, transform.c:141-155 with kvz_g_chroma_scale :56-62: H.265 table 8-10), 8 bit
static int rdoq_scaled_qp(int type, int qp)
{
  if (type == 0) return qp;
  const int q = (qp < 0 ? 0 : (qp > 57 ? 57 : qp));
  if (q < 30) return q;
  if (q >= 43) return q - 6;
  const int tab[13] = { 29, 30, 31, 32, 33, 33, 34, 34, 35, 35, 36, 36, 37 };  // kvz_g_chroma_scale[30..42]
  return tab[q - 30];
}

// context.c:366-399 kvz_context_get_sig_ctx_inc
static int rdoq_sig_ctx_inc(int pattern, int scan_idx, int pos_x, int pos_y, int log2w, int type)
{
  if (pos_x + pos_y == 0) return 0;
  if (log2w == 2) {
    const unsigned long long map = 0x8877886654325410ull;  // ctx_ind_map[16], one nibble each, entry 0 lowest
    return (int)((map >> (4 * (4 * pos_y + pos_x))) & 15);
  }
  const int offset = log2w == 3 ? (scan_idx == 0 ? 9 : 15) : (type == 0 ? 21 : 12);
  const int xs = pos_x & 3, ys = pos_y & 3;
  int cnt;
  if (pattern == 0) cnt = (xs + ys <= 2) ? ((xs + ys == 0) ? 2 : 1) : 0;
  else if (pattern == 1) cnt = (ys <= 1) ? ((ys == 0) ? 2 : 1) : 0;
  else if (pattern == 2) cnt = (xs <= 1) ? ((xs == 0) ? 2 : 1) : 0;
  else cnt = 2;
  return ((type == 0 && ((pos_x >> 2) + (pos_y >> 2)) > 0) ? 3 : 0) + offset + cnt;
}

// The block.  coef: transform coefficients (row-major w x w); dest: quantised levels (out); scan / scan_cg: kvz_g_sig_last_scan[scan_mode][log2w - 1]
// and g_sig_last_scan_cg[log2w - 2][scan_mode] as raster indices per scan position; cost3: 3 * w * w doubles of scratch.
static void rdoq_block(const rdoq_ctx *c, int qp, const int16_t *coef, int16_t *dest, int log2w, int type /* 0 luma, 2 chroma */, int scan_mode, int tr_depth, const uint32_t *scan,
                        const uint32_t *scan_cg, double *cost3)
{
  const int width = 1 << log2w, n = width * width;
  const int transform_shift = 15 - 8 - log2w;
  const int qp_scaled = rdoq_scaled_qp(type, qp);
  const int32_t q_bits = 14 + qp_scaled / 6 + transform_shift;
  const int32_t q = rdoq_quant_scale(qp_scaled % 6);
  // scalinglist.c:349-367: err_scale = 2^15 * 2^(-2 transform_shift) / q / q
  double scale = 32768.0;
  for (int i = 0; i < 2 * transform_shift; i++) scale = scale * 0.5;  // pow(2.0, -2.0 * transform_shift): exact either way
  for (int i = 0; i > 2 * transform_shift; i--) scale = scale * 2.0;
  const double temp = scale / (double)q / (double)q;
  double *cost_coeff = cost3, *cost_sig = cost3 + n, *cost_coeff0 = cost3 + 2 * n;
  const int num_blk_side = width >> 2, cg_num = n >> 4;
  double cost_coeffgroup_sig[64];
  uint32_t sig_coeffgroup_flag[64];
  for (int i = 0; i < cg_num; i++) sig_coeffgroup_flag[i] = 0;
  int ctx_set = 0, c1 = 1, c2 = 0, go_rice = 0;
  double base_cost = 0, block_uncoded_cost = 0;
  uint32_t c1_idx = 0, c2_idx = 0;
  // quant-generic.c:379-399 find_last_scanpos (zeroes dest above the last position it finds)
  int cg_last_scanpos = -1, last_scanpos = -1, cg_scanpos;
  for (cg_scanpos = cg_num - 1; cg_scanpos >= 0 && last_scanpos < 0; cg_scanpos--) {
    for (int in_cg = 15; in_cg >= 0; in_cg--) {
      const int scanpos = cg_scanpos * 16 + in_cg;
      const uint32_t blkpos = scan[scanpos];
      int32_t level_double = coef[blkpos];
      level_double = ORC_MIN(abs(level_double) * q, 0x7fffffff - (1 << (q_bits - 1)));
      if (((level_double + (1 << (q_bits - 1))) >> q_bits) > 0) {
        last_scanpos = scanpos;
        ctx_set = (scanpos > 0 && type == 0) ? 2 : 0;
        cg_last_scanpos = cg_scanpos;
        break;
      }
      dest[blkpos] = 0;
    }
    if (last_scanpos >= 0) break;
  }
  if (last_scanpos == -1) return;
  for (; cg_scanpos >= 0; cg_scanpos--) cost_coeffgroup_sig[cg_scanpos] = 0;
  // rdo.c:480-509 calc_last_bits
  int32_t last_x_bits[32], last_y_bits[32];
  {
    const int cb = log2w - 2;
    const int off = type ? 0 : (cb * 3 + ((cb + 1) >> 2)), shift = type ? cb : ((cb + 3) >> 2);
    const int bx = (type ? KVZ_HIP_CX_LAST_X_CHROMA : KVZ_HIP_CX_LAST_X_LUMA) + off, by = (type ? KVZ_HIP_CX_LAST_Y_CHROMA : KVZ_HIP_CX_LAST_Y_LUMA) + off;
    int32_t bits_x = 0, bits_y = 0;
    int k;
    for (k = 0; k < rdoq_group_idx(width - 1); k++) {
      last_x_bits[k] = bits_x + price(c, bx + (k >> shift), 0);
      bits_x += price(c, bx + (k >> shift), 1);
    }
    last_x_bits[k] = bits_x;
    for (k = 0; k < rdoq_group_idx(width - 1); k++) {
      last_y_bits[k] = bits_y + price(c, by + (k >> shift), 0);
      bits_y += price(c, by + (k >> shift), 1);
    }
    last_y_bits[k] = bits_y;
  }
  const int cg0 = KVZ_HIP_CX_SIG_CG + type;
  for (int cgs = cg_last_scanpos; cgs >= 0; cgs--) {
    const uint32_t cg_blkpos = scan_cg[cgs], cg_pos_y = cg_blkpos / num_blk_side, cg_pos_x = cg_blkpos - cg_pos_y * num_blk_side;
    // context.c:339-351 / 315-327
    uint32_t right = 0, lower = 0;
    if ((int)cg_pos_x < num_blk_side - 1) right = sig_coeffgroup_flag[cg_pos_y * num_blk_side + cg_pos_x + 1] != 0;
    if ((int)cg_pos_y < num_blk_side - 1) lower = sig_coeffgroup_flag[(cg_pos_y + 1) * num_blk_side + cg_pos_x] != 0;
    const int pattern_sig_ctx = width == 4 ? -1 : (int)(right + (lower << 1));
    double rd_coded_level_and_dist = 0, rd_uncoded_dist = 0, rd_sig_cost = 0, rd_sig_cost_0 = 0;
    int rd_nnz_before_pos0 = 0;
    for (int in_cg = 15; in_cg >= 0; in_cg--) {
      const int scanpos = cgs * 16 + in_cg;
      if (scanpos > last_scanpos) continue;
      const uint32_t blkpos = scan[scanpos];
      int32_t level_double = coef[blkpos];
      level_double = ORC_MIN(abs(level_double) * q, 0x7fffffff - (1 << (q_bits - 1)));
      const uint32_t max_abs_level = (uint32_t)((level_double + (1 << (q_bits - 1))) >> q_bits);
      const double err = (double)level_double;
      cost_coeff0[scanpos] = err * err * temp;
      block_uncoded_cost += cost_coeff0[scanpos];
      const int one_ctx = 4 * ctx_set + c1, abs_ctx = ctx_set + c2;
      int32_t level;
      if (scanpos == last_scanpos) {
        level = (int32_t)rdoq_coded_level(c, &cost_coeff[scanpos], &cost_coeff0[scanpos], &cost_sig[scanpos], level_double, max_abs_level, 0, one_ctx, abs_ctx, go_rice, c1_idx, c2_idx,
                                      q_bits, temp, 1, type);
      } else {
        const uint32_t pos_y = blkpos >> log2w, pos_x = blkpos - (pos_y << log2w);
        const int ctx_sig = rdoq_sig_ctx_inc(pattern_sig_ctx, scan_mode, (int)pos_x, (int)pos_y, log2w, type);
        level = (int32_t)rdoq_coded_level(c, &cost_coeff[scanpos], &cost_coeff0[scanpos], &cost_sig[scanpos], level_double, max_abs_level, ctx_sig, one_ctx, abs_ctx, go_rice, c1_idx,
                                      c2_idx, q_bits, temp, 0, type);
      }
      dest[blkpos] = (int16_t)level;
      base_cost += cost_coeff[scanpos];
      const int32_t base_level = c1_idx < 8 ? (2 + (c2_idx < 1)) : 1;
      if (level >= base_level && level > 3 * (1 << go_rice)) go_rice = ORC_MIN(go_rice + 1, 4);
      if (level >= 1) c1_idx++;
      if (level > 1) {
        c1 = 0;
        c2 += (c2 < 2);
        c2_idx++;
      } else if (c1 < 3 && c1 > 0 && level) {
        c1++;
      }
      if ((scanpos % 16 == 0) && scanpos > 0) {
        c2 = 0;
        go_rice = 0;
        c1_idx = 0;
        c2_idx = 0;
        ctx_set = (scanpos == 16 || type != 0) ? 0 : 2;
        if (c1 == 0) ctx_set++;
        c1 = 1;
      }
      rd_sig_cost += cost_sig[scanpos];
      if (in_cg == 0) rd_sig_cost_0 = cost_sig[scanpos];
      if (dest[blkpos]) {
        sig_coeffgroup_flag[cg_blkpos] = 1;
        rd_coded_level_and_dist += cost_coeff[scanpos] - cost_sig[scanpos];
        rd_uncoded_dist += cost_coeff0[scanpos];
        if (in_cg != 0) rd_nnz_before_pos0++;
      }
    }
    if (cgs) {
      // the flags may have changed inside the loop above only for this group: right / lower are those of the groups coded before
      const int ctx_sig = (int)(right || lower);
      if (sig_coeffgroup_flag[cg_blkpos] == 0) {
        cost_coeffgroup_sig[cgs] = c->lambda * price(c, cg0 + ctx_sig, 0);
        base_cost += cost_coeffgroup_sig[cgs] - rd_sig_cost;
      } else if (cgs < cg_last_scanpos) {
        if (rd_nnz_before_pos0 == 0) {
          base_cost -= rd_sig_cost_0;
          rd_sig_cost -= rd_sig_cost_0;
        }
        double cost_zero_cg = base_cost;
        cost_coeffgroup_sig[cgs] = c->lambda * price(c, cg0 + ctx_sig, 1);
        base_cost += cost_coeffgroup_sig[cgs];
        cost_zero_cg += c->lambda * price(c, cg0 + ctx_sig, 0);
        cost_zero_cg += rd_uncoded_dist;
        cost_zero_cg -= rd_coded_level_and_dist;
        cost_zero_cg -= rd_sig_cost;
        if (cost_zero_cg < base_cost) {
          sig_coeffgroup_flag[cg_blkpos] = 0;
          base_cost = cost_zero_cg;
          cost_coeffgroup_sig[cgs] = c->lambda * price(c, cg0 + ctx_sig, 0);
          for (int in_cg = 15; in_cg >= 0; in_cg--) {
            const int scanpos = cgs * 16 + in_cg;
            const uint32_t blkpos = scan[scanpos];
            if (dest[blkpos]) {
              dest[blkpos] = 0;
              cost_coeff[scanpos] = cost_coeff0[scanpos];
              cost_sig[scanpos] = 0;
            }
          }
        }
      }
    } else {
      sig_coeffgroup_flag[cg_blkpos] = 1;
    }
  }
  // ---- the last position (rdo.c:903-957), intra block: coded block flag of the transform unit
  double best_cost;
  int best_last_idx_p1 = 0;
  int found_last = 0;
  {
    /* rdo.c:907-915: qt_cbf_model_luma[!tr_depth] / qt_cbf_model_chroma[tr_depth].  The chroma blocks of an NxN CU arrive with tr_depth 2 (quant-generic.c:237-238
     * adds one for the partition): qt_cbf_model_chroma[2..3] are KVZ_HIP_CX_CBF_CHROMA_DEEP in the context layout */
    const int ctx_cbf = type == 0 ? KVZ_HIP_CX_CBF_LUMA + !tr_depth : (tr_depth < 2 ? KVZ_HIP_CX_CBF_CHROMA + tr_depth : KVZ_HIP_CX_CBF_CHROMA_DEEP + (tr_depth > 3 ? 3 : tr_depth) - 2);
    best_cost = block_uncoded_cost + c->lambda * price(c, ctx_cbf, 0);
    base_cost += c->lambda * price(c, ctx_cbf, 1);
  }
  for (int cgs = cg_last_scanpos; cgs >= 0; cgs--) {
    const uint32_t cg_blkpos = scan_cg[cgs];
    base_cost -= cost_coeffgroup_sig[cgs];
    if (sig_coeffgroup_flag[cg_blkpos]) {
      for (int in_cg = 15; in_cg >= 0; in_cg--) {
        const int scanpos = cgs * 16 + in_cg;
        if (scanpos > last_scanpos) continue;
        const uint32_t blkpos = scan[scanpos];
        if (dest[blkpos]) {
          const uint32_t pos_y = blkpos >> log2w, pos_x = blkpos - (pos_y << log2w);
          const uint32_t px = scan_mode == 2 ? pos_y : pos_x, py = scan_mode == 2 ? pos_x : pos_y;  // SCAN_VER swaps (rdo.c:934)
          // rdo.c:465-478 get_rate_last
          const int gx = rdoq_group_idx((int)px), gy = rdoq_group_idx((int)py);
          double ui_cost = last_x_bits[gx] + last_y_bits[gy];
          if (gx > 3) ui_cost += (double)((1 << 15) * ((gx - 2) >> 1));
          if (gy > 3) ui_cost += (double)((1 << 15) * ((gy - 2) >> 1));
          const double cost_last = c->lambda * ui_cost;
          const double total = base_cost + cost_last - cost_sig[scanpos];
          if (total < best_cost) {
            best_last_idx_p1 = scanpos + 1;
            best_cost = total;
          }
          if (dest[blkpos] > 1) { found_last = 1; break; }
          base_cost -= cost_coeff[scanpos];
          base_cost += cost_coeff0[scanpos];
        } else {
          base_cost -= cost_sig[scanpos];
        }
      }
      if (found_last) break;
    }
  }
  for (int scanpos = 0; scanpos < best_last_idx_p1; scanpos++) {
    const uint32_t blkpos = scan[scanpos];
    const int32_t level = dest[blkpos];
    dest[blkpos] = (int16_t)(coef[blkpos] < 0 ? -level : level);
  }
  for (int scanpos = best_last_idx_p1; scanpos <= last_scanpos; scanpos++) dest[scan[scanpos]] = 0;
}


/* context.c:202-213 kvz_ctx_init: init value + QP -> uc_state */
/* Q15 entropy table (rdo.c:69-80 kvz_entropy_bits), regenerated from the float table the model carries: entropy_fbits[i] * 32768 is exact */
void kvz_oracle_rdoq(int qp, double lambda, const uint8_t *ctx_states, const float *entropy_fbits, const int16_t *coef, int16_t *dest, int width, int type,
                     int scan_mode, int tr_depth)
{
  uint32_t bits[128];
  for (int i = 0; i < 128; i++) bits[i] = (uint32_t)(entropy_fbits[i] * 32768.0f);
  rdoq_ctx c = { ctx_states, bits, lambda };
  int log2w = 2;
  while ((1 << log2w) < width) log2w++;
  double *cost3 = malloc(sizeof(double) * 3 * width * width);
  /* coefficient-group scan order (tables.h:45-89 g_sig_last_scan_cg): 8x8 blocks have a 2x2 order per scan pattern; 16x16 / 32x32 blocks (always scanned
   * diagonally) the plain up-right diagonal order of the 4x4 / 8x8 grid of groups */
  uint32_t scan_cg[64];
  const int side = width >> 2;
  if (side <= 2) {
    const uint32_t d[3][4] = { { 0, 2, 1, 3 }, { 0, 1, 2, 3 }, { 0, 2, 1, 3 } };
    memcpy(scan_cg, d[scan_mode], sizeof d[0]);
  } else {
    int k = 0;
    for (int diag = 0; diag <= 2 * (side - 1); diag++)
      for (int y = diag < side ? diag : side - 1; y >= 0 && diag - y < side; y--) scan_cg[k++] = (uint32_t)(y * side + (diag - y));
  }
  rdoq_block(&c, qp, coef, dest, log2w, type, scan_mode, tr_depth, kvz_oracle_scan_table(scan_mode, log2w), scan_cg, cost3);
  free(cost3);
}
