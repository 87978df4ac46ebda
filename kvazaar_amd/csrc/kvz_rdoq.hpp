// kvz_rdoq.hpp -- rate-distortion optimised quantisation of one transform block (kvz_rdoq, rdo.c:661-1000, HM's RDOQ): what
// kvz_quantize_residual runs instead of kvz_quant when --rdoq is on (quant-generic.c:234-244; presets medium and slower).  Sign hiding off
// (its table bookkeeping, rdo.c:780-795, 990-992, is not carried), flat scaling lists, 8 bit.
//
// Per block, serial by construction (every decision feeds the context selection of the next coefficient): levels are chosen coefficient by
// coefficient from the last significant scan position down (kvz_get_coded_level: distortion err^2 * err_scale + lambda * rate, rates from
// the Q15 entropy table on the CALLER's context states -- state->cabac, not the search copy), then whole coefficient groups are tested
// against zeroing, then the best last position is chosen.  All costs are doubles combined in the reference's order (-ffp-contract=off).
// One lane per block in the batched form; the three per-position cost arrays live in caller-provided scratch (3 * w * w doubles per block).
#pragma once
#include "kvz_ops.hpp"

namespace kvz {

struct RdoqCtx {
  const u8 *ctx;        // uc_state of the contexts in KVZ_HIP_CX_* order (include/kvz_hip_types.h)
  const u32 *bits;      // kvz_entropy_bits (rdo.c:69-80): Q15 price of coding `bin` in state s = bits[s ^ bin]
  double lambda;
  const i32 *ptab = nullptr;     // prices of both bins of every context at the caller's states, [2 * idx + bin] (the CTU kernel builds it once per CTU in LDS:
                                 // the states do not move while a CTU is searched); used instead of ctx / bits when set
  KVZ_DEV i32 price(int idx, int bin) const { return ptab ? ptab[2 * idx + bin] : (i32)bits[ctx[idx] ^ bin]; }
};

// The coefficient scans by arithmetic (HEVC scans are hierarchical: 4x4 groups in group order, sixteen positions inside a group): no table in memory on the
// chain from one coefficient to the next.  diag8: the up-right diagonal order of an 8x8 grid (Tables::diag8), the group order of a 32x32 block.
struct RdoqScan {
  int log2w, mode;
  const u8 *diag8;
  KVZ_DEV static u32 in_group(int scan, int k)  // raster index inside the 4x4 group of its k-th position (tables.c kvz_g_sig_last_scan, 4x4 entries)
  {
    const unsigned long long pat = scan == 0 ? 0xfbe7ad369c258140ull : (scan == 1 ? 0xfedcba9876543210ull : 0xfb73ea62d951c840ull);
    return (u32)((pat >> (4 * k)) & 15);
  }
  KVZ_DEV u32 cg(int i) const  // raster index of the i-th group in group order (tables.h:45-89 g_sig_last_scan_cg)
  {
    if (log2w == 2) return 0;
    if (log2w == 3) return mode == 1 ? (u32)i : (u32)((0x3120 >> (4 * i)) & 3);
    if (log2w == 4) return in_group(0, i);
    return diag8[i];
  }
  KVZ_DEV u32 pos(int scanpos) const  // raster index of a scan position in the block
  {
    const u32 g = cg(scanpos >> 4), r = in_group(mode, scanpos & 15), side = 1u << (log2w - 2);
    return ((((g >> (log2w - 2)) << 2) + (r >> 2)) << log2w) + ((g & (side - 1)) << 2) + (r & 3);
  }
};

// rdo.c:345-392 kvz_get_ic_rate
KVZ_DEV i32 rdoq_ic_rate(const RdoqCtx &c, u32 abs_level, int ctx_one, int ctx_abs, int go_rice, u32 c1_idx, u32 c2_idx, int type)
{
  i32 rate = 1 << 15;
  const u32 base_level = c1_idx < 8 ? (2 + (c2_idx < 1)) : 1;  // C1FLAG_NUMBER 8, C2FLAG_NUMBER 1
  const int one0 = (type == 0 ? KVZ_HIP_CX_ONE_LUMA : KVZ_HIP_CX_ONE_CHROMA) + ctx_one, abs0 = (type == 0 ? KVZ_HIP_CX_ABS_LUMA : KVZ_HIP_CX_ABS_CHROMA) + ctx_abs;
  if (abs_level >= base_level) {
    i32 symbol = (i32)(abs_level - base_level), length;
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
      rate += c.price(one0, 1);
      if (c2_idx < 1) rate += c.price(abs0, 1);
    }
  } else if (abs_level == 1) {
    rate += c.price(one0, 0);
  } else if (abs_level == 2) {
    rate += c.price(one0, 1);
    rate += c.price(abs0, 0);
  }
  return rate;
}

// rdo.c:413-459 kvz_get_coded_level
KVZ_DEV u32 rdoq_coded_level(const RdoqCtx &c, double *coded_cost, double *coded_cost0, double *coded_cost_sig, i32 level_double, u32 max_abs_level, int ctx_sig, int ctx_one,
                             int ctx_abs, int go_rice, u32 c1_idx, u32 c2_idx, i32 q_bits, double temp, bool last, int type)
{
  double cur_cost_sig = 0;
  u32 best_abs_level = 0;
  const int sig0 = (type ? KVZ_HIP_CX_SIG_CHROMA : KVZ_HIP_CX_SIG_LUMA) + ctx_sig;
  if (!last && max_abs_level < 3) {
    *coded_cost_sig = c.lambda * c.price(sig0, 0);
    *coded_cost = *coded_cost0 + *coded_cost_sig;
    if (max_abs_level == 0) return best_abs_level;
  } else {
    *coded_cost = 1.7e+308;  // MAX_DOUBLE (global.h)
  }
  if (!last) cur_cost_sig = c.lambda * c.price(sig0, 1);
  const i32 min_abs_level = max_abs_level > 1 ? (i32)max_abs_level - 1 : 1;
  for (i32 abs_level = (i32)max_abs_level; abs_level >= min_abs_level; abs_level--) {
    const double err = (double)(level_double - (abs_level * (1 << q_bits)));
    double cur_cost = err * err * temp + c.lambda * rdoq_ic_rate(c, (u32)abs_level, ctx_one, ctx_abs, go_rice, c1_idx, c2_idx, type);
    cur_cost += cur_cost_sig;
    if (cur_cost < *coded_cost) {
      best_abs_level = (u32)abs_level;
      *coded_cost = cur_cost;
      *coded_cost_sig = cur_cost_sig;
    }
  }
  return best_abs_level;
}

KVZ_DEV int rdoq_group_idx(int pos)  // g_group_idx (rdo.c:60): index of the last-position prefix group
{
  return pos < 4 ? pos : (pos < 6 ? 4 : (pos < 8 ? 5 : (pos < 12 ? 6 : (pos < 16 ? 7 : (pos < 24 ? 8 : 9)))));
}

// quant tables of the flat lists: kvz_g_quant_scales (scalinglist.c:78)
KVZ_DEV int rdoq_quant_scale(int qp_rem)
{
  return qp_rem == 0 ? 26214 : (qp_rem == 1 ? 23302 : (qp_rem == 2 ? 20560 : (qp_rem == 3 ? 18396 : (qp_rem == 4 ? 16384 : 14564))));
}

// chroma QP of a luma QP (kvz_get_scaled_qp, transform.c:141-155 with kvz_g_chroma_scale :56-62: H.265 table 8-10), 8 bit
KVZ_DEV int rdoq_scaled_qp(int type, int qp)
{
  if (type == 0) return qp;
  const int q = iclip(0, 57, qp);
  if (q < 30) return q;
  if (q >= 43) return q - 6;
  const int tab[13] = { 29, 30, 31, 32, 33, 33, 34, 34, 35, 35, 36, 36, 37 };  // kvz_g_chroma_scale[30..42]
  return tab[q - 30];
}

// context.c:366-399 kvz_context_get_sig_ctx_inc
KVZ_DEV int rdoq_sig_ctx_inc(int pattern, int scan_idx, int pos_x, int pos_y, int log2w, int type)
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

// The block.  coef: transform coefficients (row-major w x w); dest: quantised levels (out); diag8: Tables::diag8; cost3: 3 * w * w doubles of scratch.
// -DKVZ_RDOQ_CALL: a real function call on the device instead of a dozen inlined copies in the CTU program (A/B: 20.2 k vs 22.4 k CTUs/s -- the callee's 256+
// registers leave one wavefront per SIMD; the instruction cache is not what limits the pass).
#if defined(KVZ_RDOQ_CALL) && !defined(KVZ_HOSTSIM)
#define KVZ_RDOQ_NOINLINE __attribute__((noinline))
#else
#define KVZ_RDOQ_NOINLINE
#endif
KVZ_DEV KVZ_RDOQ_NOINLINE void rdoq_block(const RdoqCtx &c, int qp, const i16 *coef, i16 *dest, int log2w, int type /* 0 luma, 2 chroma */, int scan_mode, int tr_depth, const u8 *diag8, double *cost3)
{
  const int width = 1 << log2w, n = width * width;
  const int transform_shift = 15 - 8 - log2w;
  const int qp_scaled = rdoq_scaled_qp(type, qp);
  const i32 q_bits = 14 + qp_scaled / 6 + transform_shift;
  const i32 q = rdoq_quant_scale(qp_scaled % 6);
  // scalinglist.c:349-367: err_scale = 2^15 * 2^(-2 transform_shift) / q / q
  double scale = 32768.0;
  for (int i = 0; i < 2 * transform_shift; i++) scale = scale * 0.5;  // pow(2.0, -2.0 * transform_shift): exact either way
  for (int i = 0; i > 2 * transform_shift; i--) scale = scale * 2.0;
  const double temp = scale / (double)q / (double)q;
  double *cost_coeff = cost3, *cost_sig = cost3 + n, *cost_coeff0 = cost3 + 2 * n;
  const int num_blk_side = width >> 2, cg_num = n >> 4;
  double cost_coeffgroup_sig[64];
  const RdoqScan sc{ log2w, scan_mode, diag8 };
  unsigned long long sig_groups = 0;  // sig_coeffgroup_flag, bit = raster index of the group
  int ctx_set = 0, c1 = 1, c2 = 0, go_rice = 0;
  double base_cost = 0, block_uncoded_cost = 0;
  u32 c1_idx = 0, c2_idx = 0;
  // quant-generic.c:379-399 find_last_scanpos (zeroes dest above the last position it finds)
  int cg_last_scanpos = -1, last_scanpos = -1, cg_scanpos;
  for (cg_scanpos = cg_num - 1; cg_scanpos >= 0 && last_scanpos < 0; cg_scanpos--) {
    for (int in_cg = 15; in_cg >= 0; in_cg--) {
      const int scanpos = cg_scanpos * 16 + in_cg;
      const u32 blkpos = sc.pos(scanpos);
      i32 level_double = coef[blkpos];
      level_double = imin(iabs(level_double) * q, 0x7fffffff - (1 << (q_bits - 1)));
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
  i32 last_x_bits[32], last_y_bits[32];
  {
    const int cb = log2w - 2;
    const int off = type ? 0 : (cb * 3 + ((cb + 1) >> 2)), shift = type ? cb : ((cb + 3) >> 2);
    const int bx = (type ? KVZ_HIP_CX_LAST_X_CHROMA : KVZ_HIP_CX_LAST_X_LUMA) + off, by = (type ? KVZ_HIP_CX_LAST_Y_CHROMA : KVZ_HIP_CX_LAST_Y_LUMA) + off;
    i32 bits_x = 0, bits_y = 0;
    int k;
    for (k = 0; k < rdoq_group_idx(width - 1); k++) {
      last_x_bits[k] = bits_x + c.price(bx + (k >> shift), 0);
      bits_x += c.price(bx + (k >> shift), 1);
    }
    last_x_bits[k] = bits_x;
    for (k = 0; k < rdoq_group_idx(width - 1); k++) {
      last_y_bits[k] = bits_y + c.price(by + (k >> shift), 0);
      bits_y += c.price(by + (k >> shift), 1);
    }
    last_y_bits[k] = bits_y;
  }
  const int cg0 = KVZ_HIP_CX_SIG_CG + type;
  for (int cgs = cg_last_scanpos; cgs >= 0; cgs--) {
    const u32 cg_blkpos = sc.cg(cgs), cg_pos_y = cg_blkpos / num_blk_side, cg_pos_x = cg_blkpos - cg_pos_y * num_blk_side;
    // context.c:339-351 / 315-327
    u32 right = 0, lower = 0;
    if ((int)cg_pos_x < num_blk_side - 1) right = (sig_groups >> (cg_pos_y * num_blk_side + cg_pos_x + 1)) & 1;
    if ((int)cg_pos_y < num_blk_side - 1) lower = (sig_groups >> ((cg_pos_y + 1) * num_blk_side + cg_pos_x)) & 1;
    const int pattern_sig_ctx = width == 4 ? -1 : (int)(right + (lower << 1));
    double rd_coded_level_and_dist = 0, rd_uncoded_dist = 0, rd_sig_cost = 0, rd_sig_cost_0 = 0;
    int rd_nnz_before_pos0 = 0;
    for (int in_cg = 15; in_cg >= 0; in_cg--) {
      const int scanpos = cgs * 16 + in_cg;
      if (scanpos > last_scanpos) continue;
      const u32 blkpos = sc.pos(scanpos);
      i32 level_double = coef[blkpos];
      level_double = imin(iabs(level_double) * q, 0x7fffffff - (1 << (q_bits - 1)));
      const u32 max_abs_level = (u32)((level_double + (1 << (q_bits - 1))) >> q_bits);
      const double err = (double)level_double;
      // the position's three costs stay in registers while they are used here; the arrays are only written (the later passes read them back)
      double c0v = err * err * temp, ccv, csv = 0;
      block_uncoded_cost += c0v;
      const int one_ctx = 4 * ctx_set + c1, abs_ctx = ctx_set + c2;
      i32 level;
      if (scanpos == last_scanpos) {
        level = (i32)rdoq_coded_level(c, &ccv, &c0v, &csv, level_double, max_abs_level, 0, one_ctx, abs_ctx, go_rice, c1_idx, c2_idx, q_bits, temp, true, type);
      } else {
        const u32 pos_y = blkpos >> log2w, pos_x = blkpos - (pos_y << log2w);
        const int ctx_sig = rdoq_sig_ctx_inc(pattern_sig_ctx, scan_mode, (int)pos_x, (int)pos_y, log2w, type);
        level = (i32)rdoq_coded_level(c, &ccv, &c0v, &csv, level_double, max_abs_level, ctx_sig, one_ctx, abs_ctx, go_rice, c1_idx, c2_idx, q_bits, temp, false, type);
      }
      cost_coeff[scanpos] = ccv; cost_coeff0[scanpos] = c0v; cost_sig[scanpos] = csv;
      dest[blkpos] = (i16)level;
      base_cost += ccv;
      const i32 base_level = c1_idx < 8 ? (2 + (c2_idx < 1)) : 1;
      if (level >= base_level && level > 3 * (1 << go_rice)) go_rice = imin(go_rice + 1, 4);
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
      rd_sig_cost += csv;
      if (in_cg == 0) rd_sig_cost_0 = csv;
      if (level) {
        sig_groups |= 1ull << cg_blkpos;
        rd_coded_level_and_dist += ccv - csv;
        rd_uncoded_dist += c0v;
        if (in_cg != 0) rd_nnz_before_pos0++;
      }
    }
    if (cgs) {
      // the flags may have changed inside the loop above only for this group: right / lower are those of the groups coded before
      const int ctx_sig = (int)(right || lower);
      if (((sig_groups >> cg_blkpos) & 1) == 0) {
        cost_coeffgroup_sig[cgs] = c.lambda * c.price(cg0 + ctx_sig, 0);
        base_cost += cost_coeffgroup_sig[cgs] - rd_sig_cost;
      } else if (cgs < cg_last_scanpos) {
        if (rd_nnz_before_pos0 == 0) {
          base_cost -= rd_sig_cost_0;
          rd_sig_cost -= rd_sig_cost_0;
        }
        double cost_zero_cg = base_cost;
        cost_coeffgroup_sig[cgs] = c.lambda * c.price(cg0 + ctx_sig, 1);
        base_cost += cost_coeffgroup_sig[cgs];
        cost_zero_cg += c.lambda * c.price(cg0 + ctx_sig, 0);
        cost_zero_cg += rd_uncoded_dist;
        cost_zero_cg -= rd_coded_level_and_dist;
        cost_zero_cg -= rd_sig_cost;
        if (cost_zero_cg < base_cost) {
          sig_groups &= ~(1ull << cg_blkpos);
          base_cost = cost_zero_cg;
          cost_coeffgroup_sig[cgs] = c.lambda * c.price(cg0 + ctx_sig, 0);
          for (int in_cg = 15; in_cg >= 0; in_cg--) {
            const int scanpos = cgs * 16 + in_cg;
            const u32 blkpos = sc.pos(scanpos);
            if (dest[blkpos]) {
              dest[blkpos] = 0;
              cost_coeff[scanpos] = cost_coeff0[scanpos];
              cost_sig[scanpos] = 0;
            }
          }
        }
      }
    } else {
      sig_groups |= 1ull << cg_blkpos;
    }
  }
  // ---- the last position (rdo.c:903-957), intra block: coded block flag of the transform unit
  double best_cost;
  int best_last_idx_p1 = 0;
  bool found_last = false;
  {
    // luma: qt_cbf_model_luma[!tr_depth]; chroma: qt_cbf_model_chroma[tr_depth] (rdo.c:907-915) -- entries 2..3 of the latter (the blocks of an NxN CU come
    // with tr_depth 2, quant-generic.c:237-238) sit behind the other contexts in the KVZ_HIP_CX_* layout
    const int ctx_cbf = type == 0 ? KVZ_HIP_CX_CBF_LUMA + !tr_depth : (tr_depth < 2 ? KVZ_HIP_CX_CBF_CHROMA + tr_depth : KVZ_HIP_CX_CBF_CHROMA_DEEP + imin(tr_depth, 3) - 2);
    best_cost = block_uncoded_cost + c.lambda * c.price(ctx_cbf, 0);
    base_cost += c.lambda * c.price(ctx_cbf, 1);
  }
  for (int cgs = cg_last_scanpos; cgs >= 0; cgs--) {
    const u32 cg_blkpos = sc.cg(cgs);
    base_cost -= cost_coeffgroup_sig[cgs];
    if ((sig_groups >> cg_blkpos) & 1) {
      for (int in_cg = 15; in_cg >= 0; in_cg--) {
        const int scanpos = cgs * 16 + in_cg;
        if (scanpos > last_scanpos) continue;
        const u32 blkpos = sc.pos(scanpos);
        if (dest[blkpos]) {
          const u32 pos_y = blkpos >> log2w, pos_x = blkpos - (pos_y << log2w);
          const u32 px = scan_mode == 2 ? pos_y : pos_x, py = scan_mode == 2 ? pos_x : pos_y;  // SCAN_VER swaps (rdo.c:934)
          // rdo.c:465-478 get_rate_last
          const int gx = rdoq_group_idx((int)px), gy = rdoq_group_idx((int)py);
          double ui_cost = last_x_bits[gx] + last_y_bits[gy];
          if (gx > 3) ui_cost += (double)((1 << 15) * ((gx - 2) >> 1));
          if (gy > 3) ui_cost += (double)((1 << 15) * ((gy - 2) >> 1));
          const double cost_last = c.lambda * ui_cost;
          const double total = base_cost + cost_last - cost_sig[scanpos];
          if (total < best_cost) {
            best_last_idx_p1 = scanpos + 1;
            best_cost = total;
          }
          if (dest[blkpos] > 1) { found_last = true; break; }
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
    const u32 blkpos = sc.pos(scanpos);
    const i32 level = dest[blkpos];
    dest[blkpos] = (i16)(coef[blkpos] < 0 ? -level : level);
  }
  for (int scanpos = best_last_idx_p1; scanpos <= last_scanpos; scanpos++) dest[sc.pos(scanpos)] = 0;
}

// ---------------------------------------------------------------------------------------------------------------------------------------------------
// The same block by a whole WAVEFRONT (the CTU pass, kvz_ctu.hpp recon_tus).  kvz_rdoq is a chain of decisions, but most of what it computes per coefficient does not
// depend on the chain at all.  What is serial, and stays on one lane in the reference's order:
//   * the (c1, c2, go_rice, c1_idx, c2_idx) state inside a 4x4 group, which only positions that can quantise to a non-zero level move or read;
//   * the double-precision running sums (base_cost, block_uncoded_cost, the group sums): floating-point addition is not associative, so they are added one term at a
//     time in scan order -- an addition per position, not the ~270 dependent instructions per position of the one-lane routine above;
//   * the zero-the-group and best-last-position decisions, which compare those sums.
// What is per-position and goes to sixteen lanes, one 4x4 group at a time (the group's pattern_sig_ctx is known by then): scan position -> block position, level_double,
// max_abs_level, err^2 * temp, the distortion of the two candidate levels, the significance context and lambda times the price of both its bins, the cost of
// coding a zero (coded_cost0 + cost of the zero flag) -- for a position whose max_abs_level is 0, the common case, that IS the position's result.  The per-position
// arrays of the later passes (cost_coeff, cost_coeff0, cost_sig in the caller's scratch) are written and read back sixteen positions at a time; the last-position
// pass gets its per-position rate terms the same way.  Values travel between the sixteen lanes and the chain lane through a small LDS block per wavefront
// (RdoqWaveLds): one wavefront's LDS operations execute in order, so no barrier is involved.
//
// The host simulation runs the same source: a "lane loop" is a plain loop there (KVZ_WAVE_LANES), the chain runs once.
struct RdoqWaveLds {
  double c0v[16], ccv0[16], sig0[16], sig1[16], dhi[16], dlo[16];  // per position of the group in flight: see rdoq_block_wave
  double ccv[16], csv[16];                                           // what the chain decided for it: coded cost and the significance part of it
  i32 max_abs[16];
  i32 lx_bits[32], ly_bits[32];                                      // calc_last_bits (rdo.c:480-509)
  i16 blkpos[16], level[16];
  unsigned long long sig_groups;                                     // sig_coeffgroup_flag, bit = raster index of the group (the chain lane writes, everybody reads)
  int last_scanpos, zeroed, found_last, best_last_idx_p1;
};

#ifdef KVZ_HOSTSIM
#define KVZ_WAVE_LANES(l, n) for (int l = 0; l < (n); l++)
#define KVZ_WAVE_STRIDE(i, n) for (int i = 0; i < (n); i++)
#define KVZ_WAVE_CHAIN() if (true)
#define KVZ_WAVE_ORDER()
#else
#define KVZ_WAVE_LANES(l, n) for (int l = lane, once_ = 1; once_ && l < (n); once_ = 0)
#define KVZ_WAVE_STRIDE(i, n) for (int i = lane; i < (n); i += 64)
#define KVZ_WAVE_CHAIN() if (lane == 0)
// LDS traffic of one wavefront is executed in program order; this only keeps the compiler from moving accesses across the hand-over points
#define KVZ_WAVE_ORDER() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#endif

// cost3: 3 * w * w + 64 doubles of scratch (cost_coeff | cost_sig | cost_coeff0 | cost_coeffgroup_sig).  Device: every lane of the wavefront calls it, converged, with
// wavefront-uniform arguments; lane = its index in the wavefront.  Host: one call (lane 0).
KVZ_DEV void rdoq_block_wave(const RdoqCtx &c, int qp, const i16 *coef, i16 *dest, int log2w, int type /* 0 luma, 2 chroma */, int scan_mode, int tr_depth, const u8 *diag8, double *cost3,
                             RdoqWaveLds *W, int lane)
{
  (void)lane;
  const int width = 1 << log2w, n = width * width;
  const int transform_shift = 15 - 8 - log2w;
  const int qp_scaled = rdoq_scaled_qp(type, qp);
  const i32 q_bits = 14 + qp_scaled / 6 + transform_shift;
  const i32 q = rdoq_quant_scale(qp_scaled % 6);
  double scale = 32768.0;  // scalinglist.c:349-367: err_scale = 2^15 * 2^(-2 transform_shift) / q / q
  for (int i = 0; i < 2 * transform_shift; i++) scale = scale * 0.5;
  for (int i = 0; i > 2 * transform_shift; i--) scale = scale * 2.0;
  const double temp = scale / (double)q / (double)q;
  double *cost_coeff = cost3, *cost_sig = cost3 + n, *cost_coeff0 = cost3 + 2 * n, *cost_cg_sig = cost3 + 3 * n;
  const int num_blk_side = width >> 2, cg_num = n >> 4;
  const RdoqScan sc{ log2w, scan_mode, diag8 };
  const i32 round = 1 << (q_bits - 1);
  // ---- quant-generic.c:379-399 find_last_scanpos: the highest scan position that does not quantise to zero; everything above it is zero in dest
  int my_last = -1;
  KVZ_WAVE_STRIDE(sp, n) {
    const u32 blkpos = sc.pos(sp);
    const i32 ld = imin(iabs((i32)coef[blkpos]) * q, 0x7fffffff - round);
    if (((ld + round) >> q_bits) > 0) my_last = imax(my_last, sp);
  }
#ifndef KVZ_HOSTSIM
  {  // wavefront maximum (values >= -1)
    int x = my_last + 1;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) x = imax(x, __shfl_xor(x, off));
    my_last = x - 1;
  }
#endif
  const int last_scanpos = my_last;
  KVZ_WAVE_STRIDE(sp, n) { if (sp > last_scanpos) dest[sc.pos(sp)] = 0; }
  if (last_scanpos < 0) return;
  const int cg_last_scanpos = last_scanpos >> 4;
  // rdo.c:480-509 calc_last_bits: prefix-sum of the "one more" bins; lanes 0 / 1 walk the x / y contexts
  {
    const int cb = log2w - 2;
    const int off = type ? 0 : (cb * 3 + ((cb + 1) >> 2)), shift = type ? cb : ((cb + 3) >> 2);
    const int bx = (type ? KVZ_HIP_CX_LAST_X_CHROMA : KVZ_HIP_CX_LAST_X_LUMA) + off, by = (type ? KVZ_HIP_CX_LAST_Y_CHROMA : KVZ_HIP_CX_LAST_Y_LUMA) + off;
    KVZ_WAVE_LANES(l, 2) {
      i32 *dst = l == 0 ? W->lx_bits : W->ly_bits;
      const int b0 = l == 0 ? bx : by;
      i32 bits = 0;
      int k;
      for (k = 0; k < rdoq_group_idx(width - 1); k++) {
        dst[k] = bits + c.price(b0 + (k >> shift), 0);
        bits += c.price(b0 + (k >> shift), 1);
      }
      dst[k] = bits;
    }
    KVZ_WAVE_CHAIN() { W->sig_groups = 0; W->found_last = 0; W->best_last_idx_p1 = 0; }
  }
  KVZ_WAVE_ORDER();
  const int cg0 = KVZ_HIP_CX_SIG_CG + type;
  const int sig_base = type ? KVZ_HIP_CX_SIG_CHROMA : KVZ_HIP_CX_SIG_LUMA;
  // the chain's state: meaningful on the chain lane only
  int ctx_set = (last_scanpos > 0 && type == 0) ? 2 : 0, c1 = 1, c2 = 0, go_rice = 0;
  u32 c1_idx = 0, c2_idx = 0;
  double base_cost = 0, block_uncoded_cost = 0;
  for (int cgs = cg_last_scanpos; cgs >= 0; cgs--) {
    const u32 cg_blkpos = sc.cg(cgs), cg_pos_y = cg_blkpos / num_blk_side, cg_pos_x = cg_blkpos - cg_pos_y * num_blk_side;
    const unsigned long long sig_groups_in = W->sig_groups;
    u32 right = 0, lower = 0;  // context.c:339-351 / 315-327
    if ((int)cg_pos_x < num_blk_side - 1) right = (u32)(sig_groups_in >> (cg_pos_y * num_blk_side + cg_pos_x + 1)) & 1;
    if ((int)cg_pos_y < num_blk_side - 1) lower = (u32)(sig_groups_in >> ((cg_pos_y + 1) * num_blk_side + cg_pos_x)) & 1;
    const int pattern_sig_ctx = width == 4 ? -1 : (int)(right + (lower << 1));
    // ---- per position, sixteen lanes
    KVZ_WAVE_LANES(k, 16) {
      const int scanpos = cgs * 16 + k;
      const u32 blkpos = sc.pos(scanpos);
      const i32 ld = imin(iabs((i32)coef[blkpos]) * q, 0x7fffffff - round);
      const i32 max_abs = scanpos > last_scanpos ? -1 : (ld + round) >> q_bits;  // -1: beyond the last position, not part of the block's chain
      const double err = (double)ld;
      const double c0 = err * err * temp;
      const u32 pos_y = blkpos >> log2w, pos_x = blkpos - (pos_y << log2w);
      const int ctx_sig = scanpos == last_scanpos ? 0 : rdoq_sig_ctx_inc(pattern_sig_ctx, scan_mode, (int)pos_x, (int)pos_y, log2w, type);
      const double s0 = c.lambda * c.price(sig_base + ctx_sig, 0), s1 = c.lambda * c.price(sig_base + ctx_sig, 1);
      const double e_hi = (double)(ld - (max_abs * (1 << q_bits))), e_lo = (double)(ld - ((max_abs - 1) * (1 << q_bits)));
      W->blkpos[k] = (i16)blkpos; W->max_abs[k] = max_abs;
      W->c0v[k] = c0; W->sig0[k] = s0; W->sig1[k] = s1; W->ccv0[k] = c0 + s0;
      W->dhi[k] = e_hi * e_hi * temp; W->dlo[k] = e_lo * e_lo * temp;
    }
    KVZ_WAVE_ORDER();
    // ---- the chain: rdo.c:760-840 for this group, then its coded-group decision (rdo.c:842-900)
    KVZ_WAVE_CHAIN() {
      double rd_coded_level_and_dist = 0, rd_uncoded_dist = 0, rd_sig_cost = 0, rd_sig_cost_0 = 0;
      int rd_nnz_before_pos0 = 0;
      bool any_level = false;
      for (int k = 15; k >= 0; k--) {
        const int scanpos = cgs * 16 + k;
        const i32 max_abs = W->max_abs[k];
        if (max_abs < 0) continue;
        const double c0v = W->c0v[k];
        double ccv, csv;
        i32 level = 0;
        block_uncoded_cost += c0v;
        if (max_abs == 0) { ccv = W->ccv0[k]; csv = W->sig0[k]; }  // kvz_get_coded_level: nothing but zero can be coded here (never the last position)
        else {
          // rdo.c:413-459 kvz_get_coded_level on the precomputed pieces
          const bool last = scanpos == last_scanpos;
          const int one_ctx = 4 * ctx_set + c1, abs_ctx = ctx_set + c2;
          double cur_cost_sig = 0;
          csv = 0;
          if (!last && max_abs < 3) { csv = W->sig0[k]; ccv = W->ccv0[k]; }
          else ccv = 1.7e+308;
          if (!last) cur_cost_sig = W->sig1[k];
          const i32 min_abs = max_abs > 1 ? max_abs - 1 : 1;
          for (i32 a = max_abs; a >= min_abs; a--) {
            double cur = (a == max_abs ? W->dhi[k] : W->dlo[k]) + c.lambda * rdoq_ic_rate(c, (u32)a, one_ctx, abs_ctx, go_rice, c1_idx, c2_idx, type);
            cur += cur_cost_sig;
            if (cur < ccv) { level = a; ccv = cur; csv = cur_cost_sig; }
          }
          const i32 base_level = c1_idx < 8 ? (2 + (c2_idx < 1)) : 1;
          if (level >= base_level && level > 3 * (1 << go_rice)) go_rice = imin(go_rice + 1, 4);
          if (level >= 1) c1_idx++;
          if (level > 1) { c1 = 0; c2 += (c2 < 2); c2_idx++; }
          else if (c1 < 3 && c1 > 0 && level) c1++;
        }
        W->ccv[k] = ccv; W->csv[k] = csv; W->level[k] = (i16)level;
        base_cost += ccv;
        if ((scanpos % 16 == 0) && scanpos > 0) {
          c2 = 0; go_rice = 0; c1_idx = 0; c2_idx = 0;
          ctx_set = (scanpos == 16 || type != 0) ? 0 : 2;
          if (c1 == 0) ctx_set++;
          c1 = 1;
        }
        rd_sig_cost += csv;
        if (k == 0) rd_sig_cost_0 = csv;
        if (level) {
          any_level = true;
          rd_coded_level_and_dist += ccv - csv;
          rd_uncoded_dist += c0v;
          if (k != 0) rd_nnz_before_pos0++;
        }
      }
      unsigned long long sg = sig_groups_in;
      if (any_level) sg |= 1ull << cg_blkpos;
      int zeroed = 0;
      double cg_cost = 0;
      if (cgs) {
        const int ctx_sig = (int)(right || lower);
        if (!any_level) {
          cg_cost = c.lambda * c.price(cg0 + ctx_sig, 0);
          base_cost += cg_cost - rd_sig_cost;
        } else if (cgs < cg_last_scanpos) {
          if (rd_nnz_before_pos0 == 0) { base_cost -= rd_sig_cost_0; rd_sig_cost -= rd_sig_cost_0; }
          double cost_zero_cg = base_cost;
          cg_cost = c.lambda * c.price(cg0 + ctx_sig, 1);
          base_cost += cg_cost;
          cost_zero_cg += c.lambda * c.price(cg0 + ctx_sig, 0);
          cost_zero_cg += rd_uncoded_dist;
          cost_zero_cg -= rd_coded_level_and_dist;
          cost_zero_cg -= rd_sig_cost;
          if (cost_zero_cg < base_cost) {
            sg &= ~(1ull << cg_blkpos);
            base_cost = cost_zero_cg;
            cg_cost = c.lambda * c.price(cg0 + ctx_sig, 0);
            zeroed = 1;
          }
        }
      } else sg |= 1ull << cg_blkpos;
      cost_cg_sig[cgs] = cg_cost;  // every group's entry is written here, by the lane that reads it back in the last pass (0 for the first and the last group, rdo.c:729)
      W->sig_groups = sg;
      W->zeroed = zeroed;
    }
    KVZ_WAVE_ORDER();
    // ---- the group's entries of the per-position arrays and its levels, sixteen lanes
    {
      const int zeroed = W->zeroed;
      KVZ_WAVE_LANES(k, 16) {
        const int scanpos = cgs * 16 + k;
        if (W->max_abs[k] >= 0) {
          i32 level = W->level[k];
          double ccv = W->ccv[k], csv = W->csv[k];
          if (zeroed && level) { level = 0; ccv = W->c0v[k]; csv = 0; }  // rdo.c:888-897: the group is cheaper uncoded
          cost_coeff[scanpos] = ccv; cost_sig[scanpos] = csv; cost_coeff0[scanpos] = W->c0v[k];
          dest[W->blkpos[k]] = (i16)level;
        }
      }
    }
    KVZ_WAVE_ORDER();
  }
  // ---- the last position (rdo.c:903-957), intra block: coded block flag of the transform unit
  double best_cost = 0;
  KVZ_WAVE_CHAIN() {
    const int ctx_cbf = type == 0 ? KVZ_HIP_CX_CBF_LUMA + !tr_depth : (tr_depth < 2 ? KVZ_HIP_CX_CBF_CHROMA + tr_depth : KVZ_HIP_CX_CBF_CHROMA_DEEP + imin(tr_depth, 3) - 2);
    best_cost = block_uncoded_cost + c.lambda * c.price(ctx_cbf, 0);
    base_cost += c.lambda * c.price(ctx_cbf, 1);
  }
  const unsigned long long sig_groups = W->sig_groups;
  for (int cgs = cg_last_scanpos; cgs >= 0; cgs--) {
    const u32 cg_blkpos = sc.cg(cgs);
    const bool coded = (sig_groups >> cg_blkpos) & 1;
    if (coded) {
      // per position: the level, the three costs, and for a level lambda times the rate of ending the block there (rdo.c:465-478 get_rate_last)
      KVZ_WAVE_LANES(k, 16) {
        const int scanpos = cgs * 16 + k;
        i32 level = -1;
        if (scanpos <= last_scanpos) {
          const u32 blkpos = sc.pos(scanpos);
          level = dest[blkpos];
          W->ccv[k] = cost_coeff[scanpos]; W->csv[k] = cost_sig[scanpos]; W->c0v[k] = cost_coeff0[scanpos];
          if (level) {
            const u32 pos_y = blkpos >> log2w, pos_x = blkpos - (pos_y << log2w);
            const u32 px = scan_mode == 2 ? pos_y : pos_x, py = scan_mode == 2 ? pos_x : pos_y;  // SCAN_VER swaps (rdo.c:934)
            const int gx = rdoq_group_idx((int)px), gy = rdoq_group_idx((int)py);
            double ui_cost = W->lx_bits[gx] + W->ly_bits[gy];
            if (gx > 3) ui_cost += (double)((1 << 15) * ((gx - 2) >> 1));
            if (gy > 3) ui_cost += (double)((1 << 15) * ((gy - 2) >> 1));
            W->dhi[k] = c.lambda * ui_cost;
          }
        }
        W->max_abs[k] = level;
      }
    }
    KVZ_WAVE_ORDER();
    KVZ_WAVE_CHAIN() {
      base_cost -= cost_cg_sig[cgs];
      if (coded) {
        for (int k = 15; k >= 0; k--) {
          const i32 level = W->max_abs[k];
          if (level < 0) continue;
          if (level) {
            const double total = base_cost + W->dhi[k] - W->csv[k];
            if (total < best_cost) { W->best_last_idx_p1 = cgs * 16 + k + 1; best_cost = total; }
            if (level > 1) { W->found_last = 1; break; }
            base_cost -= W->ccv[k];
            base_cost += W->c0v[k];
          } else base_cost -= W->csv[k];
        }
      }
    }
    KVZ_WAVE_ORDER();
    if (W->found_last) break;
  }
  const int best_last_idx_p1 = W->best_last_idx_p1;
  KVZ_WAVE_STRIDE(sp, last_scanpos + 1) {
    const u32 blkpos = sc.pos(sp);
    if (sp < best_last_idx_p1) { const i32 level = dest[blkpos]; dest[blkpos] = (i16)(coef[blkpos] < 0 ? -level : level); }
    else dest[blkpos] = 0;
  }
  KVZ_WAVE_ORDER();
}

// one item = one block of a batch of equally shaped blocks
struct RdoqOp {
  const Tables *tb; const u8 *ctx; double lambda; int qp; const i16 *coef; i16 *dest; int log2w, type, scan_mode, tr_depth; double *tmp;
  KVZ_DEV void operator()(int item) const
  {
    const int n = 1 << (2 * log2w);
    const RdoqCtx c{ ctx, tb->entropy_bits, lambda };
    rdoq_block(c, qp, coef + (long)item * n, dest + (long)item * n, log2w, type, scan_mode, tr_depth, tb->diag8, tmp + (long)item * 3 * n);
  }
};

}  // namespace kvz
