// kvz_residual.hpp -- the residual syntax of one transform block, kvz_encode_coeff_nxn (strategies/generic/encode_coding_tree-generic.c:40-283 with
// kvz_encode_last_significant_xy encode_coding_tree.c:63-115, kvz_context_get_sig_ctx_inc context.c:366-399, kvz_cabac_write_coeff_remain cabac.c:275-301), written
// against a SINK of bins: `s.ctx(context, value)` for a context-coded bin (contexts in the KVZ_HIP_CX_* numbering of include/kvz_hip_types.h), `s.ep(value, bits)`
// for a run of bypass bins.  Two sinks exist: the entropy coder's record list (kvz_entropy.hpp BinSink: the real mode) and the inter CTU pass's price counter
// (kvz_inter_ctu.hpp PriceSink: the counting mode of get_coeff_cabac_cost, rdo.c:220-263) -- which bins a block has does not depend on what is done with them.
#pragma once
#include "kvz_ops.hpp"
#include "kvz_tables.hpp"
#include "../../include/kvz_hip_types.h"

namespace kvz {

KVZ_DEV int entropy_sig_ctx_inc(int pattern_sig_ctx, int scan_idx, int pos_x, int pos_y, int log2_size, int type)  // context.c:366-399
{
  if (pos_x + pos_y == 0) return 0;
  if (log2_size == 2) { const unsigned long long map = 0x8877886654325410ull; return (int)((map >> (4 * (4 * pos_y + pos_x))) & 15); }  // ctx_ind_map
  const int offset = log2_size == 3 ? (scan_idx == 0 ? 9 : 15) : (type == 0 ? 21 : 12);
  const int xs = pos_x & 3, ys = pos_y & 3;
  int cnt;
  if (pattern_sig_ctx == 0) cnt = xs + ys <= 2 ? (xs + ys == 0 ? 2 : 1) : 0;
  else if (pattern_sig_ctx == 1) cnt = ys <= 1 ? (ys == 0 ? 2 : 1) : 0;
  else if (pattern_sig_ctx == 2) cnt = xs <= 1 ? (xs == 0 ? 2 : 1) : 0;
  else cnt = 2;
  return ((type == 0 && ((pos_x >> 2) + (pos_y >> 2)) > 0) ? 3 : 0) + offset + cnt;
}
KVZ_DEV int entropy_group_idx(int v)  // encoderstate.h:397 g_group_idx
{
  return v < 4 ? v : (v < 6 ? 4 : (v < 8 ? 5 : (v < 12 ? 6 : (v < 16 ? 7 : (v < 24 ? 8 : 9)))));
}

// kvz_encode_coeff_nxn_generic: the residual syntax of one transform block (sign hiding, transform skip, encryption off), cut at the coefficient group so that a caller can
// run it a group at a time (the entropy coder's bin stage keeps the lanes of a wavefront in the same piece of code that way): entropy_tu_begin -- which groups are
// significant, the last position and its syntax -- then entropy_tu_cg once per group, last to first, while t.i >= 0.
struct TuWalk {
  const i16 *coeff; int log2_size, type, scan_mode;
  unsigned long long sig_cg;  // bit cy * nbs + cx
  int i, scan_pos_sig, scan_pos_last, pos_last, c1;
};
KVZ_DEV int entropy_cg_of(const u32 *scan, int i, int log2_size)  // g_sig_last_scan_cg: the scan is group-major
{
  const int width = 1 << log2_size, p = (int)scan[i << 4];
  return (((p >> log2_size) >> 2) << (log2_size - 2)) + ((p & (width - 1)) >> 2);
}
// The scan of a block is group-major with the 4x4 pattern of its type inside every group (kvz_tables.hpp: scan[n] = group origin + pattern[n & 15]): position y * 4 + x of
// scan index i inside a group is nibble i of these constants -- diagonal, horizontal, vertical.  A group's sixteen levels are four 8-byte rows: loaded once, looked up in
// registers (a lane's loads are what the bin stage is bound by: one lane per CTU, 64 cache lines per load instruction).
KVZ_DEV unsigned long long entropy_scan_pattern(int scan_mode) { return scan_mode == 0 ? 0xfbe7ad369c258140ull : (scan_mode == 1 ? 0xfedcba9876543210ull : 0xfb73ea62d951c840ull); }
struct CgRows {
  unsigned long long r0, r1, r2, r3;
  KVZ_DEV void load(const i16 *coeff, int width, int cg_x, int cg_y)
  {
    const i16 *g = coeff + (cg_y * 4) * width + cg_x * 4;
    r0 = *(const unsigned long long *)g; r1 = *(const unsigned long long *)(g + width); r2 = *(const unsigned long long *)(g + 2 * width); r3 = *(const unsigned long long *)(g + 3 * width);
  }
  KVZ_DEV int at(int p) const  // p = y * 4 + x inside the group
  {
    const int y = p >> 2;
    const unsigned long long r = y == 0 ? r0 : (y == 1 ? r1 : (y == 2 ? r2 : r3));
    return (int)(i16)(u16)(r >> (16 * (p & 3)));
  }
};
template <class Sink> KVZ_DEV void entropy_tu_begin(Sink &s, const Tables *tb, TuWalk &t)
{
  const i16 *coeff = t.coeff;
  const int log2_size = t.log2_size, type = t.type, scan_mode = t.scan_mode;
  const int width = 1 << log2_size, nbs = width >> 2;
  const u32 *scan = tb->scan[scan_mode][log2_size - 2];
  unsigned long long sig_cg = 0;
  for (int cy = 0; cy < nbs; cy++)
    for (int cx = 0; cx < nbs; cx++) {
      bool any = false;
      for (int r = 0; r < 4; r++) any |= *(const unsigned long long *)&coeff[(cy * 4 + r) * width + cx * 4] != 0;
      if (any) sig_cg |= 1ull << (cy * nbs + cx);
    }
  int scan_cg_last = nbs * nbs - 1;
  while (!((sig_cg >> entropy_cg_of(scan, scan_cg_last, log2_size)) & 1)) scan_cg_last--;
  int scan_pos_last = scan_cg_last * 16 + 15;
  const unsigned long long pat = entropy_scan_pattern(scan_mode);
  const int last_cg = entropy_cg_of(scan, scan_cg_last, log2_size), last_cg_y = last_cg >> (log2_size - 2), last_cg_x = last_cg & (nbs - 1);  // (nbs = 2^(log2_size - 2): a division by it is ~25 instructions where the compiler cannot see that)
  CgRows rows;
  rows.load(coeff, width, last_cg_x, last_cg_y);
  while (!rows.at((int)((pat >> (4 * (scan_pos_last & 15))) & 15))) scan_pos_last--;
  const int p_last = (int)((pat >> (4 * (scan_pos_last & 15))) & 15);
  const int pos_last = (last_cg_y * 4 + (p_last >> 2)) * width + last_cg_x * 4 + (p_last & 3);
  {  // kvz_encode_last_significant_xy (encode_coding_tree.c:63-115)
    int lx = pos_last & (width - 1), ly = pos_last >> log2_size;
    const int index = log2_size - 2;
    const int ctx_offset = type ? 0 : (index * 3 + (index + 1) / 4), shift = type ? index : (index + 3) / 4;
    const int base_x = type ? KVZ_HIP_CX_LAST_X_CHROMA : KVZ_HIP_CX_LAST_X_LUMA, base_y = type ? KVZ_HIP_CX_LAST_Y_CHROMA : KVZ_HIP_CX_LAST_Y_LUMA;
    if (scan_mode == 2) { const int tmp = lx; lx = ly; ly = tmp; }
    const int gx = entropy_group_idx(lx), gy = entropy_group_idx(ly), gmax = entropy_group_idx(width - 1);
    for (int i = 0; i < gx; i++) s.ctx(base_x + ctx_offset + (i >> shift), 1);
    if (gx < gmax) s.ctx(base_x + ctx_offset + (gx >> shift), 0);
    for (int i = 0; i < gy; i++) s.ctx(base_y + ctx_offset + (i >> shift), 1);
    if (gy < gmax) s.ctx(base_y + ctx_offset + (gy >> shift), 0);
    const int min_in_group[10] = { 0, 1, 2, 3, 4, 6, 8, 12, 16, 24 };
    if (gx > 3) s.ep((u32)(lx - min_in_group[gx]), (gx - 2) / 2);
    if (gy > 3) s.ep((u32)(ly - min_in_group[gy]), (gy - 2) / 2);
  }
  t.sig_cg = sig_cg; t.i = scan_cg_last; t.scan_pos_sig = t.scan_pos_last = scan_pos_last; t.pos_last = pos_last; t.c1 = 1;
}
template <class Sink> KVZ_DEV void entropy_tu_cg(Sink &s, const Tables *tb, TuWalk &t)
{
  const i16 *coeff = t.coeff;
  const int log2_size = t.log2_size, type = t.type, scan_mode = t.scan_mode, i = t.i;
  const int width = 1 << log2_size, nbs = width >> 2;
  const u32 *scan = tb->scan[scan_mode][log2_size - 2];
  const int base_sig = type == 0 ? KVZ_HIP_CX_SIG_LUMA : KVZ_HIP_CX_SIG_CHROMA;
  unsigned long long sig_cg = t.sig_cg;
  int scan_pos_sig = t.scan_pos_sig, c1 = t.c1;
  const int scan_cg_last = t.scan_pos_last >> 4, scan_pos_last = t.scan_pos_last;
  {
    const int sub_pos = i << 4, cg_blk_pos = entropy_cg_of(scan, i, log2_size), cg_pos_y = cg_blk_pos >> (log2_size - 2), cg_pos_x = cg_blk_pos & (nbs - 1);
    int abs_coeff[16], num_non_zero = 0;
    u32 coeff_signs = 0, go_rice = 0;
    const unsigned long long pat = entropy_scan_pattern(scan_mode);
    CgRows rows{ 0, 0, 0, 0 };  // (an insignificant group is all zero: the first group's flag is inferred, its levels are still walked)
    if ((sig_cg >> cg_blk_pos) & 1) rows.load(coeff, width, cg_pos_x, cg_pos_y);
    if (scan_pos_sig == scan_pos_last) { const int v = rows.at((int)((pat >> (4 * (scan_pos_last & 15))) & 15)); abs_coeff[0] = iabs(v); coeff_signs = v < 0; num_non_zero = 1; scan_pos_sig--; }
    const int right = cg_pos_x < nbs - 1 && ((sig_cg >> (cg_pos_y * nbs + cg_pos_x + 1)) & 1);
    const int lower = cg_pos_y < nbs - 1 && ((sig_cg >> ((cg_pos_y + 1) * nbs + cg_pos_x)) & 1);
    if (i == scan_cg_last || i == 0) sig_cg |= 1ull << cg_blk_pos;
    else s.ctx(KVZ_HIP_CX_SIG_CG + type + (right || lower), (int)((sig_cg >> cg_blk_pos) & 1));  // coded_sub_block_flag (context.c:315-327)
    if ((sig_cg >> cg_blk_pos) & 1) {
      const int pattern = width == 4 ? -1 : right + (lower << 1);  // context.c:339-351
      for (; scan_pos_sig >= sub_pos; scan_pos_sig--) {
        const int p = (int)((pat >> (4 * (scan_pos_sig & 15))) & 15), pos_y = cg_pos_y * 4 + (p >> 2), pos_x = cg_pos_x * 4 + (p & 3), v = rows.at(p);
        if (scan_pos_sig > sub_pos || i == 0 || num_non_zero) s.ctx(base_sig + entropy_sig_ctx_inc(pattern, scan_mode, pos_x, pos_y, log2_size, type), v != 0);
        if (v) { abs_coeff[num_non_zero++] = iabs(v); coeff_signs = 2 * coeff_signs + (v < 0); }
      }
    } else scan_pos_sig = sub_pos - 1;
    if (num_non_zero > 0) {
      int ctx_set = (i > 0 && type == 0) ? 2 : 0;
      if (c1 == 0) ctx_set++;
      c1 = 1;
      const int base_one = (type == 0 ? KVZ_HIP_CX_ONE_LUMA : KVZ_HIP_CX_ONE_CHROMA) + 4 * ctx_set, num_c1 = num_non_zero < 8 ? num_non_zero : 8;
      int first_c2 = -1;
      for (int idx = 0; idx < num_c1; idx++) {
        const int symbol = abs_coeff[idx] > 1;
        s.ctx(base_one + c1, symbol);
        if (symbol) { c1 = 0; if (first_c2 == -1) first_c2 = idx; }
        else if (c1 < 3 && c1 > 0) c1++;
      }
      if (c1 == 0 && first_c2 != -1) s.ctx((type == 0 ? KVZ_HIP_CX_ABS_LUMA : KVZ_HIP_CX_ABS_CHROMA) + ctx_set, abs_coeff[first_c2] > 2);
      s.ep(coeff_signs, num_non_zero);
      if (c1 == 0 || num_non_zero > 8) {
        int first_coeff2 = 1;
        for (int idx = 0; idx < num_non_zero; idx++) {
          const int base_level = idx < 8 ? 2 + first_coeff2 : 1;
          if (abs_coeff[idx] >= base_level) {  // kvz_cabac_write_coeff_remain (cabac.c:275-301)
            int code_number = abs_coeff[idx] - base_level;
            if (code_number < (3 << go_rice)) {
              const u32 length = (u32)code_number >> go_rice;
              s.ep((1u << (length + 1)) - 2, (int)length + 1);
              s.ep((u32)code_number & ((1u << go_rice) - 1), (int)go_rice);
            } else {
              u32 length = go_rice;
              code_number -= 3 << go_rice;
              while (code_number >= (1 << length)) { code_number -= 1 << length; ++length; }
              s.ep((1u << (3 + length + 1 - go_rice)) - 2, (int)(3 + length + 1 - go_rice));
              s.ep((u32)code_number, (int)length);
            }
            if (abs_coeff[idx] > 3 * (1 << go_rice)) go_rice = go_rice + 1 < 4 ? go_rice + 1 : 4;
          }
          if (abs_coeff[idx] >= 2) first_coeff2 = 0;
        }
      }
    }
  }
  t.sig_cg = sig_cg; t.scan_pos_sig = scan_pos_sig; t.c1 = c1; t.i = i - 1;
}
template <class Sink> KVZ_DEV void entropy_coeff_nxn(Sink &s, const Tables *tb, const i16 *coeff, int log2_size, int type, int scan_mode)
{
  TuWalk t;
  t.coeff = coeff; t.log2_size = log2_size; t.type = type; t.scan_mode = scan_mode;
  entropy_tu_begin(s, tb, t);
  while (t.i >= 0) entropy_tu_cg(s, tb, t);
}

}  // namespace kvz
