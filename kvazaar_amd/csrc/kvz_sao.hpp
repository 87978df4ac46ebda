// kvz_sao.hpp -- SAO parameter decision on the device (kvz_sao_search_lcu, sao.c:671, for every LCU of every picture of a batch).
//
// kvazaar decides an LCU's SAO parameters right after deblocking THAT LCU (encoderstate.c:669-682), i.e. on a picture whose right and
// lower neighbours have not been deblocked yet: of the LCU's own samples the right edge (filter.c:759-790 "what is not filtered"), the
// rightmost 4 samples of every horizontal edge and the bottom edge are still to come.  With R = the reconstruction before deblocking,
// V = after all vertical edges, D = after all horizontal edges too (the picture-level order of kvz_dev.hpp's deblocking kernels), the LCU
// sees, per plane with N = 64 (luma) / 32 (chroma) and the horizontal-edge deferral of 4 samples in either plane:
//     column >= N - 4, LCU not in the last LCU column:  R   (no vertical edge reaches columns N-5 .. N-4; N-3 .. N-1 only the right edge)
//     else row >= N - 3 (luma) / N - 1 (chroma), LCU not in the last LCU row:  V   (only the bottom edge writes those rows)
//     else:  D
// (oracle/kvz_oracle_sao.c gets the same picture by literally deblocking LCU by LCU.)
//
// Three stages:
//   1. statistics (parallel, one workgroup per LCU and plane): per edge class and category the sum of (orig - rec) and the count over
//      the block's interior (sao-generic.c:50-81), per band the same over the whole block (sao.c:275-296)        -> SaoStats
//   2. candidates (parallel, folded into the tail of 1): everything of sao_search_edge_sao / sao_search_band_sao that does not depend on
//      the CABAC contexts: per class the offsets, their distortion change and the context-free part of the mode bits; the band
//      offsets / position / distortions (sao.c:214-268, 364-478)                                                   -> SaoCand
//   3. chain (one lane per picture, raster order): mode bits on the two SAO contexts as they stand at the start of the LCU
//      (CABAC_FBITS_UPDATE on state->search_cabac with update == 0, sao.c:52-177), the choice among edge / band / none
//      (sao.c:531-548), merge costs from the neighbours' parameters applied to this LCU's statistics (sao.c:551-584 -- the per-sample
//      distortion functions sao_shared_generics.h:52-130 are sums the statistics already hold), the merge decision (sao.c:712-735) and
//      the context updates of the LCU's SAO syntax (encoderstate.c:467-552) with the WPP hand-off (encoderstate.c:763-771).
// The sums of mode bits are exact in double in any order (a few binary32 table values and small integers), so splitting them into a
// context part and a context-free part changes nothing; every product / rounding is done as the reference does it
// ((int)((double)(float)bits * lambda + 0.5), -ffp-contract=off).
#pragma once
#include "kvz_ops.hpp"

namespace kvz {

#define KVZ_SAO_ABS_MAX 7  // SAO_ABS_OFFSET_MAX at 8 bits (sao.h)

struct SaoStats {        // one plane of one LCU
  i32 edge_sum[4][5], edge_cnt[4][5];  // [class][category]
  i32 band_sum[32], band_cnt[32];
};

struct SaoCand {         // context-free part of the search for one plane of one LCU
  int8_t edge_off[4][4];  // [class][category 1..4]
  i32 edge_dd[4];         // sum over categories of cnt o^2 - 2 o sum
  i32 edge_bits[4];       // sum over categories of the offset's bins (sao.c:113-123), an integer
  int8_t band_off[4];
  i32 band_pos;
  i32 band_dist;          // calc_sao_band_offsets' return value (may be INT_MAX-based garbage exactly as in the reference)
  i32 band_dd;            // sao_band_ddistortion of (band_pos, band_off)
  i32 band_bits;          // sum over the 4 offsets of their bins (sao.c:158-170)
};

// the sample of plane `color` (0 Y, 1 U, 2 V) at block position (x, y) of LCU (lx, ly) as kvazaar's SAO search sees it (see above)
struct SaoView {
  const u8 *R, *V, *D;   // the plane's pictures
  int stride, n, x0, y0, last_col, last_row, v_rows;
  KVZ_DEV int at(int x, int y) const
  {
    const long p = (long)(y0 + y) * stride + x0 + x;
    if (x >= n - 4 && !last_col) return R[p];
    if (y >= n - v_rows && !last_row) return V[p];
    return D[p];
  }
};

KVZ_DEV int sao_offset_of(int sum, int cnt)
{
  int offset = 0;
  if (cnt != 0) {
    offset = (sum + (cnt >> 1)) / cnt;
    offset = iclip(-KVZ_SAO_ABS_MAX, KVZ_SAO_ABS_MAX, offset);
  }
  return offset;
}

// stage 2 for one plane: st -> cand (one lane; ~600 integer operations, dominated by the 32 bands x <= 7 trial offsets of sao.c:224-250)
KVZ_DEV void sao_candidates(const SaoStats &st, SaoCand &c)
{
  for (int ec = 0; ec < 4; ec++) {  // sao.c:381-425
    i32 dd = 0, bits = 0;
    for (int cat = 1; cat <= 4; cat++) {
      int offset = sao_offset_of(st.edge_sum[ec][cat], st.edge_cnt[ec][cat]);
      if (cat <= 2 && offset < 0) offset = 0;  // sharpening offsets cannot be coded
      if (cat >= 3 && offset > 0) offset = 0;
      c.edge_off[ec][cat - 1] = (int8_t)offset;
      dd += st.edge_cnt[ec][cat] * offset * offset - 2 * offset * st.edge_sum[ec][cat];
      const int a = iabs(offset);
      bits += (a == 0 || a == KVZ_SAO_ABS_MAX) ? a + 1 : a + 2;
    }
    c.edge_dd[ec] = dd;
    c.edge_bits[ec] = bits;
  }
  // sao.c:214-268 calc_sao_band_offsets
  i32 dist[32];
  int8_t toff[32];
  for (int band = 0; band < 32; band++) {
    int offset = sao_offset_of(st.band_sum[band], st.band_cnt[band]);
    dist[band] = offset == 0 ? 0 : 0x7fffffff;
    toff[band] = 0;
    while (offset != 0) {  // best_dist stays INT_MAX inside the loop in the reference: every trial whose distortion is below INT_MAX overwrites, the last one wins
      const i32 temp = st.band_cnt[band] * offset * offset - 2 * offset * st.band_sum[band];
      if (temp < 0x7fffffff) { dist[band] = temp; toff[band] = (int8_t)offset; }
      offset += offset > 0 ? -1 : 1;
    }
  }
  i32 best = 0x7fffffff;
  int best_pos = 0;
  for (int band = 0; band < 28; band++) {
    const i32 temp = (i32)((u32)dist[band] + (u32)dist[band + 1] + (u32)dist[band + 2] + (u32)dist[band + 3]);  // wraps like the compiled reference
    if (temp < best) { best = temp; best_pos = band; }
  }
  c.band_pos = best_pos;
  c.band_dist = best;
  i32 dd = 0, bits = 0;
  for (int k = 0; k < 4; k++) {
    const int o = toff[best_pos + k];
    c.band_off[k] = (int8_t)o;
    dd += st.band_cnt[best_pos + k] * o * o - 2 * o * st.band_sum[best_pos + k];
    const int a = iabs(o);
    bits += a == 0 ? a + 1 : (a == KVZ_SAO_ABS_MAX ? a + 2 : a + 3);
  }
  c.band_dd = dd;
  c.band_bits = bits;
}

// ---- stage 3 ----
// packed parameter record of one plane of one LCU (the format dev_sao_kernel reads): type | class << 8 | band position << 16 | offsets[0..4] << 24..
typedef unsigned long long SaoRec;
KVZ_DEV int rec_type(SaoRec r) { return (int)(r & 0xff); }
KVZ_DEV int rec_class(SaoRec r) { return (int)((r >> 8) & 0xff); }
KVZ_DEV int rec_band(SaoRec r) { return (int)((r >> 16) & 0xff); }
KVZ_DEV int rec_off(SaoRec r, int k) { return (int)(int8_t)(r >> (24 + 8 * k)); }
KVZ_DEV SaoRec rec_make(int type, int eo_class, int band_pos, const int8_t off[4] /* categories 1..4 / bands 0..3 */)
{
  SaoRec v = (SaoRec)(type & 0xff) | ((SaoRec)(eo_class & 0xff) << 8) | ((SaoRec)(band_pos & 0xff) << 16);
  for (int k = 0; k < 4; k++) v |= (SaoRec)(u8)off[k] << (32 + 8 * k);  // offsets[0] (category 0) stays 0
  return v;
}

// distortion change of applying record r to the plane whose statistics are st (= kvz_sao_edge_ddistortion / kvz_sao_band_ddistortion)
KVZ_DEV i32 sao_apply_dd(const SaoStats &st, SaoRec r)
{
  i32 dd = 0;
  if (rec_type(r) == 2) {
    const int ec = rec_class(r);
    for (int cat = 1; cat <= 4; cat++) { const int o = rec_off(r, cat); dd += st.edge_cnt[ec][cat] * o * o - 2 * o * st.edge_sum[ec][cat]; }
  } else if (rec_type(r) == 1) {
    const int bp = rec_band(r);
    for (int k = 0; k < 4; k++) {
      const int o = rec_off(r, k + 1), b = bp + k;
      if (b < 32) dd += st.band_cnt[b] * o * o - 2 * o * st.band_sum[b];
    }
  }
  return dd;
}

struct SaoCabac { u8 merge, type; };  // uc_state of sao_merge_flag_model / sao_type_idx_model

KVZ_DEV i32 sao_rate(double bits, double lambda) { const float f = (float)bits; return (i32)((double)f * lambda + 0.5); }

// sao_search_best_mode (sao.c:491-586) for the luma plane (n = 1) or the two chroma planes (n = 2) of an LCU.  st / cand: the planes'
// records; top / left: the neighbours' packed records per plane or nullptr.  out[n]: the chosen records; merge_cost[3]: this / left / up.
KVZ_DEV void sao_best_mode(const float *fb, double lambda, SaoCabac cb, const SaoStats *st, const SaoCand *cand, int n, const SaoRec *top, const SaoRec *left, SaoRec *out,
                           i32 merge_cost[3])
{
  double head = 0.0;  // the merge flags every candidate starts with (sao.c:58-66, 98-106, 140-148)
  if (left) head += fb[cb.merge ^ 0];
  if (top) head += fb[cb.merge ^ 0];
  const double type1 = head + fb[cb.type ^ 1];
  // edge: the class with the least distortion change + rate (strictly better wins; sao.c:381-440), then sao.c:510-523
  i32 edge_best = 0x7fffffff;
  int edge_class = 0;
  for (int ec = 0; ec < 4; ec++) {
    i32 dd = 0, bits = 0;
    for (int p = 0; p < n; p++) { dd += cand[p].edge_dd[ec]; bits += cand[p].edge_bits[ec]; }
    dd += sao_rate(type1 + 1.0 + (double)bits + 2.0, lambda);
    if (dd < edge_best) { edge_best = dd; edge_class = ec; }
  }
  const i32 edge_dd = edge_best;  // recomputed in the reference from the same offsets: identical by construction
  // band: sao.c:443-478, 525-540
  i32 band_search = 0, band_bits = 0, band_apply = 0;
  for (int p = 0; p < n; p++) { band_search = (i32)((u32)band_search + (u32)cand[p].band_dist); band_bits += cand[p].band_bits; band_apply += cand[p].band_dd; }
  const i32 band_rate = sao_rate(type1 + 1.0 + (double)band_bits + 5.0 * n, lambda);
  band_search = (i32)((u32)band_search + (u32)band_rate);
  // sao.c:472: the searched offsets are only taken when band_search < MAX_INT, otherwise offsets / positions of an all-zero record
  const bool band_valid = band_search < 0x7fffffff;
  i32 band_dd;
  if (band_valid) band_dd = band_rate + band_apply;
  else {  // offsets stay 0 (sao_search_best_mode's initialisation): rate of four zero offsets per plane, no distortion change
    band_dd = sao_rate(type1 + 1.0 + 4.0 * n + 5.0 * n, lambda);
  }
  int type;
  i32 cost;
  if (edge_dd <= band_dd) { type = 2; cost = edge_dd; } else { type = 1; cost = band_dd; }
  merge_cost[0] = cost;
  const i32 nothing = sao_rate(head + fb[cb.type ^ 0], lambda);
  const bool none = cost >= nothing;
  if (none) merge_cost[0] = nothing;
  for (int p = 0; p < n; p++) {
    const int8_t zero[4] = { 0, 0, 0, 0 };
    // sao_out keeps the searched fields when the type becomes NONE (sao.c:545-548 only changes the type): neighbours that merge with it
    // copy the record, but a NONE record is never applied and its other fields never read -- so only the type is kept here
    if (none) out[p] = rec_make(0, 0, 0, zero);
    else if (type == 2) out[p] = rec_make(2, edge_class, 0, cand[p].edge_off[edge_class]);
    else out[p] = rec_make(1, 0, band_valid ? cand[p].band_pos : 0, band_valid ? cand[p].band_off : zero);
  }
  // merge candidates: 1 = left, 2 = up (sao.c:551-584)
  for (int i = 0; i < 2; i++) {
    const SaoRec *mc = i == 0 ? left : top;
    if (!mc) continue;
    double bits = fb[cb.merge ^ (i == 0 ? 1 : 0)];            // sao.c:74-87: left: one bin (1); up: two bins (0, 1)
    if (i == 1) bits += fb[cb.merge ^ 1];
    i32 dd = sao_rate(bits, lambda);
    for (int p = 0; p < n; p++) dd += sao_apply_dd(st[p], mc[p]);
    merge_cost[i + 1] = dd;
  }
}

KVZ_DEV void sao_code_bin(const u8 *next_mps, const u8 *next_lps, u8 &st, int bin) { st = (bin != (st & 1)) ? next_lps[st] : next_mps[st]; }

// One picture, raster order (the result does not depend on the order LCUs are visited in, only on the context hand-off rule).
// recs: [lcu][3] packed records (out), merge: [lcu] 0 none / 1 left / 2 up (out).
KVZ_DEV void sao_chain_picture(const float *fb, const u8 *next_mps, const u8 *next_lps, double lambda, u8 init_merge, u8 init_type, int no_wpp, int wl, int hl,
                               const SaoStats *st /* [lcu][3] */, const SaoCand *cand /* [lcu][3] */, SaoRec *recs, u8 *merge)
{
  SaoCabac cab = { init_merge, init_type }, next_row = cab;
  for (int ly = 0; ly < hl; ly++) {
    if (!no_wpp) { if (ly == 0) { cab.merge = init_merge; cab.type = init_type; } else cab = next_row; }
    for (int lx = 0; lx < wl; lx++) {
      const int i = ly * wl + lx;
      const SaoRec *top = ly ? &recs[(i - wl) * 3] : nullptr, *left = lx ? &recs[(i - 1) * 3] : nullptr;
      i32 mc_l[3] = { 0x7fffffff, 0, 0 }, mc_c[3] = { 0x7fffffff, 0, 0 };
      SaoRec out[3];
      sao_best_mode(fb, lambda, cab, &st[i * 3], &cand[i * 3], 1, top, left, &out[0], mc_l);
      sao_best_mode(fb, lambda, cab, &st[i * 3 + 1], &cand[i * 3 + 1], 2, top ? top + 1 : nullptr, left ? left + 1 : nullptr, &out[1], mc_c);
      int m = 0;  // sao.c:712-735
      const i32 own = (i32)((u32)mc_l[0] + (u32)mc_c[0]), c_left = (i32)((u32)mc_l[1] + (u32)mc_c[1]), c_up = (i32)((u32)mc_l[2] + (u32)mc_c[2]);
      if (top && c_up <= own) m = 2;
      if (left && c_left <= own && (m != 2 || c_left < c_up)) m = 1;
      for (int p = 0; p < 3; p++) recs[i * 3 + p] = m == 2 ? top[p] : (m == 1 ? left[p] : out[p]);
      merge[i] = (u8)m;
      if (lx > 0) sao_code_bin(next_mps, next_lps, cab.merge, m == 1);
      if (ly > 0 && m != 1) sao_code_bin(next_mps, next_lps, cab.merge, m == 2);
      if (!m) {
        sao_code_bin(next_mps, next_lps, cab.type, rec_type(recs[i * 3]) != 0);
        sao_code_bin(next_mps, next_lps, cab.type, rec_type(recs[i * 3 + 1]) != 0);
      }
      if (lx == 1) next_row = cab;
    }
  }
}

}  // namespace kvz
