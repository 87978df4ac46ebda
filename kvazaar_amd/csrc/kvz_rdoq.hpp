// kvz_rdoq.hpp -- rate-distortion optimised quantisation of one transform block (kvz_rdoq, rdo.c:661-1000, HM's RDOQ): what
// kvz_quantize_residual runs instead of kvz_quant when --rdoq is on (quant-generic.c:234-244; presets medium and slower).  Sign hiding off
// (its table bookkeeping, rdo.c:780-795, 990-992, is not carried), flat scaling lists, 8 bit.
//
// Per block, serial by construction (every decision feeds the context selection of the next coefficient): levels are chosen coefficient by
// coefficient from the last significant scan position down (kvz_get_coded_level: distortion err^2 * err_scale + lambda * rate, rates from
// the Q15 entropy table on the CALLER's context states -- state->cabac, not the search copy), then whole coefficient groups are tested
// against zeroing, then the best last position is chosen.  All costs are doubles combined in the reference's order (-ffp-contract=off).
// ONE routine, wavefront-cooperative (rdoq_block_wave below): the CTU pass calls it from its quantisation stage, the per-call entry points (kvz_hip_rdoq,
// kvz_hip_rdoq_blocks, kvz_hip_quantize_residual_rdoq) run it one wavefront per block (RdoqOp).  Rounds 2-3 also carried a one-lane transcription of rdo.c:661-1000 for the
// per-call path; it is gone.
#pragma once
#include "kvz_ops.hpp"

namespace kvz {

// The coefficient scans by arithmetic (HEVC scans are hierarchical: 4x4 groups in group order, sixteen positions inside a group): no table in memory on the
// chain from one coefficient to the next.  diag8: the up-right diagonal order of an 8x8 grid (Tables::diag8), the group order of a 32x32 block.
template <class PtrU8> struct RdoqScanT {
  int log2w, mode;
  PtrU8 diag8;
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
using RdoqScan = RdoqScanT<const u8 *>;

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

// ---------------------------------------------------------------------------------------------------------------------------------------------------
// The block by a whole WAVEFRONT (the CTU pass, kvz_ctu.hpp recon_tus; RdoqOp at the end of this file), one 4x4 coefficient group at a time.
//
// kvz_rdoq is a chain of decisions -- the level chosen for a coefficient moves the (c1, go_rice, number of levels) state that prices the next one -- but the chain
// carries very little.  Inside a group the rate of a candidate level depends on the state only through thirteen classes (kvz_get_ic_rate, rdo.c:345-392):
//     0..2   no level above 1 yet, c1 = 1, 2, 3                     (greater-1 context c1, the greater-2 flag still to come, escape codes with rice parameter 0)
//     3..7   a level above 1 has been seen (c1 = 0), go_rice 0..4
//     8..12  eight levels coded already (no context-coded flags left), go_rice 0..4
// and nothing else a position's decision reads depends on the positions before it.  So per group:
//   1. all 64 lanes -- lane = (position k = lane & 15, replica j = lane >> 4), every per-position quantity computed on all four replicas -- take the position's
//      coefficient, its distortions and its significance prices (the group's neighbour pattern is known by now);
//   2. replica j decides the position's level for the classes j and j + 4 (the classes 8..12 only when a group reaches eight levels): two bits per class;
//   3. the chain itself is a scalar walk over the positions that may code a level: look the decision of (position, class) up with v_readlane, move the state -- some
//      twenty scalar instructions a position, where the one-lane transcription of rdo.c:760-840 had several hundred dependent vector instructions;
//   4. every lane evaluates its position once more for the class the walk met it in, now with its costs (kvz_get_coded_level, rdo.c:413-459, exactly);
//   5. the five ordered double-precision sums of the group (block_uncoded_cost, base_cost, rd_sig_cost, rd_coded_level_and_dist, rd_uncoded_dist) are sixteen
//      additions each, in scan order as the reference adds them (floating-point addition is not associative): the terms of sum r lie along row r of a register, the
//      sum on lane 15 of the row, the register rotated by DPP row_ror:1 between additions -- all four rows at once, the fifth sum in a second register;
//   6. the zero-the-group decision (rdo.c:842-900) on wavefront-uniform values.
// Kept per group for the last pass (rdo.c:903-957: the best last position, which usually stops at the first level above 1): the classes its positions were decided
// in (4 bits each) and its context set, on the lane of the group's index; that pass recomputes the costs it meets from them.
//
// The host simulation runs the same source: lane-held storage is an array there (WaveArr), a "lane loop" a plain loop, uniform code runs once.
#ifdef KVZ_HOSTSIM
#define KVZ_LDS_PTR(T) T *
template <class T> struct WaveArr { T v[64]; };
#define KVZ_WAVE_LANES(l, n) for (int l = 0; l < (n); l++)
#define KVZ_WAVE_STRIDE(i, n) for (int i = 0; i < (n); i++)
#define KVZ_WA_SET(a, l, val) ((a).v[l] = (val))   /* lane l stores into its own slot (inside a lane loop) */
#define KVZ_WA_OWN(a, l) ((a).v[l])                 /* lane l reads its own slot */
#define KVZ_WA_PUT(a, idx, val) ((a).v[idx] = (val)) /* uniform: slot idx := val */
#define KVZ_WA_GET(a, idx) ((a).v[idx])              /* uniform read of slot idx */
#define KVZ_WA_AT(a, l, idx) ((a).v[idx])            /* lane l reads slot idx (a per-lane index) */
#define KVZ_UNI_INT(x) (x)
#else
// A pointer that is KNOWN to point into LDS: loads through it are ds_read, not flat loads (which also wait on the vector-memory counter -- behind every store in flight)
#define KVZ_LDS_PTR(T) __attribute__((address_space(3))) T *
template <class T> struct WaveArr { T v; };
#define KVZ_WAVE_LANES(l, n) for (int l = lane, once_ = 1; once_ && l < (n); once_ = 0)
#define KVZ_WAVE_STRIDE(i, n) for (int i = lane; i < (n); i += 64)
#define KVZ_WA_SET(a, l, val) ((a).v = (val))
#define KVZ_WA_OWN(a, l) ((a).v)
#define KVZ_WA_PUT(a, idx, val) ((a).v = (lane == (idx)) ? (val) : (a).v)  /* the value is uniform: a select per lane does what v_writelane would */
#define KVZ_WA_GET(a, idx) wave_readlane((a).v, (idx))
#define KVZ_WA_AT(a, l, idx) __shfl((a).v, (idx))
#define KVZ_UNI_INT(x) __builtin_amdgcn_readfirstlane((int)(x))  /* a value all lanes agree on, as a scalar: branches on it are scalar branches */
KVZ_DEV int wave_readlane(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
KVZ_DEV double wave_readlane(double v, int l)
{
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
#endif

// ONE copy in the kernel: recon_tus() is inlined at a dozen places of the CTU program, and a dozen copies of this routine made the kernel ~0.9 MB of code -- wavefronts
// of the six workgroups on a CU, each somewhere else in it, lived on instruction-cache misses.  Arguments by value (wavefront-uniform scalars and pointers).
#ifdef KVZ_HOSTSIM
#define KVZ_RDOQ_WAVE_FN inline
#else
#define KVZ_RDOQ_WAVE_FN __device__ __attribute__((noinline))
#endif

// Step 5 above.  Device: accA holds block_uncoded_cost / base_cost / rd_sig_cost / rd_coded_level_and_dist on lanes 15 / 31 / 47 / 63, accB rd_uncoded_dist on lane 15.
// Host: five scalars, the same additions in the same order.  A position outside the block's chain (beyond the last position) contributes +0.0 terms, which
// leave a sum as it is.
#ifdef KVZ_HOSTSIM
struct RdoqSums5 {
  double uncoded = 0, base = 0, sig = 0, coded = 0, udist = 0;
  void group(const WaveArr<double> &c0v, const WaveArr<double> &ccv, const WaveArr<double> &csv, const WaveArr<i32> &level, int)
  {
    sig = 0; coded = 0; udist = 0;
    for (int k = 15; k >= 0; k--) {
      const bool nz = level.v[k] > 0;
      uncoded += c0v.v[k]; base += ccv.v[k]; sig += csv.v[k];
      coded += nz ? ccv.v[k] - csv.v[k] : 0.0;
      udist += nz ? c0v.v[k] : 0.0;
    }
  }
  double get_uncoded() const { return uncoded; }
  double get_base() const { return base; }
  double get_sig() const { return sig; }
  double get_coded() const { return coded; }
  double get_udist() const { return udist; }
  void set_base(double v, int) { base = v; }
};
// ... and the one of the last pass: base_cost minus the zero flag's cost of every zero position on the way down
struct RdoqSums1 {
  double base = 0;
  const WaveArr<double> *sigc = nullptr;
  const WaveArr<i32> *lvl = nullptr;
  void begin_group(const WaveArr<double> &a, const WaveArr<i32> &l, int) { sigc = &a; lvl = &l; }
  int next_level(int k) const { while (k >= 0 && !(lvl->v[k] > 0)) k--; return k; }
  void zero_step(int L) { if (lvl->v[L] == 0) base -= sigc->v[L]; }
  void add(double v) { base += v; }
  void rotate() {}
  double get_base() const { return base; }
};
#else
KVZ_DEV double dpp_row_ror1(double v)
{
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x121 /* row_ror:1 */, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x121, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
struct RdoqSums5 {
  double accA = 0, accB = 0;
  KVZ_DEV void group(const WaveArr<double> &c0v, const WaveArr<double> &ccv, const WaveArr<double> &csv, const WaveArr<i32> &level, int lane)
  {
    const int row = lane >> 4;
    const bool nz = level.v > 0;
    double tA = row == 0 ? c0v.v : (row == 1 ? ccv.v : (row == 2 ? csv.v : (nz ? ccv.v - csv.v : 0.0)));
    double tB = (row == 0 && nz) ? c0v.v : 0.0;
    accA = row >= 2 ? 0.0 : accA;  // the group's own sums start over
    accB = 0.0;
#pragma unroll
    for (int i = 0; i < 16; i++) {
      accA += tA; accB += tB;
      tA = dpp_row_ror1(tA); tB = dpp_row_ror1(tB);
    }
  }
  KVZ_DEV double get_uncoded() const { return wave_readlane(accA, 15); }
  KVZ_DEV double get_base() const { return wave_readlane(accA, 31); }
  KVZ_DEV double get_sig() const { return wave_readlane(accA, 47); }
  KVZ_DEV double get_coded() const { return wave_readlane(accA, 63); }
  KVZ_DEV double get_udist() const { return wave_readlane(accB, 15); }
  KVZ_DEV void set_base(double v, int lane) { accA = lane == 31 ? v : accA; }
};
struct RdoqSums1 {
  double acc = 0, term = 0;  // lane 15 holds base_cost
  unsigned levels = 0;
  int lane;
  KVZ_DEV void begin_group(const WaveArr<double> &sigc, const WaveArr<i32> &lvl, int lane_)
  {
    lane = lane_;
    levels = (unsigned)__ballot(lvl.v > 0) & 0xffffu;
    term = (lvl.v != 0 || lane >= 16) ? 0.0 : -sigc.v;
  }
  KVZ_DEV int next_level(int k) const { const unsigned below = levels & ((2u << k) - 1u); return below ? 31 - __builtin_clz(below) : -1; }
  KVZ_DEV void zero_step(int) { acc += term; }
  KVZ_DEV void add(double v) { acc += lane == 15 ? v : 0.0; }
  KVZ_DEV void rotate() { term = dpp_row_ror1(term); }
  KVZ_DEV double get_base() const { return wave_readlane(acc, 15); }
};
#endif

// What the wavefront routine is given, by value: prices of both bins of every context at the caller's states ([2 * idx + bin], Q15) and the block's coefficients / levels,
// all three in LDS.
struct RdoqWaveArgs {
  KVZ_LDS_PTR(const i32) ptab;
  KVZ_LDS_PTR(const i16) coef;
  KVZ_LDS_PTR(i16) dest;
  KVZ_LDS_PTR(const u8) diag8;
  double lambda;
  int qp, log2w, type /* 0 luma, 2 chroma */, scan_mode, tr_depth;
  unsigned long long *prof = nullptr;  // -DKVZ_CTU_PROFILE builds: eight cycle counters of this routine's sections (luma blocks), else unused
  KVZ_DEV i32 price(int idx, int bin) const { return ptab[2 * idx + bin]; }
};

// rdo.c:358-371: the bypass bins of the escape code of `symbol` with rice parameter g (the loop there finds the length of the exp-Golomb suffix: floor(log2(symbol - 3 * 2^g + 2^g)))
KVZ_DEV i32 rdoq_escape_bins(i32 symbol, int g)
{
  // both forms computed, one selected: the callers' lanes disagree about the branch all the time (symbol >= 0)
  const i32 pre = (symbol >> g) + 1 + g;
  const int len = 31 - __builtin_clz((unsigned)imax(1, symbol - (3 << g) + (1 << g)));
  return symbol < (3 << g) ? pre : 3 + len + 1 - g + len;
}

// The prices a position's decision reads, per lane: both bins of the greater-1 context of its class (c1 of the class within the group's context set) and of the group's
// greater-2 context.  Plain scalars on purpose: an array indexed by the class ends up in scratch memory, and a scratch load under a full device costs more than the
// rest of the decision.
struct RdoqLevelPrices { i32 one0, one1, abs0, abs1; };
KVZ_DEV int rdoq_class_c1(int r) { return r < 3 ? r + 1 : 0; }

// kvz_get_coded_level (rdo.c:413-459) of one position in chain-state class r (see the head of this section): the level, and with COSTS its coded cost and the cost of its
// significance flag.  ma > 0.
template <bool COSTS>
KVZ_DEV i32 rdoq_decide(int r, i32 ma, bool last, double c0, double dhi, double dlo, double s0, double s1, double lambda, const RdoqLevelPrices P, double *ccv_out, double *csv_out)
{
  const int cls = r < 3 ? 0 : (r < 8 ? 1 : 2), g = cls == 0 ? 0 : (cls == 1 ? r - 3 : r - 8), base_level = 3 - cls;
  const i32 at_base = cls == 0 ? P.one1 + P.abs1 : (cls == 1 ? P.one1 : 0);  // rdo.c:373-380: what the context-coded flags of a level >= base_level cost
  const double sadd = last ? 0.0 : s1;
  // rdo.c:345-392 kvz_get_ic_rate of a candidate level a >= 1
  auto rate_of = [&](i32 a) {
    const i32 hi = rdoq_escape_bins(imax(0, a - base_level), g) * (1 << 15) + at_base, lo = a == 1 ? P.one0 : P.one1 + P.abs0;  // (a == 2 below base_level 3: greater-1 set, greater-2 clear)
    return (1 << 15) + (a >= base_level ? hi : lo);
  };
  // the candidates in the reference's order -- zero (only below 3, never at the last position), max_abs_level, max_abs_level - 1 -- each taken when strictly cheaper;
  // straight-line on purpose: as a loop over one or two candidates the lanes of a wavefront walked every branch combination
  const bool zero_ok = !last && ma < 3;
  double ccv = zero_ok ? c0 + s0 : 1.7e+308, csv = zero_ok ? s0 : 0;  // MAX_DOUBLE (global.h)
  i32 level = 0;
  double cur = dhi + lambda * rate_of(ma);
  cur += sadd;
  if (cur < ccv) { level = ma; ccv = cur; csv = sadd; }
  double cur2 = dlo + lambda * rate_of(imax(1, ma - 1));
  cur2 += sadd;
  if (ma > 1 && cur2 < ccv) { level = ma - 1; ccv = cur2; csv = sadd; }
  if (COSTS) { *ccv_out = ccv; *csv_out = csv; }
  return level;
}

// Device: every lane of the wavefront calls it, converged, with wavefront-uniform arguments; lane = its index in the wavefront.  Host: one call (lane 0).
#if defined(KVZ_CTU_PROFILE) && !defined(KVZ_HOSTSIM)
#define KVZ_RQ_PROF(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); rq_t[i] += t_ - rq_last; rq_last = __builtin_amdgcn_s_memtime(); } while (0)
#define KVZ_RQ_PROF_END() do { if (lane == 0 && c.type == 0 && c.prof) { unsigned long long all_ = 0; for (int i_ = 0; i_ < 8; i_++) { c.prof[i_] += rq_t[i_]; all_ += rq_t[i_]; } \
    c.prof[8 + c.log2w - 2] += all_; c.prof[12 + c.log2w - 2] += 1ull; } } while (0)
#else
#define KVZ_RQ_PROF(i)
#define KVZ_RQ_PROF_END()
#endif
KVZ_RDOQ_WAVE_FN void rdoq_block_wave(const RdoqWaveArgs c, int lane)
{
#if defined(KVZ_CTU_PROFILE) && !defined(KVZ_HOSTSIM)
  unsigned long long rq_t[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, rq_last = __builtin_amdgcn_s_memtime();
#endif
  const int qp = c.qp, log2w = c.log2w, type = c.type, scan_mode = c.scan_mode, tr_depth = c.tr_depth;
  KVZ_LDS_PTR(const i16) coef = c.coef;
  KVZ_LDS_PTR(i16) dest = c.dest;
  (void)lane;
  const int width = 1 << log2w, n = width * width;
  const int transform_shift = 15 - 8 - log2w;
  const int qp_scaled = rdoq_scaled_qp(type, qp);
  const i32 q_bits = 14 + qp_scaled / 6 + transform_shift;
  const i32 q = rdoq_quant_scale(qp_scaled % 6);
  // scalinglist.c:349-367: err_scale = 2^15 * 2^(-2 transform_shift) / q / q -- two IEEE divisions of a power of two, and a power of two goes through a rounding
  // unchanged: the result is ((1 / q) / q) * 2^(15 - 2 transform_shift) exactly, the first factor one of six constants (folded at compile time, IEEE double), the product
  // exact.  (Two double-precision divisions per block were some eighty instructions in front of everything else.)
  const int qr = qp_scaled % 6;
  const double inv_qq = qr == 0 ? (1.0 / 26214.0) / 26214.0 : (qr == 1 ? (1.0 / 23302.0) / 23302.0 : (qr == 2 ? (1.0 / 20560.0) / 20560.0
                      : (qr == 3 ? (1.0 / 18396.0) / 18396.0 : (qr == 4 ? (1.0 / 16384.0) / 16384.0 : (1.0 / 14564.0) / 14564.0))));
  const double temp = inv_qq * (double)(1 << (15 - 2 * transform_shift));  // 15 - 2 transform_shift = 5 .. 11
  const int num_blk_side = width >> 2;
  const RdoqScanT<KVZ_LDS_PTR(const u8)> sc{ log2w, scan_mode, c.diag8 };
  const i32 round = 1 << (q_bits - 1);
  // ---- quant-generic.c:379-399 find_last_scanpos: the highest scan position that does not quantise to zero; everything above it is zero in dest.  Sixty-four
  // positions at a time from the top: the first chunk with a level ends the search (one ballot and a count of leading zeros, not a wavefront maximum over shuffles)
  int last_scanpos = -1;
  for (int base = n > 64 ? n - 64 : 0; base >= 0 && last_scanpos < 0; base -= 64) {
#ifdef KVZ_HOSTSIM
    for (int sp = imin(base + 63, n - 1); sp >= base && last_scanpos < 0; sp--) {
      const i32 ld = imin(iabs((i32)coef[sc.pos(sp)]) * q, 0x7fffffff - round);
      if (((ld + round) >> q_bits) > 0) last_scanpos = sp;
    }
    for (int sp = base; sp < imin(base + 64, n); sp++) if (sp > last_scanpos) dest[sc.pos(sp)] = 0;
#else
    const int sp = base + lane;
    bool level = false;
    u32 blkpos = 0;
    if (sp < n) {
      blkpos = sc.pos(sp);
      const i32 ld = imin(iabs((i32)coef[blkpos]) * q, 0x7fffffff - round);
      level = ((ld + round) >> q_bits) > 0;
    }
    const unsigned long long mask = __ballot(level);
    if (mask) last_scanpos = base + 63 - __builtin_clzll(mask);
    if (sp < n && sp > last_scanpos) dest[blkpos] = 0;
#endif
  }
  last_scanpos = KVZ_UNI_INT(last_scanpos);  // a scalar for everything that follows from it (the group loop's counter first of all)
  if (last_scanpos < 0) { KVZ_RQ_PROF(0); KVZ_RQ_PROF_END(); return; }
  const int cg_last_scanpos = last_scanpos >> 4;
  // rdo.c:480-509 calc_last_bits: entry k of the x / y table on lane k (k <= 9): the prices of the prefix's 1-bins before k, and of the 0-bin that ends it
  WaveArr<i32> lx_bits, ly_bits;
  {
    const int cb = log2w - 2;
    const int off = type ? 0 : (cb * 3 + ((cb + 1) >> 2)), shift = type ? cb : ((cb + 3) >> 2);
    const int bx = (type ? KVZ_HIP_CX_LAST_X_CHROMA : KVZ_HIP_CX_LAST_X_LUMA) + off, by = (type ? KVZ_HIP_CX_LAST_Y_CHROMA : KVZ_HIP_CX_LAST_Y_LUMA) + off;
    const int kmax = rdoq_group_idx(width - 1);
#ifdef KVZ_HOSTSIM
    KVZ_WAVE_LANES(l, 64) {
      i32 bits_x = 0, bits_y = 0;
      const int kk = imin(l, kmax);
      for (int k = 0; k < kk; k++) { bits_x += c.price(bx + (k >> shift), 1); bits_y += c.price(by + (k >> shift), 1); }
      if (kk < kmax) { bits_x += c.price(bx + (kk >> shift), 0); bits_y += c.price(by + (kk >> shift), 0); }
      KVZ_WA_SET(lx_bits, l, bits_x); KVZ_WA_SET(ly_bits, l, bits_y);
    }
#else
    // every lane its own entry's two bins (four independent reads), the sums before it by a prefix scan along the row (DPP row_shr, zeros shifted in)
    const bool in = lane < kmax;
    const int cxk = (lane & 15) >> shift;
    const i32 x1 = in ? c.price(bx + cxk, 1) : 0, x0 = in ? c.price(bx + cxk, 0) : 0, y1 = in ? c.price(by + cxk, 1) : 0, y0 = in ? c.price(by + cxk, 0) : 0;
    i32 sx = x1, sy = y1;
    sx += __builtin_amdgcn_update_dpp(0, sx, 0x111, 0xF, 0xF, true); sy += __builtin_amdgcn_update_dpp(0, sy, 0x111, 0xF, 0xF, true);
    sx += __builtin_amdgcn_update_dpp(0, sx, 0x112, 0xF, 0xF, true); sy += __builtin_amdgcn_update_dpp(0, sy, 0x112, 0xF, 0xF, true);
    sx += __builtin_amdgcn_update_dpp(0, sx, 0x114, 0xF, 0xF, true); sy += __builtin_amdgcn_update_dpp(0, sy, 0x114, 0xF, 0xF, true);
    sx += __builtin_amdgcn_update_dpp(0, sx, 0x118, 0xF, 0xF, true); sy += __builtin_amdgcn_update_dpp(0, sy, 0x118, 0xF, 0xF, true);
    lx_bits.v = sx - x1 + x0; ly_bits.v = sy - y1 + y0;  // lanes up to kmax (<= 9, the first row) are read
#endif
  }
  const int cg0 = KVZ_HIP_CX_SIG_CG + type;
  const int sig_base = type ? KVZ_HIP_CX_SIG_CHROMA : KVZ_HIP_CX_SIG_LUMA;
  const int one0 = type == 0 ? KVZ_HIP_CX_ONE_LUMA : KVZ_HIP_CX_ONE_CHROMA, abs0 = type == 0 ? KVZ_HIP_CX_ABS_LUMA : KVZ_HIP_CX_ABS_CHROMA;
  // lambda times both bins of the two coded-group-flag contexts on lanes 0..3, for the whole block (a memory access on the chain costs more than the arithmetic of a position)
  WaveArr<double> cg_price;
  KVZ_WAVE_LANES(l, 64) { KVZ_WA_SET(cg_price, l, c.lambda * c.price(cg0 + ((l >> 1) & 1), l & 1)); }
  // per group, on the lane of its scan index, for the last pass: the classes its positions were decided in, its context set, the cost of its coded-group flag
  WaveArr<i32> rst_lo_of, rst_hi_of, ctx_set_of;
  WaveArr<double> cg_cost_of;
  KVZ_WAVE_LANES(l, 64) { KVZ_WA_SET(rst_lo_of, l, 0); KVZ_WA_SET(rst_hi_of, l, 0); KVZ_WA_SET(ctx_set_of, l, 0); KVZ_WA_SET(cg_cost_of, l, 0.0); }
  unsigned long long sig_groups = 0;  // sig_coeffgroup_flag, bit = raster index of the group
  unsigned long long pat_lo = 0, pat_hi = 0;  // pattern_sig_ctx of every group in scan order, two bits each (the last pass prices the zero flags again)
  // the chain's state between groups (uniform)
  int ctx_set = (last_scanpos > 0 && type == 0) ? 2 : 0, c1 = 1;
  RdoqSums5 sums;
  // One group's per-position values, computed by every lane for the position lane & 15 of group cgs (all four replicas alike): `level`, `ccv` (cost_coeff) and `csv`
  // (cost_sig) only for a position that cannot code a level; the others get theirs from rdoq_decide.
  struct Pos { i32 ma, blkpos; double c0, dhi, dlo, s0, s1; bool last; };
  auto position = [&](int cgs, int l, int pattern_sig_ctx) {
    Pos p;
    const int scanpos = cgs * 16 + (l & 15);
    const bool in_chain = scanpos <= last_scanpos;
    const u32 blkpos = in_chain ? sc.pos(scanpos) : 0;
    const i32 ld = imin(iabs((i32)coef[blkpos]) * q, 0x7fffffff - round);
    p.blkpos = (i32)blkpos;
    p.ma = in_chain ? (ld + round) >> q_bits : -1;  // -1: beyond the last position, not part of the block's chain
    p.last = scanpos == last_scanpos;
    const double err = (double)ld;
    const double e_hi = (double)(ld - (p.ma * (1 << q_bits))), e_lo = (double)(ld - ((p.ma - 1) * (1 << q_bits)));
    p.c0 = in_chain ? err * err * temp : 0.0;
    p.dhi = e_hi * e_hi * temp; p.dlo = e_lo * e_lo * temp;
    const u32 pos_y = blkpos >> log2w, pos_x = blkpos - (pos_y << log2w);
    const int ctx_sig = p.last ? 0 : rdoq_sig_ctx_inc(pattern_sig_ctx, scan_mode, (int)pos_x, (int)pos_y, log2w, type);
    p.s0 = in_chain ? c.lambda * c.price(sig_base + ctx_sig, 0) : 0.0;
    p.s1 = c.lambda * c.price(sig_base + ctx_sig, 1);
    return p;
  };
  // the level prices of class r in context set `set`, read by the lane from the price table in LDS (one 8-byte read per context: both bins)
  auto level_prices = [&](int r, int set) {
    RdoqLevelPrices P;
    const int i1 = one0 + 4 * set + rdoq_class_c1(r), i2 = abs0 + set;
    P.one0 = c.price(i1, 0); P.one1 = c.price(i1, 1); P.abs0 = c.price(i2, 0); P.abs1 = c.price(i2, 1);
    return P;
  };
  KVZ_RQ_PROF(0);
  for (int cgs = cg_last_scanpos; cgs >= 0; cgs--) {
    const u32 cg_blkpos = sc.cg(cgs), cg_pos_y = cg_blkpos / num_blk_side, cg_pos_x = cg_blkpos - cg_pos_y * num_blk_side;
    u32 right = 0, lower = 0;  // context.c:339-351 / 315-327
    if ((int)cg_pos_x < num_blk_side - 1) right = (u32)(sig_groups >> (cg_pos_y * num_blk_side + cg_pos_x + 1)) & 1;
    if ((int)cg_pos_y < num_blk_side - 1) lower = (u32)(sig_groups >> ((cg_pos_y + 1) * num_blk_side + cg_pos_x)) & 1;
    const int pattern_sig_ctx = width == 4 ? -1 : (int)(right + (lower << 1));
    if (cgs < 32) pat_lo |= (unsigned long long)(pattern_sig_ctx & 3) << (2 * cgs); else pat_hi |= (unsigned long long)(pattern_sig_ctx & 3) << (2 * (cgs - 32));
    // ---- 1. the positions
    WaveArr<double> c0v, dhi, dlo, s0v, s1v, ccv, csv;
    WaveArr<i32> max_abs, blkpos_of, level_of;
    unsigned coded = 0;  // positions of the group that may quantise to a level
    KVZ_WAVE_LANES(l, 64) {
      const Pos p = position(cgs, l, pattern_sig_ctx);
      KVZ_WA_SET(max_abs, l, p.ma); KVZ_WA_SET(blkpos_of, l, p.blkpos);
      KVZ_WA_SET(c0v, l, p.c0); KVZ_WA_SET(dhi, l, p.dhi); KVZ_WA_SET(dlo, l, p.dlo); KVZ_WA_SET(s0v, l, p.s0); KVZ_WA_SET(s1v, l, p.s1);
      KVZ_WA_SET(level_of, l, 0);
      KVZ_WA_SET(ccv, l, p.c0 + p.s0); KVZ_WA_SET(csv, l, p.s0);  // what a position that can only be zero costs (rdo.c:424-428); +0.0 twice outside the chain
#ifdef KVZ_HOSTSIM
      if (l < 16 && p.ma > 0) coded |= 1u << l;
#endif
    }
#ifndef KVZ_HOSTSIM
    coded = (unsigned)__ballot(max_abs.v > 0) & 0xffffu;
#endif
    KVZ_RQ_PROF(1);
    unsigned long long rstates = 0;
    if (coded) {
      // ---- 2. the decisions per class as bit masks over (replica, position): what the chain needs to know of a decision is whether the level is non-zero (it
      // counts), above 1 (c1 falls to 0) and whether it moves the rice parameter (rdo.c:808-811) -- three wavefront ballots per set of four classes.
      // Set t holds the classes 4 t + replica: 0 and 1 before the chain, 2 (and 3: class 12, replica 0 only) once a group has coded eight levels.
      unsigned long long NZ[4] = { 0, 0, 0, 0 }, G1[4] = { 0, 0, 0, 0 }, RC[4] = { 0, 0, 0, 0 };
      auto decide_set = [&](int t) {
#ifdef KVZ_HOSTSIM
        unsigned long long nzb = 0, g1b = 0, rcb = 0;
#endif
        bool nz = false, g1 = false, rc = false;
        KVZ_WAVE_LANES(l, 64) {
          const i32 ma = KVZ_WA_OWN(max_abs, l);
          const int r = (l >> 4) + 4 * t;
          nz = g1 = rc = false;
          if (ma > 0 && r <= 12) {
            const bool last = cgs * 16 + (l & 15) == last_scanpos;
            const i32 lv = rdoq_decide<false>(r, ma, last, KVZ_WA_OWN(c0v, l), KVZ_WA_OWN(dhi, l), KVZ_WA_OWN(dlo, l), KVZ_WA_OWN(s0v, l), KVZ_WA_OWN(s1v, l), c.lambda, level_prices(r, ctx_set), nullptr, nullptr);
            const int g = r < 3 ? 0 : (r < 8 ? r - 3 : r - 8), base_level = r < 3 ? 3 : (r < 8 ? 2 : 1);
            nz = lv > 0; g1 = lv > 1; rc = lv >= base_level && lv > (3 << g);
          }
#ifdef KVZ_HOSTSIM
          nzb |= (unsigned long long)nz << l; g1b |= (unsigned long long)g1 << l; rcb |= (unsigned long long)rc << l;
#endif
        }
#ifdef KVZ_HOSTSIM
        NZ[t] = nzb; G1[t] = g1b; RC[t] = rcb;
#else
        NZ[t] = __ballot(nz); G1[t] = __ballot(g1); RC[t] = __ballot(rc);
#endif
      };
      decide_set(0);  // classes 0..3: what a group's first levels can meet; the sets of the moved rice parameter (1) and of the ninth level on (2, 3) when the chain gets there
      KVZ_RQ_PROF(2);
      // ---- 3. the chain (rdo.c:760-840 for this group), from one EVENT to the next.  Positions that stay zero in the class they are met in move nothing; a coded
      // level always counts (c1_idx), but it changes the class only when c1 moves (the first levels of a group: 1 -> 2 -> 3, or -> 0 at a level above 1), when the
      // rice parameter moves, or when it is the eighth: the coded levels between two such events are counted with one s_bcnt1 instead of one step each (a group of
      // this content holds thirteen coded levels on average and three or four events).
      int c1_idx = 0, go_rice = 0;
      int r = c1 ? c1 - 1 : 3;  // the class the group starts in
      rstates = 0x1111111111111111ull * (unsigned)r;  // every position not met yet: the current class
      unsigned have_sets = 1;
      unsigned rem = coded;
      while (rem) {
        const int t = r >> 2, sh = 16 * (r & 3);
        if (!((have_sets >> t) & 1)) {  // 1: classes 4..7 (the rice parameter has moved), 2: 8..11 (eight levels coded), 3: class 12 (both, the parameter at its cap)
          if (t == 1) decide_set(1); else if (t == 2) decide_set(2); else decide_set(3);
          have_sets |= 1u << t;
        }
        const unsigned nz16 = (unsigned)((t == 0 ? NZ[0] : (t == 1 ? NZ[1] : (t == 2 ? NZ[2] : NZ[3]))) >> sh) & 0xffffu;
        const unsigned g116 = (unsigned)((t == 0 ? G1[0] : (t == 1 ? G1[1] : (t == 2 ? G1[2] : G1[3]))) >> sh) & 0xffffu;
        const unsigned rc16 = (unsigned)((t == 0 ? RC[0] : (t == 1 ? RC[1] : (t == 2 ? RC[2] : RC[3]))) >> sh) & 0xffffu;
        const unsigned nzr = nz16 & rem;
        if (!nzr) break;
        // where the class can change next: with c1 at 1 or 2 at every level, with c1 at 3 at a level above 1, else where the rice parameter moves (below its cap)
        const unsigned ev = ((c1_idx < 8 && c1 != 0) ? (c1 == 3 ? g116 : nz16) : (go_rice < 4 ? rc16 : 0u)) & nzr;
        const int e = ev ? 31 - __builtin_clz(ev) : -1;
        const unsigned above = e < 0 ? nzr : nzr & ~((2u << e) - 1u);  // the coded levels in front of it
        const int cnt = __builtin_popcount(above);
        if (c1_idx < 8 && c1_idx + cnt >= 8) {
          // the eighth level of the group lies among them: the positions behind it are met in class 8 + go_rice.  (c1 is 0 or 3 here -- with c1 at 1 or 2 `above` is
          // empty -- and none of these levels is above 1 when it is 3: c1 stays.)
          unsigned x = above;
          for (int j = 8 - c1_idx; j > 1; j--) x &= ~(1u << (31 - __builtin_clz(x)));
          const int p = 31 - __builtin_clz(x);
          c1_idx = 8;
          rem &= (1u << p) - 1u;
          const int r2 = 8 + go_rice;
          const unsigned xr = (unsigned)(r ^ r2) * 0x11111111u;
          rstates ^= (((unsigned long long)xr << 32) | xr) & ((1ull << (4 * p)) - 1ull);
          r = r2;
          continue;
        }
        c1_idx += cnt;
        if (c1 != 0 && cnt) c1 = (g116 & above) ? 0 : imin(3, c1 + cnt);  // (only past the eighth level can these move c1: it matters for the next group's context set)
        if (e < 0) break;
        const unsigned bit = 1u << e;
        c1_idx++;
        if (rc16 & bit) go_rice = imin(go_rice + 1, 4);
        c1 = (g116 & bit) ? 0 : (c1 == 1 || c1 == 2 ? c1 + 1 : c1);
        rem &= bit - 1u;
        const int r2 = c1_idx < 8 ? (c1 ? c1 - 1 : 3 + go_rice) : 8 + go_rice;
        if (r2 != r) {  // the positions below e are met in the new class
          const unsigned xr = (unsigned)(r ^ r2) * 0x11111111u;
          rstates ^= (((unsigned long long)xr << 32) | xr) & ((1ull << (4 * e)) - 1ull);
          r = r2;
        }
      }
      KVZ_RQ_PROF(3);
      // ---- 4. every position once more, in the class the chain met it in, with its costs
      KVZ_WAVE_LANES(l, 64) {
        const i32 ma = KVZ_WA_OWN(max_abs, l);
        if (ma > 0) {
          const bool last = cgs * 16 + (l & 15) == last_scanpos;
          double cc, cs;
          const int r = (int)((rstates >> (4 * (l & 15))) & 15);
          const i32 lv = rdoq_decide<true>(r, ma, last, KVZ_WA_OWN(c0v, l), KVZ_WA_OWN(dhi, l), KVZ_WA_OWN(dlo, l), KVZ_WA_OWN(s0v, l), KVZ_WA_OWN(s1v, l), c.lambda, level_prices(r, ctx_set), &cc, &cs);
          KVZ_WA_SET(level_of, l, lv); KVZ_WA_SET(ccv, l, cc); KVZ_WA_SET(csv, l, cs);
        }
      }
    }
    KVZ_WA_PUT(rst_lo_of, cgs, (i32)(u32)rstates); KVZ_WA_PUT(rst_hi_of, cgs, (i32)(u32)(rstates >> 32)); KVZ_WA_PUT(ctx_set_of, cgs, ctx_set);
    if (cgs > 0) {  // rdo.c:822-833, at the group's first scan position
      ctx_set = (cgs == 1 || type != 0) ? 0 : 2;
      if (c1 == 0) ctx_set++;
      c1 = 1;
    }
    KVZ_RQ_PROF(4);
    // ---- 5. the ordered sums
    sums.group(c0v, ccv, csv, level_of, lane);
    unsigned levels = 0;
#ifdef KVZ_HOSTSIM
    for (int k = 0; k < 16; k++) if (level_of.v[k] > 0) levels |= 1u << k;
#else
    levels = (unsigned)__ballot(level_of.v > 0) & 0xffffu;
#endif
    KVZ_RQ_PROF(5);
    // ---- 6. the coded-group decision (rdo.c:842-900)
    double base_cost = sums.get_base();
    const bool any_level = levels != 0;
    if (any_level) sig_groups |= 1ull << cg_blkpos;
    int zeroed = 0;
    double cg_cost = 0;
    if (cgs) {
      const int ctx_sig = (int)(right || lower);
      double rd_sig_cost = sums.get_sig();
      if (!any_level) {
        cg_cost = KVZ_WA_GET(cg_price, 2 * ctx_sig);
        base_cost += cg_cost - rd_sig_cost;
      } else if (cgs < cg_last_scanpos) {
        if ((levels & 0xfffeu) == 0) { const double rd_sig_cost_0 = KVZ_WA_GET(csv, 0); base_cost -= rd_sig_cost_0; rd_sig_cost -= rd_sig_cost_0; }  // rd_nnz_before_pos0 == 0
        double cost_zero_cg = base_cost;
        cg_cost = KVZ_WA_GET(cg_price, 2 * ctx_sig + 1);
        base_cost += cg_cost;
        cost_zero_cg += KVZ_WA_GET(cg_price, 2 * ctx_sig);
        cost_zero_cg += sums.get_udist();
        cost_zero_cg -= sums.get_coded();
        cost_zero_cg -= rd_sig_cost;
        if (KVZ_UNI_INT(cost_zero_cg < base_cost)) {
          sig_groups &= ~(1ull << cg_blkpos);
          base_cost = cost_zero_cg;
          cg_cost = KVZ_WA_GET(cg_price, 2 * ctx_sig);
          zeroed = 1;  // rdo.c:888-897: its levels are gone; the last pass skips the group
        }
      }
    } else sig_groups |= 1ull << cg_blkpos;
    KVZ_WA_PUT(cg_cost_of, cgs, cg_cost);
    sums.set_base(base_cost, lane);
    // ---- the group's levels, sixteen lanes
    KVZ_WAVE_LANES(l, 16) {
      if (KVZ_WA_OWN(max_abs, l) >= 0) dest[KVZ_WA_OWN(blkpos_of, l)] = (i16)(zeroed ? 0 : KVZ_WA_OWN(level_of, l));
    }
    KVZ_RQ_PROF(6);
  }
  // ---- the last position (rdo.c:903-957), intra block: coded block flag of the transform unit
  double best_cost;
  int best_last_idx_p1 = 0;
  RdoqSums1 walk;
  {
    const int ctx_cbf = type == 0 ? KVZ_HIP_CX_CBF_LUMA + !tr_depth : (tr_depth < 2 ? KVZ_HIP_CX_CBF_CHROMA + tr_depth : KVZ_HIP_CX_CBF_CHROMA_DEEP + imin(tr_depth, 3) - 2);
    best_cost = sums.get_uncoded() + c.lambda * c.price(ctx_cbf, 0);
    double base_cost = sums.get_base();
    base_cost += c.lambda * c.price(ctx_cbf, 1);
#ifdef KVZ_HOSTSIM
    walk.base = base_cost;
#else
    walk.lane = lane; walk.acc = base_cost;
#endif
  }
  int found_last = 0;
  for (int cgs = cg_last_scanpos; cgs >= 0 && !found_last; cgs--) {
    walk.add(-KVZ_WA_GET(cg_cost_of, cgs));
    if (!((sig_groups >> sc.cg(cgs)) & 1)) continue;
    // the group's positions again: levels and costs from the classes the first pass decided them in, and lambda times the rate of ending the block at a level
    // (rdo.c:465-478 get_rate_last)
    const int pat2 = (int)(((cgs < 32 ? pat_lo >> (2 * cgs) : pat_hi >> (2 * (cgs - 32)))) & 3);
    const unsigned long long rstates = (unsigned long long)(u32)KVZ_WA_GET(rst_lo_of, cgs) | (unsigned long long)(u32)KVZ_WA_GET(rst_hi_of, cgs) << 32;
    const int set_of_group = KVZ_WA_GET(ctx_set_of, cgs);
    WaveArr<double> c0v, ccv, sigc, lastc;
    WaveArr<i32> lvl;
    KVZ_WAVE_LANES(l, 64) {
      const Pos p = position(cgs, l, width == 4 ? -1 : pat2);
      const u32 pos_y = (u32)p.blkpos >> log2w, pos_x = (u32)p.blkpos - (pos_y << log2w);
      const u32 px = scan_mode == 2 ? pos_y : pos_x, py = scan_mode == 2 ? pos_x : pos_y;  // SCAN_VER swaps (rdo.c:934)
      const int gx = rdoq_group_idx((int)px), gy = rdoq_group_idx((int)py);
      const i32 lxb = KVZ_WA_AT(lx_bits, l, gx), lyb = KVZ_WA_AT(ly_bits, l, gy);  // every lane takes part in the exchange (converged here)
      i32 level = p.ma < 0 ? -1 : 0;
      double cc = p.c0 + p.s0, cs = p.s0, lc = 0;
      if (p.ma > 0) {
        const int r = (int)((rstates >> (4 * (l & 15))) & 15);
        level = rdoq_decide<true>(r, p.ma, p.last, p.c0, p.dhi, p.dlo, p.s0, p.s1, c.lambda, level_prices(r, set_of_group), &cc, &cs);
      }
      if (level > 0) {
        double ui_cost = lxb + lyb;
        if (gx > 3) ui_cost += (double)((1 << 15) * ((gx - 2) >> 1));
        if (gy > 3) ui_cost += (double)((1 << 15) * ((gy - 2) >> 1));
        lc = c.lambda * ui_cost;
      }
      KVZ_WA_SET(lvl, l, level); KVZ_WA_SET(c0v, l, p.c0); KVZ_WA_SET(ccv, l, cc); KVZ_WA_SET(sigc, l, cs); KVZ_WA_SET(lastc, l, lc);
    }
    KVZ_RQ_PROF(6);
    walk.begin_group(sigc, lvl, lane);
    for (int k = 15; k >= 0;) {
      const int next = walk.next_level(k);  // zero positions down to the next level: base_cost -= cost_sig each
      for (int z = k; z > next; z--) { walk.zero_step(z); walk.rotate(); }
      k = next;
      if (k < 0) break;
      const i32 level = KVZ_WA_GET(lvl, k);
      const double cs = KVZ_WA_GET(sigc, k);
      const double total = walk.get_base() + KVZ_WA_GET(lastc, k) - cs;
      if (KVZ_UNI_INT(total < best_cost)) { best_last_idx_p1 = cgs * 16 + k + 1; best_cost = total; }
      if (level > 1) { found_last = 1; break; }
      walk.add(-KVZ_WA_GET(ccv, k));
      walk.add(KVZ_WA_GET(c0v, k));
      walk.rotate();
      k--;
    }
    KVZ_RQ_PROF(6);
  }
#ifndef KVZ_HOSTSIM
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the levels written above (LDS in the CTU pass) are read back below by other lanes: one wavefront's LDS operations execute in order
#endif
  KVZ_WAVE_STRIDE(sp, last_scanpos + 1) {
    const u32 blkpos = sc.pos(sp);
    if (sp < best_last_idx_p1) { const i32 level = dest[blkpos]; dest[blkpos] = (i16)(coef[blkpos] < 0 ? -level : level); }
    else dest[blkpos] = 0;
  }
#ifndef KVZ_HOSTSIM
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
  KVZ_RQ_PROF(0);
  KVZ_RQ_PROF_END();
}

// One item = one block of a batch of equally shaped blocks, ONE WAVEFRONT PER BLOCK (the per-call entry points kvz_hip_rdoq / kvz_hip_rdoq_blocks /
// kvz_hip_quantize_residual_rdoq run the same routine as the CTU pass): the prices of both bins of every context at the caller's states, the block's coefficients and its
// levels staged in LDS around rdoq_block_wave.  Backends launch it with run_wave (64 lanes per item; the host simulation calls lane 0).
struct RdoqOp {
  const Tables *tb; const u8 *ctx; double lambda; int qp; const i16 *coef; i16 *dest; int log2w, type, scan_mode, tr_depth;
  KVZ_DEV void wave(int item, int lane) const
  {
    const int n = 1 << (2 * log2w);
#ifdef KVZ_HOSTSIM
    i32 ptab[2 * 148];
    i16 s_coef[32 * 32], s_dest[32 * 32];
    u8 s_diag8[64];
    const int lanes = 1;
#else
    __shared__ i32 ptab[2 * 148];
    __shared__ alignas(8) i16 s_coef[32 * 32];
    __shared__ alignas(8) i16 s_dest[32 * 32];
    __shared__ u8 s_diag8[64];
    const int lanes = 64;
#endif
    for (int v = lane; v < 2 * 148; v += lanes) ptab[v] = (i32)tb->entropy_bits[ctx[v >> 1] ^ (v & 1)];
    for (int i = lane; i < n; i += lanes) { s_coef[i] = coef[(long)item * n + i]; s_dest[i] = dest[(long)item * n + i]; }
    for (int i = lane; i < 64; i += lanes) s_diag8[i] = tb->diag8[i];
#ifndef KVZ_HOSTSIM
    __syncthreads();
#endif
    RdoqWaveArgs ra;
    ra.ptab = (KVZ_LDS_PTR(const i32))ptab; ra.coef = (KVZ_LDS_PTR(const i16))s_coef; ra.dest = (KVZ_LDS_PTR(i16))s_dest; ra.diag8 = (KVZ_LDS_PTR(const u8))s_diag8;
    ra.lambda = lambda; ra.qp = qp; ra.log2w = log2w; ra.type = type; ra.scan_mode = scan_mode; ra.tr_depth = tr_depth;
    rdoq_block_wave(ra, lane);
#ifndef KVZ_HOSTSIM
    __syncthreads();
#endif
    for (int i = lane; i < n; i += lanes) dest[(long)item * n + i] = s_dest[i];
  }
};

}  // namespace kvz
