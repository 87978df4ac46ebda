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
// The block by a whole WAVEFRONT (the CTU pass, kvz_ctu.hpp recon_tus; RdoqOp at the end of this file).  kvz_rdoq is a chain of decisions, but most of what it computes per coefficient does not
// depend on the chain at all.  What is serial, and is kept in the reference's order:
//   * the (c1, c2, go_rice, c1_idx, c2_idx) state inside a 4x4 group, which only positions that can quantise to a non-zero level move or read;
//   * the double-precision running sums (base_cost, block_uncoded_cost, the group sums): floating-point addition is not associative, so they are added one term at a
//     time in scan order -- an addition per position, not a few hundred dependent instructions per position as a one-lane walk would have;
//   * the zero-the-group and best-last-position decisions, which compare those sums.
// What is per-position runs on the lanes, 64 scan positions (four 4x4 groups) at a time: scan position -> block position, level_double, max_abs_level,
// err^2 * temp, the distortion of the two candidate levels; and per group, once its pattern_sig_ctx is known, the significance context and lambda times the price
// of both its bins, and the cost of coding a zero (coded_cost0 + the zero flag) -- for a position whose max_abs_level is 0, the common case, that IS its result.
//
// The chain itself is executed by ALL lanes in lock step on wavefront-uniform values: a per-position operand is fetched from the lane that holds it with
// v_readlane (a scalar register, no LDS round trip on the chain), results that a later pass needs go back into lane-held storage with v_writelane -- the level of
// a position into its own lane, the coded cost of every non-zero level into a 64-entry FIFO (one entry per lane; the last-position pass meets the non-zero
// levels in the order the first pass produced them), the cost of each coded-group flag into the lane of its group.  Nothing per-position is kept in memory: the
// last pass recomputes what is a pure function of the coefficient and the group's pattern (cost_coeff0, cost_sig) and takes the rest from the FIFO.  Levels are
// written where the caller wants them (LDS in the CTU pass) by the lanes, sixteen at a time.
//
// The host simulation runs the same source: lane-held storage is an array there (WaveArr), a "lane loop" a plain loop, the chain runs once.
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
// The ordered sums of the first pass -- block_uncoded_cost, base_cost, the group's rd_sig_cost -- take one term per scan position, in scan order; for a position
// that can only be zero (the common case) the three terms are known before the chain starts.  On the device the three sums sit on lane 15 of the first three
// 16-lane rows of one register; the group's sixteen terms of each sum are laid out along the same row, so a zero position is ONE double-precision addition for
// all three sums and a rotation of the term register within its rows (DPP row_ror:1) -- against three additions and six v_readlane.  A position that may quantise to a
// level adds its terms as a constant on the three lanes.  The host build keeps three scalars and adds in the same order.
#ifdef KVZ_HOSTSIM
struct RdoqSums3 {
  double uncoded = 0, base = 0, sig = 0;
  const WaveArr<double> *c0v = nullptr, *ccv0 = nullptr, *sig0 = nullptr;
  const WaveArr<i32> *max_abs = nullptr;
  int group = 0;
  void begin_group(int gi, const WaveArr<double> &a, const WaveArr<double> &b, const WaveArr<double> &c, const WaveArr<i32> &m, int) { c0v = &a; ccv0 = &b; sig0 = &c; max_abs = &m; sig = 0; group = gi; }
  bool may_code(int L) const { return max_abs->v[L] > 0; }
  int next_coded(int k) const { while (k >= 0 && !(max_abs->v[group * 16 + k] > 0)) k--; return k; }  // the highest position <= k of the group that may code a level, -1: none
  void zero_step(int L) { if (max_abs->v[L] == 0) { uncoded += c0v->v[L]; base += ccv0->v[L]; sig += sig0->v[L]; } }
  void add(double c0, double ccv, double csv) { uncoded += c0; base += ccv; sig += csv; }
  void rotate() {}
  double get_base() const { return base; }
  double get_sig() const { return sig; }
  double get_uncoded() const { return uncoded; }
  void set_base(double v) { base = v; }
};
// ... and the one of the last pass: base_cost minus the zero flag's cost of every zero position on the way down
struct RdoqSums1 {
  double base = 0;
  const WaveArr<double> *sigc = nullptr;
  const WaveArr<i32> *lvl = nullptr;
  void begin_group(int gi, const WaveArr<double> &a, const WaveArr<i32> &l, int) { sigc = &a; lvl = &l; group = gi; }
  bool is_level(int L) const { return lvl->v[L] > 0; }
  int group = 0;
  int next_level(int k) const { while (k >= 0 && !(lvl->v[group * 16 + k] > 0)) k--; return k; }
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
struct RdoqSums3 {
  double acc = 0, term = 0;  // per lane: lanes 15 / 31 / 47 hold the sums; `term`: the next term of each sum on those lanes
  unsigned coded = 0;        // positions of the group that may quantise to a level (wavefront-uniform)
  int lane;
  // row r of `term` := row gi of the r-th array, zero where the position is not a plain zero (beyond the last position, or handled by the caller)
  KVZ_DEV void begin_group(int gi, const WaveArr<double> &c0v, const WaveArr<double> &ccv0, const WaveArr<double> &sig0, const WaveArr<i32> &max_abs, int lane_)
  {
    lane = lane_;
    const int src = (lane & 15) + 16 * gi;
    const double r0 = __shfl(c0v.v, src), r1 = __shfl(ccv0.v, src), r2 = __shfl(sig0.v, src);
    const int ma = __shfl(max_abs.v, src);
    coded = (unsigned)(__ballot(max_abs.v > 0) >> (16 * gi)) & 0xffffu;
    term = (ma != 0 || lane >= 48) ? 0.0 : (lane < 16 ? r0 : (lane < 32 ? r1 : r2));
    acc = lane == 47 ? 0.0 : acc;
  }
  KVZ_DEV bool may_code(int L) const { return (coded >> (L & 15)) & 1; }
  KVZ_DEV int next_coded(int k) const { const unsigned below = coded & ((2u << k) - 1u); return below ? 31 - __builtin_clz(below) : -1; }
  KVZ_DEV void zero_step(int) { acc += term; }
  KVZ_DEV void add(double c0, double ccv, double csv) { acc += lane == 15 ? c0 : (lane == 31 ? ccv : (lane == 47 ? csv : 0.0)); }
  KVZ_DEV void rotate() { term = dpp_row_ror1(term); }
  KVZ_DEV double get_base() const { return wave_readlane(acc, 31); }
  KVZ_DEV double get_sig() const { return wave_readlane(acc, 47); }
  KVZ_DEV double get_uncoded() const { return wave_readlane(acc, 15); }
  KVZ_DEV void set_base(double v) { acc = lane == 31 ? v : acc; }
};
struct RdoqSums1 {
  double acc = 0, term = 0;  // lane 15 holds base_cost
  unsigned levels = 0;
  int lane;
  KVZ_DEV void begin_group(int gi, const WaveArr<double> &sigc, const WaveArr<i32> &lvl, int lane_)
  {
    lane = lane_;
    const int src = (lane & 15) + 16 * gi;
    const double r0 = __shfl(sigc.v, src);
    const int lv = __shfl(lvl.v, src);
    levels = (unsigned)(__ballot(lvl.v > 0) >> (16 * gi)) & 0xffffu;
    term = (lv != 0 || lane >= 16) ? 0.0 : -r0;
  }
  KVZ_DEV bool is_level(int L) const { return (levels >> (L & 15)) & 1; }
  KVZ_DEV int next_level(int k) const { const unsigned below = levels & ((2u << k) - 1u); return below ? 31 - __builtin_clz(below) : -1; }
  KVZ_DEV void zero_step(int) { acc += term; }
  KVZ_DEV void add(double v) { acc += lane == 15 ? v : 0.0; }
  KVZ_DEV void rotate() { term = dpp_row_ror1(term); }
  KVZ_DEV double get_base() const { return wave_readlane(acc, 15); }
};
#endif

// What the wavefront routine is given, by value: prices of both bins of every context at the caller's states ([2 * idx + bin], Q15) and the block's coefficients / levels,
// all three in LDS; cost3: scratch in memory, only touched when a block holds more than 64 non-zero levels (the FIFO's overflow), w * w doubles.
struct RdoqWaveArgs {
  KVZ_LDS_PTR(const i32) ptab;
  KVZ_LDS_PTR(const i16) coef;
  KVZ_LDS_PTR(i16) dest;
  KVZ_LDS_PTR(const u8) diag8;
  double *cost3;
  double lambda;
  int qp, log2w, type /* 0 luma, 2 chroma */, scan_mode, tr_depth;
  unsigned long long *prof = nullptr;  // -DKVZ_CTU_PROFILE builds: eight cycle counters of this routine's sections (luma blocks), else unused
  KVZ_DEV i32 price(int idx, int bin) const { return ptab[2 * idx + bin]; }
};
// Device: every lane of the wavefront calls it, converged, with wavefront-uniform arguments; lane = its index in the wavefront.  Host: one call (lane 0).
#if defined(KVZ_CTU_PROFILE) && !defined(KVZ_HOSTSIM)
#define KVZ_RQ_PROF(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); rq_t[i] += t_ - rq_last; rq_last = __builtin_amdgcn_s_memtime(); } while (0)
#define KVZ_RQ_PROF_END() do { if (lane == 0 && c.type == 0 && c.prof) { unsigned long long all_ = 0; for (int i_ = 0; i_ < 8; i_++) { atomicAdd(&c.prof[i_], rq_t[i_]); all_ += rq_t[i_]; } \
    atomicAdd(&c.prof[8 + c.log2w - 2], all_); atomicAdd(&c.prof[12 + c.log2w - 2], 1ull); } } while (0)
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
  double *cost3 = c.cost3;
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
  const int num_blk_side = width >> 2;
  const RdoqScanT<KVZ_LDS_PTR(const u8)> sc{ log2w, scan_mode, c.diag8 };
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
    my_last = __builtin_amdgcn_readfirstlane(x) - 1;
  }
#endif
  const int last_scanpos = my_last;
  KVZ_WAVE_STRIDE(sp, n) { if (sp > last_scanpos) dest[sc.pos(sp)] = 0; }
  if (last_scanpos < 0) { KVZ_RQ_PROF(0); KVZ_RQ_PROF_END(); return; }
  const int cg_last_scanpos = last_scanpos >> 4;
  // rdo.c:480-509 calc_last_bits: entry k of the x / y table on lane k (k <= 9)
  WaveArr<i32> lx_bits, ly_bits;
  {
    const int cb = log2w - 2;
    const int off = type ? 0 : (cb * 3 + ((cb + 1) >> 2)), shift = type ? cb : ((cb + 3) >> 2);
    const int bx = (type ? KVZ_HIP_CX_LAST_X_CHROMA : KVZ_HIP_CX_LAST_X_LUMA) + off, by = (type ? KVZ_HIP_CX_LAST_Y_CHROMA : KVZ_HIP_CX_LAST_Y_LUMA) + off;
    const int kmax = rdoq_group_idx(width - 1);
    KVZ_WAVE_LANES(l, 64) {
      i32 bits_x = 0, bits_y = 0;
      const int kk = imin(l, kmax);
      for (int k = 0; k < kk; k++) { bits_x += c.price(bx + (k >> shift), 1); bits_y += c.price(by + (k >> shift), 1); }
      if (kk < kmax) { bits_x += c.price(bx + (kk >> shift), 0); bits_y += c.price(by + (kk >> shift), 0); }
      KVZ_WA_SET(lx_bits, l, bits_x); KVZ_WA_SET(ly_bits, l, bits_y);
    }
  }
  const int cg0 = KVZ_HIP_CX_SIG_CG + type;
  const int sig_base = type ? KVZ_HIP_CX_SIG_CHROMA : KVZ_HIP_CX_SIG_LUMA;
  // prices the chain needs, on lanes (a memory access on the chain costs more than the arithmetic of a position): lambda times both bins of the two
  // coded-group-flag contexts on lanes 0..3 for the whole block; per group, both bins of its four greater-1 contexts (lanes 0..7) and of its greater-2 context (8, 9)
  WaveArr<double> cg_price;
  WaveArr<i32> lvl_price;
  KVZ_WAVE_LANES(l, 64) { KVZ_WA_SET(cg_price, l, c.lambda * c.price(cg0 + ((l >> 1) & 1), l & 1)); KVZ_WA_SET(lvl_price, l, 0); }
  const int one0 = type == 0 ? KVZ_HIP_CX_ONE_LUMA : KVZ_HIP_CX_ONE_CHROMA, abs0 = type == 0 ? KVZ_HIP_CX_ABS_LUMA : KVZ_HIP_CX_ABS_CHROMA;
  // lane-held storage that lives across the passes
  WaveArr<double> fifo, cg_cost_of;   // coded cost of the non-zero levels in the order they are decided; cost of the coded-group flag of group g on lane g
  WaveArr<i32> level_of;              // the level decided for the position a lane holds (valid during its super-group)
  KVZ_WAVE_LANES(l, 64) { KVZ_WA_SET(fifo, l, 0.0); KVZ_WA_SET(cg_cost_of, l, 0.0); KVZ_WA_SET(level_of, l, 0); }
  int fifo_n = 0;
  unsigned long long sig_groups = 0;  // sig_coeffgroup_flag, bit = raster index of the group
  unsigned long long pat_lo = 0, pat_hi = 0;  // pattern_sig_ctx of every group in scan order, two bits each (the last pass prices the zero flags again)
  // the chain's state (uniform)
  int ctx_set = (last_scanpos > 0 && type == 0) ? 2 : 0, c1 = 1, c2 = 0, go_rice = 0;
  u32 c1_idx = 0, c2_idx = 0;
  RdoqSums3 sums;
  KVZ_RQ_PROF(0);
  for (int sg = cg_last_scanpos >> 2; sg >= 0; sg--) {
    // ---- 64 positions, one per lane: everything that does not depend on a decision
    WaveArr<double> c0v, dhi, dlo, sig0, sig1, ccv0, pre_ccv1, pre_ccv2, pre_ccv3;
    WaveArr<i32> max_abs, blkpos_of, pre_lvl1, pre_lvl2, pre_lvl3;
    KVZ_WAVE_LANES(l, 64) {
      const int scanpos = sg * 64 + l;
      const bool in_block = scanpos < n;
      const u32 blkpos = in_block ? sc.pos(scanpos) : 0;
      const i32 ld = imin(iabs((i32)coef[blkpos]) * q, 0x7fffffff - round);
      const i32 ma = (!in_block || scanpos > last_scanpos) ? -1 : (ld + round) >> q_bits;  // -1: beyond the last position, not part of the block's chain
      const double err = (double)ld;
      const double e_hi = (double)(ld - (ma * (1 << q_bits))), e_lo = (double)(ld - ((ma - 1) * (1 << q_bits)));
      KVZ_WA_SET(blkpos_of, l, (i32)blkpos); KVZ_WA_SET(max_abs, l, ma);
      KVZ_WA_SET(c0v, l, err * err * temp);
      KVZ_WA_SET(dhi, l, e_hi * e_hi * temp); KVZ_WA_SET(dlo, l, e_lo * e_lo * temp);
      KVZ_WA_SET(sig0, l, 0.0); KVZ_WA_SET(sig1, l, 0.0); KVZ_WA_SET(ccv0, l, 0.0);
      KVZ_WA_SET(pre_ccv1, l, 0.0); KVZ_WA_SET(pre_ccv2, l, 0.0); KVZ_WA_SET(pre_ccv3, l, 0.0); KVZ_WA_SET(pre_lvl1, l, 0); KVZ_WA_SET(pre_lvl2, l, 0); KVZ_WA_SET(pre_lvl3, l, 0);
    }
    KVZ_RQ_PROF(1);
    for (int gi = 3; gi >= 0; gi--) {
      const int cgs = sg * 4 + gi;
      if (cgs > cg_last_scanpos) continue;
      const u32 cg_blkpos = sc.cg(cgs), cg_pos_y = cg_blkpos / num_blk_side, cg_pos_x = cg_blkpos - cg_pos_y * num_blk_side;
      u32 right = 0, lower = 0;  // context.c:339-351 / 315-327
      if ((int)cg_pos_x < num_blk_side - 1) right = (u32)(sig_groups >> (cg_pos_y * num_blk_side + cg_pos_x + 1)) & 1;
      if ((int)cg_pos_y < num_blk_side - 1) lower = (u32)(sig_groups >> ((cg_pos_y + 1) * num_blk_side + cg_pos_x)) & 1;
      const int pattern_sig_ctx = width == 4 ? -1 : (int)(right + (lower << 1));
      if (cgs < 32) pat_lo |= (unsigned long long)(pattern_sig_ctx & 3) << (2 * cgs); else pat_hi |= (unsigned long long)(pattern_sig_ctx & 3) << (2 * (cgs - 32));
      // the group's sixteen lanes: the significance flag's two prices, and the cost of a zero
      KVZ_WAVE_LANES(l, 64) {
        if ((l >> 4) == gi && KVZ_WA_OWN(max_abs, l) >= 0) {
          const int scanpos = sg * 64 + l;
          const u32 blkpos = (u32)KVZ_WA_OWN(blkpos_of, l), pos_y = blkpos >> log2w, pos_x = blkpos - (pos_y << log2w);
          const int ctx_sig = scanpos == last_scanpos ? 0 : rdoq_sig_ctx_inc(pattern_sig_ctx, scan_mode, (int)pos_x, (int)pos_y, log2w, type);
          const double s0 = c.lambda * c.price(sig_base + ctx_sig, 0), s1 = c.lambda * c.price(sig_base + ctx_sig, 1);
          KVZ_WA_SET(sig0, l, s0); KVZ_WA_SET(sig1, l, s1); KVZ_WA_SET(ccv0, l, KVZ_WA_OWN(c0v, l) + s0);
        }
      }
      // the group's level prices (its context set is fixed by now: it only changes where a group ends)
      KVZ_WAVE_LANES(l, 64) {
        if (l < 10) KVZ_WA_SET(lvl_price, l, l < 8 ? c.price(one0 + 4 * ctx_set + (l >> 1), l & 1) : c.price(abs0 + ctx_set, l & 1));
      }
      // The decision of every position that may quantise to a level, for the three states the chain is in until the group sees its first level above 1
      // (c1 = 1, 2, 3 with c2 = 0, go_rice = 0, fewer than eight levels so far -- by far the most common ones): rdo.c:413-459 kvz_get_coded_level with
      // rdo.c:345-392 kvz_get_ic_rate, all sixteen positions at once.  The chain then only picks the result of its state; any other state takes the general path.
      {
        const i32 p_abs0 = KVZ_WA_GET(lvl_price, 8), p_abs1 = KVZ_WA_GET(lvl_price, 9);
        i32 p_one0[3], p_one1[3];
        for (int j = 0; j < 3; j++) { p_one0[j] = KVZ_WA_GET(lvl_price, 2 * (j + 1)); p_one1[j] = KVZ_WA_GET(lvl_price, 2 * (j + 1) + 1); }
        KVZ_WAVE_LANES(l, 64) {
          const i32 ma = KVZ_WA_OWN(max_abs, l);
          if ((l >> 4) == gi && ma > 0) {
            const bool last = sg * 64 + l == last_scanpos;
            const double cur_cost_sig = last ? 0.0 : KVZ_WA_OWN(sig1, l);
            const i32 min_abs = ma > 1 ? ma - 1 : 1;
            for (int j = 0; j < 3; j++) {
              double ccv = (!last && ma < 3) ? KVZ_WA_OWN(ccv0, l) : 1.7e+308;
              i32 level = 0;
              for (i32 a = ma; a >= min_abs; a--) {
                i32 rate = 1 << 15;
                if (a >= 3) {
                  i32 symbol = a - 3, length;
                  if (symbol < 3) rate += (symbol + 1) * (1 << 15);
                  else {
                    length = 0;
                    symbol = symbol - 3;
                    while (symbol >= (1 << length)) symbol -= (1 << (length++));
                    rate += (3 + length + 1 + length) * (1 << 15);
                  }
                  rate += p_one1[j] + p_abs1;
                } else if (a == 1) rate += p_one0[j];
                else rate += p_one1[j] + p_abs0;
                double cur = (a == ma ? KVZ_WA_OWN(dhi, l) : KVZ_WA_OWN(dlo, l)) + c.lambda * rate;
                cur += cur_cost_sig;
                if (cur < ccv) { level = a; ccv = cur; }
              }
              if (j == 0) { KVZ_WA_SET(pre_ccv1, l, ccv); KVZ_WA_SET(pre_lvl1, l, level); }
              else if (j == 1) { KVZ_WA_SET(pre_ccv2, l, ccv); KVZ_WA_SET(pre_lvl2, l, level); }
              else { KVZ_WA_SET(pre_ccv3, l, ccv); KVZ_WA_SET(pre_lvl3, l, level); }
            }
          }
        }
      }
      KVZ_RQ_PROF(2);
      // ---- the chain: rdo.c:760-840 for this group, then its coded-group decision (rdo.c:842-900)
      double rd_coded_level_and_dist = 0, rd_uncoded_dist = 0, rd_sig_cost_0 = 0;
      int rd_nnz_before_pos0 = 0, any_level = 0;
      const int fifo_group_start = fifo_n;
      sums.begin_group(gi, c0v, ccv0, sig0, max_abs, lane);
      for (int k = 15; k >= 0;) {
        // the run of positions that can only be zero (or lie beyond the last position: no term), down to the next one that may code a level: one addition each
        const int next = sums.next_coded(k);
        for (int z = k; z > next; z--) { sums.zero_step(gi * 16 + z); sums.rotate(); }
        k = next;
        if (k < 0) break;
        const int L = gi * 16 + k, scanpos = cgs * 16 + k;
        {
          const i32 ma = KVZ_WA_GET(max_abs, L);
          const double c0 = KVZ_WA_GET(c0v, L);
          double ccv, csv = 0;
          i32 level = 0;
          if (c1 >= 1 && c1_idx < 8) {
            // the common states: decided by the position's lane beforehand (above)
            const bool last = scanpos == last_scanpos;
            if (c1 == 1) { ccv = KVZ_WA_GET(pre_ccv1, L); level = KVZ_WA_GET(pre_lvl1, L); }
            else if (c1 == 2) { ccv = KVZ_WA_GET(pre_ccv2, L); level = KVZ_WA_GET(pre_lvl2, L); }
            else { ccv = KVZ_WA_GET(pre_ccv3, L); level = KVZ_WA_GET(pre_lvl3, L); }
            csv = level ? (last ? 0.0 : KVZ_WA_GET(sig1, L)) : KVZ_WA_GET(sig0, L);
          } else {
            // rdo.c:413-459 kvz_get_coded_level on the precomputed pieces
            const bool last = scanpos == last_scanpos;
            double cur_cost_sig = 0;
            if (!last && ma < 3) { csv = KVZ_WA_GET(sig0, L); ccv = KVZ_WA_GET(ccv0, L); }
            else ccv = 1.7e+308;
            if (!last) cur_cost_sig = KVZ_WA_GET(sig1, L);
            const i32 min_abs = ma > 1 ? ma - 1 : 1;
            for (i32 a = ma; a >= min_abs; a--) {
              // rdo.c:345-392 kvz_get_ic_rate with the prices of this group's contexts taken from their lanes (abs_ctx = ctx_set + c2 is only ever priced with c2 == 0)
              i32 rate = 1 << 15;
              {
                const i32 base_level = c1_idx < 8 ? (2 + (c2_idx < 1)) : 1;
                if (a >= base_level) {
                  i32 symbol = a - base_level, length;
                  if (symbol < (3 << go_rice)) { length = symbol >> go_rice; rate += (length + 1 + go_rice) * (1 << 15); }
                  else {
                    length = go_rice;
                    symbol = symbol - (3 << go_rice);
                    while (symbol >= (1 << length)) symbol -= (1 << (length++));
                    rate += (3 + length + 1 - go_rice + length) * (1 << 15);
                  }
                  if (c1_idx < 8) {
                    rate += KVZ_WA_GET(lvl_price, 2 * c1 + 1);
                    if (c2_idx < 1) rate += KVZ_WA_GET(lvl_price, 9);
                  }
                } else if (a == 1) rate += KVZ_WA_GET(lvl_price, 2 * c1);
                else if (a == 2) { rate += KVZ_WA_GET(lvl_price, 2 * c1 + 1); rate += KVZ_WA_GET(lvl_price, 8); }
              }
              double cur = (a == ma ? KVZ_WA_GET(dhi, L) : KVZ_WA_GET(dlo, L)) + c.lambda * rate;
              cur += cur_cost_sig;
              if (KVZ_UNI_INT(cur < ccv)) { level = a; ccv = cur; csv = cur_cost_sig; }
            }
          }
          const i32 base_level = c1_idx < 8 ? (2 + (c2_idx < 1)) : 1;
          if (level >= base_level && level > 3 * (1 << go_rice)) go_rice = imin(go_rice + 1, 4);
          if (level >= 1) c1_idx++;
          if (level > 1) { c1 = 0; c2 += (c2 < 2); c2_idx++; }
          else if (c1 < 3 && c1 > 0 && level) c1++;
          KVZ_WA_PUT(level_of, L, level);
          sums.add(c0, ccv, csv);  // block_uncoded_cost += cost_coeff0; base_cost += cost_coeff; rd_sig_cost += cost_sig
          if (k == 0) rd_sig_cost_0 = csv;
          if (level) {
            any_level = 1;
            rd_coded_level_and_dist += ccv - csv;
            rd_uncoded_dist += c0;
            if (k != 0) rd_nnz_before_pos0++;
            if (fifo_n < 64) KVZ_WA_PUT(fifo, fifo_n, ccv);
            else {
#ifdef KVZ_HOSTSIM
              cost3[fifo_n - 64] = ccv;
#else
              if (lane == 0) cost3[fifo_n - 64] = ccv;
#endif
            }
            fifo_n++;
          }
        }
        sums.rotate();
        k--;
      }
      if (!sums.may_code(gi * 16)) rd_sig_cost_0 = KVZ_WA_GET(sig0, gi * 16);  // the group's first position only coded a zero flag
      if (cgs > 0) {  // rdo.c:822-833, at the group's first scan position
        c2 = 0; go_rice = 0; c1_idx = 0; c2_idx = 0;
        ctx_set = (cgs == 1 || type != 0) ? 0 : 2;
        if (c1 == 0) ctx_set++;
        c1 = 1;
      }
      KVZ_RQ_PROF(3);
      double base_cost = sums.get_base(), rd_sig_cost = sums.get_sig();
      if (any_level) sig_groups |= 1ull << cg_blkpos;
      int zeroed = 0;
      double cg_cost = 0;
      if (cgs) {
        const int ctx_sig = (int)(right || lower);
        if (!any_level) {
          cg_cost = KVZ_WA_GET(cg_price, 2 * ctx_sig);
          base_cost += cg_cost - rd_sig_cost;
        } else if (cgs < cg_last_scanpos) {
          if (rd_nnz_before_pos0 == 0) { base_cost -= rd_sig_cost_0; rd_sig_cost -= rd_sig_cost_0; }
          double cost_zero_cg = base_cost;
          cg_cost = KVZ_WA_GET(cg_price, 2 * ctx_sig + 1);
          base_cost += cg_cost;
          cost_zero_cg += KVZ_WA_GET(cg_price, 2 * ctx_sig);
          cost_zero_cg += rd_uncoded_dist;
          cost_zero_cg -= rd_coded_level_and_dist;
          cost_zero_cg -= rd_sig_cost;
          if (KVZ_UNI_INT(cost_zero_cg < base_cost)) {
            sig_groups &= ~(1ull << cg_blkpos);
            base_cost = cost_zero_cg;
            cg_cost = KVZ_WA_GET(cg_price, 2 * ctx_sig);
            zeroed = 1;
            fifo_n = fifo_group_start;  // rdo.c:888-897: its levels are gone; the last pass skips the group
          }
        }
      } else sig_groups |= 1ull << cg_blkpos;
      KVZ_WA_PUT(cg_cost_of, cgs, cg_cost);
      sums.set_base(base_cost);
      KVZ_RQ_PROF(4);
      // ---- the group's levels, sixteen lanes
      KVZ_WAVE_LANES(l, 64) {
        if ((l >> 4) == gi) {
          const i32 ma = KVZ_WA_OWN(max_abs, l);
          if (ma >= 0) dest[KVZ_WA_OWN(blkpos_of, l)] = (i16)((zeroed || ma == 0) ? 0 : KVZ_WA_OWN(level_of, l));
        }
      }
      KVZ_RQ_PROF(5);
    }
  }
#ifndef KVZ_HOSTSIM
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the levels just written (LDS in the CTU pass) are read back below by other lanes: one wavefront's LDS operations execute in order
#endif
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
  int fifo_r = 0, found_last = 0;
  for (int sg = cg_last_scanpos >> 2; sg >= 0 && !found_last; sg--) {
    // per position: the level; for a level its costs are recomputed (cost_coeff0, the flag's price) and lambda times the rate of ending the block there
    // (rdo.c:465-478 get_rate_last); for a zero the price of its zero flag
    WaveArr<double> c0v, sigc, lastc;
    WaveArr<i32> lvl;
    KVZ_WAVE_LANES(l, 64) {
      const int scanpos = sg * 64 + l, cgs = scanpos >> 4;
      const bool in_chain = scanpos <= last_scanpos;
      const u32 blkpos = in_chain ? sc.pos(scanpos) : 0, pos_y = blkpos >> log2w, pos_x = blkpos - (pos_y << log2w);
      const u32 px = scan_mode == 2 ? pos_y : pos_x, py = scan_mode == 2 ? pos_x : pos_y;  // SCAN_VER swaps (rdo.c:934)
      const int gx = rdoq_group_idx((int)px), gy = rdoq_group_idx((int)py);
      const i32 lxb = KVZ_WA_AT(lx_bits, l, gx), lyb = KVZ_WA_AT(ly_bits, l, gy);  // every lane takes part in the exchange (converged here)
      i32 level = -1;
      double c0 = 0, sgc = 0, lc = 0;
      if (in_chain) {
        level = dest[blkpos];
        const int pat2 = (int)(((cgs < 32 ? pat_lo >> (2 * cgs) : pat_hi >> (2 * (cgs - 32)))) & 3);
        const int ctx_sig = scanpos == last_scanpos ? 0 : rdoq_sig_ctx_inc(width == 4 ? -1 : pat2, scan_mode, (int)pos_x, (int)pos_y, log2w, type);
        sgc = scanpos == last_scanpos ? 0.0 : c.lambda * c.price(sig_base + ctx_sig, level != 0);  // cost_sig of the position: its flag as it was coded
        if (level) {
          const i32 ld = imin(iabs((i32)coef[blkpos]) * q, 0x7fffffff - round);
          const double err = (double)ld;
          c0 = err * err * temp;
          double ui_cost = lxb + lyb;
          if (gx > 3) ui_cost += (double)((1 << 15) * ((gx - 2) >> 1));
          if (gy > 3) ui_cost += (double)((1 << 15) * ((gy - 2) >> 1));
          lc = c.lambda * ui_cost;
        }
      }
      KVZ_WA_SET(lvl, l, level); KVZ_WA_SET(c0v, l, c0); KVZ_WA_SET(sigc, l, sgc); KVZ_WA_SET(lastc, l, lc);
    }
    KVZ_RQ_PROF(6);
    for (int gi = 3; gi >= 0 && !found_last; gi--) {
      const int cgs = sg * 4 + gi;
      if (cgs > cg_last_scanpos) continue;
      walk.add(-KVZ_WA_GET(cg_cost_of, cgs));
      if (!((sig_groups >> sc.cg(cgs)) & 1)) continue;
      walk.begin_group(gi, sigc, lvl, lane);
      for (int k = 15; k >= 0;) {
        const int next = walk.next_level(k);  // zero positions down to the next level: base_cost -= cost_sig each
        for (int z = k; z > next; z--) { walk.zero_step(gi * 16 + z); walk.rotate(); }
        k = next;
        if (k < 0) break;
        const int L = gi * 16 + k;
        const i32 level = KVZ_WA_GET(lvl, L);
        const double csv = KVZ_WA_GET(sigc, L);
        const double total = walk.get_base() + KVZ_WA_GET(lastc, L) - csv;
        if (KVZ_UNI_INT(total < best_cost)) { best_last_idx_p1 = cgs * 16 + k + 1; best_cost = total; }
        if (level > 1) { found_last = 1; break; }
        double ccv;
        if (fifo_r < 64) ccv = KVZ_WA_GET(fifo, fifo_r); else ccv = cost3[fifo_r - 64];
        fifo_r++;
        walk.add(-ccv);
        walk.add(KVZ_WA_GET(c0v, L));
        walk.rotate();
        k--;
      }
    }
    KVZ_RQ_PROF(7);
  }
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
  const Tables *tb; const u8 *ctx; double lambda; int qp; const i16 *coef; i16 *dest; int log2w, type, scan_mode, tr_depth; double *tmp;
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
    ra.cost3 = tmp + (long)item * 3 * n; ra.lambda = lambda; ra.qp = qp; ra.log2w = log2w; ra.type = type; ra.scan_mode = scan_mode; ra.tr_depth = tr_depth;
    rdoq_block_wave(ra, lane);
#ifndef KVZ_HOSTSIM
    __syncthreads();
#endif
    for (int i = lane; i < n; i += lanes) dest[(long)item * n + i] = s_dest[i];
  }
};

}  // namespace kvz
