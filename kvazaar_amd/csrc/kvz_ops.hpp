// kvz_ops.hpp -- device "ops" of the hip strategy: one functor per kernel, one item per thread.
//
// Every op is a POD of pointers + scalars with `void operator()(int item) const`; the launcher maps
// item = blockIdx.x * blockDim.x + threadIdx.x over a 1-D grid (kvz_launch.hpp).  Items of one launch are
// independent except for integer atomicAdd accumulation into an output word, so the arithmetic is
// schedule-independent and bit-exact by construction.
//
// The same source is compiled for the host by tests/hostsim (KVZ_HOSTSIM): there each op is run by a plain
// `for (item...)` loop so that the index arithmetic of every kernel is checked against the oracle on a
// machine without a GPU.  That build is test infrastructure only; libkvz_hip.so contains no host compute path.
//
// Semantics follow kvazaar v2.3.2's generic strategies (file:line cited per op, relative to
// /root/reference/src); KVZ_BIT_DEPTH = 8 throughout.
#pragma once
#include <stdint.h>

#ifdef KVZ_HOSTSIM
#define KVZ_DEV inline
#define KVZ_ATOMIC_ADD(p, v) (*(p) += (v))
#else
#include <hip/hip_runtime.h>
#define KVZ_DEV __device__ __forceinline__
#define KVZ_ATOMIC_ADD(p, v) atomicAdd((p), (v))
#endif

namespace kvz {

typedef uint8_t u8;
typedef int16_t i16;
typedef uint16_t u16;
typedef uint32_t u32;
typedef int32_t i32;

#ifdef KVZ_HOSTSIM
#define KVZ_HD inline
#else
#define KVZ_HD __host__ __device__ __forceinline__
#endif
KVZ_HD int iabs(int v) { return v < 0 ? -v : v; }
KVZ_HD int iclip(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }
KVZ_HD int imin(int a, int b) { return a < b ? a : b; }
KVZ_HD int imax(int a, int b) { return a > b ? a : b; }
KVZ_DEV u8 clip_pixel(int v) { return (u8)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

// ------------------------------------------------------------------------------------------------
// Constant tables, uploaded once (kvz_tables.hpp builds them on the host).
// ------------------------------------------------------------------------------------------------
// Layout the entries of Tables::mref_tab address (kvz_ctu.hpp CtuShared): a row of the extended main references is kMrefStride bytes with ref_main[0] at kMrefOrg,
// the top / left reference arrays of a plane are kMrefRefRow bytes apart, the filtered set lies kMrefFiltered bytes behind the unfiltered one.
constexpr unsigned kMrefStride = 36, kMrefOrg = 16, kMrefRefRow = 68, kMrefFiltered = 408;

struct Tables {
  i16 dct[4][32 * 32];   // [log2n-2] row-major n*n (dct-generic.c:46-120)
  i16 dst4[16];          // dct-generic.c:38-44
  // The 16- and 32-point matrices again as signed bytes (|v| <= 90), the table operand of the MFMA transforms (kvz_mfma.hpp):
  // [0: n = 16, 1: n = 32][0: C row-major, 1: C transposed]; rows of n bytes in the k-slot order of the instruction -- natural for
  // n = 16, and for n = 32 byte 16 h + j of a row = entry 8 (j >> 2) + 4 h + (j & 3).  dct_sum: the row sums.
  alignas(16) int8_t dct_i8[2][2][32 * 32];
  alignas(16) i32 dct_sum[2][2][32];
  // ... and 16 x 16 block-diagonal matrices for the small transforms, so that one 16 x 16 product carries several blocks:
  // [0: diag(C4 x4), 1: diag(C8 x2), 2: diag(DST4 x4)][0: T, 1: T transposed]
  alignas(16) int8_t bd_i8[3][2][16 * 16];
  alignas(16) i32 bd_sum[3][2][16];
  // The small matrices as int16 pairs for v_dot2_i32_i16 (kvz_dev.hpp dev_transform_rows_kernel): [0: C4, 1: C8, 2: DST4][0: rows of M, 1: rows of M^T]
  // [output][pair] = M[k][2 i] | M[k][2 i + 1] << 16
  u32 small_pairs[3][2][8][4];
  u32 pairs16[2][16][8];  // the same for C16
  u32 scan[3][4][1024];  // [scan_idx][log2-2] (tables.c kvz_g_sig_last_scan), sizes 4..32
  // intra.c:47-82 num_ref_pixels_top / num_ref_pixels_left: reference samples available above-right / below-left of the 4x4
  // unit at [y / 4][x / 4] of a CTU, regenerated from the z-order of the units (kvz_tables.hpp)
  u8 avail_top[16][16], avail_left[16][16];
  // CABAC context state machine (H.265 table 9-41) on kvazaar's packed state (state << 1 | MPS, cabac.c:40-62):
  // [0] after coding the more probable symbol, [1] after the less probable one
  u8 ctx_next[2][128];
  // CTU pass, rough search: where every entry of the extended main references of the angular modes 11..25 comes from (CtuShared::mref; intra-generic.c:97-123) for an
  // 8x8 [0] / 16x16 [1] CU.  Entry i (mode 11 + i / NQ, q = i % NQ - W, NQ = 2 W + 2) = source byte offset from CtuShared::ref (unfiltered top 0 / left 68, the
  // filtered ones 408 / 476 behind them) | destination byte offset from CtuShared::mref << 16
  u32 mref_tab[2][512];
  u8 diag8[64];  // up-right diagonal order of an 8x8 grid: the coefficient groups of a 32x32 block (tables.h:66-79 g_sig_last_scan_32x32)
  u32 entropy_bits[128];       // kvz_entropy_bits (rdo.c:69-80): Q15 price of a bin, index = context state ^ bin
  int8_t luma_filter[4][8];    // filter.c:66-72
  int8_t chroma_filter[8][4];  // filter.c:74-84
};

// ------------------------------------------------------------------------------------------------
// Picture costs
// ------------------------------------------------------------------------------------------------

// SAD of a w x h rectangle, one item per (block, row).  picture-generic.c:98-111 reg_sad, :475-501 sad_NxN,
// :687-700 ver_sad (b_stride = 0), and -- with clamp = 1 -- the frame-edge glue of image.c:279-397
// (cor_sad / ver_sad / hor_sad pieces == SAD against the edge-replicated reference).
struct SadRectOp {
  const u8 *a; int a_stride; long a_bstride;
  const u8 *b; int b_stride; long b_bstride;
  int w, h;
  int clamp, ref_w, ref_h, ref_x, ref_y;  // clamp: b is the frame origin, block at (ref_x, ref_y)
  u32 *out;                               // out[block] += ...
  KVZ_DEV void operator()(int item) const
  {
    const int blk = item / h, y = item - blk * h;
    const u8 *pa = a + blk * a_bstride + (long)y * a_stride;
    u32 s = 0;
    if (!clamp) {
      const u8 *pb = b + blk * b_bstride + (long)y * b_stride;
      for (int x = 0; x < w; x++) s += iabs((int)pa[x] - (int)pb[x]);
    } else {
      const int ry = iclip(0, ref_h - 1, ref_y + y);
      const u8 *pb = b + blk * b_bstride + (long)ry * b_stride;
      for (int x = 0; x < w; x++) s += iabs((int)pa[x] - (int)pb[iclip(0, ref_w - 1, ref_x + x)]);
    }
    KVZ_ATOMIC_ADD(&out[blk], s);
  }
};

// picture-generic.c:729-752 hor_sad_generic: left/right columns replicated from ref[left] / ref[w-right-1]
struct HorSadOp {
  const u8 *pic; int pic_stride; const u8 *ref; int ref_stride; int w, h, left, right; u32 *out;
  KVZ_DEV void operator()(int y) const
  {
    const u8 *p = pic + (long)y * pic_stride, *r = ref + (long)y * ref_stride;
    u32 s = 0;
    for (int x = 0; x < w; x++) {
      int rx = x;
      if (left) rx = x < left ? left : x; else if (right) rx = x >= w - right ? w - right - 1 : x;
      s += iabs((int)p[x] - (int)r[rx]);
    }
    KVZ_ATOMIC_ADD(out, s);
  }
};

// picture-generic.c:536-551 pixels_calc_ssd, one item per (block, row)
struct SsdOp {
  const u8 *a; int a_stride; long a_bstride; const u8 *b; int b_stride; long b_bstride; int w; u32 *out;
  KVZ_DEV void operator()(int item) const
  {
    const int blk = item / w, y = item - blk * w;
    const u8 *pa = a + blk * a_bstride + (long)y * a_stride, *pb = b + blk * b_bstride + (long)y * b_stride;
    u32 s = 0;
    for (int x = 0; x < w; x++) { int d = (int)pa[x] - (int)pb[x]; s += (u32)(d * d); }
    KVZ_ATOMIC_ADD(&out[blk], s);
  }
};

// 4x4 Hadamard SATD: picture-generic.c:117-196, (sum + 1) >> 1.
KVZ_DEV u32 satd4(const u8 *a, int as, const u8 *b, int bs)
{
  int t[16];
  for (int r = 0; r < 4; r++) {
    int d0 = (int)a[r * as] - b[r * bs], d1 = (int)a[r * as + 1] - b[r * bs + 1], d2 = (int)a[r * as + 2] - b[r * bs + 2],
        d3 = (int)a[r * as + 3] - b[r * bs + 3];
    t[4 * r] = d0 + d1 + d2 + d3; t[4 * r + 1] = d0 - d1 + d2 - d3; t[4 * r + 2] = d0 + d1 - d2 - d3; t[4 * r + 3] = d0 - d1 - d2 + d3;
  }
  int sum = 0;
  for (int c = 0; c < 4; c++) {
    int p = t[c], q = t[4 + c], r = t[8 + c], s = t[12 + c];
    sum += iabs(p + q + r + s) + iabs(p - q + r - s) + iabs(p + q - r - s) + iabs(p - q - r + s);
  }
  return (u32)((sum + 1) >> 1);
}

// 8-point Hadamard butterfly on 8 registers (natural order; only sum |.| is used afterwards).
#define KVZ_HAD8(v0, v1, v2, v3, v4, v5, v6, v7)                                                   \
  {                                                                                                \
    int a0 = v0 + v4, a1 = v1 + v5, a2 = v2 + v6, a3 = v3 + v7, a4 = v0 - v4, a5 = v1 - v5, a6 = v2 - v6, a7 = v3 - v7; \
    int b0 = a0 + a2, b1 = a1 + a3, b2 = a0 - a2, b3 = a1 - a3, b4 = a4 + a6, b5 = a5 + a7, b6 = a4 - a6, b7 = a5 - a7; \
    v0 = b0 + b1; v1 = b0 - b1; v2 = b2 + b3; v3 = b2 - b3; v4 = b4 + b5; v5 = b4 - b5; v6 = b6 + b7; v7 = b6 - b7;     \
  }

// 8x8 Hadamard SATD: picture-generic.c:252-340, (sum + 2) >> 2.
KVZ_DEV u32 satd8(const u8 *a, int as, const u8 *b, int bs)
{
  int m[8][8];
  for (int r = 0; r < 8; r++) {
    for (int c = 0; c < 8; c++) m[r][c] = (int)a[r * as + c] - (int)b[r * bs + c];
    KVZ_HAD8(m[r][0], m[r][1], m[r][2], m[r][3], m[r][4], m[r][5], m[r][6], m[r][7]);
  }
  int sum = 0;
  for (int c = 0; c < 8; c++) {
    KVZ_HAD8(m[0][c], m[1][c], m[2][c], m[3][c], m[4][c], m[5][c], m[6][c], m[7][c]);
    for (int r = 0; r < 8; r++) sum += iabs(m[r][c]);
  }
  return (u32)((sum + 2) >> 2);
}

// One tile of a SATD decomposition: size 4 or 8 at (x, y) of the block.
struct SatdTile { i16 x, y, size, pad; };

// SATD over a tile list, one item per (block, tile).  Covers satd_NxN (strategies-picture.h:53-69, all 8x8 tiles or
// one 4x4), satd_any_size (:75-113: 4x4 first column/row, 8x8 rest) and satd_any_size_quad incl. its row-offset
// quirk (picture-generic.c:404-471) -- the tile list is built on the host by kvz_satd_tiles().
struct SatdTilesOp {
  const u8 *a; int a_stride; long a_bstride;
  const u8 *b; int b_stride; long b_bstride;
  const SatdTile *tiles; int n_tiles;
  u32 *out;
  KVZ_DEV void operator()(int item) const
  {
    const int blk = item / n_tiles, t = item - blk * n_tiles;
    const SatdTile tl = tiles[t];
    const u8 *pa = a + blk * a_bstride + (long)tl.y * a_stride + tl.x;
    const u8 *pb = b + blk * b_bstride + (long)tl.y * b_stride + tl.x;
    const u32 s = tl.size == 4 ? satd4(pa, a_stride, pb, b_stride) : satd8(pa, a_stride, pb, b_stride);
    KVZ_ATOMIC_ADD(&out[blk], s);
  }
};

// picture-generic.c:553-614 bipred_average_{px_px,im_im,px_im}: (a + b + 64) >> 7 of 14-bit samples
struct BipredOp {
  u8 *dst; int dst_stride; const u8 *px0; const i16 *im0; const u8 *px1; const i16 *im1; int w;
  KVZ_DEV void operator()(int i) const
  {
    const int y = i / w, x = i - y * w;
    const i16 s0 = px0 ? (i16)(px0[i] << 6) : im0[i];
    const i16 s1 = px1 ? (i16)(px1[i] << 6) : im1[i];
    dst[y * dst_stride + x] = clip_pixel(((int)s0 + (int)s1 + 64) >> 7);
  }
};

// picture-generic.c:755-778 pixel_var_generic.  Double precision with the reference's summation order (one item,
// sequential): the operation order is part of the bit-exact contract, so this is deliberately not a tree reduction.
struct PixelVarOp {
  const u8 *buf; u32 len; double *out;
  KVZ_DEV void operator()(int) const
  {
    double sum = 0;
    for (u32 i = 0; i < len; i++) sum += (double)buf[i];
    const double mean = sum / (double)len;
    double var = 0;
    for (u32 i = 0; i < len; i++) {
      const double t = (double)buf[i] - mean;
#ifdef KVZ_HOSTSIM
      var += t * t;
#else
      var = __dadd_rn(var, __dmul_rn(t, t));  // no FMA contraction
#endif
    }
    *out = var / (double)len;
  }
};

// ------------------------------------------------------------------------------------------------
// Transforms.  dct-generic.c:255-577: the partial butterflies are exact regroupings of integer matrix
// products, so each pass is one dot product per output coefficient.
//   forward pass : dst[k*n + j] = (short)((sum_i C[k][i]*src[j*n+i] + add) >> shift)      (wraps)
//   inverse pass : dst[j*n + i] = clip16((sum_k C[k][i]*src[k*n+j] + add) >> shift)
// One item per output coefficient of every block.
// ------------------------------------------------------------------------------------------------
struct TransformPassOp {
  const i16 *C; int n, log2n, shift, inverse;
  const i16 *src; i16 *dst;  // blocks are contiguous n*n
  KVZ_DEV void operator()(int item) const
  {
    const int nn = n * n, blk = item >> (2 * log2n), e = item & (nn - 1);
    const i16 *s = src + (long)blk * nn;
    const int add = 1 << (shift - 1);
    int acc = 0;
    if (!inverse) {
      const int k = e >> log2n, j = e & (n - 1);
      for (int i = 0; i < n; i++) acc += (int)C[k * n + i] * (int)s[j * n + i];
      dst[(long)blk * nn + e] = (i16)((acc + add) >> shift);
    } else {
      const int j = e >> log2n, i = e & (n - 1);
      for (int k = 0; k < n; k++) acc += (int)C[k * n + i] * (int)s[k * n + j];
      dst[(long)blk * nn + e] = (i16)iclip(-32768, 32767, (acc + add) >> shift);
    }
  }
};

// transform.c:164-196 transform skip (4x4 only in practice)
struct TransformSkipOp {
  int shift, inverse; const i16 *src; i16 *dst;
  KVZ_DEV void operator()(int i) const
  {
    if (!inverse) dst[i] = (i16)((uint16_t)src[i] << shift);
    else dst[i] = (i16)(((int)src[i] + (1 << (shift - 1))) >> shift);
  }
};

// ------------------------------------------------------------------------------------------------
// Quantisation (quant-generic.c)
// ------------------------------------------------------------------------------------------------
struct QuantScalars {
  int q_bits, add, flat_q;      // forward: quant-generic.c:57-66
  int dq_shift, dq_scale;       // inverse flat path: :335-339
  int dq_list, dq_qp_per;       // scaling-list path :309-333 (dq_shift already includes the +4)
};

// quant-generic.c:68-81: level = (|c|*q + add) >> q_bits, sign, clip16; ac_sum accumulates the unsigned levels
struct QuantOp {
  QuantScalars q; const i16 *coef; const i16 *qcoef_tab /* or null */; i16 *out; u32 *ac_sum; int per_block;
  KVZ_DEV void operator()(int i) const
  {
    const int c = coef[i];
    const int64_t a = (int64_t)iabs(c);
    const int qs = qcoef_tab ? (int)qcoef_tab[i % per_block] : q.flat_q;
    int level = (int)((a * qs + q.add) >> q.q_bits);
    KVZ_ATOMIC_ADD(&ac_sum[i / per_block], (u32)level);
    if (c < 0) level = -level;
    out[i] = (i16)iclip(-32768, 32767, level);
  }
};

// quant-generic.c:84-179 sign-bit hiding, one item per block (sequential over coefficient groups: `last_cg`
// carries state from the highest group downwards).
struct SignHideOp {
  QuantScalars q; const i16 *coef; const i16 *qcoef_tab; i16 *qc; const u32 *ac_sum; const u32 *scan; int per_block;
  KVZ_DEV int delta_u(const i16 *cf, int n) const
  {
    const int64_t a = (int64_t)iabs((int)cf[n]);
    const int qs = qcoef_tab ? (int)qcoef_tab[n] : q.flat_q;
    const int level = (int)((a * qs + q.add) >> q.q_bits);
    return (int)((a * qs - ((int64_t)level << q.q_bits)) >> (q.q_bits - 8));
  }
  KVZ_DEV void operator()(int blk) const
  {
    if (ac_sum[blk] < 2) return;
    const i16 *cf = coef + (long)blk * per_block;
    i16 *o = qc + (long)blk * per_block;
    int last_cg = -1;
    for (int subset = (per_block - 1) >> 4; subset >= 0; subset--) {
      int first_nz = 16, last_nz = -1, abssum = 0, n;
      const int subpos = subset << 4;
      for (n = 15; n >= 0; n--) if (o[scan[n + subpos]]) { last_nz = n; break; }
      for (n = 0; n < 16; n++) if (o[scan[n + subpos]]) { first_nz = n; break; }
      for (n = first_nz; n <= last_nz; n++) abssum += o[scan[n + subpos]];
      if (last_nz >= 0 && last_cg == -1) last_cg = 1;
      if (last_nz - first_nz >= 4) {
        const int signbit = o[scan[subpos + first_nz]] > 0 ? 0 : 1;
        if (signbit != (abssum & 1)) {
          int min_cost = 0x7fffffff, min_pos = -1, cur_cost = 0x7fffffff;
          int final_change = 0, cur_change = 0;
          for (n = (last_cg == 1 ? last_nz : 15); n >= 0; n--) {
            const int blkpos = (int)scan[n + subpos];
            if (o[blkpos] != 0) {
              const int du = delta_u(cf, blkpos);
              if (du > 0) { cur_cost = -du; cur_change = 1; }
              else if (n == first_nz && iabs(o[blkpos]) == 1) { cur_cost = 0x7fffffff; }
              else { cur_cost = du; cur_change = -1; }
            } else if (n < first_nz && ((cf[blkpos] >= 0) ? 0 : 1) != signbit) {
              cur_cost = 0x7fffffff;
            } else { cur_cost = -delta_u(cf, blkpos); cur_change = 1; }
            if (cur_cost < min_cost) { min_cost = cur_cost; final_change = cur_change; min_pos = blkpos; }
          }
          if (o[min_pos] == 32767 || o[min_pos] == -32768) final_change = -1;
          if (cf[min_pos] >= 0) o[min_pos] = (i16)(o[min_pos] + final_change); else o[min_pos] = (i16)(o[min_pos] - final_change);
        }
      }
      if (last_cg == 1) last_cg = 0;
    }
  }
};

// quant-generic.c:298-340 kvz_dequant_generic
struct DequantOp {
  QuantScalars q; const i16 *qc; const i16 *dq_tab; i16 *out; int per_block;
  KVZ_DEV void operator()(int i) const
  {
    int c;
    if (q.dq_list) {
      const int d = dq_tab[i % per_block];
      if (q.dq_shift > q.dq_qp_per) c = (((int)qc[i] * d) + (1 << (q.dq_shift - q.dq_qp_per - 1))) >> (q.dq_shift - q.dq_qp_per);
      else c = iclip(-32768, 32767, (int)qc[i] * d) << (q.dq_qp_per - q.dq_shift);
    } else {
      c = ((int)qc[i] * q.dq_scale + (1 << (q.dq_shift - 1))) >> q.dq_shift;
    }
    out[i] = (i16)iclip(-32768, 32767, c);
  }
};

// quant-generic.c:214-222 residual = ref - pred (int16), blocks packed w*w
struct ResidualOp {
  const u8 *ref; const u8 *pred; int stride, w; i16 *out;
  KVZ_DEV void operator()(int i) const { const int y = i / w, x = i - y * w; out[i] = (i16)((int)ref[y * stride + x] - (int)pred[y * stride + x]); }
};

// flag[0] |= any coefficient non-zero (quant-generic.c:246-255)
struct AnyNonzeroOp {
  const i16 *c; u32 *flag;
  KVZ_DEV void operator()(int i) const { if (c[i] != 0) KVZ_ATOMIC_ADD(flag, 1u); }
};

// quant-generic.c:259-290: rec = clip(residual + pred) when has_coeffs && !early_skip, else rec = pred
struct ReconOp {
  const i16 *resid; const u8 *pred; int in_stride; u8 *rec; int out_stride, w; const u32 *flag; int early_skip;
  KVZ_DEV void operator()(int i) const
  {
    const int y = i / w, x = i - y * w;
    const int p = pred[y * in_stride + x];
    if (*flag && !early_skip) rec[y * out_stride + x] = (u8)iclip(0, 255, (int)(i16)(resid[i] + p));
    else rec[y * out_stride + x] = (u8)p;
  }
};

// quant-generic.c:342-349
struct AbsSumOp {
  const i16 *c; u32 *out;
  KVZ_DEV void operator()(int i) const { KVZ_ATOMIC_ADD(out, (u32)iabs((int)c[i])); }
};

// nal-generic.c:57-82 array_checksum (SEI decoded picture hash, method "checksum"): sum over the plane of sample ^ mask(x, y),
// 32-bit wrap-around.  One item per row.
struct PlaneChecksumOp {
  const u8 *data; int width, stride, y0; u32 *out;  // rows y0.. of the plane, data pointing at row y0
  KVZ_DEV void operator()(int r) const
  {
    const u8 *row = data + (long)r * stride;
    const int y = y0 + r;
    u32 sum = 0;
    for (int x = 0; x < width; x++) sum += (u32)(row[x] ^ (u8)((x & 0xff) ^ (y & 0xff) ^ (x >> 8) ^ (y >> 8)));
    KVZ_ATOMIC_ADD(out, sum);
  }
};

// nal-generic.c:41-55 array_md5_generic: MD5 (RFC 1321, extras/libmd5.c) of the plane's width * height bytes.  A hash is a serial chain over
// 64-byte blocks, so one item = one whole message (or one chunk of it: the chaining value goes in and out through `state`); batches of
// planes run one chain per lane.  K[i] = floor(2^32 |sin(i + 1)|) and the per-round rotations are the RFC's tables.
KVZ_DEV u32 md5_rotl(u32 x, int c) { return (x << c) | (x >> (32 - c)); }
KVZ_DEV void md5_block(u32 st[4], const u32 m[16])
{
  const u32 K[64] = { 0xd76aa478, 0xe8c7b756, 0x242070db, 0xc1bdceee, 0xf57c0faf, 0x4787c62a, 0xa8304613, 0xfd469501, 0x698098d8, 0x8b44f7af, 0xffff5bb1, 0x895cd7be, 0x6b901122,
                      0xfd987193, 0xa679438e, 0x49b40821, 0xf61e2562, 0xc040b340, 0x265e5a51, 0xe9b6c7aa, 0xd62f105d, 0x02441453, 0xd8a1e681, 0xe7d3fbc8, 0x21e1cde6, 0xc33707d6,
                      0xf4d50d87, 0x455a14ed, 0xa9e3e905, 0xfcefa3f8, 0x676f02d9, 0x8d2a4c8a, 0xfffa3942, 0x8771f681, 0x6d9d6122, 0xfde5380c, 0xa4beea44, 0x4bdecfa9, 0xf6bb4b60,
                      0xbebfbc70, 0x289b7ec6, 0xeaa127fa, 0xd4ef3085, 0x04881d05, 0xd9d4d039, 0xe6db99e5, 0x1fa27cf8, 0xc4ac5665, 0xf4292244, 0x432aff97, 0xab9423a7, 0xfc93a039,
                      0x655b59c3, 0x8f0ccc92, 0xffeff47d, 0x85845dd1, 0x6fa87e4f, 0xfe2ce6e0, 0xa3014314, 0x4e0811a1, 0xf7537e82, 0xbd3af235, 0x2ad7d2bb, 0xeb86d391 };
  const int R[4][4] = { { 7, 12, 17, 22 }, { 5, 9, 14, 20 }, { 4, 11, 16, 23 }, { 6, 10, 15, 21 } };
  u32 a = st[0], b = st[1], c = st[2], d = st[3];
  for (int i = 0; i < 64; i++) {
    u32 f;
    int g;
    if (i < 16) { f = (b & c) | (~b & d); g = i; }
    else if (i < 32) { f = (d & b) | (~d & c); g = (5 * i + 1) & 15; }
    else if (i < 48) { f = b ^ c ^ d; g = (3 * i + 5) & 15; }
    else { f = c ^ (b | ~d); g = (7 * i) & 15; }
    const u32 t = d;
    d = c;
    c = b;
    b = b + md5_rotl(a + f + K[i] + m[g], R[i >> 4][i & 3]);
    a = t;
  }
  st[0] += a; st[1] += b; st[2] += c; st[3] += d;
}
// `blocks` whole 64-byte blocks from p (4-byte aligned), then -- when finish -- the tail bytes, the 0x80 / zero padding and the bit length of
// the whole message (total_bytes); digest = the state words little-endian.
KVZ_DEV void md5_run(u32 st[4], const u8 *p, long blocks, int tail, int finish, unsigned long long total_bytes)
{
  for (long b = 0; b < blocks; b++) {
    u32 m[16];
    const u32 *w = reinterpret_cast<const u32 *>(p + b * 64);
    for (int i = 0; i < 16; i++) m[i] = w[i];
    md5_block(st, m);
  }
  if (!finish) return;
  u8 buf[128];
  const u8 *t = p + blocks * 64;
  for (int i = 0; i < 128; i++) buf[i] = i < tail ? t[i] : (i == tail ? 0x80 : 0);
  const int last = tail < 56 ? 1 : 2;
  const unsigned long long bits = total_bytes * 8;
  for (int i = 0; i < 8; i++) buf[last * 64 - 8 + i] = (u8)(bits >> (8 * i));
  for (int k = 0; k < last; k++) {
    u32 m[16];
    for (int i = 0; i < 16; i++) m[i] = (u32)buf[k * 64 + 4 * i] | ((u32)buf[k * 64 + 4 * i + 1] << 8) | ((u32)buf[k * 64 + 4 * i + 2] << 16) | ((u32)buf[k * 64 + 4 * i + 3] << 24);
    md5_block(st, m);
  }
}
struct Md5Op {  // one item: a chunk of one message
  const u8 *data; long blocks; int tail, finish; unsigned long long total; u32 *state /* [4] in/out */;
  KVZ_DEV void operator()(int) const
  {
    u32 st[4] = { state[0], state[1], state[2], state[3] };
    md5_run(st, data, blocks, tail, finish, total);
    for (int i = 0; i < 4; i++) state[i] = st[i];
  }
};

// quant-generic.c:351-375: sum of Q8.8 weights[min(|c|,3)]; the /256.0 happens on the host (exact)
struct FastCoeffCostOp {
  const i16 *c; uint64_t weights; u32 *out;
  KVZ_DEV void operator()(int i) const
  {
    int a = iabs((int)c[i]);
    if (a > 3) a = 3;
    KVZ_ATOMIC_ADD(out, (u32)((weights >> (16 * a)) & 0xffff));
  }
};

// quant-generic.c:379-399 find_last_scanpos_generic, one item (sequential scan from the end).
// res[0] = last_scanpos (or -1), res[1] = cg_last_scanpos, res[2] = cg_scanpos, res[3] = ctx_set, res[4] = blkpos
struct FindLastOp {
  const i16 *coef; i16 *dest; const i16 *quant_coeff; const u32 *scan; int type, q_bits, cg_size, cg_num; i32 *res;
  KVZ_DEV void operator()(int) const
  {
    res[0] = -1; res[1] = -1; res[3] = 0; res[4] = -1;
    int cg;
    for (cg = cg_num - 1; cg >= 0; cg--) {
      for (int in_cg = cg_size - 1; in_cg >= 0; in_cg--) {
        const int scanpos = cg * cg_size + in_cg;
        const u32 blkpos = scan[scanpos];
        const int qv = quant_coeff[blkpos];
        int ld = imin(iabs((int)coef[blkpos]) * qv, 0x7fffffff - (1 << (q_bits - 1)));
        const u32 mx = (u32)((ld + (1 << (q_bits - 1))) >> q_bits);
        if (mx > 0) {
          res[0] = scanpos; res[1] = cg; res[2] = cg; res[3] = (scanpos > 0 && type == 0) ? 2 : 0; res[4] = (i32)blkpos;
          return;
        }
        dest[blkpos] = 0;
      }
    }
    res[2] = cg;  // -1
  }
};

// ------------------------------------------------------------------------------------------------
// Intra prediction (intra-generic.c).  One item per predicted pixel; refs are [2w+1] with index 0 = corner.
// ------------------------------------------------------------------------------------------------

// Sample of the (extended) main reference at block-relative index i >= most_negative_index (intra-generic.c:82-104):
// i >= -1 comes from the main reference, i <= -2 is projected from the side reference with the inverse angle.
KVZ_DEV int angular_ref(const u8 *main_ref, const u8 *side_ref, int i, int inv)
{
  // one load behind a pointer select: lanes of a row disagree about the side all the time
  const u8 *p = i >= -1 ? main_ref + (i + 1) : side_ref + ((128 + (-1 - i) * inv) >> 8);
  return *p;
}

// Value of angular mode `mode` at pixel (x, y) of a w x w block (intra-generic.c:49-155).
KVZ_DEV u8 angular_pixel(int mode, int x, int y, const u8 *above, const u8 *left)
{
  const int disp_tab[9] = { 0, 2, 5, 9, 13, 17, 21, 26, 32 };
  const int inv_tab[9] = { 0, 4096, 1638, 910, 630, 482, 390, 315, 256 };
  const bool vertical = mode >= 18;
  const int mode_disp = vertical ? mode - 26 : 10 - mode;
  const int ad = iabs(mode_disp);
  const int sample_disp = mode_disp < 0 ? -disp_tab[ad] : disp_tab[ad];
  const u8 *main_ref = vertical ? above : left, *side_ref = vertical ? left : above;
  const int px = vertical ? x : y, py = vertical ? y : x;  // horizontal modes = vertical on swapped refs, transposed
  if (sample_disp == 0) return main_ref[px + 1];
  const int delta_pos = (py + 1) * sample_disp;
  const int di = delta_pos >> 5, df = delta_pos & 31;
  const int inv = inv_tab[ad];
  const int r1 = angular_ref(main_ref, side_ref, px + di, inv);
  if (!df) return (u8)r1;
  const int r2 = angular_ref(main_ref, side_ref, px + di + 1, inv);
  return (u8)(((32 - df) * r1 + df * r2 + 16) >> 5);
}

// intra-generic.c:165-201
KVZ_DEV u8 planar_pixel(int log2w, int x, int y, const u8 *top, const u8 *left)
{
  const int w = 1 << log2w;
  const int hor = (w - 1 - x) * left[y + 1] + (x + 1) * top[w + 1];
  const int ver = (w - 1 - y) * top[x + 1] + (y + 1) * left[w + 1];
  return (u8)((ver + hor + w) >> (log2w + 1));
}

KVZ_DEV int dc_value(int log2w, const u8 *top, const u8 *left)
{
  const int w = 1 << log2w;
  int sum = 0;
  for (int i = 0; i < w; i++) sum += top[i + 1] + left[i + 1];
  return (u8)((sum + w) >> (log2w + 1));
}

// intra-generic.c:210-241 (luma DC with [1 2 1] / [1 3] boundary smoothing)
KVZ_DEV u8 filtered_dc_pixel(int dc, int x, int y, const u8 *top, const u8 *left)
{
  if (x == 0 && y == 0) return (u8)((left[1] + 2 * dc + top[1] + 2) / 4);
  if (y == 0) return (u8)((top[x + 1] + 3 * dc + 2) / 4);
  if (x == 0) return (u8)((left[y + 1] + 3 * dc + 2) / 4);
  return (u8)dc;
}

// kind: 0 angular(mode), 1 planar, 2 filtered dc.  Blocks: refs at above + blk*ref_bstride, dst contiguous w*w.
struct IntraPredOp {
  int kind, log2w; const int8_t *modes /* per block, kind 0 */; const u8 *above; const u8 *left; long ref_bstride; u8 *dst;
  KVZ_DEV void operator()(int item) const
  {
    const int w = 1 << log2w, blk = item >> (2 * log2w), e = item & (w * w - 1), y = e >> log2w, x = e & (w - 1);
    const u8 *t = above + blk * ref_bstride, *l = left + blk * ref_bstride;
    u8 v;
    if (kind == 0) v = angular_pixel(modes[blk], x, y, t, l);
    else if (kind >= 3) v = angular_pixel(kind - 3, x, y, t, l);  // one mode (kind - 3) for every block of the batch
    else if (kind == 1) v = planar_pixel(log2w, x, y, t, l);
    else v = filtered_dc_pixel(dc_value(log2w, t, l), x, y, t, l);
    dst[item] = v;
  }
};

// ------------------------------------------------------------------------------------------------
// Interpolation (ipol-generic.c).  G_f(r, c) = sum_i f[i] * S(r, c - 3 + i) is the horizontal 8-tap response
// centred on column c; every output of the sampling and FME functions is one of
//   H : fin(G_f(r, c))                       horizontal only
//   V : fin(sum_j v[j] * S(r - 3 + j, c))    vertical only
//   HV: fin((sum_j v[j] * G_f(r - 3 + j, c)) >> 6)
// with fin(t) = clip8(((int16)t + 32) >> 6)  (picture-generic.c:42-60 after the 14-bit rounding).
// ------------------------------------------------------------------------------------------------
KVZ_DEV int hor8(const int8_t *f, const u8 *src, int stride, int r, int c)
{
  const u8 *p = src + (long)r * stride + c - 3;
  int t = 0;
  for (int i = 0; i < 8; i++) t += f[i] * (int)p[i];
  return t;
}
KVZ_DEV int hv8(const int8_t *hf, const int8_t *vf, const u8 *src, int stride, int r, int c)
{
  int t = 0;
  for (int j = 0; j < 8; j++) t += vf[j] * (int)(i16)hor8(hf, src, stride, r - 3 + j, c);
  return t;
}
KVZ_DEV u8 fin14(int t) { return clip_pixel(((int)(i16)t + 32) >> 6); }

// ipol-generic.c:134-211 sample_quarterpel_luma{,_hi}; :681-758 sample_octpel_chroma{,_hi}.  One item per pixel.
struct SampleOp {
  const Tables *tb; int chroma; const u8 *src; int src_stride, w; u8 *dst8; i16 *dst16; int dst_stride; int fx, fy;
  KVZ_DEV void operator()(int item) const
  {
    const int y = item / w, x = item - y * w;
    int v = 0;
    if (!chroma) {
      v = hv8(tb->luma_filter[fx], tb->luma_filter[fy], src, src_stride, y, x) >> 6;
    } else {
      const int8_t *hf = tb->chroma_filter[fx], *vf = tb->chroma_filter[fy];
      for (int j = 0; j < 4; j++) {
        const u8 *p = src + (long)(y - 1 + j) * src_stride + x - 1;
        int t = 0;
        for (int i = 0; i < 4; i++) t += hf[i] * (int)p[i];
        v += vf[j] * (int)(i16)t;
      }
      v >>= 6;
    }
    if (dst8) dst8[y * dst_stride + x] = clip_pixel((v + 32) >> 6);
    else dst16[y * dst_stride + x] = (i16)v;
  }
};

// One output plane of an FME block filter.
struct FmePlane { int mode /* 0 H, 1 V, 2 HV */, hf, vf, roff, coff; };

// ipol-generic.c:213-679: the four filter_{hpel,qpel}_blocks_{hor_ver,diag}_luma functions, 4 planes each,
// one item per (plane, pixel).  Plane descriptors are derived on the host (kvz_fme_planes()).
struct FmeOp {
  const Tables *tb; const u8 *src; int src_stride, w, h; u8 *filtered /* 4 x 64*64, stride 64 */; FmePlane pl[4];
  KVZ_DEV void operator()(int item) const
  {
    const int per = w * h, p = item / per, e = item - p * per, y = e / w, x = e - y * w;
    const FmePlane P = pl[p];
    const int r = y + P.roff, c = x + P.coff;
    int t;
    if (P.mode == 0) t = hor8(tb->luma_filter[P.hf], src, src_stride, r, c);
    else if (P.mode == 1) {
      t = 0;
      const int8_t *vf = tb->luma_filter[P.vf];
      for (int j = 0; j < 8; j++) t += vf[j] * (int)src[(long)(r - 3 + j) * src_stride + c];
    } else t = hv8(tb->luma_filter[P.hf], tb->luma_filter[P.vf], src, src_stride, r, c) >> 6;
    filtered[p * 4096 + y * 64 + x] = fin14(t);
  }
};

// Horizontal intermediates the reference leaves in the caller's buffers (hor_intermediate[p] stride 64 and
// hor_first_cols[p]): IM[y][x] = G_f(y - 3, x + 1), COL[y] = G_f(y - 3, 0) for y in [y0, h + 8).
// One item per (row, column+1): column index 0 writes COL.
struct FmeHorOp {
  const Tables *tb; const u8 *src; int src_stride, w, h, y0, filt; i16 *im; i16 *col;
  KVZ_DEV void operator()(int item) const
  {
    const int cols = w + 1, yy = item / cols, c = item - yy * cols, y = yy + y0;
    const i16 v = (i16)hor8(tb->luma_filter[filt], src, src_stride, y - 3, c);
    if (c == 0) col[y] = v; else im[y * 64 + c - 1] = v;
  }
};

// ipol-generic.c:761-814 get_extended_block: edge-replicated window copy, one item per output byte.
struct ExtBlockOp {
  const u8 *src; int src_w, src_h, src_s, x0, y0 /* frame coords of buf[0] */, ext_s, rows /* real rows */, total_rows; u8 *buf;
  KVZ_DEV void operator()(int item) const
  {
    const int y = item / ext_s, x = item - y * ext_s;
    if (y >= rows) { buf[item] = 0; return; }  // pad_b_simd rows are zeroed
    buf[item] = src[(long)iclip(0, src_h - 1, y0 + y) * src_s + iclip(0, src_w - 1, x0 + x)];
  }
};

// ------------------------------------------------------------------------------------------------
// SAO (sao-generic.c, sao_shared_generics.h)
// ------------------------------------------------------------------------------------------------
KVZ_DEV int sgn3(int v) { return (v > 0) - (v < 0); }
KVZ_DEV int eo_cat(int a, int b, int c)
{
  const int idx = 2 + sgn3(c - a) + sgn3(c - b);  // {1,2,0,3,4}[idx]
  return idx == 0 ? 1 : idx == 1 ? 2 : idx == 2 ? 0 : idx;
}
KVZ_DEV void eo_offsets(int eo_class, int &ax, int &ay, int &bx, int &by)
{
  // sao.h:71-76 g_sao_edge_offsets
  ax = eo_class == 0 ? -1 : eo_class == 1 ? 0 : eo_class == 2 ? -1 : 1;
  ay = eo_class == 0 ? 0 : -1;
  bx = -ax; by = -ay;
}

// sao-generic.c:50-81 calc_sao_edge_dir (mode 0: stats[cat] += orig - c, stats[5+cat] += 1) and
// sao_shared_generics.h:52-91 sao_edge_ddistortion (mode 1: stats[0] += delta^2 - diff^2), interior pixels only.
struct SaoEdgeOp {
  int mode, eo_class, bw, bh; const u8 *orig; const u8 *rec; i32 offsets[5]; i32 *stats;
  KVZ_DEV void operator()(int item) const
  {
    const int iw = bw - 2, y = item / iw + 1, x = item - (y - 1) * iw + 1;
    int ax, ay, bx, by;
    eo_offsets(eo_class, ax, ay, bx, by);
    const int c = rec[y * bw + x], a = rec[(y + ay) * bw + x + ax], b = rec[(y + by) * bw + x + bx];
    const int cat = eo_cat(a, b, c), diff = (int)orig[y * bw + x] - c;
    if (mode == 0) {
      KVZ_ATOMIC_ADD(&stats[cat], diff);
      KVZ_ATOMIC_ADD(&stats[5 + cat], 1);
    } else {
      const int off = offsets[cat];
      if (off != 0) { const int d = diff - off; KVZ_ATOMIC_ADD(&stats[0], d * d - diff * diff); }
    }
  }
};

// sao_shared_generics.h:93-130 sao_band_ddistortion
struct SaoBandOp {
  int shift, band_pos; i32 bands[4]; const u8 *orig; const u8 *rec; i32 *out;
  KVZ_DEV void operator()(int i) const
  {
    const int band = ((int)rec[i] >> shift) - band_pos;
    const int off = (band >= 0 && band <= 3) ? bands[band] : 0;
    if (off != 0) { const int diff = (int)orig[i] - (int)rec[i], d = diff - off; KVZ_ATOMIC_ADD(out, d * d - diff * diff); }
  }
};

// sao-generic.c:84-124 sao_reconstruct_color (+ sao.c:180-202 band LUT evaluated per pixel)
struct SaoReconOp {
  int type, eo_class, band_pos, offset_base /* 0 or 5 (V plane) */; i32 offsets[10];
  const u8 *rec; int stride; u8 *out; int out_stride, bw;
  KVZ_DEV void operator()(int item) const
  {
    const int y = item / bw, x = item - y * bw;
    const u8 *c = rec + (long)y * stride + x;
    int v = c[0];
    if (type == 1) {
      const int d = (v >> 3) - band_pos;
      if (d >= 0 && d <= 3) v = iclip(0, 255, v + offsets[d + 1 + offset_base]);
    } else {
      int ax, ay, bx, by;
      eo_offsets(eo_class, ax, ay, bx, by);
      const int cat = eo_cat(c[ay * stride + ax], c[by * stride + bx], v);
      v = iclip(0, 255, v + offsets[cat + offset_base]);
    }
    out[(long)y * out_stride + x] = (u8)v;
  }
};

}  // namespace kvz
