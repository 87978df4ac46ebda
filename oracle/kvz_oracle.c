/*
 * kvz_oracle.c -- TEST INFRASTRUCTURE (see kvz_oracle.h).  Plain-C restatement of the generic
 * strategy functions of ultravideo/kvazaar v2.3.2.  All paths below are relative to
 * /root/reference/src.  KVZ_BIT_DEPTH is 8 throughout (kvazaar.h:90-98), so every
 * ">> (KVZ_BIT_DEPTH - 8)" of the reference is a no-op and is omitted.
 *
 * Parity status: pinned.  tests/test_oracle_golden.py checks the golden values of the reference's
 * unit tests; tests/test_oracle_vs_ref.py checks every function here bit-for-bit against the compiled
 * reference (oracle/_ref/libkvazaar_ref.so), generic and AVX2 strategies.
 */
#include "kvz_oracle.h"

#include <stdlib.h>
#include <math.h>
#include <string.h>

#define ORC_MIN(a, b) ((a) < (b) ? (a) : (b))
#define ORC_MAX(a, b) ((a) > (b) ? (a) : (b))
#define ORC_CLIP(lo, hi, v) ORC_MAX((lo), ORC_MIN((hi), (v)))

/* ------------------------------------------------------------------------------------------------
 * Tables
 * ---------------------------------------------------------------------------------------------- */

/* First column of the transposed 32-point matrix (strategies/generic/dct-generic.c:170, row k, col 0):
 * the 32 distinct magnitudes c_k*cos(k*pi/64) of the HEVC core transform.  Every entry of the
 * 32/16/8/4-point matrices (dct-generic.c:46-120) follows from these by the cosine symmetries. */
static const int16_t dct_mag[32] = { 64, 90, 90, 90, 89, 88, 87, 85, 83, 82, 80, 78, 75, 73, 70, 67,
                                     64, 61, 57, 54, 50, 46, 43, 38, 36, 31, 25, 22, 18, 13, 9,  4 };
static int16_t g_dct[4][32 * 32]; /* index 0..3 -> n = 4,8,16,32 */
static const int16_t g_dst4[16] = { 29, 55, 74, 84, 74, 74, 0, -74, 84, -29, -74, 55, 55, -84, 74, -29 }; /* dct-generic.c:38-44 */
static uint32_t g_scan[3][5][1024];
static int g_tables_ready = 0;

static int dct32_entry(int k, int n)
{
  if (k == 0) return 64;
  int m = (k * (2 * n + 1)) % 128; /* angle in units of pi/64 */
  if (m < 32) return dct_mag[m];
  if (m < 64) return -dct_mag[64 - m];
  if (m < 96) return -dct_mag[m - 64];
  return dct_mag[128 - m];
}

/* Scan tables (tables.c:9-67 kvz_g_sig_last_scan): coefficients are visited 4x4 coefficient group by
 * coefficient group; groups and the positions inside a group follow the same pattern
 * (0 = up-right diagonal, 1 = horizontal/raster, 2 = vertical/column-major). */
static void scan_pattern(int type, int size, int *xs, int *ys)
{
  int n = 0;
  if (type == 0) {
    for (int d = 0; d < 2 * size - 1; d++)
      for (int y = ORC_MIN(d, size - 1); y >= 0 && d - y < size; y--) { xs[n] = d - y; ys[n] = y; n++; }
  } else if (type == 1) {
    for (int y = 0; y < size; y++) for (int x = 0; x < size; x++) { xs[n] = x; ys[n] = y; n++; }
  } else {
    for (int x = 0; x < size; x++) for (int y = 0; y < size; y++) { xs[n] = x; ys[n] = y; n++; }
  }
}

static void init_tables(void)
{
  if (g_tables_ready) return;
  for (int t = 0; t < 4; t++) {
    int n = 4 << t, step = 32 / n;
    for (int k = 0; k < n; k++) for (int j = 0; j < n; j++) g_dct[t][k * n + j] = (int16_t)dct32_entry(k * step, j);
  }
  for (int type = 0; type < 3; type++) {
    for (int l2 = 1; l2 <= 5; l2++) {
      int size = 1 << l2;
      uint32_t *out = g_scan[type][l2 - 1];
      if (l2 == 1) { /* 2x2: tables.c g_sig_last_scan_*_0 */
        int xs[4], ys[4];
        scan_pattern(type, 2, xs, ys);
        for (int i = 0; i < 4; i++) out[i] = ys[i] * 2 + xs[i];
        continue;
      }
      int cgs = size / 4, gx[64], gy[64], px[16], py[16], n = 0;
      scan_pattern(type, cgs, gx, gy);
      scan_pattern(type, 4, px, py);
      for (int g = 0; g < cgs * cgs; g++)
        for (int i = 0; i < 16; i++) out[n++] = (gy[g] * 4 + py[i]) * size + gx[g] * 4 + px[i];
    }
  }
  g_tables_ready = 1;
}

const int16_t *kvz_oracle_dct_matrix(int n)
{
  init_tables();
  return n == 4 ? g_dct[0] : n == 8 ? g_dct[1] : n == 16 ? g_dct[2] : n == 32 ? g_dct[3] : NULL;
}
const int16_t *kvz_oracle_dst_matrix(void) { return g_dst4; }
const uint32_t *kvz_oracle_scan_table(int scan_idx, int log2_size)
{
  init_tables();
  if (scan_idx < 0 || scan_idx > 2 || log2_size < 1 || log2_size > 5) return NULL;
  return g_scan[scan_idx][log2_size - 1];
}

/* ------------------------------------------------------------------------------------------------
 * Picture: SAD / SATD / SSD   (strategies/generic/picture-generic.c)
 * ---------------------------------------------------------------------------------------------- */

/* picture-generic.c:98-111 reg_sad_generic */
unsigned kvz_oracle_reg_sad(const uint8_t *d1, const uint8_t *d2, int w, int h, unsigned s1, unsigned s2)
{
  unsigned sad = 0;
  for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) sad += abs(d1[y * s1 + x] - d2[y * s2 + x]);
  return sad;
}

/* picture-generic.c:475-501 SAD_NXN */
unsigned kvz_oracle_sad_nxn(int n, const uint8_t *b1, const uint8_t *b2)
{
  unsigned sum = 0;
  for (int i = 0; i < n * n; i++) sum += abs(b1[i] - b2[i]);
  return sum;
}

/* picture-generic.c:117-196 hadamard_4x4_generic: 4x4 Hadamard of the difference, sum of |.|, (s+1)>>1.
 * The butterfly order of the reference only permutes/negates outputs; the sum of absolute values equals
 * that of H4 * D * H4 with the natural-order Hadamard matrix. */
static int hadamard4(const int *d /* 16 */)
{
  int t[16], sum = 0;
  for (int r = 0; r < 4; r++) {
    int a = d[4 * r], b = d[4 * r + 1], c = d[4 * r + 2], e = d[4 * r + 3];
    t[4 * r] = a + b + c + e; t[4 * r + 1] = a - b + c - e; t[4 * r + 2] = a + b - c - e; t[4 * r + 3] = a - b - c + e;
  }
  for (int c = 0; c < 4; c++) {
    int a = t[c], b = t[4 + c], g = t[8 + c], e = t[12 + c];
    sum += abs(a + b + g + e) + abs(a - b + g - e) + abs(a + b - g - e) + abs(a - b - g + e);
  }
  return (sum + 1) >> 1;
}

/* picture-generic.c:212-225 kvz_satd_4x4_subblock_generic */
static unsigned satd4_sub(const uint8_t *b1, int s1, const uint8_t *b2, int s2)
{
  int d[16];
  for (int y = 0; y < 4; y++) for (int x = 0; x < 4; x++) d[4 * y + x] = b1[y * s1 + x] - b2[y * s2 + x];
  return (unsigned)hadamard4(d);
}

/* picture-generic.c:252-340 satd_8x8_subblock_generic: 8x8 Hadamard, sum |.|, (s+2)>>2 */
static unsigned satd8_sub(const uint8_t *b1, int s1, const uint8_t *b2, int s2)
{
  int m[64], sum = 0;
  for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x++) m[8 * y + x] = b1[y * s1 + x] - b2[y * s2 + x];
  for (int pass = 0; pass < 2; pass++) {
    int st = pass ? 8 : 1, ln = pass ? 1 : 8; /* pass 0: along rows, pass 1: along columns */
    for (int l = 0; l < 8; l++) {
      int *v = m + l * ln;
      for (int half = 4; half >= 1; half >>= 1)
        for (int base = 0; base < 8; base += 2 * half)
          for (int i = 0; i < half; i++) {
            int a = v[(base + i) * st], b = v[(base + i + half) * st];
            v[(base + i) * st] = a + b; v[(base + i + half) * st] = a - b;
          }
    }
  }
  for (int i = 0; i < 64; i++) sum += abs(m[i]);
  return (unsigned)((sum + 2) >> 2);
}

/* picture-generic.c:201-208 satd_4x4_generic; strategies-picture.h:53-69 SATD_NxN for n>=8 */
unsigned kvz_oracle_satd_nxn(int n, const uint8_t *b1, const uint8_t *b2)
{
  if (n == 4) return satd4_sub(b1, 4, b2, 4);
  unsigned sum = 0;
  for (int y = 0; y < n; y += 8) for (int x = 0; x < n; x += 8) sum += satd8_sub(b1 + y * n + x, n, b2 + y * n + x, n);
  return sum;
}

/* picture-generic.c:512-534 SAD_DUAL_NXN; preds is kvz_pixel(*)[32*32] (strategies-picture.h:48) */
void kvz_oracle_sad_nxn_dual(int n, const uint8_t *preds, const uint8_t *orig, unsigned num_modes, unsigned *costs_out)
{
  (void)num_modes;
  costs_out[0] = kvz_oracle_sad_nxn(n, preds, orig);
  costs_out[1] = kvz_oracle_sad_nxn(n, preds + 1024, orig);
}

/* picture-generic.c:369-402 SATD_DUAL_NXN / satd_4x4_dual_generic */
void kvz_oracle_satd_nxn_dual(int n, const uint8_t *preds, const uint8_t *orig, unsigned num_modes, unsigned *costs_out)
{
  (void)num_modes;
  costs_out[0] = kvz_oracle_satd_nxn(n, preds, orig);
  costs_out[1] = kvz_oracle_satd_nxn(n, preds + 1024, orig);
}

/* strategies-picture.h:75-113 SATD_ANY_SIZE: first column / first row in 4x4 when w%8 / h%8, rest 8x8 */
unsigned kvz_oracle_satd_any_size(int w, int h, const uint8_t *b1, int s1, const uint8_t *b2, int s2)
{
  unsigned sum = 0;
  if (w % 8 != 0) {
    for (int y = 0; y < h; y += 4) sum += satd4_sub(b1 + y * s1, s1, b2 + y * s2, s2);
    b1 += 4; b2 += 4; w -= 4;
  }
  if (h % 8 != 0) {
    for (int x = 0; x < w; x += 4) sum += satd4_sub(b1 + x, s1, b2 + x, s2);
    b1 += 4 * s1; b2 += 4 * s2; h -= 4;
  }
  for (int y = 0; y < h; y += 8) for (int x = 0; x < w; x += 8) sum += satd8_sub(b1 + y * s1 + x, s1, b2 + y * s2 + x, s2);
  return sum;
}

/* picture-generic.c:404-471 SATD_ANY_SIZE_MULTI_GENERIC(quad_generic, 4).  Restated with the reference's
 * row-offset quirk kept: after the "first row" step height has already been reduced by 4, so the 8x8 loop
 * `for (y = height % 8; ...)` starts at y = 0 (not 4) measured from the ORIGINAL block top, re-covering
 * the first 4 rows and never reaching the last 4 (picture-generic.c:446-449).  `valid` is ignored. */
void kvz_oracle_satd_any_size_quad(int w, int h, const uint8_t *const *preds, int stride, const uint8_t *orig,
                                   int orig_stride, unsigned num_modes, unsigned *costs_out, int8_t *valid)
{
  (void)num_modes; (void)valid;
  for (int b = 0; b < 4; b++) {
    unsigned cost = 0;
    int width = w, height = h;
    const int wm8 = width % 8;
    if (wm8 != 0) {
      for (int y = 0; y < height; y += 4) cost += satd4_sub(orig + y * orig_stride, orig_stride, preds[b] + y * stride, stride);
      width -= 4;
    }
    if (height % 8 != 0) {
      /* note: pred pointer restarts at preds[b] (column 0), orig at column 0 as well */
      for (int x = 0; x < width; x += 4) cost += satd4_sub(orig + x, orig_stride, preds[b] + x, stride);
      height -= 4;
    }
    for (int y = height % 8; y < height; y += 8)
      for (int x = wm8; x < width; x += 8)
        cost += satd8_sub(orig + y * orig_stride + wm8 + (x - wm8), orig_stride, preds[b] + y * stride + wm8 + (x - wm8), stride);
    costs_out[b] = cost;
  }
}

/* picture-generic.c:536-551 pixels_calc_ssd_generic */
unsigned kvz_oracle_pixels_calc_ssd(const uint8_t *ref, const uint8_t *rec, int ref_stride, int rec_stride, int width)
{
  int ssd = 0;
  for (int y = 0; y < width; y++)
    for (int x = 0; x < width; x++) { int d = ref[x + y * ref_stride] - rec[x + y * rec_stride]; ssd += d * d; }
  return (unsigned)ssd;
}

/* picture-generic.c:687-700 ver_sad_generic */
uint32_t kvz_oracle_ver_sad(const uint8_t *pic, const uint8_t *ref, int32_t bw, int32_t bh, uint32_t pic_stride)
{
  unsigned sad = 0;
  for (int y = 0; y < bh; y++) for (int x = 0; x < bw; x++) sad += abs(pic[y * pic_stride + x] - ref[x]);
  return sad;
}

/* picture-generic.c:713-726 hor_sad */
static unsigned hor_sad_col(const uint8_t *pic, const uint8_t *ref, int bw, int bh, unsigned pic_stride, unsigned ref_stride)
{
  unsigned sad = 0;
  for (int y = 0; y < bh; y++) for (int x = 0; x < bw; x++) sad += abs(pic[y * pic_stride + x] - ref[y * ref_stride]);
  return sad;
}

/* picture-generic.c:729-752 hor_sad_generic */
uint32_t kvz_oracle_hor_sad(const uint8_t *pic, const uint8_t *ref, int32_t w, int32_t h, uint32_t pic_stride,
                            uint32_t ref_stride, uint32_t left, uint32_t right)
{
  uint32_t r = 0;
  if (left) {
    r += hor_sad_col(pic, ref + left, left, h, pic_stride, ref_stride);
    r += kvz_oracle_reg_sad(pic + left, ref + left, w - left, h, pic_stride, ref_stride);
  } else if (right) {
    r += kvz_oracle_reg_sad(pic, ref, w - right, h, pic_stride, ref_stride);
    r += hor_sad_col(pic + w - right, ref + w - right - 1, right, h, pic_stride, ref_stride);
  } else {
    r += kvz_oracle_reg_sad(pic, ref, w, h, pic_stride, ref_stride);
  }
  return r;
}

/* picture-generic.c:755-778 pixel_var_generic (double; same operation order) */
double kvz_oracle_pixel_var(const uint8_t *arr, uint32_t len)
{
  double var = 0, sum = 0;
  for (uint32_t i = 0; i < len; i++) sum += arr[i];
  double mean = sum / (double)len;
  for (uint32_t i = 0; i < len; i++) { double t = (double)arr[i] - mean; var += t * t; }
  return var / len;
}

/* picture-generic.c:62-80 kvz_fast_clip_32bit_to_pixel / :42-60 16-bit variant: 0 below, 255 above */
static uint8_t clip_pixel(int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }

/* picture-generic.c:553-614 bipred_average_{px_px,im_im,px_im}: (a + b + 64) >> 7 on 14-bit samples */
void kvz_oracle_bipred_average_plane(uint8_t *dst, unsigned dst_stride, const uint8_t *px0, const int16_t *im0,
                                     const uint8_t *px1, const int16_t *im1, unsigned w, unsigned h)
{
  for (unsigned i = 0; i < w * h; i++) {
    unsigned y = i / w, x = i % w;
    int16_t a = px0 ? (int16_t)(px0[i] << 6) : im0[i];
    int16_t b = px1 ? (int16_t)(px1[i] << 6) : im1[i];
    dst[y * dst_stride + x] = clip_pixel((a + b + 64) >> 7);
  }
}

/* ------------------------------------------------------------------------------------------------
 * DCT / DST   (strategies/generic/dct-generic.c)
 * The partial butterflies (:255-577) are exact integer regroupings of a matrix product (no int32
 * overflow is reachable: 32 * 90 * 32768 < 2^31), so both passes are restated as plain products.
 *   forward pass (:255-279 etc.): dst[k*N + j] = (short)((sum_n C[k][n]*src[j*N+n] + add) >> shift)  -- wraps, no clip
 *   inverse pass (:281-309 etc.): dst[j*N + n] = clip16((sum_k C[k][n]*src[k*N+j] + add) >> shift)
 * ---------------------------------------------------------------------------------------------- */
static void fwd_pass(const int16_t *C, int n, const int16_t *src, int16_t *dst, int shift)
{
  const int add = 1 << (shift - 1);
  for (int j = 0; j < n; j++)
    for (int k = 0; k < n; k++) {
      int s = 0;
      for (int i = 0; i < n; i++) s += C[k * n + i] * src[j * n + i];
      dst[k * n + j] = (int16_t)((s + add) >> shift);
    }
}
static void inv_pass(const int16_t *C, int n, const int16_t *src, int16_t *dst, int shift)
{
  const int add = 1 << (shift - 1);
  for (int j = 0; j < n; j++)
    for (int i = 0; i < n; i++) {
      int s = 0;
      for (int k = 0; k < n; k++) s += C[k * n + i] * src[k * n + j];
      dst[j * n + i] = (int16_t)ORC_CLIP(-32768, 32767, (s + add) >> shift);
    }
}

/* dct-generic.c:579-629: shifts log2N-1+(bd-8), log2N+6 forward; 7, 12-(bd-8) inverse */
void kvz_oracle_transform(int kind, int8_t bitdepth, const int16_t *in, int16_t *out)
{
  static const int sizes[5] = { 4, 8, 16, 32, 4 };
  int16_t tmp[32 * 32];
  const int inverse = kind >= KVZ_HIP_IDCT_4;
  const int idx = inverse ? kind - KVZ_HIP_IDCT_4 : kind;
  const int n = sizes[idx];
  const int16_t *C = idx == 4 ? g_dst4 : kvz_oracle_dct_matrix(n);
  int log2n = n == 4 ? 2 : n == 8 ? 3 : n == 16 ? 4 : 5;
  if (!inverse) {
    fwd_pass(C, n, in, tmp, log2n - 1 + (bitdepth - 8));
    fwd_pass(C, n, tmp, out, log2n + 6);
  } else {
    inv_pass(C, n, in, tmp, 7);
    inv_pass(C, n, tmp, out, 12 - (bitdepth - 8));
  }
}

/* ------------------------------------------------------------------------------------------------
 * Quantisation   (strategies/generic/quant-generic.c, transform.c)
 * ---------------------------------------------------------------------------------------------- */
static const int16_t g_quant_scales[6] = { 26214, 23302, 20560, 18396, 16384, 14564 }; /* scalinglist.c:78 */
static const int16_t g_inv_quant_scales[6] = { 40, 45, 51, 57, 64, 72 };               /* scalinglist.c:79 */
static const uint8_t g_chroma_scale[58] = { /* transform.c:56-62 */
  0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 29, 30, 31, 32,
  33, 33, 34, 34, 35, 35, 36, 36, 37, 37, 38, 39, 40, 41, 42, 43, 44, 45, 46, 47, 48, 49, 50, 51 };

/* transform.c:141-155 kvz_get_scaled_qp */
int32_t kvz_oracle_get_scaled_qp(int8_t type, int8_t qp, int8_t qp_offset)
{
  if (type == 0) return qp + qp_offset;
  int32_t q = ORC_CLIP(-qp_offset, 57, qp);
  return q < 0 ? q + qp_offset : g_chroma_scale[q] + qp_offset;
}

static int log2_of(int w) { return w == 4 ? 2 : w == 8 ? 3 : w == 16 ? 4 : w == 32 ? 5 : w == 2 ? 1 : 6; }

/* quant-generic.c:50-180 kvz_quant_generic (incl. sign-bit hiding :84-179) */
void kvz_oracle_quant(const kvz_hip_quant_params *p, const int16_t *coef, int16_t *q_coef, int32_t width,
                      int32_t height, int8_t type, int8_t scan_idx, int8_t block_type)
{
  (void)block_type;
  const int log2_tr = log2_of(width);
  const uint32_t *scan = kvz_oracle_scan_table(scan_idx, log2_tr);
  const int32_t qp_scaled = kvz_oracle_get_scaled_qp(type, (int8_t)p->qp, (int8_t)((p->bitdepth - 8) * 6));
  const int32_t transform_shift = 15 - p->bitdepth - log2_tr;
  const int32_t q_bits = 14 + qp_scaled / 6 + transform_shift;
  const int32_t add = (p->slice_is_intra ? 171 : 85) << (q_bits - 9);
  const int32_t q_bits8 = q_bits - 8;
  const int32_t flat_q = g_quant_scales[qp_scaled % 6];
  uint32_t ac_sum = 0;

  for (int n = 0; n < width * height; n++) {
    int32_t level = coef[n];
    int64_t abs_level = (int64_t)abs(level);
    int32_t sign = level < 0 ? -1 : 1;
    int32_t q = p->quant_coeff ? p->quant_coeff[n] : flat_q;
    level = (int32_t)((abs_level * q + add) >> q_bits);
    ac_sum += level;
    level *= sign;
    q_coef[n] = (int16_t)ORC_CLIP(-32768, 32767, level);
  }
  if (!p->signhide || ac_sum < 2) return;

  int32_t delta_u[32 * 32];
  for (int n = 0; n < width * height; n++) {
    int64_t abs_level = (int64_t)abs((int32_t)coef[n]);
    int32_t q = p->quant_coeff ? p->quant_coeff[n] : flat_q;
    int32_t level = (int32_t)((abs_level * q + add) >> q_bits);
    delta_u[n] = (int32_t)((abs_level * q - ((int64_t)level << q_bits)) >> q_bits8);
  }
  /* NOTE: the reference computes (level << q_bits) in int32 (quant-generic.c:93); for |coef| <= 32767
   * and the reachable q_bits (<= 29) level << q_bits <= abs_level*q + add < 2^31 never overflows, so the
   * 64-bit form above is identical. */
  int last_cg = -1;
  for (int subset = (width * height - 1) >> 4; subset >= 0; subset--) {
    int first_nz = 16, last_nz = -1, subpos = subset << 4, abssum = 0, n;
    for (n = 15; n >= 0; n--) if (q_coef[scan[n + subpos]]) { last_nz = n; break; }
    for (n = 0; n < 16; n++) if (q_coef[scan[n + subpos]]) { first_nz = n; break; }
    for (n = first_nz; n <= last_nz; n++) abssum += q_coef[scan[n + subpos]];
    if (last_nz >= 0 && last_cg == -1) last_cg = 1;
    if (last_nz - first_nz >= 4) {
      int32_t signbit = q_coef[scan[subpos + first_nz]] > 0 ? 0 : 1;
      if (signbit != (abssum & 1)) {
        int32_t min_cost_inc = 0x7fffffff, min_pos = -1, cur_cost = 0x7fffffff;
        int16_t final_change = 0, cur_change = 0;
        for (n = (last_cg == 1 ? last_nz : 15); n >= 0; n--) {
          uint32_t blk = scan[n + subpos];
          if (q_coef[blk] != 0) {
            if (delta_u[blk] > 0) { cur_cost = -delta_u[blk]; cur_change = 1; }
            else if (n == first_nz && abs(q_coef[blk]) == 1) { cur_cost = 0x7fffffff; }
            else { cur_cost = delta_u[blk]; cur_change = -1; }
          } else if (n < first_nz && ((coef[blk] >= 0) ? 0 : 1) != signbit) {
            cur_cost = 0x7fffffff;
          } else { cur_cost = -delta_u[blk]; cur_change = 1; }
          if (cur_cost < min_cost_inc) { min_cost_inc = cur_cost; final_change = cur_change; min_pos = (int32_t)blk; }
        }
        if (q_coef[min_pos] == 32767 || q_coef[min_pos] == -32768) final_change = -1;
        if (coef[min_pos] >= 0) q_coef[min_pos] += final_change; else q_coef[min_pos] -= final_change;
      }
    }
    if (last_cg == 1) last_cg = 0;
  }
}

/* quant-generic.c:298-340 kvz_dequant_generic */
void kvz_oracle_dequant(const kvz_hip_quant_params *p, const int16_t *q_coef, int16_t *coef, int32_t width,
                        int32_t height, int8_t type, int8_t block_type)
{
  (void)block_type;
  const int log2_tr = log2_of(width);
  const int32_t transform_shift = 15 - p->bitdepth - log2_tr;
  const int32_t qp_scaled = kvz_oracle_get_scaled_qp(type, (int8_t)p->qp, (int8_t)((p->bitdepth - 8) * 6));
  int32_t shift = 20 - 14 - transform_shift;
  if (p->scaling_list) {
    shift += 4;
    if (shift > qp_scaled / 6) {
      int32_t add = 1 << (shift - qp_scaled / 6 - 1);
      for (int n = 0; n < width * height; n++) {
        int32_t c = ((q_coef[n] * p->dequant_coeff[n]) + add) >> (shift - qp_scaled / 6);
        coef[n] = (int16_t)ORC_CLIP(-32768, 32767, c);
      }
    } else {
      for (int n = 0; n < width * height; n++) {
        int32_t c = ORC_CLIP(-32768, 32767, q_coef[n] * p->dequant_coeff[n]);
        coef[n] = (int16_t)ORC_CLIP(-32768, 32767, c << (qp_scaled / 6 - shift));
      }
    }
  } else {
    int32_t scale = g_inv_quant_scales[qp_scaled % 6] << (qp_scaled / 6);
    int32_t add = 1 << (shift - 1);
    for (int n = 0; n < width * height; n++) {
      int32_t c = (q_coef[n] * scale + add) >> shift;
      coef[n] = (int16_t)ORC_CLIP(-32768, 32767, c);
    }
  }
}

/* transform.c:164-196 kvz_transformskip / kvz_itransformskip */
static void transformskip(int bitdepth, const int16_t *block, int16_t *coeff, int n)
{
  int shift = 15 - bitdepth - log2_of(n);
  for (int i = 0; i < n * n; i++) coeff[i] = (int16_t)((uint16_t)block[i] << shift);
}
static void itransformskip(int bitdepth, int16_t *block, const int16_t *coeff, int n)
{
  int shift = 15 - bitdepth - log2_of(n), offset = 1 << (shift - 1);
  for (int i = 0; i < n * n; i++) block[i] = (int16_t)((coeff[i] + offset) >> shift);
}

/* quant-generic.c:198-292 kvz_quantize_residual_generic, rdoq disabled (kvz_rdoq stays on the host,
 * SURVEY.md 8a).  Transform choice follows strategies-dct.c:78-116: 4x4 intra luma uses the DST. */
int kvz_oracle_quantize_residual(const kvz_hip_quant_params *p, int width, int color, int scan_order,
                                 int use_trskip, int in_stride, int out_stride, const uint8_t *ref_in,
                                 const uint8_t *pred_in, uint8_t *rec_out, int16_t *coeff_out, int early_skip)
{
  int16_t residual[32 * 32], coeff[32 * 32] = { 0 };  /* (zeroed only to keep -Wmaybe-uninitialized quiet: every used entry is written below) */
  int has_coeffs = 0;
  for (int y = 0; y < width; y++)
    for (int x = 0; x < width; x++) residual[x + y * width] = (int16_t)(ref_in[x + y * in_stride] - pred_in[x + y * in_stride]);

  const int idx = width == 4 ? ((color == 0 && p->cu_is_intra) ? 4 : 0) : width == 8 ? 1 : width == 16 ? 2 : 3;
  if (use_trskip) transformskip(p->bitdepth, residual, coeff, width);
  else kvz_oracle_transform(idx, (int8_t)p->bitdepth, residual, coeff);

  kvz_oracle_quant(p, coeff, coeff_out, width, width, color == 0 ? 0 : 2, (int8_t)scan_order, (int8_t)(p->cu_is_intra ? 1 : 2));
  for (int i = 0; i < width * width; i++) if (coeff_out[i] != 0) { has_coeffs = 1; break; }

  if (has_coeffs && !early_skip) {
    kvz_oracle_dequant(p, coeff_out, coeff, width, width, color == 0 ? 0 : (color == 1 ? 2 : 3), (int8_t)(p->cu_is_intra ? 1 : 2));
    if (use_trskip) itransformskip(p->bitdepth, residual, coeff, width);
    else kvz_oracle_transform(KVZ_HIP_IDCT_4 + idx, (int8_t)p->bitdepth, coeff, residual);
    for (int y = 0; y < width; y++)
      for (int x = 0; x < width; x++) {
        int16_t val = (int16_t)(residual[x + y * width] + pred_in[x + y * in_stride]);
        rec_out[x + y * out_stride] = (uint8_t)ORC_CLIP(0, 255, val);
      }
  } else if (rec_out != pred_in) {
    for (int y = 0; y < width; y++) for (int x = 0; x < width; x++) rec_out[x + y * out_stride] = pred_in[x + y * in_stride];
  }
  return has_coeffs;
}

/* quant-generic.c:342-349 */
/* nal-generic.c:57-82 array_checksum_generic: the picture-hash SEI's per-plane checksum (returned as the 32-bit value the
 * reference then stores big-endian in checksum_out[0..3]) */
/* nal-generic.c:41-55 array_md5_generic: MD5 (RFC 1321; the reference uses extras/libmd5.c) of width * height contiguous bytes.  Restated
 * from the RFC: four rounds of 16 steps with T[i] = floor(2^32 |sin(i + 1)|), message words little-endian, length appended in bits. */
void kvz_oracle_plane_md5(const uint8_t *data, int height, int width, int stride, uint8_t *out16)
{
  static const uint8_t S[64] = { 7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 5, 9, 14, 20, 5, 9, 14, 20, 5, 9, 14, 20, 5, 9, 14, 20,
                                 4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21 };
  uint32_t T[64], h[4] = { 0x67452301u, 0xefcdab89u, 0x98badcfeu, 0x10325476u };
  for (int i = 0; i < 64; i++) T[i] = (uint32_t)(uint64_t)floor(fabs(sin((double)(i + 1))) * 4294967296.0);
  (void)stride;
  const uint64_t len = (uint64_t)width * height, padded = ((len + 8) / 64 + 1) * 64;
  for (uint64_t off = 0; off < padded; off += 64) {
    uint8_t blk[64];
    for (int i = 0; i < 64; i++) {
      const uint64_t p = off + i;
      blk[i] = p < len ? data[p] : (p == len ? 0x80 : (p >= padded - 8 ? (uint8_t)((len * 8) >> (8 * (p - (padded - 8)))) : 0));
    }
    uint32_t M[16], a = h[0], b = h[1], c = h[2], d = h[3];
    for (int i = 0; i < 16; i++) M[i] = (uint32_t)blk[4 * i] | ((uint32_t)blk[4 * i + 1] << 8) | ((uint32_t)blk[4 * i + 2] << 16) | ((uint32_t)blk[4 * i + 3] << 24);
    for (int i = 0; i < 64; i++) {
      uint32_t f;
      int g;
      switch (i >> 4) {
        case 0: f = (b & c) | (~b & d); g = i; break;
        case 1: f = (d & b) | (~d & c); g = (5 * i + 1) % 16; break;
        case 2: f = b ^ c ^ d; g = (3 * i + 5) % 16; break;
        default: f = c ^ (b | ~d); g = (7 * i) % 16; break;
      }
      const uint32_t x = a + f + T[i] + M[g], tmp = d;
      d = c; c = b; b = b + ((x << S[i]) | (x >> (32 - S[i]))); a = tmp;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d;
  }
  for (int i = 0; i < 16; i++) out16[i] = (uint8_t)(h[i >> 2] >> (8 * (i & 3)));
}

uint32_t kvz_oracle_plane_checksum(const uint8_t *data, int height, int width, int stride)
{
  uint32_t sum = 0;
  for (int y = 0; y < height; y++)
    for (int x = 0; x < width; x++) sum += (uint32_t)(data[y * stride + x] ^ (uint8_t)((x & 0xff) ^ (y & 0xff) ^ (x >> 8) ^ (y >> 8)));
  return sum;
}

uint32_t kvz_oracle_coeff_abs_sum(const int16_t *coeffs, size_t length)
{
  uint32_t sum = 0;
  for (size_t i = 0; i < length; i++) sum += abs(coeffs[i]);
  return sum;
}

/* quant-generic.c:351-375 fast_coeff_cost_generic: weights = 4 x Q8.8 packed in a u64 */
double kvz_oracle_fast_coeff_cost(const int16_t *coeff, int32_t width, uint64_t weights)
{
  uint32_t sum = 0;
  for (int i = 0; i < width * width; i++) {
    uint32_t a = (uint32_t)abs(coeff[i]);
    if (a > 3) a = 3;
    sum += (uint32_t)((weights >> (16 * a)) & 0xffff);
  }
  return (double)sum / 256.0;
}

/* quant-generic.c:379-399 find_last_scanpos_generic */
void kvz_oracle_find_last_scanpos(const int16_t *coef, int16_t *dest_coeff, int8_t type, int32_t q_bits,
                                  const int16_t *quant_coeff, int32_t *sig_coeff_inc_out, uint32_t cg_size,
                                  uint16_t *ctx_set, const uint32_t *scan, int32_t *cg_last_scanpos,
                                  int32_t *last_scanpos, uint32_t cg_num, int32_t *cg_scanpos, int32_t width,
                                  int8_t scan_mode)
{
  (void)width; (void)scan_mode;
  for (*cg_scanpos = (int32_t)cg_num - 1; *cg_scanpos >= 0; (*cg_scanpos)--) {
    for (int32_t in_cg = (int32_t)cg_size - 1; in_cg >= 0; in_cg--) {
      int32_t scanpos = *cg_scanpos * (int32_t)cg_size + in_cg;
      uint32_t blkpos = scan[scanpos];
      int32_t q = quant_coeff[blkpos];
      int32_t level_double = coef[blkpos];
      level_double = ORC_MIN(abs(level_double) * q, 0x7fffffff - (1 << (q_bits - 1)));
      uint32_t max_abs_level = (uint32_t)((level_double + (1 << (q_bits - 1))) >> q_bits);
      if (max_abs_level > 0) {
        *last_scanpos = scanpos;
        *ctx_set = (scanpos > 0 && type == 0) ? 2 : 0;
        *cg_last_scanpos = *cg_scanpos;
        sig_coeff_inc_out[blkpos] = 0;
        return;
      }
      dest_coeff[blkpos] = 0;
    }
  }
}

/* ------------------------------------------------------------------------------------------------
 * Intra prediction   (strategies/generic/intra-generic.c)
 * ---------------------------------------------------------------------------------------------- */

/* intra-generic.c:49-155 kvz_angular_pred_generic */
void kvz_oracle_angular_pred(int log2_width, int intra_mode, const uint8_t *in_ref_above, const uint8_t *in_ref_left, uint8_t *dst)
{
  static const int8_t disp_tab[9] = { 0, 2, 5, 9, 13, 17, 21, 26, 32 };
  static const int16_t inv_tab[9] = { 0, 4096, 1638, 910, 630, 482, 390, 315, 256 };
  uint8_t tmp_ref[2 * 32];
  const int width = 1 << log2_width;
  const int vertical = intra_mode >= 18;
  const int mode_disp = vertical ? intra_mode - 26 : 10 - intra_mode;
  const int sample_disp = (mode_disp < 0 ? -1 : 1) * disp_tab[abs(mode_disp)];
  const uint8_t *ref_main, *ref_side;

  if (sample_disp < 0) {
    ref_side = (vertical ? in_ref_left : in_ref_above) + 1;
    ref_main = (vertical ? in_ref_above : in_ref_left) + 1;
    for (int x = -1; x < width; x++) tmp_ref[x + width] = ref_main[x];
    ref_main = &tmp_ref[width];
    int col_disp = 128;
    const int inv = inv_tab[abs(mode_disp)];
    const int most_negative = (width * sample_disp) >> 5;
    for (int x = -2; x >= most_negative; x--) {
      col_disp += inv;
      tmp_ref[x + width] = ref_side[(col_disp >> 8) - 1];
    }
  } else {
    ref_main = (vertical ? in_ref_above : in_ref_left) + 1;
  }

  if (sample_disp != 0) {
    int delta_pos = 0;
    for (int y = 0; y < width; y++) {
      delta_pos += sample_disp;
      const int di = delta_pos >> 5, df = delta_pos & 31;
      for (int x = 0; x < width; x++) {
        if (df) dst[y * width + x] = (uint8_t)(((32 - df) * ref_main[x + di] + df * ref_main[x + di + 1] + 16) >> 5);
        else dst[y * width + x] = ref_main[x + di];
      }
    }
  } else {
    for (int y = 0; y < width; y++) for (int x = 0; x < width; x++) dst[y * width + x] = ref_main[x];
  }
  if (!vertical) {
    for (int y = 0; y < width - 1; y++)
      for (int x = y + 1; x < width; x++) { uint8_t t = dst[y * width + x]; dst[y * width + x] = dst[x * width + y]; dst[x * width + y] = t; }
  }
}

/* intra-generic.c:165-201 kvz_intra_pred_planar_generic */
void kvz_oracle_intra_pred_planar(int log2_width, const uint8_t *ref_top, const uint8_t *ref_left, uint8_t *dst)
{
  const int width = 1 << log2_width;
  const int top_right = ref_top[width + 1], bottom_left = ref_left[width + 1];
  for (int y = 0; y < width; y++)
    for (int x = 0; x < width; x++) {
      int hor = (width - 1 - x) * ref_left[y + 1] + (x + 1) * top_right;
      int ver = (width - 1 - y) * ref_top[x + 1] + (y + 1) * bottom_left;
      dst[y * width + x] = (uint8_t)((ver + hor + width) >> (log2_width + 1));
    }
}

/* intra-generic.c:210-241 kvz_intra_pred_filtered_dc_generic */
void kvz_oracle_intra_pred_filtered_dc(int log2_width, const uint8_t *ref_top, const uint8_t *ref_left, uint8_t *out)
{
  const int width = 1 << log2_width;
  int sum = 0;
  for (int i = 0; i < width; i++) sum += ref_top[i + 1] + ref_left[i + 1];
  const int dc = (uint8_t)((sum + width) >> (log2_width + 1));
  out[0] = (uint8_t)((ref_left[1] + 2 * dc + ref_top[1] + 2) / 4);
  for (int x = 1; x < width; x++) out[x] = (uint8_t)((ref_top[x + 1] + 3 * dc + 2) / 4);
  for (int y = 1; y < width; y++) {
    out[y * width] = (uint8_t)((ref_left[y + 1] + 3 * dc + 2) / 4);
    for (int x = 1; x < width; x++) out[y * width + x] = (uint8_t)dc;
  }
}

/* ------------------------------------------------------------------------------------------------
 * Interpolation   (strategies/generic/ipol-generic.c, filter.c:66-84)
 * ---------------------------------------------------------------------------------------------- */
static const int8_t g_luma_filter[4][8] = { { 0, 0, 0, 64, 0, 0, 0, 0 }, { -1, 4, -10, 58, 17, -5, 1, 0 },
                                            { -1, 4, -11, 40, 40, -11, 4, -1 }, { 0, 1, -5, 17, 58, -10, 4, -1 } };
static const int8_t g_chroma_filter[8][4] = { { 0, 64, 0, 0 }, { -2, 58, 10, -2 }, { -4, 54, 16, -2 }, { -6, 46, 28, -4 },
                                              { -4, 36, 36, -4 }, { -4, 28, 46, -6 }, { -2, 16, 54, -4 }, { -2, 10, 58, -2 } };

static int32_t tap8_px(const int8_t *f, const uint8_t *d, int stride) { int32_t t = 0; for (int i = 0; i < 8; i++) t += f[i] * d[i * stride]; return t; }
static int32_t tap8_im(const int8_t *f, const int16_t *d, int stride) { int32_t t = 0; for (int i = 0; i < 8; i++) t += f[i] * d[i * stride]; return t; }
static int32_t tap4_px(const int8_t *f, const uint8_t *d, int stride) { int32_t t = 0; for (int i = 0; i < 4; i++) t += f[i] * d[i * stride]; return t; }
static int32_t tap4_im(const int8_t *f, const int16_t *d, int stride) { int32_t t = 0; for (int i = 0; i < 4; i++) t += f[i] * d[i * stride]; return t; }
/* picture-generic.c:42-60 kvz_fast_clip_16bit_to_pixel on an int16 value */
static uint8_t clip16_pixel(int16_t v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }

/* ipol-generic.c:134-211 sample_quarterpel_luma{,_hi}_generic */
static void qpel_luma_common(const uint8_t *src, int src_stride, int w, int h, uint8_t *dst8, int16_t *dst16, int dst_stride, const int16_t mv[2])
{
  const int8_t *hf = g_luma_filter[mv[0] & 3], *vf = g_luma_filter[mv[1] & 3];
  static int16_t hor[71][64];
  for (int y = 0; y < h + 7; y++)
    for (int x = 0; x < w; x++) hor[y][x] = (int16_t)tap8_px(hf, &src[src_stride * (y - 3) + x - 3], 1);
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      int32_t v = tap8_im(vf, &hor[y][x], 64) >> 6;
      if (dst8) dst8[y * dst_stride + x] = clip_pixel((v + 32) >> 6);
      else dst16[y * dst_stride + x] = (int16_t)v;
    }
}
void kvz_oracle_sample_quarterpel_luma(const uint8_t *src, int16_t src_stride, int w, int h, uint8_t *dst, int16_t dst_stride, int8_t hor_flag, int8_t ver_flag, const int16_t mv[2])
{ (void)hor_flag; (void)ver_flag; qpel_luma_common(src, src_stride, w, h, dst, NULL, dst_stride, mv); }
void kvz_oracle_sample_quarterpel_luma_hi(const uint8_t *src, int16_t src_stride, int w, int h, int16_t *dst, int16_t dst_stride, int8_t hor_flag, int8_t ver_flag, const int16_t mv[2])
{ (void)hor_flag; (void)ver_flag; qpel_luma_common(src, src_stride, w, h, NULL, dst, dst_stride, mv); }

/* ipol-generic.c:681-758 sample_octpel_chroma{,_hi}_generic */
static void opel_chroma_common(const uint8_t *src, int src_stride, int w, int h, uint8_t *dst8, int16_t *dst16, int dst_stride, const int16_t mv[2])
{
  const int8_t *hf = g_chroma_filter[mv[0] & 7], *vf = g_chroma_filter[mv[1] & 7];
  static int16_t hor[35][32];
  for (int y = 0; y < h + 3; y++)
    for (int x = 0; x < w; x++) hor[y][x] = (int16_t)tap4_px(hf, &src[src_stride * (y - 1) + x - 1], 1);
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      int32_t v = tap4_im(vf, &hor[y][x], 32) >> 6;
      if (dst8) dst8[y * dst_stride + x] = clip_pixel((v + 32) >> 6);
      else dst16[y * dst_stride + x] = (int16_t)v;
    }
}
void kvz_oracle_sample_octpel_chroma(const uint8_t *src, int16_t src_stride, int w, int h, uint8_t *dst, int16_t dst_stride, int8_t hor_flag, int8_t ver_flag, const int16_t mv[2])
{ (void)hor_flag; (void)ver_flag; opel_chroma_common(src, src_stride, w, h, dst, NULL, dst_stride, mv); }
void kvz_oracle_sample_octpel_chroma_hi(const uint8_t *src, int16_t src_stride, int w, int h, int16_t *dst, int16_t dst_stride, int8_t hor_flag, int8_t ver_flag, const int16_t mv[2])
{ (void)hor_flag; (void)ver_flag; opel_chroma_common(src, src_stride, w, h, NULL, dst, dst_stride, mv); }

#define IM(p) (hor_intermediate + (p) * KVZ_HIP_IPOL_IM_PLANE)
#define COL(p) (hor_first_cols + (p) * KVZ_HIP_IPOL_COL_LEN)
#define FIL(p) (filtered + (p) * 64 * 64)
static uint8_t fin(int32_t v14) { return clip16_pixel((int16_t)(((int16_t)v14 + 32) >> 6)); }

/* ipol-generic.c:213-326 kvz_filter_hpel_blocks_hor_ver_luma_generic */
void kvz_oracle_filter_hpel_blocks_hor_ver_luma(const uint8_t *src, int16_t src_stride, int w, int h, uint8_t *filtered,
                                                int16_t *hor_intermediate, int8_t fme_level, int16_t *hor_first_cols,
                                                int8_t hpel_off_x, int8_t hpel_off_y)
{
  (void)hpel_off_x; (void)hpel_off_y;
  const int8_t *fir0 = g_luma_filter[0], *fir2 = g_luma_filter[2];
  const int first_y = fme_level > 1 ? 0 : 1;
  int x, y;
  for (y = 0; y < h + 8; y++) for (x = 0; x < w; x++) IM(0)[y * 64 + x] = (int16_t)tap8_px(fir0, &src[src_stride * (y - 3) + x - 2], 1);
  for (y = 0; y < h + 8; y++) COL(0)[y] = (int16_t)tap8_px(fir0, &src[src_stride * (y - 3) - 3], 1);
  for (y = first_y; y < h + 8; y++) for (x = 0; x < w; x++) IM(1)[y * 64 + x] = (int16_t)tap8_px(fir2, &src[src_stride * (y - 3) + x - 2], 1);
  for (y = first_y; y < h + 8; y++) COL(2)[y] = (int16_t)tap8_px(fir2, &src[src_stride * (y - 3) - 3], 1);
  /* right, left */
  for (y = 0; y < h; y++) for (x = 0; x < w; x++) FIL(1)[y * 64 + x] = fin(IM(1)[4 * 64 + y * 64 + x]);
  for (y = 0; y < h; y++) {
    FIL(0)[y * 64] = fin(COL(2)[y + 4]);
    for (x = 1; x < w; x++) FIL(0)[y * 64 + x] = FIL(1)[y * 64 + x - 1];
  }
  /* top, bottom */
  for (y = 0; y < h; y++) for (x = 0; x < w; x++) FIL(2)[y * 64 + x] = fin((int16_t)tap8_px(fir2, &src[src_stride * (y - 3) + x + 1], src_stride));
  for (y = 0; y < h - 1; y++) for (x = 0; x < w; x++) FIL(3)[y * 64 + x] = FIL(2)[(y + 1) * 64 + x];
  for (x = 0; x < w; x++) FIL(3)[y * 64 + x] = fin((int16_t)tap8_px(fir2, &src[src_stride * (y - 3 + 1) + x + 1], src_stride));
}

/* ipol-generic.c:328-407 kvz_filter_hpel_blocks_diag_luma_generic */
void kvz_oracle_filter_hpel_blocks_diag_luma(const uint8_t *src, int16_t src_stride, int w, int h, uint8_t *filtered,
                                             int16_t *hor_intermediate, int8_t fme_level, int16_t *hor_first_cols,
                                             int8_t hpel_off_x, int8_t hpel_off_y)
{
  (void)src; (void)src_stride; (void)fme_level; (void)hpel_off_x; (void)hpel_off_y;
  const int8_t *fir2 = g_luma_filter[2];
  int x, y;
  for (y = 0; y < h; y++) for (x = 0; x < w; x++) FIL(1)[y * 64 + x] = fin(tap8_im(fir2, &IM(1)[y * 64 + x], 64) >> 6);
  for (y = 0; y < h; y++) {
    FIL(0)[y * 64] = fin(tap8_im(fir2, &COL(2)[y], 1) >> 6);
    for (x = 1; x < w; x++) FIL(0)[y * 64 + x] = FIL(1)[y * 64 + x - 1];
  }
  for (y = 0; y < h - 1; y++) for (x = 0; x < w; x++) FIL(3)[y * 64 + x] = FIL(1)[(y + 1) * 64 + x];
  for (x = 0; x < w; x++) FIL(3)[y * 64 + x] = fin(tap8_im(fir2, &IM(1)[(y + 1) * 64 + x], 64) >> 6);
  for (y = 0; y < h - 1; y++) for (x = 0; x < w; x++) FIL(2)[y * 64 + x] = FIL(0)[(y + 1) * 64 + x];
  for (x = 1; x < w; x++) FIL(2)[y * 64 + x] = FIL(3)[y * 64 + x - 1];
  FIL(2)[y * 64] = fin(tap8_im(fir2, &COL(2)[y + 1], 1) >> 6);
}

/* ipol-generic.c:409-567 kvz_filter_qpel_blocks_hor_ver_luma_generic */
void kvz_oracle_filter_qpel_blocks_hor_ver_luma(const uint8_t *src, int16_t src_stride, int w, int h, uint8_t *filtered,
                                                int16_t *hor_intermediate, int8_t fme_level, int16_t *hor_first_cols,
                                                int8_t hpel_off_x, int8_t hpel_off_y)
{
  (void)fme_level;
  const int8_t *fir0 = g_luma_filter[0], *fir1 = g_luma_filter[1], *fir2 = g_luma_filter[2], *fir3 = g_luma_filter[3];
  const int8_t *hor_fir_l = hpel_off_x != 0 ? fir1 : fir3, *hor_fir_r = hpel_off_x != 0 ? fir3 : fir1;
  int16_t *hor_pos_l = IM(3), *hor_pos_r = IM(4), *col_pos_l = COL(1), *col_pos_r = COL(3);
  const int16_t *hor_hpel_pos = hpel_off_x != 0 ? IM(1) : IM(0);
  const int16_t *col_pos_hor = hpel_off_x != 0 ? COL(2) : COL(0);
  const int off_x_fir_l = hpel_off_x < 1 ? 0 : 1, off_x_fir_r = hpel_off_x < 0 ? 0 : 1;
  const int off_y_fir_t = hpel_off_y < 1 ? 0 : 1, off_y_fir_b = hpel_off_y < 0 ? 0 : 1;
  const int sample_off_y = hpel_off_y < 0 ? 0 : 1;
  int x, y;
  for (y = 0; y < h + 8; y++) for (x = 0; x < w; x++) hor_pos_l[y * 64 + x] = (int16_t)tap8_px(hor_fir_l, &src[src_stride * (y - 3) + x - 2], 1);
  for (y = 0; y < h + 8; y++) col_pos_l[y] = (int16_t)tap8_px(hor_fir_l, &src[src_stride * (y - 3) - 3], 1);
  for (y = 0; y < h + 8; y++) for (x = 0; x < w; x++) hor_pos_r[y * 64 + x] = (int16_t)tap8_px(hor_fir_r, &src[src_stride * (y - 3) + x - 2], 1);
  for (y = 0; y < h + 8; y++) col_pos_r[y] = (int16_t)tap8_px(hor_fir_r, &src[src_stride * (y - 3) - 3], 1);

  const int8_t *ver_fir_l = hpel_off_y != 0 ? fir2 : fir0, *ver_fir_r = ver_fir_l;
  const int8_t *ver_fir_t = hpel_off_y != 0 ? fir1 : fir3, *ver_fir_b = hpel_off_y != 0 ? fir3 : fir1;
  for (y = 0; y < h; y++) {
    if (!off_x_fir_l) FIL(0)[y * 64] = fin(tap8_im(ver_fir_l, &col_pos_l[y + sample_off_y], 1) >> 6);
    for (x = !off_x_fir_l; x < w; x++)
      FIL(0)[y * 64 + x] = fin(tap8_im(ver_fir_l, &hor_pos_l[(y + sample_off_y) * 64 + x - !off_x_fir_l], 64) >> 6);
  }
  for (y = 0; y < h; y++) {
    if (!off_x_fir_r) FIL(1)[y * 64] = fin(tap8_im(ver_fir_r, &col_pos_r[y + sample_off_y], 1) >> 6);
    for (x = !off_x_fir_r; x < w; x++)
      FIL(1)[y * 64 + x] = fin(tap8_im(ver_fir_r, &hor_pos_r[(y + sample_off_y) * 64 + x - !off_x_fir_r], 64) >> 6);
  }
  const int sample_off_x = hpel_off_x > -1 ? 1 : 0;
  for (y = 0; y < h; y++) {
    if (!sample_off_x) FIL(2)[y * 64] = fin(tap8_im(ver_fir_t, &col_pos_hor[y + off_y_fir_t], 1) >> 6);
    for (x = !sample_off_x; x < w; x++)
      FIL(2)[y * 64 + x] = fin(tap8_im(ver_fir_t, &hor_hpel_pos[(y + off_y_fir_t) * 64 + x - !sample_off_x], 64) >> 6);
  }
  for (y = 0; y < h; y++) {
    if (!sample_off_x) FIL(3)[y * 64] = fin(tap8_im(ver_fir_b, &col_pos_hor[y + off_y_fir_b], 1) >> 6);
    for (x = !sample_off_x; x < w; x++)
      FIL(3)[y * 64 + x] = fin(tap8_im(ver_fir_b, &hor_hpel_pos[(y + off_y_fir_b) * 64 + x - !sample_off_x], 64) >> 6);
  }
}

/* ipol-generic.c:569-679 kvz_filter_qpel_blocks_diag_luma_generic */
void kvz_oracle_filter_qpel_blocks_diag_luma(const uint8_t *src, int16_t src_stride, int w, int h, uint8_t *filtered,
                                             int16_t *hor_intermediate, int8_t fme_level, int16_t *hor_first_cols,
                                             int8_t hpel_off_x, int8_t hpel_off_y)
{
  (void)src; (void)src_stride; (void)fme_level;
  const int8_t *fir1 = g_luma_filter[1], *fir3 = g_luma_filter[3];
  const int16_t *hor_pos_l = IM(3), *hor_pos_r = IM(4), *col_pos_l = COL(1), *col_pos_r = COL(3);
  const int8_t *ver_fir_t = hpel_off_y != 0 ? fir1 : fir3, *ver_fir_b = hpel_off_y != 0 ? fir3 : fir1;
  const int off_x_fir_l = hpel_off_x < 1 ? 0 : 1, off_x_fir_r = hpel_off_x < 0 ? 0 : 1;
  const int off_y_fir_t = hpel_off_y < 1 ? 0 : 1, off_y_fir_b = hpel_off_y < 0 ? 0 : 1;
  const int16_t *hp[4] = { hor_pos_l, hor_pos_r, hor_pos_l, hor_pos_r };
  const int16_t *cp[4] = { col_pos_l, col_pos_r, col_pos_l, col_pos_r };
  const int8_t *vf[4] = { ver_fir_t, ver_fir_t, ver_fir_b, ver_fir_b };
  const int offx[4] = { off_x_fir_l, off_x_fir_r, off_x_fir_l, off_x_fir_r };
  const int offy[4] = { off_y_fir_t, off_y_fir_t, off_y_fir_b, off_y_fir_b };
  for (int p = 0; p < 4; p++)
    for (int y = 0; y < h; y++) {
      if (!offx[p]) FIL(p)[y * 64] = fin(tap8_im(vf[p], &cp[p][y + offy[p]], 1) >> 6);
      for (int x = !offx[p]; x < w; x++)
        FIL(p)[y * 64 + x] = fin(tap8_im(vf[p], &hp[p][(y + offy[p]) * 64 + x - !offx[p]], 64) >> 6);
    }
}

/* ipol-generic.c:761-814 kvz_get_extended_block_generic */
int kvz_oracle_get_extended_block(const kvz_hip_epol_params *a, const uint8_t *src, uint8_t *buf)
{
  int min_y = a->blk_y - a->pad_t, max_y = a->blk_y + a->blk_h + a->pad_b + a->pad_b_simd - 1;
  int min_x = a->blk_x - a->pad_l, max_x = a->blk_x + a->blk_w + a->pad_r - 1;
  if (!((min_y < 0) || (max_y >= a->src_h) || (min_x < 0) || (max_x >= a->src_w))) return 0;
  const int ext_s = a->pad_l + a->blk_w + a->pad_r;
  int cnt_l = ORC_CLIP(0, ext_s, -min_x);
  int cnt_r = ORC_CLIP(0, ext_s, max_x - (a->src_w - 1));
  int cnt_m = ORC_CLIP(0, ext_s, ext_s - cnt_l - cnt_r);
  int y;
  for (y = -a->pad_t; y < a->blk_h + a->pad_b; y++) {
    int cy = ORC_CLIP(0, a->src_h - 1, a->blk_y + y);
    const uint8_t *sl = src + cy * a->src_s, *sr = src + cy * a->src_s + a->src_w - 1, *sm = src + cy * a->src_s + ORC_MAX(min_x, 0);
    uint8_t *dl = buf + (y + a->pad_t) * ext_s, *dm = dl + cnt_l, *dr = dm + cnt_m;
    for (int i = 0; i < cnt_l; i++) dl[i] = *sl;
    for (int i = 0; i < cnt_m; i++) dm[i] = sm[i];
    for (int i = 0; i < cnt_r; i++) dr[i] = *sr;
  }
  for (int ys = 0; ys < a->pad_b_simd; ys++) memset(buf + (y + a->pad_t + ys) * ext_s, 0, ext_s);
  /* the reference also zeroes one byte past the block for the AVX2 over-read (:805); callers size buf for it */
  buf[(a->blk_h + a->pad_b + a->pad_t + a->pad_b_simd - 1) * ext_s + a->pad_l + a->blk_w + a->pad_r] = 0;
  return 1;
}

/* ------------------------------------------------------------------------------------------------
 * SAO   (strategies/generic/sao-generic.c, sao_shared_generics.h, sao.h:71-76, sao.c:180-202)
 * ---------------------------------------------------------------------------------------------- */
static const int g_sao_ofs[4][2][2] = { { { -1, 0 }, { 1, 0 } }, { { 0, -1 }, { 0, 1 } }, { { -1, -1 }, { 1, 1 } }, { { 1, -1 }, { -1, 1 } } }; /* {x,y} */
static int sgn3(int x) { return (x > 0) - (x < 0); }
/* sao_shared_generics.h:42-50 */
static int eo_cat(int a, int b, int c) { static const int map[5] = { 1, 2, 0, 3, 4 }; return map[2 + sgn3(c - a) + sgn3(c - b)]; }

/* sao_shared_generics.h:52-91 */
int kvz_oracle_sao_edge_ddistortion(int bitdepth, const uint8_t *orig, const uint8_t *rec, int bw, int bh, int eo_class, const int offsets[5])
{
  int sum = 0;
  const int bit_offset = bitdepth != 8 ? 1 << (bitdepth - 9) : 0;
  const int ax = g_sao_ofs[eo_class][0][0], ay = g_sao_ofs[eo_class][0][1], bx = g_sao_ofs[eo_class][1][0], by = g_sao_ofs[eo_class][1][1];
  for (int y = 1; y < bh - 1; y++)
    for (int x = 1; x < bw - 1; x++) {
      int c = rec[y * bw + x], a = rec[(y + ay) * bw + x + ax], b = rec[(y + by) * bw + x + bx];
      int offset = offsets[eo_cat(a, b, c)];
      if (offset != 0) {
        int diff = (orig[y * bw + x] - c + bit_offset) >> (bitdepth - 8);
        int delta = diff - offset;
        sum += delta * delta - diff * diff;
      }
    }
  return sum;
}

/* sao-generic.c:50-81 */
void kvz_oracle_calc_sao_edge_dir(int bitdepth, const uint8_t *orig, const uint8_t *rec, int eo_class, int bw, int bh, int cat_sum_cnt[10])
{
  const int offset = bitdepth != 8 ? 1 << (bitdepth - 9) : 0;
  const int ax = g_sao_ofs[eo_class][0][0], ay = g_sao_ofs[eo_class][0][1], bx = g_sao_ofs[eo_class][1][0], by = g_sao_ofs[eo_class][1][1];
  for (int y = 1; y < bh - 1; y++)
    for (int x = 1; x < bw - 1; x++) {
      int c = rec[y * bw + x], a = rec[(y + ay) * bw + x + ax], b = rec[(y + by) * bw + x + bx];
      int cat = eo_cat(a, b, c);
      cat_sum_cnt[cat] += (orig[y * bw + x] - c + offset) >> (bitdepth - 8);
      cat_sum_cnt[5 + cat] += 1;
    }
}

/* sao-generic.c:84-124 sao_reconstruct_color_generic (+ sao.c:180-202 kvz_calc_sao_offset_array) */
void kvz_oracle_sao_reconstruct_color(const kvz_hip_sao_params *sao, const uint8_t *rec, uint8_t *new_rec, int stride, int new_stride, int bw, int bh, int color)
{
  const int offset_v = color == 2 ? 5 : 0;
  if (sao->type == 1) {
    int lut[256];
    const int values = 1 << sao->bitdepth, shift = sao->bitdepth - 5, band_pos = color == 2 ? 1 : 0, cur_bp = sao->band_position[band_pos];
    for (int val = 0; val < values; val++) {
      int d = (val >> shift) - cur_bp;
      lut[val] = (d >= 0 && d <= 3) ? ORC_CLIP(0, values - 1, val + sao->offsets[d + 1 + 5 * band_pos]) : val;
    }
    for (int y = 0; y < bh; y++) for (int x = 0; x < bw; x++) new_rec[y * new_stride + x] = (uint8_t)lut[rec[y * stride + x]];
  } else {
    const int ec = sao->eo_class;
    const int ax = g_sao_ofs[ec][0][0], ay = g_sao_ofs[ec][0][1], bx = g_sao_ofs[ec][1][0], by = g_sao_ofs[ec][1][1];
    for (int y = 0; y < bh; y++)
      for (int x = 0; x < bw; x++) {
        const uint8_t *c = &rec[y * stride + x];
        int cat = eo_cat(c[ay * stride + ax], c[by * stride + bx], c[0]);
        new_rec[y * new_stride + x] = (uint8_t)ORC_CLIP(0, 255, c[0] + sao->offsets[cat + offset_v]);
      }
  }
}

/* SAO applied to a whole picture: kvz_sao_reconstruct (sao.c:302-361) for every CTU and plane with that CTU's parameters.
 * `in` is the deblocked picture (tight planar 4:2:0), neighbours are always taken from it (SAO never reads its own output,
 * H.265 8.7.3); `out` receives the result.  luma[ctu] / chroma[ctu] in raster CTU order; chroma carries U offsets in
 * offsets[0..4] / band_position[0] and V in offsets[5..9] / band_position[1] (sao.h:55-63). */
void kvz_oracle_sao_frame(int width, int height, const uint8_t *in, uint8_t *out, const kvz_hip_sao_params *luma, const kvz_hip_sao_params *chroma)
{
  const int wc = (width + 63) / 64, hc = (height + 63) / 64;
  memcpy(out, in, (size_t)width * height * 3 / 2);
  for (int color = 0; color < 3; color++) {
    const int sh = color ? 1 : 0, fw = width >> sh, fh = height >> sh, lcu = 64 >> sh;
    const size_t plane = color == 0 ? 0 : (color == 1 ? (size_t)width * height : (size_t)width * height * 5 / 4);
    for (int cy = 0; cy < hc; cy++)
      for (int cx = 0; cx < wc; cx++) {
        const kvz_hip_sao_params *sao = color ? &chroma[cy * wc + cx] : &luma[cy * wc + cx];
        if (sao->type == 0) continue;
        int x = cx * lcu, y = cy * lcu, w = fw - x < lcu ? fw - x : lcu, h = fh - y < lcu ? fh - y : lcu;
        if (sao->type == 2) {  /* sao.c:324-349: rows / columns whose neighbour would lie outside the picture are left alone */
          const int ax = g_sao_ofs[sao->eo_class][0][0], ay = g_sao_ofs[sao->eo_class][0][1], bx = g_sao_ofs[sao->eo_class][1][0], by = g_sao_ofs[sao->eo_class][1][1];
          if (x + w + ax > fw || x + w + bx > fw) w -= 1;
          if (x + ax < 0 || x + bx < 0) { x += 1; w -= 1; }
          if (y + h + ay > fh || y + h + by > fh) h -= 1;
          if (y + ay < 0 || y + by < 0) { y += 1; h -= 1; }
        }
        if (w > 0 && h > 0) kvz_oracle_sao_reconstruct_color(sao, in + plane + (size_t)y * fw + x, out + plane + (size_t)y * fw + x, fw, fw, w, h, color);
      }
  }
}

/* sao_shared_generics.h:93-130 */
int kvz_oracle_sao_band_ddistortion(int bitdepth, const uint8_t *orig, const uint8_t *rec, int bw, int bh, int band_pos, const int sao_bands[4])
{
  const int shift = bitdepth - 5;
  int sum = 0;
  for (int i = 0; i < bw * bh; i++) {
    int band = (rec[i] >> shift) - band_pos, offset = 0;
    if (band >= 0 && band <= 3) offset = sao_bands[band];
    if (offset != 0) {
      int diff = orig[i] - rec[i], delta = diff - offset;
      sum += delta * delta - diff * diff;
    }
  }
  return sum;
}

/* ------------------------------------------------------------------------------------------------
 * Frame-edge SAD glue (image.c:279-397 image_interpolated_sad + :407-440 kvz_image_calc_sad).
 * The reference splits a block whose motion vector leaves the reference frame into corner / vertical /
 * horizontal / regular parts (cor_sad, kvz_ver_sad, kvz_hor_sad, kvz_reg_sad).  Every part compares the
 * current block against the reference sampled with edge replication, so the whole construction is
 * "SAD against the clamped reference coordinate" -- restated directly in that form.
 * ---------------------------------------------------------------------------------------------- */
unsigned kvz_oracle_image_calc_sad(const uint8_t *pic, int pic_stride, const uint8_t *ref, int ref_w, int ref_h,
                                   int ref_stride, int pic_x, int pic_y, int ref_x, int ref_y, int bw, int bh)
{
  unsigned sad = 0;
  for (int y = 0; y < bh; y++)
    for (int x = 0; x < bw; x++) {
      int rx = ORC_CLIP(0, ref_w - 1, ref_x + x), ry = ORC_CLIP(0, ref_h - 1, ref_y + y);
      sad += abs(pic[(pic_y + y) * pic_stride + pic_x + x] - ref[ry * ref_stride + rx]);
    }
  return sad;
}
