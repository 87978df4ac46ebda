// kvz_tables.hpp -- host-side construction of the constant tables the ops read (kvz::Tables) and of the
// small per-call descriptors (SATD tile lists, FME plane descriptors, quantiser scalars).
//
// The transform matrices are generated from the 32 distinct HEVC magnitudes (first column of the transposed
// 32-point matrix, /root/reference/src/strategies/generic/dct-generic.c:170); the scan tables from the
// up-right-diagonal / raster / column-major patterns applied to 4x4 coefficient groups (tables.c:9-67).
// tests/test_hostsim.py checks the generated tables against the reference's through the oracle.
#pragma once
#include <string.h>

#include "kvz_ops.hpp"

namespace kvz {

inline int dct32_entry(int k, int n)
{
  static const int mag[32] = { 64, 90, 90, 90, 89, 88, 87, 85, 83, 82, 80, 78, 75, 73, 70, 67,
                               64, 61, 57, 54, 50, 46, 43, 38, 36, 31, 25, 22, 18, 13, 9,  4 };
  if (k == 0) return 64;
  const int m = (k * (2 * n + 1)) % 128;  // angle in units of pi/64
  if (m < 32) return mag[m];
  if (m < 64) return -mag[64 - m];
  if (m < 96) return -mag[m - 64];
  return mag[128 - m];
}

inline void scan_pattern(int type, int size, int *xs, int *ys)
{
  int n = 0;
  if (type == 0) {
    for (int d = 0; d < 2 * size - 1; d++)
      for (int y = d < size - 1 ? d : size - 1; y >= 0 && d - y < size; y--) { xs[n] = d - y; ys[n] = y; n++; }
  } else if (type == 1) {
    for (int y = 0; y < size; y++) for (int x = 0; x < size; x++) { xs[n] = x; ys[n] = y; n++; }
  } else {
    for (int x = 0; x < size; x++) for (int y = 0; y < size; y++) { xs[n] = x; ys[n] = y; n++; }
  }
}

// IEEE binary16 bit pattern of a small integer (|v| < 2048: exactly representable)
inline u16 half_bits_of_int(int v)
{
  if (v == 0) return 0;
  const u16 sign = v < 0 ? 0x8000 : 0;
  unsigned a = (unsigned)(v < 0 ? -v : v);
  int e = 0;
  while ((a >> (e + 1)) != 0) e++;  // a in [2^e, 2^(e+1))
  return (u16)(sign | ((e + 15) << 10) | ((a << (10 - e)) & 0x3ff));
}

// Deblocking thresholds, H.265 Table 8-12 (filter.c:46-65 kvz_g_tc_table_8x8 / kvz_g_beta_table_8x8) and the chroma QP
// mapping of Table 8-10 (transform.c:56-62 kvz_g_chroma_scale)
inline int deblock_tc(int q)  // q in [0, 53]
{
  static const unsigned char first_q_of_next[] = { 18, 27, 31, 35, 38, 40, 42, 43, 44, 45, 46 };  // tc' steps 0 -> 1 -> ... -> 10 -> 11
  static const unsigned char from_46[] = { 11, 13, 14, 16, 18, 20, 22, 24 };
  if (q >= 46) return from_46[q - 46];
  int v = 0;
  while (q >= first_q_of_next[v]) v++;
  return v;
}
inline int deblock_beta(int q) { return q < 16 ? 0 : (q <= 28 ? q - 10 : 2 * q - 38); }  // q in [0, 51]
inline int chroma_qp_of(int qp)
{
  static const unsigned char t[58] = { 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 29, 30, 31, 32,
                                       33, 33, 34, 34, 35, 35, 36, 36, 37, 37, 38, 39, 40, 41, 42, 43, 44, 45, 46, 47, 48, 49, 50, 51 };
  return t[qp < 0 ? 0 : (qp > 57 ? 57 : qp)];
}

// Entropy bits of the HEVC CABAC state machine, fixed point 1<<15 (HM 12.0 sm_entropyBits, as tabulated in
// rdo.c:69-80 kvz_entropy_bits); CTX_ENTROPY_FBITS = value / 32768 (rdo.c:83, cabac.h:131).  Standard constant data.
static const uint32_t kEntropyBits[128] = {
  0x08000, 0x08000, 0x076da, 0x089a0, 0x06e92, 0x09340, 0x0670a, 0x09cdf, 0x06029, 0x0a67f, 0x059dd, 0x0b01f, 0x05413, 0x0b9bf, 0x04ebf, 0x0c35f,
  0x049d3, 0x0ccff, 0x04546, 0x0d69e, 0x0410d, 0x0e03e, 0x03d22, 0x0e9de, 0x0397d, 0x0f37e, 0x03619, 0x0fd1e, 0x032ee, 0x106be, 0x02ffa, 0x1105d,
  0x02d37, 0x119fd, 0x02aa2, 0x1239d, 0x02836, 0x12d3d, 0x025f2, 0x136dd, 0x023d1, 0x1407c, 0x021d2, 0x14a1c, 0x01ff2, 0x153bc, 0x01e2f, 0x15d5c,
  0x01c87, 0x166fc, 0x01af7, 0x1709b, 0x0197f, 0x17a3b, 0x0181d, 0x183db, 0x016d0, 0x18d7b, 0x01595, 0x1971b, 0x0146c, 0x1a0bb, 0x01354, 0x1aa5a,
  0x0124c, 0x1b3fa, 0x01153, 0x1bd9a, 0x01067, 0x1c73a, 0x00f89, 0x1d0da, 0x00eb7, 0x1da79, 0x00df0, 0x1e419, 0x00d34, 0x1edb9, 0x00c82, 0x1f759,
  0x00bda, 0x200f9, 0x00b3c, 0x20a99, 0x00aa5, 0x21438, 0x00a17, 0x21dd8, 0x00990, 0x22778, 0x00911, 0x23118, 0x00898, 0x23ab8, 0x00826, 0x24458,
  0x007ba, 0x24df7, 0x00753, 0x25797, 0x006f2, 0x26137, 0x00696, 0x26ad7, 0x0063f, 0x27477, 0x005ed, 0x27e17, 0x0059f, 0x287b6, 0x00554, 0x29156,
  0x0050e, 0x29af6, 0x004cc, 0x2a497, 0x0048d, 0x2ae35, 0x00451, 0x2b7d6, 0x00418, 0x2c176, 0x003e2, 0x2cb15, 0x003af, 0x2d4b5, 0x0037f, 0x2de55
};

inline void build_tables(Tables *t)
{
  memset(t, 0, sizeof(*t));
  for (int l = 0; l < 4; l++) {
    const int n = 4 << l, step = 32 / n;
    for (int k = 0; k < n; k++) for (int j = 0; j < n; j++) t->dct[l][k * n + j] = (i16)dct32_entry(k * step, j);
  }
  for (int l = 2; l < 4; l++) {
    const int n = 4 << l;
    for (int which = 0; which < 2; which++)
      for (int r = 0; r < n; r++) {
        int sum = 0;
        for (int slot = 0; slot < n; slot++) {
          const int h = slot >> 4, j = slot & 15, k = n == 32 ? 8 * (j >> 2) + 4 * h + (j & 3) : slot;  // kvz_mfma.hpp DevMma<32>::row
          const int v = which == 0 ? t->dct[l][r * n + k] : t->dct[l][k * n + r];
          t->dct_i8[l - 2][which][r * n + slot] = (int8_t)v;
          sum += v;
        }
        t->dct_sum[l - 2][which][r] = sum;
      }
  }
  {  // a unit to the above-right / below-left is usable iff it comes earlier in z-order (cu.h:385-421) than the unit itself
    auto z = [](int x4, int y4) { unsigned r = 0; for (int b = 0; b < 4; b++) r |= (((x4 >> b) & 1u) << (2 * b)) | (((y4 >> b) & 1u) << (2 * b + 1)); return r; };
    for (int r = 0; r < 16; r++)
      for (int c = 0; c < 16; c++) {
        int n = 0;
        if (r == 0) t->avail_top[r][c] = 64;
        else { for (int cc = c; cc < 16 && z(cc, r - 1) < z(c, r); cc++) n++; t->avail_top[r][c] = (u8)(4 * n); }
        n = 0;
        if (c == 0) t->avail_left[r][c] = (u8)(64 - 4 * r);
        else { for (int rr = r; rr < 16 && z(c - 1, rr) < z(c, r); rr++) n++; t->avail_left[r][c] = (u8)(4 * n); }
      }
  }
  {  // H.265 table 9-41 transIdxLps; transIdxMps = min(state + 1, 62)
    static const u8 lps[64] = { 0, 0, 1, 2, 2, 4, 4, 5, 6, 7, 8, 9, 9, 11, 11, 12, 13, 13, 15, 15, 16, 16, 18, 18, 19, 19, 21, 21, 22, 22, 23, 24,
                                24, 25, 26, 26, 27, 27, 28, 29, 29, 30, 30, 30, 31, 32, 32, 33, 33, 33, 34, 34, 35, 35, 35, 36, 36, 36, 37, 37, 37, 38, 38, 63 };
    for (int u = 0; u < 128; u++) {
      const int st = u >> 1, mps = u & 1;
      t->ctx_next[0][u] = (u8)(st >= 62 ? u : ((st + 1) << 1) | mps);
      t->ctx_next[1][u] = (u8)(st == 0 ? (mps ^ 1) : (lps[st] << 1) | mps);
    }
  }
  {
    int n = 0;
    for (int d = 0; d < 15; d++) for (int x = 0; x <= d; x++) if (x < 8 && d - x < 8) t->diag8[n++] = (u8)((d - x) * 8 + x);
  }
  for (int l2 = 3; l2 <= 4; l2++) {  // Tables::mref_tab (kvz_ctu.hpp build_mref; the layout constants kMref* below are asserted against the kernel's there)
    static const int inv_tab[9] = { 0, 4096, 1638, 910, 630, 482, 390, 315, 256 };
    const int w = 1 << l2, nq = 2 * w + 2, thres = l2 == 3 ? 7 : 1;
    for (int i = 0; i < 15 * nq; i++) {
      const int mode = 11 + i / nq, q = i % nq - w, vertical = mode >= 18, md = vertical ? mode - 26 : 10 - mode, ad = md < 0 ? -md : md;
      const int d26 = mode > 26 ? mode - 26 : 26 - mode, d10 = mode > 10 ? mode - 10 : 10 - mode, filt = (d26 < d10 ? d26 : d10) > thres;
      int idx = q >= 0 ? q : (128 + (-q) * inv_tab[ad]) >> 8;
      if (idx > 2 * w) idx = 2 * w;
      const int main_side = vertical ? 0 : 1, side = q >= 0 ? main_side : 1 - main_side;  // 0 top, 1 left
      const unsigned src = (filt ? kMrefFiltered : 0u) + kMrefRefRow * side + idx, dst = kMrefStride * (i / nq) + kMrefOrg + i % nq - w;
      t->mref_tab[l2 - 3][i] = src | dst << 16;
    }
    for (int i = 15 * nq; i < 512; i++) t->mref_tab[l2 - 3][i] = t->mref_tab[l2 - 3][15 * nq - 1];  // the lanes past the end repeat the last entry: no bounds test in the kernel
  }
  static const i16 dst4[16] = { 29, 55, 74, 84, 74, 74, 0, -74, 84, -29, -74, 55, 55, -84, 74, -29 };
  memcpy(t->dst4, dst4, sizeof dst4);
  for (int kind = 0; kind < 3; kind++) {
    const int n = kind == 1 ? 8 : 4;
    const i16 *m = kind == 2 ? t->dst4 : t->dct[kind == 1 ? 1 : 0];
    for (int g = 0; g < 16 / n; g++)
      for (int k = 0; k < n; k++)
        for (int j = 0; j < n; j++) {
          t->bd_i8[kind][0][(g * n + k) * 16 + g * n + j] = (int8_t)m[k * n + j];
          t->bd_i8[kind][1][(g * n + j) * 16 + g * n + k] = (int8_t)m[k * n + j];
          t->bd_sum[kind][0][g * n + k] += m[k * n + j];
          t->bd_sum[kind][1][g * n + j] += m[k * n + j];
        }
  }
  for (int kind = 0; kind < 3; kind++) {
    const int n = kind == 1 ? 8 : 4;
    const i16 *m = kind == 2 ? t->dst4 : t->dct[kind == 1 ? 1 : 0];
    for (int k = 0; k < n; k++)
      for (int i = 0; i < n / 2; i++) {
        t->small_pairs[kind][0][k][i] = (u32)(u16)m[k * n + 2 * i] | ((u32)(u16)m[k * n + 2 * i + 1] << 16);
        t->small_pairs[kind][1][k][i] = (u32)(u16)m[(2 * i) * n + k] | ((u32)(u16)m[(2 * i + 1) * n + k] << 16);
      }
  }
  for (int k = 0; k < 16; k++)
    for (int i = 0; i < 8; i++) {
      const i16 *m = t->dct[2];
      t->pairs16[0][k][i] = (u32)(u16)m[k * 16 + 2 * i] | ((u32)(u16)m[k * 16 + 2 * i + 1] << 16);
      t->pairs16[1][k][i] = (u32)(u16)m[(2 * i) * 16 + k] | ((u32)(u16)m[(2 * i + 1) * 16 + k] << 16);
    }
  for (int type = 0; type < 3; type++)
    for (int l2 = 2; l2 <= 5; l2++) {
      const int size = 1 << l2, cgs = size / 4;
      int gx[64], gy[64], px[16], py[16], n = 0;
      scan_pattern(type, cgs, gx, gy);
      scan_pattern(type, 4, px, py);
      for (int g = 0; g < cgs * cgs; g++)
        for (int i = 0; i < 16; i++) t->scan[type][l2 - 2][n++] = (u32)((gy[g] * 4 + py[i]) * size + gx[g] * 4 + px[i]);
    }
  static const int8_t lf[4][8] = { { 0, 0, 0, 64, 0, 0, 0, 0 }, { -1, 4, -10, 58, 17, -5, 1, 0 }, { -1, 4, -11, 40, 40, -11, 4, -1 }, { 0, 1, -5, 17, 58, -10, 4, -1 } };
  static const int8_t cf[8][4] = { { 0, 64, 0, 0 }, { -2, 58, 10, -2 }, { -4, 54, 16, -2 }, { -6, 46, 28, -4 }, { -4, 36, 36, -4 }, { -4, 28, 46, -6 }, { -2, 16, 54, -4 }, { -2, 10, 58, -2 } };
  for (int i = 0; i < 128; i++) t->entropy_bits[i] = kEntropyBits[i];
  memcpy(t->luma_filter, lf, sizeof lf);
  memcpy(t->chroma_filter, cf, sizeof cf);
}

inline int ilog2(int w) { int l = 0; while ((1 << l) < w) l++; return l; }

// transform.c:141-155 kvz_get_scaled_qp
inline int scaled_qp(int type, int qp, int qp_offset)
{
  static const uint8_t chroma_scale[58] = { 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 29, 30, 31, 32,
                                            33, 33, 34, 34, 35, 35, 36, 36, 37, 37, 38, 39, 40, 41, 42, 43, 44, 45, 46, 47, 48, 49, 50, 51 };
  if (type == 0) return qp + qp_offset;
  int q = qp < -qp_offset ? -qp_offset : (qp > 57 ? 57 : qp);
  return q < 0 ? q + qp_offset : chroma_scale[q] + qp_offset;
}

// quant-generic.c:57-66 (forward) and :303-339 (inverse) scalars for a width x width block of plane type
// (`type` as the reference passes it: 0 luma, 2/3 chroma).
inline QuantScalars quant_scalars(int qp, int bitdepth, int slice_is_intra, int scaling_list, int width, int type)
{
  static const int quant_scales[6] = { 26214, 23302, 20560, 18396, 16384, 14564 };  // scalinglist.c:78
  static const int inv_quant_scales[6] = { 40, 45, 51, 57, 64, 72 };                // scalinglist.c:79
  QuantScalars q;
  const int log2_tr = ilog2(width);
  const int qps = scaled_qp(type, qp, (bitdepth - 8) * 6);
  const int transform_shift = 15 - bitdepth - log2_tr;
  q.q_bits = 14 + qps / 6 + transform_shift;
  q.add = (slice_is_intra ? 171 : 85) << (q.q_bits - 9);
  q.flat_q = quant_scales[qps % 6];
  q.dq_shift = 20 - 14 - transform_shift + (scaling_list ? 4 : 0);
  q.dq_scale = inv_quant_scales[qps % 6] << (qps / 6);
  q.dq_list = scaling_list;
  q.dq_qp_per = qps / 6;
  return q;
}

// Tile lists.  kind 0: n x n as strategies-picture.h:53-69 (8x8 tiles, or one 4x4); kind 1: satd_any_size
// (:75-113); kind 2: satd_any_size_quad with the reference's offsets (picture-generic.c:404-471): the first-row
// pass restarts at column 0 and the 8x8 pass starts at row 0 even when the first 4 rows were already covered.
inline int satd_tiles(int kind, int w, int h, SatdTile *out)
{
  int n = 0;
  if (kind == 0) {
    if (w == 4) { out[n++] = SatdTile{ 0, 0, 4, 0 }; return n; }
    for (int y = 0; y < h; y += 8) for (int x = 0; x < w; x += 8) out[n++] = SatdTile{ (i16)x, (i16)y, 8, 0 };
    return n;
  }
  if (kind == 1) {
    int x0 = 0, y0 = 0;
    if (w % 8 != 0) { for (int y = 0; y < h; y += 4) out[n++] = SatdTile{ 0, (i16)y, 4, 0 }; x0 = 4; }
    if (h % 8 != 0) { for (int x = x0; x < w; x += 4) out[n++] = SatdTile{ (i16)x, 0, 4, 0 }; y0 = 4; }
    for (int y = y0; y < h; y += 8) for (int x = x0; x < w; x += 8) out[n++] = SatdTile{ (i16)x, (i16)y, 8, 0 };
    return n;
  }
  int width = w, height = h;
  const int wm8 = w % 8;
  if (wm8 != 0) { for (int y = 0; y < height; y += 4) out[n++] = SatdTile{ 0, (i16)y, 4, 0 }; width -= 4; }
  if (height % 8 != 0) { for (int x = 0; x < width; x += 4) out[n++] = SatdTile{ (i16)x, 0, 4, 0 }; height -= 4; }
  for (int y = height % 8; y < height; y += 8) for (int x = wm8; x < width; x += 8) out[n++] = SatdTile{ (i16)x, (i16)y, 8, 0 };
  return n;
}

// Plane descriptors of the four FME block filters (ipol-generic.c:213-679), derived in kvz_ops.hpp's notation:
//   which 0: hpel hor_ver  left  = H(f2)(y+1, x)      right = H(f2)(y+1, x+1)   top = V(f2)(y, x+1)   bottom = V(f2)(y+1, x+1)
//   which 1: hpel diag     tl = HV(f2,f2)(y, x)   tr = HV(f2,f2)(y, x+1)   bl = HV(f2,f2)(y+1, x)   br = HV(f2,f2)(y+1, x+1)
//   which 2: qpel hor_ver  l = HV(hl, vl)(y+soy, x+ofl)   r = HV(hr, vl)(y+soy, x+ofr)   t = HV(hh, vt)(y+oft, x+sox)   b = HV(hh, vb)(y+ofb, x+sox)
//   which 3: qpel diag     tl/tr/bl/br = HV(hl|hr, vt|vb)(y+oft|ofb, x+ofl|ofr)
KVZ_HD void fme_planes(int which, int ox, int oy, FmePlane pl[4])
{
  const int hl = ox != 0 ? 1 : 3, hr = ox != 0 ? 3 : 1, hh = ox != 0 ? 2 : 0;
  const int vl = oy != 0 ? 2 : 0, vt = oy != 0 ? 1 : 3, vb = oy != 0 ? 3 : 1;
  const int ofl = ox < 1 ? 0 : 1, ofr = ox < 0 ? 0 : 1, oft = oy < 1 ? 0 : 1, ofb = oy < 0 ? 0 : 1;
  const int soy = oy < 0 ? 0 : 1, sox = ox > -1 ? 1 : 0;
  if (which == 0) {
    pl[0] = FmePlane{ 0, 2, 0, 1, 0 }; pl[1] = FmePlane{ 0, 2, 0, 1, 1 };
    pl[2] = FmePlane{ 1, 0, 2, 0, 1 }; pl[3] = FmePlane{ 1, 0, 2, 1, 1 };
  } else if (which == 1) {
    pl[0] = FmePlane{ 2, 2, 2, 0, 0 }; pl[1] = FmePlane{ 2, 2, 2, 0, 1 };
    pl[2] = FmePlane{ 2, 2, 2, 1, 0 }; pl[3] = FmePlane{ 2, 2, 2, 1, 1 };
  } else if (which == 2) {
    pl[0] = FmePlane{ 2, hl, vl, soy, ofl }; pl[1] = FmePlane{ 2, hr, vl, soy, ofr };
    pl[2] = FmePlane{ 2, hh, vt, oft, sox }; pl[3] = FmePlane{ 2, hh, vb, ofb, sox };
  } else {
    pl[0] = FmePlane{ 2, hl, vt, oft, ofl }; pl[1] = FmePlane{ 2, hr, vt, oft, ofr };
    pl[2] = FmePlane{ 2, hl, vb, ofb, ofl }; pl[3] = FmePlane{ 2, hr, vb, ofb, ofr };
  }
}

}  // namespace kvz
