// kvz_api_impl.hpp -- the per-call ("drop-in") entry points of the flat strategy API, written once against a
// Backend policy:
//   HipBackend  (kvz_hip.hip)          pinned staging arena -> one H2D copy -> kernel launches on the calling
//                                      thread's stream -> one D2H copy -> stream sync.  This is the product.
//   HostBackend (tests/hostsim)        same call sequences with every op run by a host loop; test infrastructure.
//
// Synchronous semantics as the reference requires: caller owns all buffers, outputs are complete on
// return, nothing is retained, callable concurrently from many threads (one arena + stream per thread).
//
// A call stages its inputs contiguously (strided inputs are gathered row by row so only w*h bytes travel),
// reserves zero-initialised accumulators / outputs / scratch behind them, and runs a short sequence of ops.
#pragma once
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>

#include "../../include/kvz_hip_types.h"
#include "kvz_ops.hpp"
#include "kvz_entropy.hpp"
#include "kvz_rdoq.hpp"
#include "kvz_tables.hpp"

namespace kvz {

// Backend concept:
//   void begin();                                   start a call (resets the arena)
//   T *in(const T *host, size_t n);                 stage n elements, returns the device address
//   T *in_rows(const T *host, int w, int h, long stride);   gather h rows of w elements (device stride = w)
//   T *zeroed(size_t n);                            zero-initialised, uploaded AND downloaded (accumulators)
//   T *out(size_t n);                               downloaded only
//   T *scratch(size_t n);                           device only
//   void upload();                                  after the last in()/zeroed()
//   void run(const Op &op, int n_items);
//   void download();                                copies zeroed()+out() regions back and waits
//   const T *host(const T *dev);                    staging address of a device address (valid after download)
//   const Tables *tables();                         device address of the constant tables

template <class B> struct Api {
  // ---------------------------------------------------------------- picture
  static unsigned reg_sad(B &be, const u8 *d1, const u8 *d2, int w, int h, unsigned s1, unsigned s2)
  {
    if (w <= 0 || h <= 0) return 0;
    be.begin();
    const u8 *a = be.in_rows(d1, w, h, s1), *b = be.in_rows(d2, w, h, s2);
    u32 *o = be.template zeroed<u32>(1);
    be.upload();
    be.run(SadRectOp{ a, w, 0, b, w, 0, w, h, 0, 0, 0, 0, 0, o }, h);
    be.download();
    return *be.host(o);
  }

  static unsigned sad_nxn(B &be, int n, const u8 *b1, const u8 *b2) { return reg_sad(be, b1, b2, n, n, n, n); }

  static u32 satd_tiles_call(B &be, int kind, int w, int h, const u8 *b1, int s1, const u8 *b2, int s2)
  {
    SatdTile tiles[160];
    const int nt = satd_tiles(kind, w, h, tiles);
    be.begin();
    const u8 *a = be.in_rows(b1, w, h, s1), *b = be.in_rows(b2, w, h, s2);
    const SatdTile *t = be.in(tiles, nt);
    u32 *o = be.template zeroed<u32>(1);
    be.upload();
    be.run(SatdTilesOp{ a, w, 0, b, w, 0, t, nt, o }, nt);
    be.download();
    return *be.host(o);
  }
  static unsigned satd_nxn(B &be, int n, const u8 *b1, const u8 *b2) { return satd_tiles_call(be, 0, n, n, b1, n, b2, n); }
  static unsigned satd_any_size(B &be, int w, int h, const u8 *b1, int s1, const u8 *b2, int s2) { return satd_tiles_call(be, 1, w, h, b1, s1, b2, s2); }

  // preds = two slabs of 32*32 (strategies-picture.h:48 pred_buffer)
  static void nxn_dual(B &be, int satd, int n, const u8 *preds, const u8 *orig, unsigned *costs)
  {
    SatdTile tiles[64];
    const int nt = satd ? satd_tiles(0, n, n, tiles) : 0;
    be.begin();
    u8 *p = be.template in_raw<u8>(2 * n * n);
    memcpy(be.host_rw(p), preds, n * n);
    memcpy(be.host_rw(p) + n * n, preds + 1024, n * n);
    const u8 *o = be.in(orig, n * n);
    const SatdTile *t = satd ? be.in(tiles, nt) : nullptr;
    u32 *c = be.template zeroed<u32>(2);
    be.upload();
    if (satd) be.run(SatdTilesOp{ p, n, (long)n * n, o, n, 0, t, nt, c }, 2 * nt);
    else be.run(SadRectOp{ p, n, (long)n * n, o, n, 0, n, n, 0, 0, 0, 0, 0, c }, 2 * n);
    be.download();
    costs[0] = be.host(c)[0];
    costs[1] = be.host(c)[1];
  }

  static void satd_any_size_quad(B &be, int w, int h, const u8 *const *preds, int stride, const u8 *orig, int orig_stride, unsigned *costs)
  {
    SatdTile tiles[200];
    const int nt = satd_tiles(2, w, h, tiles);
    be.begin();
    u8 *p = be.template in_raw<u8>(4 * w * h);
    for (int k = 0; k < 4; k++) for (int y = 0; y < h; y++) memcpy(be.host_rw(p) + (k * h + y) * w, preds[k] + (long)y * stride, w);
    const u8 *o = be.in_rows(orig, w, h, orig_stride);
    const SatdTile *t = be.in(tiles, nt);
    u32 *c = be.template zeroed<u32>(4);
    be.upload();
    be.run(SatdTilesOp{ o, w, 0, p, w, (long)w * h, t, nt, c }, 4 * nt);
    be.download();
    for (int k = 0; k < 4; k++) costs[k] = be.host(c)[k];
  }

  static unsigned pixels_calc_ssd(B &be, const u8 *ref, const u8 *rec, int rs, int cs, int width)
  {
    be.begin();
    const u8 *a = be.in_rows(ref, width, width, rs), *b = be.in_rows(rec, width, width, cs);
    u32 *o = be.template zeroed<u32>(1);
    be.upload();
    be.run(SsdOp{ a, width, 0, b, width, 0, width, o }, width);
    be.download();
    return *be.host(o);
  }

  static u32 ver_sad(B &be, const u8 *pic, const u8 *ref, int bw, int bh, u32 pic_stride)
  {
    be.begin();
    const u8 *a = be.in_rows(pic, bw, bh, pic_stride), *b = be.in(ref, bw);
    u32 *o = be.template zeroed<u32>(1);
    be.upload();
    be.run(SadRectOp{ a, bw, 0, b, 0, 0, bw, bh, 0, 0, 0, 0, 0, o }, bh);
    be.download();
    return *be.host(o);
  }

  static u32 hor_sad(B &be, const u8 *pic, const u8 *ref, int w, int h, u32 ps, u32 rs, u32 left, u32 right)
  {
    be.begin();
    const u8 *a = be.in_rows(pic, w, h, ps), *b = be.in_rows(ref, w, h, rs);
    u32 *o = be.template zeroed<u32>(1);
    be.upload();
    be.run(HorSadOp{ a, w, b, w, w, h, (int)left, (int)right, o }, h);
    be.download();
    return *be.host(o);
  }

  // image.c:407 kvz_image_calc_sad: SAD against the edge-replicated reference frame (only the touched window is staged)
  static unsigned image_calc_sad(B &be, const u8 *pic, int pic_stride, const u8 *ref, int ref_w, int ref_h, int ref_stride,
                                 int pic_x, int pic_y, int ref_x, int ref_y, int bw, int bh)
  {
    const int x0 = iclip(0, ref_w - 1, ref_x), x1 = iclip(0, ref_w - 1, ref_x + bw - 1);
    const int y0 = iclip(0, ref_h - 1, ref_y), y1 = iclip(0, ref_h - 1, ref_y + bh - 1);
    const int ww = x1 - x0 + 1, wh = y1 - y0 + 1;
    be.begin();
    const u8 *a = be.in_rows(pic + (long)pic_y * pic_stride + pic_x, bw, bh, pic_stride);
    const u8 *b = be.in_rows(ref + (long)y0 * ref_stride + x0, ww, wh, ref_stride);
    u32 *o = be.template zeroed<u32>(1);
    be.upload();
    // the staged window is a ww x wh "frame" whose origin is (x0, y0)
    be.run(SadRectOp{ a, bw, 0, b, ww, 0, bw, bh, 1, ww, wh, ref_x - x0, ref_y - y0, o }, bh);
    be.download();
    return *be.host(o);
  }

  static double pixel_var(B &be, const u8 *buf, u32 len)
  {
    be.begin();
    const u8 *a = be.in(buf, len);
    double *o = be.template out<double>(1);
    be.upload();
    be.run(PixelVarOp{ a, len, o }, 1);
    be.download();
    return *be.host(o);
  }

  static void bipred_average_plane(B &be, u8 *dst, unsigned dst_stride, const u8 *px0, const i16 *im0, const u8 *px1, const i16 *im1, unsigned w, unsigned h)
  {
    const int n = (int)(w * h);
    be.begin();
    const u8 *p0 = px0 ? be.in(px0, n) : nullptr, *p1 = px1 ? be.in(px1, n) : nullptr;
    const i16 *i0 = im0 ? be.in(im0, n) : nullptr, *i1 = im1 ? be.in(im1, n) : nullptr;
    u8 *o = be.template out<u8>(n);
    be.upload();
    be.run(BipredOp{ o, (int)w, p0, i0, p1, i1, (int)w }, n);
    be.download();
    for (unsigned y = 0; y < h; y++) memcpy(dst + (long)y * dst_stride, be.host(o) + y * w, w);
  }

  // ---------------------------------------------------------------- transforms
  static const i16 *matrix(B &be, int kind_idx)
  {
    const Tables *t = be.tables();
    return kind_idx == 4 ? t->dst4 : t->dct[kind_idx];
  }
  // runs both passes on device buffers: src -> tmp -> dst, `blocks` blocks of n*n
  static void transform_dev(B &be, int kind, int bitdepth, const i16 *src, i16 *tmp, i16 *dst, int blocks)
  {
    static const int sizes[5] = { 4, 8, 16, 32, 4 };
    const int inverse = kind >= KVZ_HIP_IDCT_4, idx = inverse ? kind - KVZ_HIP_IDCT_4 : kind;
    const int n = sizes[idx], l2 = ilog2(n);
    const i16 *C = matrix(be, idx);
    const int s1 = inverse ? 7 : l2 - 1 + (bitdepth - 8), s2 = inverse ? 12 - (bitdepth - 8) : l2 + 6;
    be.run(TransformPassOp{ C, n, l2, s1, inverse, src, tmp }, blocks * n * n);
    be.run(TransformPassOp{ C, n, l2, s2, inverse, tmp, dst }, blocks * n * n);
  }
  static void transform(B &be, int kind, int8_t bitdepth, const i16 *in, i16 *out)
  {
    static const int sizes[5] = { 4, 8, 16, 32, 4 };
    const int n = sizes[kind >= KVZ_HIP_IDCT_4 ? kind - KVZ_HIP_IDCT_4 : kind];
    be.begin();
    const i16 *s = be.in(in, n * n);
    i16 *o = be.template out<i16>(n * n);
    i16 *tmp = be.template scratch<i16>(n * n);
    be.upload();
    transform_dev(be, kind, bitdepth, s, tmp, o, 1);
    be.download();
    memcpy(out, be.host(o), n * n * sizeof(i16));
  }

  // ---------------------------------------------------------------- quant
  static void quant_dev(B &be, const kvz_hip_quant_params *p, const QuantScalars &q, const i16 *coef, const i16 *qtab, i16 *out, u32 *ac, int n, int scan_idx, int log2w)
  {
    be.run(QuantOp{ q, coef, qtab, out, ac, n }, n);
    if (p->signhide) be.run(SignHideOp{ q, coef, qtab, out, ac, be.tables()->scan[scan_idx][log2w - 2], n }, 1);
  }
  static void quant(B &be, const kvz_hip_quant_params *p, const i16 *coef, i16 *q_coef, int width, int height, int type, int scan_idx)
  {
    const int n = width * height;
    const QuantScalars q = quant_scalars(p->qp, p->bitdepth, p->slice_is_intra, p->scaling_list, width, type);
    be.begin();
    const i16 *c = be.in(coef, n);
    const i16 *qt = p->quant_coeff ? be.in(p->quant_coeff, n) : nullptr;
    u32 *ac = be.template zeroed<u32>(1);
    i16 *o = be.template out<i16>(n);
    be.upload();
    quant_dev(be, p, q, c, qt, o, ac, n, scan_idx, ilog2(width));
    be.download();
    memcpy(q_coef, be.host(o), n * sizeof(i16));
  }
  static void dequant(B &be, const kvz_hip_quant_params *p, const i16 *q_coef, i16 *coef, int width, int height, int type)
  {
    const int n = width * height;
    const QuantScalars q = quant_scalars(p->qp, p->bitdepth, p->slice_is_intra, p->scaling_list, width, type);
    be.begin();
    const i16 *c = be.in(q_coef, n);
    const i16 *dt = (p->scaling_list && p->dequant_coeff) ? be.in(p->dequant_coeff, n) : nullptr;
    i16 *o = be.template out<i16>(n);
    be.upload();
    be.run(DequantOp{ q, c, dt, o, n }, n);
    be.download();
    memcpy(coef, be.host(o), n * sizeof(i16));
  }

  // quant-generic.c:198-292 kvz_quantize_residual_generic with rdoq off (kvz_rdoq is host code in the reference)
  static int quantize_residual(B &be, const kvz_hip_quant_params *p, int width, int color, int scan_order, int use_trskip, int in_stride,
                               int out_stride, const u8 *ref_in, const u8 *pred_in, u8 *rec_out, i16 *coeff_out, int early_skip)
  {
    const int n = width * width, l2 = ilog2(width);
    const int idx = width == 4 ? ((color == 0 && p->cu_is_intra) ? 4 : 0) : l2 - 2;
    const QuantScalars qf = quant_scalars(p->qp, p->bitdepth, p->slice_is_intra, p->scaling_list, width, color == 0 ? 0 : 2);
    const QuantScalars qi = quant_scalars(p->qp, p->bitdepth, p->slice_is_intra, p->scaling_list, width, color == 0 ? 0 : (color == 1 ? 2 : 3));
    be.begin();
    const u8 *ref = be.in_rows(ref_in, width, width, in_stride), *pred = be.in_rows(pred_in, width, width, in_stride);
    const i16 *qt = p->quant_coeff ? be.in(p->quant_coeff, n) : nullptr;
    const i16 *dt = (p->scaling_list && p->dequant_coeff) ? be.in(p->dequant_coeff, n) : nullptr;
    u32 *acc = be.template zeroed<u32>(2);  // [0] ac_sum, [1] non-zero count
    i16 *co = be.template out<i16>(n);
    u8 *rec = be.template out<u8>(n);
    i16 *resid = be.template scratch<i16>(n), *coeff = be.template scratch<i16>(n), *tmp = be.template scratch<i16>(n);
    be.upload();
    be.run(ResidualOp{ ref, pred, width, width, resid }, n);
    const int ts_shift = 15 - p->bitdepth - l2;
    if (use_trskip) be.run(TransformSkipOp{ ts_shift, 0, resid, coeff }, n);
    else transform_dev(be, idx, p->bitdepth, resid, tmp, coeff, 1);
    quant_dev(be, p, qf, coeff, qt, co, acc, n, scan_order, l2);
    be.run(AnyNonzeroOp{ co, acc + 1 }, n);
    // the inverse path always runs on the device; ReconOp selects pred when there are no coefficients / early skip
    be.run(DequantOp{ qi, co, dt, coeff, n }, n);
    if (use_trskip) be.run(TransformSkipOp{ ts_shift, 1, coeff, resid }, n);
    else transform_dev(be, KVZ_HIP_IDCT_4 + idx, p->bitdepth, coeff, tmp, resid, 1);
    be.run(ReconOp{ resid, pred, width, rec, width, width, acc + 1, early_skip }, n);
    be.download();
    const int has = be.host(acc)[1] != 0;
    memcpy(coeff_out, be.host(co), n * sizeof(i16));
    if ((has && !early_skip) || rec_out != pred_in)
      for (int y = 0; y < width; y++) memcpy(rec_out + (long)y * out_stride, be.host(rec) + y * width, width);
    return has;
  }

  // ... and with rdoq on (quant-generic.c:234-244: kvz_rdoq in place of kvz_quant) for an intra block, flat lists, sign hiding and transform skip off: the
  // whole function in ONE round trip -- residual, transform, kvz_rdoq on the caller's context states, dequantisation, inverse transform, reconstruction
  static int quantize_residual_rdoq(B &be, const kvz_hip_quant_params *p, double lambda, const u8 *ctx_states, int tr_depth, int width, int color, int scan_order, int in_stride,
                                    int out_stride, const u8 *ref_in, const u8 *pred_in, u8 *rec_out, i16 *coeff_out, int early_skip)
  {
    const int n = width * width, l2 = ilog2(width);
    const int idx = width == 4 ? ((color == 0 && p->cu_is_intra) ? 4 : 0) : l2 - 2;
    const QuantScalars qi = quant_scalars(p->qp, p->bitdepth, p->slice_is_intra, 0, width, color == 0 ? 0 : (color == 1 ? 2 : 3));
    be.begin();
    const u8 *ref = be.in_rows(ref_in, width, width, in_stride), *pred = be.in_rows(pred_in, width, width, in_stride);
    const u8 *cx = be.in(ctx_states, 160);
    u32 *acc = be.template zeroed<u32>(2);  // [1] non-zero count
    i16 *co = be.template out<i16>(n);
    u8 *rec = be.template out<u8>(n);
    i16 *resid = be.template scratch<i16>(n), *coeff = be.template scratch<i16>(n), *tmp = be.template scratch<i16>(n);
    be.upload();
    be.run(ResidualOp{ ref, pred, width, width, resid }, n);
    transform_dev(be, idx, p->bitdepth, resid, tmp, coeff, 1);
    be.run_wave(RdoqOp{ be.tables(), cx, lambda, p->qp, coeff, co, l2, color == 0 ? 0 : 2, scan_order, tr_depth }, 1);  // writes every level of the block
    be.run(AnyNonzeroOp{ co, acc + 1 }, n);
    be.run(DequantOp{ qi, co, nullptr, coeff, n }, n);
    transform_dev(be, KVZ_HIP_IDCT_4 + idx, p->bitdepth, coeff, tmp, resid, 1);
    be.run(ReconOp{ resid, pred, width, rec, width, width, acc + 1, early_skip }, n);
    be.download();
    const int has = be.host(acc)[1] != 0;
    memcpy(coeff_out, be.host(co), n * sizeof(i16));
    if ((has && !early_skip) || rec_out != pred_in)
      for (int y = 0; y < width; y++) memcpy(rec_out + (long)y * out_stride, be.host(rec) + y * width, width);
    return has;
  }

  static u32 coeff_abs_sum(B &be, const i16 *coeffs, size_t length)
  {
    be.begin();
    const i16 *c = be.in(coeffs, length);
    u32 *o = be.template zeroed<u32>(1);
    be.upload();
    be.run(AbsSumOp{ c, o }, (int)length);
    be.download();
    return *be.host(o);
  }
  // ---------------------------------------------------------------- encode
  // kvz_encode_coeff_nxn (strategies-encode.h:49-65) as far as it is data-parallel-friendly per call: the block's syntax as bin records (kvz_entropy.hpp); the
  // arithmetic coder state lives in the caller's cabac_data_t.  Returns the number of records the block has (more than `capacity`: the list is truncated).
  static int coeff_nxn_bins(B &be, const i16 *coeff, int width, int type, int scan_mode, u32 *records, int capacity)
  {
    int log2 = 2;
    while ((1 << log2) < width) log2++;
    be.begin();
    const i16 *c = be.in(coeff, (size_t)width * width);
    u32 *o = be.template zeroed<u32>((size_t)capacity + 1);
    be.upload();
    be.run(CoeffBinsOp{ be.tables(), c, log2, type, scan_mode, o, (u32)capacity }, 1);
    be.download();
    const int n = (int)be.host(o)[0];
    memcpy(records, be.host(o) + 1, (size_t)(n < capacity ? n : capacity) * sizeof(u32));
    return n;
  }
  // ---------------------------------------------------------------- nal
  // kvz_array_checksum is called once per whole plane (nal.c:79-84; 2 MB of luma at 1080p, 8 MB at 4K): the plane goes through the
  // staging arena in row chunks, the mask only depends on absolute (x, y) and the u32 sum wraps the same way in any order.
  static u32 plane_checksum(B &be, const u8 *data, int height, int width, int stride)
  {
    if (height <= 0 || width <= 0) return 0;
    const size_t room = be.cap - 4096;
    if ((size_t)width > room) { fprintf(stderr, "kvz_hip: plane_checksum: a row of %d bytes does not fit the staging arena\n", width); abort(); }
    const int rows = (int)(room / (size_t)width);
    u32 total = 0;
    for (int y0 = 0; y0 < height; y0 += rows) {
      const int n = height - y0 < rows ? height - y0 : rows;
      be.begin();
      const u8 *d = be.in_rows(data + (long)y0 * stride, width, n, stride);
      u32 *o = be.template zeroed<u32>(1);
      be.upload();
      be.run(PlaneChecksumOp{ d, width, width, y0, o }, n);
      be.download();
      total += *be.host(o);
    }
    return total;
  }

  // kvz_array_md5 (nal-generic.c:41-55): width * height contiguous bytes from `data` (the reference ignores the stride too); the message
  // goes through the staging arena in chunks of whole 64-byte blocks, the chaining value stays on the device's side of the arena.
  static void plane_md5(B &be, const u8 *data, int height, int width, u8 out[16])
  {
    const unsigned long long total = (unsigned long long)(width > 0 && height > 0 ? (long)width * height : 0);
    const size_t room = (be.cap - 4096) & ~(size_t)63;
    u32 state[4] = { 0x67452301u, 0xefcdab89u, 0x98badcfeu, 0x10325476u };
    unsigned long long done = 0;
    for (;;) {
      const unsigned long long left = total - done;
      const int finish = left <= room;
      const size_t n = finish ? (size_t)left : room;
      be.begin();
      const u8 *d = be.in(data + done, n ? n : 1);
      u32 *st = be.template in_raw<u32>(4);
      memcpy(be.host_rw(st), state, sizeof state);
      be.mark_download_from(st);
      be.dl_end = be.cur;
      be.upload();
      be.run(Md5Op{ d, (long)(n / 64), finish ? (int)(n % 64) : 0, finish, total, st }, 1);
      be.download();
      memcpy(state, be.host(st), sizeof state);
      done += n;
      if (finish) break;
    }
    for (int i = 0; i < 16; i++) out[i] = (u8)(state[i >> 2] >> (8 * (i & 3)));
  }

  // kvz_rdoq (rdo.c:661-1000) for `count` intra blocks of one shape: what kvz_quantize_residual runs instead of kvz_quant with --rdoq
  // (quant-generic.c:234-244).  Serial per block: one item each.
  static void rdoq(B &be, int qp, double lambda, const u8 *ctx_states, const i16 *coef, i16 *dest, int width, int type, int scan_mode, int tr_depth, int count)
  {
    const int n = width * width;
    be.begin();
    const i16 *c = be.in(coef, (size_t)n * count);
    const u8 *cx = be.in(ctx_states, 160);
    i16 *d = be.template in_raw<i16>((size_t)n * count);  // uploaded: blocks without a significant level keep the caller's values, as in the reference
    memcpy(be.host_rw(d), dest, (size_t)n * count * sizeof(i16));
    be.mark_download_from(d);
    be.dl_end = be.cur;
    be.upload();
    int log2w = 2;
    while ((1 << log2w) < width) log2w++;
    be.run_wave(RdoqOp{ be.tables(), cx, lambda, qp, c, d, log2w, type, scan_mode, tr_depth }, count);
    be.download();
    memcpy(dest, be.host(d), (size_t)n * count * sizeof(i16));
  }

  static double fast_coeff_cost(B &be, const i16 *coeff, int width, uint64_t weights)
  {
    be.begin();
    const i16 *c = be.in(coeff, width * width);
    u32 *o = be.template zeroed<u32>(1);
    be.upload();
    be.run(FastCoeffCostOp{ c, weights, o }, width * width);
    be.download();
    return (double)*be.host(o) / 256.0;
  }
  static void find_last_scanpos(B &be, const i16 *coef, i16 *dest_coeff, int type, int q_bits, const i16 *quant_coeff, i32 *sig_coeff_inc,
                                u32 cg_size, uint16_t *ctx_set, const u32 *scan, i32 *cg_last_scanpos, i32 *last_scanpos, u32 cg_num, i32 *cg_scanpos, int width)
  {
    const int n = width * width;
    be.begin();
    const i16 *c = be.in(coef, n), *qc = be.in(quant_coeff, n);
    const u32 *sc = be.in(scan, n);
    i16 *d = be.template in_raw<i16>(n);
    memcpy(be.host_rw(d), dest_coeff, n * sizeof(i16));
    be.mark_download_from(d);
    i32 *res = be.template out<i32>(8);
    be.upload();
    be.run(FindLastOp{ c, d, qc, sc, type, q_bits, (int)cg_size, (int)cg_num, res }, 1);
    be.download();
    memcpy(dest_coeff, be.host(d), n * sizeof(i16));
    const i32 *r = be.host(res);
    *cg_scanpos = r[2];
    if (r[0] >= 0) { *last_scanpos = r[0]; *cg_last_scanpos = r[1]; *ctx_set = (uint16_t)r[3]; sig_coeff_inc[r[4]] = 0; }
  }

  // ---------------------------------------------------------------- intra
  static void intra_pred(B &be, int kind, int log2w, int mode, const u8 *above, const u8 *left, u8 *dst)
  {
    const int w = 1 << log2w, nref = 2 * w + 1;
    const int8_t m = (int8_t)mode;
    be.begin();
    const u8 *a = be.in(above, nref), *l = be.in(left, nref);
    const int8_t *md = be.in(&m, 1);
    u8 *o = be.template out<u8>(w * w);
    be.upload();
    be.run(IntraPredOp{ kind, log2w, md, a, l, 0, o }, w * w);
    be.download();
    memcpy(dst, be.host(o), w * w);
  }

  // ---------------------------------------------------------------- ipol
  // stages the (w + lpad + rpad) x (h + tpad + bpad) source window and returns the device address of src[0]
  static const u8 *stage_window(B &be, const u8 *src, int stride, int w, int h, int l, int r, int t, int b, int *dev_stride)
  {
    const int ww = w + l + r, wh = h + t + b;
    const u8 *base = be.in_rows(src - (long)t * stride - l, ww, wh, stride);
    *dev_stride = ww;
    return base + (long)t * ww + l;
  }
  static void sample(B &be, int chroma, int hi, const u8 *src, int src_stride, int w, int h, void *dst, int dst_stride, const i16 mv[2])
  {
    const int fx = chroma ? (mv[0] & 7) : (mv[0] & 3), fy = chroma ? (mv[1] & 7) : (mv[1] & 3);
    int ds;
    be.begin();
    const u8 *s = chroma ? stage_window(be, src, src_stride, w, h, 1, 2, 1, 2, &ds) : stage_window(be, src, src_stride, w, h, 3, 4, 3, 4, &ds);
    u8 *o8 = hi ? nullptr : be.template out<u8>(w * h);
    i16 *o16 = hi ? be.template out<i16>(w * h) : nullptr;
    be.upload();
    be.run(SampleOp{ be.tables(), chroma, s, ds, w, o8, o16, w, fx, fy }, w * h);
    be.download();
    for (int y = 0; y < h; y++) {
      if (hi) memcpy((i16 *)dst + (long)y * dst_stride, be.host(o16) + y * w, w * sizeof(i16));
      else memcpy((u8 *)dst + (long)y * dst_stride, be.host(o8) + y * w, w);
    }
  }
  // which: 0 hpel hor_ver, 1 hpel diag, 2 qpel hor_ver, 3 qpel diag (ipol-generic.c:213-679)
  static void fme_blocks(B &be, int which, const u8 *src, int src_stride, int w, int h, u8 *filtered, i16 *hor_intermediate, int fme_level,
                         i16 *hor_first_cols, int ox, int oy)
  {
    const int IMP = (64 + 7 + 1) * 64 + 1, COLN = 64 + 7 + 1;
    FmeOp op;
    int ds;
    be.begin();
    // rows -3 .. h+4, columns -3 .. w+4 are touched by the four functions
    const u8 *s = stage_window(be, src, src_stride, w, h, 3, 5, 3, 5, &ds);
    u8 *f = be.template out<u8>(4 * 4096);
    i16 *im[2] = { nullptr, nullptr }, *col[2] = { nullptr, nullptr };
    const int writes_im = which == 0 || which == 2;
    if (writes_im) for (int k = 0; k < 2; k++) { im[k] = be.template out<i16>(IMP); col[k] = be.template out<i16>(COLN); }
    be.upload();
    op.tb = be.tables(); op.src = s; op.src_stride = ds; op.w = w; op.h = h; op.filtered = f;
    fme_planes(which, ox, oy, op.pl);
    be.run(op, 4 * w * h);
    int filt[2] = { 0, 2 }, slot[2] = { 0, 1 }, cslot[2] = { 0, 2 }, y0[2] = { 0, fme_level > 1 ? 0 : 1 };
    if (which == 2) { filt[0] = ox != 0 ? 1 : 3; filt[1] = ox != 0 ? 3 : 1; slot[0] = 3; slot[1] = 4; cslot[0] = 1; cslot[1] = 3; y0[1] = 0; }
    if (writes_im) for (int k = 0; k < 2; k++) be.run(FmeHorOp{ be.tables(), s, ds, w, h, y0[k], filt[k], im[k], col[k] }, (h + 8 - y0[k]) * (w + 1));
    be.download();
    for (int p = 0; p < 4; p++) for (int y = 0; y < h; y++) memcpy(filtered + p * 4096 + y * 64, be.host(f) + p * 4096 + y * 64, w);
    if (writes_im) for (int k = 0; k < 2; k++) {
      for (int y = y0[k]; y < h + 8; y++) {
        memcpy(hor_intermediate + (long)slot[k] * IMP + y * 64, be.host(im[k]) + y * 64, w * sizeof(i16));
        hor_first_cols[cslot[k] * COLN + y] = be.host(col[k])[y];
      }
    }
  }
  static int get_extended_block(B &be, const kvz_hip_epol_params *a, const u8 *src, u8 *buf)
  {
    const int min_y = a->blk_y - a->pad_t, max_y = a->blk_y + a->blk_h + a->pad_b + a->pad_b_simd - 1;
    const int min_x = a->blk_x - a->pad_l, max_x = a->blk_x + a->blk_w + a->pad_r - 1;
    if (!(min_y < 0 || max_y >= a->src_h || min_x < 0 || max_x >= a->src_w)) return 0;
    const int ext_s = a->pad_l + a->blk_w + a->pad_r, rows = a->pad_t + a->blk_h + a->pad_b, total = rows + a->pad_b_simd;
    const int x0 = iclip(0, a->src_w - 1, min_x), x1 = iclip(0, a->src_w - 1, max_x);
    const int y0 = iclip(0, a->src_h - 1, min_y), y1 = iclip(0, a->src_h - 1, a->blk_y + a->blk_h + a->pad_b - 1);
    const int ww = x1 - x0 + 1, wh = y1 - y0 + 1;
    be.begin();
    const u8 *s = be.in_rows(src + (long)y0 * a->src_s + x0, ww, wh, a->src_s);
    u8 *o = be.template out<u8>(ext_s * total);
    be.upload();
    be.run(ExtBlockOp{ s, ww, wh, ww, min_x - x0, min_y - y0, ext_s, rows, total, o }, ext_s * total);
    be.download();
    memcpy(buf, be.host(o), ext_s * total);
    buf[(a->blk_h + a->pad_b + a->pad_t + a->pad_b_simd - 1) * ext_s + a->pad_l + a->blk_w + a->pad_r] = 0;  // ipol-generic.c:805
    return 1;
  }

  // ---------------------------------------------------------------- sao
  static void sao_edge(B &be, int mode, const u8 *orig, const u8 *rec, int bw, int bh, int eo_class, const int *offsets, i32 *stats_out, int nstats)
  {
    if (bw < 3 || bh < 3) return;
    SaoEdgeOp op;
    be.begin();
    op.orig = be.in(orig, bw * bh); op.rec = be.in(rec, bw * bh);
    op.stats = be.template zeroed<i32>(10);
    be.upload();
    op.mode = mode; op.eo_class = eo_class; op.bw = bw; op.bh = bh;
    for (int i = 0; i < 5; i++) op.offsets[i] = offsets ? offsets[i] : 0;
    be.run(op, (bw - 2) * (bh - 2));
    be.download();
    for (int i = 0; i < nstats; i++) stats_out[i] += be.host(op.stats)[i];
  }
  static int sao_band_ddistortion(B &be, int bitdepth, const u8 *orig, const u8 *rec, int bw, int bh, int band_pos, const int bands[4])
  {
    SaoBandOp op;
    be.begin();
    op.orig = be.in(orig, bw * bh); op.rec = be.in(rec, bw * bh);
    op.out = be.template zeroed<i32>(1);
    be.upload();
    op.shift = bitdepth - 5; op.band_pos = band_pos;
    for (int i = 0; i < 4; i++) op.bands[i] = bands[i];
    be.run(op, bw * bh);
    be.download();
    return *be.host(op.out);
  }
  static void sao_reconstruct_color(B &be, const kvz_hip_sao_params *sao, const u8 *rec, u8 *new_rec, int stride, int new_stride, int bw, int bh, int color)
  {
    SaoReconOp op;
    int ds;
    be.begin();
    const int ring = sao->type == 2 ? 1 : 0;  // edge classes read the 1-pixel ring around the block (sao.c:321-348)
    op.rec = stage_window(be, rec, stride, bw, bh, ring, ring, ring, ring, &ds);
    op.out = be.template out<u8>(bw * bh);
    be.upload();
    op.type = sao->type; op.eo_class = sao->eo_class; op.band_pos = sao->band_position[color == 2 ? 1 : 0]; op.offset_base = color == 2 ? 5 : 0;
    for (int i = 0; i < 10; i++) op.offsets[i] = sao->offsets[i];
    op.stride = ds; op.out_stride = bw; op.bw = bw;
    be.run(op, bw * bh);
    be.download();
    for (int y = 0; y < bh; y++) memcpy(new_rec + (long)y * new_stride, be.host(op.out) + y * bw, bw);
  }
};

}  // namespace kvz
