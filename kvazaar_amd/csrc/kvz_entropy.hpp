// kvz_entropy.hpp -- kvazaar's entropy coder in its REAL mode on the device: the slice data of I pictures from what the CTU pass left in HBM.
//
// Reference: encoder_state_worker_encode_lcu_bitstream (encoderstate.c:676-745: SAO syntax :467-552, kvz_encode_coding_tree, end_of_slice_segment_flag,
// end_of_subset_one_bit, kvz_cabac_finish + byte alignment per substream), kvz_encode_coding_tree of an I slice (encode_coding_tree.c:745-900, :467-652, :193-310,
// :117-190, :63-115), kvz_encode_coeff_nxn (strategies/generic/encode_coding_tree-generic.c:40-283, cabac.c:275-301) and the arithmetic coder (cabac.c:85-270);
// WPP: a row's contexts start from the row above after its second CTU (encoderstate.c:763-771).  oracle/kvz_oracle_entropy.inc is the CPU restatement, byte-identical
// to the reference encoder's slice data.
//
// The coder is serial per substream, but its two halves are not equally serial, so the work is cut at the bin:
//   1. entropy_ctu_bins      one lane per CTU, all CTUs of all pictures at once: walks the CTU's syntax and writes its BINS as 32-bit records -- a context-coded bin
//                            (context index, value), a run of bypass bins, a terminating bin.  Which context a bin uses never depends on a context's state, so this
//                            half has no dependency between CTUs at all (neighbour CU depths / modes come from the frame-level maps).  A CTU's list has a fixed
//                            capacity (6-9 k records at QP 22, 24 k for noise at QP 12); a list that outgrows it is reported and the host runs the chunk again.
//   2. entropy_row_contexts  one lane per picture: the context states each CTU row starts from -- the state machine run over the first two CTUs' context-coded bins
//                            of every row, row after row (with --no-wpp there is nothing to do).
//   3. entropy_code_row_wide one lane per substream (picture x CTU row; one per picture without WPP): the arithmetic coder proper over the row's records -- range
//                            subdivision, renormalisation, the code value leaving 32 bits at a time with its carries -- into scratch sized by an upper bound
//                            stage 1 keeps (6 bits per context-coded bin, 7 per terminating bin, the bypass bins);
//   4. emulation prevention  one workgroup per substream counts the 0x03 bytes it needs (a property of the finished bytes, decided by position); a copy kernel then
//                            packs the substreams back to back, with them.
// The same functions compile for the host (tests/hostsim) where every lane is a loop iteration.
#pragma once
#include "kvz_ops.hpp"
#include "kvz_sao.hpp"
#include "kvz_residual.hpp"
#include "../../include/kvz_hip_types.h"
#include "../../include/kvz_hip_dev.h"

namespace kvz {

#define KVZ_ENTROPY_CTXS 168
// contexts of the inter syntax, behind the KVZ_HIP_CX_* ones (cabac.h:63-100: cu_skip_flag_model[3], cu_merge_flag_ext_model, cu_merge_idx_ext_model, cu_pred_mode_model,
// cu_mvd_model[2], mvp_idx_model[2], inter_dir[5], cu_qt_root_cbf_model)
enum { KVZ_EB_CX_SKIP = 150, KVZ_EB_CX_MERGE_FLAG = 153, KVZ_EB_CX_MERGE_IDX = 154, KVZ_EB_CX_PRED_MODE = 155, KVZ_EB_CX_MVD = 156, KVZ_EB_CX_MVP_IDX = 158,
       KVZ_EB_CX_INTER_DIR = 160, KVZ_EB_CX_ROOT_CBF = 165 };

struct EntropyJob {
  int W, H, wc, hc, n_frames, no_wpp;
  const u8 *depth, *mode;      // [frames][(H/8)*(W/8)]
  const u8 *part, *mode4;      // [frames][(H/8)*(W/8)], [frames][(H/4)*(W/4)] or null (no NxN CUs)
  const i16 *coeff;            // [frames][ctu][KVZ_HIP_CTU_COEFFS]
  const SaoRec *sao;           // [frames][ctu][3] packed decisions or null (SAO off)
  const u8 *sao_merge;         // [frames][ctu]: 0 none, 1 left, 2 up
  const u8 *not_last;          // [frames] or null: 1 = the picture is a tile that is not its slice's last: it ends in end_of_subset_one_bit, not end_of_slice_segment_flag
  u32 *bins;                   // [frames * ctus][cap] records, cap a multiple of 16 (lists are read a 64-byte line at a time)
  u32 *nbins;                  // [frames * ctus] records the CTU produced; above cap the list is truncated (still counted) and the host runs the chunk again with room
  u32 cap;
  u32 *nbits;                  // [frames * ctus] upper bound of the bits the CTU's records make the coder emit (sizes the substreams' scratch)
  u8 *row_ctx;                 // [frames][hc][KVZ_ENTROPY_CTXS] context states at the start of every row
  // B pictures (kvz_hip_dev_entropy_code_inter): the CU records of the pictures and of their reference pictures, one per 4x4 unit; then depth / mode / part are unused
  const kvz_hip_cu_info *cu, *ref_cu;  // [frames][(H/4)*(W/4)]
  int poc;                     // temporal MV predictors need poc > 1 (inter.c:1290)
  u8 ctx_init[KVZ_ENTROPY_CTXS];  // the slice's initial states: kvz_hip_intra_cost_model::ctx_init (KVZ_HIP_CX_* order); B slices: + the KVZ_EB_CX_* contexts
};
#define KVZ_EB_CTX(ctx, v) ((u32)(ctx) | ((u32)(v) << 8))
#define KVZ_EB_EP(value, n) (0x40000000u | ((u32)(n) << 16) | ((u32)(value) & 0xffffu))
#define KVZ_EB_TRM(v) (0x80000000u | ((u32)(v) << 8) | 168u)  /* the value where a context-coded bin has it, on the coder's pseudo-context (KVZ_ENTROPY_CTX_NEUTRAL) */

struct BinSink {
  u32 *out; u32 n, cap;
  u32 bits;  // upper bound of the bits these records make the coder emit: 6 per context-coded bin (the longest renormalisation), 7 per terminating bin, bypass bins as they are
  // Runs of bypass bins that follow each other (the signs of a group, then the prefix and the suffix of every remaining level) are gathered into records of up to 16 bins:
  // kvz_cabac_encode_bins_ep of a run codes the same bytes however the run is cut (low' = (low << n) + range * value is exact), and the coder's serial chain is one step
  // per record.  pend_v / pend_n: the run being gathered.
  u32 pend_v = 0; int pend_n = 0;
  KVZ_DEV void put(u32 r) { if (n < cap) out[n] = r; n++; }
  KVZ_DEV void finish() { if (pend_n > 0) { put(KVZ_EB_EP(pend_v, pend_n)); pend_n = 0; pend_v = 0; } }
  KVZ_DEV void ctx(int c, int v) { finish(); put(KVZ_EB_CTX(c, v ? 1 : 0)); bits += 6; }
  KVZ_DEV void ep(u32 value, int bits)
  {
    this->bits += (u32)bits;
    while (bits > 0) {
      const int take = bits < 16 - pend_n ? bits : 16 - pend_n;  // the leading `take` bins of the run join the record being gathered
      bits -= take;
      pend_v = (pend_v << take) | (value >> bits);
      pend_n += take;
      value &= (1u << bits) - 1;
      if (pend_n == 16) finish();
    }
  }
  KVZ_DEV void trm(int v) { finish(); put(KVZ_EB_TRM(v)); bits += 7; }
  // the residual of the transform block whose levels start `off` into the CTU's block: here and now
  KVZ_DEV void tu(const Tables *tb, const i16 *ctu, int off, int log2_size, int type, int scan_mode) { entropy_coeff_nxn(*this, tb, ctu + off, log2_size, type, scan_mode); }
};
// ... or later: a sink that queues what follows the first transform block of a CU -- records as they are, blocks as descriptors (kind 3: offset | log2 - 2 << 13 |
// chroma << 15 | scan << 16) -- so that the bin stage can run the blocks of all its lanes a coefficient group at a time (entropy_ctu_bins_phased).  A CU queues at most
// 26 items (a 64x64 CU: two flags, then four times two chroma flags, a luma flag and three blocks).
#define KVZ_EB_TU(off, log2_size, type, scan) (0xc0000000u | (u32)(off) | ((u32)((log2_size) - 2) << 13) | ((u32)((type) != 0) << 15) | ((u32)(scan) << 16))
struct DeferSink {
  enum { QCAP = 32 };
  BinSink out;
  u32 *q;  // QCAP items of this lane, `qs` words apart (LDS: item-major, the lanes of a wavefront side by side)
  int qs, head, tail;
  KVZ_DEV bool queued() const { return head < tail; }
  KVZ_DEV u32 front() const { return q[head * qs]; }
  KVZ_DEV void item(u32 r) { if (head == tail) head = tail = 0; q[tail++ * qs] = r; }
  KVZ_DEV void ctx(int c, int v) { if (queued()) item(KVZ_EB_CTX(c, v ? 1 : 0)); else out.ctx(c, v); }
  KVZ_DEV void ep(u32 value, int bits)
  {
    if (!queued()) { out.ep(value, bits); return; }
    while (bits > 16) { bits -= 16; item(KVZ_EB_EP(value >> bits, 16)); value &= (1u << bits) - 1; }
    if (bits > 0) item(KVZ_EB_EP(value, bits));
  }
  KVZ_DEV void trm(int v) { if (queued()) item(KVZ_EB_TRM(v)); else out.trm(v); }
  KVZ_DEV void tu(const Tables *, const i16 *, int off, int log2_size, int type, int scan_mode) { item(KVZ_EB_TU(off, log2_size, type, scan_mode)); }
  // the records in front of the queue go out; true: a block is at its head now
  KVZ_DEV bool flush_records()
  {
    while (head < tail && (front() >> 30) != 3u) {
      const u32 r = q[head++ * qs], kind = r >> 30;
      if (kind == 0) out.ctx((int)(r & 0xff), (int)((r >> 8) & 1)); else if (kind == 1) out.ep(r & 0xffffu, (int)((r >> 16) & 0x3fu)); else out.trm((int)((r >> 8) & 1));
    }
    return head < tail;
  }
};

KVZ_DEV unsigned entropy_zorder(int x, int y)  // cu.h:385-421 with width 64: Morton index of the 4x4 block times 16
{
  unsigned r = 0;
  for (int b = 0; b < 4; b++) r |= (((unsigned)(x >> (2 + b)) & 1u) << (2 * b)) | (((unsigned)(y >> (2 + b)) & 1u) << (2 * b + 1));
  return r * 16;
}
KVZ_DEV bool entropy_any(const i16 *c, int n)
{
  // levels are stored as 16-bit values, blocks are 8-byte aligned (16 levels at least)
  const unsigned long long *q = (const unsigned long long *)c;
  for (int i = 0; i < n / 4; i++) if (q[i]) return true;
  return false;
}
KVZ_DEV int entropy_scan_order(int mode, int depth)  // encoderstate.c:1761-1775 kvz_get_scan_order, intra
{
  if (depth >= 3) {
    if (mode >= 6 && mode <= 14) return 2;
    if (mode >= 22 && mode <= 30) return 1;
  }
  return 0;
}

// Per-call form (strategies-encode.h:49-65 kvz_encode_coeff_nxn): the bins of ONE transform block.  out[0] = number of records, out[1..] the records (as many as fit
// `cap`); the caller -- integration/kvazaar/strategies/hip/encode-hip.c -- feeds them to the cabac_data_t it was handed.
struct CoeffBinsOp {
  const Tables *tb; const i16 *coeff; int log2_size, type, scan_mode; u32 *out; u32 cap;
  KVZ_DEV void operator()(int) const
  {
    BinSink s{ out + 1, 0, cap, 0 };
    entropy_coeff_nxn(s, tb, coeff, log2_size, type, scan_mode);
    s.finish();
    out[0] = s.n;
  }
};

struct EntropyCtu {  // one CTU of one picture
  const EntropyJob &J;
  const Tables *tb;
  const u8 *depth, *mode, *part, *mode4;
  const i16 *ctu;
  int w8, w4;
  KVZ_DEV int mode_at(int x, int y) const
  {
    if (part && part[(y >> 3) * w8 + (x >> 3)]) return mode4[(y >> 2) * w4 + (x >> 2)];
    return mode[(y >> 3) * w8 + (x >> 3)];
  }
  // cbf_is_set(cu->cbf, depth, plane) read off the levels (cu.h:510-569: the block or any block below it)
  KVZ_DEV bool cbf(int c, int xl, int yl, int depth) const
  {
    const int w = 64 >> depth;
    const int cw = depth >= 3 ? 4 : w / 2;
    if (c == 0) return entropy_any(ctu + entropy_zorder(xl, yl), w * w);
    return entropy_any(ctu + (c == 1 ? 4096 : 5120) + entropy_zorder((xl & ~7) / 2, (yl & ~7) / 2), cw * cw);
  }
  // encode_transform_unit (encode_coding_tree.c:117-190) of the block at (x, y), tree depth `depth`
  template <class S> KVZ_DEV void transform_unit(S &s, int x, int y, int depth, bool cb_y, bool cu_u, bool cu_v) const
  {
    const int xl = x & 63, yl = y & 63, log2w = 6 - depth, log2c = depth == 4 ? 2 : log2w - 1;
    if (cb_y) s.tu(tb, ctu, (int)entropy_zorder(xl, yl), log2w, 0, entropy_scan_order(mode_at(x, y), depth));
    if (depth == 4 && (x % 8 == 0 || y % 8 == 0)) return;  // the 4x4 chroma blocks follow the last luma block, under the first PU's mode
    const int cscan = entropy_scan_order(mode_at(x & ~7, y & ~7), depth), cxl = (xl & ~7) / 2, cyl = (yl & ~7) / 2;
    if (cu_u) s.tu(tb, ctu, 4096 + (int)entropy_zorder(cxl, cyl), log2c, 2, cscan);
    if (cu_v) s.tu(tb, ctu, 5120 + (int)entropy_zorder(cxl, cyl), log2c, 2, cscan);
  }
  // encode_transform_coeff (encode_coding_tree.c:193-310) of an intra CU: one level of implicit split at most (64x64 CUs, NxN CUs), no split_transform_flag is ever coded
  template <class S> KVZ_DEV void transform_tree(S &s, int x, int y, int depth, bool nxn) const
  {
    const int xl = x & 63, yl = y & 63;
    const bool cu_u = cbf(1, xl, yl, depth), cu_v = cbf(2, xl, yl, depth);
    s.ctx(KVZ_HIP_CX_CBF_CHROMA, cu_u);
    s.ctx(KVZ_HIP_CX_CBF_CHROMA, cu_v);
    if (depth > 0 && !nxn) {  // one transform block
      const bool cb_y = cbf(0, xl, yl, depth);
      s.ctx(KVZ_HIP_CX_CBF_LUMA + 1, cb_y);
      if (cb_y | cu_u | cu_v) transform_unit(s, x, y, depth, cb_y, cu_u, cu_v);
      return;
    }
    const int o = 64 >> (depth + 1);
    for (int q = 0; q < 4; q++) {
      const int qx = x + (q & 1) * o, qy = y + (q >> 1) * o, d = depth + 1;
      bool cb_u = cu_u, cb_v = cu_v;
      if (d < 4) {  // chroma flags of the 32x32 units, when the CU's were set
        cb_u = cu_u && cbf(1, qx & 63, qy & 63, d);
        cb_v = cu_v && cbf(2, qx & 63, qy & 63, d);
        if (cu_u) s.ctx(KVZ_HIP_CX_CBF_CHROMA + 1, cb_u);
        if (cu_v) s.ctx(KVZ_HIP_CX_CBF_CHROMA + 1, cb_v);
      }
      const bool cb_y = cbf(0, qx & 63, qy & 63, d);
      s.ctx(KVZ_HIP_CX_CBF_LUMA, cb_y);
      if (cb_y | cb_u | cb_v) transform_unit(s, qx, qy, d, cb_y, cb_u, cb_v);
    }
  }
  // intra.c:84-126 kvz_intra_get_dir_luma_predictor
  KVZ_DEV static void mpm_candidates(int l, int a, int preds[3])
  {
    if (l == a) {
      if (l > 1) { preds[0] = l; preds[1] = ((l + 29) % 32) + 2; preds[2] = ((l - 1) % 32) + 2; }
      else { preds[0] = 0; preds[1] = 1; preds[2] = 26; }
    } else {
      preds[0] = l; preds[1] = a;
      if (l && a) preds[2] = 0; else preds[2] = (l + a) < 2 ? 26 : 1;
    }
  }
  // the leaf of kvz_encode_coding_tree: part_mode, encode_intra_coding_unit (encode_coding_tree.c:467-652), the transform tree
  template <class S> KVZ_DEV void coding_unit(S &s, int x, int y, int depth) const
  {
    const bool nxn = depth == 3 && part && part[(y >> 3) * w8 + (x >> 3)];
    if (depth == 3) s.ctx(KVZ_HIP_CX_PART, !nxn);
    const int n_pu = nxn ? 4 : 1;
    int preds[4][3], mpm[4], modes[4];
    for (int j = 0; j < n_pu; j++) {
      const int px = x + 4 * (j & 1), py = y + 4 * (j >> 1);
      modes[j] = mode_at(px, py);
      const int l = px > 0 ? mode_at(px - 1, py) : 1, a = (py % 64 > 0 && py > 0) ? mode_at(px, py - 1) : 1;
      mpm_candidates(l, a, preds[j]);
      mpm[j] = -1;
      for (int i = 2; i >= 0; i--) if (preds[j][i] == modes[j]) mpm[j] = i;
    }
    for (int j = 0; j < n_pu; j++) s.ctx(KVZ_HIP_CX_INTRA, mpm[j] != -1);
    for (int j = 0; j < n_pu; j++) {
      if (mpm[j] != -1) {
        s.ep(mpm[j] == 0 ? 0 : 1, 1);
        if (mpm[j] != 0) s.ep(mpm[j] == 1 ? 0 : 1, 1);
      } else {
        int *q = preds[j], t;
        if (q[0] > q[1]) { t = q[0]; q[0] = q[1]; q[1] = t; }
        if (q[0] > q[2]) { t = q[0]; q[0] = q[2]; q[2] = t; }
        if (q[1] > q[2]) { t = q[1]; q[1] = q[2]; q[2] = t; }
        int rem = modes[j];
        for (int i = 2; i >= 0; i--) rem = rem > q[i] ? rem - 1 : rem;
        s.ep((u32)rem, 5);
      }
    }
    s.ctx(KVZ_HIP_CX_CHROMA, 0);  // intra_chroma_pred_mode 4: the luma mode
    transform_tree(s, x, y, depth, nxn);
  }
  // kvz_encode_coding_tree (encode_coding_tree.c:745-900) without recursion: a stack of (x, y, depth) nodes, children pushed in reverse coding order
  // ... one node of it: a split node pushes its children, a leaf is coded (node: x offset >> 3 in bits 0..3, y offset >> 3 in bits 4..7, depth in bits 8..9)
  template <class S> KVZ_DEV void coding_tree(S &s, int cx, int cy) const
  {
    u16 stack[16];
    int sp = 0;
    stack[sp++] = 0;
    while (sp > 0) tree_step(s, stack, sp, cx, cy);
  }
  template <class S, class STK> KVZ_DEV void tree_step(S &s, STK &stack, int &sp, int cx, int cy) const
  {
    {
      const int node = stack[--sp], depth = node >> 8, x = cx + ((node & 15) << 3), y = cy + (((node >> 4) & 15) << 3);
      const int w = 64 >> depth, half = w >> 1;
      const int cur_depth = this->depth[(y >> 3) * w8 + (x >> 3)];
      const bool split_flag = cur_depth > depth;
      const bool border_x = J.W < x + w, border_y = J.H < y + w, border = border_x || border_y;
      const bool border_split_x = J.W >= x + 8 + half, border_split_y = J.H >= y + 8 + half;
      if (depth != 3) {
        if (!border) {
          int sm = 0;
          if (x > 0 && this->depth[(y >> 3) * w8 + ((x - 1) >> 3)] > depth) sm++;
          if (y > 0 && this->depth[((y - 1) >> 3) * w8 + (x >> 3)] > depth) sm++;
          s.ctx(KVZ_HIP_CX_SPLIT + sm, split_flag);
        }
        if (split_flag || border) {
          const int h8 = half >> 3, base = (node & 0xff) | ((depth + 1) << 8);
          if (!border || (border_split_x && border_split_y)) stack[sp++] = (u16)(base + h8 + (h8 << 4));
          if (!border_y || border_split_y) stack[sp++] = (u16)(base + (h8 << 4));
          if (!border_x || border_split_x) stack[sp++] = (u16)(base + h8);
          stack[sp++] = (u16)base;
          return;
        }
      }
      coding_unit(s, x, y, depth);
    }
  }
  // encode_sao_color (encoderstate.c:467-517) from the packed decision records (kvz_sao.hpp SaoRec: type | class << 8 | band << 16 | five offsets from bit 24)
  template <class S> KVZ_DEV static void sao_color(S &s, SaoRec first, SaoRec own, int color)
  {
    const int type = (int)(first & 0xff);
    if (color != 2) {
      s.ctx(KVZ_HIP_CX_SAO_TYPE, type != 0);
      if (type == 1) s.ep(0, 1); else if (type == 2) s.ep(1, 1);
    }
    if (type == 0) return;
    int off[5];
    for (int k = 0; k < 5; k++) off[k] = (int)(int8_t)(own >> (24 + 8 * k));
    for (int i = 1; i <= 4; i++) {  // kvz_cabac_write_unary_max_symbol_ep (cabac.c:526-551), max 7
      const int symbol = iabs(off[i]);
      if (!symbol) { s.ep(0, 1); continue; }
      const int n = symbol + (symbol < 7);
      s.ep(((1u << symbol) - 1) << (symbol < 7), n);
    }
    if (type == 1) {
      for (int i = 1; i <= 4; i++) if (off[i] != 0) s.ep(off[i] < 0 ? 1 : 0, 1);
      s.ep((u32)((own >> 16) & 0xff), 5);
    } else if (color != 2) {
      s.ep((u32)((first >> 8) & 0xff), 2);
    }
  }
};

// ---- B pictures: kvz_encode_coding_tree with the inter syntax, from the frame-level CU records of the inter CTU pass (kvz_hip_dev_inter_ctu_pass) ----
// One reference picture per list, 2Nx2N CUs, max_merge 5 (BASELINE config 4's presets).  What the records do not hold are the MV predictors the MVDs are coded against:
// kvz_inter_get_mv_cand_cua (inter.c:1330-1352) derived again here from the neighbours' records -- all of them final by now, availability is a matter of coding order.
struct EntropyCtuB {
  const EntropyJob &J;
  const Tables *tb;
  const kvz_hip_cu_info *cu, *ref_cu;  // the picture's, the reference picture's
  const i16 *ctu;
  int w4, cx, cy;
  KVZ_DEV const kvz_hip_cu_info &at(int x, int y) const { return cu[(y >> 2) * w4 + (x >> 2)]; }
  KVZ_DEV static bool a0_coded(int x, int y, int width, int height)  // inter.c:686-741 is_a0_cand_coded
  {
    int size = (width & -width) < (height & -height) ? (width & -width) : (height & -height);
    if (height != size) y = y + height - size;
    while (size < 64) {
      const int parent = 2 * size, idx = ((x & (parent - 1)) != 0) + 2 * ((y & (parent - 1)) != 0);
      if (idx == 0) return true;
      if (idx == 1 || idx == 3) return false;
      y -= size; size = parent;
    }
    return false;
  }
  KVZ_DEV static bool b0_coded(int x, int y, int width, int height)  // inter.c:743-798 is_b0_cand_coded
  {
    int size = (width & -width) < (height & -height) ? (width & -width) : (height & -height);
    if (width != size) x = x + width - size;
    while (size < 64) {
      const int parent = 2 * size, idx = ((x & (parent - 1)) != 0) + 2 * ((y & (parent - 1)) != 0);
      if (idx == 0 || idx == 2) return true;
      if (idx == 3) return false;
      x -= size; size = parent;
    }
    return true;
  }
  KVZ_DEV static bool add_mvp(const kvz_hip_cu_info &c, bool valid, int reflist, i16 out[2])  // inter.c:1121-1145 add_mvp_candidate, one reference picture
  {
    if (!valid) return false;
    for (int i = 0; i < 2; i++) {
      const int l = i == 0 ? reflist : !reflist;
      if ((c.mv_dir & (1 << l)) == 0) continue;
      out[0] = l ? c.mv[1][0] : c.mv[0][0]; out[1] = l ? c.mv[1][1] : c.mv[0][1];  // (selects: an index computed at run time puts the record on the stack)
      return true;
    }
    return false;
  }
  KVZ_DEV void mv_candidates(int x, int y, int w, int h, int reflist, i16 mv_cand[2][2]) const  // inter.c:1147-1352
  {
    const int xl = x - cx, yl = y - cy;
    kvz_hip_cu_info a[2], b[3], col;
    bool va[2] = { false, false }, vb[3] = { false, false, false }, vcol = false;
    if (x != 0) {
      const kvz_hip_cu_info &a1 = at(x - 1, y + h - 1);
      if (a1.type == 2) { a[1] = a1; va[1] = true; }
      if (yl + h < 64 && y + h < J.H) { const kvz_hip_cu_info &a0 = at(x - 1, y + h); if (a0.type == 2 && a0_coded(x, y, w, h)) { a[0] = a0; va[0] = true; } }
    }
    if (y != 0) {
      if (x + w < J.W && (xl + w < 64 || yl == 0)) { const kvz_hip_cu_info &b0 = at(x + w, y - 1); if (b0.type == 2 && b0_coded(x, y, w, h)) { b[0] = b0; vb[0] = true; } }
      const kvz_hip_cu_info &b1 = at(x + w - 1, y - 1);
      if (b1.type == 2) { b[1] = b1; vb[1] = true; }
      if (x != 0) { const kvz_hip_cu_info &b2 = at(x - 1, y - 1); if (b2.type == 2) { b[2] = b2; vb[2] = true; } }
    }
    if (J.poc >= 1) {  // inter.c:1204-1260: H (below right, not across the CTU row) before C3 (centre), at 16-sample granularity in the reference picture
      const int xbr = x + w, ybr = y + h, xc = x + w / 2, yc = y + h / 2;
      bool vh = false;
      if (xbr < J.W && ybr < J.H && ybr % 64 != 0) { const kvz_hip_cu_info &c = ref_cu[(((ybr >> 4) << 4) >> 2) * w4 + (((xbr >> 4) << 4) >> 2)]; if (c.type == 2) { col = c; vh = true; } }
      vcol = vh;
      if (!vh && xc < J.W && yc < J.H) { const kvz_hip_cu_info &c = ref_cu[(((yc >> 4) << 4) >> 2) * w4 + (((xc >> 4) << 4) >> 2)]; if (c.type == 2) { col = c; vcol = true; } }
    }
    int n = 0, nb_b = 0;
    for (int i = 0; i < 2; i++) if (add_mvp(a[i], va[i], reflist, mv_cand[n])) { n++; break; }
    if (n == 0) for (int i = 0; i < 2; i++) if (add_mvp(a[i], va[i], reflist, mv_cand[n])) { n++; break; }
    for (int i = 0; i < 3; i++) if (add_mvp(b[i], vb[i], reflist, mv_cand[n < 2 ? n : 1])) { nb_b++; break; }
    n += nb_b;
    if (va[0] || va[1]) nb_b = 1; else if (n != 2) nb_b = 0;
    if (!nb_b) for (int i = 0; i < 3; i++) if (add_mvp(b[i], vb[i], reflist, mv_cand[n < 2 ? n : 1])) { n++; break; }
    if (n == 2 && mv_cand[0][0] == mv_cand[1][0] && mv_cand[0][1] == mv_cand[1][1]) n = 1;
    if (J.poc > 1 && n < 2 && vcol) {
      int col_list = reflist;
      if ((col.mv_dir & (col_list + 1)) == 0) col_list = 1 - col_list;
      mv_cand[n][0] = col_list ? col.mv[1][0] : col.mv[0][0]; mv_cand[n][1] = col_list ? col.mv[1][1] : col.mv[0][1];
      n++;
    }
    while (n < 2) { mv_cand[n][0] = 0; mv_cand[n][1] = 0; n++; }
  }
  template <class S> KVZ_DEV static void merge_idx(S &s, int idx)  // encode_coding_tree.c:323-338
  {
    for (int ui = 0; ui < 4; ui++) {
      const int symbol = ui != idx;
      if (ui == 0) s.ctx(KVZ_EB_CX_MERGE_IDX, symbol); else s.ep((u32)symbol, 1);
      if (!symbol) break;
    }
  }
  template <class S> KVZ_DEV static void ex_golomb(S &s, u32 symbol, u32 count)  // cabac.c:556-586 kvz_cabac_write_ep_ex_golomb
  {
    u32 bins = 0;
    int num_bins = 0;
    while (symbol >= (1u << count)) { bins = 2 * bins + 1; ++num_bins; symbol -= 1u << count; ++count; }
    bins = 2 * bins; ++num_bins;
    bins = (bins << count) | symbol;
    s.ep(bins, num_bins + (int)count);
  }
  template <class S> KVZ_DEV static void mvd(S &s, int hor, int ver)  // encode_coding_tree.c:1062-1112 kvz_encode_mvd
  {
    const u32 ah = (u32)iabs(hor), av = (u32)iabs(ver);
    s.ctx(KVZ_EB_CX_MVD, hor != 0);
    s.ctx(KVZ_EB_CX_MVD, ver != 0);
    if (hor) s.ctx(KVZ_EB_CX_MVD + 1, ah > 1);
    if (ver) s.ctx(KVZ_EB_CX_MVD + 1, av > 1);
    if (hor) { if (ah > 1) ex_golomb(s, ah - 2, 1); s.ep(hor > 0 ? 0 : 1, 1); }
    if (ver) { if (av > 1) ex_golomb(s, av - 2, 1); s.ep(ver > 0 ? 0 : 1, 1); }
  }
  KVZ_DEV static bool cbf_set(u32 cbf, int depth, int plane) { const u32 masks[5] = { 0x1f, 0x0f, 0x07, 0x03, 0x1 }; return (cbf & (masks[depth] << (5 * plane))) != 0; }  // cu.h:510-569
  template <class S> KVZ_DEV void coding_unit(S &s, int x, int y, int depth) const
  {
    const kvz_hip_cu_info &cur = at(x, y);
    const int w = 64 >> depth, xl = x - cx, yl = y - cy, log2w = 6 - depth, log2c = depth == 3 ? 2 : log2w - 1;
    int skip_ctx = 0;
    if (x) skip_ctx += at(x - 1, y).skipped;
    if (y) skip_ctx += at(x, y - 1).skipped;
    s.ctx(KVZ_EB_CX_SKIP + skip_ctx, cur.skipped);
    if (cur.skipped) { merge_idx(s, cur.merge_idx); return; }
    s.ctx(KVZ_EB_CX_PRED_MODE, cur.type == 1);
    if (cur.type == 2 || depth == 3) s.ctx(KVZ_HIP_CX_PART, 1);  // 2Nx2N
    const bool cb_y = cbf_set(cur.cbf, depth, 0), cb_u = cbf_set(cur.cbf, depth, 1), cb_v = cbf_set(cur.cbf, depth, 2);
    int scan = 0;
    if (cur.type == 2) {
      s.ctx(KVZ_EB_CX_MERGE_FLAG, cur.merged);
      if (cur.merged) merge_idx(s, cur.merge_idx);
      else {
        const int inter_dir = cur.mv_dir - 1;
        s.ctx(KVZ_EB_CX_INTER_DIR + depth, inter_dir == 2);
        if (inter_dir < 2) s.ctx(KVZ_EB_CX_INTER_DIR + 4, inter_dir);
        for (int l = 0; l < 2; l++) {
          if (!(cur.mv_dir & (1 << l))) continue;
          i16 cand[2][2];
          mv_candidates(x, y, w, w, l, cand);
          const int k = cur.mv_cand[l];
          mvd(s, cur.mv[l][0] - cand[k][0], cur.mv[l][1] - cand[k][1]);
          s.ctx(KVZ_EB_CX_MVP_IDX, k);
        }
      }
      const bool any = cb_y || cb_u || cb_v;
      if (!cur.merged) s.ctx(KVZ_EB_CX_ROOT_CBF, any);
      if (!any) return;
      s.ctx(KVZ_HIP_CX_CBF_CHROMA, cb_u);
      s.ctx(KVZ_HIP_CX_CBF_CHROMA, cb_v);
      if (cb_u || cb_v) s.ctx(KVZ_HIP_CX_CBF_LUMA + 1, cb_y);
    } else {
      int l = 1, a = 1;
      if (x > 0 && at(x - 1, y).type == 1) l = at(x - 1, y).mode;
      if (y % 64 > 0 && y > 0 && at(x, y - 1).type == 1) a = at(x, y - 1).mode;
      int preds[3], mpm = -1;
      EntropyCtu::mpm_candidates(l, a, preds);
      for (int i = 2; i >= 0; i--) if (preds[i] == cur.mode) mpm = i;
      s.ctx(KVZ_HIP_CX_INTRA, mpm != -1);
      if (mpm != -1) { s.ep(mpm == 0 ? 0 : 1, 1); if (mpm != 0) s.ep(mpm == 1 ? 0 : 1, 1); }
      else {
        int t;
        if (preds[0] > preds[1]) { t = preds[0]; preds[0] = preds[1]; preds[1] = t; }
        if (preds[0] > preds[2]) { t = preds[0]; preds[0] = preds[2]; preds[2] = t; }
        if (preds[1] > preds[2]) { t = preds[1]; preds[1] = preds[2]; preds[2] = t; }
        int rem = cur.mode;
        for (int i = 2; i >= 0; i--) rem = rem > preds[i] ? rem - 1 : rem;
        s.ep((u32)rem, 5);
      }
      s.ctx(KVZ_HIP_CX_CHROMA, 0);
      s.ctx(KVZ_HIP_CX_CBF_CHROMA, cb_u);
      s.ctx(KVZ_HIP_CX_CBF_CHROMA, cb_v);
      s.ctx(KVZ_HIP_CX_CBF_LUMA + 1, cb_y);
      scan = entropy_scan_order(cur.mode, depth);
    }
    if (cb_y) s.tu(tb, ctu, (int)entropy_zorder(xl, yl), log2w, 0, scan);
    if (cb_u) s.tu(tb, ctu, 4096 + (int)entropy_zorder(xl / 2, yl / 2), log2c, 2, scan);
    if (cb_v) s.tu(tb, ctu, 5120 + (int)entropy_zorder(xl / 2, yl / 2), log2c, 2, scan);
  }
  template <class S> KVZ_DEV void coding_tree(S &s) const
  {
    u16 stack[16];
    int sp = 0;
    stack[sp++] = 0;
    while (sp > 0) tree_step(s, stack, sp);
  }
  template <class S, class STK> KVZ_DEV void tree_step(S &s, STK &stack, int &sp) const
  {
    {
      const int node = stack[--sp], depth = node >> 8, x = cx + ((node & 15) << 3), y = cy + (((node >> 4) & 15) << 3);
      const int w = 64 >> depth, half = w >> 1;
      const bool split_flag = at(x, y).depth > depth;
      const bool border_x = J.W < x + w, border_y = J.H < y + w, border = border_x || border_y;
      const bool border_split_x = J.W >= x + 8 + half, border_split_y = J.H >= y + 8 + half;
      if (depth != 3) {
        if (!border) {
          int sm = 0;
          if (x > 0 && at(x - 1, y).depth > depth) sm++;
          if (y > 0 && at(x, y - 1).depth > depth) sm++;
          s.ctx(KVZ_HIP_CX_SPLIT + sm, split_flag);
        }
        if (split_flag || border) {
          const int h8 = half >> 3, base = (node & 0xff) | ((depth + 1) << 8);
          if (!border || (border_split_x && border_split_y)) stack[sp++] = (u16)(base + h8 + (h8 << 4));
          if (!border_y || border_split_y) stack[sp++] = (u16)(base + (h8 << 4));
          if (!border_x || border_split_x) stack[sp++] = (u16)(base + h8);
          stack[sp++] = (u16)base;
          return;
        }
      }
      coding_unit(s, x, y, depth);
    }
  }
};

// stage 1: the bins of CTU `item` (frame-major, raster CTU order inside a frame)
KVZ_DEV void entropy_ctu_bins(const EntropyJob &J, const Tables *tb, long item)
{
  const int ctus = J.wc * J.hc, f = (int)(item / ctus), k = (int)(item - (long)f * ctus), lx = k % J.wc, ly = k / J.wc;
  const long cells8 = (long)(J.H >> 3) * (J.W >> 3), cells4 = (long)(J.H >> 2) * (J.W >> 2);
  BinSink s{ J.bins + item * J.cap, 0, J.cap, 0 };
  if (J.sao) {  // encode_sao (encoderstate.c:519-552)
    const int merge = J.sao_merge[item];
    if (lx > 0) s.ctx(KVZ_HIP_CX_SAO_MERGE, merge == 1);
    if (ly > 0 && merge != 1) s.ctx(KVZ_HIP_CX_SAO_MERGE, merge == 2);
    if (!merge) {
      const SaoRec *r = J.sao + item * 3;
      EntropyCtu::sao_color(s, r[0], r[0], 0);
      EntropyCtu::sao_color(s, r[1], r[1], 1);
      EntropyCtu::sao_color(s, r[1], r[2], 2);
    }
  }
  if (J.cu) {  // a B picture: the inter syntax from the CU records
    const EntropyCtuB cb{ J, tb, J.cu + f * cells4, J.ref_cu + f * cells4, J.coeff + item * KVZ_HIP_CTU_COEFFS, J.W >> 2, lx * 64, ly * 64 };
    cb.coding_tree(s);
  } else {
    const EntropyCtu c{ J, tb, J.depth + f * cells8, J.mode + f * cells8, J.part ? J.part + f * cells8 : nullptr, J.mode4 ? J.mode4 + f * cells4 : nullptr,
                        J.coeff + item * KVZ_HIP_CTU_COEFFS, J.W >> 3, J.W >> 2 };
    c.coding_tree(s, lx * 64, ly * 64);
  }
  const bool last_col = lx == J.wc - 1, last_row = ly == J.hc - 1, end_of_picture = last_col && last_row;
  const bool end_of_slice = end_of_picture && !(J.not_last && J.not_last[f]);
  s.trm(end_of_slice);                                                              // end_of_slice_segment_flag (encoderstate.c:699-712)
  if ((end_of_picture || (!J.no_wpp && last_col)) && !end_of_slice) s.trm(1);       // end_of_subset_one_bit: the substream ends, the slice does not (:716-724)
  J.nbins[item] = s.n;
  J.nbits[item] = s.bits;
}

// ... the same list written another way.  One lane per CTU walking its whole syntax leaves a wavefront's 64 lanes in 64 different places of the program (7 of 64 lanes
// active per instruction on average: profiles/r04_l_pmc_leg_entropy.json) -- and nine tenths of the walk is the residual of the transform blocks.  Here the lanes of a
// wavefront agree on WHAT they do next, most urgent first: (1) a coefficient group of the block a lane is in (entropy_tu_cg: the bulk, now run by every lane that has a
// block open), (2) flush queued records and open the next queued block (entropy_tu_begin), (3) the next node of the coding tree, whose CU queues its flags and blocks
// (DeferSink).  Every lane takes one step of the piece it is at per round (waiting for the wavefront to agree on ONE piece per round -- groups first -- left a lane
// that had finished its block idle until the longest block of the wavefront was through: 53.5 against 40.9 ms for 768 1080p pictures); the records of a lane come out
// in the same order either way.
#ifdef KVZ_HOSTSIM
KVZ_DEV bool entropy_any_lane(bool v) { return v; }  // a lane is a loop iteration: it agrees with itself
#else
__device__ __forceinline__ bool entropy_any_lane(bool v) { return __builtin_amdgcn_ballot_w64(v) != 0; }
#endif
struct EntropyBinsLds {  // per workgroup of 64 lanes: the lanes' queues and tree stacks, item-major
  u32 q[DeferSink::QCAP][64];
  u16 stack[16][64];
};
struct LaneStack { u16 *p; int stride; KVZ_DEV u16 &operator[](int i) { return p[i * stride]; } };
KVZ_DEV void entropy_ctu_bins_phased(const EntropyJob &J, const Tables *tb, long item, bool live, u32 *queue, u16 *stack_mem, int stride)
{
  const int ctus = J.wc * J.hc;
  if (!live) item = 0;  // (a lane past the end keeps the wavefront's votes company and writes nothing)
  const int f = (int)(item / ctus), k = (int)(item - (long)f * ctus), lx = k % J.wc, ly = k / J.wc;
  const long cells8 = (long)(J.H >> 3) * (J.W >> 3), cells4 = (long)(J.H >> 2) * (J.W >> 2);
  DeferSink s{ BinSink{ J.bins + item * J.cap, 0, live ? J.cap : 0u, 0 }, queue, stride, 0, 0 };
  if (live && J.sao) {  // encode_sao (encoderstate.c:519-552)
    const int merge = J.sao_merge[item];
    if (lx > 0) s.ctx(KVZ_HIP_CX_SAO_MERGE, merge == 1);
    if (ly > 0 && merge != 1) s.ctx(KVZ_HIP_CX_SAO_MERGE, merge == 2);
    if (!merge) {
      const SaoRec *r = J.sao + item * 3;
      EntropyCtu::sao_color(s, r[0], r[0], 0);
      EntropyCtu::sao_color(s, r[1], r[1], 1);
      EntropyCtu::sao_color(s, r[1], r[2], 2);
    }
  }
  const i16 *ctu = J.coeff + item * KVZ_HIP_CTU_COEFFS;
  const EntropyCtuB cb{ J, tb, J.cu ? J.cu + f * cells4 : nullptr, J.cu ? J.ref_cu + f * cells4 : nullptr, ctu, J.W >> 2, lx * 64, ly * 64 };
  const EntropyCtu ci{ J, tb, J.cu ? nullptr : J.depth + f * cells8, J.cu ? nullptr : J.mode + f * cells8, J.part ? J.part + f * cells8 : nullptr,
                       J.mode4 ? J.mode4 + f * cells4 : nullptr, ctu, J.W >> 3, J.W >> 2 };
  LaneStack stack{ stack_mem, stride };
  int sp = 0;
  if (live) stack[sp++] = 0;
  TuWalk t;
  t.i = -1;
  bool open = false;
  // every lane takes a step of whatever it is at in every round: the wavefront runs the three pieces one after the other, each for the lanes that are there
  for (;;) {
    const bool cg = open, queued = !open && s.queued(), more = !open && !queued && sp > 0;
    if (!entropy_any_lane(cg || queued || more)) break;
    if (entropy_any_lane(cg)) {
      if (cg) {
        entropy_tu_cg(s.out, tb, t);
        if (t.i < 0) { open = false; s.head++; }
      }
    }
    if (entropy_any_lane(queued)) {
      if (queued && s.flush_records()) {
        const u32 d = s.front();
        t.coeff = ctu + (d & 0x1fffu); t.log2_size = 2 + (int)((d >> 13) & 3u); t.type = (d >> 15) & 1u ? 2 : 0; t.scan_mode = (int)((d >> 16) & 3u);
        entropy_tu_begin(s.out, tb, t);
        open = true;
      }
    }
    if (entropy_any_lane(more)) {
      if (more) { if (J.cu) cb.tree_step(s, stack, sp); else ci.tree_step(s, stack, sp, lx * 64, ly * 64); }
    }
  }
  if (!live) return;
  const bool last_col = lx == J.wc - 1, last_row = ly == J.hc - 1, end_of_picture = last_col && last_row;
  const bool end_of_slice = end_of_picture && !(J.not_last && J.not_last[f]);
  s.trm(end_of_slice);                                                              // end_of_slice_segment_flag (encoderstate.c:699-712)
  if ((end_of_picture || (!J.no_wpp && last_col)) && !end_of_slice) s.trm(1);       // end_of_subset_one_bit: the substream ends, the slice does not (:716-724)
  J.nbins[item] = s.out.n;
  J.nbits[item] = s.out.bits;
}

// Four records per load: a lane walks its own list, so what bounds it is the latency of its loads -- 16 bytes at a time, the next four requested before these are coded
struct alignas(16) Rec4 { u32 v[4]; };
struct Rec16 { u32 w[16]; };  // one 64-byte line of a list: consumed under constant indices (indexing the line with a variable would put it in scratch)
KVZ_DEV Rec16 entropy_load16(const Rec4 *b, u32 block)
{
  Rec16 r;
#pragma unroll
  for (int k = 0; k < 4; k++) { const Rec4 q = b[4 * block + k]; r.w[4 * k] = q.v[0]; r.w[4 * k + 1] = q.v[1]; r.w[4 * k + 2] = q.v[2]; r.w[4 * k + 3] = q.v[3]; }
  return r;
}
struct EntropyTabs {       // where the state machine's table lives: an LDS copy on the device, the original on the host
  const u8 *next;          // Tables::ctx_next as [2][128]
};

// stage 2: the contexts every row of picture f starts from (WPP)
KVZ_DEV void entropy_row_contexts(const EntropyJob &J, const EntropyTabs T, int f, u8 *ctx /* KVZ_ENTROPY_CTXS bytes of work memory */)
{
  const int ctus = J.wc * J.hc;
  u8 *out = J.row_ctx + (long)f * J.hc * KVZ_ENTROPY_CTXS;
  for (int i = 0; i < KVZ_ENTROPY_CTXS; i++) { ctx[i] = J.ctx_init[i]; out[i] = ctx[i]; }
  for (int r = 0; r + 1 < J.hc; r++) {
    if (J.wc >= 2) {  // a picture one CTU wide never reaches "lcu->index == 1": its rows keep the slice's initial states
      for (int x = 0; x < 2; x++) {
        const long item = (long)f * ctus + r * J.wc + x;
        const Rec4 *b = (const Rec4 *)(J.bins + item * J.cap);
        const u32 n = J.nbins[item] < J.cap ? J.nbins[item] : J.cap;
        Rec16 cur = entropy_load16(b, 0);  // (capacities are multiples of 16 records: a whole line is always there to read)
        for (u32 i0 = 0; i0 < n; i0 += 16) {
          const Rec16 nxt = i0 + 16 < n ? entropy_load16(b, (i0 >> 4) + 1) : cur;
          const u32 m = n - i0 < 16 ? n - i0 : 16;
#pragma unroll
          for (u32 k = 0; k < 16; k++) {  // unrolled: the line stays in registers under constant indices, and the body is three LDS accesses
            const u32 rec = cur.w[k];
            if (k >= m || (rec >> 30)) continue;
            const int c = (int)(rec & 0xff), bin = (int)((rec >> 8) & 1), st = ctx[c];
            ctx[c] = T.next[(bin != (st & 1) ? 128 : 0) + st];
          }
          cur = nxt;
        }
      }
    } else {
      for (int i = 0; i < KVZ_ENTROPY_CTXS; i++) ctx[i] = J.ctx_init[i];
    }
    for (int i = 0; i < KVZ_ENTROPY_CTXS; i++) out[(r + 1) * KVZ_ENTROPY_CTXS + i] = ctx[i];
  }
}

// Range of the less probable symbol, H.265 table 9-46 (cabac.c:66-82 kvz_g_auc_lpst_table), four bytes per state
#ifdef KVZ_HOSTSIM
static const u32 kLpsPacked[64] = {
#else
__device__ const u32 kLpsPacked[64] = {
#endif
  0xF0D0B080u, 0xE3C5A780u, 0xD8BB9E80u, 0xCDB2967Bu, 0xC3A98E74u, 0xB9A0876Fu, 0xAF988069u, 0xA6907A64u,
  0x9E89745Fu, 0x96826E5Au, 0x8E7B6855u, 0x87756351u, 0x806F5E4Du, 0x7A695949u, 0x74645545u, 0x6E5F5042u,
  0x685A4C3Eu, 0x6356483Bu, 0x5E514538u, 0x594D4135u, 0x55493E33u, 0x50453B30u, 0x4C42382Eu, 0x483F352Bu,
  0x453B3229u, 0x41383027u, 0x3E362D25u, 0x3B332B23u, 0x38302921u, 0x352E2720u, 0x322B251Eu, 0x3029231Du,
  0x2D27211Bu, 0x2B251F1Au, 0x29231E18u, 0x27211C17u, 0x25201B16u, 0x231E1A15u, 0x211D1814u, 0x1F1B1713u,
  0x1E1A1612u, 0x1C191511u, 0x1B171410u, 0x1916130Fu, 0x1815120Eu, 0x1714110Eu, 0x1613100Du, 0x15120F0Cu,
  0x14110E0Cu, 0x13100E0Bu, 0x120F0D0Bu, 0x110F0C0Au, 0x100E0C0Au, 0x0F0D0B09u, 0x0E0C0B09u, 0x0E0C0A08u,
  0x0D0B0908u, 0x0C0B0907u, 0x0C0A0907u, 0x0B0A0807u, 0x0B090806u, 0x0A090706u, 0x09080706u, 0x02020202u };

// ---- stage 3, the wide form: the same code value, moved out 32 bits at a time, every record through ONE branch-free step ----
// What a wavefront of substreams costs per record is the instructions of the step times the paths its lanes take; the byte-wise coder above has three record kinds, two
// renormalisation paths and a byte path with emulation prevention in it, and sixteen lanes are enough to visit all of them at nearly every step.  Here:
//  * a terminating bin is a context-coded bin on a context that never moves: state 63 | MPS 0, whose LPS range is 2 in every range class (cabac.c:193-210 against :104-133:
//    range -= 2, "LPS" = low += range, seven bits of renormalisation = clz(2) - 23); bypass records use the same pseudo-context and ignore what it says.  So every record
//    reads a context, and what differs between the kinds is three selects;
//  * the code value is a number whose digits leave at the top (cabac.c:138-177 moves a byte when 13 bits have gathered and keeps 0xff bytes back for a carry).  WHEN digits
//    leave does not change them: `low` is 64 bits wide here, 32 bits go at a time (a lane may from 32 pending bits on and must at 38; when one lane must, every lane that may
//    does), the carry is the bit above the pending ones, looked at when digits leave, and goes into the unit kept back from the last time -- and on through memory in the one
//    case in 2^32 where that unit is all ones;
//  * emulation prevention (bitstream.c:212-223) is a property of the finished bytes: a pass of its own over the substream (entropy_escape_chunk below);
//  * the context's state for the NEXT record is fetched while this one is coded: per state one 64-bit table entry (the four LPS ranges, both successors, the state itself),
//    read as soon as the successor of the current context is known, and the state byte of the record after that one step earlier.
#define KVZ_ENTROPY_CTX_NEUTRAL KVZ_ENTROPY_CTXS
static_assert(KVZ_ENTROPY_CTX_NEUTRAL == 168, "KVZ_EB_TRM names it");
#define KVZ_ENTROPY_CTX_STRIDE 172  // bytes of context states per lane: 43 dwords, odd, so lanes reading the same context hit different LDS banks
#define KVZ_EB_NOP KVZ_EB_EP(0, 0)   // a run of no bypass bins: low << 0 + range * 0
// entry of state s = sigma << 1 | mps
KVZ_DEV unsigned long long entropy_state_entry(const u8 *next /* [2][128] */, int s)
{
  const int nm = s >= 126 ? s : next[s], nl = s >= 126 ? s : next[128 + s];
  return (unsigned long long)kLpsPacked[s >> 1] | (unsigned long long)nm << 32 | (unsigned long long)nl << 40 | (unsigned long long)s << 48;
}
struct WideLine { u32 w[16]; };
// a lane's cursor over the records of its substream, a 64-byte line at a time; slots past a CTU's last record read as KVZ_EB_NOP
struct WideCursor {
  const u32 *bins, *nbins; u32 cap;
  long ctu, end;   // CTU index (into bins / nbins), one past the last
  u32 n, i0;
  KVZ_DEV void enter() { if (ctu < end) { n = nbins[ctu] < cap ? nbins[ctu] : cap; i0 = 0; } }
  KVZ_DEV void open(const EntropyJob &J, long first, long count) { bins = J.bins; nbins = J.nbins; cap = J.cap; ctu = first; end = first + count; n = 0; i0 = 0; enter(); }
  KVZ_DEV bool fetch(WideLine &l)  // false: past the end (the line is all KVZ_EB_NOP)
  {
    const bool real = ctu < end;
    u32 valid = 0;
    if (real) {
      const Rec4 *b = (const Rec4 *)(bins + ctu * cap) + (i0 >> 2);
#pragma unroll
      for (int k = 0; k < 4; k++) { const Rec4 q = b[k]; l.w[4 * k] = q.v[0]; l.w[4 * k + 1] = q.v[1]; l.w[4 * k + 2] = q.v[2]; l.w[4 * k + 3] = q.v[3]; }
      valid = n > i0 ? n - i0 : 0;
      i0 += 16;
      if (i0 >= n) { ctu++; enter(); }
    }
#pragma unroll
    for (u32 q = 0; q < 16; q++) l.w[q] = q < valid ? l.w[q] : KVZ_EB_NOP;
    return real;
  }
};
#ifdef KVZ_HOSTSIM
#define KVZ_WAVE_ANY(c) (c)
static int entropy_wide_must() { static const int t = getenv("KVZ_HOSTSIM_WIDE_EARLY") ? 32 : 38; return t; }  // test hook: every lane as early as it may / as late as it must
#else
#define KVZ_WAVE_ANY(c) (__builtin_amdgcn_ballot_w64(c) != 0)
__device__ __forceinline__ constexpr int entropy_wide_must() { return 38; }
#endif
template <int W> struct WideCoder {  // W: bits per unit moved out (32; 8 in tests, where a carry into an all-ones unit happens all the time)
  static_assert(W == 32 || W == 8, "");
  unsigned long long low;   // bits [0, 8 + pend), and above them what has been carried out of them since the last unit left (0 or 1; with byte units up to 2)
  u32 range, cache, n;      // cache: the unit moved out last, kept back for a carry; n: units in memory
  int pend, have;           // pend = 24 - bits_left of cabac.c: bits above bit 8 that have not left yet
  u8 *out;                  // 4-byte aligned; null: count only
  KVZ_DEV void start(u8 *o) { low = 0; range = 510; pend = 1; have = 0; cache = 0; n = 0; out = o; }
  KVZ_DEV u32 load_unit(long j) const
  {
    if (W == 8) return out[j];
    const u8 *p = out + 4 * j;
    return (u32)p[0] << 24 | (u32)p[1] << 16 | (u32)p[2] << 8 | (u32)p[3];
  }
  KVZ_DEV void store_unit(long j, u32 v)
  {
    if (!out) return;
    if (W == 8) { out[j] = (u8)v; return; }
#ifdef KVZ_HOSTSIM
    u8 *p = out + 4 * j;
    p[0] = (u8)(v >> 24); p[1] = (u8)(v >> 16); p[2] = (u8)(v >> 8); p[3] = (u8)v;
#else
    reinterpret_cast<u32 *>(out)[j] = __builtin_bswap32(v);
#endif
  }
  KVZ_DEV void push(u32 carry)  // the kept unit, with what was carried into it since it left `low`, goes to memory
  {
    const u32 mask = W == 32 ? 0xffffffffu : 0xffu;
    const unsigned long long sum = (unsigned long long)cache + carry;
    if ((sum >> W) != 0 && out) {  // it overflowed: the carry goes on through the units in memory (it always stops: the code value's top bit never carries out)
      for (long j = (long)n - 1; j >= 0; j--) { const u32 u = (load_unit(j) + 1) & mask; store_unit(j, u); if (u) break; }
    }
    store_unit(n, (u32)sum & mask);
    n++;
  }
  KVZ_DEV void move_unit()
  {
    const int top = 8 + pend;
    const u32 carry = (u32)(low >> top);
    const u32 unit = (u32)(low >> (top - W)) & (W == 32 ? 0xffffffffu : 0xffu);
    low &= (1ull << (top - W)) - 1;
    pend -= W;
    if (have) push(carry);
    cache = unit; have = 1;
  }
  KVZ_DEV void move_units()
  {
    if (W == 32) { if (KVZ_WAVE_ANY(pend >= entropy_wide_must()) && pend >= 32) move_unit(); }
    else while (pend >= W) move_unit();
  }
  // kvz_cabac_finish, the stop bit and the zero bits up to the byte boundary (encoderstate.c:726-732); returns the bytes of the substream before emulation prevention
  KVZ_DEV u32 finish()
  {
    const int top = 8 + pend;
    const u32 carry = (u32)(low >> top);
    low &= (1ull << top) - 1;
    if (have) push(carry);
    unsigned long long w = ((low >> 8) << 1) | 1ull;
    int bits = pend + 1;
    const int pad = (8 - (bits & 7)) & 7;
    w <<= pad; bits += pad;
    u32 bytes = (W / 8) * n;
    for (int sh = bits - 8; sh >= 0; sh -= 8) { if (out) out[bytes] = (u8)(w >> sh); bytes++; }
    return bytes;
  }
};
KVZ_DEV int entropy_rec_ctx(u32 rec) { return (rec & 0xc0000000u) == 0x40000000u ? KVZ_ENTROPY_CTX_NEUTRAL : (int)(rec & 0xff); }  // (a bypass record has bins where the others have their context)
// the substream `item` without emulation prevention, at `out` (room for its upper bound, 4-byte aligned); ctx: KVZ_ENTROPY_CTX_STRIDE bytes of work memory;
// tab: entropy_state_entry of the 128 states.  Returns the bytes written.
template <int W> KVZ_DEV u32 entropy_code_row_wide(const EntropyJob &J, const unsigned long long *tab, long item, u8 *ctx, u8 *out)
{
  const int ctus = J.wc * J.hc;
  const int f = J.no_wpp ? (int)item : (int)(item / J.hc), row = J.no_wpp ? 0 : (int)(item - (long)f * J.hc);
  const long first = (long)f * ctus + (long)row * J.wc, count = J.no_wpp ? ctus : J.wc;
  const u8 *start = J.no_wpp ? J.ctx_init : J.row_ctx + ((long)f * J.hc + row) * KVZ_ENTROPY_CTXS;
  for (int i = 0; i < KVZ_ENTROPY_CTXS; i++) ctx[i] = start[i];
  ctx[KVZ_ENTROPY_CTX_NEUTRAL] = 126;
  WideCoder<W> a;
  a.start(out);
  WideCursor cu;
  cu.open(J, first, count);
  WideLine cur, nxt;
  bool cur_real = cu.fetch(cur), nxt_real = cu.fetch(nxt);
  unsigned long long E = tab[ctx[entropy_rec_ctx(cur.w[0])]];  // the entry of the record about to be coded
  u32 P = ctx[entropy_rec_ctx(cur.w[1])];                       // the state of the next record's context as it was before this record
  while (KVZ_WAVE_ANY(cur_real)) {
#pragma unroll
    for (int q = 0; q < 16; q++) {
      const u32 rec = cur.w[q], rec1 = q < 15 ? cur.w[q + 1] : nxt.w[0], rec2 = q < 14 ? cur.w[q + 2] : nxt.w[q - 14];
      const int c = entropy_rec_ctx(rec), c1 = entropy_rec_ctx(rec1), c2 = entropy_rec_ctx(rec2);
      // masks instead of branches: mb = the record is a run of bypass bins, ml = the bin is the less probable symbol (a terminating bin's value sits where a
      // context-coded bin's does, KVZ_EB_TRM; a bypass record's pseudo-context does not care)
      const u32 mb = (rec & 0xc0000000u) == 0x40000000u ? ~0u : 0u;
      const u32 hi = (u32)(E >> 32);  // successor after an MPS | after an LPS << 8 | the state << 16
      const u32 ml = 0u - (((rec >> 8) ^ (hi >> 16)) & 1u);
      const u32 lps = ((u32)E >> ((a.range >> 3) & 24)) & 0xff;
      // the context moves on, and the next record's entry is asked for before this record's arithmetic
      const u32 s_new = (hi >> (ml & 8)) & 0xff;
      ctx[c] = (u8)s_new;
      const u32 s1 = c1 == c ? s_new : P;
      const unsigned long long E1 = tab[s1];
      P = ctx[c2];
      // the interval (cabac.c:104-133, :231-254): an LPS shifts by clz(lps) - 23 and continues with lps, an MPS by one if the range fell below 256
      const u32 rm = a.range - lps;
      const u32 nbits = (u32)__builtin_clz(lps) - 23u, small = rm < 256 ? 1u : 0u;
      const u32 sh_ctx = (nbits & ml) | (small & ~ml);
      const u32 sh = (((rec >> 16) & 0x3f) & mb) | (sh_ctx & ~mb);
      const u32 add = ((a.range * (rec & 0xffffu)) & mb) | (((rm & ml) << sh_ctx) & ~mb);
      a.range = (a.range & mb) | ((((lps & ml) | (rm & ~ml)) << sh_ctx) & ~mb);
      a.low = (a.low << sh) + add;
      a.pend += (int)sh;
      a.move_units();
      E = E1;
    }
    cur = nxt; cur_real = nxt_real;
    nxt_real = cu.fetch(nxt);
  }
  return a.finish();
}

// Emulation prevention (bitstream.c:212-223: a byte below 4 that follows two zero bytes gets 0x03 in front, and the count of zeros starts again) over the finished bytes
// b[0, n), by position: with z zero bytes directly in front of b[i], the serial rule puts an 0x03 in front of b[i] exactly when b[i] < 4, z >= 2 and z is even (inside a
// run of zeros the count restarts after every insertion, so every second zero from the third on gets one; a byte 1..3 behind the run gets one when the run's length is
// even).  A thread takes the bytes [lo, hi): counts its insertions (dst null) or writes its part at dst, which is where b[lo] lands.
KVZ_DEV u32 entropy_escape_chunk(const u8 *b, u32 lo, u32 hi, u8 *dst)
{
  if (lo >= hi) return 0;
  u32 z = 0;
  for (u32 j = lo; j > 0 && b[j - 1] == 0; j--) z++;
  u32 ins = 0;
  for (u32 i = lo; i < hi; i++) {
    const u32 v = b[i];
    if (v < 4 && z >= 2 && (z & 1) == 0) { if (dst) dst[i - lo + ins] = 3; ins++; }
    if (dst) dst[i - lo + ins] = (u8)v;
    z = v == 0 ? z + 1 : 0;
  }
  return ins;
}
// does the serial rule put an 0x03 in front of b[i]?  (entropy_escape_chunk's rule, asked of one position)
KVZ_DEV bool entropy_escape_at(const u8 *b, u32 i)
{
  if (i < 2 || b[i] >= 4 || b[i - 1] != 0 || b[i - 2] != 0) return false;
  u32 z = 2;
  for (u32 j = i - 2; j > 0 && b[j - 1] == 0; j--) z++;
  return (z & 1) == 0;
}

#ifndef KVZ_HOSTSIM
__global__ void __launch_bounds__(64) dev_entropy_bins_kernel(const EntropyJob J, const Tables *tb, long total)
{
  const long item = (long)blockIdx.x * 64 + threadIdx.x;
  if (item < total) entropy_ctu_bins(J, tb, item);
}
#ifndef KVZ_ENTROPY_BINS_WAVES
#define KVZ_ENTROPY_BINS_WAVES 4  /* 128 VGPRs: four wavefronts per SIMD, which is also what 10 KB of LDS per workgroup allow (88 ms against 102 at three; five and more: no gain) */
#endif
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(KVZ_ENTROPY_BINS_WAVES, KVZ_ENTROPY_BINS_WAVES))) dev_entropy_bins_phased_kernel(const EntropyJob J, const Tables *tb, long total, int columns)
{
  __shared__ EntropyBinsLds L;
  // columns: 0 -- every CTU; 1 -- the first two CTUs of every row (what the row contexts are made from: that part goes first, and stage 2 runs beside the rest);
  // 2 -- the others.  `total`: the lanes of this launch
  long item = (long)blockIdx.x * 64 + threadIdx.x;
  const bool live = item < total;
  if (columns != 0 && live) {
    const int per_row = columns == 1 ? 2 : J.wc - 2;
    const long row = item / per_row;  // picture-major: f * hc + ly
    item = row * J.wc + (columns == 1 ? 0 : 2) + (item - row * per_row);
  }
  entropy_ctu_bins_phased(J, tb, item, live, &L.q[0][threadIdx.x], &L.stack[0][threadIdx.x], 64);
}
// Stage 2 is one long dependent chain per picture (its rows' first two CTUs, row after row): few lanes per workgroup spread the pictures over the chip
template <int LANES> struct EntropyLds {
  u8 ctx[LANES][KVZ_ENTROPY_CTXS];
  u8 next[256];
};
template <int LANES> __global__ void __launch_bounds__(LANES) dev_entropy_row_ctx_kernel(const EntropyJob J, const Tables *tb)
{
  __shared__ EntropyLds<LANES> L;
  for (int i = threadIdx.x; i < 256; i += LANES) L.next[i] = tb->ctx_next[i >> 7][i & 127];
  __syncthreads();
  const int f = blockIdx.x * LANES + threadIdx.x;
  if (f < J.n_frames) entropy_row_contexts(J, EntropyTabs{ L.next }, f, L.ctx[threadIdx.x]);
}
// Stage 3: every substream coded once, at out + offsets[item] (room for its upper bound, 4-byte aligned), without emulation prevention; sizes: the bytes it came to.
// A step costs a wavefront the same whatever the number of its lanes: 64 substreams per wavefront (measured: 48 ms for the 26 112 substreams of 1 536 1080p pictures
// at 64 lanes, 54 at 32, 66 at 16; the byte-at-a-time coder this replaces: 105 ms at its best width, 16)
template <int LANES> struct EntropyWideLds {
  u8 ctx[LANES][KVZ_ENTROPY_CTX_STRIDE];
  unsigned long long tab[128];
};
static_assert(sizeof(EntropyWideLds<64>) == 64 * KVZ_ENTROPY_CTX_STRIDE + 1024 && sizeof(EntropyWideLds<64>) <= 20480, "the launch pads this to a CTU-pass workgroup's LDS (kvz_dev.hpp)");
template <int LANES> __global__ void __launch_bounds__(LANES) dev_entropy_code_wide_kernel(const EntropyJob J, const Tables *tb, long total, u32 *sizes, const unsigned long long *offsets, u8 *out)
{
  __shared__ EntropyWideLds<LANES> L;
  for (int i = threadIdx.x; i < 128; i += LANES) L.tab[i] = entropy_state_entry(&tb->ctx_next[0][0], i);
  __syncthreads();
  const long item = (long)blockIdx.x * LANES + threadIdx.x;
  if (item >= total) return;
  // A chain, not a rate: this wavefront needs a fifth of a SIMD's issue slots, but every one of them late is a step late.  Beside another kernel's wavefronts (the next
  // batch's CTU pass, kvz_hip_batch_entropy_code_then) it goes first
#ifndef KVZ_ENTROPY_NO_SETPRIO  /* (developer switch: tools/chain_stress.py's hunt) */
  __builtin_amdgcn_s_setprio(3);
#endif
  sizes[item] = entropy_code_row_wide<32>(J, L.tab, item, L.ctx[threadIdx.x], out + offsets[item]);
}
// one workgroup per substream: the emulation prevention bytes it needs; sizes: in the bytes the coder wrote, out the substream's final size
__global__ void __launch_bounds__(256) dev_entropy_escape_count_kernel(const u8 *src, const unsigned long long *src_off, u32 *sizes, u32 *ins)
{
  __shared__ u32 total;
  if (threadIdx.x == 0) total = 0;
  __syncthreads();
  const u8 *s = src + src_off[blockIdx.x];
  const u32 n = sizes[blockIdx.x];
  u32 mine = 0;
  for (u32 i = threadIdx.x; i < n; i += 256) mine += entropy_escape_at(s, i) ? 1u : 0u;
  if (mine) atomicAdd(&total, mine);
  __syncthreads();
  if (threadIdx.x == 0) { ins[blockIdx.x] = total; sizes[blockIdx.x] = n + total; }
}
// ... and moved back to back, with them: one workgroup per substream (sizes: final; ins: the emulation prevention bytes among them)
__global__ void __launch_bounds__(256) dev_entropy_compact_kernel(const u8 *src, const unsigned long long *src_off, const u32 *sizes, const u32 *ins, const unsigned long long *dst_off, u8 *dst)
{
  const u8 *s = src + src_off[blockIdx.x];
  u8 *d = dst + dst_off[blockIdx.x];
  const u32 n = sizes[blockIdx.x] - ins[blockIdx.x];
  if (ins[blockIdx.x] == 0) {  // (nearly every substream)
    for (u32 i = threadIdx.x; i < n; i += 256) d[i] = s[i];
    return;
  }
  __shared__ u32 before[256];
  const u32 chunk = (n + 255) / 256, lo = threadIdx.x * chunk < n ? threadIdx.x * chunk : n, hi = lo + chunk < n ? lo + chunk : n;
  before[threadIdx.x] = entropy_escape_chunk(s, lo, hi, nullptr);
  __syncthreads();
  if (threadIdx.x == 0) { u32 acc = 0; for (int t = 0; t < 256; t++) { const u32 v = before[t]; before[t] = acc; acc += v; } }
  __syncthreads();
  entropy_escape_chunk(s, lo, hi, d + lo + before[threadIdx.x]);
}
// Where every substream starts, on the device: the exclusive prefix sum of a per-substream byte count -- the worst-case room of a substream's code (the bound of each
// of its per_stream CTUs' bits, rounded as the host's scratch_bytes loop rounds it: dev_entropy_stream_room_kernel) or its final size.  The host used to compute both lists
// and copy them up, and a small host-to-device copy queues behind whatever the copy engine holds -- the next batch's pictures on their way (kvz_hip_batch_upload_all_async).
__global__ void __launch_bounds__(256) dev_entropy_stream_room_kernel(const u32 *bits, long streams, long per_stream, u32 *room)
{
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= streams) return;
  unsigned long long b = 0;
  for (long k = 0; k < per_stream; k++) b += bits[i * per_stream + k];
  room[i] = (u32)((((b + 7) / 8 + 16) * 3 / 2 + 15) & ~15ull);
}
#define KVZ_ENTROPY_SCAN_THREADS 256  /* four wavefronts: a 1024-thread workgroup needs a whole CU's wave slots at once, and queued beside a persistent pass it stood at the head of its queue until the pass ended */
__global__ void __launch_bounds__(KVZ_ENTROPY_SCAN_THREADS) dev_entropy_offsets_kernel(const u32 *bytes, long streams, unsigned long long *offsets)  // one workgroup
{
  constexpr int T = KVZ_ENTROPY_SCAN_THREADS;
  __shared__ unsigned long long part[T];
  // thread t takes the entries t, t + T, ... block by block: coalesced reads; the scan runs block by block with a running base
  unsigned long long base = 0;
  for (long at = 0; at < streams; at += T) {
    const long i = at + threadIdx.x;
    const unsigned long long v = i < streams ? bytes[i] : 0;
    part[threadIdx.x] = v;
    __syncthreads();
    for (int d = 1; d < T; d <<= 1) {  // inclusive scan of the block
      const unsigned long long w = threadIdx.x >= (unsigned)d ? part[threadIdx.x - d] : 0;
      __syncthreads();
      part[threadIdx.x] += w;
      __syncthreads();
    }
    if (i < streams) offsets[i] = base + part[threadIdx.x] - v;
    base += part[T - 1];
    __syncthreads();
  }
}
#endif

}  // namespace kvz
