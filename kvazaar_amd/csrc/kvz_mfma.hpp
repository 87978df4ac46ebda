// kvz_mfma.hpp -- exact integer matrix products on the matrix cores, shared by the CTU kernel (kvz_ctu.hpp) and the batch
// transform kernels (kvz_dev.hpp).  Device code only.
//
// int8 MFMA with int32 accumulators (v_mfma_i32_32x32x32_i8 / v_mfma_i32_16x16x32_i8): every entry of the transform matrices fits a
// signed byte (|v| <= 90); a 16-bit operand x is split x = 256 h + l into its signed high byte h and its unsigned low byte l, and
// because the instruction wants signed bytes the low part goes in as l - 128 (= l ^ 0x80) with 128 * (sum of the matrix entries it meets)
// preloaded into the accumulator -- together with the rounding constant of the stage.  Everything is exact integer arithmetic
// (|sums| <= 32 * 90 * 128), so results are bit-identical to dct-generic.c whatever the order of summation.
// Splitting four values into byte planes is four v_perm_b32; there are no conversions on either side of the product.
//
// The accumulator layout (lane = column, registers = rows) is the B operand layout of the next product and, read as A, the transposed
// matrix, so a result can feed the next product without leaving registers: the contraction index k only has to sit in the same
// (lane group, byte) slot on both operands, and which k that is is free.  Slot i of a lane carries k = row(lane, i), the row its
// accumulator register i holds; the table operands are stored in that order (Tables::dct_i8, kvz_tables.hpp).
#pragma once
#include <hip/hip_runtime.h>

#include "kvz_ops.hpp"

namespace kvz {

typedef int dev_int4 __attribute__((ext_vector_type(4)));
typedef int dev_int16 __attribute__((ext_vector_type(16)));

// byte planes of four 16-bit values (the low halves of v0..v3): lo = (l0 l1 l2 l3) ^ 0x80808080, hi = (h0 h1 h2 h3)
__device__ __forceinline__ void dev_byte_planes(int v0, int v1, int v2, int v3, u32 &lo, u32 &hi)
{
  const u32 t01 = __builtin_amdgcn_perm((u32)v1, (u32)v0, 0x05010400u), t23 = __builtin_amdgcn_perm((u32)v3, (u32)v2, 0x05010400u);
  lo = __builtin_amdgcn_perm(t23, t01, 0x05040100u) ^ 0x80808080u;
  hi = __builtin_amdgcn_perm(t23, t01, 0x07060302u);
}

template <int N> struct DevMma;
template <> struct DevMma<16> {  // v_mfma_i32_16x16x32_i8, the upper half of the k slots left at zero
  static constexpr int NREG = 4;
  static __device__ __forceinline__ int idx(int lane) { return lane & 15; }
  static __device__ __forceinline__ int row(int lane, int r) { return 4 * (lane >> 4) + r; }
  static __device__ __forceinline__ int k0(int lane, int) { return 4 * (lane >> 4); }  // first of the 4 consecutive k of step 0
  static constexpr int STEPS = 1;
};
template <> struct DevMma<32> {  // v_mfma_i32_32x32x32_i8
  static constexpr int NREG = 16;
  static __device__ __forceinline__ int idx(int lane) { return lane & 31; }
  static __device__ __forceinline__ int row(int lane, int r) { return 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3); }
  static __device__ __forceinline__ int k0(int lane, int step) { return 8 * step + 4 * (lane >> 5); }  // slots 4 step .. 4 step + 3
  static constexpr int STEPS = 4;
};

// The 16-point product's operands that do not depend on the block -- this lane's slot of the table and the biases of its accumulator registers --, so that a wavefront
// with several blocks loads them once
struct DevTable16 {
  long tv;       // the table's bytes of this lane's k slots (the upper four slots stay zero: the contraction is over 16)
  int c;         // 128 x the sum of the table row this lane's column meets (register matrix x table)
  dev_int4 s4;   // 128 x the sums of the table rows this lane's four accumulator rows are (table x register matrix)
};
__device__ __forceinline__ DevTable16 dev_table16(const int8_t *tab, const i32 *sums, int lane)
{
  typedef DevMma<16> M;
  DevTable16 t;
  t.tv = (long)(unsigned long long)*reinterpret_cast<const u32 *>(tab + M::idx(lane) * 16 + 4 * (lane >> 4));
  t.c = 128 * sums[M::idx(lane)];
  t.s4 = *reinterpret_cast<const dev_int4 *>(sums + M::row(lane, 0));
  for (int i = 0; i < 4; i++) t.s4[i] *= 128;
  return t;
}
// l, h: the byte planes of the lane's four 16-bit values (dev_byte_planes)
__device__ __forceinline__ void dev_product16_planes(u32 l, u32 h, const DevTable16 &t, bool table_is_a, int add, int *out)
{
  const long lo = (long)(unsigned long long)l, hi = (long)(unsigned long long)h;
  dev_int4 acc = { 0, 0, 0, 0 };
  // ONE accumulator: the high plane's product, times 256 plus the bias, is what the low plane's product adds to
  if (table_is_a) {
    acc = __builtin_amdgcn_mfma_i32_16x16x32_i8(t.tv, hi, acc, 0, 0, 0);
    for (int i = 0; i < 4; i++) acc[i] = acc[i] * 256 + (t.s4[i] + add);
    acc = __builtin_amdgcn_mfma_i32_16x16x32_i8(t.tv, lo, acc, 0, 0, 0);
  } else {
    acc = __builtin_amdgcn_mfma_i32_16x16x32_i8(hi, t.tv, acc, 0, 0, 0);
    for (int i = 0; i < 4; i++) acc[i] = acc[i] * 256 + (t.c + add);
    acc = __builtin_amdgcn_mfma_i32_16x16x32_i8(lo, t.tv, acc, 0, 0, 0);
  }
  for (int r = 0; r < 4; r++) out[r] = acc[r];
}
__device__ __forceinline__ void dev_product16(const int *v, const DevTable16 &t, bool table_is_a, int add, int *out)
{
  u32 l, h;
  dev_byte_planes(v[0], v[1], v[2], v[3], l, h);
  dev_product16_planes(l, h, t, table_is_a, add, out);
}

// out[] = (register matrix v, accumulator layout) x (table) when table_is_a == false, (table) x (register matrix) otherwise, + add.
// tab: the table's rows in slot order (16 or 32 signed bytes per row), sums: the row sums of the table.
template <int N> __device__ __forceinline__ void dev_product(const int *v, const int8_t *tab, const i32 *sums, bool table_is_a, int lane, int add, int *out)
{
  typedef DevMma<N> M;
  if constexpr (N == 32) {
    dev_int4 lo, hi;
    for (int q = 0; q < 4; q++) { u32 l, h; dev_byte_planes(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3], l, h); lo[q] = (int)l; hi[q] = (int)h; }
    const dev_int4 tv = *reinterpret_cast<const dev_int4 *>(tab + M::idx(lane) * 32 + (lane >> 5) * 16);
    // ONE accumulator: the high plane's product first, then 256 x it + the bias as the accumulator the low plane's product adds to -- half the registers of two
    // accumulators and a combining pass (the 32-point products were where the CTU kernel's register allocation spilled)
    dev_int16 acc;
    for (int r = 0; r < 16; r++) acc[r] = 0;
    if (table_is_a) {
      acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(tv, hi, acc, 0, 0, 0);
      for (int q = 0; q < 4; q++) {
        const dev_int4 s4 = *reinterpret_cast<const dev_int4 *>(sums + M::row(lane, 4 * q));
        for (int i = 0; i < 4; i++) acc[4 * q + i] = acc[4 * q + i] * 256 + (128 * s4[i] + add);
      }
      acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(tv, lo, acc, 0, 0, 0);
    } else {
      const int c = 128 * sums[M::idx(lane)] + add;
      acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(hi, tv, acc, 0, 0, 0);
      for (int r = 0; r < 16; r++) acc[r] = acc[r] * 256 + c;
      acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(lo, tv, acc, 0, 0, 0);
    }
    for (int r = 0; r < 16; r++) out[r] = acc[r];
  } else {
    const DevTable16 t = dev_table16(tab, sums, lane);
    dev_product16(v, t, table_is_a, add, out);
  }
}

// Both passes of a 16- or 32-point DCT / IDCT of one N x N int16 block, chained through registers; `x` and `o` may be the same
// block (every load precedes every store: the stores depend on products that consumed all loads).
//   forward: D0^T = X T^T, K = T D0^T  (dct-generic.c partial_butterfly_*: the intermediate wraps to int16 -- only its two low bytes are used)
//   inverse: U = X^T T,    O = U^T T   (partial_butterfly_inverse_*: both stages clip to int16)
template <int N> __device__ __forceinline__ void mfma_transform_block(const i16 *x, i16 *o, bool inverse, const Tables *tb, int lane)
{
  typedef DevMma<N> M;
  constexpr int L2 = N == 16 ? 4 : 5;
  const int8_t *T = tb->dct_i8[L2 - 4][0], *Tt = tb->dct_i8[L2 - 4][1];
  const i32 *sT = tb->dct_sum[L2 - 4][0], *sTt = tb->dct_sum[L2 - 4][1];
  const int col = M::idx(lane);
  int v[M::NREG], t[M::NREG];
  if (!inverse) {
    for (int st = 0; st < M::STEPS; st++) {  // A operand = rows of X: lane <-> row, four consecutive k per step
      const short4 q = *reinterpret_cast<const short4 *>(x + col * N + M::k0(lane, st));
      v[4 * st] = q.x; v[4 * st + 1] = q.y; v[4 * st + 2] = q.z; v[4 * st + 3] = q.w;
    }
    dev_product<N>(v, T, sT, false, lane, 1 << (L2 - 2), t);
    for (int r = 0; r < M::NREG; r++) v[r] = t[r] >> (L2 - 1);
    dev_product<N>(v, T, sT, true, lane, 1 << (L2 + 5), t);
    for (int r = 0; r < M::NREG; r++) o[M::row(lane, r) * N + col] = (i16)(t[r] >> (L2 + 6));
  } else {
    for (int st = 0; st < M::STEPS; st++)    // A operand = rows of X^T: lane <-> column of X, four consecutive rows per step
      for (int i = 0; i < 4; i++) v[4 * st + i] = x[(M::k0(lane, st) + i) * N + col];
    dev_product<N>(v, Tt, sTt, false, lane, 64, t);
    for (int r = 0; r < M::NREG; r++) v[r] = iclip(-32768, 32767, t[r] >> 7);
    dev_product<N>(v, Tt, sTt, false, lane, 2048, t);   // the accumulator read as A is U^T
    for (int r = 0; r < M::NREG; r++) o[M::row(lane, r) * N + col] = (i16)iclip(-32768, 32767, t[r] >> 12);
  }
}

// NB 16x16 blocks that follow each other in x / o (LDS), one wavefront: the table operands are loaded once and the blocks' chains are independent, so the products of
// one block fill the latency of the others'.  Same arithmetic as mfma_transform_block<16>.
template <int NB> __device__ __forceinline__ void mfma_transform_blocks16(const i16 *x, i16 *o, bool inverse, const Tables *tb, int lane)
{
  typedef DevMma<16> M;
  const int col = M::idx(lane), k0 = M::k0(lane, 0);
  const DevTable16 t = dev_table16(tb->dct_i8[0][inverse ? 1 : 0], tb->dct_sum[0][inverse ? 1 : 0], lane);
  int v[NB][4], w[NB][4];
  const u16 *xu = reinterpret_cast<const u16 *>(x);
#pragma unroll
  for (int b = 0; b < NB; b++) {  // the byte planes come straight out of the 16-bit values as they lie in memory: no sign extension in between
    u32 l, h;
    if (!inverse) {
      const uint2 q = *reinterpret_cast<const uint2 *>(x + b * 256 + col * 16 + k0);
      l = __builtin_amdgcn_perm(q.y, q.x, 0x06040200u) ^ 0x80808080u;
      h = __builtin_amdgcn_perm(q.y, q.x, 0x07050301u);
    } else dev_byte_planes((int)xu[b * 256 + k0 * 16 + col], (int)xu[b * 256 + (k0 + 1) * 16 + col], (int)xu[b * 256 + (k0 + 2) * 16 + col], (int)xu[b * 256 + (k0 + 3) * 16 + col], l, h);
    dev_product16_planes(l, h, t, false, inverse ? 64 : 4, w[b]);
  }
#pragma unroll
  for (int b = 0; b < NB; b++) for (int r = 0; r < 4; r++) v[b][r] = inverse ? iclip(-32768, 32767, w[b][r] >> 7) : w[b][r] >> 3;
#pragma unroll
  for (int b = 0; b < NB; b++) dev_product16(v[b], t, !inverse, inverse ? 2048 : 512, w[b]);
#pragma unroll
  for (int b = 0; b < NB; b++) for (int r = 0; r < 4; r++) o[b * 256 + M::row(lane, r) * 16 + col] = (i16)(inverse ? iclip(-32768, 32767, w[b][r] >> 12) : w[b][r] >> 10);
}

}  // namespace kvz
