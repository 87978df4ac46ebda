// kvz_mfma.hpp -- exact integer matrix products on the matrix cores, shared by the CTU kernel (kvz_ctu.hpp) and the batch
// transform kernels (kvz_dev.hpp).  Device code only.
//
// The accumulator layout of v_mfma (lane = column, registers = 4 consecutive rows per k-step) is the B operand layout of the
// next product and, read as A, the transposed matrix, so a result can feed the next product without leaving registers.
// Exactness: 16-bit operands are split x = 256 (x >> 8) + (x & 255); every operand is then an integer binary16 holds exactly
// (as is every entry of the transform matrices, |v| <= 90), products are exact in binary32 and all partial sums stay below
// 32 * 90 * 255 < 2^24 -- the binary32 accumulators hold exact integers whatever the summation order.
#pragma once
#include <hip/hip_runtime.h>

#include "kvz_ops.hpp"

namespace kvz {

typedef _Float16 dev_half4 __attribute__((ext_vector_type(4)));
typedef float dev_float4 __attribute__((ext_vector_type(4)));
typedef float dev_float16 __attribute__((ext_vector_type(16)));

template <int N> struct DevMma;
template <> struct DevMma<16> {  // v_mfma_f32_16x16x16_f16
  typedef dev_float4 Acc;
  static constexpr int NREG = 4, STEPS = 1;
  static __device__ __forceinline__ int idx(int lane) { return lane & 15; }
  static __device__ __forceinline__ int k0(int lane, int) { return 4 * (lane >> 4); }
  static __device__ __forceinline__ int row(int lane, int r) { return 4 * (lane >> 4) + r; }
  static __device__ __forceinline__ Acc mma(dev_half4 a, dev_half4 b, Acc c) { return __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0); }
};
template <> struct DevMma<32> {  // v_mfma_f32_32x32x8_f16
  typedef dev_float16 Acc;
  static constexpr int NREG = 16, STEPS = 4;
  static __device__ __forceinline__ int idx(int lane) { return lane & 31; }
  static __device__ __forceinline__ int k0(int lane, int step) { return 8 * step + 4 * (lane >> 5); }
  static __device__ __forceinline__ int row(int lane, int r) { return 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3); }
  static __device__ __forceinline__ Acc mma(dev_half4 a, dev_half4 b, Acc c) { return __builtin_amdgcn_mfma_f32_32x32x8f16(a, b, c, 0, 0, 0); }
};

// out[] = (register matrix v, accumulator layout) x (table) when table_is_a == false, (table) x (register matrix) otherwise
template <int N> __device__ __forceinline__ void dev_product(const int *v, const u16 *table, bool table_is_a, int lane, int *out)
{
  typedef DevMma<N> M;
  typename M::Acc lo = { 0 }, hi = { 0 };
  for (int st = 0; st < M::STEPS; st++) {
    const dev_half4 tv = *reinterpret_cast<const dev_half4 *>(table + M::idx(lane) * N + M::k0(lane, st));
    dev_half4 dl, dh;
    for (int i = 0; i < 4; i++) { const int x = v[4 * st + i]; dh[i] = (_Float16)(x >> 8); dl[i] = (_Float16)(x & 255); }
    if (table_is_a) { lo = M::mma(tv, dl, lo); hi = M::mma(tv, dh, hi); }
    else { lo = M::mma(dl, tv, lo); hi = M::mma(dh, tv, hi); }
  }
  for (int r = 0; r < M::NREG; r++) out[r] = (int)hi[r] * 256 + (int)lo[r];
}


// Both passes of a 16- or 32-point DCT / IDCT of one N x N int16 block, chained through registers; `x` and `o` may be the same
// block (every load precedes every store: the stores depend on products that consumed all loads).  T / Tt = the transform
// matrix / its transpose as halves, rows contiguous (Tables::dct_h).
//   forward: D0^T = X T^T, K = T D0^T  (dct-generic.c partial_butterfly_*: the intermediate wraps to int16)
//   inverse: U = X^T T,    O = U^T T   (partial_butterfly_inverse_*: both stages clip to int16)
template <int N> __device__ __forceinline__ void mfma_transform_block(const i16 *x, i16 *o, bool inverse, const u16 *T, const u16 *Tt, int lane)
{
  typedef DevMma<N> M;
  constexpr int L2 = N == 16 ? 4 : 5;
  const int col = M::idx(lane);
  int v[M::NREG], t[M::NREG];
  if (!inverse) {
    for (int st = 0; st < M::STEPS; st++) {  // A operand = rows of X: lane <-> row, four consecutive k per step
      const short4 q = *reinterpret_cast<const short4 *>(x + col * N + M::k0(lane, st));
      v[4 * st] = q.x; v[4 * st + 1] = q.y; v[4 * st + 2] = q.z; v[4 * st + 3] = q.w;
    }
    dev_product<N>(v, T, false, lane, t);
    { const int shift = L2 - 1, add = 1 << (shift - 1); for (int r = 0; r < M::NREG; r++) v[r] = (int)(i16)((t[r] + add) >> shift); }
    dev_product<N>(v, T, true, lane, t);
    { const int shift = L2 + 6, add = 1 << (shift - 1); for (int r = 0; r < M::NREG; r++) o[M::row(lane, r) * N + col] = (i16)((t[r] + add) >> shift); }
  } else {
    for (int st = 0; st < M::STEPS; st++)    // A operand = rows of X^T: lane <-> column of X, four consecutive rows per step
      for (int i = 0; i < 4; i++) v[4 * st + i] = x[(M::k0(lane, st) + i) * N + col];
    dev_product<N>(v, Tt, false, lane, t);
    for (int r = 0; r < M::NREG; r++) v[r] = iclip(-32768, 32767, (t[r] + 64) >> 7);
    dev_product<N>(v, Tt, false, lane, t);   // the accumulator read as A is U^T
    for (int r = 0; r < M::NREG; r++) o[M::row(lane, r) * N + col] = (i16)iclip(-32768, 32767, (t[r] + 2048) >> 12);
  }
}

}  // namespace kvz
