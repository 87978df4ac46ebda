// kvz_dev.hpp -- device-resident batch kernels behind include/kvz_hip_dev.h.  Included by kvz_hip.hip.
//
// These are the HBM-streaming forms of the primitives: every byte of the inputs is read once with 8- or 16-byte loads
// that are contiguous across the lanes of a wavefront, the arithmetic happens in registers (v_sad_u8, packed int16
// Hadamard, MFMA), and one value per block goes back.  Rooflines: 2 n^2 bytes per SAD / SATD block,
// 4 n^2 bytes per transform block (+ 4 n^3 integer multiply-adds for the 16 / 32-point ones on the matrix cores).
#pragma once
#include <hip/hip_runtime.h>

#include <functional>
#include <mutex>
#include <chrono>
#include <vector>

#include "../../include/kvz_hip_dev.h"
#include "kvz_mfma.hpp"
#include "kvz_ops.hpp"
#include "kvz_sao.hpp"
#include "kvz_entropy.hpp"

namespace kvz {

// Sum over aligned groups of G consecutive lanes (G a power of two <= 64); the result is valid in the group's first lane.
__device__ __forceinline__ u32 load_u32_any(const u8 *p) { u32 v; __builtin_memcpy(&v, p, 4); return v; }  // any byte alignment
template <int G> __device__ __forceinline__ u32 group_sum(u32 v)
{
  for (int off = G / 2; off > 0; off >>= 1) v += __shfl_down(v, off, G);
  return v;
}

// ---------------------------------------------------------------------------------------------------------------
// SAD: one lane per 16 bytes of each input (dwordx4 loads, lane-contiguous), v_sad_u8 per dword, then a segmented sum
// over the n^2 / 16 lanes of a block.  64x64 blocks span four wavefronts: one atomic per wavefront into the zeroed output.
template <int N> __global__ void __launch_bounds__(256) dev_sad_kernel(const uint4 *a, const uint4 *b, const long chunks, u32 *out)
{
  constexpr int G = N * N / 16;  // lanes per block
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  u32 v = 0;
  if (i < chunks) {
    const uint4 x = a[i], y = b[i];
    v = __builtin_amdgcn_sad_u8(x.x, y.x, 0);
    v = __builtin_amdgcn_sad_u8(x.y, y.y, v);
    v = __builtin_amdgcn_sad_u8(x.z, y.z, v);
    v = __builtin_amdgcn_sad_u8(x.w, y.w, v);
  }
  if (G <= 64) {
    v = group_sum<(G <= 64 ? G : 64)>(v);
    if ((threadIdx.x & (G - 1)) == 0 && i < chunks) out[i / G] = v;
  } else {
    v = group_sum<64>(v);
    if ((threadIdx.x & 63) == 0 && i < chunks) atomicAdd(&out[i / G], v);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// SATD: one lane per 8x8 tile.  The lane reads its eight rows of both blocks as 8-byte loads (the lanes of a block row
// read consecutive addresses), forms the differences as packed int16 and runs the 8x8 Hadamard in registers: two packed
// stages along the rows, three down the columns, the last row stage folded into the magnitude sum
// (|p + q| + |p - q| = 2 max(|p|, |q|)).  Per tile (sum + 2) >> 2 (picture-generic.c:252-340), then the sum over the
// (n/8)^2 tiles of the block (strategies-picture.h:53-69).
typedef short dev_pk16 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u32 dev_absmax(dev_pk16 a)
{
  const dev_pk16 m = __builtin_elementwise_max(a, -a);
  return (u32)imax((int)m.x, (int)m.y);
}
__device__ __forceinline__ void dev_diff_row(uint2 x, uint2 y, dev_pk16 *d)
{
  const u32 xs[2] = { x.x, x.y }, ys[2] = { y.x, y.y };
  for (int h = 0; h < 2; h++) {
    dev_pk16 lo, hi;
    lo.x = (short)((int)(xs[h] & 0xff) - (int)(ys[h] & 0xff));
    lo.y = (short)((int)((xs[h] >> 8) & 0xff) - (int)((ys[h] >> 8) & 0xff));
    hi.x = (short)((int)((xs[h] >> 16) & 0xff) - (int)((ys[h] >> 16) & 0xff));
    hi.y = (short)((int)(xs[h] >> 24) - (int)(ys[h] >> 24));
    d[2 * h] = lo; d[2 * h + 1] = hi;
  }
}
__device__ __forceinline__ u32 dev_satd8_regs(dev_pk16 d[8][4])
{
  for (int r = 0; r < 8; r++) {
    const dev_pk16 a0 = d[r][0] + d[r][2], a1 = d[r][1] + d[r][3], a2 = d[r][0] - d[r][2], a3 = d[r][1] - d[r][3];
    d[r][0] = a0 + a1; d[r][1] = a0 - a1; d[r][2] = a2 + a3; d[r][3] = a2 - a3;
  }
  u32 sum = 0;
  for (int j = 0; j < 4; j++) {
    const dev_pk16 a0 = d[0][j] + d[4][j], a1 = d[1][j] + d[5][j], a2 = d[2][j] + d[6][j], a3 = d[3][j] + d[7][j];
    const dev_pk16 a4 = d[0][j] - d[4][j], a5 = d[1][j] - d[5][j], a6 = d[2][j] - d[6][j], a7 = d[3][j] - d[7][j];
    const dev_pk16 b0 = a0 + a2, b1 = a1 + a3, b2 = a0 - a2, b3 = a1 - a3, b4 = a4 + a6, b5 = a5 + a7, b6 = a4 - a6, b7 = a5 - a7;
    sum += dev_absmax(b0 + b1) + dev_absmax(b0 - b1) + dev_absmax(b2 + b3) + dev_absmax(b2 - b3);
    sum += dev_absmax(b4 + b5) + dev_absmax(b4 - b5) + dev_absmax(b6 + b7) + dev_absmax(b6 - b7);
  }
  return 2 * sum;
}
template <int N> __global__ void __launch_bounds__(256) dev_satd_kernel(const u8 *a, const u8 *b, const long tiles, u32 *out)
{
  constexpr int TW = N / 8, T = TW * TW;  // tiles per block row / per block
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  u32 v = 0;
  if (t < tiles) {
    const long blk = t / T;
    const int tt = (int)(t % T), ty = tt / TW, tx = tt % TW;
    const long base = blk * (N * N) + (long)(ty * 8) * N + tx * 8;
    dev_pk16 d[8][4];
    for (int r = 0; r < 8; r++) {
      const uint2 x = *reinterpret_cast<const uint2 *>(a + base + r * N), y = *reinterpret_cast<const uint2 *>(b + base + r * N);
      dev_diff_row(x, y, d[r]);
    }
    v = (dev_satd8_regs(d) + 2) >> 2;
  }
  if (T <= 64) {
    v = group_sum<(T <= 64 ? T : 64)>(v);
    if ((threadIdx.x & (T - 1)) == 0 && t < tiles) out[t / T] = v;
  }
}
__global__ void __launch_bounds__(256) dev_satd4_kernel(const u8 *a, const u8 *b, const long blocks, u32 *out)
{
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= blocks) return;
  const uint4 x = reinterpret_cast<const uint4 *>(a)[i], y = reinterpret_cast<const uint4 *>(b)[i];
  u8 pa[16], pb[16];
  const u32 xs[4] = { x.x, x.y, x.z, x.w }, ys[4] = { y.x, y.y, y.z, y.w };
  for (int k = 0; k < 16; k++) { pa[k] = (u8)(xs[k >> 2] >> (8 * (k & 3))); pb[k] = (u8)(ys[k >> 2] >> (8 * (k & 3))); }
  out[i] = satd4(pa, 4, pb, 4);
}

// ---------------------------------------------------------------------------------------------------------------
// 16- and 32-point DCT / IDCT on the matrix cores: one wavefront per block, both passes chained through registers.
// With T the transform matrix (Tables::dct_i8, signed bytes) and X the n x n input:
//   forward: D0^T = X T^T, K = T D0^T   -> out = K       (dct-generic.c partial_butterfly_*, intermediate wraps to int16)
//   inverse: U = X^T T,    O = U^T T    -> out = O       (partial_butterfly_inverse_*, both stages clip to int16)
// The accumulator layout of v_mfma (lane = column, registers = 4 consecutive rows per k-step) is the B operand layout of
// the next product and, read as A, the transposed matrix, so the first result feeds the second product directly.
// Exactness: int8 operands, int32 accumulators (kvz_mfma.hpp): 16-bit values go in as a signed high and a biased low byte plane.
// Memory side: a wavefront moves its block between HBM and LDS with 16-byte accesses (2 KB per 32x32 block = two dwordx4 per lane) and feeds the
// matrix cores from LDS -- the operand layouts want 2-byte column gathers (inverse input) and 2-byte row scatters (every output), which cost
// 16 narrow global accesses per lane when done against HBM directly.
template <int N> __global__ void __launch_bounds__(256) dev_transform_mfma_kernel(const i16 *in, i16 *out, const int count, const int inverse, const Tables *tb)
{
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if constexpr (N == 16) {
    // FOUR blocks per wavefront (2 KB: two dwordx4 per lane each way); nothing is shared between wavefronts, so they synchronise on their own (a wavefront's LDS
    // operations execute in order: a compiler fence, no s_barrier), load the table operands once and run the four blocks' product chains side by side
    __shared__ alignas(16) i16 s_blk[4][4 * 256];
    i16 *sb = s_blk[wave];
    const long blk = ((long)blockIdx.x * 4 + wave) * 4;
    const long left = (long)count - blk;  // wavefront-uniform
    if (left <= 0) return;
    const uint4 *src = reinterpret_cast<const uint4 *>(in + blk * 256);
    uint4 *dst = reinterpret_cast<uint4 *>(out + blk * 256);
#pragma unroll
    for (int k = 0; k < 2; k++) { const int j = k * 64 + lane; if ((j >> 5) < left) reinterpret_cast<uint4 *>(sb)[j] = src[j]; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    mfma_transform_blocks16<4>(sb, sb, inverse != 0, tb, lane);  // (blocks past the end: whatever the LDS holds, never stored)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int k = 0; k < 2; k++) { const int j = k * 64 + lane; if ((j >> 5) < left) dst[j] = reinterpret_cast<const uint4 *>(sb)[j]; }
  } else {
    __shared__ alignas(16) i16 s_blk[4][N * N];
    const long blk = (long)blockIdx.x * 4 + wave;
    const bool have = blk < count;  // wavefront-uniform
    i16 *sb = s_blk[wave];
    if (have) { const uint4 *src = reinterpret_cast<const uint4 *>(in + blk * (N * N)); for (int k = 0; k < 2; k++) reinterpret_cast<uint4 *>(sb)[k * 64 + lane] = src[k * 64 + lane]; }
    __syncthreads();
    if (have) mfma_transform_block<N>(sb, sb, inverse != 0, tb, lane);
    __syncthreads();
    if (have) { uint4 *dst = reinterpret_cast<uint4 *>(out + blk * (N * N)); for (int k = 0; k < 2; k++) dst[k * 64 + lane] = reinterpret_cast<const uint4 *>(sb)[k * 64 + lane]; }
  }
}

// 4- and 8-point transforms (and the 4x4 DST) on the vector ALU, one lane per block ROW: a row of int16 arrives as packed pairs (one 8- or
// 16-byte load, lane-contiguous), an output is NB / 2 v_dot2_i32_i16 against pairs of matrix entries held in scalar registers with the
// rounding constant as the initial accumulator; the transposition between (and after / before) the two passes goes through a padded LDS tile:
// NB 2-byte writes, one 8- or 16-byte read per row.  forward (dct-generic.c:559-579): pass, transpose, pass, transpose; inverse: transpose, pass,
// transpose, pass.  Four rows per lane (loads of all four in flight at once), 1024 rows per workgroup.
template <int NB> __global__ void __launch_bounds__(256) dev_transform_rows_kernel(const i16 *in, i16 *out, const long rows_total, const int inverse, const u32 *pairs /* [2][NB][NB / 2], rows padded to 4 pairs for NB = 4 */)
{
  constexpr int R = NB == 16 ? 2 : 4, PW = NB / 2, KS = NB == 16 ? 8 : 4, WS = NB == 16 ? 128 : 32, BS = NB * NB + (NB == 4 ? 4 : 8), BLOCKS = R * 256 / NB, L2 = NB == 4 ? 2 : (NB == 8 ? 3 : 4);
  typedef short s16x2 __attribute__((ext_vector_type(2)));
  __shared__ alignas(16) i16 s_t[BLOCKS * BS];
  const int tid = threadIdx.x;
  const u32 *cp = pairs + (inverse ? WS : 0);  // uniform: the matrix pairs come through the scalar cache (this table is never rewritten)
  u32 p[R][PW];
  long row[R];
  int base[R];  // element offset of the row's block in the tile
  const int r = tid % NB;  // the row's index in its block (256 is a multiple of NB: the same for every row of the lane)
  auto rd = [&](const i16 *ptr, u32 *x) {  // one row = PW dwords
    if (NB == 4) { const uint2 v = *reinterpret_cast<const uint2 *>(ptr); x[0] = v.x; x[1] = v.y; }
    else for (int h = 0; h < PW / 4; h++) { const uint4 v = reinterpret_cast<const uint4 *>(ptr)[h]; x[4 * h] = v.x; x[4 * h + 1] = v.y; x[4 * h + 2] = v.z; x[4 * h + 3] = v.w; }
  };
  auto wr = [&](i16 *ptr, const u32 *x) {
    if (NB == 4) *reinterpret_cast<uint2 *>(ptr) = make_uint2(x[0], x[1]);
    else for (int h = 0; h < PW / 4; h++) reinterpret_cast<uint4 *>(ptr)[h] = make_uint4(x[4 * h], x[4 * h + 1], x[4 * h + 2], x[4 * h + 3]);
  };
#pragma unroll
  for (int q = 0; q < R; q++) {
    row[q] = ((long)blockIdx.x * R + q) * 256 + tid;
    base[q] = ((q * 256 + tid) / NB) * BS;
    for (int i = 0; i < PW; i++) p[q][i] = 0;
    if (row[q] < rows_total) rd(in + row[q] * NB, p[q]);
  }
  auto pass = [&](const u32 *x, int shift, bool clip, int *y) {
#pragma unroll
    for (int k = 0; k < NB; k++) {
      int acc = 1 << (shift - 1);
#pragma unroll
      for (int i = 0; i < PW; i++) acc = __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, x[i]), __builtin_bit_cast(s16x2, cp[k * KS + i]), acc, false);
      acc >>= shift;
      y[k] = clip ? iclip(-32768, 32767, acc) : acc;
    }
  };
  auto put = [&](int q, const int *y) {  // element e of the lane's row -> [e][row]
#pragma unroll
    for (int e = 0; e < NB; e++) s_t[base[q] + e * NB + r] = (i16)y[e];
  };
  auto get = [&](int q, u32 *x) { rd(&s_t[base[q] + r * NB], x); };  // the lane's row of the tile
  int y[NB];
  if (!inverse) {
#pragma unroll
    for (int q = 0; q < R; q++) { pass(p[q], L2 - 1, false, y); put(q, y); }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < R; q++) get(q, p[q]);
    __syncthreads();
#pragma unroll
    for (int q = 0; q < R; q++) { pass(p[q], L2 + 6, false, y); put(q, y); }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < R; q++) get(q, p[q]);
  } else {
#pragma unroll
    for (int q = 0; q < R; q++) { for (int e = 0; e < NB; e++) y[e] = (int)(i16)(p[q][e >> 1] >> (16 * (e & 1))); put(q, y); }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < R; q++) get(q, p[q]);
    __syncthreads();
#pragma unroll
    for (int q = 0; q < R; q++) { pass(p[q], 7, true, y); put(q, y); }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < R; q++) {
      u32 x[PW];
      get(q, x);
      pass(x, 12, true, y);
      for (int i = 0; i < PW; i++) p[q][i] = __builtin_amdgcn_perm((u32)y[2 * i + 1], (u32)y[2 * i], 0x05040100u);
    }
  }
#pragma unroll
  for (int q = 0; q < R; q++)
    if (row[q] < rows_total) wr(out + row[q] * NB, p[q]);
}

// 4- and 8-point transforms (and the 4x4 DST): 16 / n blocks sit on the diagonal of one 16x16 problem, the matrix is the
// matching block-diagonal one (Tables::bd_i8), everything else as above.  Off-diagonal results are exact zeros and never stored.
template <int NB /* block size: 4 or 8 */> __global__ void __launch_bounds__(256)
dev_transform_small_mfma_kernel(const i16 *in, i16 *out, const int count, const int inverse, const int8_t *T, const int8_t *Tt, const i32 *sT, const i32 *sTt)
{
  typedef DevMma<16> M;
  constexpr int L2 = NB == 4 ? 2 : 3, G = 16 / NB;
  const int lane = threadIdx.x & 63;
  const long first = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * G;  // first block of this wavefront
  if (first >= count) return;  // wavefront-uniform
  const int col = M::idx(lane), g = col / NB, cc = col - g * NB, k0 = M::k0(lane, 0);
  const bool have = first + g < count;          // the lane's own block exists
  const bool diag = k0 / NB == g;               // the lane's four k lie in its block
  const i16 *x = in + (first + g) * (NB * NB);
  i16 *o = out + (first + g) * (NB * NB);
  int v[4], t[4];
  for (int i = 0; i < 4; i++) {
    const int kk = k0 - g * NB + i;
    v[i] = (have && diag) ? (int)(inverse ? x[kk * NB + cc] : x[cc * NB + kk]) : 0;
  }
  if (!inverse) {
    dev_product<16>(v, T, sT, false, lane, 1 << (L2 - 2), t);
    for (int r = 0; r < 4; r++) v[r] = t[r] >> (L2 - 1);
    dev_product<16>(v, T, sT, true, lane, 1 << (L2 + 5), t);
    for (int r = 0; r < 4; r++) v[r] = t[r] >> (L2 + 6);
  } else {
    dev_product<16>(v, Tt, sTt, false, lane, 64, t);
    for (int r = 0; r < 4; r++) v[r] = iclip(-32768, 32767, t[r] >> 7);
    dev_product<16>(v, Tt, sTt, false, lane, 2048, t);
    for (int r = 0; r < 4; r++) v[r] = iclip(-32768, 32767, t[r] >> 12);
  }
  for (int r = 0; r < 4; r++) {
    const int row = M::row(lane, r);
    if (have && row / NB == g) o[(row - g * NB) * NB + cc] = (i16)v[r];
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Angular prediction of a batch of blocks with one mode (intra-generic.c:49-155), row-wise: a lane produces groups of 4 horizontally
// adjacent samples of the *vertical* problem -- for modes >= 18 that is the block itself, for modes < 18 the block with the references
// swapped and the result transposed.  Along a row the displacement (delta_int, delta_fract) is one number, so a group needs 5 consecutive
// bytes of the main reference: two aligned LDS dwords funnel-shifted into place, then the two-tap interpolation on two samples per 32-bit
// multiply (the 16-bit halves cannot carry: (32 - f) a + f b + 16 <= 8176).  Horizontal modes: the four lanes of a quad hold the same four
// columns of four consecutive rows; they exchange their dwords by DPP (quad_perm broadcasts), each picks one byte column (v_perm_b32) and
// holds four horizontally adjacent samples of the transposed block, which go through an LDS tile as dwords to be stored lane-contiguously.
// A workgroup stages the references of its blocks (one contiguous byte range per side, copied as dwords whatever its alignment); modes with
// a negative angle first build each block's extended main reference ext[k + W] = ref_main[k], k = -W .. W, the negative indices projected
// from the side reference (intra-generic.c:82-104); the others read the staged bytes directly.
// Bytes per block: 2 (2 W + 1) read, W^2 written.
constexpr int kAngularGroupsPerLane = 8;  // 2048 groups = 8 KB of output per workgroup: enough bytes in flight per barrier phase to cover the load latency
template <int L2> __global__ void __launch_bounds__(256) dev_angular_kernel(const u8 *above, const u8 *left, const int count, const int mode, u8 *out)
{
  constexpr int W = 1 << L2, NG = kAngularGroupsPerLane, GPB = W * W / 4, BLOCKS = 256 * NG / GPB, RS = 2 * W + 1, ES = (2 * W + 1 + 8 + 3) & ~3;
  constexpr int RAW = (BLOCKS * RS + 3 + 3) / 4 + 2;
  __shared__ u32 s_raw[2][RAW];
  __shared__ alignas(8) u8 s_ext[BLOCKS][ES];
  __shared__ u32 s_tile[BLOCKS * W * W / 4];
  const int disp_tab[9] = { 0, 2, 5, 9, 13, 17, 21, 26, 32 };
  const int inv_tab[9] = { 0, 4096, 1638, 910, 630, 482, 390, 315, 256 };
  const bool vertical = mode >= 18;
  const int mode_disp = vertical ? mode - 26 : 10 - mode, ad = iabs(mode_disp);
  const int disp = mode_disp < 0 ? -disp_tab[ad] : disp_tab[ad], inv = inv_tab[ad];
  const long blk0 = (long)blockIdx.x * BLOCKS;
  const int nblk = (int)(count - blk0 < BLOCKS ? count - blk0 : BLOCKS);
  const long g0 = blk0 * RS;
  const int mis = (int)(g0 & 3), ndw = (mis + nblk * RS + 3) >> 2;
  const u8 *main_g = vertical ? above : left, *side_g = vertical ? left : above;
  const int tid = threadIdx.x;
  for (int i = tid; i < ndw; i += 256) {
    s_raw[0][i] = reinterpret_cast<const u32 *>(main_g + g0 - mis)[i];
    if (disp < 0) s_raw[1][i] = reinterpret_cast<const u32 *>(side_g + g0 - mis)[i];
  }
  __syncthreads();
  if (disp < 0) {
    const int lowest = (W * disp) >> 5, span = W + 1 - lowest;  // indices lowest .. W in use
    const u8 *rm = reinterpret_cast<const u8 *>(s_raw[0]) + mis, *rs = reinterpret_cast<const u8 *>(s_raw[1]) + mis;
    constexpr int LPB = BLOCKS >= 256 ? 1 : 256 / BLOCKS;  // lanes per block (no division by the run-time span)
    for (int bb = tid / LPB; bb < nblk; bb += 256 / LPB)
      for (int j = tid % LPB; j < span; j += LPB) {
        const int k = j + lowest;
        s_ext[bb][k + W] = k >= -1 ? rm[bb * RS + k + 1] : rs[bb * RS + ((128 + (-1 - k) * inv) >> 8)];
      }
    __syncthreads();
  }
  u32 *out32 = reinterpret_cast<u32 *>(out + blk0 * (W * W));
#pragma unroll
  for (int g = 0; g < NG; g++) {
    const int vl = tid + 256 * g, b = vl / GPB, q = vl % GPB;
    const int py = vertical ? q / (W / 4) : q % W, px = 4 * (vertical ? q % (W / 4) : q / W);
    // ref_main[k] = base[bo + k], base 4-byte aligned
    const u8 *base = disp < 0 ? &s_ext[b][0] : reinterpret_cast<const u8 *>(s_raw[0]);
    const int bo = disp < 0 ? W : mis + b * RS + 1;
    const int delta = (py + 1) * disp, di = delta >> 5;
    const u32 f = (u32)(delta & 31), c = 32 - f;
    const int j0 = bo + px + di, o = j0 & 3;
    const u32 *src = reinterpret_cast<const u32 *>(base + (j0 & ~3));
    const unsigned long long w = (((unsigned long long)src[1] << 32) | src[0]) >> (8 * o);
    const u32 A = (u32)w, B = (u32)(w >> 8);  // samples k and k + 1
    const u32 ev = (((A & 0x00ff00ffu) * c + (B & 0x00ff00ffu) * f + 0x00100010u) >> 5) & 0x00ff00ffu;
    const u32 od = ((((A >> 8) & 0x00ff00ffu) * c + ((B >> 8) & 0x00ff00ffu) * f + 0x00100010u) >> 5) & 0x00ff00ffu;
    const u32 res = ev | (od << 8);
    if (vertical) {
      if (b < nblk) out32[vl] = res;
    } else {
      // res = samples (px .. px + 3, py) of the vertical problem = out(x = py, y = px .. px + 3); the quad holds py & ~3 .. + 3
      const int r = (int)res, i = tid & 3;
      const u32 r0 = (u32)__builtin_amdgcn_update_dpp(0, r, 0x00, 0xf, 0xf, false), r1 = (u32)__builtin_amdgcn_update_dpp(0, r, 0x55, 0xf, 0xf, false);
      const u32 r2 = (u32)__builtin_amdgcn_update_dpp(0, r, 0xaa, 0xf, 0xf, false), r3 = (u32)__builtin_amdgcn_update_dpp(0, r, 0xff, 0xf, 0xf, false);
      const u32 sel = 0x0400u + (u32)i * 0x0101u;  // byte i of the low operand, byte i of the high one
      const u32 t01 = __builtin_amdgcn_perm(r1, r0, sel), t23 = __builtin_amdgcn_perm(r3, r2, sel);
      s_tile[(b * (W * W) + (px + i) * W + (py & ~3)) >> 2] = __builtin_amdgcn_perm(t23, t01, 0x05040100u);  // out(y = px + i, x = py & ~3 .. + 3)
    }
  }
  if (!vertical) {
    __syncthreads();
#pragma unroll
    for (int g = 0; g < NG; g++) {
      const int vl = tid + 256 * g;
      if (vl / GPB < nblk) out32[vl] = s_tile[vl];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Integer-pel motion cost surface: SAD of one BW x BW block of the current picture against the reference at every
// displacement of a (2 range + 1)^2 window -- kvz_image_calc_sad (image.c:407) per candidate, i.e. the reference is
// edge-replicated outside the frame (image.c:279-397).  One workgroup per block: the block and the (BW + 2 range)^2 window are
// staged in LDS once (clamped addressing does the replication), then a lane scores FOUR horizontally adjacent candidates at a time with
// v_qsad_pk_u16_u8: one instruction takes 8 window bytes and 4 block bytes and adds the SADs at byte offsets 0, 1, 2, 3 to four packed
// 16-bit sums -- 16 absolute differences per instruction, and the candidate's column needs no shifting into place.  The 16-bit sums are
// flushed into 32-bit ones every 256 / BW rows (256 / BW * BW * 255 < 65 536).  Window re-use between candidates never touches HBM again.
constexpr int kSadSurfaceLanes = 320;  // five wavefronts: the 33 x 9 = 297 groups of a +-16 search fit one round
template <int BW> __global__ void __launch_bounds__(kSadSurfaceLanes)
dev_sad_surface_kernel(const u8 *cur, const u8 *ref, const int W, const int H, const int range, const i16 *blk_xy, u32 *out)
{
  constexpr int MAXR = 32, WD = (BW + 2 * MAXR + 8) / 4;  // dwords per staged window row (+ the bytes a partial last group of four reads)
  constexpr int FL = 256 / BW >= BW ? BW : 256 / BW;       // rows per flush of the 16-bit sums
  constexpr int NT = kSadSurfaceLanes;
  __shared__ u32 s_cur[BW * BW / 4];
  __shared__ alignas(8) u32 s_win[(BW + 2 * MAXR) * WD];
  const int b = blockIdx.x, bx = blk_xy[2 * b], by = blk_xy[2 * b + 1], side = 2 * range + 1, wrows = BW + 2 * range, wcols = BW + 2 * range;
  u8 *win8 = reinterpret_cast<u8 *>(s_win);
  for (int i = threadIdx.x; i < BW * BW / 4; i += NT) s_cur[i] = load_u32_any(cur + (long)(by + i / (BW / 4)) * W + bx + 4 * (i % (BW / 4)));
  const int wdw = (wcols + 3) >> 2;  // dwords of a window row that hold samples
  if (bx - range >= 0 && bx - range + 4 * wdw <= W && by - range >= 0 && by - range + wrows <= H) {  // the window lies inside the picture: dword copies
    for (int i = threadIdx.x; i < wrows * wdw; i += NT) {
      const int r = i / wdw, c = i - r * wdw;
      s_win[r * WD + c] = load_u32_any(ref + (long)(by - range + r) * W + bx - range + 4 * c);
    }
  } else {
    for (int i = threadIdx.x; i < wrows * wcols; i += NT) {
      const int r = i / wcols, c = i % wcols;
      const int y = iclip(0, H - 1, by - range + r), x = iclip(0, W - 1, bx - range + c);
      win8[r * (WD * 4) + c] = ref[(long)y * W + x];
    }
  }
  __syncthreads();
  const int groups = (side + 3) >> 2;  // groups of four candidates along x
  for (int t = threadIdx.x; t < side * groups; t += NT) {
    const int dy = t / groups, d0 = t - dy * groups;  // window-relative displacement: row dy, columns 4 d0 .. 4 d0 + 3
    u32 sad[4] = { 0, 0, 0, 0 };
    for (int r0 = 0; r0 < BW; r0 += FL) {
      unsigned long long acc = 0;
#pragma unroll 4
      for (int r = r0; r < r0 + FL; r++) {
        const u32 *wr = s_win + (dy + r) * WD + d0;
        u32 lo = wr[0];
#pragma unroll
        for (int k = 0; k < BW / 4; k++) {
          const u32 hi = wr[k + 1];
          acc = __builtin_amdgcn_qsad_pk_u16_u8(((unsigned long long)hi << 32) | lo, s_cur[r * (BW / 4) + k], acc);
          lo = hi;
        }
      }
      for (int i = 0; i < 4; i++) sad[i] += (u32)(acc >> (16 * i)) & 0xffffu;
    }
    for (int i = 0; i < 4; i++)
      if (4 * d0 + i < side) out[(long)b * side * side + dy * side + 4 * d0 + i] = sad[i];
  }
}

// ---------------------------------------------------------------------------------------------------------------
// SAO applied to whole pictures (kvz_sao_reconstruct, sao.c:302-361, for every CTU and plane): out = in + offset[category], the
// category from the sample and its two neighbours along the CTU's edge class, or from its band (sao-generic.c:84-124).
// Neighbours always come from `in` (the deblocked picture); samples whose neighbour would lie outside the picture keep their
// value (sao.c:324-349).  One lane per 4 samples of a plane row: dword load of the centre, byte loads of the six outer
// neighbours it does not already hold, dword store; one workgroup per plane row.
// Per (frame, CTU, plane) parameter record in 8 bytes: type | class | band position | offsets[0..4]
// One wavefront per CTU (blockIdx = group of four CTU columns | CTU row | frame).  Sixteen samples per lane, two per 32-bit operation: the bytes of the centre dword and of the two neighbour
// dwords (unaligned loads at the class's displacement) are spread over 16-bit halves (even / odd bytes), sign(c - n) + 1 = clamp(c + 1 - n, 0, 2)
// is three packed instructions per neighbour and pair, and the category -> offset table of the CTU's record is applied to all four samples by
// two v_perm_b32 (byte look-ups); band offsets use the same look-up on clamp(band - position + 1, 0, 5).
__device__ __forceinline__ u32 sao_add_offsets(u32 c4, u32 off4)  // clip(c + (int8)off) on four bytes
{
  const dev_pk16 ce = __builtin_bit_cast(dev_pk16, c4 & 0x00ff00ffu), co = __builtin_bit_cast(dev_pk16, (c4 >> 8) & 0x00ff00ffu);
  const dev_pk16 oe = __builtin_bit_cast(dev_pk16, off4 << 8) >> 8, oo = __builtin_bit_cast(dev_pk16, off4) >> 8;  // sign-extended int8 per half (even: the cross-half bits shifted in are shifted out again)
  const dev_pk16 lo = { 0, 0 }, hi = { 255, 255 };
  const dev_pk16 re = __builtin_elementwise_min(__builtin_elementwise_max(ce + oe, lo), hi), ro = __builtin_elementwise_min(__builtin_elementwise_max(co + oo, lo), hi);
  return __builtin_bit_cast(u32, re) | (__builtin_bit_cast(u32, ro) << 8);
}
__device__ __forceinline__ u32 sao_sign_idx(u32 c4, u32 a4, u32 b4)  // per byte: 2 + sign(c - a) + sign(c - b)
{
  const dev_pk16 one = { 1, 1 }, lo = { 0, 0 }, two = { 2, 2 };
  const dev_pk16 ce = __builtin_bit_cast(dev_pk16, c4 & 0x00ff00ffu) + one, co = __builtin_bit_cast(dev_pk16, (c4 >> 8) & 0x00ff00ffu) + one;
  auto sg = [&](dev_pk16 c1, u32 n) { return __builtin_elementwise_min(__builtin_elementwise_max(c1 - __builtin_bit_cast(dev_pk16, n), lo), two); };
  const dev_pk16 ie = sg(ce, a4 & 0x00ff00ffu) + sg(ce, b4 & 0x00ff00ffu), io = sg(co, (a4 >> 8) & 0x00ff00ffu) + sg(co, (b4 >> 8) & 0x00ff00ffu);
  return __builtin_bit_cast(u32, ie) | (__builtin_bit_cast(u32, io) << 8);
}
// 128-bit helpers: four dwords = sixteen samples
__device__ __forceinline__ uint4 load_u128_any(const u8 *p) { uint4 v; __builtin_memcpy(&v, p, 16); return v; }
__device__ __forceinline__ uint4 shl8_u128(uint4 v)  // byte k <- byte k - 1
{
  return make_uint4(v.x << 8, __builtin_amdgcn_alignbyte(v.y, v.x, 3), __builtin_amdgcn_alignbyte(v.z, v.y, 3), __builtin_amdgcn_alignbyte(v.w, v.z, 3));
}
__device__ __forceinline__ uint4 shr8_u128(uint4 v)  // byte k <- byte k + 1
{
  return make_uint4(__builtin_amdgcn_alignbyte(v.y, v.x, 1), __builtin_amdgcn_alignbyte(v.z, v.y, 1), __builtin_amdgcn_alignbyte(v.w, v.z, 1), v.w >> 8);
}
// TAIL: plane widths that are not multiples of 16 (the last lane of a row then holds 4, 8 or 12 samples, moved as single dwords)
template <bool TAIL> __global__ void __launch_bounds__(256) dev_sao_kernel(const u8 *in, u8 *out, const int W, const int H, const unsigned long long *packed, const kvz_hip_sao_params *luma,
                                                                          const kvz_hip_sao_params *chroma)
{
  // One wavefront per CTU, four horizontally adjacent CTUs per workgroup (their rows share 128-byte lines), sixteen samples per lane and
  // pass: luma in four passes of 16 rows (4 lanes across a row), U and V in one pass each (2 lanes across, 32 rows).  The parameter record
  // of a pass -- type, class, offsets -- is uniform over the wavefront, so no lane ever runs another CTU's branch.  Phase 1 only issues the
  // loads of all six passes (nothing in it consumes a loaded value), phase 2 computes and stores.
  const int lane = threadIdx.x & 63, cxi = blockIdx.x * 4 + (threadIdx.x >> 6), cyi = blockIdx.y, frame = blockIdx.z;
  const int wc = (W + 63) >> 6, hc = (H + 63) >> 6;
  if (cxi >= wc) return;
  const long frame_off = (long)frame * ((long)W * H * 3 / 2);
  // The records: packed by the SAO decision kernels of the batch (kvz_sao.hpp SaoRec), or built here from the caller's parameter structures
  const long ctu = (long)frame * wc * hc + (long)cyi * wc + cxi;
  unsigned long long rec3[3];
  for (int c = 0; c < 3; c++) {
    if (packed) rec3[c] = packed[ctu * 3 + c];
    else {
      const kvz_hip_sao_params *q = c ? &chroma[ctu] : &luma[ctu];
      const int base = c == 2 ? 5 : 0;
      unsigned long long v = (unsigned long long)(q->type & 0xff) | ((unsigned long long)(q->eo_class & 0xff) << 8) | ((unsigned long long)(q->band_position[c == 2 ? 1 : 0] & 0xff) << 16);
      for (int k = 0; k < 5; k++) v |= (unsigned long long)(u8)(int8_t)q->offsets[base + k] << (24 + 8 * k);
      rec3[c] = v;
    }
  }
  uint4 c16[6], a16[6], b16[6];
  long at[6];      // byte index of the pass's first sample, -1: outside the picture
  int geo[6];      // nvalid (dwords inside the picture) | (adj_a + 1) << 4 | (adj_b + 1) << 8 | first-sample-at-x0 << 12 | last-sample-at-row-end << 13 | row-at-picture-edge << 14
#pragma unroll
  for (int p = 0; p < 6; p++) {
    const int color = p < 4 ? 0 : p - 3, sh = color ? 1 : 0, fw = W >> sh, fh = H >> sh, bw = 64 >> sh;
    const long plane = frame_off + (color == 0 ? 0 : (color == 1 ? (long)W * H : (long)W * H * 5 / 4));
    const u32 rlo = (u32)rec3[color];
    const int type = (int)(rlo & 0xff), cls = (int)((rlo >> 8) & 0xff);
    const int x = cxi * bw + 16 * (color ? lane & 1 : lane & 3), y = cyi * bw + (color ? lane >> 1 : (lane >> 2) + 16 * p);
    at[p] = -1; geo[p] = 0;
    c16[p] = a16[p] = b16[p] = make_uint4(0, 0, 0, 0);
    if (x < fw && y < fh) {
      const u8 *row = in + plane + (long)y * fw;
      const int nvalid = (!TAIL || fw - x >= 16) ? 4 : (fw - x) >> 2;
      at[p] = plane + (long)y * fw + x;
      const int dx = cls == 1 ? 0 : (cls == 3 ? 1 : -1);           // a = (dx, -1 or 0), b = (-dx, +1 or 0)  (sao.h:71-76)
      // clamped rows: samples whose neighbour row is outside keep their value
      const int ra = (cls == 0 || y == 0) ? 0 : -fw, rb = (cls == 0 || y + 1 == fh) ? 0 : fw;
      // neighbour samples; at the ends of the row the load is moved inside the row and the bytes shifted into place in phase 2 (the byte
      // that falls off belongs to a sample that keeps its value)
      const int last = x + 4 * nvalid;  // one past the last valid sample of this lane
      const int adj_a = (x + dx < 0) ? 1 : ((last + dx > fw) ? -1 : 0), adj_b = (x - dx < 0) ? 1 : ((last - dx > fw) ? -1 : 0);
      geo[p] = nvalid | ((adj_a + 1) << 4) | ((adj_b + 1) << 8) | ((x == 0) << 12) | ((last == fw) << 13) | ((y == 0 || y + 1 == fh) << 14);
      const u8 *pa = row + ra + x + dx + adj_a, *pb = row + rb + x - dx + adj_b;
      if (!TAIL || nvalid == 4) {
        c16[p] = *reinterpret_cast<const uint4 *>(row + x);
        if (type == 2) { a16[p] = load_u128_any(pa); b16[p] = load_u128_any(pb); }
      } else {
        if (nvalid > 0) { c16[p].x = *reinterpret_cast<const u32 *>(row + x); if (type == 2) { a16[p].x = load_u32_any(pa); b16[p].x = load_u32_any(pb); } }
        if (nvalid > 1) { c16[p].y = *reinterpret_cast<const u32 *>(row + x + 4); if (type == 2) { a16[p].y = load_u32_any(pa + 4); b16[p].y = load_u32_any(pb + 4); } }
        if (nvalid > 2) { c16[p].z = *reinterpret_cast<const u32 *>(row + x + 8); if (type == 2) { a16[p].z = load_u32_any(pa + 8); b16[p].z = load_u32_any(pb + 8); } }
      }
    }
  }
#pragma unroll
  for (int p = 0; p < 6; p++) {
    const int color = p < 4 ? 0 : p - 3;
    const unsigned long long rec = rec3[color];
    const u32 rlo = (u32)rec, rhi = (u32)(rec >> 32);             // bytes: type, class, band position, offsets[0] | offsets[1..4]
    const int type = (int)(rlo & 0xff), cls = (int)((rlo >> 8) & 0xff);
    if (at[p] < 0) continue;
    const int nvalid = geo[p] & 15, adj_a = ((geo[p] >> 4) & 3) - 1, adj_b = ((geo[p] >> 8) & 3) - 1;
    u32 res[4];
    uint4 a = a16[p], b = b16[p];
    u32 keep_first = 0xffffffffu, keep_last = 0xffffffffu, whole = 0xffffffffu;
    if (type == 2) {
      if (adj_a > 0) a = shl8_u128(a); else if (adj_a < 0) a = shr8_u128(a);
      if (adj_b > 0) b = shl8_u128(b); else if (adj_b < 0) b = shr8_u128(b);
      // samples with a neighbour outside the picture keep their value (sao.c:324-349)
      if (cls != 0 && ((geo[p] >> 14) & 1)) whole = 0;
      if (cls != 1) { if ((geo[p] >> 12) & 1) keep_first = 0xffffff00u; if ((geo[p] >> 13) & 1) keep_last = 0x00ffffffu; }
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const u32 c4 = i == 0 ? c16[p].x : (i == 1 ? c16[p].y : (i == 2 ? c16[p].z : c16[p].w));
      const u32 a4 = i == 0 ? a.x : (i == 1 ? a.y : (i == 2 ? a.z : a.w)), b4 = i == 0 ? b.x : (i == 1 ? b.y : (i == 2 ? b.z : b.w));
      res[i] = c4;
      if (type == 1) {
        const u32 bh = (1u - ((rlo >> 16) & 0xff)) & 0xffffu;
        const dev_pk16 bias = __builtin_bit_cast(dev_pk16, bh | (bh << 16));  // + 1 - band position in both halves
        const u32 band = (c4 >> 3) & 0x1f1f1f1fu;
        const dev_pk16 lo = { 0, 0 }, five = { 5, 5 };
        const dev_pk16 te = __builtin_elementwise_min(__builtin_elementwise_max(__builtin_bit_cast(dev_pk16, band & 0x00ff00ffu) + bias, lo), five);
        const dev_pk16 to = __builtin_elementwise_min(__builtin_elementwise_max(__builtin_bit_cast(dev_pk16, (band >> 8) & 0x00ff00ffu) + bias, lo), five);
        const u32 t4 = __builtin_bit_cast(u32, te) | (__builtin_bit_cast(u32, to) << 8);
        const u32 sel = __builtin_amdgcn_perm(0x00000c07u, 0x0605040cu, t4);  // 0 -> zero, 1..4 -> offsets[1..4], 5 -> zero
        res[i] = sao_add_offsets(c4, __builtin_amdgcn_perm(rhi, rlo, sel));
      } else if (type == 2) {
        const u32 idx4 = sao_sign_idx(c4, a4, b4);
        const u32 sel = __builtin_amdgcn_perm(0x00000007u, 0x06030504u, idx4);  // {1,2,0,3,4}[idx] as byte positions of the record
        u32 keep = whole;
        if (i == 0) keep &= keep_first;
        if (i == nvalid - 1) keep &= keep_last;
        res[i] = sao_add_offsets(c4, __builtin_amdgcn_perm(rhi, rlo, sel) & keep);
      }
    }
    if (!TAIL || nvalid == 4) *reinterpret_cast<uint4 *>(out + at[p]) = make_uint4(res[0], res[1], res[2], res[3]);
    else {
      if (nvalid > 0) *reinterpret_cast<u32 *>(out + at[p]) = res[0];
      if (nvalid > 1) *reinterpret_cast<u32 *>(out + at[p] + 4) = res[1];
      if (nvalid > 2) *reinterpret_cast<u32 *>(out + at[p] + 8) = res[2];
    }
  }
}
inline void launch_sao(hipStream_t stream, const u8 *in, u8 *out, int W, int H, int n_frames, const unsigned long long *packed, const kvz_hip_sao_params *luma, const kvz_hip_sao_params *chroma)
{
  const long fb = (long)W * H * 3 / 2, lcus = (long)((W + 63) >> 6) * ((H + 63) >> 6);
  for (int f0 = 0; f0 < n_frames; f0 += 32768) {  // gridDim.z <= 65535
    const int nf = n_frames - f0 < 32768 ? n_frames - f0 : 32768;
    const dim3 grid((unsigned)((((W + 63) >> 6) + 3) >> 2), (unsigned)((H + 63) >> 6), (unsigned)nf);
    const unsigned long long *pk = packed ? packed + 3 * lcus * f0 : nullptr;
    const kvz_hip_sao_params *lu = luma ? luma + lcus * f0 : nullptr, *ch = chroma ? chroma + lcus * f0 : nullptr;
    if ((W >> 1) % 16 == 0) hipLaunchKernelGGL(dev_sao_kernel<false>, grid, dim3(256), 0, stream, in + f0 * fb, out + f0 * fb, W, H, pk, lu, ch);
    else hipLaunchKernelGGL(dev_sao_kernel<true>, grid, dim3(256), 0, stream, in + f0 * fb, out + f0 * fb, W, H, pk, lu, ch);
  }
}

struct DeblockGeom { int W, H, beta, tc, tc_c; long frame_bytes; const u8 *cu_depth; const kvz_hip_cu_dbk *info; int slice_b, tc1; /* inter pictures: per-4x4 records, tc at strength 1 */ };

__device__ __forceinline__ bool deblock_edge_on(const DeblockGeom &g, long frame, int x, int y, bool vertical)  // filter.c:202-216
{
  const int w8 = g.W >> 3, d = g.cu_depth[frame * (long)w8 * (g.H >> 3) + (long)(y >> 3) * w8 + (x >> 3)];
  const int tu_w = 64 >> (d ? d : 1);
  return ((vertical ? x : y) & (tu_w - 1)) == 0;
}
// Inter pictures: is the 8x8 unit at (ux, uy) on a transform or prediction edge (filter.c:202-257), and how strong is the 4-sample part whose first
// sample on the q side is (x, y) (filter.c:405-493)?  One record per 4x4 unit.
__device__ __forceinline__ const kvz_hip_cu_dbk *dbk_unit(const DeblockGeom &g, long frame, int x, int y)
{
  return g.info + frame * ((long)(g.W >> 2) * (g.H >> 2)) + (long)(y >> 2) * (g.W >> 2) + (x >> 2);
}
__device__ __forceinline__ bool dbk_edge_on_inter(const DeblockGeom &g, long frame, int ux, int uy, bool vertical)
{
  const kvz_hip_cu_dbk *u = dbk_unit(g, frame, ux, uy);
  const int pos = vertical ? ux : uy;
  if ((pos & ((64 >> u->tr_depth) - 1)) == 0) return true;  // transform edge
  const int cu_w = 64 >> u->depth, rel = pos & (cu_w - 1);  // prediction edges of the CU's partitioning, in quarters of the CU (cu.c:63-72)
  if (rel == 0) return true;
  const int ps = u->part_size;
  int q = -1;  // the second partition's offset along this axis, if it has one
  if (vertical) q = (ps == 2 || ps == 3) ? 2 : (ps == 6 ? 1 : (ps == 7 ? 3 : -1));
  else q = (ps == 1 || ps == 3) ? 2 : (ps == 4 ? 1 : (ps == 5 ? 3 : -1));
  return q >= 0 && rel == q * cu_w / 4;
}
__device__ __forceinline__ int dbk_strength(const DeblockGeom &g, long frame, int x, int y, bool vertical, bool tu_boundary)
{
  const kvz_hip_cu_dbk *q = dbk_unit(g, frame, x, y), *p = dbk_unit(g, frame, vertical ? x - 1 : x, vertical ? y : y - 1);
  if (q->type == 1 || p->type == 1) return 2;
  if (tu_boundary && (q->cbf_y || p->cbf_y)) return 1;
  auto far = [](int a, int b) { return iabs(a - b) >= 4; };
  if (p->mv_dir != 3 && q->mv_dir != 3) {
    const int lq = q->mv_dir - 1, lp = p->mv_dir - 1;
    if (far(q->mv[lq][0], p->mv[lp][0]) || far(q->mv[lq][1], p->mv[lp][1])) return 1;
    if (q->mv_ref[lq] != p->mv_ref[lp]) return 1;
  }
  if (!g.slice_b) return 0;
  // B slices (filter.c:428-489): undefined vectors count as zero, references compared as pictures
  const int rp0 = (p->mv_dir & 1) ? p->ref_id[0] : -1, rp1 = (p->mv_dir & 2) ? p->ref_id[1] : -1;
  const int rq0 = (q->mv_dir & 1) ? q->ref_id[0] : -1, rq1 = (q->mv_dir & 2) ? q->ref_id[1] : -1;
  const int q0x = (q->mv_dir & 1) ? q->mv[0][0] : 0, q0y = (q->mv_dir & 1) ? q->mv[0][1] : 0, q1x = (q->mv_dir & 2) ? q->mv[1][0] : 0, q1y = (q->mv_dir & 2) ? q->mv[1][1] : 0;
  const int p0x = (p->mv_dir & 1) ? p->mv[0][0] : 0, p0y = (p->mv_dir & 1) ? p->mv[0][1] : 0, p1x = (p->mv_dir & 2) ? p->mv[1][0] : 0, p1y = (p->mv_dir & 2) ? p->mv[1][1] : 0;
  if (!((rp0 == rq0 && rp1 == rq1) || (rp0 == rq1 && rp1 == rq0))) return 1;
  const bool straight = far(q0x, p0x) || far(q0y, p0y) || far(q1x, p1x) || far(q1y, p1y);
  const bool crossed = far(q1x, p0x) || far(q1y, p0y) || far(q0x, p1x) || far(q0y, p1y);
  if (rp0 != rp1) return (rp0 == rq0 ? straight : crossed) ? 1 : 0;
  return (straight && crossed) ? 1 : 0;
}
// Luma filter of one 4-line part (filter.c:386-561): sample k of a line, the edge between k = 3 and k = 4.
// Two lines per operation: P[p][k] = sample k of lines 2 p (low half) and 2 p + 1 (high half) as int16.  Every intermediate fits
// 16 bits (|9 (m4 - m3) - 3 (m5 - m2) + 8| <= 3068, sums of up to 8 samples + 4 <= 2044); per-line conditions become 0 / -1 half masks.
__device__ __forceinline__ dev_pk16 pk_splat(int v) { const dev_pk16 r = { (short)v, (short)v }; return r; }
__device__ __forceinline__ dev_pk16 pk_abs(dev_pk16 a) { return __builtin_elementwise_max(a, -a); }
__device__ __forceinline__ dev_pk16 pk_clip(dev_pk16 lo, dev_pk16 hi, dev_pk16 v) { return __builtin_elementwise_min(__builtin_elementwise_max(v, lo), hi); }
__device__ __forceinline__ dev_pk16 pk_select(dev_pk16 mask, dev_pk16 a, dev_pk16 b)  // mask halves 0 / -1: a where set, else b
{
  const u32 m = __builtin_bit_cast(u32, mask);
  return __builtin_bit_cast(dev_pk16, (__builtin_bit_cast(u32, a) & m) | (__builtin_bit_cast(u32, b) & ~m));
}
__device__ __forceinline__ void deblock_luma_lines_pk(dev_pk16 P[2][8], int beta, int tc)
{
  // decisions on lines 0 and 3 (filter.c:386-561): second differences of every line, then the two that count
  const dev_pk16 dpA = pk_abs(P[0][1] - P[0][2] - P[0][2] + P[0][3]), dqA = pk_abs(P[0][4] - P[0][5] - P[0][5] + P[0][6]);
  const dev_pk16 dpB = pk_abs(P[1][1] - P[1][2] - P[1][2] + P[1][3]), dqB = pk_abs(P[1][4] - P[1][5] - P[1][5] + P[1][6]);
  const int dp0 = dpA.x, dq0 = dqA.x, dp3 = dpB.y, dq3 = dqB.y;
  const int dp = dp0 + dp3, dq = dq0 + dq3;
  if (dp + dq >= beta) return;
  const int p00 = P[0][0].x, p03 = P[0][3].x, p04 = P[0][4].x, p07 = P[0][7].x, p30 = P[1][0].y, p33 = P[1][3].y, p34 = P[1][4].y, p37 = P[1][7].y;
  const bool strong = 2 * (dp0 + dq0) < (beta >> 2) && 2 * (dp3 + dq3) < (beta >> 2) &&
                      iabs(p03 - p04) < ((5 * tc + 1) >> 1) && iabs(p33 - p34) < ((5 * tc + 1) >> 1) &&
                      iabs(p00 - p03) + iabs(p04 - p07) < (beta >> 3) && iabs(p30 - p33) + iabs(p34 - p37) < (beta >> 3);
  const int side = (beta + (beta >> 1)) >> 3;
  const dev_pk16 zero = pk_splat(0), maxv = pk_splat(255);
#pragma unroll
  for (int p = 0; p < 2; p++) {
    const dev_pk16 m0 = P[p][0], m1 = P[p][1], m2 = P[p][2], m3 = P[p][3], m4 = P[p][4], m5 = P[p][5], m6 = P[p][6], m7 = P[p][7];
    if (strong) {
      const dev_pk16 t2 = pk_splat(2 * tc), c2 = pk_splat(2), c4 = pk_splat(4);
      const dev_pk16 s34 = m3 + m4;
      P[p][1] = pk_clip(m1 - t2, m1 + t2, (m0 + m0 + m1 + m1 + m1 + m2 + s34 + c4) >> 3);
      P[p][2] = pk_clip(m2 - t2, m2 + t2, (m1 + m2 + s34 + c2) >> 2);
      P[p][3] = pk_clip(m3 - t2, m3 + t2, (m1 + m2 + m2 + s34 + s34 + m5 + c4) >> 3);
      P[p][4] = pk_clip(m4 - t2, m4 + t2, (m2 + s34 + s34 + m5 + m5 + m6 + c4) >> 3);
      P[p][5] = pk_clip(m5 - t2, m5 + t2, (s34 + m5 + m6 + c2) >> 2);
      P[p][6] = pk_clip(m6 - t2, m6 + t2, (s34 + m5 + m6 + m6 + m6 + m7 + m7 + c4) >> 3);
    } else {
      const dev_pk16 d43 = m4 - m3, d52 = m5 - m2;
      dev_pk16 delta = ((d43 << 3) + d43 - d52 - d52 - d52 + pk_splat(8)) >> 4;
      const dev_pk16 on = (pk_abs(delta) - pk_splat(tc * 10)) >> 15;  // -1 where |delta| < 10 tc
      const dev_pk16 tcv = pk_splat(tc), tc2 = pk_splat(tc >> 1), one = pk_splat(1);
      delta = pk_clip(-tcv, tcv, delta);
      P[p][3] = pk_select(on, pk_clip(zero, maxv, m3 + delta), m3);
      P[p][4] = pk_select(on, pk_clip(zero, maxv, m4 - delta), m4);
      if (dp < side) P[p][2] = pk_select(on, pk_clip(zero, maxv, m2 + pk_clip(-tc2, tc2, ((((m1 + m3 + one) >> 1) - m2 + delta) >> 1))), m2);
      if (dq < side) P[p][5] = pk_select(on, pk_clip(zero, maxv, m5 + pk_clip(-tc2, tc2, ((((m6 + m4 + one) >> 1) - m5 - delta) >> 1))), m5);
    }
  }
}
// Luma.  VERTICAL: part (x = 8 * x8, y = 4 * y4): 4 rows of the 8 samples x - 4 .. x + 3 (one 8-byte access per row, 4-byte
// aligned); !VERTICAL: part (x = 4 * x4, y = 8 * y8): the 8 rows y - 4 .. y + 3 of 4 samples (one dword per row).
template <bool VERTICAL> __global__ void __launch_bounds__(256) dev_deblock_luma_kernel(u8 *frames, const DeblockGeom g, const long parts)
{
  const long p = (long)blockIdx.x * 256 + threadIdx.x;
  if (p >= parts) return;
  const int nx = VERTICAL ? g.W >> 3 : g.W >> 2, ny = VERTICAL ? g.H >> 2 : g.H >> 3;
  const long frame = p / ((long)nx * ny);
  const int r = (int)(p % ((long)nx * ny)), ix = r % nx, iy = r / nx;
  const int x = VERTICAL ? 8 * ix : 4 * ix, y = VERTICAL ? 4 * iy : 8 * iy;
  if ((VERTICAL ? x : y) == 0) return;
  int tc = g.tc;
  if (g.info) {  // inter picture: edge test on the 8x8 unit, strength per 4-sample part
    const int ux = x & ~7, uy = y & ~7;
    if (!dbk_edge_on_inter(g, frame, ux, uy, VERTICAL)) return;
    const kvz_hip_cu_dbk *u = dbk_unit(g, frame, ux, uy);
    const int strength = dbk_strength(g, frame, x, y, VERTICAL, (((VERTICAL ? ux : uy)) & ((64 >> u->tr_depth) - 1)) == 0);
    if (!strength) return;
    if (strength == 1) tc = g.tc1;
  } else if (!deblock_edge_on(g, frame, x, y, VERTICAL)) return;
  u8 *Y = frames + frame * g.frame_bytes;
  dev_pk16 P[2][8];
  if (VERTICAL) {
    u32 lo[4], hi[4];
    for (int i = 0; i < 4; i++) {
      const u32 *row = reinterpret_cast<const u32 *>(Y + (long)(y + i) * g.W + x - 4);
      lo[i] = row[0]; hi[i] = row[1];
    }
    for (int p = 0; p < 2; p++)
      for (int k = 0; k < 4; k++) {  // byte k of line 2 p -> low half, of line 2 p + 1 -> high half
        const u32 sel = 0x0c000c00u | (u32)k | ((u32)(4 + k) << 16);
        P[p][k] = __builtin_bit_cast(dev_pk16, __builtin_amdgcn_perm(lo[2 * p + 1], lo[2 * p], sel));
        P[p][4 + k] = __builtin_bit_cast(dev_pk16, __builtin_amdgcn_perm(hi[2 * p + 1], hi[2 * p], sel));
      }
  } else {
    for (int k = 0; k < 8; k++) {
      const u32 v = *reinterpret_cast<const u32 *>(Y + (long)(y - 4 + k) * g.W + x);
      P[0][k] = __builtin_bit_cast(dev_pk16, __builtin_amdgcn_perm(v, v, 0x0c010c00u));
      P[1][k] = __builtin_bit_cast(dev_pk16, __builtin_amdgcn_perm(v, v, 0x0c030c02u));
    }
  }
  deblock_luma_lines_pk(P, g.beta, tc);
  if (VERTICAL) {
    for (int p = 0; p < 2; p++) {
      // halves (k0 | k1 << 8) and (k2 | k3 << 8) of both lines, then one dword per line
      const u32 a01 = __builtin_bit_cast(u32, P[p][0]) | (__builtin_bit_cast(u32, P[p][1]) << 8), a23 = __builtin_bit_cast(u32, P[p][2]) | (__builtin_bit_cast(u32, P[p][3]) << 8);
      const u32 a45 = __builtin_bit_cast(u32, P[p][4]) | (__builtin_bit_cast(u32, P[p][5]) << 8), a67 = __builtin_bit_cast(u32, P[p][6]) | (__builtin_bit_cast(u32, P[p][7]) << 8);
      u32 *r0 = reinterpret_cast<u32 *>(Y + (long)(y + 2 * p) * g.W + x - 4), *r1 = reinterpret_cast<u32 *>(Y + (long)(y + 2 * p + 1) * g.W + x - 4);
      r0[0] = __builtin_amdgcn_perm(a23, a01, 0x05040100u); r0[1] = __builtin_amdgcn_perm(a67, a45, 0x05040100u);
      r1[0] = __builtin_amdgcn_perm(a23, a01, 0x07060302u); r1[1] = __builtin_amdgcn_perm(a67, a45, 0x07060302u);
    }
  } else {
    for (int k = 1; k < 7; k++)
      *reinterpret_cast<u32 *>(Y + (long)(y - 4 + k) * g.W + x) = __builtin_amdgcn_perm(__builtin_bit_cast(u32, P[1][k]), __builtin_bit_cast(u32, P[0][k]), 0x06040200u);
  }
}
// Chroma (filter.c:567-632, 170-190): edges on the 8x8 chroma grid, 4 samples per part, both planes (part index carries the plane)
template <bool VERTICAL> __global__ void __launch_bounds__(256) dev_deblock_chroma_kernel(u8 *frames, const DeblockGeom g, const long parts)
{
  const long p = (long)blockIdx.x * 256 + threadIdx.x;
  if (p >= parts) return;
  const int cw = g.W >> 1, ch = g.H >> 1;
  const int nx = VERTICAL ? (cw + 7) >> 3 : cw >> 2, ny = VERTICAL ? ch >> 2 : (ch + 7) >> 3;  // same counts as deblock_frames_on
  const long per_frame = 2L * nx * ny, frame = p / per_frame;
  const int r = (int)(p % per_frame), plane = r / (nx * ny), q = r % (nx * ny), ix = q % nx, iy = q / nx;
  const int xc = VERTICAL ? 8 * ix : 4 * ix, yc = VERTICAL ? 4 * iy : 8 * iy;
  if ((VERTICAL ? xc : yc) == 0) return;
  if (g.info) {  // filter.c:567-632: the unit's edge test, then only next to an intra CU (strength 2)
    if (!dbk_edge_on_inter(g, frame, 2 * xc, 2 * yc, VERTICAL)) return;
    const kvz_hip_cu_dbk *q = dbk_unit(g, frame, 2 * xc, 2 * yc), *pp = dbk_unit(g, frame, VERTICAL ? 2 * xc - 2 : 2 * xc, VERTICAL ? 2 * yc : 2 * yc - 2);
    if (q->type != 1 && pp->type != 1) return;
  } else if (!deblock_edge_on(g, frame, 2 * xc, 2 * yc, VERTICAL)) return;
  u8 *P = frames + frame * g.frame_bytes + (long)g.W * g.H + (long)plane * cw * ch;
  const int across = VERTICAL ? 1 : cw, along = VERTICAL ? cw : 1;
  for (int i = 0; i < 4; i++) {
    u8 *s = P + (long)yc * cw + xc + i * along;
    const int m2 = s[-2 * across], m3 = s[-across], m4 = s[0], m5 = s[across];
    const int delta = iclip(-g.tc_c, g.tc_c, (((m4 - m3) * 4) + m2 - m5 + 4) >> 3);
    s[-across] = (u8)iclip(0, 255, m3 + delta);
    s[0] = (u8)iclip(0, 255, m4 - delta);
  }
}

// passes: 1 = the vertical edges, 2 = the horizontal edges, 3 = both (in that order)
inline void deblock_frames_on(hipStream_t stream, u8 *frames, int width, int height, int n_frames, const u8 *cu_depth, int qp, int beta_off, int tc_off, int passes = 3,
                              const kvz_hip_cu_dbk *info = nullptr, int slice_b = 0)
{
  if (n_frames <= 0) return;
  DeblockGeom g;
  g.W = width; g.H = height; g.cu_depth = cu_depth; g.frame_bytes = (long)width * height * 3 / 2;
  g.info = info; g.slice_b = slice_b;
  g.tc1 = deblock_tc(iclip(0, 53, qp + 2 * tc_off));  // filter.c:496-497 with strength 1
  g.beta = deblock_beta(iclip(0, 51, qp + 2 * beta_off));
  g.tc = deblock_tc(iclip(0, 53, qp + 2 + 2 * tc_off));                  // filter.c:496-497 with strength 2
  g.tc_c = deblock_tc(iclip(0, 53, chroma_qp_of(qp) + 2 + 2 * tc_off));  // filter.c:592-595
  const int cw = width >> 1, ch = height >> 1;
  auto grid = [](long n) { return dim3((unsigned)((n + 255) / 256)); };
  const long lv = (long)n_frames * (width >> 3) * (height >> 2), lh = (long)n_frames * (width >> 2) * (height >> 3);
  const long cv = 2L * n_frames * ((cw + 7) >> 3) * (ch >> 2), chh = 2L * n_frames * (cw >> 2) * ((ch + 7) >> 3);
  if (lv && (passes & 1)) hipLaunchKernelGGL(dev_deblock_luma_kernel<true>, grid(lv), dim3(256), 0, stream, frames, g, lv);
  if (cv && (passes & 1)) hipLaunchKernelGGL(dev_deblock_chroma_kernel<true>, grid(cv), dim3(256), 0, stream, frames, g, cv);
  if (lh && (passes & 2)) hipLaunchKernelGGL(dev_deblock_luma_kernel<false>, grid(lh), dim3(256), 0, stream, frames, g, lh);
  if (chh && (passes & 2)) hipLaunchKernelGGL(dev_deblock_chroma_kernel<false>, grid(chh), dim3(256), 0, stream, frames, g, chh);
  KVZ_HIP_CHECK(hipGetLastError());
}

// ---------------------------------------------------------------------------------------------------------------
// The "checksum" picture hash (nal-generic.c:57-82 array_checksum) of a batch: per plane the 32-bit wrap-around sum of sample ^ mask(x, y),
// mask = (x & 0xff) ^ (y & 0xff) ^ (x >> 8) ^ (y >> 8).  One workgroup per (plane, group of rows): a lane takes 16 samples of a row per step -- inside an
// aligned group of 16 the mask is one byte xor the position in the group, so four v_sad_u8 against 0 sum them -- and walks down the group's rows; the
// workgroup's sum goes out with one atomic add.  No divisions by run-time values (the first version spent its time in 64-bit index arithmetic: 160 GB/s).
__global__ void __launch_bounds__(256) dev_checksum_kernel(const u8 *frames, const int W, const int H, const long frame_bytes, const int rows_per_wg, u32 *out)
{
  __shared__ u32 s_sum[4];
  const int fp = blockIdx.y, frame = fp / 3, plane = fp - 3 * frame;
  const int pw = plane ? W >> 1 : W, ph = plane ? H >> 1 : H;
  const u8 *base = frames + frame * frame_bytes + (plane == 0 ? 0 : (plane == 1 ? (long)W * H : (long)W * H * 5 / 4));
  const int y0 = blockIdx.x * rows_per_wg, y1 = y0 + rows_per_wg < ph ? y0 + rows_per_wg : ph;
  u32 v = 0;
  if (y0 < ph) {
    if ((pw & 15) == 0) {
      const int units = pw >> 4;  // 16-sample units per row
      for (int i = threadIdx.x; i < units * (y1 - y0); i += 256) {
        const int ry = i / units, u = i - ry * units, y = y0 + ry, x = 16 * u;
        const uint4 w4 = *reinterpret_cast<const uint4 *>(base + (long)y * pw + x);
        const u32 c = (((x & 0xff) ^ (y & 0xff) ^ (x >> 8) ^ (y >> 8)) & 0xff) * 0x01010101u;
        v = __builtin_amdgcn_sad_u8(w4.x ^ c ^ 0x03020100u, 0u, v);
        v = __builtin_amdgcn_sad_u8(w4.y ^ c ^ 0x07060504u, 0u, v);
        v = __builtin_amdgcn_sad_u8(w4.z ^ c ^ 0x0b0a0908u, 0u, v);
        v = __builtin_amdgcn_sad_u8(w4.w ^ c ^ 0x0f0e0d0cu, 0u, v);
      }
    } else {  // widths that are multiples of 4 only: dword by dword
      const int units = pw >> 2;
      for (int i = threadIdx.x; i < units * (y1 - y0); i += 256) {
        const int ry = i / units, u = i - ry * units, y = y0 + ry, x = 4 * u;
        const u32 w = *reinterpret_cast<const u32 *>(base + (long)y * pw + x);
        const u32 c = (((x & 0xff) ^ (y & 0xff) ^ (x >> 8) ^ (y >> 8)) & 0xff) * 0x01010101u;
        v = __builtin_amdgcn_sad_u8(w ^ c ^ 0x03020100u, 0u, v);
      }
    }
  }
  v = group_sum<64>(v);
  if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0 && y0 < ph) atomicAdd(&out[fp], s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3]);
}
inline void launch_checksums(hipStream_t stream, const u8 *frames, int W, int H, int n_frames, u32 *out)
{
  const int rows = 32;  // 60 KB of a 1080p luma plane per workgroup
  const long fb = (long)W * H * 3 / 2;
  for (int f0 = 0; f0 < n_frames; f0 += 16384) {  // gridDim.y <= 65535
    const int nf = n_frames - f0 < 16384 ? n_frames - f0 : 16384;
    hipLaunchKernelGGL(dev_checksum_kernel, dim3((unsigned)((H + rows - 1) / rows), (unsigned)(3 * nf)), dim3(256), 0, stream, frames + (long)f0 * fb, W, H, fb, rows, out + 3L * f0);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// SAO parameter decision (kvz_sao.hpp): statistics + context-free candidates, one workgroup per (LCU, plane) ...
struct SaoGeom { int W, H, wl, hl; long frame_bytes; };
__global__ void __launch_bounds__(256) dev_sao_stats_kernel(const u8 *src, const u8 *R, const u8 *V, const u8 *D, const SaoGeom g, SaoStats *stats, SaoCand *cand)
{
  // accumulators: {sum, count} of a statistic packed into one 64-bit LDS word (sum in the high half, count in the low half: one ds_add_u64 per
  // sample and statistic; the count never carries into the sum -- at most 4096 samples).  Category 0 of the edge classes is never read by the
  // decision (its offset is 0 by definition, sao.c:417-418), so samples of category 0 -- the majority -- touch the band histogram only.
  // Sixteen private copies of the 52 accumulators, interleaved so that copy c lives in LDS banks 2 c, 2 c + 1: the lanes of a wavefront that hit the same
  // statistic no longer serialise on one address (one atomic per cycle and CU was the whole kernel: 34 ms per 1 536 pictures).
  constexpr int NACC = 4 * 5 + 32, COPIES = 16;
  __shared__ unsigned long long acc[NACC * COPIES];
  __shared__ SaoStats st;
  __shared__ u8 s_rec[64 * 64], s_org[64 * 64];
  const long item = blockIdx.x;  // (frame, lcu, plane)
  const int color = (int)(item % 3);
  const long lcu_all = item / 3;
  const int lcus = g.wl * g.hl, lcu = (int)(lcu_all % lcus);
  const long frame = lcu_all / lcus;
  const int lx = lcu % g.wl, ly = lcu / g.wl, sh = color ? 1 : 0, n = 64 >> sh, fw = g.W >> sh, fh = g.H >> sh;
  const long plane = frame * g.frame_bytes + (color == 0 ? 0 : (color == 1 ? (long)g.W * g.H : (long)g.W * g.H * 5 / 4));
  const int bw = imin(n, fw - lx * n), bh = imin(n, fh - ly * n);  // sao.c:598-605, 645-650
  SaoView view{ R + plane, V + plane, D + plane, fw, n, lx * n, ly * n, lx == g.wl - 1, ly == g.hl - 1, color ? 1 : 3 };
  for (int i = threadIdx.x; i < NACC * COPIES; i += 256) acc[i] = 0;
  const int copy = threadIdx.x & (COPIES - 1);
  for (int p = threadIdx.x; p < bw * bh; p += 256) {
    const int x = p % bw, y = p / bw;
    s_rec[p] = (u8)view.at(x, y);
    s_org[p] = src[plane + (long)(ly * n + y) * fw + lx * n + x];
  }
  __syncthreads();
  for (int p = threadIdx.x; p < bw * bh; p += 256) {
    const int x = p % bw, y = p / bw, c = s_rec[p], diff = (int)s_org[p] - c;
    const unsigned long long add = ((unsigned long long)(long long)diff << 32) + 1ull;  // sum += diff (two's complement in the high half), count += 1
    atomicAdd(&acc[(20 + (c >> 3)) * COPIES + copy], add);
    if (x >= 1 && x < bw - 1 && y >= 1 && y < bh - 1) {  // sao-generic.c:68-69: the block's interior
#pragma unroll
      for (int ec = 0; ec < 4; ec++) {
        int ax, ay, bx, by;
        eo_offsets(ec, ax, ay, bx, by);
        const int cat = eo_cat(s_rec[p + ay * bw + ax], s_rec[p + by * bw + bx], c);
        if (cat) atomicAdd(&acc[(ec * 5 + cat) * COPIES + copy], add);
      }
    }
  }
  __syncthreads();
  if (threadIdx.x < 4 * 5 + 32) {  // unpack: the low half's carries never reach bit 32 (counts <= 4096), the high half is the signed sum
    unsigned long long v = 0;
    for (int k = 0; k < COPIES; k++) v += acc[threadIdx.x * COPIES + k];
    const i32 sum = (i32)(v >> 32), cnt = (i32)(v & 0xffffffffu);
    if (threadIdx.x < 20) { st.edge_sum[threadIdx.x / 5][threadIdx.x % 5] = sum; st.edge_cnt[threadIdx.x / 5][threadIdx.x % 5] = cnt; }
    else { st.band_sum[threadIdx.x - 20] = sum; st.band_cnt[threadIdx.x - 20] = cnt; }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < (int)(sizeof(SaoStats) / sizeof(i32)); i += 256) reinterpret_cast<i32 *>(&stats[item])[i] = reinterpret_cast<const i32 *>(&st)[i];
  if (threadIdx.x == 0) {
    SaoCand c;
    sao_candidates(st, c);
    cand[item] = c;
  }
}
// ... and the chain over the LCUs of a picture: one lane per picture
__global__ void __launch_bounds__(64) dev_sao_chain_kernel(const SaoStats *stats, const SaoCand *cand, const SaoGeom g, const int n_frames, const float *fbits, const Tables *tb,
                                                           const double lambda, const int init_merge, const int init_type, const int no_wpp, SaoRec *recs, u8 *merge)
{
  const int f = blockIdx.x * 64 + threadIdx.x;
  if (f >= n_frames) return;
  const long base = (long)f * g.wl * g.hl;
  sao_chain_picture(fbits, tb->ctx_next[0], tb->ctx_next[1], lambda, (u8)init_merge, (u8)init_type, no_wpp, g.wl, g.hl, stats + base * 3, cand + base * 3, recs + base * 3,
                    merge + base);
}

}  // namespace kvz
#include "kvz_fme.hpp"
#include "kvz_me.hpp"
#include "kvz_inter_kernels.hpp"
#include "kvz_inter_host.hpp"

namespace kvz {
__global__ void dev_cu_dbk_kernel(const kvz_hip_cu_info *cu, int count, kvz_hip_cu_dbk *out)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const kvz_hip_cu_info c = cu[i];
  kvz_hip_cu_dbk o;
  __builtin_memset(&o, 0, sizeof o);
  const unsigned masks[5] = { 0x1f, 0x0f, 0x07, 0x03, 0x1 };
  o.type = c.type; o.depth = c.depth; o.tr_depth = c.tr_depth; o.part_size = 0;
  o.cbf_y = (uint8_t)((c.cbf & masks[c.tr_depth > 4 ? 4 : c.tr_depth]) != 0);
  if (c.type == 2) {
    o.mv_dir = c.mv_dir;
    for (int l = 0; l < 2; l++) { o.mv_ref[l] = (int8_t)((c.mv_dir & (1 << l)) ? c.mv_ref[l] : 0); o.ref_id[l] = 0; o.mv[l][0] = c.mv[l][0]; o.mv[l][1] = c.mv[l][1]; }
  }
  out[i] = o;
}
// the device-side scratch of kvz_hip_dev_inter_ctu_pass, grown on demand and kept between calls
struct InterScratch {
  InterSlab *slabs = nullptr; int n_slabs = 0;
  ICtx *ctx = nullptr; unsigned *done = nullptr; uint32_t *items = nullptr; long n_ctus = 0;
  unsigned *ticket = nullptr;  // [0] ticket, [1] error
  InterModel *model = nullptr;
};
// One set per calling thread and device, like the stream the work is queued on (be() is thread_local): two threads, or two devices of one process, never share
// ticket / done / slab buffers, and a buffer is only ever freed by the thread whose (synchronised) stream used it.
// A worker thread that exits frees its sets (kvz_runtime.hpp ThreadState): registered the first time a thread asks for one.
inline InterScratch &inter_scratch()
{
  static thread_local InterScratch s[64];
  static thread_local bool registered = false;
  if (!registered) {
    registered = true;
    InterScratch *all = s;
    thread_state().cleanups.push_back([all]() {
      for (int d = 0; d < 64; d++) {
        InterScratch &x = all[d];
        if (!x.slabs && !x.ctx && !x.ticket) continue;
        (void)hipSetDevice(d);
        (void)hipFree(x.slabs); (void)hipFree(x.ctx); (void)hipFree(x.done); (void)hipFree(x.items); (void)hipFree(x.ticket); (void)hipFree(x.model);
        x = InterScratch();
      }
    });
  }
  return s[current_device() & 63];
}
}  // namespace kvz
namespace kvz {

// ---------------------------------------------------------------------------------------------------------------
// Picture-hash MD5 (nal.c:88-101 kvz_image_md5 -> nal-generic.c:41-55 per plane): a serial chain of 64-byte blocks per plane, so the batch
// brings the parallelism: one lane per (frame, plane).  Each lane streams its plane with 64-byte reads (one cache line per block).
__global__ void __launch_bounds__(64) dev_md5_kernel(const u8 *frames, const int W, const int H, const long n_planes, u8 *out /* [n_planes][16] */)
{
  const long i = (long)blockIdx.x * 64 + threadIdx.x;
  if (i >= n_planes) return;
  const long frame = i / 3;
  const int plane = (int)(i % 3);
  const long bytes = plane ? (long)(W >> 1) * (H >> 1) : (long)W * H;
  const u8 *p = frames + frame * ((long)W * H * 3 / 2) + (plane == 0 ? 0 : (plane == 1 ? (long)W * H : (long)W * H * 5 / 4));
  u32 st[4] = { 0x67452301u, 0xefcdab89u, 0x98badcfeu, 0x10325476u };
  md5_run(st, p, bytes / 64, (int)(bytes % 64), 1, (unsigned long long)bytes);
  for (int k = 0; k < 16; k++) out[i * 16 + k] = (u8)(st[k >> 2] >> (8 * (k & 3)));
}

// timing events per calling thread and device (an event belongs to the device it was created on)
struct DevTimer { hipEvent_t e0 = nullptr, e1 = nullptr; };
static DevTimer &thread_timer(int which)
{
  static thread_local DevTimer t[2][64];
  static thread_local bool registered = false;
  if (!registered) {
    registered = true;
    DevTimer *all = &t[0][0];
    thread_state().cleanups.push_back([all]() {
      for (int i = 0; i < 128; i++) if (all[i].e0) { (void)hipSetDevice(i & 63); (void)hipEventDestroy(all[i].e0); (void)hipEventDestroy(all[i].e1); all[i] = DevTimer(); }
    });
  }
  return t[which][current_device() & 63];
}
static DevTimer &dev_timer() { return thread_timer(0); }
static DevTimer &inter_timer() { return thread_timer(1); }
static float &inter_kernel_ms() { static thread_local float ms = 0; return ms; }
static int &inter_share() { static thread_local int parts = 1; return parts; }  // kvz_hip_dev_inter_set_share  // the last inter CTU pass's kernel, HIP events on its stream

}  // namespace kvz

// ---- assembling a picture from tile pictures (the receive side of the sharded inter configuration's exchange, kvazaar_amd/sharding.py) ----
// Every tile arrives as its own planar Y|U|V picture in a fixed-size slot; one launch pastes all of them into the full planar frame.  One workgroup per
// (tile, row of any plane): a row is w or w/2 contiguous bytes on both sides, copied sixteen bytes per lane where both ends are 16-byte aligned.
namespace kvz {
struct PasteTable { int n; int x[64], y[64], w[64], h[64], slot[64]; };
__global__ void __launch_bounds__(256) dev_paste_tiles_kernel(u8 *frame, const int W, const int H, const u8 *slots, const long slot_bytes, const PasteTable tb)
{
  const int ti = blockIdx.y;
  const int x = tb.x[ti], y = tb.y[ti], w = tb.w[ti], h = tb.h[ti];
  const u8 *src0 = slots + (long)tb.slot[ti] * slot_bytes;
  for (int row = blockIdx.x; row < 2 * h; row += gridDim.x) {  // h luma rows, then h/2 of U, h/2 of V
    const int plane = row < h ? 0 : (row < h + h / 2 ? 1 : 2), r = plane == 0 ? row : (plane == 1 ? row - h : row - h - h / 2);
    const int pw = plane ? w >> 1 : w, fw = plane ? W >> 1 : W;
    const u8 *src = src0 + (plane == 0 ? 0 : (plane == 1 ? (long)w * h : (long)w * h + (long)(w >> 1) * (h >> 1))) + (long)r * pw;
    u8 *dst = frame + (plane == 0 ? 0 : (plane == 1 ? (long)W * H : (long)W * H + (long)(W >> 1) * (H >> 1))) + (long)((plane ? y >> 1 : y) + r) * fw + (plane ? x >> 1 : x);
    if ((((unsigned long long)src | (unsigned long long)dst | (unsigned)pw) & 15) == 0) {
      for (int i = threadIdx.x * 16; i < pw; i += 256 * 16) *reinterpret_cast<uint4 *>(dst + i) = *reinterpret_cast<const uint4 *>(src + i);
    } else {
      for (int i = threadIdx.x; i < pw; i += 256) dst[i] = src[i];
    }
  }
}
}  // namespace kvz

extern "C" {

void *kvz_hip_dev_alloc(size_t bytes)
{
  kvz::runtime_init(-1);
  void *p = nullptr;
  KVZ_HIP_CHECK(hipMalloc(&p, bytes ? bytes : 1));
  return p;
}
void kvz_hip_dev_free(void *p) { if (p) KVZ_HIP_CHECK(hipFree(p)); }
void kvz_hip_dev_upload(void *d, const void *h, size_t n)
{
  KVZ_HIP_CHECK(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, be().stream));
  KVZ_HIP_CHECK(hipStreamSynchronize(be().stream));
}
void kvz_hip_dev_download(void *h, const void *d, size_t n)
{
  KVZ_HIP_CHECK(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, be().stream));
  KVZ_HIP_CHECK(hipStreamSynchronize(be().stream));
}
void kvz_hip_dev_sync(void) { KVZ_HIP_CHECK(hipStreamSynchronize(be().stream)); }
void kvz_hip_dev_copy(void *d, const void *s, size_t n) { KVZ_HIP_CHECK(hipMemcpyAsync(d, s, n, hipMemcpyDeviceToDevice, be().stream)); }
void kvz_hip_dev_timer_start(void)
{
  kvz::DevTimer &t = kvz::dev_timer();
  if (!t.e0) { KVZ_HIP_CHECK(hipEventCreate(&t.e0)); KVZ_HIP_CHECK(hipEventCreate(&t.e1)); }
  KVZ_HIP_CHECK(hipEventRecord(t.e0, be().stream));
}
float kvz_hip_dev_timer_stop(void)
{
  kvz::DevTimer &t = kvz::dev_timer();
  float ms = 0;
  KVZ_HIP_CHECK(hipEventRecord(t.e1, be().stream));
  KVZ_HIP_CHECK(hipEventSynchronize(t.e1));
  KVZ_HIP_CHECK(hipEventElapsedTime(&ms, t.e0, t.e1));
  return ms;
}

#define KVZ_DEV_LAUNCH(kernel, items, ...)                                                                        \
  do {                                                                                                            \
    const long items_ = (items);                                                                                  \
    if (items_ > 0) {                                                                                             \
      hipLaunchKernelGGL(kernel, dim3((unsigned)((items_ + 255) / 256)), dim3(256), 0, be().stream, __VA_ARGS__); \
      KVZ_HIP_CHECK(hipGetLastError());                                                                           \
    }                                                                                                             \
  } while (0)

int kvz_hip_dev_sad_nxn(int n, const uint8_t *a, const uint8_t *b, int count, uint32_t *out)
{
  const long chunks = (long)count * n * n / 16;
  const uint4 *pa = reinterpret_cast<const uint4 *>(a), *pb = reinterpret_cast<const uint4 *>(b);
  switch (n) {
  case 8: KVZ_DEV_LAUNCH(kvz::dev_sad_kernel<8>, chunks, pa, pb, chunks, out); break;
  case 16: KVZ_DEV_LAUNCH(kvz::dev_sad_kernel<16>, chunks, pa, pb, chunks, out); break;
  case 32: KVZ_DEV_LAUNCH(kvz::dev_sad_kernel<32>, chunks, pa, pb, chunks, out); break;
  case 64:
    KVZ_HIP_CHECK(hipMemsetAsync(out, 0, (size_t)count * sizeof(uint32_t), be().stream));
    KVZ_DEV_LAUNCH(kvz::dev_sad_kernel<64>, chunks, pa, pb, chunks, out);
    break;
  default: fprintf(stderr, "kvz_hip_dev_sad_nxn: unsupported n=%d\n", n); return -1;
  }
  return 0;
}

int kvz_hip_dev_satd_nxn(int n, const uint8_t *a, const uint8_t *b, int count, uint32_t *out)
{
  const long tiles = n >= 8 ? (long)count * (n / 8) * (n / 8) : count;
  switch (n) {
  case 4: KVZ_DEV_LAUNCH(kvz::dev_satd4_kernel, tiles, a, b, tiles, out); break;
  case 8: KVZ_DEV_LAUNCH(kvz::dev_satd_kernel<8>, tiles, a, b, tiles, out); break;
  case 16: KVZ_DEV_LAUNCH(kvz::dev_satd_kernel<16>, tiles, a, b, tiles, out); break;
  case 32: KVZ_DEV_LAUNCH(kvz::dev_satd_kernel<32>, tiles, a, b, tiles, out); break;
  case 64: KVZ_DEV_LAUNCH(kvz::dev_satd_kernel<64>, tiles, a, b, tiles, out); break;
  default: fprintf(stderr, "kvz_hip_dev_satd_nxn: unsupported n=%d\n", n); return -1;
  }
  return 0;
}

int kvz_hip_dev_transform(int kind, const int16_t *in, int16_t *tmp, int16_t *out, int count, int use_matrix_cores)
{
  static const int sizes[5] = { 4, 8, 16, 32, 4 };
  const int inverse = kind >= KVZ_HIP_IDCT_4, idx = inverse ? kind - KVZ_HIP_IDCT_4 : kind, n = sizes[idx];
  if (use_matrix_cores == 2 && n == 16) {  // 16-point blocks, A/B: one lane per row on v_dot2_i32_i16 (4.9 / 4.4 TB/s forward / inverse against 5.1 / 5.0 TB/s on the matrix cores, four blocks per wavefront)
    const long rows = (long)count * 16;
    KVZ_DEV_LAUNCH(kvz::dev_transform_rows_kernel<16>, (rows + 511) / 512 * 256, in, out, rows, inverse, &kvz::device_tables()->pairs16[0][0][0]);
    return 0;
  }
  if (use_matrix_cores && (n == 16 || n == 32) && idx != 4) {
    if (n == 16) KVZ_DEV_LAUNCH(kvz::dev_transform_mfma_kernel<16>, ((long)count + 15) / 16 * 256, in, out, count, inverse, kvz::device_tables());  // four blocks per wavefront
    else KVZ_DEV_LAUNCH(kvz::dev_transform_mfma_kernel<32>, ((long)count + 3) / 4 * 256, in, out, count, inverse, kvz::device_tables());
    return 0;
  }
  if (use_matrix_cores == 1) {  // the small sizes run on the vector ALU (v_dot2_i32_i16); use_matrix_cores == 2 keeps the block-diagonal MFMA form for A/B
    const kvz::Tables *tb = kvz::device_tables();
    const int kind_sp = idx == 4 ? 2 : (n == 8 ? 1 : 0);
    const long rows = (long)count * n, threads = (rows + 1023) / 1024 * 256;
    if (n == 4) KVZ_DEV_LAUNCH(kvz::dev_transform_rows_kernel<4>, threads, in, out, rows, inverse, &tb->small_pairs[kind_sp][0][0][0]);
    else KVZ_DEV_LAUNCH(kvz::dev_transform_rows_kernel<8>, threads, in, out, rows, inverse, &tb->small_pairs[kind_sp][0][0][0]);
    return 0;
  }
  if (use_matrix_cores) {
    const kvz::Tables *tb = kvz::device_tables();
    const int kind_bd = idx == 4 ? 2 : (n == 8 ? 1 : 0), per_wave = 16 / n;
    const long threads = ((long)count + 4 * per_wave - 1) / (4 * per_wave) * 256;
    if (n == 4) KVZ_DEV_LAUNCH(kvz::dev_transform_small_mfma_kernel<4>, threads, in, out, count, inverse, tb->bd_i8[kind_bd][0], tb->bd_i8[kind_bd][1], tb->bd_sum[kind_bd][0], tb->bd_sum[kind_bd][1]);
    else KVZ_DEV_LAUNCH(kvz::dev_transform_small_mfma_kernel<8>, threads, in, out, count, inverse, tb->bd_i8[kind_bd][0], tb->bd_i8[kind_bd][1], tb->bd_sum[kind_bd][0], tb->bd_sum[kind_bd][1]);
    return 0;
  }
  if (!tmp) { fprintf(stderr, "kvz_hip_dev_transform: the scalar path needs tmp\n"); return -1; }
  A::transform_dev(be(), kind, 8, in, tmp, out, count);
  return 0;
}

int kvz_hip_dev_angular_pred(int log2_width, int mode, const uint8_t *ref_above, const uint8_t *ref_left, int count, uint8_t *out)
{
  if (count <= 0) return 0;
  const int blocks_per_wg = 256 * kvz::kAngularGroupsPerLane / ((1 << (2 * log2_width)) / 4);
  const long threads = ((long)count + blocks_per_wg - 1) / blocks_per_wg * 256;
  switch (log2_width) {
  case 2: KVZ_DEV_LAUNCH(kvz::dev_angular_kernel<2>, threads, ref_above, ref_left, count, mode, out); break;
  case 3: KVZ_DEV_LAUNCH(kvz::dev_angular_kernel<3>, threads, ref_above, ref_left, count, mode, out); break;
  case 4: KVZ_DEV_LAUNCH(kvz::dev_angular_kernel<4>, threads, ref_above, ref_left, count, mode, out); break;
  case 5: KVZ_DEV_LAUNCH(kvz::dev_angular_kernel<5>, threads, ref_above, ref_left, count, mode, out); break;
  default: fprintf(stderr, "kvz_hip_dev_angular_pred: unsupported log2_width=%d\n", log2_width); return -1;
  }
  return 0;
}

void kvz_hip_dev_deblock_frames(uint8_t *frames, int width, int height, int n_frames, const uint8_t *cu_depth, int qp, int beta_offset_div2,
                                int tc_offset_div2)
{
  kvz::deblock_frames_on(be().stream, frames, width, height, n_frames, cu_depth, qp, beta_offset_div2, tc_offset_div2);
}

void kvz_hip_dev_deblock_frames_inter(uint8_t *frames, int width, int height, int n_frames, const kvz_hip_cu_dbk *info, int qp, int beta_offset_div2, int tc_offset_div2,
                                      int slice_is_b)
{
  kvz::deblock_frames_on(be().stream, frames, width, height, n_frames, nullptr, qp, beta_offset_div2, tc_offset_div2, 3, info, slice_is_b);
}

// The loop filters between two pictures of a device-resident chain with inter prediction: kvz_hip_batch_loop_filters' pipeline (three pictures R / V / D, the
// statistics kernel, the decision chain, the SAO kernel) with the deblocking edges and strengths taken from kvz_hip_cu_dbk records.
namespace kvz {
struct LoopScratch {
  u8 *ver = nullptr, *dbk = nullptr; size_t pic_bytes = 0;
  SaoStats *stats = nullptr; SaoCand *cand = nullptr; SaoRec *recs = nullptr; u8 *merge = nullptr; size_t lcus = 0;
  float *fbits = nullptr;
};
// per calling thread and device (see inter_scratch): kvz_hip_dev_entropy_code_inter reads the SAO decisions the LAST kvz_hip_dev_loop_filters_inter of the same thread
// left on the same device
inline LoopScratch &loop_scratch()
{
  static thread_local LoopScratch s[64];
  static thread_local bool registered = false;
  if (!registered) {
    registered = true;
    LoopScratch *all = s;
    thread_state().cleanups.push_back([all]() {
      for (int d = 0; d < 64; d++) {
        LoopScratch &x = all[d];
        if (!x.ver && !x.stats && !x.fbits) continue;
        (void)hipSetDevice(d);
        (void)hipFree(x.ver); (void)hipFree(x.dbk); (void)hipFree(x.stats); (void)hipFree(x.cand); (void)hipFree(x.recs); (void)hipFree(x.merge); (void)hipFree(x.fbits);
        x = LoopScratch();
      }
    });
  }
  return s[current_device() & 63];
}
}  // namespace kvz
int kvz_hip_dev_loop_filters_inter(const uint8_t *src, uint8_t *rec, int width, int height, int n_pictures, const kvz_hip_cu_dbk *info, int qp, int slice_is_b, int deblock,
                                   int beta_offset_div2, int tc_offset_div2, int sao, int no_wpp, kvz_hip_sao_params *luma, kvz_hip_sao_params *chroma, uint8_t *merge)
{
  if (n_pictures <= 0) return 0;
  if (!src || !rec || !info || width <= 0 || height <= 0 || (width & 7) || (height & 7) || qp < 0 || qp > 51) { fprintf(stderr, "kvz_hip_dev_loop_filters_inter: bad argument\n"); return -1; }
  hipStream_t st = be().stream;
  if (!sao) {
    if (deblock) kvz::deblock_frames_on(st, rec, width, height, n_pictures, nullptr, qp, beta_offset_div2, tc_offset_div2, 3, info, slice_is_b);
    return 0;
  }
  const int wc = (width + 63) >> 6, hc = (height + 63) >> 6;
  const long frame_px = (long)width * height * 3 / 2;
  const size_t pic_bytes = (size_t)frame_px * n_pictures, lcus = (size_t)wc * hc * n_pictures;
  kvz::LoopScratch &sc = kvz::loop_scratch();
  if (pic_bytes > sc.pic_bytes) {
    if (sc.ver) { KVZ_HIP_CHECK(hipFree(sc.ver)); KVZ_HIP_CHECK(hipFree(sc.dbk)); }
    KVZ_HIP_CHECK(hipMalloc((void **)&sc.ver, pic_bytes));
    KVZ_HIP_CHECK(hipMalloc((void **)&sc.dbk, pic_bytes));
    sc.pic_bytes = pic_bytes;
  }
  if (lcus > sc.lcus) {
    if (sc.stats) { KVZ_HIP_CHECK(hipFree(sc.stats)); KVZ_HIP_CHECK(hipFree(sc.cand)); KVZ_HIP_CHECK(hipFree(sc.recs)); KVZ_HIP_CHECK(hipFree(sc.merge)); }
    KVZ_HIP_CHECK(hipMalloc((void **)&sc.stats, lcus * 3 * sizeof(kvz::SaoStats)));
    KVZ_HIP_CHECK(hipMalloc((void **)&sc.cand, lcus * 3 * sizeof(kvz::SaoCand)));
    KVZ_HIP_CHECK(hipMalloc((void **)&sc.recs, lcus * 3 * sizeof(kvz::SaoRec)));
    KVZ_HIP_CHECK(hipMalloc((void **)&sc.merge, lcus));
    sc.lcus = lcus;
  }
  if (!sc.fbits) {
    float fbits[128];
    for (int i = 0; i < 128; i++) fbits[i] = (float)kvz::kEntropyBits[i] / 32768.0f;
    KVZ_HIP_CHECK(hipMalloc((void **)&sc.fbits, sizeof fbits));
    KVZ_HIP_CHECK(hipMemcpy(sc.fbits, fbits, sizeof fbits, hipMemcpyHostToDevice));
  }
  // R = rec (kept until the SAO kernel has read D), V = after the vertical edges, D = after all edges (sao.c:632-735 reads all three around an LCU's borders)
  KVZ_HIP_CHECK(hipMemcpyAsync(sc.ver, rec, pic_bytes, hipMemcpyDeviceToDevice, st));
  if (deblock) kvz::deblock_frames_on(st, sc.ver, width, height, n_pictures, nullptr, qp, beta_offset_div2, tc_offset_div2, 1, info, slice_is_b);
  KVZ_HIP_CHECK(hipMemcpyAsync(sc.dbk, sc.ver, pic_bytes, hipMemcpyDeviceToDevice, st));
  if (deblock) kvz::deblock_frames_on(st, sc.dbk, width, height, n_pictures, nullptr, qp, beta_offset_div2, tc_offset_div2, 2, info, slice_is_b);
  const kvz::SaoGeom g{ width, height, wc, hc, frame_px };
  const double lambda = 0.57 * pow(2.0, (qp - 12) / 3.0);  // rate_control.c:678-691 at the picture's QP
  // context.c:38-39 INIT_SAO_MERGE_FLAG / INIT_SAO_TYPE_IDX of the slice type (B: 153 / 160, I: 153 / 200)
  const int cx_merge = kvz::ctx_state(qp, 153), cx_type = kvz::ctx_state(qp, slice_is_b ? 160 : 200);
  hipLaunchKernelGGL(kvz::dev_sao_stats_kernel, dim3((unsigned)(lcus * 3)), dim3(256), 0, st, src, rec, sc.ver, sc.dbk, g, sc.stats, sc.cand);
  hipLaunchKernelGGL(kvz::dev_sao_chain_kernel, dim3((unsigned)((n_pictures + 63) / 64)), dim3(64), 0, st, (const kvz::SaoStats *)sc.stats, (const kvz::SaoCand *)sc.cand, g, n_pictures, sc.fbits,
                     kvz::device_tables(), lambda, cx_merge, cx_type, no_wpp, sc.recs, sc.merge);
  kvz::launch_sao(st, sc.dbk, rec, width, height, n_pictures, sc.recs, nullptr, nullptr);
  KVZ_HIP_CHECK(hipGetLastError());
  if (luma || chroma || merge) {  // host copies of the decisions (what the encoder writes as SAO syntax)
    std::vector<kvz::SaoRec> recs(lcus * 3);
    KVZ_HIP_CHECK(hipMemcpyAsync(recs.data(), sc.recs, lcus * 3 * sizeof(kvz::SaoRec), hipMemcpyDeviceToHost, st));
    if (merge) KVZ_HIP_CHECK(hipMemcpyAsync(merge, sc.merge, lcus, hipMemcpyDeviceToHost, st));
    KVZ_HIP_CHECK(hipStreamSynchronize(st));
    for (size_t i = 0; i < lcus; i++) {
      auto unpack = [&](kvz_hip_sao_params *o, int plane, int slot) {
        const kvz::SaoRec r = recs[i * 3 + plane];
        if (slot == 0) { memset(o, 0, sizeof *o); o->bitdepth = 8; o->type = (int)(r & 0xff); o->eo_class = (int)((r >> 8) & 0xff); }
        o->band_position[slot] = (int)((r >> 16) & 0xff);
        for (int k = 0; k < 5; k++) o->offsets[5 * slot + k] = (int)(int8_t)(r >> (24 + 8 * k));
      };
      if (luma) unpack(&luma[i], 0, 0);
      if (chroma) { unpack(&chroma[i], 1, 0); unpack(&chroma[i], 2, 1); }
    }
  }
  return 0;
}

int kvz_hip_dev_sad_surface(const uint8_t *cur, const uint8_t *ref, int width, int height, int bw, int range, const int16_t *blk_xy, int count,
                             uint32_t *out)
{
  if (count <= 0) return 0;
  if (range < 0 || range > 32) { fprintf(stderr, "kvz_hip_dev_sad_surface: range %d not in [0, 32]\n", range); return -1; }
  const dim3 grid((unsigned)count), block(kvz::kSadSurfaceLanes);
  switch (bw) {
  case 8: hipLaunchKernelGGL(kvz::dev_sad_surface_kernel<8>, grid, block, 0, be().stream, cur, ref, width, height, range, blk_xy, out); break;
  case 16: hipLaunchKernelGGL(kvz::dev_sad_surface_kernel<16>, grid, block, 0, be().stream, cur, ref, width, height, range, blk_xy, out); break;
  case 32: hipLaunchKernelGGL(kvz::dev_sad_surface_kernel<32>, grid, block, 0, be().stream, cur, ref, width, height, range, blk_xy, out); break;
  case 64: hipLaunchKernelGGL(kvz::dev_sad_surface_kernel<64>, grid, block, 0, be().stream, cur, ref, width, height, range, blk_xy, out); break;
  default: fprintf(stderr, "kvz_hip_dev_sad_surface: unsupported block width %d\n", bw); return -1;
  }
  KVZ_HIP_CHECK(hipGetLastError());
  return 0;
}

int kvz_hip_dev_fme_costs(const uint8_t *cur, const uint8_t *ref, int width, int height, const kvz_hip_fme_pu *pus, int count, int max_pu_size, int steps, uint32_t *out)
{
  if (count <= 0) return 0;
  const dim3 grid((unsigned)count), block(256);
  const kvz::Tables *tb = kvz::device_tables();
  if (max_pu_size <= 16) hipLaunchKernelGGL(kvz::dev_fme_kernel<16>, grid, block, 0, be().stream, cur, ref, width, height, pus, steps, tb, out);
  else if (max_pu_size <= 32) hipLaunchKernelGGL(kvz::dev_fme_kernel<32>, grid, block, 0, be().stream, cur, ref, width, height, pus, steps, tb, out);
  else if (max_pu_size <= 64) hipLaunchKernelGGL(kvz::dev_fme_kernel<64>, grid, block, 0, be().stream, cur, ref, width, height, pus, steps, tb, out);
  else { fprintf(stderr, "kvz_hip_dev_fme_costs: PUs larger than 64 samples do not exist\n"); return -1; }
  KVZ_HIP_CHECK(hipGetLastError());
  return 0;
}

void kvz_hip_dev_cu_dbk_from_info(const kvz_hip_cu_info *cu, int count, kvz_hip_cu_dbk *out)
{
  if (count <= 0) return;
  hipLaunchKernelGGL(kvz::dev_cu_dbk_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, be().stream, cu, count, out);
  KVZ_HIP_CHECK(hipGetLastError());
}

int kvz_hip_dev_inter_ctu_pass(const uint8_t *src, const uint8_t *ref, const kvz_hip_cu_info *ref_cu, uint8_t *rec, kvz_hip_cu_info *cu, int16_t *coeff, int width,
                               int height, int n_pictures, const kvz_hip_inter_params *p)
{
  return kvz_hip_dev_inter_ctu_pass_tiles(src, ref, ref_cu, rec, cu, coeff, width, height, n_pictures, p, nullptr, 0);
}
int kvz_hip_dev_inter_ctu_pass_tiles(const uint8_t *src, const uint8_t *ref, const kvz_hip_cu_info *ref_cu, uint8_t *rec, kvz_hip_cu_info *cu, int16_t *coeff, int width,
                                     int height, int n_pictures, const kvz_hip_inter_params *p, const int32_t *tile_xy, int n_references)
{
  if (n_pictures <= 0) return 0;
  if (!p || p->struct_size != sizeof(kvz_hip_inter_params)) {
    fprintf(stderr, "kvz_hip_dev_inter_ctu_pass: kvz_hip_inter_params.struct_size %u is not this library's %zu (zero the struct, set struct_size = sizeof, build against the library's headers)\n", p ? p->struct_size : 0u, sizeof(kvz_hip_inter_params));
    return -1;
  }
  if (!p || width <= 0 || height <= 0 || (width & 7) || (height & 7) || width > 64 * 255 || height > 64 * 255 || n_pictures > 65535) { fprintf(stderr, "kvz_hip_dev_inter_ctu_pass: bad geometry\n"); return -1; }
  if (p->qp < 0 || p->qp > 51) { fprintf(stderr, "kvz_hip_dev_inter_ctu_pass: picture QP %d outside 0..51\n", p->qp); return -1; }
  if (p->fme_level < 0 || p->fme_level > 4 || p->pu_depth_inter_max < 1 || p->pu_depth_inter_max > 3 || p->poc < 1 || p->fast_residual_cost < 0 || p->fast_residual_cost > 51) { fprintf(stderr, "kvz_hip_dev_inter_ctu_pass: unsupported parameters\n"); return -1; }
  if (p->ref_width || p->ref_height) {  // the pictures are tiles of a ref_width x ref_height frame
    if ((p->ref_width & 7) || (p->ref_height & 7) || p->tile_x < 0 || p->tile_y < 0 || (p->tile_x & 7) || (p->tile_y & 7) || p->tile_x + width > p->ref_width || p->tile_y + height > p->ref_height) {
      fprintf(stderr, "kvz_hip_dev_inter_ctu_pass: the %dx%d tile at (%d, %d) does not lie in the %dx%d reference frame\n", width, height, p->tile_x, p->tile_y, p->ref_width, p->ref_height);
      return -1;
    }
  }
  hipStream_t st = be().stream;
  const int wc = (width + 63) / 64, hc = (height + 63) / 64, ctus = wc * hc;
  const long total = (long)ctus * n_pictures;
  kvz::InterScratch &sc = kvz::inter_scratch();
  int dev_id = 0, n_cu = 256;
  KVZ_HIP_CHECK(hipGetDevice(&dev_id));
  KVZ_HIP_CHECK(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev_id));
  const char *env = getenv("KVZ_HIP_INTER_WG_PER_CU");
  // the kernel's build: with the residual coder's contexts in the LDS context sets only where the picture's coefficients are priced with them (kvz_inter_ctu.hpp)
  const bool cabac_build = !(p->qp < p->fast_residual_cost && p->qp < 50);
  const void *kernel = cabac_build ? (const void *)kvz::inter_ctu_ticket_kernel_cabac : (const void *)kvz::inter_ctu_ticket_kernel_fast;
  // resident workgroups (= wavefronts) per CU: what the kernel's registers and LDS allow -- a persistent grid, one workgroup per slot
  int fit = 0;
  KVZ_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&fit, kernel, KVZ_ICTU_THREADS, 0));
  // what THIS kernel build fits is the ceiling: a share (kvz_hip_dev_inter_set_share, per calling thread) divides it, the developer's environment value is clamped to it
  if (fit <= 0) fit = 8;
  int per_cu = fit / kvz::inter_share();
  if (env && atoi(env) > 0) per_cu = atoi(env) < fit ? atoi(env) : fit;
  if (per_cu < 1) per_cu = 1;
  int n_wg = n_cu * per_cu;
  if (getenv("KVZ_HIP_INTER_VERBOSE")) fprintf(stderr, "kvz_hip inter pass: %s build, %d workgroups per CU x %d CUs\n", cabac_build ? "cabac" : "fast", per_cu, n_cu);
  if ((long)n_wg > total) n_wg = (int)total;
  if (n_wg > sc.n_slabs) {
    if (sc.slabs) KVZ_HIP_CHECK(hipFree(sc.slabs));
    KVZ_HIP_CHECK(hipMalloc((void **)&sc.slabs, (size_t)n_wg * sizeof(kvz::InterSlab)));
    sc.n_slabs = n_wg;
  }
  if (total > sc.n_ctus) {
    if (sc.ctx) { KVZ_HIP_CHECK(hipFree(sc.ctx)); KVZ_HIP_CHECK(hipFree(sc.done)); KVZ_HIP_CHECK(hipFree(sc.items)); }
    KVZ_HIP_CHECK(hipMalloc((void **)&sc.ctx, (size_t)total * sizeof(kvz::ICtx)));
    KVZ_HIP_CHECK(hipMalloc((void **)&sc.done, (size_t)total * sizeof(unsigned)));
    KVZ_HIP_CHECK(hipMalloc((void **)&sc.items, (size_t)total * sizeof(uint32_t)));
    sc.n_ctus = total;
  }
  if (!sc.ticket) { KVZ_HIP_CHECK(hipMalloc((void **)&sc.ticket, 2 * sizeof(unsigned))); KVZ_HIP_CHECK(hipMalloc((void **)&sc.model, sizeof(kvz::InterModel))); }
  // the ticket list: anti-diagonals x + 2 y ascending (raster order per picture without WPP), pictures interleaved
  std::vector<uint32_t> items;
  items.reserve((size_t)total);
  if (p->no_wpp) {
    for (int y = 0; y < hc; y++) for (int x = 0; x < wc; x++) for (int f = 0; f < n_pictures; f++) items.push_back((uint32_t)f << 16 | (uint32_t)y << 8 | (uint32_t)x);
  } else {
    for (int d = 0; d <= (wc - 1) + 2 * (hc - 1); d++)
      for (int y = 0; y < hc; y++) { const int x = d - 2 * y; if (x < 0 || x >= wc) continue; for (int f = 0; f < n_pictures; f++) items.push_back((uint32_t)f << 16 | (uint32_t)y << 8 | (uint32_t)x); }
  }
  kvz::InterModel m;
  float fbits[128];
  for (int i = 0; i < 128; i++) fbits[i] = (float)kvz::kEntropyBits[i] / 32768.0f;
  kvz::inter_model_init(&m, p->qp, p->poc, kvz_hip_default_coeff_weights(p->qp) /* 0 from QP 50 on, where kvz_fast_coeff_cost is never used (rdo.c:311-340) */, fbits, p->mv_constraint, p->sao, p->deblock, p->fme_level, p->pu_depth_inter_max, p->no_wpp, p->fast_residual_cost,
                        width, height, p->ref_width, p->ref_height, p->tile_x, p->tile_y, p->no_tmvp);
  KVZ_HIP_CHECK(hipMemcpyAsync(sc.items, items.data(), (size_t)total * sizeof(uint32_t), hipMemcpyHostToDevice, st));
  KVZ_HIP_CHECK(hipMemcpyAsync(sc.model, &m, sizeof m, hipMemcpyHostToDevice, st));
  KVZ_HIP_CHECK(hipMemsetAsync(sc.done, 0, (size_t)total * sizeof(unsigned), st));
  KVZ_HIP_CHECK(hipMemsetAsync(sc.ticket, 0, 2 * sizeof(unsigned), st));
  KVZ_HIP_CHECK(hipStreamSynchronize(st));  // `items` and `m` are stack / heap objects of this call
  kvz::InterFrames F;
  F.W = width; F.H = height; F.wc = wc; F.hc = hc; F.frame_px = (long)width * height * 3 / 2; F.cells = (long)(width / 4) * (height / 4);
  F.src = src; F.ref = ref; F.ref_cu = ref_cu; F.rec = rec; F.cu = cu; F.coeff = coeff; F.ctx_out = sc.ctx; F.slabs = sc.slabs;
  F.prof = nullptr;
  F.tile_xy = (p->ref_width || p->ref_height) ? tile_xy : nullptr;
  F.ref_count = n_references > 0 ? n_references : 0;
#ifdef KVZ_ICTU_PROFILE
  static unsigned long long *d_prof = nullptr;
  if (!d_prof) KVZ_HIP_CHECK(hipMalloc((void **)&d_prof, kvz::IP_COUNT * sizeof(unsigned long long)));
  KVZ_HIP_CHECK(hipMemsetAsync(d_prof, 0, kvz::IP_COUNT * sizeof(unsigned long long), st));
  F.prof = d_prof;
#endif
  kvz::InterSched sched;
  sched.items = sc.items; sched.ticket = sc.ticket; sched.done = sc.done; sched.error = sc.ticket + 1; sched.total = (unsigned)total; sched.no_wpp = p->no_wpp;
  sched.wait_ticks = 3000000000ull;  // 30 s of the 100 MHz clock
  kvz::DevTimer &tm = kvz::inter_timer();
  if (!tm.e0) { KVZ_HIP_CHECK(hipEventCreate(&tm.e0)); KVZ_HIP_CHECK(hipEventCreate(&tm.e1)); }
  KVZ_HIP_CHECK(hipEventRecord(tm.e0, st));
  if (cabac_build) hipLaunchKernelGGL(kvz::inter_ctu_ticket_kernel_cabac, dim3((unsigned)n_wg), dim3(KVZ_ICTU_THREADS), 0, st, F, sc.model, kvz::device_tables(), sched);
  else hipLaunchKernelGGL(kvz::inter_ctu_ticket_kernel_fast, dim3((unsigned)n_wg), dim3(KVZ_ICTU_THREADS), 0, st, F, sc.model, kvz::device_tables(), sched);
  KVZ_HIP_CHECK(hipGetLastError());
  KVZ_HIP_CHECK(hipEventRecord(tm.e1, st));
  unsigned flags[2] = { 0, 0 };
  KVZ_HIP_CHECK(hipMemcpyAsync(flags, sc.ticket, sizeof flags, hipMemcpyDeviceToHost, st));
  KVZ_HIP_CHECK(hipStreamSynchronize(st));
  KVZ_HIP_CHECK(hipEventElapsedTime(&kvz::inter_kernel_ms(), tm.e0, tm.e1));
#ifdef KVZ_ICTU_PROFILE
  {
    unsigned long long hp[kvz::IP_COUNT];
    KVZ_HIP_CHECK(hipMemcpy(hp, d_prof, sizeof hp, hipMemcpyDeviceToHost));
    static const char *names[kvz::IP_COUNT] = { "merge MC+SATD", "early skip", "integer ME", "fractional ME", "candidates", "intra search", "intra recon", "inter quant/recon", "mock+rd cost", "copies", "io", "total search", "winner MC", "context copies", "zero-coeff alternative", "final syntax" };
    for (int i = 0; i < kvz::IP_COUNT; i++) fprintf(stderr, "ictu-profile %-18s %10.3f ms (sum over workgroups) %5.1f %%\n", names[i], hp[i] / 1e5, 100.0 * hp[i] / (double)hp[kvz::IP_TOTAL]);
  }
#endif
  return flags[1] ? -2 : 0;
}

float kvz_hip_dev_inter_kernel_ms(void) { return kvz::inter_kernel_ms(); }
void kvz_hip_dev_inter_set_share(int parts) { kvz::inter_share() = parts > 1 ? parts : 1; }
int kvz_hip_dev_inter_slots_per_cu(void)
{
  kvz::runtime_init(-1);
  int fit = 0;
  KVZ_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&fit, (const void *)kvz::inter_ctu_ticket_kernel_fast, KVZ_ICTU_THREADS, 0));
  return fit > 0 ? fit : 8;
}

int kvz_hip_dev_pu_search(const uint8_t *cur, const uint8_t *ref, int width, int height, const kvz_hip_me_pu *pus, int count, int max_pu_size,
                          const kvz_hip_me_params *params, kvz_hip_me_result *out)
{
  if (count <= 0) return 0;
  if (!params || params->fme_level < 0 || params->fme_level > 4) { fprintf(stderr, "kvz_hip_dev_pu_search: fme_level must be 0 .. 4\n"); return -1; }
  const dim3 grid((unsigned)count), block(256);
  const kvz::Tables *tb = kvz::device_tables();
  // PU sizes are mixed within a picture's list: the instantiation is sized for the largest one (pu-depth-inter 1-3: 32)
  if (max_pu_size <= 32) hipLaunchKernelGGL(kvz::dev_pu_search_kernel<32>, grid, block, 0, be().stream, cur, ref, width, height, pus, *params, tb, out);
  else if (max_pu_size <= 64) hipLaunchKernelGGL(kvz::dev_pu_search_kernel<64>, grid, block, 0, be().stream, cur, ref, width, height, pus, *params, tb, out);
  else { fprintf(stderr, "kvz_hip_dev_pu_search: PUs larger than 64 samples do not exist\n"); return -1; }
  KVZ_HIP_CHECK(hipGetLastError());
  return 0;
}

int kvz_hip_dev_inter_pred(const uint8_t *ref0, const uint8_t *ref1, uint8_t *pred, int width, int height, const kvz_hip_mc_pu *pus, int count, int max_pu_size)
{
  if (count <= 0) return 0;
  const dim3 grid((unsigned)count), block(256);
  const kvz::Tables *tb = kvz::device_tables();
  // PUs up to 16x16 with a picture width that is a multiple of 8 (all of kvazaar's): one wavefront per PU
  if (max_pu_size <= 16 && (width & 7) == 0 && !getenv("KVZ_HIP_MC_WORKGROUP_PER_PU"))
    hipLaunchKernelGGL(kvz::dev_inter_pred_wave_kernel, dim3((unsigned)((count + 3) / 4)), block, 0, be().stream, ref0, ref1, pred, width, height, pus, count, tb);
  else if (max_pu_size <= 16) hipLaunchKernelGGL(kvz::dev_inter_pred_kernel<16>, grid, block, 0, be().stream, ref0, ref1, pred, width, height, pus, tb);
  else if (max_pu_size <= 32) hipLaunchKernelGGL(kvz::dev_inter_pred_kernel<32>, grid, block, 0, be().stream, ref0, ref1, pred, width, height, pus, tb);
  else if (max_pu_size <= 64) hipLaunchKernelGGL(kvz::dev_inter_pred_kernel<64>, grid, block, 0, be().stream, ref0, ref1, pred, width, height, pus, tb);
  else { fprintf(stderr, "kvz_hip_dev_inter_pred: PUs larger than 64 samples do not exist\n"); return -1; }
  KVZ_HIP_CHECK(hipGetLastError());
  return 0;
}

void kvz_hip_dev_sao_frames(const uint8_t *in, uint8_t *out, int width, int height, int n_frames, const kvz_hip_sao_params *luma,
                            const kvz_hip_sao_params *chroma)
{
  if (n_frames <= 0) return;
  kvz::launch_sao(be().stream, in, out, width, height, n_frames, nullptr, luma, chroma);
  KVZ_HIP_CHECK(hipGetLastError());
}

int kvz_hip_dev_paste_tiles(uint8_t *frame, int width, int height, const uint8_t *slots, long slot_bytes, const int32_t *tiles /* host: n x (x, y, w, h, slot) */, int n, void *stream)
{
  if (n < 0 || n > 64) { fprintf(stderr, "kvz_hip_dev_paste_tiles: at most 64 tiles per launch\n"); return -1; }
  if (n == 0) return 0;
  kvz::PasteTable tb;
  tb.n = n;
  int hmax = 0;
  for (int i = 0; i < n; i++) {
    tb.x[i] = tiles[5 * i]; tb.y[i] = tiles[5 * i + 1]; tb.w[i] = tiles[5 * i + 2]; tb.h[i] = tiles[5 * i + 3]; tb.slot[i] = tiles[5 * i + 4];
    if (tb.h[i] > hmax) hmax = tb.h[i];
  }
  hipLaunchKernelGGL(kvz::dev_paste_tiles_kernel, dim3(2 * hmax < 512 ? 2 * hmax : 512, n), dim3(256), 0, stream ? (hipStream_t)stream : be().stream, frame, width, height, slots, slot_bytes, tb);
  KVZ_HIP_CHECK(hipGetLastError());
  return 0;
}

void kvz_hip_dev_picture_checksums(const uint8_t *frames, int width, int height, int n_frames, uint32_t *out)
{
  if (n_frames <= 0) return;
  KVZ_HIP_CHECK(hipMemsetAsync(out, 0, (size_t)n_frames * 3 * sizeof(uint32_t), be().stream));
  kvz::launch_checksums(be().stream, frames, width, height, n_frames, out);
  KVZ_HIP_CHECK(hipGetLastError());
}

void kvz_hip_batch_deblock(kvz_hip_batch *b, int qp, int beta_offset_div2, int tc_offset_div2)
{
  kvz::batch_enter(b);
  kvz::deblock_frames_on(b->stream, b->d_rec, b->F.W, b->F.H, b->n_frames, b->d_depth, qp, beta_offset_div2, tc_offset_div2);
}

void kvz_hip_batch_loop_filters(kvz_hip_batch *b, const kvz_hip_intra_cost_model *model, int deblock, int beta_offset_div2, int tc_offset_div2, int sao)
{
  if (!kvz::cost_model_known(model, "kvz_hip_batch_loop_filters")) { b->failed = 1; return; }  // no return value: the batch reports it (kvz_hip_batch_sync -> -1)
  kvz::batch_enter(b);
  const kvz::CtuFrames &F = b->F;
  const int n = b->n_frames;
  if (!sao) {
    if (deblock) kvz::deblock_frames_on(b->stream, b->d_rec, F.W, F.H, n, b->d_depth, model->qp, beta_offset_div2, tc_offset_div2);
    return;
  }
  const size_t pic_bytes = (size_t)F.frame_px * n, lcus = (size_t)F.wc * F.hc * n;
  if (!b->d_ver) {  // first use: the two intermediate pictures and the decision's records
    KVZ_HIP_CHECK(hipMalloc((void **)&b->d_ver, pic_bytes));
    KVZ_HIP_CHECK(hipMalloc((void **)&b->d_dbk, pic_bytes));
    KVZ_HIP_CHECK(hipMalloc((void **)&b->d_sao_stats, lcus * 3 * sizeof(kvz::SaoStats)));
    KVZ_HIP_CHECK(hipMalloc((void **)&b->d_sao_cand, lcus * 3 * sizeof(kvz::SaoCand)));
    KVZ_HIP_CHECK(hipMalloc((void **)&b->d_sao_recs, lcus * 3 * sizeof(kvz::SaoRec)));
    KVZ_HIP_CHECK(hipMalloc((void **)&b->d_sao_merge, lcus));
    KVZ_HIP_CHECK(hipMalloc((void **)&b->d_sao_fbits, 128 * sizeof(float)));
  }
  // R = d_rec (kept), V = d_ver, D = d_dbk
  KVZ_HIP_CHECK(hipMemcpyAsync(b->d_ver, b->d_rec, pic_bytes, hipMemcpyDeviceToDevice, b->stream));
  if (deblock) kvz::deblock_frames_on(b->stream, b->d_ver, F.W, F.H, n, b->d_depth, model->qp, beta_offset_div2, tc_offset_div2, 1);
  KVZ_HIP_CHECK(hipMemcpyAsync(b->d_dbk, b->d_ver, pic_bytes, hipMemcpyDeviceToDevice, b->stream));
  if (deblock) kvz::deblock_frames_on(b->stream, b->d_dbk, F.W, F.H, n, b->d_depth, model->qp, beta_offset_div2, tc_offset_div2, 2);
  KVZ_HIP_CHECK(hipMemcpyAsync(b->d_sao_fbits, model->entropy_fbits, 128 * sizeof(float), hipMemcpyHostToDevice, b->stream));
  const kvz::SaoGeom g{ F.W, F.H, F.wc, F.hc, F.frame_px };
  hipLaunchKernelGGL(kvz::dev_sao_stats_kernel, dim3((unsigned)(lcus * 3)), dim3(256), 0, b->stream, b->d_src, b->d_rec, b->d_ver, b->d_dbk, g, (kvz::SaoStats *)b->d_sao_stats, (kvz::SaoCand *)b->d_sao_cand);
  hipLaunchKernelGGL(kvz::dev_sao_chain_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, b->stream, (const kvz::SaoStats *)b->d_sao_stats, (const kvz::SaoCand *)b->d_sao_cand, g, n, b->d_sao_fbits, kvz::device_tables(),
                     model->lambda, (int)model->ctx_init[KVZ_HIP_CX_SAO_MERGE], (int)model->ctx_init[KVZ_HIP_CX_SAO_TYPE], model->no_wpp, b->d_sao_recs, b->d_sao_merge);
  // the SAO'd picture becomes the batch's reconstruction (R is not needed any more)
  kvz::launch_sao(b->stream, b->d_dbk, b->d_rec, F.W, F.H, n, b->d_sao_recs, nullptr, nullptr);
  KVZ_HIP_CHECK(hipGetLastError());
}

int kvz_hip_batch_sao_params(kvz_hip_batch *b, int frame, kvz_hip_sao_params *luma, kvz_hip_sao_params *chroma, uint8_t *merge)
{
  kvz::batch_enter(b);
  const size_t lcus = (size_t)b->F.wc * b->F.hc;
  if (!b->d_sao_recs) { fprintf(stderr, "kvz_hip_batch_sao_params: kvz_hip_batch_loop_filters(..., sao = 1) has not run on this batch\n"); return -1; }
  std::vector<kvz::SaoRec> recs(lcus * 3);
  KVZ_HIP_CHECK(hipStreamSynchronize(b->stream));
  KVZ_HIP_CHECK(hipMemcpy(recs.data(), b->d_sao_recs + (size_t)frame * lcus * 3, lcus * 3 * sizeof(kvz::SaoRec), hipMemcpyDeviceToHost));
  if (merge) KVZ_HIP_CHECK(hipMemcpy(merge, b->d_sao_merge + (size_t)frame * lcus, lcus, hipMemcpyDeviceToHost));
  for (size_t i = 0; i < lcus; i++) {
    auto unpack = [&](kvz_hip_sao_params *o, int plane, int slot) {
      const kvz::SaoRec r = recs[i * 3 + plane];
      if (slot == 0) { memset(o, 0, sizeof *o); o->bitdepth = 8; o->type = (int)(r & 0xff); o->eo_class = (int)((r >> 8) & 0xff); }
      o->band_position[slot] = (int)((r >> 16) & 0xff);
      for (int k = 0; k < 5; k++) o->offsets[5 * slot + k] = (int)(int8_t)(r >> (24 + 8 * k));
    };
    if (luma) unpack(&luma[i], 0, 0);
    if (chroma) { unpack(&chroma[i], 1, 0); unpack(&chroma[i], 2, 1); }
  }
  return kvz::batch_check(b);
}

// The entropy coder in its real mode (kvz_entropy.hpp): the slice data of every picture of the batch, substream by substream, from the device-resident results of the
// CTU pass (and of the loop filters' SAO decision).  Pictures are coded in chunks whose bin records fit a scratch budget (KVZ_HIP_ENTROPY_SCRATCH_MB, default 49152:
// 25 MB per 1080p picture at the default capacity of 12 288 records per CTU); a CTU that produces more records than that makes its chunk run again with the room it needs.
namespace kvz {
// kvz_init_contexts for a B slice (context.c:36-193 row 0 of every table, :202-305) in the entropy coder's numbering: KVZ_HIP_CX_* then KVZ_EB_CX_* (kvz_entropy.hpp)
inline void entropy_b_slice_contexts(int qp, uint8_t out[KVZ_ENTROPY_CTXS])
{
  static const uint8_t split[3] = { 107, 139, 126 }, skip[3] = { 197, 185, 201 }, mvd[2] = { 169, 198 }, inter_dir[5] = { 95, 79, 63, 31, 31 };
  uint8_t init[KVZ_ENTROPY_CTXS];
  memset(init, 154, sizeof init);
  for (int i = 0; i < 3; i++) { init[KVZ_HIP_CX_SPLIT + i] = split[i]; init[KVZ_EB_CX_SKIP + i] = skip[i]; }
  init[KVZ_HIP_CX_PART] = 154; init[KVZ_HIP_CX_INTRA] = 183; init[KVZ_HIP_CX_CHROMA] = 152;
  init[KVZ_HIP_CX_CBF_LUMA] = 153; init[KVZ_HIP_CX_CBF_LUMA + 1] = 111;
  init[KVZ_HIP_CX_CBF_CHROMA] = 149; init[KVZ_HIP_CX_CBF_CHROMA + 1] = 92; init[KVZ_HIP_CX_CBF_CHROMA_DEEP] = 167; init[KVZ_HIP_CX_CBF_CHROMA_DEEP + 1] = 154;
  b_slice_residual_init_values(init + KVZ_HIP_CX_SIG_CG);  // kvz_inter_host.hpp: the tables' row 0
  init[KVZ_HIP_CX_SAO_MERGE] = 153; init[KVZ_HIP_CX_SAO_TYPE] = 160;
  init[KVZ_EB_CX_MERGE_FLAG] = 154; init[KVZ_EB_CX_MERGE_IDX] = 137; init[KVZ_EB_CX_PRED_MODE] = 134;
  init[KVZ_EB_CX_MVD] = mvd[0]; init[KVZ_EB_CX_MVD + 1] = mvd[1]; init[KVZ_EB_CX_MVP_IDX] = 168;
  for (int i = 0; i < 5; i++) init[KVZ_EB_CX_INTER_DIR + i] = inter_dir[i];
  init[KVZ_EB_CX_ROOT_CBF] = 79;
  for (int i = 0; i < KVZ_ENTROPY_CTXS; i++) out[i] = (uint8_t)ctx_state(qp, init[i]);
}
// scratch of the entropy coder, kept between calls (grow-only; hipMalloc / hipFree per call cost more than a small batch's kernels): one caller at a time
struct EntropyScratch {
  std::mutex lock;
  using Buf = DevBuf;
  Buf bins, nbins, nbits, sizes, ins, offsets, bound_offsets, room, rowctx, scratch, out, not_last;
  hipStream_t side = nullptr;          // stage 2 beside stage 1's second part
  hipEvent_t ev_first = nullptr, ev_rows = nullptr, ev_pre = nullptr;  // ev_pre: everything in front of stage 3 has run (see chain_queued)
  hipEvent_t ev_tail = nullptr;        // behind a call's last kernel, when the call returned with its download in flight
  bool ev_tail_set = false;
  static void *need(Buf &b, size_t bytes)
  {
    if (bytes > b.bytes) {
      if (b.p) KVZ_HIP_CHECK(hipFree(b.p));
      b.bytes = bytes + bytes / 8;
      KVZ_HIP_CHECK(hipMalloc(&b.p, b.bytes));
    }
    return b.p;
  }
};
inline EntropyScratch &entropy_scratch(int device) { static EntropyScratch s[64]; return s[device & 63]; }  // buffers live on the device they were allocated on
}  // namespace kvz
namespace kvz {
// The stages of kvz_entropy.hpp over n_frames pictures, in chunks whose bin records fit the scratch budget.  job(f0, nf): the chunk's inputs (everything of EntropyJob but
// the scratch pointers); not_last: host flags or null.
// own_out: a compaction buffer of the caller's instead of the shared one (grown here); with `defer` the call returns when the LAST chunk's download has been queued on
// `stream` -- the caller synchronises the stream before it reads `out` -- which needs such a buffer: the next call, on another stream, compacts into the shared one at once.
inline long entropy_code_pictures(hipStream_t stream, int device, int n, int wc, int hc, int no_wpp, const uint8_t *not_last, const std::function<EntropyJob(int, int)> &job,
                                  uint8_t *out, size_t capacity, uint32_t *substream_bytes, const std::function<void()> &chain_queued = nullptr,
                                  EntropyScratch::Buf *own_out = nullptr, bool defer = false)
{
  EntropyScratch &S = entropy_scratch(device);
  std::lock_guard<std::mutex> guard(S.lock);
  if (S.ev_tail_set) KVZ_HIP_CHECK(hipStreamWaitEvent(stream, S.ev_tail, 0));  // the previous call's last kernels (on its own stream) read scratch this call writes
  static const bool times = getenv("KVZ_HIP_ENTROPY_TIMES") != nullptr;  // developer: host-side phase clock on stderr
  auto t_last = std::chrono::steady_clock::now();
  auto mark = [&](const char *what) { if (!times) return; const auto now = std::chrono::steady_clock::now(); fprintf(stderr, "kvz_hip entropy: %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count()); t_last = now; };
  const int ctus = wc * hc, rows = no_wpp ? 1 : hc;
  size_t budget = 49152;
  if (const char *e = getenv("KVZ_HIP_ENTROPY_SCRATCH_MB")) { const long v = atol(e); if (v > 0) budget = (size_t)v; }
  budget <<= 20;
  uint32_t cap = 12288;
  if (const char *e = getenv("KVZ_HIP_ENTROPY_CAP")) { const long v = atol(e); if (v > 0) cap = ((uint32_t)v + 15u) & ~15u; }
  std::vector<uint32_t> counts, bound_bits, sizes;
  std::vector<unsigned long long> offsets, bound_offsets;
  size_t total = 0;
  long rc = 0;
  for (int f0 = 0; f0 < n && rc == 0;) {
    int nf = (int)(budget / ((size_t)ctus * cap * sizeof(uint32_t)));
    if (nf < 1) nf = 1;
    if (nf > n - f0) nf = n - f0;
    const long items = (long)nf * ctus, streams = (long)nf * rows;
    uint32_t *d_bins = (uint32_t *)S.need(S.bins, (size_t)items * cap * sizeof(uint32_t)), *d_nbins = (uint32_t *)S.need(S.nbins, (size_t)items * sizeof(uint32_t));
    uint32_t *d_nbits = (uint32_t *)S.need(S.nbits, (size_t)items * sizeof(uint32_t)), *d_sizes = (uint32_t *)S.need(S.sizes, (size_t)streams * sizeof(uint32_t));
    unsigned long long *d_offsets = (unsigned long long *)S.need(S.offsets, (size_t)streams * sizeof(unsigned long long));
    unsigned long long *d_bound_offsets = (unsigned long long *)S.need(S.bound_offsets, (size_t)streams * sizeof(unsigned long long));
    uint8_t *d_rowctx = (uint8_t *)S.need(S.rowctx, (size_t)nf * hc * KVZ_ENTROPY_CTXS), *d_out = nullptr, *d_scratch = nullptr;
    uint8_t *d_not_last = not_last ? (uint8_t *)S.need(S.not_last, (size_t)nf) : nullptr;
    if (not_last) KVZ_HIP_CHECK(hipMemcpyAsync(d_not_last, not_last + f0, (size_t)nf, hipMemcpyHostToDevice, stream));
    EntropyJob J = job(f0, nf);
    J.bins = d_bins; J.nbins = d_nbins; J.nbits = d_nbits; J.cap = cap; J.row_ctx = d_rowctx; J.not_last = d_not_last;
    static const bool serial_bins = [] { const char *e = getenv("KVZ_HIP_ENTROPY_BINS"); return e && !strcmp(e, "serial"); }();  // developer: the one-lane-walks-it-all form of stage 1
    // WPP: the row contexts (stage 2: one lane per picture, a long chain) need the bins of the first two CTUs of every row only -- those and stage 2 run on a second
    // stream beside the bins of all the other CTUs
    const bool early_rows = !serial_bins && !no_wpp && wc > 2;
    if (serial_bins) hipLaunchKernelGGL(dev_entropy_bins_kernel, dim3((unsigned)((items + 63) / 64)), dim3(64), 0, stream, J, device_tables(), items);
    else if (!early_rows) hipLaunchKernelGGL(dev_entropy_bins_phased_kernel, dim3((unsigned)((items + 63) / 64)), dim3(64), 0, stream, J, device_tables(), items, 0);
    else {
      const long first = (long)nf * hc * 2, rest = items - first;
      if (!S.side) { KVZ_HIP_CHECK(hipStreamCreateWithFlags(&S.side, hipStreamNonBlocking)); KVZ_HIP_CHECK(hipEventCreateWithFlags(&S.ev_first, hipEventDisableTiming)); KVZ_HIP_CHECK(hipEventCreateWithFlags(&S.ev_rows, hipEventDisableTiming)); }
      KVZ_HIP_CHECK(hipEventRecord(S.ev_first, stream));  // (what the caller queued before -- the pass, the loop filters -- comes first on both streams)
      KVZ_HIP_CHECK(hipStreamWaitEvent(S.side, S.ev_first, 0));
      hipLaunchKernelGGL(dev_entropy_bins_phased_kernel, dim3((unsigned)((first + 63) / 64)), dim3(64), 0, S.side, J, device_tables(), first, 1);
      hipLaunchKernelGGL(dev_entropy_row_ctx_kernel<8>, dim3((unsigned)((nf + 7) / 8)), dim3(8), 0, S.side, J, device_tables());
      KVZ_HIP_CHECK(hipEventRecord(S.ev_rows, S.side));
      hipLaunchKernelGGL(dev_entropy_bins_phased_kernel, dim3((unsigned)((rest + 63) / 64)), dim3(64), 0, stream, J, device_tables(), rest, 2);
      KVZ_HIP_CHECK(hipStreamWaitEvent(stream, S.ev_rows, 0));  // the counts of every CTU are read next
    }
    counts.resize((size_t)items); bound_bits.resize((size_t)items);
    KVZ_HIP_CHECK(hipMemcpyAsync(counts.data(), d_nbins, (size_t)items * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    KVZ_HIP_CHECK(hipMemcpyAsync(bound_bits.data(), d_nbits, (size_t)items * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    KVZ_HIP_CHECK(hipStreamSynchronize(stream));
    mark("bins + counts down");
    uint32_t most = 0;
    for (uint32_t c : counts) most = c > most ? c : most;
    const bool again = most > cap;
    if (!again) {
      // room per substream: the bound of its CTUs' bits, the coder's flush, and an emulation prevention byte after every two bytes at worst
      bound_offsets.resize((size_t)streams);
      unsigned long long scratch_bytes = 0;
      const long per_stream = no_wpp ? ctus : wc;
      for (long i = 0; i < streams; i++) {
        unsigned long long bits = 0;
        for (long k = 0; k < per_stream; k++) bits += bound_bits[(size_t)(i * per_stream + k)];
        bound_offsets[(size_t)i] = scratch_bytes;
        scratch_bytes += (((bits + 7) / 8 + 16) * 3 / 2 + 15) & ~15ull;
      }
      d_scratch = (uint8_t *)S.need(S.scratch, scratch_bytes ? scratch_bytes : 16);
      mark("host: bounds");
      uint32_t *d_room = (uint32_t *)S.need(S.room, (size_t)streams * sizeof(uint32_t));
      hipLaunchKernelGGL(dev_entropy_stream_room_kernel, dim3((unsigned)((streams + 255) / 256)), dim3(256), 0, stream, d_nbits, streams, per_stream, d_room);
      hipLaunchKernelGGL(dev_entropy_offsets_kernel, dim3(1), dim3(KVZ_ENTROPY_SCAN_THREADS), 0, stream, d_room, streams, d_bound_offsets);  // == bound_offsets, without a copy
      if (chain_queued && f0 + nf >= n) {
        if (!S.ev_pre) KVZ_HIP_CHECK(hipEventCreateWithFlags(&S.ev_pre, hipEventDisableTiming));
        KVZ_HIP_CHECK(hipEventRecord(S.ev_pre, stream));
      }
      static const int lanes = [] { const char *e = getenv("KVZ_HIP_ENTROPY_LANES"); const int v = e ? atoi(e) : 64; return v == 32 || v == 16 || v == 8 ? v : 64; }();
      uint32_t *d_ins = (uint32_t *)S.need(S.ins, (size_t)streams * sizeof(uint32_t));
      if (!early_rows && !no_wpp) hipLaunchKernelGGL(dev_entropy_row_ctx_kernel<8>, dim3((unsigned)((nf + 7) / 8)), dim3(8), 0, stream, J, device_tables());
      {
        const dim3 grid((unsigned)((streams + lanes - 1) / lanes)), block((unsigned)lanes);
        // LDS: a workgroup of this stage takes exactly what a workgroup of the CTU pass takes (unused dynamic LDS on top of its 12 KB).  A pass started beside this stage
        // (kvz_hip_batch_entropy_code_then) is one persistent launch whose workgroups never leave: the 12 KB hole a workgroup of this stage left behind fitted none of them,
        // and every CU that had hosted one ran the rest of the pass with seven workgroups instead of eight
        const unsigned pad = 20480u - (unsigned)(lanes * KVZ_ENTROPY_CTX_STRIDE + 1024);
        if (lanes == 64) hipLaunchKernelGGL(dev_entropy_code_wide_kernel<64>, grid, block, pad, stream, J, device_tables(), streams, d_sizes, d_bound_offsets, d_scratch);
        else if (lanes == 32) hipLaunchKernelGGL(dev_entropy_code_wide_kernel<32>, grid, block, pad, stream, J, device_tables(), streams, d_sizes, d_bound_offsets, d_scratch);
        else if (lanes == 8) hipLaunchKernelGGL(dev_entropy_code_wide_kernel<8>, grid, block, pad, stream, J, device_tables(), streams, d_sizes, d_bound_offsets, d_scratch);
        else hipLaunchKernelGGL(dev_entropy_code_wide_kernel<16>, grid, block, pad, stream, J, device_tables(), streams, d_sizes, d_bound_offsets, d_scratch);
      }
      // from here on the device is nearly idle -- stage 3 is a few hundred wavefronts on their own chains, then a copy --: the caller's moment to queue other work
      // The caller's work is a persistent pass that takes every free workgroup slot the moment it starts: stage 3 must have its slots first (it then runs 39 ms beside
      // the pass; started behind it, it waits 310 ms for the pass to end).  So the pass is queued only when everything in front of stage 3 has run -- stage 3 is then
      // dispatched at once, the pass a launch latency later.
      if (chain_queued && f0 + nf >= n) { KVZ_HIP_CHECK(hipEventSynchronize(S.ev_pre)); chain_queued(); }
      hipLaunchKernelGGL(dev_entropy_escape_count_kernel, dim3((unsigned)streams), dim3(256), 0, stream, d_scratch, d_bound_offsets, d_sizes, d_ins);  // (behind the caller's pass, if it queued one: see kvz_hip_batch_entropy_code_then)
      sizes.resize((size_t)streams);
      KVZ_HIP_CHECK(hipMemcpyAsync(sizes.data(), d_sizes, (size_t)streams * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
      KVZ_HIP_CHECK(hipStreamSynchronize(stream));
      mark("row contexts + coder");
      offsets.resize((size_t)streams);
      unsigned long long chunk_bytes = 0;
      for (long i = 0; i < streams; i++) { offsets[(size_t)i] = chunk_bytes; chunk_bytes += sizes[(size_t)i]; }
      if (total + chunk_bytes > capacity) {
        fprintf(stderr, "kvz_hip_batch_entropy_code: the output buffer is too small (%zu bytes needed so far)\n", (size_t)(total + chunk_bytes));
        rc = -1;
      } else {
        d_out = (uint8_t *)S.need(own_out ? *own_out : S.out, chunk_bytes ? chunk_bytes : 1);
        hipLaunchKernelGGL(dev_entropy_offsets_kernel, dim3(1), dim3(KVZ_ENTROPY_SCAN_THREADS), 0, stream, d_sizes, streams, d_offsets);  // == offsets, without a copy
        hipLaunchKernelGGL(dev_entropy_compact_kernel, dim3((unsigned)streams), dim3(256), 0, stream, d_scratch, d_bound_offsets, d_sizes, d_ins, d_offsets, d_out);
        KVZ_HIP_CHECK(hipGetLastError());
        const bool in_flight = defer && own_out && f0 + nf >= n;
        if (in_flight) {
          if (!S.ev_tail) KVZ_HIP_CHECK(hipEventCreateWithFlags(&S.ev_tail, hipEventDisableTiming));
          KVZ_HIP_CHECK(hipEventRecord(S.ev_tail, stream));
          S.ev_tail_set = true;
        }
        KVZ_HIP_CHECK(hipMemcpyAsync(out + total, d_out, chunk_bytes, hipMemcpyDeviceToHost, stream));
        if (!in_flight) KVZ_HIP_CHECK(hipStreamSynchronize(stream));
        mark("compact + slice data down");
        memcpy(substream_bytes + (size_t)f0 * rows, sizes.data(), (size_t)streams * sizeof(uint32_t));
        total += chunk_bytes;
        f0 += nf;
      }
    }
    if (again) {
      cap = (most + 1023u) & ~1023u;  // this chunk again, with room for its largest CTU
    }
  }
  return rc ? rc : (long)total;
}
}  // namespace kvz

void kvz_hip_batch_entropy_defer_download(kvz_hip_batch *b, int on) { if (b) b->entropy_deferred = on != 0; }

long kvz_hip_batch_entropy_code(kvz_hip_batch *b, const kvz_hip_intra_cost_model *model, int sao, uint8_t *out, size_t capacity, uint32_t *substream_bytes)
{
  return kvz_hip_batch_entropy_code_tiles(b, model, sao, nullptr, out, capacity, substream_bytes);
}
long kvz_hip_batch_entropy_code_tiles(kvz_hip_batch *b, const kvz_hip_intra_cost_model *model, int sao, const uint8_t *not_last, uint8_t *out, size_t capacity,
                                      uint32_t *substream_bytes)
{
  return kvz_hip_batch_entropy_code_then(b, model, sao, not_last, out, capacity, substream_bytes, nullptr, nullptr);
}
long kvz_hip_batch_entropy_code_then(kvz_hip_batch *b, const kvz_hip_intra_cost_model *model, int sao, const uint8_t *not_last, uint8_t *out, size_t capacity,
                                     uint32_t *substream_bytes, kvz_hip_batch *next, const kvz_hip_intra_cost_model *next_model)
{
  if (!b || !kvz::cost_model_known(model, "kvz_hip_batch_entropy_code")) return -1;
  if (next && !kvz::cost_model_known(next_model, "kvz_hip_batch_entropy_code_then (next_model)")) return -3;  // nothing has been queued anywhere
  kvz::batch_enter(b);
  const kvz::CtuFrames &F = b->F;
  const int ctus = F.wc * F.hc;
  // `next`'s pass is queued exactly once on every path below (the caller synchronises `next` whatever this call returns): by the coder at its quiet moment, or here
  int launched = 0;
  bool started = false;
  // ... and what this call queues behind it on its own stream waits for that pass to end: the coder's last kernels could not run beside a persistent pass anyway (no free
  // workgroup slot), and kernels standing at the head of ANOTHER hardware queue while the pass runs are what one two-batch chain in forty stalled on (DESIGN.md section 8
  // item 8: none with one hardware queue) -- behind an event the queue holds a barrier packet instead of a dispatch
  auto start_next = [&] {
    if (next && !started) {
      started = true;
      launched = kvz_hip_intra_frames(next, next_model);
      kvz::batch_enter(b);
      if (launched > 0) KVZ_HIP_CHECK(hipStreamWaitEvent(b->stream, next->ev1, 0));
    }
  };
  struct StartOnExit { decltype(start_next) &f; ~StartOnExit() { f(); } } start_on_exit{ start_next };
  if (sao && !b->d_sao_recs) { fprintf(stderr, "kvz_hip_batch_entropy_code: kvz_hip_batch_loop_filters(..., sao = 1) has not run on this batch\n"); return -1; }
  if (model->search_nxn && !b->d_part) { fprintf(stderr, "kvz_hip_batch_entropy_code: the batch has no NxN partition maps\n"); return -1; }
  KVZ_HIP_CHECK(hipStreamSynchronize(b->stream));
  if (kvz::batch_check(b) != 0) return -1;
  const long cells8 = (long)(F.H >> 3) * (F.W >> 3), cells4 = (long)(F.H >> 2) * (F.W >> 2);
  auto job = [&](int f0, int nf) {
    kvz::EntropyJob J;
    memset(&J, 0, sizeof J);
    J.W = F.W; J.H = F.H; J.wc = F.wc; J.hc = F.hc; J.n_frames = nf; J.no_wpp = model->no_wpp;
    J.depth = b->d_depth + f0 * cells8; J.mode = b->d_mode + f0 * cells8;
    J.part = model->search_nxn ? b->d_part + f0 * cells8 : nullptr; J.mode4 = model->search_nxn ? b->d_mode4 + f0 * cells4 : nullptr;
    J.coeff = b->d_coeff + (size_t)f0 * ctus * KVZ_HIP_CTU_COEFFS;
    J.sao = sao ? (const kvz::SaoRec *)b->d_sao_recs + (size_t)f0 * ctus * 3 : nullptr; J.sao_merge = sao ? b->d_sao_merge + (size_t)f0 * ctus : nullptr;
    memcpy(J.ctx_init, model->ctx_init, sizeof model->ctx_init < sizeof J.ctx_init ? sizeof model->ctx_init : sizeof J.ctx_init);
    return J;
  };
  const long total = kvz::entropy_code_pictures(b->stream, b->device, b->n_frames, F.wc, F.hc, model->no_wpp, not_last, job, out, capacity, substream_bytes,
                                                next ? std::function<void()>(start_next) : std::function<void()>(),
                                                b->entropy_deferred ? &b->entropy_out : nullptr, b->entropy_deferred != 0);
  start_next();  // the coder failed before its last chunk: the pass starts now
  return next && launched < 0 ? -2 : total;
}

// ... of B pictures: the CU records, levels and (with sao) the SAO decisions of the last kvz_hip_dev_loop_filters_inter as the inter CTU pass / the loop filters left them
long kvz_hip_dev_entropy_code_inter(const kvz_hip_cu_info *cu, const kvz_hip_cu_info *ref_cu, const int16_t *coeff, int width, int height, int n_pictures,
                                    const kvz_hip_inter_params *params, uint8_t *out, size_t capacity, uint32_t *substream_bytes)
{
  if (n_pictures <= 0) return 0;
  if (params && params->struct_size != sizeof(kvz_hip_inter_params)) { fprintf(stderr, "kvz_hip_dev_entropy_code_inter: kvz_hip_inter_params.struct_size %u is not this library's %zu\n", params->struct_size, sizeof(kvz_hip_inter_params)); return -1; }
  if (!cu || !ref_cu || !coeff || !params || width <= 0 || height <= 0 || (width & 7) || (height & 7) || params->qp < 0 || params->qp > 51 || params->poc < 1) {
    fprintf(stderr, "kvz_hip_dev_entropy_code_inter: bad argument\n");
    return -1;
  }
  const int wc = (width + 63) >> 6, hc = (height + 63) >> 6, ctus = wc * hc;
  const long cells4 = (long)(height >> 2) * (width >> 2);
  kvz::LoopScratch &ls = kvz::loop_scratch();
  if (params->sao && (!ls.recs || ls.lcus < (size_t)ctus * n_pictures)) { fprintf(stderr, "kvz_hip_dev_entropy_code_inter: kvz_hip_dev_loop_filters_inter(..., sao = 1) has not run on these pictures\n"); return -1; }
  uint8_t init[KVZ_ENTROPY_CTXS];
  kvz::entropy_b_slice_contexts(params->qp, init);
  int device = 0;
  KVZ_HIP_CHECK(hipGetDevice(&device));
  auto job = [&](int f0, int nf) {
    kvz::EntropyJob J;
    memset(&J, 0, sizeof J);
    J.W = width; J.H = height; J.wc = wc; J.hc = hc; J.n_frames = nf; J.no_wpp = params->no_wpp;
    J.cu = cu + f0 * cells4; J.ref_cu = ref_cu + f0 * cells4; J.poc = params->no_tmvp ? 0 : params->poc;  // (the coder only asks the POC whether temporal predictors exist)
    J.coeff = coeff + (size_t)f0 * ctus * KVZ_HIP_CTU_COEFFS;
    J.sao = params->sao ? ls.recs + (size_t)f0 * ctus * 3 : nullptr; J.sao_merge = params->sao ? ls.merge + (size_t)f0 * ctus : nullptr;
    memcpy(J.ctx_init, init, sizeof init);
    return J;
  };
  return kvz::entropy_code_pictures(be().stream, device, n_pictures, wc, hc, params->no_wpp, nullptr, job, out, capacity, substream_bytes);
}

void kvz_hip_dev_picture_md5(const uint8_t *frames, int width, int height, int n_frames, uint8_t *out)
{
  if (n_frames <= 0) return;
  const long n = 3L * n_frames;
  hipLaunchKernelGGL(kvz::dev_md5_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, be().stream, frames, width, height, n, out);
  KVZ_HIP_CHECK(hipGetLastError());
}

int kvz_hip_batch_md5(kvz_hip_batch *b, uint8_t *host_out)
{
  kvz::batch_enter(b);
  const long n = 3L * b->n_frames;
  if (n <= 0) return 0;
  uint8_t *d = nullptr;
  KVZ_HIP_CHECK(hipMallocAsync((void **)&d, (size_t)n * 16, b->stream));
  hipLaunchKernelGGL(kvz::dev_md5_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, b->stream, b->d_rec, b->F.W, b->F.H, n, d);
  KVZ_HIP_CHECK(hipGetLastError());
  KVZ_HIP_CHECK(hipMemcpyAsync(host_out, d, (size_t)n * 16, hipMemcpyDeviceToHost, b->stream));
  KVZ_HIP_CHECK(hipFreeAsync(d, b->stream));
  KVZ_HIP_CHECK(hipStreamSynchronize(b->stream));
  return kvz::batch_check(b);
}

int kvz_hip_batch_checksums(kvz_hip_batch *b, uint32_t *host_out)
{
  kvz::batch_enter(b);
  const int n = b->n_frames;
  if (n <= 0) return 0;
  uint32_t *d = nullptr;
  KVZ_HIP_CHECK(hipMallocAsync((void **)&d, (size_t)n * 3 * sizeof(uint32_t), b->stream));
  KVZ_HIP_CHECK(hipMemsetAsync(d, 0, (size_t)n * 3 * sizeof(uint32_t), b->stream));
  kvz::launch_checksums(b->stream, b->d_rec, b->F.W, b->F.H, n, d);
  KVZ_HIP_CHECK(hipGetLastError());
  KVZ_HIP_CHECK(hipMemcpyAsync(host_out, d, (size_t)n * 3 * sizeof(uint32_t), hipMemcpyDeviceToHost, b->stream));
  KVZ_HIP_CHECK(hipFreeAsync(d, b->stream));
  KVZ_HIP_CHECK(hipStreamSynchronize(b->stream));
  return kvz::batch_check(b);
}

}  // extern "C"
