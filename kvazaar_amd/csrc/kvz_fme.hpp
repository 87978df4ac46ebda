// kvz_fme.hpp -- the fractional motion search's arithmetic as a streaming kernel: search_frac (search_inter.c:974-1130) per PU =
// get_extended_block (ipol-generic.c:761-814) -> filter_{hpel,qpel}_blocks_{hor_ver,diag}_luma (ipol-generic.c:213-679) ->
// satd_any_size / satd_any_size_quad (picture-generic.c:404-471) -- fused: one workgroup per PU stages the (w + 8) x (h + 8) reference
// window in LDS once (clamped addressing IS the extended block), builds each step's four candidate planes from shared 14-bit horizontal
// intermediates and scores them against the PU's source block with the 8x8 Hadamard; the planes never go to HBM.  What leaves the
// device per PU: the integer-position SATD and four SATDs per step; the motion-vector bit costs and the choice between steps stay with
// the caller (search_inter.c:1104-1160), which feeds the best half-pel offset back for the quarter-pel steps.
//
// Every candidate plane is HV(hf, vf) at (row offset, column offset) in the notation of kvz_ops.hpp (FmePlane, kvz_tables.hpp
// fme_planes) with luma_filter[0] = {0,0,0,64,0,0,0,0} as the identity: H-only and V-only planes are the same expression
// ((64 g) >> 6 == g exactly), so one code path produces all sixteen planes.
// Algorithmic bytes per PU: (w + 8)(h + 8) window + w h source read, 4 bytes per cost written.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/kvz_hip_dev.h"
#include "kvz_ops.hpp"
#include "kvz_tables.hpp"

namespace kvz {

template <int MAXN> __global__ void __launch_bounds__(256) dev_fme_kernel(const u8 *cur, const u8 *ref, const int W, const int H, const kvz_hip_fme_pu *pus, const int steps,
                                                                         const Tables *tb, u32 *out)
{
  constexpr int WS = MAXN + 8;                 // window stride
  __shared__ alignas(8) u8 s_win[(MAXN + 8) * WS];
  __shared__ i16 s_g[(MAXN + 8) * (MAXN + 1)];
  __shared__ alignas(8) u8 s_cur[MAXN * MAXN];
  __shared__ alignas(8) u8 s_pred[4][MAXN * MAXN];
  __shared__ u32 s_cost[4];
  const kvz_hip_fme_pu pu = pus[blockIdx.x];
  const int w = pu.w, h = pu.h, tid = threadIdx.x;
  u32 *o = out + (long)blockIdx.x * KVZ_HIP_FME_COSTS;
  if (w > MAXN || h > MAXN) return;  // routed to a larger instantiation by the host
  // window origin in the picture: ext_origin - (3, 3), ext_origin = (x + mv.x - 1, y + mv.y - 1)  (search_inter.c:1016-1030)
  const int X0 = pu.x + pu.mv_x - 4, Y0 = pu.y + pu.mv_y - 4;
  for (int i = tid; i < (h + 8) * (w + 8); i += 256) {
    const int r = i / (w + 8), c = i - r * (w + 8);
    s_win[r * WS + c] = ref[(long)iclip(0, H - 1, Y0 + r) * W + iclip(0, W - 1, X0 + c)];
  }
  for (int i = tid; i < w * h; i += 256) { const int y = i / w, x = i - y * w; s_cur[i] = cur[(long)(pu.y + y) * W + pu.x + x]; }
  const int tiles = (w >> 3) * (h >> 3), tw = w >> 3;
  auto score = [&](int planes) {  // SATD of s_pred[0..planes) against s_cur: one lane per (plane, 8x8 tile)
    if (tid < 4) s_cost[tid] = 0;
    __syncthreads();
    for (int t = tid; t < planes * tiles; t += 256) {
      const int p = t / tiles, tt = t - p * tiles, ty = tt / tw, tx = tt - ty * tw, base = ty * 8 * w + tx * 8;
      dev_pk16 d[8][4];
      for (int r = 0; r < 8; r++)
        dev_diff_row(*reinterpret_cast<const uint2 *>(&s_pred[p][base + r * w]), *reinterpret_cast<const uint2 *>(&s_cur[base + r * w]), d[r]);
      atomicAdd(&s_cost[p], (dev_satd8_regs(d) + 2) >> 2);
    }
    __syncthreads();
  };
  __syncthreads();
  // integer position: the block at ext_origin + (1, 1) (search_inter.c:1059-1063)
  for (int i = tid; i < w * h; i += 256) { const int y = i / w, x = i - y * w; s_pred[0][i] = s_win[(y + 4) * WS + x + 4]; }
  __syncthreads();
  score(1);
  if (tid == 0) o[0] = s_cost[0];
  for (int step = 0; step < 4; step++) {
    if (!(steps & (1 << step))) continue;
    FmePlane pl[4];
    fme_planes(step, pu.hpel_x, pu.hpel_y, pl);
    int done = 0;  // bit p: plane p produced
    for (int first = 0; first < 4; first++) {
      if (done & (1 << first)) continue;
      const int hf = pl[first].hf;
      // G[r][c] = sum_i f[i] * S(r - 3, c - 3 + i) in window coordinates: rows 0 .. h + 7, columns 0 .. w
      const int8_t *f = tb->luma_filter[hf];
      __syncthreads();
      for (int i = tid; i < (h + 8) * (w + 1); i += 256) {
        const int r = i / (w + 1), c = i - r * (w + 1);
        const u8 *p = &s_win[r * WS + c];
        int t = 0;
        for (int k = 0; k < 8; k++) t += f[k] * (int)p[k];
        s_g[r * (MAXN + 1) + c] = (i16)t;
      }
      __syncthreads();
      for (int p = first; p < 4; p++) {
        if (pl[p].hf != hf) continue;
        done |= 1 << p;
        const int8_t *vf = tb->luma_filter[pl[p].vf];
        const int ro = pl[p].roff, co = pl[p].coff;
        for (int i = tid; i < w * h; i += 256) {
          const int y = i / w, x = i - y * w;
          int t = 0;
          for (int j = 0; j < 8; j++) t += vf[j] * (int)s_g[(y + ro + j) * (MAXN + 1) + x + co];
          s_pred[p][i] = fin14(t >> 6);
        }
      }
    }
    __syncthreads();
    score(4);
    if (tid < 4) o[1 + 4 * step + tid] = s_cost[tid];
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Motion-compensated prediction of PU lists (inter.c:371-575 inter_recon_unipred / kvz_inter_recon_bipred): per PU and reference list
// the 8-tap (luma, quarter-pel) / 4-tap (chroma, eighth-pel) separable filters of ipol-generic.c:134-211, 681-758 on the clamped
// reference window, 14-bit intermediates; one list: clip((v + 32) >> 6), two lists: clip((v0 + v1 + 64) >> 7)
// (picture-generic.c:553-668).  The reference branches -- integer vectors are copied (inter.c:411-428, inter_cp_with_ext_border at
// the picture edge), fractional ones filtered, and kvz_bipred_average mixes pixel and 14-bit operands -- but a filter with the identity
// taps {0,0,0,64,0,0,0,0} gives 64 s exactly, so every branch is the same expression and one code path serves them all
// (tests/test_gpu_fme.py follows the reference's branches through the oracle).
// One workgroup per PU: window -> LDS, horizontal pass -> 14-bit LDS plane, vertical pass -> 14-bit accumulator, final rounding ->
// the PU's samples of the prediction picture; luma, then U and V.  Bytes per PU and list: (w + 7)(h + 7) + 2 (w/2 + 3)(h/2 + 3) read,
// 1.5 w h written.
template <int MAXN> __global__ void __launch_bounds__(256) dev_inter_pred_kernel(const u8 *ref0, const u8 *ref1, u8 *pred, const int W, const int H, const kvz_hip_mc_pu *pus,
                                                                                const Tables *tb)
{
  constexpr int WS = MAXN + 8;
  __shared__ u8 s_win[(MAXN + 7) * WS];
  __shared__ i16 s_g[(MAXN + 7) * MAXN];
  __shared__ i16 s_acc[MAXN * MAXN];
  const kvz_hip_mc_pu pu = pus[blockIdx.x];
  const int tid = threadIdx.x;
  if (pu.w > MAXN || pu.h > MAXN) return;
  const int lists = (pu.use[0] != 0) + (pu.use[1] != 0);
  for (int plane = 0; plane < 3; plane++) {
    const int sh = plane ? 1 : 0, fw = W >> sh, fh = H >> sh, w = pu.w >> sh, h = pu.h >> sh, taps = plane ? 4 : 8, before = plane ? 1 : 3;
    const long poff = plane == 0 ? 0 : (plane == 1 ? (long)W * H : (long)W * H * 5 / 4);
    bool first = true;
    for (int l = 0; l < 2; l++) {
      if (!pu.use[l]) continue;
      const u8 *src = (l ? ref1 : ref0) + poff;
      const int mvx = pu.mv[l][0], mvy = pu.mv[l][1];
      // luma: integer part mv >> 2, fraction mv & 3; chroma (half resolution): mv >> 3, mv & 7  (inter.c:66-67, 178-199)
      const int X0 = (pu.x >> sh) + (mvx >> (2 + sh)) - before, Y0 = (pu.y >> sh) + (mvy >> (2 + sh)) - before;
      const int8_t *hf = plane ? tb->chroma_filter[mvx & 7] : tb->luma_filter[mvx & 3], *vf = plane ? tb->chroma_filter[mvy & 7] : tb->luma_filter[mvy & 3];
      const int wr = h + taps - 1, wc = w + taps - 1;
      __syncthreads();
      for (int i = tid; i < wr * wc; i += 256) {
        const int r = i / wc, c = i - r * wc;
        s_win[r * WS + c] = src[(long)iclip(0, fh - 1, Y0 + r) * fw + iclip(0, fw - 1, X0 + c)];
      }
      __syncthreads();
      for (int i = tid; i < wr * w; i += 256) {
        const int r = i / w, c = i - r * w;
        int t = 0;
        for (int k = 0; k < taps; k++) t += hf[k] * (int)s_win[r * WS + c + k];
        s_g[r * MAXN + c] = (i16)t;
      }
      __syncthreads();
      for (int i = tid; i < w * h; i += 256) {
        const int y = i / w, x = i - y * w;
        int t = 0;
        for (int k = 0; k < taps; k++) t += vf[k] * (int)s_g[(y + k) * MAXN + x];
        const int v = (int)(i16)(t >> 6);  // the 14-bit sample (kvz_pixel_im)
        if (lists == 1) pred[poff + (long)((pu.y >> sh) + y) * fw + (pu.x >> sh) + x] = clip_pixel((v + 32) >> 6);
        else if (first) s_acc[i] = (i16)v;
        else pred[poff + (long)((pu.y >> sh) + y) * fw + (pu.x >> sh) + x] = clip_pixel(((int)s_acc[i] + v + 64) >> 7);
      }
      first = false;
    }
  }
}

// ---- the same for PUs up to 16x16, ONE WAVEFRONT per PU (four PUs per workgroup, no workgroup barrier) --------------------------------------------------
// A 16x16 PU is 1.2 KB of traffic: with a 256-lane workgroup and four barriers per plane the launch and barrier overheads were the cost (0.04 of the HBM roof).
// Here a wavefront owns its PU: the window goes to LDS as aligned dwords (byte by byte, clamped, only where it touches the picture edge), a lane filters runs of
// four adjacent outputs from four dwords -- the 8 (4) taps of the horizontal pass are two (one) v_dot4_i32_i8 on pixels biased by -128 (+ 128 * 64 afterwards:
// every filter sums to 64), the vertical pass on the 14-bit intermediates, kept TRANSPOSED in LDS so that a column's samples are adjacent, is v_dot2_i32_i16 --
// and the finished block leaves through LDS as one dword per lane.  Same integers as the kernel above, whatever the order of the sums.
typedef short kvz_short2 __attribute__((ext_vector_type(2)));
KVZ_DEV int dot4_i8(int a, int b, int c) { return __builtin_amdgcn_sdot4(a, b, c, false); }
KVZ_DEV int dot2_i16(unsigned a, unsigned b, int c)
{
  kvz_short2 va, vb;
  __builtin_memcpy(&va, &a, 4); __builtin_memcpy(&vb, &b, 4);
  return __builtin_amdgcn_sdot2(va, vb, c, false);
}
struct McWaveLds {
  alignas(16) u8 win[23 * 32];        // luma window: 23 rows of up to 28 bytes from a dword-aligned column; chroma: two windows of 11 rows x 16 bytes behind each other
  alignas(16) i16 gT[16 * 24];        // horizontal pass, transposed: [column][row], rows padded to 24; chroma: two planes of [8][12]
};

__global__ void __launch_bounds__(256) dev_inter_pred_wave_kernel(const u8 *ref0, const u8 *ref1, u8 *pred, const int W, const int H, const kvz_hip_mc_pu *pus, const int count,
                                                                  const Tables *tb)
{
  __shared__ McWaveLds lds_all[4];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, pi = blockIdx.x * 4 + wave;
  if (pi >= count) return;  // wavefront-uniform; nothing below synchronises across wavefronts
  McWaveLds &L = lds_all[wave];
  const kvz_hip_mc_pu pu = pus[pi];
  const int lists = (pu.use[0] != 0) + (pu.use[1] != 0);
  if (pu.w > 16 || pu.h > 16 || lists == 0) return;
  for (int pass = 0; pass < 2; pass++) {  // pass 0: luma; pass 1: U (lanes 0..31) and V (lanes 32..63) side by side
    const int sh = pass, fw = W >> sh, fh = H >> sh, w = pu.w >> sh, h = pu.h >> sh, taps = pass ? 4 : 8, before = pass ? 1 : 3;
    const int half = pass ? lane >> 5 : 0, hl = pass ? lane & 31 : lane, nl = pass ? 32 : 64;  // which plane of the pass, lane within it, lanes per plane
    const long poff = pass == 0 ? 0 : (half == 0 ? (long)W * H : (long)W * H * 5 / 4);
    const int wstride = pass ? 16 : 32, gstride = pass ? 12 : 24;
    u8 *win = L.win + (pass ? half * 11 * 16 : 0);
    i16 *gT = L.gT + (pass ? half * 8 * 12 : 0);
    int acc0[4] = { 0, 0, 0, 0 };  // the first list's 14-bit samples of this lane's run (two lists)
    bool first = true;
    for (int l = 0; l < 2; l++) {
      if (!pu.use[l]) continue;
      const u8 *src = (l ? ref1 : ref0) + poff;
      const int mvx = pu.mv[l][0], mvy = pu.mv[l][1];
      const int X0 = (pu.x >> sh) + (mvx >> (2 + sh)) - before, Y0 = (pu.y >> sh) + (mvy >> (2 + sh)) - before;
      const int8_t *hfp = pass ? tb->chroma_filter[mvx & 7] : tb->luma_filter[mvx & 3], *vfp = pass ? tb->chroma_filter[mvy & 7] : tb->luma_filter[mvy & 3];
      const int wr = h + taps - 1, wc = w + taps - 1;
      const int X0a = X0 & ~3, xo = X0 - X0a, ndw = (xo + wc + 3) >> 2;  // dword-aligned column the window rows are staged from
      __builtin_amdgcn_wave_barrier();
      if (X0a >= 0 && X0a + 4 * ndw <= fw && Y0 >= 0 && Y0 + wr <= fh && (fw & 3) == 0) {
        // eight dword slots per row (a row has at most seven): row and slot of a lane by shifts
        for (int i = hl; i < wr * 8; i += nl) {
          const int r = i >> 3, d = i & 7;
          if (d < ndw) *reinterpret_cast<unsigned *>(win + r * wstride + 4 * d) = *reinterpret_cast<const unsigned *>(src + (long)(Y0 + r) * fw + X0a + 4 * d);
        }
      } else {  // the window leaves the picture: clamped addressing = the edge-replicated extended block (ipol-generic.c:761-814)
        for (int i = hl; i < wr * 32; i += nl) {
          const int r = i >> 5, c = i & 31;
          if (c < wc) win[r * wstride + xo + c] = src[(long)iclip(0, fh - 1, Y0 + r) * fw + iclip(0, fw - 1, X0 + c)];
        }
      }
      __builtin_amdgcn_wave_barrier();
      // horizontal pass: runs of four adjacent outputs of a window row -> gT[column][row]
      {
        unsigned f0, f1 = 0;
        __builtin_memcpy(&f0, hfp, 4);
        if (!pass) __builtin_memcpy(&f1, hfp + 4, 4);
        const int lruns = w == 16 ? 2 : (w == 8 ? 1 : 0);  // log2 of the runs per row (w / 4)
        for (int t = hl; t < (wr << lruns); t += nl) {
          const int r = t >> lruns, c0 = (t & ((1 << lruns) - 1)) * 4, o = xo + c0, sh0 = o & 3;
          const unsigned *p = reinterpret_cast<const unsigned *>(win + r * wstride + (o & ~3));
          const unsigned d0 = p[0] ^ 0x80808080u, d1 = p[1] ^ 0x80808080u, d2 = p[2] ^ 0x80808080u, d3 = pass ? 0u : p[3] ^ 0x80808080u;
          // the string shifted to the run's first byte once (dynamic byte shift), the four outputs then at constant offsets
          const unsigned e0 = __builtin_amdgcn_alignbyte(d1, d0, sh0), e1 = __builtin_amdgcn_alignbyte(d2, d1, sh0), e2 = __builtin_amdgcn_alignbyte(d3, d2, sh0);
#pragma unroll
          for (int j = 0; j < 4; j++) {
            int g = dot4_i8((int)__builtin_amdgcn_alignbyte(e1, e0, j), (int)f0, 128 * 64);
            if (!pass) g = dot4_i8((int)__builtin_amdgcn_alignbyte(e2, e1, j), (int)f1, g);
            gT[(c0 + j) * gstride + r] = (i16)g;
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
      // vertical pass: runs of four outputs down a column, from the column's adjacent 14-bit samples
      {
        int vf[8];
        for (int k = 0; k < 8; k++) vf[k] = k < taps ? (int)vfp[k] : 0;
        unsigned vp[4];
        for (int k = 0; k < 4; k++) vp[k] = ((unsigned)vf[2 * k] & 0xffffu) | ((unsigned)vf[2 * k + 1] << 16);
        const int lruns = h == 16 ? 2 : (h == 8 ? 1 : 0), t = hl;  // runs per column = h / 4; w * runs <= 64 (32 per chroma plane): one task per lane
        if (t < (w << lruns)) {
          const int x = t >> lruns, y0 = (t & ((1 << lruns) - 1)) * 4;
          const i16 *col = gT + x * gstride + y0;  // y0 is a multiple of 4 and gstride is even: dword-aligned
          unsigned q[6];
#pragma unroll
          for (int k = 0; k < 6; k++) q[k] = reinterpret_cast<const unsigned *>(col)[k];  // samples y0 .. y0 + 11
          int v[4];
#pragma unroll
          for (int j = 0; j < 4; j++) {
            // samples y0 + j .. y0 + j + 7 as four pairs: aligned pairs for even j, shifted by one sample for odd j
            int a = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
              const int base = (j >> 1) + k;
              const unsigned pr = (j & 1) ? __builtin_amdgcn_alignbyte(q[base + 1 < 6 ? base + 1 : 5], q[base], 2) : q[base];
              a = dot2_i16(pr, vp[k], a);
            }
            v[j] = (int)(i16)(a >> 6);
          }
          if (lists == 1 || !first) {
            u8 o4[4];
#pragma unroll
            for (int j = 0; j < 4; j++) o4[j] = lists == 1 ? clip_pixel((v[j] + 32) >> 6) : clip_pixel((acc0[j] + v[j] + 64) >> 7);
            // the block leaves through LDS (the window is dead by now) so that a lane stores four bytes of a ROW
#pragma unroll
            for (int j = 0; j < 4; j++) win[(y0 + j) * w + x] = o4[j];
          } else {
#pragma unroll
            for (int j = 0; j < 4; j++) acc0[j] = v[j];
          }
        }
      }
      first = false;
    }
    __builtin_amdgcn_wave_barrier();
    for (int i = hl; i < (w * h) >> 2; i += nl) {
      const int lw = w == 16 ? 4 : (w == 8 ? 3 : 2), y = (4 * i) >> lw, x = (4 * i) & (w - 1);
      *reinterpret_cast<unsigned *>(pred + poff + (long)((pu.y >> sh) + y) * fw + (pu.x >> sh) + x) = *reinterpret_cast<const unsigned *>(win + 4 * i);
    }
  }
}

}  // namespace kvz
