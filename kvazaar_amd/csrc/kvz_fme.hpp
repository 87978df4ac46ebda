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

}  // namespace kvz
