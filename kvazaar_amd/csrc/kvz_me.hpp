// kvz_me.hpp -- the motion search of a prediction unit, whole and on the device: search_pu_inter_ref (search_inter.c:1237-1435) followed by the fractional
// refinement of its result (search_inter.c:1866-1917 -> search_frac :974-1130), with every decision of the reference on the way (include/kvz_hip_dev.h
// kvz_hip_dev_pu_search states the contract; oracle/kvz_oracle_inter.inc me_integer / me_fractional is the CPU restatement it is checked against).
//
// One workgroup of four wavefronts per PU.  The search is a chain of ROUNDS -- the starting points (up to 7 probes), the two rounds of the early-termination
// cross (4 + 3), the hexagon's first ring (6), its steps (3 each) and the final square (8) -- and inside a round the probes' SADs do not depend on one
// another: only the reference's accept / reject sequence (check_mv_cost, search_inter.c:180-232: each probe against the best so far, the MVD bits only when
// the SAD alone does not rule it out) is order-dependent.  So a round computes all its SADs in parallel -- one wavefront per probe, a run of w h / 64 samples
// per lane, v_sad_u8 on dwords, a DPP wave sum --, then every thread replays the decisions on those numbers, uniformly, so that no broadcast is needed and a
// round costs one barrier.  After the starting points a (w + 16)^2 search window around the best of them is staged in LDS (clamped addressing = image.c:279-397's
// edge replication) and follows the search: it is reloaded around the current centre when a probe of the coming round would leave it; probes outside it (the
// starting points themselves) read the reference picture through L2.  The source block sits in LDS.  The fractional part is the
// fused pipeline of kvz_fme.hpp (window -> shared 14-bit horizontal intermediates -> four planes per step -> 8x8 Hadamard), followed by search_frac's own
// bookkeeping including its `unsigned` cost accumulator.
// Algorithmic bytes per PU: w h source + ~30 probes x w h reference samples (L2 hits after the first touch) + the (w + 8)^2 window; 64 B result.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/kvz_hip_dev.h"
#include "kvz_ops.hpp"
#include "kvz_tables.hpp"

namespace kvz {

struct MeState {  // the best so far (check_mv_cost's out-parameters), identical in every thread
  int mvx, mvy;
  double cost, bits;
};

KVZ_DEV unsigned me_golomb_bits(unsigned s)  // search_inter.c:234-249
{
  unsigned bins = 0;
  if (s >= 1u << 8) { bins += 16; s >>= 8; }
  if (s >= 1u << 4) { bins += 8; s >>= 4; }
  if (s >= 1u << 2) { bins += 4; s >>= 2; }
  if (s >= 1u << 1) bins += 2;
  return bins;
}
KVZ_DEV int me_mvd_bits(int dx, int dy)  // search_inter.c:328-341 get_mvd_coding_cost, whole bits
{
  const unsigned ax = (unsigned)(dx < 0 ? -dx : dx), ay = (unsigned)(dy < 0 ? -dy : dy);
  return (int)(4 + (ax == 1) + (ay == 1) + me_golomb_bits(ax) + me_golomb_bits(ay));
}

template <int MAXN> __global__ void __launch_bounds__(256) dev_pu_search_kernel(const u8 *cur, const u8 *ref, const int W, const int H, const kvz_hip_me_pu *pus,
                                                                               const kvz_hip_me_params prm, const Tables *tb, kvz_hip_me_result *out)
{
  constexpr int WS = MAXN + 8;
  __shared__ alignas(8) u8 s_win[(MAXN + 8) * WS];
  __shared__ i16 s_g[(MAXN + 8) * (MAXN + 1)];
  __shared__ alignas(8) u8 s_cur[MAXN * MAXN];
  __shared__ alignas(8) u8 s_pred[4][MAXN * MAXN];
  __shared__ u32 s_cost[4];
  __shared__ u32 s_sad[2][8];
  const kvz_hip_me_pu pu = pus[blockIdx.x];
  const int w = pu.w, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wh = w * w, l2w = 31 - __builtin_clz((unsigned)w);
  if (w > MAXN || pu.h != w) return;  // routed to a larger instantiation by the host; only square PUs
  for (int i = tid; i < wh; i += 256) s_cur[i] = cur[(long)(pu.y + (i >> l2w)) * W + pu.x + (i & (w - 1))];

  // fracmv_within_tile (search_inter.c:75-152), mv-constraint none
  auto allowed = [&](int x, int y) -> bool {
    if (!prm.mv_constraint) return true;
    const bool frac_luma = (x % 4 != 0) || (y % 4 != 0), frac_chroma = (x % 8 != 0) || (y % 8 != 0);
    int margin = frac_luma ? 4 : (frac_chroma ? 2 : 0);
    if (prm.sao) margin += 10; else if (prm.deblock) margin += 8;
    const int lx = ((pu.x + w + margin) * 4 + x) / 256 - pu.x / 64, ly = ((pu.y + w + margin) * 4 + y) / 256 - pu.y / 64;
    return !(ly > 1) && !(lx + ly > 2);
  };
  // calc_mvd_cost with no merge candidates = the cheaper predictor's MVD bits (search_inter.c:343-423); x, y in quarter samples
  auto mvd_bits = [&](int x, int y) -> int {
    const int c1 = me_mvd_bits(x - pu.mv_cand[0][0], y - pu.mv_cand[0][1]);
    const bool same = pu.mv_cand[0][0] == pu.mv_cand[1][0] && pu.mv_cand[0][1] == pu.mv_cand[1][1];
    const int c2 = same ? c1 : me_mvd_bits(x - pu.mv_cand[1][0], y - pu.mv_cand[1][1]);
    return c1 < c2 ? c1 : c2;
  };
  auto mvp_index = [&](int x, int y) -> int {  // select_mv_cand without cost_out
    if (pu.mv_cand[0][0] == pu.mv_cand[1][0] && pu.mv_cand[0][1] == pu.mv_cand[1][1]) return 0;
    return me_mvd_bits(x - pu.mv_cand[1][0], y - pu.mv_cand[1][1]) < me_mvd_bits(x - pu.mv_cand[0][0], y - pu.mv_cand[0][1]) ? 1 : 0;
  };

  // One round: the SADs of n <= 8 integer displacements (px[k], py[k]) into s_sad[buf][0..n).  One wavefront per probe (probes wave, wave + 4), every lane a run
  // of w h / 64 samples; reference samples from the search window in LDS (wx0, wy0: its origin in the picture; (w + 16)^2 samples around the search centre, aliased
  // on the prediction planes the fractional part uses later) when the probe lies inside it, through L2 with clamped addressing otherwise.  s_sad is double
  // buffered, so a round costs one barrier.
  constexpr int R = 8, WW = MAXN + 2 * R;
  u8 *s_sw = &s_pred[0][0];
  int px[8], py[8];
  int wx0 = 0, wy0 = 0, buf = 0;
  bool have_window = false;
  const int per = wh >> 6, run = lane * per, ryy = run >> l2w, rxx = run & (w - 1), ww = w + 2 * R;
  auto load_window = [&](int mx, int my) {  // centred on the integer displacement (mx, my)
    wx0 = pu.x + mx - R; wy0 = pu.y + my - R;
    __syncthreads();  // earlier probes are done with the old window
    for (int i = tid; i < ww * ww; i += 256) {
      const int r = i / ww, c = i - r * ww;
      s_sw[r * WW + c] = ref[(long)iclip(0, H - 1, wy0 + r) * W + iclip(0, W - 1, wx0 + c)];
    }
    have_window = true;
    __syncthreads();
  };
  auto probe = [&](int n) {
    buf ^= 1;
#pragma unroll
    for (int k0 = 0; k0 < 8; k0 += 4) {
      const int k = k0 + wave;
      if (k < n) {
        int kx = 0, ky = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) if (j == wave) { kx = px[k0 + j]; ky = py[k0 + j]; }
        const int X = pu.x + kx, Y = pu.y + ky;
        u32 part = 0;
        if (have_window && X >= wx0 && X + w <= wx0 + ww && Y >= wy0 && Y + w <= wy0 + ww) {
          const u8 *p = s_sw + (Y - wy0 + ryy) * WW + (X - wx0 + rxx);
          if (per == 1) { const int c = s_cur[run], r = p[0]; part = (u32)(c > r ? c - r : r - c); }
          else {
            const int sh = (X - wx0) & 3;  // rxx is a multiple of four and rows start aligned: the misalignment is the same for every lane
            const u32 *q = reinterpret_cast<const u32 *>(p - sh);
            u32 lo = q[0];
            for (int j = 0; j < per / 4; j++) {
              const u32 hi = q[j + 1];
              const u32 r4 = sh == 0 ? lo : (u32)(((unsigned long long)hi << 32 | lo) >> (8 * sh));
              part = __builtin_amdgcn_sad_u8(*reinterpret_cast<const u32 *>(&s_cur[run + 4 * j]), r4, part);
              lo = hi;
            }
          }
        } else if (per == 1) {
          const int c = s_cur[run], r = ref[(long)iclip(0, H - 1, Y + ryy) * W + iclip(0, W - 1, X + rxx)];
          part = (u32)(c > r ? c - r : r - c);
        } else if (X >= 0 && X + w <= W && Y >= 0 && Y + w <= H) {
          const u8 *row = ref + (long)(Y + ryy) * W + X + rxx;
          for (int j = 0; j < per / 4; j++) { u32 r4; __builtin_memcpy(&r4, row + 4 * j, 4); part = __builtin_amdgcn_sad_u8(*reinterpret_cast<const u32 *>(&s_cur[run + 4 * j]), r4, part); }
        } else {
          const u8 *row = ref + (long)iclip(0, H - 1, Y + ryy) * W;
          for (int j = 0; j < per; j++) { const int c = s_cur[run + j], r = row[iclip(0, W - 1, X + rxx + j)]; part += (u32)(c > r ? c - r : r - c); }
        }
        int x = (int)part;  // wave64 sum: row_shr 8 / 4 / 2 / 1 inside rows of 16 lanes, then the four row totals
        x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, true);
        x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, true);
        x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, true);
        x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, true);
        const u32 total = (u32)(__builtin_amdgcn_readlane(x, 15) + __builtin_amdgcn_readlane(x, 31) + __builtin_amdgcn_readlane(x, 47) + __builtin_amdgcn_readlane(x, 63));
        if (lane == 0) s_sad[buf][k] = total;
      }
    }
    __syncthreads();
  };
  // the window follows the search: reloaded around (mx, my) when a probe of the coming round would leave it
  auto keep_window = [&](int n, int mx, int my) {
    bool inside = have_window;
#pragma unroll
    for (int j = 0; j < 8; j++)
      if (j < n) inside = inside && pu.x + px[j] >= wx0 && pu.x + px[j] + w <= wx0 + ww && pu.y + py[j] >= wy0 && pu.y + py[j] + w <= wy0 + ww;
    if (!inside) load_window(mx, my);
  };

  MeState best;
  best.mvx = 0; best.mvy = 0; best.cost = 1.7e+308; best.bits = 2147483647.0;
  // check_mv_cost on the SAD of probe k (search_inter.c:180-232)
  auto consider = [&](int k, int x, int y) -> bool {
    if (!allowed(x * 4, y * 4)) return false;
    double cost = (double)s_sad[buf][k];
    if (cost + 0.001 >= best.cost) return false;
    const double bits = (double)mvd_bits(x * 4, y * 4);
    cost += bits * prm.lambda_sqrt;
    if (cost + 0.001 >= best.cost) return false;
    best.mvx = x * 4; best.mvy = y * 4; best.cost = cost; best.bits = bits;
    return true;
  };

  // ---- select_starting_point (search_inter.c:285-312) on the co-located motion when it is allowed, else (0, 0) ----
  int ex = 0, ey = 0;
  if (pu.has_start && allowed(pu.start_mv[0], pu.start_mv[1])) { ex = pu.start_mv[0]; ey = pu.start_mv[1]; }
  best.mvx = ex; best.mvy = ey;
  ex >>= 2; ey >>= 2;
  {
    int n = 0;
    auto push = [&](int x, int y) {
#pragma unroll
      for (int j = 0; j < 8; j++) if (j == n) { px[j] = x; py[j] = y; }
      n++;
    };
#pragma unroll
    for (int j = 0; j < 8; j++) { px[j] = 0; py[j] = 0; }
    push(0, 0);
    bool extra = ex != 0 || ey != 0;
    if (extra)
      for (int i = 0; i < pu.num_merge; i++)
        if (pu.merge_dir[i] != 3 && ((pu.merge_mv[i][0] + 2) >> 2) == ex && ((pu.merge_mv[i][1] + 2) >> 2) == ey) { extra = false; break; }
    if (extra) push(ex, ey);
    for (int i = 0; i < pu.num_merge; i++) {
      if (pu.merge_dir[i] == 3) continue;
      const int x = (pu.merge_mv[i][0] + 2) >> 2, y = (pu.merge_mv[i][1] + 2) >> 2;
      if (x == 0 && y == 0) continue;
      push(x, y);
    }
    __syncthreads();  // s_cur is complete
    probe(n);
#pragma unroll
    for (int k = 0; k < 8; k++) if (k < n) consider(k, px[k], py[k]);
    load_window(best.mvx >> 2, best.mvy >> 2);
  }

  // ---- early_terminate (search_inter.c:425-486), me-early-termination sensitive ----
  bool skip_me = false;
  {
    const int hx[7] = { 0, -1, 0, 1, 0, -1, 0 }, hy[7] = { -1, 0, 1, 0, -1, 0, 0 };
    int mx = best.mvx >> 2, my = best.mvy >> 2, first = 0, last = 3;
    for (int k = 0; k < 2; k++) {
      const double threshold = best.cost * 0.95;
      int best_index = 6;
#pragma unroll
      for (int j = 0; j < 4; j++) { const int i = first + j <= last ? first + j : 6; px[j] = mx + hx[i]; py[j] = my + hy[i]; }
      for (int j = 4; j < 8; j++) { px[j] = 0; py[j] = 0; }
      keep_window(last - first + 1, mx, my);
      probe(last - first + 1);
#pragma unroll
      for (int j = 0; j < 4; j++) if (first + j <= last && consider(j, px[j], py[j])) best_index = first + j;
      mx += hx[best_index]; my += hy[best_index];
      if (best.cost >= threshold) { skip_me = true; break; }
      first = (best_index + 3) % 4;
      last = first + 2;
    }
  }

  // ---- hexagon_search (search_inter.c:712-800), unlimited steps ----
  if (!skip_me) {
    const int lx[9] = { 0, 1, 2, 1, -1, -2, -1, 1, 2 }, ly[9] = { 0, -2, 0, 2, 2, 0, -2, -2, 0 };
    const int sx[9] = { 0, 0, -1, 1, 0, -1, 1, -1, 1 }, sy[9] = { 0, -1, 0, 0, 1, -1, -1, 1, 1 };
    int mx = best.mvx >> 2, my = best.mvy >> 2, best_index = 0;
#pragma unroll
    for (int j = 0; j < 6; j++) { px[j] = mx + lx[j + 1]; py[j] = my + ly[j + 1]; }
    px[6] = px[7] = 0; py[6] = py[7] = 0;
    keep_window(6, mx, my);
    probe(6);
#pragma unroll
    for (int j = 0; j < 6; j++) if (consider(j, px[j], py[j])) best_index = j + 1;
    while (best_index != 0) {
      const int start = best_index == 1 ? 6 : (best_index == 8 ? 1 : best_index - 1);
      mx += lx[best_index]; my += ly[best_index];
      best_index = 0;
#pragma unroll
      for (int j = 0; j < 3; j++) { px[j] = mx + lx[start + j]; py[j] = my + ly[start + j]; }
      keep_window(3, mx, my);
      probe(3);
#pragma unroll
      for (int j = 0; j < 3; j++) if (consider(j, px[j], py[j])) best_index = start + j;
    }
#pragma unroll
    for (int j = 0; j < 8; j++) { px[j] = mx + sx[j + 1]; py[j] = my + sy[j + 1]; }
    keep_window(8, mx, my);
    probe(8);
#pragma unroll
    for (int j = 0; j < 8; j++) consider(j, px[j], py[j]);
  }

  // ---- the fused fractional pipeline of kvz_fme.hpp around the integer result ----
  const int imx = best.mvx >> 2, imy = best.mvy >> 2;
  const int X0 = pu.x + imx - 4, Y0 = pu.y + imy - 4;
  const int tiles = (w >> 3) * (w >> 3), tw = w >> 3;
  auto score = [&](int planes) {  // SATD of s_pred[0..planes) against s_cur: one lane per (plane, 8x8 tile)
    __syncthreads();
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
  const bool found = best.cost < 1.7e+308;
  const bool need_window = found && (prm.fme_level == 0 || (allowed(best.mvx, best.mvy) && (allowed(best.mvx + 3, best.mvy + 3) || allowed(best.mvx - 3, best.mvy - 3))));
  if (need_window) {
    __syncthreads();
    for (int i = tid; i < (w + 8) * (w + 8); i += 256) {
      const int r = i / (w + 8), c = i - r * (w + 8);
      s_win[r * WS + c] = ref[(long)iclip(0, H - 1, Y0 + r) * W + iclip(0, W - 1, X0 + c)];
    }
    __syncthreads();
    for (int i = tid; i < wh; i += 256) s_pred[0][i] = s_win[((i >> l2w) + 4) * WS + (i & (w - 1)) + 4];
    score(1);
  }
  if (found && prm.fme_level == 0) {  // search_inter.c:1381-1393: the result re-priced with SATD
    best.cost = (double)s_cost[0];
    best.cost += best.bits * prm.lambda_sqrt;
  }
  kvz_hip_me_result res;
  res.mv[0] = best.mvx; res.mv[1] = best.mvy; res.cost = best.cost; res.bits = best.bits;
  res.mvp = mvp_index(best.mvx, best.mvy);
  res.valid = allowed(best.mvx, best.mvy) && found;
  res.frac_mv[0] = res.frac_mv[1] = 0; res.frac_mvp = 0; res.frac_valid = 0; res.frac_cost = 0; res.frac_bits = 0;

  if (prm.fme_level > 0 && res.valid && (allowed(best.mvx + 3, best.mvy + 3) || allowed(best.mvx - 3, best.mvy - 3))) {
    // search_frac (search_inter.c:974-1166): the two half-sample steps, with fme_level 4 the two quarter-sample steps around the best half-sample position as well;
    // `costs` is unsigned there, the sums with the motion cost truncate
    const int sqx[9] = { 0, -1, 1, 0, 0, -1, 1, -1, 1 }, sqy[9] = { 0, 0, 0, -1, 1, -1, -1, 1, 1 };
    int mx = imx, my = imy;
    double bitcost = (double)mvd_bits(mx * 4, my * 4);
    u32 c0 = (u32)((double)s_cost[0] + bitcost * prm.lambda_sqrt);
    double cost = (double)c0;
    mx *= 2; my *= 2;
    int best_index = 0, i0 = 1, off_x = 0, off_y = 0;
    const int steps = prm.fme_level;
    for (int step = 0; step < steps; step++) {
      const int unit = step < 2 ? 2 : 1;  // quarter samples per step of this precision
      FmePlane pl[4];
      fme_planes(step, off_x, off_y, pl);
      int done = 0;
      for (int first = 0; first < 4; first++) {
        if (done & (1 << first)) continue;
        const int hf = pl[first].hf;
        const int8_t *f = tb->luma_filter[hf];
        __syncthreads();
        for (int i = tid; i < (w + 8) * (w + 1); i += 256) {
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
          for (int i = tid; i < wh; i += 256) {
            const int y = i >> l2w, x = i & (w - 1);
            int t = 0;
            for (int j = 0; j < 8; j++) t += vf[j] * (int)s_g[(y + ro + j) * (MAXN + 1) + x + co];
            s_pred[p][i] = fin14(t >> 6);
          }
        }
      }
      score(4);
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int cx = mx + sqx[i0 + j], cy = my + sqy[i0 + j];
        if (!allowed(cx * unit, cy * unit)) continue;
        const double b = (double)mvd_bits(cx * unit, cy * unit);
        const u32 cj = (u32)((double)s_cost[j] + b * prm.lambda_sqrt);
        if ((double)cj < cost) { cost = (double)cj; bitcost = b; best_index = i0 + j; }
      }
      i0 += 4;
      if (step == 1 || step == steps - 1) {  // search_inter.c:1146-1162
        mx += sqx[best_index]; my += sqy[best_index];
        if (step == (steps - 1 < 1 ? steps - 1 : 1)) { mx *= 2; my *= 2; off_x = sqx[best_index]; off_y = sqy[best_index]; best_index = 0; i0 = 1; }
      }
    }
    res.frac_mv[0] = mx; res.frac_mv[1] = my; res.frac_cost = cost; res.frac_bits = bitcost;
    res.frac_mvp = mvp_index(mx, my);
    res.frac_valid = allowed(mx, my);
  }
  if (tid == 0) out[blockIdx.x] = res;
}

}  // namespace kvz
