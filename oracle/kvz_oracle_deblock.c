/*
 * kvz_oracle_deblock.c -- TEST INFRASTRUCTURE (see kvz_oracle.h): CPU restatement of kvazaar's deblocking filter for the
 * all-intra, constant-QP configuration of the batched pass.
 *
 * Reference: filter.c:783 kvz_filter_deblock_lcu and everything it calls.  kvazaar walks LCU by LCU (vertical edges of the
 * LCU, the 4 deferred rightmost pixels of the previous LCU's horizontal edges, then the horizontal edges); because edges
 * of one direction lie 8 pixels apart and a filter reads 4 / writes at most 3 pixels on either side, that order equals
 * H.265 8.7.2's picture-level order -- every vertical edge first, then every horizontal edge on the result -- which is what
 * this file (and the device kernel) does.  tests/test_oracle_vs_ref.py checks it against the compiled reference.
 *
 * Restrictions taken from the configuration: every CU is intra 2Nx2N (boundary strength 2 on every filtered edge,
 * filter.c:418-421, 622), one QP for the picture (filter.c:277-279), 8-bit, 4:2:0, no PCM / lossless.
 */
#include <stdlib.h>

#include "kvz_oracle.h"

/* H.265 Table 8-12 as kvazaar stores it (filter.c:46-65): tc' as run lengths, beta' as two ramps */
static int tc_prime(int q)
{
  static const unsigned char run_end[] = { 18, 27, 31, 35, 38, 40, 42, 43, 44, 45, 46 };  /* first Q with the next value: 0,1,2,...,10 */
  static const unsigned char tail[] = { 11, 13, 14, 16, 18, 20, 22, 24 };               /* Q = 46..53 */
  if (q >= 46) return tail[q - 46];
  int v = 0;
  while (q >= run_end[v]) v++;
  return v;
}
static int beta_prime(int q) { return q < 16 ? 0 : (q <= 28 ? q - 10 : 2 * q - 38); }

static int clip3(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }

static const unsigned char chroma_qp[58] = { /* H.265 Table 8-10, transform.c:56-62 */
  0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 29, 30, 31, 32,
  33, 33, 34, 34, 35, 35, 36, 36, 37, 37, 38, 39, 40, 41, 42, 43, 44, 45, 46, 47, 48, 49, 50, 51 };

/* filter.c:202-216 is_tu_boundary (is_pu_boundary adds nothing for 2Nx2N: CU edges are TU edges) for the 8x8 unit at (x, y) */
static int edge_is_filtered(const uint8_t *cu_depth, int w8, int x, int y, int vertical)
{
  const int d = cu_depth[(y >> 3) * w8 + (x >> 3)], tr_depth = d ? d : 1, tu_w = 64 >> tr_depth;
  return ((vertical ? x : y) & (tu_w - 1)) == 0;
}

/* One 4-sample part of a luma edge (filter.c:386-561): p[i][k] = sample k (0..7, edge between 3 and 4) of line i, in place */
static void luma_part(uint8_t *px, int step_across, int step_along, int beta, int tc)
{
  int b[4][8];
  for (int i = 0; i < 4; i++) for (int k = 0; k < 8; k++) b[i][k] = px[i * step_along + (k - 4) * step_across];
  const int dp0 = abs(b[0][1] - 2 * b[0][2] + b[0][3]), dq0 = abs(b[0][4] - 2 * b[0][5] + b[0][6]);
  const int dp3 = abs(b[3][1] - 2 * b[3][2] + b[3][3]), dq3 = abs(b[3][4] - 2 * b[3][5] + b[3][6]);
  const int dp = dp0 + dp3, dq = dq0 + dq3;
  if (dp + dq >= beta) return;
  const int strong = 2 * (dp0 + dq0) < (beta >> 2) && 2 * (dp3 + dq3) < (beta >> 2) &&
                     abs(b[0][3] - b[0][4]) < ((5 * tc + 1) >> 1) && abs(b[3][3] - b[3][4]) < ((5 * tc + 1) >> 1) &&
                     abs(b[0][0] - b[0][3]) + abs(b[0][4] - b[0][7]) < (beta >> 3) && abs(b[3][0] - b[3][3]) + abs(b[3][4] - b[3][7]) < (beta >> 3);
  const int side_threshold = (beta + (beta >> 1)) >> 3;
  for (int i = 0; i < 4; i++) {
    const int m0 = b[i][0], m1 = b[i][1], m2 = b[i][2], m3 = b[i][3], m4 = b[i][4], m5 = b[i][5], m6 = b[i][6], m7 = b[i][7];
    int o[8] = { m0, m1, m2, m3, m4, m5, m6, m7 };
    if (strong) {  /* filter.c:95-118 */
      o[1] = clip3(m1 - 2 * tc, m1 + 2 * tc, (2 * m0 + 3 * m1 + m2 + m3 + m4 + 4) >> 3);
      o[2] = clip3(m2 - 2 * tc, m2 + 2 * tc, (m1 + m2 + m3 + m4 + 2) >> 2);
      o[3] = clip3(m3 - 2 * tc, m3 + 2 * tc, (m1 + 2 * m2 + 2 * m3 + 2 * m4 + m5 + 4) >> 3);
      o[4] = clip3(m4 - 2 * tc, m4 + 2 * tc, (m2 + 2 * m3 + 2 * m4 + 2 * m5 + m6 + 4) >> 3);
      o[5] = clip3(m5 - 2 * tc, m5 + 2 * tc, (m3 + m4 + m5 + m6 + 2) >> 2);
      o[6] = clip3(m6 - 2 * tc, m6 + 2 * tc, (m3 + m4 + m5 + 3 * m6 + 2 * m7 + 4) >> 3);
    } else {       /* filter.c:126-165 */
      int delta = (9 * (m4 - m3) - 3 * (m5 - m2) + 8) >> 4;
      if (abs(delta) < tc * 10) {
        const int tc2 = tc >> 1;
        delta = clip3(-tc, tc, delta);
        o[3] = clip3(0, 255, m3 + delta);
        o[4] = clip3(0, 255, m4 - delta);
        if (dp < side_threshold) o[2] = clip3(0, 255, m2 + clip3(-tc2, tc2, (((m1 + m3 + 1) >> 1) - m2 + delta) >> 1));
        if (dq < side_threshold) o[5] = clip3(0, 255, m5 + clip3(-tc2, tc2, (((m6 + m4 + 1) >> 1) - m5 - delta) >> 1));
      }
    }
    for (int k = 1; k < 7; k++) px[i * step_along + (k - 4) * step_across] = (uint8_t)o[k];
  }
}

/* One 4-sample part of a chroma edge (filter.c:170-190, 622-629) */
static void chroma_part(uint8_t *px, int step_across, int step_along, int tc)
{
  for (int i = 0; i < 4; i++) {
    uint8_t *s = px + i * step_along;
    const int m2 = s[-2 * step_across], m3 = s[-step_across], m4 = s[0], m5 = s[step_across];
    const int delta = clip3(-tc, tc, (((m4 - m3) * 4) + m2 - m5 + 4) >> 3);
    s[-step_across] = (uint8_t)clip3(0, 255, m3 + delta);
    s[0] = (uint8_t)clip3(0, 255, m4 - delta);
  }
}

void kvz_oracle_deblock_frame(int width, int height, int qp, int beta_offset_div2, int tc_offset_div2, uint8_t *y, uint8_t *u, uint8_t *v,
                              const uint8_t *cu_depth)
{
  kvz_oracle_deblock_frame_passes(width, height, qp, beta_offset_div2, tc_offset_div2, y, u, v, cu_depth, 3);
}

/* passes: 1 = every vertical edge, 2 = every horizontal edge, 3 = both in that order (the intermediate pictures the device's SAO
 * statistics are assembled from, kvazaar_amd/csrc/kvz_sao.hpp) */
void kvz_oracle_deblock_frame_passes(int width, int height, int qp, int beta_offset_div2, int tc_offset_div2, uint8_t *y, uint8_t *u, uint8_t *v,
                                     const uint8_t *cu_depth, int passes)
{
  const int w8 = width >> 3, cw = width >> 1, ch = height >> 1;
  const int beta = beta_prime(clip3(0, 51, qp + (beta_offset_div2 << 1)));
  const int tc = tc_prime(clip3(0, 53, qp + 2 * (2 - 1) + (tc_offset_div2 << 1)));                       /* filter.c:496-497, strength 2 */
  const int tc_c = tc_prime(clip3(0, 53, chroma_qp[qp] + 2 * (2 - 1) + (tc_offset_div2 << 1)));        /* filter.c:592-595 */
  for (int dir = 0; dir < 2; dir++) {  /* 0: vertical edges (filtering across x), 1: horizontal edges */
    const int vertical = dir == 0;
    if (!(passes & (1 << dir))) continue;
    for (int ey = 0; ey < height; ey += 8)
      for (int ex = 0; ex < width; ex += 8) {
        if ((vertical ? ex : ey) == 0) continue;                        /* picture border, filter.c:648-649 */
        if (!edge_is_filtered(cu_depth, w8, ex, ey, vertical)) continue;
        for (int part = 0; part < 2; part++) {                          /* 8 samples of edge = two 4-sample parts */
          uint8_t *p = y + (ey + (vertical ? 4 * part : 0)) * width + ex + (vertical ? 0 : 4 * part);
          luma_part(p, vertical ? 1 : width, vertical ? width : 1, beta, tc);
        }
        /* chroma: only edges on the 8x8 chroma grid (filter.c:680), 4 chroma samples per 8 luma samples of edge */
        const int xc = ex >> 1, yc = ey >> 1;
        if (((vertical ? xc : yc) & 7) == 0) {
          chroma_part(u + yc * cw + xc, vertical ? 1 : cw, vertical ? cw : 1, tc_c);
          chroma_part(v + yc * cw + xc, vertical ? 1 : cw, vertical ? cw : 1, tc_c);
        }
      }
  }
  (void)ch;
}

/* ---- the reference's own order: one LCU at a time (filter.c:783 kvz_filter_deblock_lcu) ---------------------------------
 * Needed where the INTERMEDIATE picture matters: kvazaar searches an LCU's SAO parameters right after deblocking that LCU
 * (encoderstate.c:669-682), i.e. on a picture whose right / lower neighbours have not been deblocked yet (oracle/kvz_oracle_sao.c).
 * Statement for statement: vertical edges of the LCU (filter.c:699-714), the deferred rightmost 4 samples of the horizontal edges of the
 * LCU to the left (filter.c:725-757), the horizontal edges of the LCU without their rightmost 4 samples unless the LCU ends the picture
 * (filter.c:648-683). */
typedef struct {
  int width, height, beta, tc, tc_c; uint8_t *y, *u, *v; const uint8_t *cu_depth;
  /* pictures with inter CUs (kvz_oracle_deblock_lcu_inter): per-part boundary strengths from `info` instead of the constant 2 */
  const kvz_hip_cu_dbk *info; int slice_b, qp, tc_offset_div2;
} dbk_t;
static const kvz_hip_cu_dbk *unit_at(const kvz_hip_cu_dbk *info, int width, int x, int y);
static int inter_edge_on(const kvz_hip_cu_dbk *info, int width, int x, int y, int vertical, int *tu_boundary);
static int inter_strength(const kvz_hip_cu_dbk *q, const kvz_hip_cu_dbk *p, int tu_boundary, int slice_b);
static int unit_edge(const dbk_t *d, int x, int y, int vertical, int *tu_boundary)
{
  if (d->info) return inter_edge_on(d->info, d->width, x, y, vertical, tu_boundary);
  *tu_boundary = 1;
  return edge_is_filtered(d->cu_depth, d->width >> 3, x, y, vertical);
}

static void edge_luma(const dbk_t *d, int x, int y, int length, int vertical, int tu_boundary)   /* filter.c:386-561 on `length` samples */
{
  for (int part = 0; part < length / 4; part++) {
    const int px = x + (vertical ? 0 : 4 * part), py = y + (vertical ? 4 * part : 0);
    int tc = d->tc;
    if (d->info) {  /* filter.c:405-497: the strength of this 4-sample part */
      const int s = inter_strength(unit_at(d->info, d->width, px, py), unit_at(d->info, d->width, vertical ? px - 1 : px, vertical ? py : py - 1), tu_boundary, d->slice_b);
      if (!s) continue;
      tc = tc_prime(clip3(0, 53, d->qp + 2 * (s - 1) + (d->tc_offset_div2 << 1)));
    }
    uint8_t *p = d->y + py * d->width + px;
    luma_part(p, vertical ? 1 : d->width, vertical ? d->width : 1, d->beta, tc);
  }
}
static void edge_chroma(const dbk_t *d, int xc, int yc, int length, int vertical)  /* filter.c:567-624 */
{
  const int cw = d->width >> 1;
  for (int part = 0; part < length / 4; part++) {
    const int off = (yc + (vertical ? 4 * part : 0)) * cw + xc + (vertical ? 0 : 4 * part);
    if (d->info) {  /* filter.c:610: chroma only at strength 2, i.e. next to an intra CU */
      const int lx = 2 * (xc + (vertical ? 0 : 4 * part)), ly = 2 * (yc + (vertical ? 4 * part : 0));
      if (unit_at(d->info, d->width, lx, ly)->type != 1 && unit_at(d->info, d->width, vertical ? lx - 1 : lx, vertical ? ly : ly - 1)->type != 1) continue;
    }
    chroma_part(d->u + off, vertical ? 1 : cw, vertical ? cw : 1, d->tc_c);
    chroma_part(d->v + off, vertical ? 1 : cw, vertical ? cw : 1, d->tc_c);
  }
}
static void deblock_unit(const dbk_t *d, int x, int y, int vertical, int tu_boundary)  /* filter.c:638-683 with width = height = 8 */
{
  if (x == 0 && vertical) return;
  if (y == 0 && !vertical) return;
  int length = 8, length_c = 4;
  if (!vertical) {
    const int x_right = x + 8;
    if (x_right % 64 == 0 && x_right != d->width) { length = 4; length_c = 0; }  /* deferred to the next LCU */
  }
  edge_luma(d, x, y, length, vertical, tu_boundary);
  const int xc = x >> 1, yc = y >> 1;
  if (((vertical ? xc : yc) & 7) == 0) edge_chroma(d, xc, yc, length_c, vertical);
}
static void deblock_lcu_inside(const dbk_t *d, int x, int y, int vertical)  /* filter.c:699-714 */
{
  const int end_x = x + 64 < d->width ? x + 64 : d->width, end_y = y + 64 < d->height ? y + 64 : d->height;
  for (int ey = y; ey < end_y; ey += 8)
    for (int ex = x; ex < end_x; ex += 8) {
      int tu_boundary;
      if (unit_edge(d, ex, ey, vertical, &tu_boundary)) deblock_unit(d, ex, ey, vertical, tu_boundary);
    }
}
static void deblock_lcu_rightmost(const dbk_t *d, int x_px, int y_px)  /* filter.c:725-757 */
{
  const int x = x_px - 4, end = y_px + 64 < d->height ? y_px + 64 : d->height;
  int tu_boundary;
  for (int y = y_px; y < end; y += 8)
    if (y > 0 && unit_edge(d, x, y, 0, &tu_boundary)) edge_luma(d, x, y, 4, 0, tu_boundary);
  const int xc = (x_px >> 1) - 4, yc0 = y_px >> 1, end_c = yc0 + 32 < (d->height >> 1) ? yc0 + 32 : d->height >> 1;
  for (int yc = yc0; yc < end_c; yc += 8)
    if (yc > 0 && unit_edge(d, xc << 1, yc << 1, 0, &tu_boundary)) edge_chroma(d, xc, yc, 4, 0);
}

void kvz_oracle_deblock_lcu(int width, int height, int qp, int beta_offset_div2, int tc_offset_div2, uint8_t *y, uint8_t *u, uint8_t *v,
                            const uint8_t *cu_depth, int x_px, int y_px)
{
  dbk_t d = { width, height, beta_prime(clip3(0, 51, qp + (beta_offset_div2 << 1))), tc_prime(clip3(0, 53, qp + 2 + (tc_offset_div2 << 1))),
              tc_prime(clip3(0, 53, chroma_qp[qp] + 2 + (tc_offset_div2 << 1))), y, u, v, cu_depth, NULL, 0, qp, tc_offset_div2 };
  deblock_lcu_inside(&d, x_px, y_px, 1);
  if (x_px > 0) deblock_lcu_rightmost(&d, x_px, y_px);
  deblock_lcu_inside(&d, x_px, y_px, 0);
}
/* the same LCU step on a picture with inter CUs: edges and per-part strengths from one kvz_hip_cu_dbk per 4x4 unit */
void kvz_oracle_deblock_lcu_inter(int width, int height, int qp, int beta_offset_div2, int tc_offset_div2, uint8_t *y, uint8_t *u, uint8_t *v,
                                  const kvz_hip_cu_dbk *info, int slice_is_b, int x_px, int y_px)
{
  dbk_t d = { width, height, beta_prime(clip3(0, 51, qp + (beta_offset_div2 << 1))), tc_prime(clip3(0, 53, qp + 2 + (tc_offset_div2 << 1))),
              tc_prime(clip3(0, 53, chroma_qp[qp] + 2 + (tc_offset_div2 << 1))), y, u, v, NULL, info, slice_is_b, qp, tc_offset_div2 };
  deblock_lcu_inside(&d, x_px, y_px, 1);
  if (x_px > 0) deblock_lcu_rightmost(&d, x_px, y_px);
  deblock_lcu_inside(&d, x_px, y_px, 0);
}

/* ---- pictures with inter CUs: boundary strengths from motion data (filter.c:405-493), prediction-unit edges (filter.c:225-257) -----------
 * info: one kvz_hip_cu_dbk per 4x4 unit.  Picture-level order as above (all vertical edges, then all horizontal ones); the edge grid, the
 * 4-sample parts and the filters are those of the intra case, only the per-part strength (and with it tc, or no filtering at all) differs. */
static const kvz_hip_cu_dbk *unit_at(const kvz_hip_cu_dbk *info, int width, int x, int y) { return info + (size_t)(y >> 2) * (width >> 2) + (x >> 2); }

static int inter_edge_on(const kvz_hip_cu_dbk *info, int width, int x, int y, int vertical, int *tu_boundary)
{
  static const int8_t second_x[8] = { -1, -1, 2, 2, -1, -1, 1, 3 }, second_y[8] = { -1, 2, -1, 2, 1, 3, -1, -1 };  /* cu.c:63-72, second partition's offset in CU quarters */
  const kvz_hip_cu_dbk *u = unit_at(info, width, x, y);
  const int pos = vertical ? x : y, cu_w = 64 >> u->depth, rel = pos & (cu_w - 1);
  *tu_boundary = (pos & ((64 >> u->tr_depth) - 1)) == 0;
  if (*tu_boundary || rel == 0) return 1;
  const int q = vertical ? second_x[u->part_size] : second_y[u->part_size];
  return q >= 0 && rel == q * cu_w / 4;
}

static int far4(int a, int b) { return abs(a - b) >= 4; }

static int inter_strength(const kvz_hip_cu_dbk *q, const kvz_hip_cu_dbk *p, int tu_boundary, int slice_b)
{
  if (q->type == 1 || p->type == 1) return 2;
  if (tu_boundary && (q->cbf_y || p->cbf_y)) return 1;
  if (p->mv_dir != 3 && q->mv_dir != 3) {
    if (far4(q->mv[q->mv_dir - 1][0], p->mv[p->mv_dir - 1][0]) || far4(q->mv[q->mv_dir - 1][1], p->mv[p->mv_dir - 1][1])) return 1;
    if (q->mv_ref[q->mv_dir - 1] != p->mv_ref[p->mv_dir - 1]) return 1;
  }
  if (!slice_b) return 0;
  int16_t mq[2][2], mp[2][2];
  for (int l = 0; l < 2; l++) for (int k = 0; k < 2; k++) { mq[l][k] = (q->mv_dir & (1 << l)) ? q->mv[l][k] : 0; mp[l][k] = (p->mv_dir & (1 << l)) ? p->mv[l][k] : 0; }
  const int rp0 = (p->mv_dir & 1) ? p->ref_id[0] : -1, rp1 = (p->mv_dir & 2) ? p->ref_id[1] : -1;
  const int rq0 = (q->mv_dir & 1) ? q->ref_id[0] : -1, rq1 = (q->mv_dir & 2) ? q->ref_id[1] : -1;
  if ((rp0 == rq0 && rp1 == rq1) || (rp0 == rq1 && rp1 == rq0)) {
    const int straight = far4(mq[0][0], mp[0][0]) || far4(mq[0][1], mp[0][1]) || far4(mq[1][0], mp[1][0]) || far4(mq[1][1], mp[1][1]);
    const int crossed = far4(mq[1][0], mp[0][0]) || far4(mq[1][1], mp[0][1]) || far4(mq[0][0], mp[1][0]) || far4(mq[0][1], mp[1][1]);
    if (rp0 != rp1) return (rp0 == rq0) ? straight : crossed;
    return straight && crossed;
  }
  return 1;
}

void kvz_oracle_deblock_frame_inter(int width, int height, int qp, int beta_offset_div2, int tc_offset_div2, uint8_t *y, uint8_t *u, uint8_t *v,
                                    const kvz_hip_cu_dbk *info, int slice_is_b)
{
  const int cw = width >> 1;
  const int beta = beta_prime(clip3(0, 51, qp + (beta_offset_div2 << 1)));
  const int tc_c = tc_prime(clip3(0, 53, chroma_qp[qp] + 2 + (tc_offset_div2 << 1)));
  for (int dir = 0; dir < 2; dir++) {
    const int vertical = dir == 0;
    for (int ey = 0; ey < height; ey += 8)
      for (int ex = 0; ex < width; ex += 8) {
        int tu_boundary;
        if ((vertical ? ex : ey) == 0) continue;
        if (!inter_edge_on(info, width, ex, ey, vertical, &tu_boundary)) continue;
        for (int part = 0; part < 2; part++) {
          const int px = ex + (vertical ? 0 : 4 * part), py = ey + (vertical ? 4 * part : 0);
          const int s = inter_strength(unit_at(info, width, px, py), unit_at(info, width, vertical ? px - 1 : px, vertical ? py : py - 1), tu_boundary, slice_is_b);
          if (!s) continue;
          const int tc = tc_prime(clip3(0, 53, qp + 2 * (s - 1) + (tc_offset_div2 << 1)));
          luma_part(y + py * width + px, vertical ? 1 : width, vertical ? width : 1, beta, tc);
        }
        const int xc = ex >> 1, yc = ey >> 1;
        if (((vertical ? xc : yc) & 7) == 0) {  /* filter.c:680, then :610: only next to an intra CU */
          const kvz_hip_cu_dbk *q = unit_at(info, width, ex, ey), *p = unit_at(info, width, vertical ? ex - 1 : ex, vertical ? ey : ey - 1);
          if (q->type == 1 || p->type == 1) {
            chroma_part(u + yc * cw + xc, vertical ? 1 : cw, vertical ? cw : 1, tc_c);
            chroma_part(v + yc * cw + xc, vertical ? 1 : cw, vertical ? cw : 1, tc_c);
          }
        }
      }
  }
}
