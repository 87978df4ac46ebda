/* batch_sim.c -- TEST INFRASTRUCTURE ONLY.  The subset of include/kvz_hip_batch.h that integration/kvazaar/search_lcu_hip.c calls, served by the ORACLE's
 * CTU pass (oracle/kvz_oracle_ctu.c) on the host.  oracle/Makefile links it into oracle/_ref/kvazaar_hipsim, a copy of the integrated encoder in which these
 * definitions shadow libkvz_hip.so's: it lets the CPU test suite check the BINDING's own logic (picture gathering, CU-array / coded-block-flag rebuild, tile views,
 * NxN partitions) against the reference encoder's bitstream on machines without a GPU.  It is never linked into, loaded by or shipped with the product; no -m gpu
 * test, bench leg or smoke() uses it -- those run oracle/_ref/kvazaar_hip, where the same calls reach the device. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/kvz_hip_batch.h"
#include "kvz_oracle.h"

struct kvz_hip_batch {
  int w, h, n;
  uint8_t *src, *rec, *depth, *mode, *part, *mode4;
  int16_t *coeff;
  double *cost;
  int has_parts;
  kvz_hip_sao_params *sao_luma, *sao_chroma;  /* the last kvz_hip_batch_loop_filters(..., sao = 1)'s decisions, one record per LCU and frame */
  uint8_t *sao_merge;
  int has_sao;
};

kvz_hip_batch *kvz_hip_batch_create(int width, int height, int n_frames)
{
  kvz_hip_batch *b = calloc(1, sizeof *b);
  const size_t px = (size_t)width * height * 3 / 2, nctu = (size_t)((width + 63) / 64) * ((height + 63) / 64);
  b->w = width; b->h = height; b->n = n_frames;
  b->src = calloc(px, n_frames); b->rec = calloc(px, n_frames);
  b->depth = calloc((size_t)(width / 8) * (height / 8), n_frames); b->mode = calloc((size_t)(width / 8) * (height / 8), n_frames);
  b->part = calloc((size_t)(width / 8) * (height / 8), n_frames); b->mode4 = calloc((size_t)(width / 4) * (height / 4), n_frames);
  b->coeff = calloc(nctu * KVZ_HIP_CTU_COEFFS * sizeof(int16_t), n_frames);
  b->cost = calloc(nctu * sizeof(double), n_frames);
  b->sao_luma = calloc(nctu * sizeof(kvz_hip_sao_params), n_frames); b->sao_chroma = calloc(nctu * sizeof(kvz_hip_sao_params), n_frames);
  b->sao_merge = calloc(nctu, n_frames);
  return b;
}
void kvz_hip_batch_destroy(kvz_hip_batch *b)
{
  if (!b) return;
  free(b->src); free(b->rec); free(b->depth); free(b->mode); free(b->part); free(b->mode4); free(b->coeff); free(b->cost); free(b->sao_luma); free(b->sao_chroma); free(b->sao_merge); free(b);
}
int kvz_hip_batch_ctus_per_frame(const kvz_hip_batch *b) { return ((b->w + 63) / 64) * ((b->h + 63) / 64); }
void kvz_hip_batch_upload(kvz_hip_batch *b, int frame, const uint8_t *y, const uint8_t *u, const uint8_t *v)
{
  const size_t ys = (size_t)b->w * b->h, cs = ys / 4;
  uint8_t *dst = b->src + (size_t)frame * (ys + 2 * cs);
  memcpy(dst, y, ys); memcpy(dst + ys, u, cs); memcpy(dst + ys + cs, v, cs);
}
int kvz_hip_intra_frames(kvz_hip_batch *b, const kvz_hip_intra_cost_model *model)
{
  const size_t ys = (size_t)b->w * b->h, cs = ys / 4, px = ys + 2 * cs, ncu = (size_t)(b->w / 8) * (b->h / 8), nctu = (size_t)kvz_hip_batch_ctus_per_frame(b);
  for (int f = 0; f < b->n; f++) {
    const uint8_t *s = b->src + f * px;
    uint8_t *r = b->rec + f * px;
    kvz_oracle_intra_frame_nxn(model, b->w, b->h, s, s + ys, s + ys + cs, r, r + ys, r + ys + cs, b->coeff + f * nctu * KVZ_HIP_CTU_COEFFS, b->depth + f * ncu, b->mode + f * ncu,
                               b->cost + f * nctu, b->part + f * ncu, b->mode4 + f * ncu * 4);
  }
  b->has_parts = model->search_nxn != 0;
  return 1;
}
int kvz_hip_batch_sync(kvz_hip_batch *b) { (void)b; return 0; }
int kvz_hip_batch_download(kvz_hip_batch *b, int frame, uint8_t *rec_y, uint8_t *rec_u, uint8_t *rec_v, int16_t *coeff, uint8_t *cu_depth, uint8_t *cu_mode, double *ctu_cost)
{
  const size_t ys = (size_t)b->w * b->h, cs = ys / 4, px = ys + 2 * cs, ncu = (size_t)(b->w / 8) * (b->h / 8), nctu = (size_t)kvz_hip_batch_ctus_per_frame(b);
  if (rec_y) memcpy(rec_y, b->rec + frame * px, ys);
  if (rec_u) memcpy(rec_u, b->rec + frame * px + ys, cs);
  if (rec_v) memcpy(rec_v, b->rec + frame * px + ys + cs, cs);
  if (coeff) memcpy(coeff, b->coeff + frame * nctu * KVZ_HIP_CTU_COEFFS, nctu * KVZ_HIP_CTU_COEFFS * sizeof(int16_t));
  if (cu_depth) memcpy(cu_depth, b->depth + frame * ncu, ncu);
  if (cu_mode) memcpy(cu_mode, b->mode + frame * ncu, ncu);
  if (ctu_cost) memcpy(ctu_cost, b->cost + frame * nctu, nctu * sizeof(double));
  return 0;
}
int kvz_hip_batch_download_partitions(kvz_hip_batch *b, int frame, uint8_t *cu_part, uint8_t *cu_mode4)
{
  const size_t ncu = (size_t)(b->w / 8) * (b->h / 8);
  if (!b->has_parts) return -1;
  if (cu_part) memcpy(cu_part, b->part + frame * ncu, ncu);
  if (cu_mode4) memcpy(cu_mode4, b->mode4 + frame * ncu * 4, ncu * 4);
  return 0;
}
void *kvz_hip_host_alloc(size_t bytes) { return malloc(bytes ? bytes : 1); }
void kvz_hip_host_free(void *p) { free(p); }

/* ---- the calls of the binding's B-picture path (include/kvz_hip_dev.h), served by kvz_oracle_inter_picture: "device" memory is host memory here ---- */
#include "../include/kvz_hip_dev.h"
void *kvz_hip_dev_alloc(size_t bytes) { return malloc(bytes ? bytes : 1); }
void kvz_hip_dev_free(void *p) { free(p); }
void kvz_hip_dev_upload(void *dev_dst, const void *host_src, size_t bytes) { memcpy(dev_dst, host_src, bytes); }
void kvz_hip_dev_download(void *host_dst, const void *dev_src, size_t bytes) { memcpy(host_dst, dev_src, bytes); }
int kvz_hip_dev_inter_ctu_pass(const uint8_t *src, const uint8_t *ref, const kvz_hip_cu_info *ref_cu, uint8_t *rec, kvz_hip_cu_info *cu, int16_t *coeff, int width,
                               int height, int n_pictures, const kvz_hip_inter_params *p)
{
  _Static_assert(sizeof(kvz_hip_cu_info) == sizeof(kvz_oracle_cu), "the two CU records are one layout");
  if (p->qp < 0 || p->qp > 51 || p->fast_residual_cost < 0 || p->fast_residual_cost > 51 || p->poc < 1 || width % 8 || height % 8) return -1;
  const size_t px = (size_t)width * height * 3 / 2, cells = (size_t)(width / 4) * (height / 4), nctu = (size_t)((width + 63) / 64) * ((height + 63) / 64);
  kvz_hip_intra_cost_model m;  /* for kvz_f_entropy_bits and the coefficient weights of the QP only */
  kvz_hip_intra_cost_model_init(p->qp, kvz_hip_default_coeff_weights(p->qp), &m);
  kvz_oracle_lowdelay_cfg cfg;
  memset(&cfg, 0, sizeof cfg);
  cfg.fme_level = p->fme_level; cfg.pu_depth_inter_max = p->pu_depth_inter_max; cfg.sao = p->sao; cfg.deblock = p->deblock; cfg.mv_constraint = p->mv_constraint;
  cfg.no_wpp = p->no_wpp; cfg.fast_residual_cost = p->fast_residual_cost;
  for (int f = 0; f < n_pictures; f++)
    kvz_oracle_inter_picture(p->qp, p->poc, &cfg, m.entropy_fbits, kvz_hip_default_coeff_weights(p->qp), width, height, src + f * px, ref + f * px,
                             (const kvz_oracle_cu *)ref_cu + f * cells, rec + f * px, (kvz_oracle_cu *)cu + f * cells, coeff ? coeff + f * nctu * KVZ_HIP_CTU_COEFFS : NULL);
  return 0;
}

/* ---- the entropy coder's real mode (include/kvz_hip_batch.h kvz_hip_batch_entropy_code), served by kvz_oracle_entropy_intra_picture ---- */
long kvz_hip_batch_entropy_code_tiles(kvz_hip_batch *b, const kvz_hip_intra_cost_model *model, int sao, const uint8_t *not_last, uint8_t *out, size_t capacity,
                                      uint32_t *substream_bytes);
long kvz_hip_batch_entropy_code(kvz_hip_batch *b, const kvz_hip_intra_cost_model *model, int sao, uint8_t *out, size_t capacity, uint32_t *substream_bytes)
{
  return kvz_hip_batch_entropy_code_tiles(b, model, sao, NULL, out, capacity, substream_bytes);
}
long kvz_hip_batch_entropy_code_tiles(kvz_hip_batch *b, const kvz_hip_intra_cost_model *model, int sao, const uint8_t *not_last, uint8_t *out, size_t capacity,
                                      uint32_t *substream_bytes)
{
  if (sao && !b->has_sao) return -1;
  const size_t ys = (size_t)b->w * b->h, ncu = (size_t)(b->w / 8) * (b->h / 8), nctu = (size_t)kvz_hip_batch_ctus_per_frame(b);
  const int rows = model->no_wpp ? 1 : (b->h + 63) / 64;
  size_t total = 0;
  (void)ys;
  for (int f = 0; f < b->n; f++) {
    const size_t n = kvz_oracle_entropy_intra_tile(model, b->w, b->h, b->depth + f * ncu, b->mode + f * ncu, model->search_nxn ? b->part + f * ncu : NULL,
                                                   model->search_nxn ? b->mode4 + f * ncu * 4 : NULL, b->coeff + f * nctu * KVZ_HIP_CTU_COEFFS,
                                                   sao ? b->sao_luma + f * nctu : NULL, sao ? b->sao_chroma + f * nctu : NULL, sao ? b->sao_merge + f * nctu : NULL,
                                                   not_last ? not_last[f] : 0, out + total, capacity - total, substream_bytes + (size_t)f * rows);
    if (total + n > capacity) return -1;
    total += n;
  }
  return (long)total;
}

/* kvz_hip_batch_loop_filters: deblocking + the SAO decision (+ SAO) of every picture, by the oracle; what the sim keeps of it are the decisions */
void kvz_hip_batch_loop_filters(kvz_hip_batch *b, const kvz_hip_intra_cost_model *model, int deblock, int beta_offset_div2, int tc_offset_div2, int sao)
{
  const size_t ys = (size_t)b->w * b->h, px = ys * 3 / 2, ncu = (size_t)(b->w / 8) * (b->h / 8), nctu = (size_t)kvz_hip_batch_ctus_per_frame(b);
  for (int f = 0; f < b->n; f++) {
    uint8_t *r = b->rec + f * px;
    if (sao) kvz_oracle_sao_search_frame(model, b->w, b->h, b->src + f * px, r, b->depth + f * ncu, deblock, beta_offset_div2, tc_offset_div2, b->sao_luma + f * nctu,
                                         b->sao_chroma + f * nctu, b->sao_merge + f * nctu);
    else if (deblock) kvz_oracle_deblock_frame(b->w, b->h, model->qp, beta_offset_div2, tc_offset_div2, r, r + ys, r + ys + ys / 4, b->depth + f * ncu);
  }
  b->has_sao = sao != 0;
}
