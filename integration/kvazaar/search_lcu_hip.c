/*
 * search_lcu_hip.c -- kvazaar-side binding of the BATCHED pass (include/kvz_hip_batch.h): what a kvazaar maintainer would put
 * in front of kvz_search_lcu (search.c:1209) to let the device search and reconstruct whole pictures.
 *
 * kvz_search_lcu(state, x, y, ...) leaves three things behind for the rest of encoder_state_worker_encode_lcu_search
 * (encoderstate.c:659-720: deblocking, SAO, kvz_encode_coding_tree): the LCU's cu_info in frame->cu_array, its reconstruction in
 * frame->rec and its quantised coefficients in state->coeff (copy_lcu_to_cu_data / copy_coeffs, search.c:1180-1249).  For the
 * configuration the batched pass implements -- I slices of an `ultrafast` .. `medium`-like setup, 8-bit 4:2:0, constant QP, with or without WPP; since round 3 also
 * the B pictures of a low-delay GOP (search_lcu_inter below: kvz_hip_dev_inter_ctu_pass) --
 * this file fills exactly those from one kvz_hip_intra_frames() run per picture: the first LCU of a picture to get here runs the
 * pass for the whole picture (one-frame batch; the throughput path batches many pictures, this binding is about correctness),
 * every LCU then copies its part.  Everything else falls through to the original function.  The bitstream is the reference's,
 * byte for byte (tests/test_e2e_dropin.py).
 *
 * In this repository the splice is done at link time (-Wl,--wrap=kvz_search_lcu, oracle/Makefile) because reference sources
 * are not modified; enabled at run time by KVZ_HIP_BATCH_SEARCH=1.
 */
#include <pthread.h>
#include <unistd.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "global.h" // IWYU pragma: keep
#include "cabac.h"
#include "cu.h"
#include "encoder.h"
#include "encoderstate.h"
#include "fast_coeff_cost.h"
#include "bitstream.h"
#include "image.h"
#include "videoframe.h"

#include "kvz_hip_batch.h"
#include "kvz_hip_dev.h"

void __real_kvz_search_lcu(encoder_state_t *const state, const int x, const int y, const yuv_t *const hor_buf, const yuv_t *const ver_buf);
void __real_kvz_encode_coding_tree(encoder_state_t *const state, uint16_t x, uint16_t y, uint8_t depth);
void __real_kvz_encoder_state_worker_write_bitstream(void *opaque);

#define KVZ_HIP_MAX_LCU_ROWS 136  /* 8192 / 64 + a margin */
typedef struct {
  const videoframe_t *frame;  /* key: the (tile) frame and the picture it holds */
  int32_t num;
  int width, height, qp;
  uint8_t *src;                /* the source picture as tight planes Y|U|V (pinned), filled by the thread that registered the picture */
  uint8_t *rec, *depth, *mode; /* tight planes Y|U|V; one byte per 8x8 (pinned) */
  uint8_t *part, *mode4;       /* model.search_nxn: NxN flag per 8x8 CU, luma mode per 4x4 unit */
  int16_t *coeff;              /* KVZ_HIP_CTU_COEFFS per LCU, raster LCU order */
  /* KVZ_HIP_BATCH_ENTROPY=1: the picture's slice data, coded on the device (kvz_hip_batch_entropy_code) -- one substream per LCU row (WPP) or one for the picture;
   * then the levels stay on the device, kvz_encode_coding_tree is skipped and the row coders' streams are replaced by these bytes before the slice header is written */
  uint8_t *ent; size_t ent_cap;
  uint32_t ent_sizes[KVZ_HIP_MAX_LCU_ROWS];
  int ent_ready;               /* this picture's slice data is in ent */
  int ent_hold;                /* ... and has not been handed to the bitstream yet: the slot must not be reused */
  int ent_not_last;            /* the picture is a tile, and not its slice's last */
  int ent_sao, ent_deblock, ent_beta, ent_tc;  /* SAO syntax to write: the device's own loop filters run first (their decisions are what gets coded) */
  kvz_hip_intra_cost_model model;
  int state;                   /* FREE -> PENDING (registered, waiting for a pass) -> COMPUTING (in the leader's batch) -> READY */
  int outstanding;             /* LCUs of the picture that have not copied their part yet; the slot is only reused at 0 (under g_lock) */
} picture_result;
enum { SLOT_FREE = 0, SLOT_PENDING, SLOT_COMPUTING, SLOT_READY };
static int g_entropy = -1;  /* KVZ_HIP_BATCH_ENTROPY */

/* One slot per picture in flight: (owf + 1) x tiles of them at most, so the table grows on demand and a slot is never recycled
 * while an LCU of its picture is still to come (every LCU of a picture passes through kvz_search_lcu exactly once).
 *
 * Frame batching.  With --owf N kvazaar has N + 1 pictures in flight, and the first LCU of each of them asks for its picture's results at about
 * the same time.  The device pass is a latency machine (a lone 1080p picture takes ~75 ms, sixty-four of them ~100 ms), so pictures are
 * gathered: the thread that finds no pass running becomes the leader, waits until half of the pictures in flight have registered (at most
 * KVZ_HIP_BATCH_WINDOW_US, default 40 ms), runs ONE kvz_hip_intra_frames over everything pending with the same geometry and model (up to
 * KVZ_HIP_BATCH_MAX, default 64), and wakes the rest.  Pictures that register while a pass runs form the next batch. */
static picture_result **g_slots;
static int g_n_slots;
static pthread_mutex_t g_lock = PTHREAD_MUTEX_INITIALIZER;
static pthread_cond_t g_cond = PTHREAD_COND_INITIALIZER;
static int g_leader_active;
#define N_BATCH_SIZES 4
static const int g_batch_sizes[N_BATCH_SIZES] = { 1, 4, 16, 64 };
static kvz_hip_batch *g_batches[N_BATCH_SIZES];  /* one batch per capacity, created on first use; all of the geometry g_batch_w x g_batch_h */
static int g_batch_w, g_batch_h;

/* 1 when the picture can be searched by the batched pass: every option below changes the search in a way the pass does not model */
static int eligible(const encoder_state_t *state)
{
  static int enabled = -1;
  if (enabled < 0) { const char *e = getenv("KVZ_HIP_BATCH_SEARCH"); enabled = e ? atoi(e) : 0; }
  if (!enabled) return 0;
  const encoder_control_t *ctrl = state->encoder_control;
  const kvz_config *cfg = &ctrl->cfg;
#define REQUIRE(cond) do { if (!(cond)) { if (enabled > 1) fprintf(stderr, "search_lcu_hip: not eligible: %s\n", #cond); return 0; } } while (0)
  REQUIRE(state->frame->slicetype == KVZ_SLICE_I || state->frame->slicetype == KVZ_SLICE_B);
  REQUIRE(ctrl->bitdepth == 8 && ctrl->chroma_format == KVZ_CSP_420);
  REQUIRE(cfg->rdo == 0 && !cfg->signhide_enable && !cfg->trskip_enable && cfg->tr_depth_intra == 0);
  /* --rdoq (preset medium): kvz_rdoq in every quantisation (rdoq-skip 0), priced on the CABAC model (fast-residual-cost 0) */
  REQUIRE(!cfg->rdoq_enable || (!cfg->rdoq_skip && !(state->qp < cfg->fast_residual_cost_limit && state->qp < MAX_FAST_COEFF_COST_QP)));
  REQUIRE(!cfg->lossless && !cfg->implicit_rdpcm && cfg->scaling_list == KVZ_SCALING_LIST_OFF);
  REQUIRE(!cfg->full_intra_search);
  /* all-intra: GOP layer 0 only; 1-3 = preset `fast`, 1-4 = `medium` (8x8 CUs also tried as four 4x4 PUs) */
  REQUIRE((cfg->pu_depth_intra.min[0] == 2 || cfg->pu_depth_intra.min[0] == 1) && (cfg->pu_depth_intra.max[0] == 3 || cfg->pu_depth_intra.max[0] == 4));
  REQUIRE(cfg->cu_split_termination == KVZ_CU_SPLIT_TERMINATION_ZERO && cfg->combine_intra_cus);
  REQUIRE(cfg->target_bitrate <= 0 && !cfg->vaq && !cfg->roi.file_path && !cfg->set_qp_in_cu && state->frame->max_qp_delta_depth < 0);
  REQUIRE(!cfg->ml_pu_depth_intra);  /* (intra_bit_allocation, which lp GOPs switch on, only acts under rate control: rate_control.c:352-705) */
  if (state->frame->slicetype == KVZ_SLICE_I) return 1;
  /* B pictures: the inter CTU pass (include/kvz_hip_dev.h kvz_hip_dev_inter_ctu_pass) is the search of `--preset faster|veryfast|superfast|ultrafast --gop lp-gNd*t1`:
   * one reference picture -- the previous one -- in both lists, hexagon search with the `sensitive` early termination, any fme level, bi-prediction through
   * merge candidates only, early skip, 2Nx2N PUs of 8..32 samples, coefficients priced as kvz_get_coeff_cost does with the configuration's fast-residual-cost.  --owf 0: the pass searches the whole
   * picture when its first LCU arrives, so the reference picture has to be complete by then. */
  const encoder_state_config_frame_t *fr = state->frame;
  REQUIRE(cfg->owf == 0 && cfg->tiles_width_count * cfg->tiles_height_count <= 1 && cfg->slices == KVZ_SLICES_NONE);
  REQUIRE(fr->ref->used_size == 1 && fr->ref_LX_size[0] == 1 && fr->ref_LX_size[1] == 1 && fr->ref->pocs[0] == fr->poc - 1);
  REQUIRE(cfg->gop_len > 0 && cfg->gop_lowdelay && cfg->bipred && cfg->fast_bipred && cfg->tmvp_enable);
  REQUIRE(cfg->ime_algorithm == KVZ_IME_HEXBS && cfg->me_early_termination == KVZ_ME_EARLY_TERMINATION_SENSITIVE && cfg->me_max_steps == (uint32_t)-1);
  REQUIRE(cfg->fme_level >= 0 && cfg->fme_level <= 4 && !cfg->mv_rdo && cfg->mv_constraint == KVZ_MV_CONSTRAIN_NONE);
  REQUIRE(cfg->early_skip && cfg->max_merge == 5 && cfg->zero_coeff_rdo && !cfg->smp_enable && !cfg->amp_enable && !cfg->rdoq_enable);
  REQUIRE(cfg->pu_depth_inter.min[0] == 1 && (cfg->pu_depth_inter.max[0] == 2 || cfg->pu_depth_inter.max[0] == 3));
  REQUIRE(cfg->pu_depth_intra.min[0] == 2 && cfg->pu_depth_intra.max[0] == 3);
  {  /* search.c:674-687 takes the depths of the picture's GOP layer where the configuration sets them (>= 0): the pass is handed layer 0's */
    const int layer = cfg->gop_len != 0 ? cfg->gop[fr->gop_offset].layer - 1 : 0;
    REQUIRE(layer >= 0 && layer < KVZ_MAX_GOP_LAYERS);
    REQUIRE((cfg->pu_depth_inter.min[layer] < 0 || cfg->pu_depth_inter.min[layer] == cfg->pu_depth_inter.min[0]) && (cfg->pu_depth_inter.max[layer] < 0 || cfg->pu_depth_inter.max[layer] == cfg->pu_depth_inter.max[0]));
    REQUIRE((cfg->pu_depth_intra.min[layer] < 0 || cfg->pu_depth_intra.min[layer] == cfg->pu_depth_intra.min[0]) && (cfg->pu_depth_intra.max[layer] < 0 || cfg->pu_depth_intra.max[layer] == cfg->pu_depth_intra.max[0]));
  }
  /* the pass prices with the built-in fast-coefficient-cost weights (kvz_hip_default_coeff_weights): a custom --fast-coeff-table stays with kvz_search_lcu.  The test is on the
   * PARSED table: encoder.c:168 clears cfg.fast_coeff_table_fn once the file has been read into ctrl->fast_coeff_table, so the file name says nothing here. */
  REQUIRE(state->qp < 0 || state->qp >= MAX_FAST_COEFF_COST_QP || ctrl->fast_coeff_table.wts_by_qp[state->qp] == kvz_hip_default_coeff_weights(state->qp));
  REQUIRE(cfg->fast_residual_cost_limit >= 0 && cfg->fast_residual_cost_limit <= 51 && !cfg->intra_rdo_et);  /* the pass prices coefficients either way (rdo.c:311-340) */
  REQUIRE(state->tile->frame->width % 8 == 0 && state->tile->frame->height % 8 == 0);
#undef REQUIRE
  return 2;
}

static int env_int(const char *name, int dflt)
{
  const char *e = getenv(name);
  return e && atoi(e) > 0 ? atoi(e) : dflt;
}

/* the leader's work: one pass over `n` gathered pictures (same geometry and model) */
static kvz_hip_batch *g_last_batch;  /* the batch object run_pictures used (the leader codes its pictures afterwards) */
static int g_last_batch_capacity;
static void run_pictures(picture_result **list, int n)
{
  const int w = list[0]->width, h = list[0]->height;
  const size_t ys = (size_t)w * h, cs = ys / 4;
  if (g_batch_w != w || g_batch_h != h) {
    for (int i = 0; i < N_BATCH_SIZES; i++) if (g_batches[i]) { kvz_hip_batch_destroy(g_batches[i]); g_batches[i] = NULL; }
    g_batch_w = w; g_batch_h = h;
  }
  int bi = 0;
  while (g_batch_sizes[bi] < n) bi++;
  if (!g_batches[bi]) {
    g_batches[bi] = kvz_hip_batch_create(w, h, g_batch_sizes[bi]);
    if (!g_batches[bi]) { fprintf(stderr, "search_lcu_hip: cannot create a %dx%d batch of %d pictures\n", w, h, g_batch_sizes[bi]); abort(); }
  }
  kvz_hip_batch *b = g_batches[bi];
  g_last_batch = b; g_last_batch_capacity = g_batch_sizes[bi];
  /* slots beyond n keep whatever picture they held last: searched again, never read */
  for (int i = 0; i < n; i++) kvz_hip_batch_upload(b, i, list[i]->src, list[i]->src + ys, list[i]->src + ys + cs);
  if (kvz_hip_intra_frames(b, &list[0]->model) < 0) { fprintf(stderr, "search_lcu_hip: the library cannot run this model\n"); abort(); }
  /* the library reports an invalid run instead of aborting; this binding has no other search to fall back to for pictures whose
   * LCUs are already being handed out, so it stops the encoder */
  int bad = kvz_hip_batch_sync(b) != 0;
  const int entropy = g_entropy > 0 && list[0]->ent_hold;  /* (set with the slot: the configuration has no SAO syntax) */
  for (int i = 0; i < n && !bad; i++)
    bad = kvz_hip_batch_download(b, i, list[i]->rec, list[i]->rec + ys, list[i]->rec + ys + cs, entropy ? NULL : list[i]->coeff, list[i]->depth, list[i]->mode, NULL) != 0;
  for (int i = 0; i < n && !bad && list[0]->model.search_nxn; i++) bad = kvz_hip_batch_download_partitions(b, i, list[i]->part, list[i]->mode4) != 0;
  if (bad) { fprintf(stderr, "search_lcu_hip: the device pass failed\n"); abort(); }
  {  /* KVZ_HIP_BATCH_TRACE=<file>: "pictures passes largest-batch" so far (tests check the path was taken, and that pictures were gathered) */
    static int pictures, passes, largest;
    const char *trace = getenv("KVZ_HIP_BATCH_TRACE");
    pictures += n; passes++;
    if (n > largest) largest = n;
    if (trace) { FILE *f = fopen(trace, "w"); if (f) { fprintf(f, "%d %d %d\n", pictures, passes, largest); fclose(f); } }
  }
}

/* the leader, after the pictures' LCUs have been released to the workers: the slice data of the same pictures (nothing needs it before the frame's bitstream is written) */
static void run_entropy(picture_result **list, int n)
{
  kvz_hip_batch *b = g_last_batch;
  const int w = list[0]->width, h = list[0]->height;
  /* every picture of the batch object is coded (slots beyond n hold older pictures); only the first n are read */
  const int cap_n = g_last_batch_capacity, rows = list[0]->model.no_wpp ? 1 : (h + 63) / 64;
  const size_t cap = (size_t)cap_n * ((size_t)w * h * 2 + 65536);
  static uint8_t *all; static size_t all_cap; static uint32_t *sizes; static size_t sizes_cap;
  if (all_cap < cap) { free(all); all = malloc(cap); all_cap = cap; }
  if (sizes_cap < (size_t)cap_n * rows) { free(sizes); sizes = malloc((size_t)cap_n * rows * sizeof *sizes); sizes_cap = (size_t)cap_n * rows; }
  if (!all || !sizes || rows > KVZ_HIP_MAX_LCU_ROWS) { fprintf(stderr, "search_lcu_hip: out of memory\n"); abort(); }
  uint8_t not_last[64] = { 0 };
  for (int i = 0; i < n; i++) not_last[i] = (uint8_t)list[i]->ent_not_last;
  const int sao = list[0]->ent_sao;
  if (sao) kvz_hip_batch_loop_filters(b, &list[0]->model, list[0]->ent_deblock, list[0]->ent_beta, list[0]->ent_tc, 1);  /* (the pictures before the filters are on the host already) */
  if (kvz_hip_batch_entropy_code_tiles(b, &list[0]->model, sao, not_last, all, cap, sizes) < 0) { fprintf(stderr, "search_lcu_hip: the device entropy coder failed\n"); abort(); }
  size_t at = 0;
  for (int i = 0; i < n; i++) {
    size_t bytes = 0;
    for (int r = 0; r < rows; r++) { list[i]->ent_sizes[r] = sizes[(size_t)i * rows + r]; bytes += sizes[(size_t)i * rows + r]; }
    if (list[i]->ent_cap < bytes) { free(list[i]->ent); list[i]->ent = malloc(bytes); list[i]->ent_cap = bytes; }
    if (!list[i]->ent) { fprintf(stderr, "search_lcu_hip: out of memory\n"); abort(); }
    memcpy(list[i]->ent, all + at, bytes);
    at += bytes;
  }
  {  /* KVZ_HIP_ENTROPY_TRACE=<file>: pictures whose slice data the device has written so far */
    static int coded;
    const char *trace = getenv("KVZ_HIP_ENTROPY_TRACE");
    coded += n;
    if (trace) { FILE *f = fopen(trace, "w"); if (f) { fprintf(f, "%d\n", coded); fclose(f); } }
  }
}

/* the picture's results: registered on first request, computed with whatever else is pending */
static picture_result *picture_of(const encoder_state_t *state)
{
  const videoframe_t *frame = state->tile->frame;
  pthread_mutex_lock(&g_lock);
  picture_result *r = NULL;
  for (int i = 0; i < g_n_slots && !r; i++)
    if (g_slots[i]->outstanding > 0 && g_slots[i]->frame == frame && g_slots[i]->num == state->frame->num) r = g_slots[i];
  if (!r) {
    for (int i = 0; i < g_n_slots && !r; i++)
      if (g_slots[i]->outstanding == 0 && !g_slots[i]->ent_hold) r = g_slots[i];
    if (!r) {
      picture_result **grown = realloc(g_slots, (size_t)(g_n_slots + 1) * sizeof *g_slots);
      r = grown ? calloc(1, sizeof *r) : NULL;
      if (!r) { fprintf(stderr, "search_lcu_hip: out of memory\n"); abort(); }
      g_slots = grown;
      g_slots[g_n_slots++] = r;
    }
    const int w = frame->width, h = frame->height, wc = (w + 63) / 64, hc = (h + 63) / 64;
    const size_t ys = (size_t)w * h, cs = ys / 4;
    if (r->width != w || r->height != h) {  /* pinned: the transfers of a 64-picture batch are 800 MB */
      kvz_hip_host_free(r->src); kvz_hip_host_free(r->rec); kvz_hip_host_free(r->depth); kvz_hip_host_free(r->mode); kvz_hip_host_free(r->coeff);
      kvz_hip_host_free(r->part); kvz_hip_host_free(r->mode4);
      r->part = kvz_hip_host_alloc((size_t)(w / 8) * (h / 8));
      r->mode4 = kvz_hip_host_alloc((size_t)(w / 4) * (h / 4));
      r->src = kvz_hip_host_alloc(ys + 2 * cs);
      r->rec = kvz_hip_host_alloc(ys + 2 * cs);
      r->depth = kvz_hip_host_alloc((size_t)(w / 8) * (h / 8));
      r->mode = kvz_hip_host_alloc((size_t)(w / 8) * (h / 8));
      r->coeff = kvz_hip_host_alloc((size_t)wc * hc * KVZ_HIP_CTU_COEFFS * sizeof(int16_t));
      r->width = w; r->height = h;
    }
    r->frame = frame; r->num = state->frame->num; r->qp = state->qp;
    r->outstanding = wc * hc;
    if (g_entropy < 0) { const char *e = getenv("KVZ_HIP_BATCH_ENTROPY"); g_entropy = e ? atoi(e) : 0; }
    r->ent_ready = 0;
    /* the device writes the slice data of configurations whose LCUs carry no SAO syntax (the SAO decision is made on the host here) and one slice per picture; a
     * tile that is not the picture's last ends in end_of_subset_one_bit instead of end_of_slice_segment_flag (encoderstate.c:699-724) */
    {
      const kvz_config *c = &state->encoder_control->cfg;
      const int tiles = c->tiles_width_count * c->tiles_height_count;
      /* SAO syntax: the device then makes the SAO decision of the pictures as well (kvz_hip_batch_loop_filters, the same decision kvz_sao_search_lcu makes on the host
       * from the same pictures) -- `--sao full` without tiles (the loop filters of a tile read its neighbours) */
      r->ent_sao = c->sao_type == KVZ_SAO_FULL && tiles <= 1;
      r->ent_deblock = c->deblock_enable != 0; r->ent_beta = c->deblock_beta; r->ent_tc = c->deblock_tc;
      r->ent_hold = g_entropy > 0 && (c->sao_type == KVZ_SAO_OFF || r->ent_sao) && c->slices == KVZ_SLICES_NONE;
      r->ent_not_last = tiles > 1 && state->tile->id != tiles - 1;
    }
    r->state = SLOT_PENDING;
    const kvz_config *cfg = &state->encoder_control->cfg;
    kvz_hip_intra_cost_model_init(state->qp, kvz_fast_coeff_get_weights(state), &r->model);
    r->model.coeff_cabac = !(state->qp < cfg->fast_residual_cost_limit && state->qp < MAX_FAST_COEFF_COST_QP);  /* rdo.c:311-340 */
    r->model.search_32x32 = cfg->pu_depth_intra.min[0] == 1;  /* 32x32 CUs are searched, not only merged (search.c:794) */
    r->model.rdoq = cfg->rdoq_enable != 0;
    r->model.search_nxn = cfg->pu_depth_intra.max[0] == 4;  /* depth 4 of search_cu: the NxN partition of 8x8 CUs (search.c:691, 794) */
    r->model.no_wpp = !cfg->wpp;  /* kvazaar switches WPP off when tiles are used (cfg.c:925-978) */
    /* kvz_picture planes carry a stride; the batch takes tight planes.  (Outside the lock: only this thread knows the slot is being filled --
     * nobody gathers a slot before it is PENDING ... so mark it pending only afterwards.) */
    r->state = SLOT_FREE;
    pthread_mutex_unlock(&g_lock);
    const kvz_picture *pic = frame->source;
    for (int row = 0; row < h; row++) memcpy(r->src + (size_t)row * w, pic->y + (size_t)row * pic->stride, w);
    for (int row = 0; row < h / 2; row++) {
      memcpy(r->src + ys + (size_t)row * (w / 2), pic->u + (size_t)row * (pic->stride / 2), w / 2);
      memcpy(r->src + ys + cs + (size_t)row * (w / 2), pic->v + (size_t)row * (pic->stride / 2), w / 2);
    }
    pthread_mutex_lock(&g_lock);
    r->state = SLOT_PENDING;
  }
  while (r->state != SLOT_READY) {
    if (r->state == SLOT_PENDING && !g_leader_active) {
      g_leader_active = 1;
      pthread_mutex_unlock(&g_lock);
      /* the gather window: until half of the pictures kvazaar keeps in flight (--owf + 1) are pending -- the other half is then being entropy-coded
       * on the host while this pass runs -- but never longer than KVZ_HIP_BATCH_WINDOW_US (default 40 ms, half a lone picture's pass) */
      const int max_n = env_int("KVZ_HIP_BATCH_MAX", 64) < 64 ? env_int("KVZ_HIP_BATCH_MAX", 64) : 64;
      /* ... and never for more pictures than can register at all: every picture in the window is brought here by a worker thread that then blocks on the
       * condition variable below, so with few workers (--threads 1 / 2) the others cannot arrive; tiles are pictures of their own.  The wait also ends as
       * soon as nothing new has registered for 1 ms: a full window is only paid while pictures keep arriving. */
      const kvz_config *cfg = &state->encoder_control->cfg;
      const int tiles = cfg->tiles_width_count * cfg->tiles_height_count, in_flight = (cfg->owf + 1) * (tiles > 0 ? tiles : 1);
      const int workers = cfg->threads > 0 ? cfg->threads : 1;
      int want = in_flight / 2 > 0 ? in_flight / 2 : 1;
      if (want > workers) want = workers;
      if (want > max_n) want = max_n;
      const int window_us = env_int("KVZ_HIP_BATCH_WINDOW_US", 40000);
      for (int waited = 0, last_pending = -1, quiet_us = 0;; waited += 500) {
        pthread_mutex_lock(&g_lock);
        int pending = 0;
        for (int i = 0; i < g_n_slots; i++) pending += g_slots[i]->state == SLOT_PENDING;
        quiet_us = pending == last_pending ? quiet_us + 500 : 0;
        last_pending = pending;
        if (pending >= want || waited >= window_us || quiet_us >= 1000) break;  /* leaves with the lock held */
        pthread_mutex_unlock(&g_lock);
        usleep(500);
      }
      picture_result *list[64];
      int n = 0;
      list[n++] = r;
      r->state = SLOT_COMPUTING;
      for (int i = 0; i < g_n_slots && n < max_n; i++) {
        picture_result *o = g_slots[i];
        if (o->state == SLOT_PENDING && o->width == r->width && o->height == r->height && memcmp(&o->model, &r->model, sizeof o->model) == 0) { o->state = SLOT_COMPUTING; list[n++] = o; }
      }
      pthread_mutex_unlock(&g_lock);
      run_pictures(list, n);
      pthread_mutex_lock(&g_lock);
      for (int i = 0; i < n; i++) list[i]->state = SLOT_READY;
      pthread_cond_broadcast(&g_cond);
      if (g_entropy > 0 && list[0]->ent_hold) {  /* the LCUs go out now; the device codes the pictures meanwhile, and the next pass waits for that */
        pthread_mutex_unlock(&g_lock);
        run_entropy(list, n);
        pthread_mutex_lock(&g_lock);
        for (int i = 0; i < n; i++) list[i]->ent_ready = 1;
      }
      g_leader_active = 0;
      pthread_cond_broadcast(&g_cond);
    } else {
      pthread_cond_wait(&g_cond, &g_lock);
    }
  }
  pthread_mutex_unlock(&g_lock);
  return r;
}

static int any_level(const int16_t *c, int n)
{
  for (int i = 0; i < n; i++) if (c[i]) return 1;
  return 0;
}
static unsigned zorder16(int x, int y) /* cu.h:385-421 xy_to_zorder for 4-sample units, in coefficients */
{
  unsigned r = 0;
  for (int b = 0; b < 4; b++) r |= (((unsigned)(x >> (2 + b)) & 1u) << (2 * b)) | (((unsigned)(y >> (2 + b)) & 1u) << (2 * b + 1));
  return r * 16;
}

/* ---- B pictures: one kvz_hip_dev_inter_ctu_pass per picture.  --owf 0 keeps one picture in flight, so a single set of buffers serves: the first LCU of a
 * picture to get here uploads the source, the reference picture (after its loop filters) and the reference's CU array, runs the pass and downloads the
 * reconstruction, CU records and coefficients; the other LCUs wait for it. */
static struct {
  int w, h, ready, busy;
  int32_t num;
  const void *owner;  /* the picture's videoframe: with the frame number the key of what the buffers hold (two encoders of one process count the same numbers) */
  uint8_t *src, *ref, *rec;
  kvz_hip_cu_info *ref_cu, *cu;
  int16_t *coeff;
  void *d_src, *d_ref, *d_rec, *d_ref_cu, *d_cu, *d_coeff;
} g_inter = { .num = -1 };

static void tight_planes(uint8_t *dst, const kvz_picture *pic, int w, int h)
{
  const size_t ys = (size_t)w * h, cs = ys / 4;
  for (int row = 0; row < h; row++) memcpy(dst + (size_t)row * w, pic->y + (size_t)row * pic->stride, w);
  for (int row = 0; row < h / 2; row++) {
    memcpy(dst + ys + (size_t)row * (w / 2), pic->u + (size_t)row * (pic->stride / 2), w / 2);
    memcpy(dst + ys + cs + (size_t)row * (w / 2), pic->v + (size_t)row * (pic->stride / 2), w / 2);
  }
}

static void inter_picture(const encoder_state_t *state)
{
  const videoframe_t *frame = state->tile->frame;
  const kvz_config *cfg = &state->encoder_control->cfg;
  const int w = frame->width, h = frame->height, wc = (w + 63) / 64, hc = (h + 63) / 64;
  const size_t bytes = (size_t)w * h * 3 / 2, cells = (size_t)(w / 4) * (h / 4), ncoeff = (size_t)wc * hc * KVZ_HIP_CTU_COEFFS;
  if (g_inter.w != w || g_inter.h != h) {
    kvz_hip_host_free(g_inter.src); kvz_hip_host_free(g_inter.ref); kvz_hip_host_free(g_inter.rec);
    kvz_hip_host_free(g_inter.ref_cu); kvz_hip_host_free(g_inter.cu); kvz_hip_host_free(g_inter.coeff);
    kvz_hip_dev_free(g_inter.d_src); kvz_hip_dev_free(g_inter.d_ref); kvz_hip_dev_free(g_inter.d_rec);
    kvz_hip_dev_free(g_inter.d_ref_cu); kvz_hip_dev_free(g_inter.d_cu); kvz_hip_dev_free(g_inter.d_coeff);
    g_inter.src = kvz_hip_host_alloc(bytes); g_inter.ref = kvz_hip_host_alloc(bytes); g_inter.rec = kvz_hip_host_alloc(bytes);
    g_inter.ref_cu = kvz_hip_host_alloc(cells * sizeof(kvz_hip_cu_info)); g_inter.cu = kvz_hip_host_alloc(cells * sizeof(kvz_hip_cu_info));
    g_inter.coeff = kvz_hip_host_alloc(ncoeff * sizeof(int16_t));
    g_inter.d_src = kvz_hip_dev_alloc(bytes); g_inter.d_ref = kvz_hip_dev_alloc(bytes); g_inter.d_rec = kvz_hip_dev_alloc(bytes);
    g_inter.d_ref_cu = kvz_hip_dev_alloc(cells * sizeof(kvz_hip_cu_info)); g_inter.d_cu = kvz_hip_dev_alloc(cells * sizeof(kvz_hip_cu_info));
    g_inter.d_coeff = kvz_hip_dev_alloc(ncoeff * sizeof(int16_t));
    if (!g_inter.src || !g_inter.ref || !g_inter.rec || !g_inter.ref_cu || !g_inter.cu || !g_inter.coeff || !g_inter.d_src || !g_inter.d_ref || !g_inter.d_rec ||
        !g_inter.d_ref_cu || !g_inter.d_cu || !g_inter.d_coeff) { fprintf(stderr, "search_lcu_hip: out of memory\n"); abort(); }
    g_inter.w = w; g_inter.h = h;
  }
  tight_planes(g_inter.src, frame->source, w, h);
  tight_planes(g_inter.ref, state->frame->ref->images[0], w, h);
  /* what the search reads of the reference picture's CUs (inter.c:1204-1260 temporal candidates, search_inter.c:1286-1339 the starting point): type and motion */
  const cu_array_t *rca = state->frame->ref->cu_arrays[0];
  for (int y = 0; y < h; y += 4)
    for (int x = 0; x < w; x += 4) {
      const cu_info_t *c = kvz_cu_array_at_const(rca, x, y);
      kvz_hip_cu_info *o = &g_inter.ref_cu[(size_t)(y / 4) * (w / 4) + x / 4];
      memset(o, 0, sizeof *o);
      o->type = c->type; o->depth = c->depth; o->tr_depth = c->tr_depth; o->cbf = c->cbf;
      if (c->type == CU_INTER) {
        o->skipped = c->skipped; o->merged = c->merged; o->merge_idx = c->merge_idx; o->mv_dir = c->inter.mv_dir;
        for (int l = 0; l < 2; l++) { o->mv_ref[l] = c->inter.mv_ref[l]; o->mv[l][0] = c->inter.mv[l][0]; o->mv[l][1] = c->inter.mv[l][1]; }
      } else {
        o->mode = (uint8_t)c->intra.mode;
      }
    }
  kvz_hip_dev_upload(g_inter.d_src, g_inter.src, bytes);
  kvz_hip_dev_upload(g_inter.d_ref, g_inter.ref, bytes);
  kvz_hip_dev_upload(g_inter.d_ref_cu, g_inter.ref_cu, cells * sizeof(kvz_hip_cu_info));
  kvz_hip_inter_params prm;
  memset(&prm, 0, sizeof prm);
  prm.struct_size = sizeof prm;
  prm.qp = state->qp; prm.poc = state->frame->poc;
  prm.mv_constraint = cfg->owf && cfg->wpp;  /* search_inter.c:75-152 */
  prm.sao = cfg->sao_type != 0; prm.deblock = cfg->deblock_enable != 0;
  prm.fme_level = cfg->fme_level; prm.pu_depth_inter_max = cfg->pu_depth_inter.max[0]; prm.no_wpp = !cfg->wpp;
  prm.fast_residual_cost = cfg->fast_residual_cost_limit;
  const int rc = kvz_hip_dev_inter_ctu_pass(g_inter.d_src, g_inter.d_ref, g_inter.d_ref_cu, g_inter.d_rec, g_inter.d_cu, g_inter.d_coeff, w, h, 1, &prm);
  if (rc != 0) { fprintf(stderr, "search_lcu_hip: the inter CTU pass failed (%d)\n", rc); abort(); }
  kvz_hip_dev_download(g_inter.rec, g_inter.d_rec, bytes);
  kvz_hip_dev_download(g_inter.cu, g_inter.d_cu, cells * sizeof(kvz_hip_cu_info));
  kvz_hip_dev_download(g_inter.coeff, g_inter.d_coeff, ncoeff * sizeof(int16_t));
  {  /* KVZ_HIP_INTER_TRACE=<file>: B pictures searched on the device so far */
    static int pictures;
    const char *trace = getenv("KVZ_HIP_INTER_TRACE");
    pictures++;
    if (trace) { FILE *f = fopen(trace, "w"); if (f) { fprintf(f, "%d\n", pictures); fclose(f); } }
  }
}

static void copy_rec_and_coeff(encoder_state_t *state, int x, int y, int w, int h, const uint8_t *rec, const int16_t *coeff)
{
  videoframe_t *frame = state->tile->frame;
  const size_t ys = (size_t)w * h, cs = ys / 4;
  /* reconstruction before deblocking (copy_lcu_to_cu_data) */
  for (int row = 0; row < 64 && y + row < h; row++) {
    const int n = x + 64 <= w ? 64 : w - x;
    memcpy(&frame->rec->y[x + (size_t)(y + row) * frame->rec->stride], rec + (size_t)(y + row) * w + x, n);
  }
  for (int row = 0; row < 32 && y / 2 + row < h / 2; row++) {
    const int n = x / 2 + 32 <= w / 2 ? 32 : w / 2 - x / 2;
    memcpy(&frame->rec->u[x / 2 + (size_t)(y / 2 + row) * (frame->rec->stride / 2)], rec + ys + (size_t)(y / 2 + row) * (w / 2) + x / 2, n);
    memcpy(&frame->rec->v[x / 2 + (size_t)(y / 2 + row) * (frame->rec->stride / 2)], rec + ys + cs + (size_t)(y / 2 + row) * (w / 2) + x / 2, n);
  }
  /* coefficients (copy_coeffs): lcu_t z-order, the layout the library returns */
  memcpy(state->coeff->y, coeff, 4096 * sizeof(int16_t));
  memcpy(state->coeff->u, coeff + 4096, 1024 * sizeof(int16_t));
  memcpy(state->coeff->v, coeff + 5120, 1024 * sizeof(int16_t));
}

static void search_lcu_inter(encoder_state_t *state, int x, int y)
{
  videoframe_t *frame = state->tile->frame;
  pthread_mutex_lock(&g_lock);
  while (g_inter.busy) pthread_cond_wait(&g_cond, &g_lock);
  if (g_inter.num != state->frame->num || g_inter.owner != (const void *)frame || !g_inter.ready) {
    g_inter.busy = 1; g_inter.ready = 0; g_inter.num = state->frame->num; g_inter.owner = frame;
    pthread_mutex_unlock(&g_lock);
    inter_picture(state);
    pthread_mutex_lock(&g_lock);
    g_inter.busy = 0; g_inter.ready = 1;
    pthread_cond_broadcast(&g_cond);
  }
  pthread_mutex_unlock(&g_lock);
  const int w = g_inter.w, h = g_inter.h, wc = (w + 63) / 64;
  /* CU info (kvz_cu_array_copy_from_lcu): one cu_info_t per 4x4 unit, as search_cu leaves them (search.c:1000-1063, lcu_fill_inter / lcu_fill_cbf) */
  for (int yy = 0; yy < 64 && y + yy < h; yy += 4)
    for (int xx = 0; xx < 64 && x + xx < w; xx += 4) {
      const kvz_hip_cu_info *c = &g_inter.cu[(size_t)((y + yy) / 4) * (w / 4) + (x + xx) / 4];
      cu_info_t *cu = kvz_cu_array_at(frame->cu_array, x + xx, y + yy);
      memset(cu, 0, sizeof *cu);
      cu->type = c->type; cu->depth = c->depth; cu->part_size = SIZE_2Nx2N; cu->tr_depth = c->tr_depth; cu->cbf = c->cbf; cu->qp = (uint8_t)state->qp;
      if (c->type == CU_INTER) {
        cu->skipped = c->skipped; cu->merged = c->merged; cu->merge_idx = c->merge_idx;
        cu->inter.mv_dir = c->mv_dir; cu->inter.mv_cand0 = c->mv_cand[0]; cu->inter.mv_cand1 = c->mv_cand[1];
        for (int l = 0; l < 2; l++) { cu->inter.mv_ref[l] = c->mv_ref[l]; cu->inter.mv[l][0] = c->mv[l][0]; cu->inter.mv[l][1] = c->mv[l][1]; }
      } else {
        cu->intra.mode = (int8_t)c->mode; cu->intra.mode_chroma = (int8_t)c->mode;
      }
    }
  copy_rec_and_coeff(state, x, y, w, h, g_inter.rec, g_inter.coeff + (size_t)((y / 64) * wc + x / 64) * KVZ_HIP_CTU_COEFFS);
}

/* The LCU's part of a picture whose slice data the device has written: CU info (depth, modes, partition) for the loop filters and for the neighbours' sake, the
 * reconstruction -- no levels, no coded block flags: their only reader, kvz_encode_coding_tree, does not run for this picture (deblocking of intra CUs is strength 2
 * whatever the flags, filter.c:405-431). */
static void search_lcu_without_levels(encoder_state_t *state, picture_result *r, int x, int y)
{
  videoframe_t *frame = state->tile->frame;
  const int w = r->width, h = r->height, w8 = w / 8;
  for (int yy = 0; yy < 64 && y + yy < h; yy += 8)
    for (int xx = 0; xx < 64 && x + xx < w; xx += 8) {
      const int depth = r->depth[((y + yy) / 8) * w8 + (x + xx) / 8], mode = r->mode[((y + yy) / 8) * w8 + (x + xx) / 8];
      const int nxn = r->model.search_nxn && r->part[((y + yy) / 8) * w8 + (x + xx) / 8];
      for (int sy = 0; sy < 8; sy += 4)
        for (int sx = 0; sx < 8; sx += 4) {
          cu_info_t *cu = kvz_cu_array_at(frame->cu_array, x + xx + sx, y + yy + sy);
          const int pm = nxn ? r->mode4[((y + yy + sy) / 4) * (w / 4) + (x + xx + sx) / 4] : mode;
          memset(cu, 0, sizeof *cu);
          cu->type = CU_INTRA; cu->depth = nxn ? 3 : depth; cu->part_size = nxn ? SIZE_NxN : SIZE_2Nx2N; cu->tr_depth = nxn ? 4 : (depth > 0 ? depth : 1);
          cu->qp = (uint8_t)state->qp; cu->intra.mode = (int8_t)pm; cu->intra.mode_chroma = (int8_t)pm;
        }
    }
  static const int16_t no_levels[KVZ_HIP_CTU_COEFFS];
  copy_rec_and_coeff(state, x, y, w, h, r->rec, no_levels);
  pthread_mutex_lock(&g_lock);
  r->outstanding--;
  pthread_mutex_unlock(&g_lock);
}

/* kvz_encode_coding_tree (encode_coding_tree.c:745) is what encoder_state_worker_encode_lcu_bitstream spends its time in; for pictures the device has coded it has
 * nothing to do (the few bins the worker still writes around it -- end_of_slice_segment_flag, the substream's flush -- go into streams that are replaced below) */
void __wrap_kvz_encode_coding_tree(encoder_state_t *const state, uint16_t x, uint16_t y, uint8_t depth)
{
  if (g_entropy > 0 && depth == 0) {
    const videoframe_t *frame = state->tile->frame;
    int coded = 0;
    pthread_mutex_lock(&g_lock);
    for (int i = 0; i < g_n_slots && !coded; i++) coded = g_slots[i]->ent_hold && g_slots[i]->frame == frame && g_slots[i]->num == state->frame->num;
    pthread_mutex_unlock(&g_lock);
    if (coded) return;
  }
  __real_kvz_encode_coding_tree(state, x, y, depth);
}

/* the leaves of a frame's encoder-state tree, in the order the bitstream takes them: replace what the host coders wrote by the device's substreams */
static void replace_leaf_streams(encoder_state_t *state)
{
  if (!state->is_leaf) {
    for (int i = 0; state->children[i].encoder_control; ++i) replace_leaf_streams(&state->children[i]);
    return;
  }
  const videoframe_t *frame = state->tile->frame;
  picture_result *r = NULL;
  pthread_mutex_lock(&g_lock);
  for (int i = 0; i < g_n_slots && !r; i++)
    if (g_slots[i]->ent_hold && g_slots[i]->frame == frame && g_slots[i]->num == state->frame->num) r = g_slots[i];
  while (r && !r->ent_ready) pthread_cond_wait(&g_cond, &g_lock);  /* the leader is still coding the batch this picture was in */
  pthread_mutex_unlock(&g_lock);
  if (!r) return;
  const int row = (state->type == ENCODER_STATE_TYPE_WAVEFRONT_ROW && !r->model.no_wpp) ? state->wfrow->lcu_offset_y : 0;
  size_t at = 0;
  for (int k = 0; k < row; k++) at += r->ent_sizes[k];
  kvz_bitstream_clear(&state->stream);
  for (uint32_t k = 0; k < r->ent_sizes[row]; k++) kvz_bitstream_writebyte(&state->stream, r->ent[at + k]);  /* the bytes carry their emulation prevention already */
}
static void release_entropy_slots(encoder_state_t *state)
{
  if (!state->is_leaf) {
    for (int i = 0; state->children[i].encoder_control; ++i) release_entropy_slots(&state->children[i]);
    return;
  }
  pthread_mutex_lock(&g_lock);
  for (int i = 0; i < g_n_slots; i++)
    if (g_slots[i]->ent_hold && g_slots[i]->outstanding == 0 && g_slots[i]->frame == state->tile->frame && g_slots[i]->num == state->frame->num) { g_slots[i]->ent_hold = 0; g_slots[i]->ent_ready = 0; }
  pthread_mutex_unlock(&g_lock);
}
/* kvz_encoder_state_worker_write_bitstream (encoder_state-bitstream.c:1138) runs once per frame when all its LCUs are through: the slice header it writes takes the entry
 * points from the leaves' streams, so they are swapped first */
void __wrap_kvz_encoder_state_worker_write_bitstream(void *opaque)
{
  encoder_state_t *state = (encoder_state_t *)opaque;
  if (g_entropy > 0) replace_leaf_streams(state);
  __real_kvz_encoder_state_worker_write_bitstream(opaque);
  if (g_entropy > 0) release_entropy_slots(state);
}

void __wrap_kvz_search_lcu(encoder_state_t *const state, const int x, const int y, const yuv_t *const hor_buf, const yuv_t *const ver_buf)
{
  const int path = eligible(state);
  if (!path) { __real_kvz_search_lcu(state, x, y, hor_buf, ver_buf); return; }
  /* what kvz_search_lcu leaves in state->search_cabac for the stages after it: a counting-mode copy of the row's contexts taken at the start of
   * the LCU (search.c:1211-1212), update flag off as search_cu leaves it -- kvz_sao_search_lcu prices its mode bits on it (sao.c:52-177) */
  memcpy(&state->search_cabac, &state->cabac, sizeof(cabac_data_t));
  state->search_cabac.only_count = 1;
  state->search_cabac.update = 0;
  if (path == 2) { search_lcu_inter(state, x, y); return; }
  picture_result *r = picture_of(state);
  if (r->ent_hold) { search_lcu_without_levels(state, r, x, y); return; }
  videoframe_t *frame = state->tile->frame;
  const int w = r->width, h = r->height, w8 = w / 8, wc = (w + 63) / 64;
  const size_t ys = (size_t)w * h, cs = ys / 4;
  const int16_t *coeff = r->coeff + (size_t)((y / 64) * wc + x / 64) * KVZ_HIP_CTU_COEFFS;
  const int16_t *plane[3] = { coeff, coeff + 4096, coeff + 5120 };

  /* CU info (kvz_cu_array_copy_from_lcu): one cu_info_t per 4x4, every unit of a CU alike; coded block flags per transform unit
   * as kvz_intra_recon_cu leaves them (intra.c:623-696): the unit's bit at its depth, and for a 64x64 CU the depth-0 bit on its
   * first unit when any of the four has the plane coded */
  for (int yy = 0; yy < 64 && y + yy < h; yy += 8)
    for (int xx = 0; xx < 64 && x + xx < w; xx += 8) {
      const int depth = r->depth[((y + yy) / 8) * w8 + (x + xx) / 8], mode = r->mode[((y + yy) / 8) * w8 + (x + xx) / 8];
      if (r->model.search_nxn && r->part[((y + yy) / 8) * w8 + (x + xx) / 8]) {
        /* an NxN CU (search.c:691-700, 794-800): four 4x4 PUs with a mode and a luma transform block each (tr_depth 4), the 4x4 chroma blocks with the first
         * PU (transform.c:306-312); coded block flags at depth 4 on the unit that owns the block */
        for (int j = 0; j < 4; j++) {
          const int sx = 4 * (j & 1), sy = 4 * (j >> 1), pm = r->mode4[((y + yy + sy) / 4) * (w / 4) + (x + xx + sx) / 4];
          cu_info_t *cu = kvz_cu_array_at(frame->cu_array, x + xx + sx, y + yy + sy);
          memset(cu, 0, sizeof *cu);
          cu->type = CU_INTRA; cu->depth = 3; cu->part_size = SIZE_NxN; cu->tr_depth = 4; cu->qp = (uint8_t)state->qp;
          cu->intra.mode = (int8_t)pm; cu->intra.mode_chroma = (int8_t)pm;
          if (any_level(plane[0] + zorder16(xx + sx, yy + sy), 16)) cbf_set(&cu->cbf, 4, COLOR_Y);
          if (j == 0) {
            if (any_level(plane[1] + zorder16(xx / 2, yy / 2), 16)) cbf_set(&cu->cbf, 4, COLOR_U);
            if (any_level(plane[2] + zorder16(xx / 2, yy / 2), 16)) cbf_set(&cu->cbf, 4, COLOR_V);
          }
        }
        continue;
      }
      const int td = depth < 1 ? 1 : depth, tw = 64 >> td, tx = xx & ~(tw - 1), ty = yy & ~(tw - 1), cw = td == 3 ? 4 : tw / 2;
      uint16_t cbf = 0;
      for (int c = 0; c < 3; c++) {
        const int n = c ? cw * cw : tw * tw;
        const int16_t *lv = plane[c] + (c ? zorder16(tx / 2, ty / 2) : zorder16(tx, ty));
        if (any_level(lv, n)) cbf_set(&cbf, td, (color_t)c);
      }
      if (depth == 0 && tx == 0 && ty == 0)
        for (int c = 0; c < 3; c++) {
          int any = 0;
          for (int q = 0; q < 4; q++) any |= any_level(plane[c] + (c ? q * 256 : q * 1024), c ? 256 : 1024);
          if (any) cbf_set(&cbf, 0, (color_t)c);
        }
      for (int sy = 0; sy < 8; sy += 4)
        for (int sx = 0; sx < 8; sx += 4) {
          cu_info_t *cu = kvz_cu_array_at(frame->cu_array, x + xx + sx, y + yy + sy);
          memset(cu, 0, sizeof *cu);
          cu->type = CU_INTRA; cu->depth = depth; cu->part_size = SIZE_2Nx2N; cu->tr_depth = depth > 0 ? depth : 1;
          cu->cbf = cbf; cu->qp = (uint8_t)state->qp;
          cu->intra.mode = (int8_t)mode; cu->intra.mode_chroma = (int8_t)mode;
        }
    }
  copy_rec_and_coeff(state, x, y, w, h, r->rec, coeff);
  pthread_mutex_lock(&g_lock);
  r->outstanding--;  /* this LCU is done with the slot's buffers */
  pthread_mutex_unlock(&g_lock);
}
