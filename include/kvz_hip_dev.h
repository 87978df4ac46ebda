/*
 * kvz_hip_dev.h -- device-resident batch entry points of the hot-path primitives.
 *
 * The drop-in entry points of kvz_hip.h keep the reference's synchronous, host-pointer semantics (one block per
 * call); these take DEVICE pointers and a block count, so that a caller that already keeps frames / CTU data in HBM
 * (the batched CTU pass, a future inter pass, the micro-benchmarks of SURVEY.md 8d) pays no staging.  Layouts are the
 * reference's: blocks are the contiguous n*n arrays kvz_sad_NxN / kvz_satd_NxN / kvz_dct_NxN take
 * (strategies-picture.h:115-131, strategies-dct.h:44), `count` of them back to back.
 *
 * All work is queued on the calling thread's stream; kvz_hip_dev_sync() waits for it.  Entry points that take a shape argument return 0, or -1 -- with a
 * message on stderr and nothing queued -- for a shape they do not have (HIP failures stay fatal, as everywhere in the library).  Results are bit-exact with the
 * per-call entry points (tests/test_gpu_dev.py compares against the oracle).
 */
#ifndef KVZ_HIP_DEV_H_
#define KVZ_HIP_DEV_H_

#include "kvz_hip_types.h"

#ifdef __cplusplus
extern "C" {
#endif

/* memory + stream helpers (abort with a message on failure, like the rest of the runtime) */
void *kvz_hip_dev_alloc(size_t bytes);
void  kvz_hip_dev_free(void *p);
void  kvz_hip_dev_upload(void *dev_dst, const void *host_src, size_t bytes);
void  kvz_hip_dev_download(void *host_dst, const void *dev_src, size_t bytes);
void  kvz_hip_dev_sync(void);
/* device-to-device copy on the same stream (the read + write streaming reference of bench_kernels.py) */
void  kvz_hip_dev_copy(void *dev_dst, const void *dev_src, size_t bytes);
/* Event pair on the calling thread's stream: milliseconds the device spent between start and stop. */
void  kvz_hip_dev_timer_start(void);
float kvz_hip_dev_timer_stop(void);

/* kvz_sad_NxN / kvz_satd_NxN (picture-generic.c:475-501, 252-340 + strategies-picture.h:53-69), n in {4 (satd only), 8, 16, 32, 64}:
 * out[i] = cost of blocks a[i], b[i].  2 n^2 bytes read per block, 4 written. */
int  kvz_hip_dev_sad_nxn(int n, const uint8_t *a, const uint8_t *b, int count, uint32_t *out);
int  kvz_hip_dev_satd_nxn(int n, const uint8_t *a, const uint8_t *b, int count, uint32_t *out);

/* kvz_dct_NxN / kvz_idct_NxN / 4x4 DST (dct-generic.c:559-630), 8-bit: `kind` = enum kvz_hip_transform_kind.
 * use_matrix_cores == 1 (the product path): 16- and 32-point blocks on the matrix cores (v_mfma_i32_16x16x32_i8 / 32x32x32_i8 on byte planes, exact integer
 * arithmetic; four 16-point blocks or one 32-point block per wavefront), 4- and 8-point blocks on the vector ALU, one lane per block row (v_dot2_i32_i16, transposes
 * through LDS).
 * == 2 (kept for A/B): the small sizes on the matrix cores too, 4 or 2 blocks on the diagonal of a 16 x 16 product; 16-point blocks on the vector ALU.
 * == 0: one lane per coefficient, two launches through `tmp` (count * n^2 int16 of scratch; may be NULL otherwise). */
int  kvz_hip_dev_transform(int kind, const int16_t *in, int16_t *tmp, int16_t *out, int count, int use_matrix_cores);

/* kvz_angular_pred (intra-generic.c:49-155): block i is predicted from ref_above + i * (2w+1) and ref_left + i * (2w+1)
 * into out + i * w * w.  (4w + 2) + w^2 bytes per block. */
int  kvz_hip_dev_angular_pred(int log2_width, int mode, const uint8_t *ref_above, const uint8_t *ref_left, int count, uint8_t *out);

/* Deblocking of all-intra, constant-QP pictures in place: kvz_filter_deblock_lcu (filter.c:783) over every LCU of every
 * frame.  frames = n_frames x [Y | U | V] tight planar 4:2:0 (the batch layout), cu_depth = n_frames x [H/8][W/8] CU depths
 * as the CTU pass returns them; beta / tc offsets are cfg.deblock_beta / cfg.deblock_tc (cfg.c: 0, 0).  Not a strategy in
 * the reference: kvazaar calls it per LCU from encoder_state_worker_encode_lcu_search
 * (encoderstate.c:669-671) between reconstruction and SAO.  ~3 w h bytes read and written per frame. */
void kvz_hip_dev_deblock_frames(uint8_t *frames, int width, int height, int n_frames, const uint8_t *cu_depth, int qp, int beta_offset_div2,
                                int tc_offset_div2);

/* Deblocking of pictures that contain inter CUs (P / B slices): as kvz_hip_dev_deblock_frames, with the boundary strengths of filter.c:405-493
 * -- 2 next to an intra CU, 1 on a transform edge with coded luma coefficients or across different motion (vectors >= 1 sample apart, other reference
 * pictures, the B-slice list permutations), 0 otherwise -- from one record per 4x4 unit (raster, stride width / 4, frames back to back): what the
 * filter reads of cu_info_t.  Edges are the transform-unit edges (64 >> tr_depth) and the prediction-unit edges of part_size on the 8x8 grid
 * (filter.c:202-257); chroma is only filtered at strength 2 (filter.c:610).  slice_is_b: the extra B-slice rules (filter.c:428-489). */
typedef struct kvz_hip_cu_dbk {
  uint8_t type;        /* cu_info_t::type: 1 = CU_INTRA, 2 = CU_INTER */
  uint8_t depth, tr_depth, part_size;
  uint8_t cbf_y;       /* cbf_is_set(cu->cbf, cu->tr_depth, COLOR_Y) */
  uint8_t mv_dir;      /* inter.mv_dir: 1 = L0, 2 = L1, 3 = both */
  int8_t  mv_ref[2];   /* inter.mv_ref: index into the reference lists */
  int16_t ref_id[2];   /* state->frame->ref_LX[list][mv_ref[list]]: which picture that is (only compared for equality) */
  int16_t mv[2][2];    /* inter.mv[list] = {x, y}, quarter samples */
} kvz_hip_cu_dbk;
void kvz_hip_dev_deblock_frames_inter(uint8_t *frames, int width, int height, int n_frames, const kvz_hip_cu_dbk *info, int qp, int beta_offset_div2,
                                      int tc_offset_div2, int slice_is_b);

/* Both loop filters of n pictures with inter prediction, in place -- what encoder_state_worker_encode_lcu_search runs after the search of every LCU (encoderstate.c:659-720:
 * kvz_filter_deblock_lcu, kvz_sao_search_lcu) and what kvz_sao_reconstruct applies afterwards: `rec` goes in as the inter CTU pass returned it and comes out as the picture the next
 * one predicts from.  src: the source pictures (the SAO decision measures against them); info: kvz_hip_dev_cu_dbk_from_info of the pass's CU records; qp: the picture QP (lambda and
 * the SAO contexts' initial states follow from it and slice_is_b); deblock / sao: cfg.deblock_enable, cfg.sao_type != 0 (`full`); no_wpp: one coder per picture.  The decision is
 * kvazaar's LCU by LCU: statistics on the partly deblocked picture exactly as the encoder sees it at that point, merge candidates, the SAO syntax priced on the evolving contexts of
 * the picture's substreams.  luma / chroma / merge (HOST pointers or NULL): the decisions, n_pictures x LCUs in raster order, as kvz_hip_batch_sao_params returns them.
 * Returns -1 on a bad argument. */
int  kvz_hip_dev_loop_filters_inter(const uint8_t *src, uint8_t *rec, int width, int height, int n_pictures, const kvz_hip_cu_dbk *info, int qp, int slice_is_b, int deblock,
                                    int beta_offset_div2, int tc_offset_div2, int sao, int no_wpp, kvz_hip_sao_params *luma, kvz_hip_sao_params *chroma, uint8_t *merge);

/* Integer-pel motion cost surface (the candidate scoring of the inter search, search_inter.c:1000-1005 -> kvz_image_calc_sad,
 * image.c:407): for block b = the bw x bw block of `cur` at (blk_xy[2b], blk_xy[2b+1]) and every displacement (dx, dy) in
 * [-range, range]^2,   out[b * side^2 + (dy + range) * side + dx + range] = kvz_image_calc_sad(cur, ref, x, y, x + dx, y + dy, bw, bw)
 * with side = 2 range + 1 -- the reference picture is edge-replicated outside the frame as image.c:279-397 does.  Both
 * pictures are width x height luma planes (stride = width); bw in {8, 16, 32, 64}, range <= 32, blocks inside the picture. */
int  kvz_hip_dev_sad_surface(const uint8_t *cur, const uint8_t *ref, int width, int height, int bw, int range, const int16_t *blk_xy, int count,
                             uint32_t *out);

/* Fractional motion search, the arithmetic of search_frac (search_inter.c:974-1130) for `count` prediction units of one picture pair:
 * for PU i -- the w x h block of `cur` at (x, y), w and h multiples of 8 up to 64 -- with integer-pel motion vector (mv_x, mv_y):
 *   out[i][0]            = kvz_satd_any_size(w, h, source block, reference block at the integer position)   (search_inter.c:1059)
 *   out[i][1 + 4 s + j]  = cost j of kvz_satd_any_size_quad over the four planes of step s                    (search_inter.c:1088-1118)
 * for every step s set in the bit mask `steps`: 0 = kvz_filter_hpel_blocks_hor_ver_luma (left, right, top, bottom), 1 = ..._hpel_blocks_diag_luma
 * (top-left, top-right, bottom-left, bottom-right), 2 / 3 = the quarter-pel functions around the half-pel offset (hpel_x, hpel_y) in
 * {-1, 0, 1}^2 (search_frac's sample_off_x / _y).  The reference window is read with clamped addressing, which is what
 * kvz_get_extended_block (ipol-generic.c:761-814) materialises.  Both pictures are width x height luma planes, stride = width.
 * `veryfast` (fme_level 2) uses steps = 3 in one call; quarter-pel presets call again with steps = 12 after choosing the half-pel offset. */
typedef struct kvz_hip_fme_pu {
  int16_t x, y, w, h;
  int16_t mv_x, mv_y;      /* integer-pel units (best_mv >> 2) */
  int8_t  hpel_x, hpel_y;  /* only read by steps 2 and 3 */
  int16_t reserved;
} kvz_hip_fme_pu;
#define KVZ_HIP_FME_COSTS 17
int  kvz_hip_dev_fme_costs(const uint8_t *cur, const uint8_t *ref, int width, int height, const kvz_hip_fme_pu *pus, int count, int max_pu_size, int steps, uint32_t *out);

/* Motion-compensated prediction (inter.c:371-575 inter_recon_unipred / kvz_inter_recon_bipred -> kvz_sample_quarterpel_luma(_hi),
 * kvz_sample_octpel_chroma(_hi), kvz_bipred_average): for PU i -- w x h luma samples at (x, y), multiples of 8 up to 64 -- the prediction
 * from reference list 0 (use[0]) and / or list 1 (use[1]) with quarter-pel motion vectors mv[list] = {x, y}, written into the PU's samples of
 * all three planes of `pred`.  ref0 / ref1 / pred: tight planar 4:2:0 pictures (Y | U | V) of width x height; references are read with
 * clamped addressing (what kvz_get_extended_block / inter_cp_with_ext_border materialise at the picture edge). */
typedef struct kvz_hip_mc_pu {
  int16_t x, y, w, h;
  int16_t mv[2][2];
  int8_t  use[2];
  int16_t reserved;
} kvz_hip_mc_pu;
int  kvz_hip_dev_inter_pred(const uint8_t *ref0, const uint8_t *ref1, uint8_t *pred, int width, int height, const kvz_hip_mc_pu *pus, int count, int max_pu_size);

/* cu_info_t (cu.h:130-170) of one 4x4 unit as the inter CTU pass reads and writes it: type 0 not set / 1 intra / 2 inter, the CU's depth, the intra mode,
 * tr_depth, the coded block flags in kvazaar's packing (cu.h:262-300: 5 depth bits per plane), the inter flags and motion.  Motion fields of a list mv_dir does
 * not use are 0 / 255. */
typedef struct kvz_hip_cu_info {
  uint8_t type, depth, mode, tr_depth; uint16_t cbf;
  uint8_t skipped, merged, merge_idx, mv_dir, mv_ref[2], mv_cand[2];
  int16_t mv[2][2];
} kvz_hip_cu_info;

/* The CTU pass of pictures with inter prediction -- search_cu of a B slice (search.c:646-1063) and everything below it, for picture k of n_pictures independent
 * sequences at once (BASELINE config 4: `--preset veryfast --gop lp-g4d3t1`, each B picture predicting from the previous picture of its sequence in both lists):
 *   src      [n] the pictures to encode, tight planar 4:2:0 (Y|U|V), width and height multiples of 8
 *   ref      [n] their reference pictures (the previous picture of each sequence AFTER its loop filters), same layout
 *   ref_cu   [n] the reference pictures' CU info, one record per 4x4 unit, raster order, stride width / 4 (temporal candidates, the motion search's starting point)
 *   rec      [n] out: the reconstruction before the loop filters
 *   cu       [n] out: CU info of the encoded pictures (what the next picture takes as ref_cu, and what deblocking reads)
 *   coeff    [n] out or NULL: KVZ_HIP_CTU_COEFFS quantised coefficients per CTU (raster CTU order; inside a CTU as lcu_coeff_t: Y | U | V, z-order)
 * All device pointers.  The pass is oracle/kvz_oracle_inter.inc's search_cu_b on the device, CTU for CTU identical to the reference encoder (tests/test_gpu_inter_ctu.py).
 * Coefficients are priced as kvz_get_coeff_cost does (rdo.c:311-340): kvz_fast_coeff_cost while the picture QP lies below params->fast_residual_cost (28 in BASELINE
 * config 4's preset, whose QP 22 runs its pictures at 21-25), the residual coder in counting mode on the search contexts from there on.  Returns -1 on a bad argument, -2 when a CTU
 * hand-off timed out. */
typedef struct kvz_hip_inter_params {
  uint32_t struct_size;        /* sizeof(kvz_hip_inter_params) of the caller's headers: set it after zeroing the struct; an unknown size is refused with -1 */
  int32_t qp;                  /* the picture's QP (state->frame->QP; kvz_oracle_lowdelay_qp states how kvazaar derives it from --qp and the GOP) */
  int32_t poc;                 /* picture order count inside the intra period (> 0); temporal AMVP candidates need poc > 1 (inter.c:1290) */
  int32_t mv_constraint;       /* cfg.owf && cfg.wpp */
  int32_t sao, deblock;        /* cfg.sao_type != 0, cfg.deblock_enable (the margin of that restriction) */
  int32_t fme_level;           /* cfg.fme_level (--subme) 0 .. 4: 4 `faster`, 2 `veryfast`, 0 `ultrafast` -- the steps search_frac takes (search_inter.c:1088) */
  int32_t pu_depth_inter_max;  /* 3 `veryfast`, 2 `ultrafast` */
  int32_t no_wpp;              /* one coder runs through the picture in raster order (--no-wpp) */
  int32_t fast_residual_cost;  /* cfg.fast_residual_cost_limit: 28 `ultrafast` .. `veryfast`, 0 `faster` -- below it (and below 50) coefficients are priced by kvz_fast_coeff_cost */
  /* Tiles (kvazaar --tiles CxR; encoderstate.c:944-979): the pictures handed to the pass are ONE TILE each -- an independent sub-picture for prediction, neighbours, contexts
   * and CTU order -- while `ref` / `ref_cu` are whole FRAMES of ref_width x ref_height (frame_bytes = ref_width * ref_height * 3 / 2, (ref_width / 4) * (ref_height / 4)
   * records): motion vectors may leave the tile (search_inter.c:94-187: mv-constraint none), reference samples are read at tile offset + position and replicated at the
   * FRAME's edges (inter.c:80-81, search_inter.c:217-218), the co-located record of the search's starting point likewise (search_inter.c:1286-1287).  The temporal merge /
   * AMVP candidates are read at the TILE-LOCAL position of the frame's array and checked against the frame's size, as the reference does (inter.c:836-905 takes x, y of
   * the tile and encoder_control->in.width).  All zero: the picture is the frame. */
  int32_t ref_width, ref_height, tile_x, tile_y;
  int32_t no_tmvp;             /* !cfg.tmvp_enable: no temporal merge / AMVP candidates (inter.c:1295-1302, 1471-1476) -- kvazaar switches TMVP off whenever tiles are used (cfg.c:920-975) */
} kvz_hip_inter_params;
int  kvz_hip_dev_inter_ctu_pass(const uint8_t *src, const uint8_t *ref, const kvz_hip_cu_info *ref_cu, uint8_t *rec, kvz_hip_cu_info *cu, int16_t *coeff, int width,
                                int height, int n_pictures, const kvz_hip_inter_params *params);
/* Threads and devices: the pass keeps its work memory (ticket list, done flags, contexts, the workgroups' level slabs) per CALLING THREAD and per DEVICE, like the
 * stream it is queued on -- two threads, or two devices of one process, never share it; calls of one thread are serialised on its stream.  `rec` is read back by the pass
 * itself (a finished CU is the intra reference of the next): it must not alias `ref`.  The struct grows at the end: zero it, set struct_size = sizeof(kvz_hip_inter_params), then fill it -- a caller
 * built against other headers than the library's is refused (-1) instead of handing the pass whatever lies behind a shorter struct. */
/* ... with every picture's own tile origin: tile_xy (DEVICE pointer, n_pictures x {x, y}, multiples of 8 inside the reference frame; NULL: params->tile_x / tile_y for
 * all) -- the tiles of one size of MANY places of the grid in one launch (without WPP a tile offers one CTU at a time: the launch needs that many more chains).
 * n_references (0: one per picture): `ref` / `ref_cu` hold that many frames and picture p predicts from frame p % n_references -- several tiles of the same frame. */
int  kvz_hip_dev_inter_ctu_pass_tiles(const uint8_t *src, const uint8_t *ref, const kvz_hip_cu_info *ref_cu, uint8_t *rec, kvz_hip_cu_info *cu, int16_t *coeff, int width,
                                      int height, int n_pictures, const kvz_hip_inter_params *params, const int32_t *tile_xy, int n_references);
/* milliseconds the kernel of the calling thread's last kvz_hip_dev_inter_ctu_pass took on the device (HIP events on its stream around the launch) */
float kvz_hip_dev_inter_kernel_ms(void);
/* The pass is a persistent launch of as many workgroups per CU as the kernel build it launches fits (its occupancy on the current device).  Passes that are to run
 * side by side -- the two tile sizes of a uniform tile grid, from two host threads -- each take a share: kvz_hip_dev_inter_set_share(parts) makes the CALLING THREAD's
 * later passes launch 1 / parts of what their kernel fits (at least one workgroup per CU; parts <= 1: all of it, the default).  Per thread, like the stream and the
 * scratch of the pass: nothing process-global changes.  ($KVZ_HIP_INTER_WG_PER_CU, a developer override of the count, is clamped to the occupancy.) */
void kvz_hip_dev_inter_set_share(int parts);
/* workgroups of the pass's kernel (the build without the residual coder's contexts) that fit one CU of the calling thread's device: informational */
int kvz_hip_dev_inter_slots_per_cu(void);
/* The slice data of n B pictures -- kvz_encode_coding_tree with the inter syntax (encode_coding_tree.c:745-900, kvz_encode_inter_prediction_unit :311-421, kvz_encode_mvd
 * :1062-1112), the residual coder and the arithmetic coder, as kvz_hip_batch_entropy_code does it for I pictures (kvz_hip_batch.h) -- from what the inter CTU pass left on
 * the device: cu (its CU records), ref_cu (the reference pictures' records: the temporal MV predictor), coeff (its levels; the pass must have been given a coeff buffer).
 * params: the pass's (qp, poc, no_wpp; sao != 0: the SAO syntax of the decisions the last kvz_hip_dev_loop_filters_inter(..., sao = 1) made on the same pictures).  The MV
 * predictors the MVDs are coded against are derived again from the records (kvz_inter_get_mv_cand_cua, inter.c:1330-1352).  out (HOST) / substream_bytes (HOST, n_pictures x
 * (CTU rows | 1)): as kvz_hip_batch_entropy_code.  Returns the total size, -1 on failure.
 * With sao != 0 the call reads the SAO records and merge flags that kvz_hip_dev_loop_filters_inter left in the CALLING THREAD's scratch of the current device: the two
 * calls must come from the same host thread, the filter call on these pictures directly before -- a thread that filtered other pictures in between codes THEIR SAO syntax,
 * a thread that never filtered gets -1. */
long kvz_hip_dev_entropy_code_inter(const kvz_hip_cu_info *cu, const kvz_hip_cu_info *ref_cu, const int16_t *coeff, int width, int height, int n_pictures,
                                    const kvz_hip_inter_params *params, uint8_t *out, size_t capacity, uint32_t *substream_bytes);
/* what the deblocking filter reads (kvz_hip_cu_dbk) of `count` CU records: type, depth, tr_depth, the luma coded block flag at tr_depth, motion */
void kvz_hip_dev_cu_dbk_from_info(const kvz_hip_cu_info *cu, int count, kvz_hip_cu_dbk *out);

/* The motion search of one reference picture for `count` prediction units, whole -- search_pu_inter_ref (search_inter.c:1237-1435) and the fractional
 * refinement of its result (search_inter.c:1866-1917 -> search_frac :974-1130) with every decision the reference takes on the way:
 *   the starting point (select_starting_point :285-312: the best of (0,0), the co-located motion of the previous picture and the single-list merge candidates),
 *   the early termination (early_terminate :425-486, `sensitive`: two rounds of a small cross, stop when a round gains less than 5 %),
 *   the hexagon search (hexagon_search :712-800, unlimited steps) -- every probe = check_mv_cost (:180-232): edge-replicated SAD (image.c:407), then the MVD
 *   bit cost of the cheaper AMVP predictor (calc_mvd_cost :381-423 / get_mvd_coding_cost :328-341) times lambda_sqrt, against the best so far with the
 *   reference's 0.001 guard --, for fme_level 0 the SATD re-pricing of the result (:1381-1393),
 *   for fme_level 2 the two half-pel steps (hor / ver neighbours, diagonal neighbours) with the truncation of the reference's `unsigned` cost accumulator, for
 *   fme_level 4 the two quarter-pel steps around the best half-pel position as well,
 *   and throughout the motion-vector restriction of overlapped pictures (fracmv_within_tile :75-152 with mv-constraint none: cfg.owf && cfg.wpp).
 * What the caller supplies per PU is what depends on the neighbourhood: the two AMVP predictors (kvz_inter_get_mv_cand), the merge candidates' motion
 * (kvz_inter_get_merge_cand; only candidates that use one list take part) and the co-located CU's motion.  cur / ref: width x height luma planes, stride = width.
 * One workgroup per PU; PUs are squares of 8, 16, 32 or 64 samples inside the picture, none larger than max_pu_size.  Returns -1 on a bad argument. */
typedef struct kvz_hip_me_pu {
  int16_t x, y, w, h;
  int16_t mv_cand[2][2];   /* AMVP predictors, quarter samples */
  int16_t start_mv[2];     /* motion of the co-located CU of the reference picture (search_inter.c:1286-1339), quarter samples; read when has_start */
  uint8_t has_start;
  uint8_t num_merge;       /* merge candidates, in list order */
  uint8_t merge_dir[5];    /* inter_merge_cand_t::dir: 1 = L0, 2 = L1, 3 = both (ignored by the search) */
  uint8_t reserved;
  int16_t merge_mv[5][2];  /* motion of the candidate's list (dir 1 or 2), quarter samples */
} kvz_hip_me_pu;
typedef struct kvz_hip_me_params {
  double  lambda_sqrt;     /* state->lambda_sqrt of the picture */
  int32_t mv_constraint;   /* cfg.owf && cfg.wpp */
  int32_t sao, deblock;    /* cfg.sao_type != 0, cfg.deblock_enable: the margin of that restriction */
  int32_t fme_level;       /* 0 .. 4: 0 (`ultrafast`), 2 (`veryfast`), 4 (`faster`) */
} kvz_hip_me_params;
typedef struct kvz_hip_me_result {
  int32_t mv[2];           /* after the integer search, quarter samples */
  int32_t mvp, valid;      /* select_mv_cand of that vector; valid = it becomes an AMVP candidate (vector allowed and a cost was found, :1404-1405) */
  double  cost, bits;
  int32_t frac_mv[2];      /* after search_frac; frac_valid 0: not refined (fme_level 0, no quarter-sample step possible, or the result not allowed) -- the integer result stands */
  int32_t frac_mvp, frac_valid;
  double  frac_cost, frac_bits;
} kvz_hip_me_result;
int  kvz_hip_dev_pu_search(const uint8_t *cur, const uint8_t *ref, int width, int height, const kvz_hip_me_pu *pus, int count, int max_pu_size,
                           const kvz_hip_me_params *params, kvz_hip_me_result *out);

/* SAO applied to whole pictures: kvz_sao_reconstruct (sao.c:302-361) for every CTU and plane of n_frames tight planar 4:2:0
 * frames.  in = the deblocked pictures, out = a different buffer of the same layout (SAO reads pre-SAO neighbours);
 * luma / chroma = n_frames x CTUs (raster order) parameter records, chroma carrying U in offsets[0..4] / band_position[0] and V
 * in offsets[5..9] / band_position[1] like sao_info_t (sao.h:55-63).  The parameter decision (sao.c:671 kvz_sao_search_lcu)
 * is not part of this entry point.  2 x 1.5 w h bytes per frame. */
void kvz_hip_dev_sao_frames(const uint8_t *in, uint8_t *out, int width, int height, int n_frames, const kvz_hip_sao_params *luma,
                            const kvz_hip_sao_params *chroma);

/* Picture-hash SEI checksums (nal.c:73-86 kvz_image_checksum): out[3 * f + p] = kvz_array_checksum of plane p of frame f
 * (nal-generic.c:57-82) for n_frames tight planar 4:2:0 frames; width a multiple of 8.  1.5 w h bytes read per frame. */
void kvz_hip_dev_picture_checksums(const uint8_t *frames, int width, int height, int n_frames, uint32_t *out);

/* Picture-hash MD5 (nal.c:88-101 kvz_image_md5, `--hash md5`): out[(3 f + p) * 16 ..] = the 16 digest bytes of plane p of frame f
 * (kvz_array_md5, nal-generic.c:41-55).  One serial chain per plane, one lane each: throughput comes from the number of planes in flight. */
void kvz_hip_dev_picture_md5(const uint8_t *frames, int width, int height, int n_frames, uint8_t *out);

/* Assembles a planar 4:2:0 picture from tile pictures: `tiles` (HOST memory) holds n records (x, y, w, h, slot); tile i's planar Y|U|V picture of w x h lies at
 * slots + slot * slot_bytes (device memory) and is pasted at (x, y) of the width x height frame (device memory).  One launch on `stream` (a hipStream_t; NULL = the
 * calling thread's stream of this library).  n <= 64.  Used by the reference-frame exchange of the tile-sharded inter configuration (kvazaar_amd/sharding.py). */
int kvz_hip_dev_paste_tiles(uint8_t *frame, int width, int height, const uint8_t *slots, long slot_bytes, const int32_t *tiles, int n, void *stream);

#ifdef __cplusplus
}
#endif
#endif
