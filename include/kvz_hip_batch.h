/*
 * kvz_hip_batch.h -- batched, device-resident entry points of libkvz_hip.so (group 3 of kvz_hip.h).
 *
 * The per-call strategy functions (kvz_hip.h) are bit-exact drop-ins but pay a PCIe round trip per call; kvazaar
 * makes ~3 500 such calls per CTU.  The throughput path keeps whole batches of frames in HBM and runs kvazaar's
 * per-CTU flow (search.c:1209 kvz_search_lcu and everything below it that SURVEY.md section 8a lists for the
 * all-intra `ultrafast` configuration: angular / planar / DC prediction, SATD mode costs, DCT 4..32, quantisation,
 * dequantisation, IDCT, reconstruction, SSD and fast coefficient cost) as ONE workgroup per CTU out of LDS.
 *
 * Frame independence: with `-p 1` every frame is an IDR picture (encoderstate.c:1599-1620), so a batch of N frames
 * is N independent problems; inside a frame CTU (x, y) needs its left, above and above-right neighbours, i.e. the
 * WPP order of encoderstate.c:793-903.  One persistent kernel launch per call draws the CTUs of EVERY frame of the batch from a
 * ticket list in that order (anti-diagonals x + 2y = const; raster order per picture when the model says no WPP) and workgroups
 * hand results to each other through per-CTU border records.
 *
 * Decisions use kvazaar's own cost formulas, in double precision with the reference's operation order, on CABAC contexts
 * that evolve as kvazaar's do (kvz_hip_intra_cost_model): the reconstruction, CU quadtree, modes and coefficients
 * are those of `kvazaar --preset ultrafast -p 1`, picture for picture (tests/test_encoder_parity.py checks the pass against
 * digests of the reference CLI's --debug output).  The checker for every intermediate is oracle/kvz_oracle_ctu.c.
 */
#ifndef KVZ_HIP_BATCH_H_
#define KVZ_HIP_BATCH_H_

#include "kvz_hip_types.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct kvz_hip_batch kvz_hip_batch; /* device buffers for n_frames pictures of width x height (yuv420p, 8 bit) */

/* width and height must be multiples of 8 (kvazaar pads its input the same way).  NULL on failure. */
kvz_hip_batch *kvz_hip_batch_create(int width, int height, int n_frames);
/* ... on a given device (0 .. kvz_hip_device_count() - 1; -1 = the calling thread's current one, which is what kvz_hip_batch_create does).  The reference runs the
 * tiles of a picture as jobs of ONE process (encoderstate.c:944-1013 builds one sub-encoder per tile, threadqueue.c:275-355 runs them on its workers): a host that
 * wants tile i on GPU i % count creates tile i's batch with this and calls it from any thread -- every entry point taking a kvz_hip_batch binds the calling thread
 * to the batch's device first, and its buffers, stream and kernels stay there.  NULL when the device does not exist. */
kvz_hip_batch *kvz_hip_batch_create_on(int device, int width, int height, int n_frames);
void           kvz_hip_batch_destroy(kvz_hip_batch *b);

/* Host <-> HBM.  Planes are tightly packed (stride = width; chroma width/2 x height/2). */
void kvz_hip_batch_upload(kvz_hip_batch *b, int frame, const uint8_t *y, const uint8_t *u, const uint8_t *v);
/* All n_frames pictures at once, asynchronously: `src` holds them back to back, each Y | U | V tightly packed (kvz_image_alloc's planar buffer, image.c:62-95),
 * preferably in kvz_hip_host_alloc'ed memory.  The copy runs on a queue of its own and starts when the batch's LAST pass has ended -- the pass is the only reader
 * of the source pictures (SAO's statistics aside: not while kvz_hip_batch_loop_filters with sao is in flight) -- so it overlaps the batch's own deblocking and
 * entropy coding and another batch's pass; the batch's NEXT kvz_hip_intra_frames waits for it.  `src` must stay untouched until that pass has been synced.
 * What a host feeding the device from its reader thread (encoder.c / input frame queue) does per batch instead of n_frames synchronous uploads.
 * All batches of a device share ONE upload queue.  The HIP runtime runs a process's streams on GPU_MAX_HW_QUEUES hardware queues (default 4) and streams sharing one
 * wait for each other: a double-buffered chain (two batches, the entropy coder's side stream, the upload queue, the calling thread's own stream) needs more, so the
 * library sets GPU_MAX_HW_QUEUES=8 when it is loaded unless the variable is already set -- effective when that happens before the process's first HIP call. */
void kvz_hip_batch_upload_all_async(kvz_hip_batch *b, const uint8_t *src);
/* Any output pointer may be NULL.  coeff: KVZ_HIP_CTU_COEFFS int16 per CTU (raster CTU order, lcu_t z-order inside);
 * cu_depth / cu_mode: one byte per 8x8 block (raster, stride width/8); ctu_cost: one double per CTU.
 * Returns 0, or -1 when the run that produced the data was invalid (see kvz_hip_batch_sync). */
int  kvz_hip_batch_download(kvz_hip_batch *b, int frame, uint8_t *rec_y, uint8_t *rec_u, uint8_t *rec_v, int16_t *coeff,
                            uint8_t *cu_depth, uint8_t *cu_mode, double *ctu_cost);

/* After a pass with model.search_nxn: cu_part, one byte per 8x8 block (raster, stride width/8): 1 = the CU is coded as four 4x4 prediction units (part_size NxN);
 * cu_mode4, the luma mode of every 4x4 unit (raster, stride width/4; four equal entries for a 2Nx2N CU -- cu_mode of kvz_hip_batch_download holds the first PU's).
 * Either may be NULL.  Returns 0, or -1 when the run was invalid or no such pass has run on the batch. */
int  kvz_hip_batch_download_partitions(kvz_hip_batch *b, int frame, uint8_t *cu_part, uint8_t *cu_mode4);

/* Pinned host memory (hipHostMalloc) for the asynchronous transfers below; plain malloc'ed buffers work too but serialise. */
void *kvz_hip_host_alloc(size_t bytes);
void  kvz_hip_host_free(void *p);
/* Queues the download of what the entropy coder needs of EVERY frame of the batch (encode_coding_tree.c:745 reads the coefficients,
 * cu_info and -- for the next stages -- the reconstruction) on the batch's stream, behind whatever it holds: with pinned
 * destinations the copies overlap the kernels of other batches' streams (two batches alternating = double buffering).
 * Layouts as kvz_hip_batch_download, frames back to back; NULL skips a buffer.  kvz_hip_batch_sync() waits and validates. */
void kvz_hip_batch_download_all_async(kvz_hip_batch *b, uint8_t *rec, int16_t *coeff, uint8_t *cu_depth, uint8_t *cu_mode);

/* Work queued on `b` from now on starts only after everything queued on `other` so far has finished (both on the same device).  The
 * CTU pass is ONE persistent launch that takes every workgroup slot of the device until its tickets run out: a second batch's pass launched
 * right behind it moves into the slots as they free up, and the first batch's small follow-up kernels (deblocking, SAO, checksums) then wait
 * for the whole second pass -- and with them its download.  Double-buffered pipelines therefore order the next batch's pass after the
 * previous batch's follow-up kernels, and queue the download behind that point (bench.py chain_d2h): the copy engines then run during
 * the next pass. */
void kvz_hip_batch_order_after(kvz_hip_batch *b, kvz_hip_batch *other);
/* The share of the device this batch's persistent pass takes: `num` of every `den` workgroup slots per CU (1 <= num <= den; (1, 1) = all of them, the default).
 * Batches of different geometry that are meant to run side by side -- the two tile sizes of kvazaar's uniform tile grid (encoderstate.c:944-979: 960x1088 and
 * 960x1072 at 3840x2160 --tiles 4x2) -- each take their share; with the default the first launch fills the device and the second one only moves in when the
 * first one's workgroups retire, i.e. the two passes run one after the other, each on fewer serial CTU chains than the device has slots.  Takes effect with
 * the next kvz_hip_intra_frames. */
void kvz_hip_batch_set_device_share(kvz_hip_batch *b, int num, int den);

/* The hot path: search + reconstruct every CTU of every frame in the batch.  Asynchronous on the batch's stream;
 * kvz_hip_batch_sync() waits.  Returns the number of kernel launches issued (1; one per CTU anti-diagonal with the older
 * schedule behind KVZ_HIP_SCHED=wave), or -1 -- with a message on stderr and nothing launched -- for a model the batch cannot run (rdoq without coeff_cabac;
 * search_32x32 / rdoq / search_nxn / no_wpp under KVZ_HIP_SCHED=wave). */
int  kvz_hip_intra_frames(kvz_hip_batch *b, const kvz_hip_intra_cost_model *model);
/* Waits for the batch's stream.  Returns 0, or -1 when a CTU hand-off wait inside a pass timed out (workgroups wait for their
 * neighbours' results with a wall-clock bound -- 30 s per wait, KVZ_HIP_WAIT_MS overrides -- so that a lost hand-off cannot hang
 * the GPU): the batch's results are then invalid and every later sync / download / checksum call of the batch reports -1 too.
 * HIP runtime failures (no device, out of memory, launch failure) stay fatal, as in the per-call path. */
int  kvz_hip_batch_sync(kvz_hip_batch *b);
/* After a pass reported -1: once a hand-off wait has timed out the rest of that pass only drains its tickets (no CTU is searched on stale neighbour data, nothing
 * waits again) and the batch stays invalid -- sticky, so that no later call can mistake its contents for results -- until this call clears the condition.  The
 * pictures uploaded to the batch are untouched: the caller may simply run the pass again.  Returns 0. */
int  kvz_hip_batch_reset(kvz_hip_batch *b);
/* developer diagnostics: every CTU's hand-off flag (= the number of the last pass that completed it) into out [n_frames x CTUs per frame]; returns the number of the batch's last pass */
unsigned kvz_hip_batch_debug_flags(kvz_hip_batch *b, unsigned *out);
/* Device time of the launches of the last kvz_hip_intra_frames call, from HIP events recorded on the batch's own
 * stream around the launch sequence (milliseconds); call after kvz_hip_batch_sync(). */
float kvz_hip_batch_last_kernel_ms(kvz_hip_batch *b);
int   kvz_hip_batch_ctus_per_frame(const kvz_hip_batch *b);
/* Developer aid: per-stage shader-cycle counters of a -DKVZ_CTU_PROFILE build of the library (zeros otherwise). */
int   kvz_hip_batch_profile(kvz_hip_batch *b, unsigned long long *out, int n);

/* Deblocks the batch's reconstruction in place on the batch's stream (kvz_filter_deblock_lcu, filter.c:783, over every LCU of
 * every frame; see kvz_hip_dev_deblock_frames in kvz_hip_dev.h) using the CU depths the last kvz_hip_intra_frames() left.
 * kvazaar runs it right after an LCU's reconstruction (encoderstate.c:669-671); downloads after this call return the
 * deblocked picture.  The next kvz_hip_intra_frames() overwrites the reconstruction. */
void kvz_hip_batch_deblock(kvz_hip_batch *b, int qp, int beta_offset_div2, int tc_offset_div2);

/* Both in-loop filters the way the encoder runs them per LCU (encoderstate.c:669-682): deblocking (when `deblock`) and -- when `sao` --
 * the SAO parameter decision kvz_sao_search_lcu (sao.c:671, `--sao full`: edge and band) on the partly deblocked picture each LCU sees at
 * that moment, with the merge-left / merge-up choice and the bit costs on the two SAO contexts as the LCU's real syntax moves them
 * (encoderstate.c:467-552), followed by kvz_sao_reconstruct (sao.c:302-361) of every LCU.  On the batch's stream; afterwards the batch's
 * reconstruction is the final picture (what `kvazaar --debug` writes), the decided parameters are read with kvz_hip_batch_sao_params().
 * model: the cost model of the CTU pass (lambda, entropy table, no_wpp, the SAO contexts' initial states in ctx_init).  With sao == 0 this
 * is kvz_hip_batch_deblock.  Three statistic sets per LCU (4 edge classes x 5 categories, 32 bands: {sum, count}) are gathered by one
 * workgroup per (LCU, plane); the decision chain -- serial per picture like the coder it follows -- runs one lane per picture. */
void kvz_hip_batch_loop_filters(kvz_hip_batch *b, const kvz_hip_intra_cost_model *model, int deblock, int beta_offset_div2, int tc_offset_div2, int sao);
/* The SAO parameters of one frame as the last kvz_hip_batch_loop_filters(..., sao = 1) decided them: one record per LCU in raster order
 * (chroma: U in offsets[0..4] / band_position[0], V in offsets[5..9] / band_position[1], like sao_info_t); merge: 0 none, 1 left, 2 up.
 * LCUs whose type is 0 (none) carry zeros in the other fields.  Any pointer may be NULL.  0 / -1 like kvz_hip_batch_sync. */
int  kvz_hip_batch_sao_params(kvz_hip_batch *b, int frame, kvz_hip_sao_params *luma, kvz_hip_sao_params *chroma, uint8_t *merge);

/* Picture-hash checksums (nal.c:73-86 kvz_image_checksum) of every frame's current reconstruction: host_out[3 * f + plane].
 * Queued behind whatever the batch's stream holds (CTU pass, deblocking) and waited for. */
int  kvz_hip_batch_checksums(kvz_hip_batch *b, uint32_t *host_out);  /* 0 / -1 like kvz_hip_batch_sync */

/* ... and the MD5 flavour of the picture hash (nal.c:88-101 kvz_image_md5, `--hash md5`): host_out[(3 f + plane) * 16 ..] = 16 digest bytes. */
int  kvz_hip_batch_md5(kvz_hip_batch *b, uint8_t *host_out);

/* Cost model of an I slice at `qp` (kvz_hip_intra_cost_model, adaptive contexts): HEVC context init values
 * (context.c:96-134), kvz_ctx_init (context.c:202-213), the HM entropy table (rdo.c:69-80), lambda of
 * rate_control.c:678-691.  coeff_weights = kvz_fast_coeff_get_weights(state) of the encoder (fast_coeff_cost.c:84-88). */
void kvz_hip_intra_cost_model_init(int qp, uint64_t coeff_weights, kvz_hip_intra_cost_model *model);
/* kvz_fast_coeff_get_weights for kvazaar's built-in table (fast_coeff_cost.h:48-101 packed by fast_coeff_cost.c:39-52); 0 for
 * QP >= MAX_FAST_COEFF_COST_QP (50), where kvazaar never uses the fast estimate (rdo.c:311-340). */
/* The entropy coder in its REAL mode -- what kvazaar's encoder_state_worker_encode_lcu_bitstream writes (encoderstate.c:676-745: SAO syntax, kvz_encode_coding_tree
 * encode_coding_tree.c:745, kvz_encode_coeff_nxn strategies/generic/encode_coding_tree-generic.c:40, the arithmetic coder cabac.c:85-270, the per-substream flush and
 * byte alignment) for every picture of the batch, on the device, from the results of the last kvz_hip_intra_frames (CU depths / modes / partitions, levels) and -- with
 * sao != 0 -- of the last kvz_hip_batch_loop_filters(..., sao = 1): I slices of the configurations the pass covers.  model: the pass's (ctx_init = the slice's initial
 * context states, no_wpp, search_nxn).  out (HOST, `capacity` bytes) receives the slice data: picture after picture, and inside a picture one substream per CTU row
 * (WPP; the rows' contexts start from the row above after its second CTU, encoderstate.c:763-771) or one for the whole picture (no_wpp); substream_bytes (HOST) their sizes,
 * n_frames x (CTU rows | 1) entries -- the entry points of the slice header (encoder_state-bitstream.c:935-954).  The bytes are those of kvazaar's per-row bitstream_t
 * objects, emulation prevention bytes included (bitstream.c:212-223): what follows the slice header in the NAL unit, verbatim.  Only these bytes cross PCIe instead of
 * the levels (12 KB per CTU).  Returns the total size, -1 on failure. */
long kvz_hip_batch_entropy_code(kvz_hip_batch *b, const kvz_hip_intra_cost_model *model, int sao, uint8_t *out, size_t capacity, uint32_t *substream_bytes);
/* ... when the batch's pictures are TILES of larger pictures (each tile a picture of its own for the pass, model.no_wpp as kvazaar codes tiles): not_last[f] != 0 (HOST,
 * n_frames entries) marks a tile that is not the last of its slice -- its substream ends in end_of_subset_one_bit instead of end_of_slice_segment_flag
 * (encoderstate.c:699-724).  not_last == NULL: every picture is a slice of its own (the function above). */
long kvz_hip_batch_entropy_code_tiles(kvz_hip_batch *b, const kvz_hip_intra_cost_model *model, int sao, const uint8_t *not_last, uint8_t *out, size_t capacity,
                                      uint32_t *substream_bytes);
/* ... and a pipeline's form of it: once the coder's first stage is through and its third is queued -- a few hundred wavefronts that each follow one substream's chain,
 * then the download: the device is nearly idle from there on --, the CTU pass of `next` is started with `next_model` (kvz_hip_intra_frames(next, next_model); another
 * batch on the same device), so that it runs beside the rest of this call.  Started any earlier, the pass -- one persistent launch that takes every workgroup slot it
 * finds -- would keep the coder's first stage, which wants the whole device, waiting until it is done.  next == NULL: kvz_hip_batch_entropy_code_tiles.
 * Returns: the total size; -1 the coder failed (as kvz_hip_batch_entropy_code_tiles); -2 the coder succeeded but kvz_hip_intra_frames(next, next_model) returned an
 * error; -3 next_model is NULL or of an unknown struct_size -- checked first, nothing has been queued on either batch.  In every other case, -1 and -2 included,
 * `next`'s pass HAS been queued exactly once when the call returns (at the coder's quiet moment, or at the end of the call when the coder never reached it): the
 * caller synchronises `next` (kvz_hip_batch_sync) whatever this call returned. */
long kvz_hip_batch_entropy_code_then(kvz_hip_batch *b, const kvz_hip_intra_cost_model *model, int sao, const uint8_t *not_last, uint8_t *out, size_t capacity,
                                     uint32_t *substream_bytes, kvz_hip_batch *next, const kvz_hip_intra_cost_model *next_model);
/* With `on` != 0 the three calls above return on this batch as soon as the slice data's download has been QUEUED on the batch's stream (the substream sizes are final and
 * the return value is the total; the bytes in `out` -- pinned memory, or the copy is not asynchronous -- are valid after kvz_hip_batch_sync(b) or the batch's next call
 * that synchronises its stream).  In a pipeline of two batches in turn the next batch's deblocking and coder then start while this batch's 0.4 MB per picture are still
 * on their way down (13 ms per 1 536 1080p pictures that sat between two passes).  The batch compacts into a device buffer of its own in this mode. */
void kvz_hip_batch_entropy_defer_download(kvz_hip_batch *b, int on);
uint64_t kvz_hip_default_coeff_weights(int qp);

#ifdef __cplusplus
}
#endif
#endif
