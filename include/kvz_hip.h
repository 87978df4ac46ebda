/*
 * kvz_hip.h -- C ABI of libkvz_hip.so, the MI355X (gfx950) "hip" strategy for kvazaar's per-CTU hot path.
 *
 * The library is a drop-in for ONE thing: the function-pointer strategy layer of ultravideo/kvazaar v2.3.2
 * (src/strategyselector.h:99, src/strategies/strategies-{picture,dct,quant,intra,ipol,sao}.h).  Everything here is
 * extern "C", plain pointers and sizes; no kvazaar struct and no torch type crosses the boundary.  File:line
 * citations are relative to /root/reference/src and name the reference interface each entry point replaces.
 * INTEGRATION.md shows the registration shim (strategies/hip/[x]-hip.c) a kvazaar maintainer adds on their side.
 *
 * Three groups:
 *   1. Typedef-exact entry points -- can be handed to kvz_strategyselector_register() as they are
 *      (struct-free reference signatures: all of strategies-dct.h, strategies-intra.h, most of strategies-picture.h,
 *      fast_coeff_cost / coeff_abs_sum of strategies-quant.h).
 *   2. Flat entry points -- reference functions whose signature carries host structs (encoder_state_t,
 *      encoder_control_t, cu_info_t, lcu_t, sao_info_t, kvz_epol_args): same arguments with the structs replaced
 *      by the PODs of kvz_hip_types.h; the registration shim fills those PODs from the structs.
 *   3. Batched, device-resident entry points (kvz_hip_batch.h) -- where the throughput lives: whole frames of
 *      CTUs per launch, inputs already in HBM.
 *
 * Semantics of groups 1 and 2: synchronous, caller owns every buffer, ordinary pageable host
 * pointers with arbitrary alignment/stride, nothing retained, fully re-entrant (one HIP stream + pinned staging arena
 * per calling thread, created lazily; the strategy API has no init/teardown hook).  Results are bit-exact with the
 * reference's `generic` strategy.  There is NO CPU fallback: if no gfx950 device is usable the first call prints
 * the HIP error to stderr and abort()s (the reference's strategy functions have no error channel).
 */
#ifndef KVZ_HIP_H_
#define KVZ_HIP_H_

#include "kvz_hip_types.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- runtime ---------------------------------------------------------------------------------------------------- */
int         kvz_hip_device_count(void);        /* usable devices; 0 when there is none (no abort)                    */
int         kvz_hip_init(int device);          /* bind this process to `device` (default 0 / $KVZ_HIP_DEVICE); 1 = ok */
/* Devices and threads.  The first call that touches HIP picks the process DEFAULT device (kvz_hip_init's argument; < 0: $KVZ_HIP_DEVICE, else $LOCAL_RANK, else 0).
 * Every calling thread has a current device: the default until the thread selects another one with kvz_hip_set_thread_device() or calls an entry point that takes a
 * kvz_hip_batch (kvz_hip_batch.h), which binds the thread to the batch's device.  The per-call entry points of this header and the device-pointer entry points of
 * kvz_hip_dev.h run on the calling thread's current device with a stream, staging arenas and grow-only scratch of their own per (thread, device); a worker thread
 * that exits gives them back.  So ONE process can drive several GPUs -- kvazaar's tile threads (encoderstate.c:944-1013, threadqueue.c:275-355) with tile i on GPU
 * i % kvz_hip_device_count() -- and one process per GPU (bench.py --gpus N under torch.distributed.run) keeps working through the default. */
int         kvz_hip_set_thread_device(int device);  /* 1 = ok, 0 = no such device (the thread keeps its current one)         */
int         kvz_hip_thread_device(void);            /* the calling thread's current device                                    */
const char *kvz_hip_version(void);
unsigned long long kvz_hip_call_count(void); /* per-call entry points served so far (KVZ_HIP_STATS=1 prints it at exit) */

/* ---- 1. typedef-exact: strategies-picture.h ---------------------------------------------------------------------- */
/* reg_sad_func (strategies-picture.h:115-117), "reg_sad" (picture-generic.c:98) */
unsigned kvz_hip_reg_sad(const uint8_t *data1, const uint8_t *data2, int width, int height, unsigned stride1, unsigned stride2);
/* cost_pixel_nxn_func (:118), "sad_NxN" (picture-generic.c:475-501) and "satd_NxN" (:201-208, strategies-picture.h:53-69) */
unsigned kvz_hip_sad_4x4(const uint8_t *b1, const uint8_t *b2);
unsigned kvz_hip_sad_8x8(const uint8_t *b1, const uint8_t *b2);
unsigned kvz_hip_sad_16x16(const uint8_t *b1, const uint8_t *b2);
unsigned kvz_hip_sad_32x32(const uint8_t *b1, const uint8_t *b2);
unsigned kvz_hip_sad_64x64(const uint8_t *b1, const uint8_t *b2);
unsigned kvz_hip_satd_4x4(const uint8_t *b1, const uint8_t *b2);
unsigned kvz_hip_satd_8x8(const uint8_t *b1, const uint8_t *b2);
unsigned kvz_hip_satd_16x16(const uint8_t *b1, const uint8_t *b2);
unsigned kvz_hip_satd_32x32(const uint8_t *b1, const uint8_t *b2);
unsigned kvz_hip_satd_64x64(const uint8_t *b1, const uint8_t *b2);
/* cost_pixel_nxn_multi_func (:124): preds = kvz_pixel(*)[32*32], two candidates (picture-generic.c:369-402, 512-534) */
void kvz_hip_sad_4x4_dual(const uint8_t (*preds)[32 * 32], const uint8_t *orig, unsigned num_modes, unsigned *costs_out);
void kvz_hip_sad_8x8_dual(const uint8_t (*preds)[32 * 32], const uint8_t *orig, unsigned num_modes, unsigned *costs_out);
void kvz_hip_sad_16x16_dual(const uint8_t (*preds)[32 * 32], const uint8_t *orig, unsigned num_modes, unsigned *costs_out);
void kvz_hip_sad_32x32_dual(const uint8_t (*preds)[32 * 32], const uint8_t *orig, unsigned num_modes, unsigned *costs_out);
void kvz_hip_sad_64x64_dual(const uint8_t (*preds)[32 * 32], const uint8_t *orig, unsigned num_modes, unsigned *costs_out);
void kvz_hip_satd_4x4_dual(const uint8_t (*preds)[32 * 32], const uint8_t *orig, unsigned num_modes, unsigned *costs_out);
void kvz_hip_satd_8x8_dual(const uint8_t (*preds)[32 * 32], const uint8_t *orig, unsigned num_modes, unsigned *costs_out);
void kvz_hip_satd_16x16_dual(const uint8_t (*preds)[32 * 32], const uint8_t *orig, unsigned num_modes, unsigned *costs_out);
void kvz_hip_satd_32x32_dual(const uint8_t (*preds)[32 * 32], const uint8_t *orig, unsigned num_modes, unsigned *costs_out);
void kvz_hip_satd_64x64_dual(const uint8_t (*preds)[32 * 32], const uint8_t *orig, unsigned num_modes, unsigned *costs_out);
/* cost_pixel_any_size_func (:119-123), "satd_any_size" (strategies-picture.h:75-113) */
unsigned kvz_hip_satd_any_size(int width, int height, const uint8_t *block1, int stride1, const uint8_t *block2, int stride2);
/* cost_pixel_any_size_multi_func (:125), "satd_any_size_quad" (picture-generic.c:404-471, row-offset quirk kept) */
void kvz_hip_satd_any_size_quad(int width, int height, const uint8_t *const *preds /* ABI == const kvz_pixel ** */, int stride, const uint8_t *orig, int orig_stride,
                                unsigned num_modes, unsigned *costs_out, int8_t *valid);
/* pixels_calc_ssd_func (:127), ver_sad_func (:129-131), hor_sad_func (:132-134), pixel_var_func (:150) */
unsigned kvz_hip_pixels_calc_ssd(const uint8_t *ref, const uint8_t *rec, int ref_stride, int rec_stride, int width);
uint32_t kvz_hip_ver_sad(const uint8_t *pic_data, const uint8_t *ref_data, int32_t block_width, int32_t block_height, uint32_t pic_stride);
uint32_t kvz_hip_hor_sad(const uint8_t *pic_data, const uint8_t *ref_data, int32_t width, int32_t height, uint32_t pic_stride,
                         uint32_t ref_stride, uint32_t left, uint32_t right);
double   kvz_hip_pixel_var(const uint8_t *buf, uint32_t len);
/* get_optimized_sad_func (:128, optimized_sad_func_ptr_t.h:13-17): width-specialised reg_sad or NULL */
typedef uint32_t (*kvz_hip_optimized_sad_fn)(const uint8_t *pic, const uint8_t *ref, int32_t height, uint32_t stride1, uint32_t stride2);
kvz_hip_optimized_sad_fn kvz_hip_get_optimized_sad(int32_t width);

/* ---- 1. typedef-exact: strategies-dct.h:44 dct_func ---------------------------------------------------------------- */
void kvz_hip_fast_forward_dst_4x4(int8_t bitdepth, const int16_t *input, int16_t *output);  /* dct-generic.c:601-609 */
void kvz_hip_dct_4x4(int8_t bitdepth, const int16_t *input, int16_t *output);               /* :559-568 DCT_NXN_GENERIC */
void kvz_hip_dct_8x8(int8_t bitdepth, const int16_t *input, int16_t *output);
void kvz_hip_dct_16x16(int8_t bitdepth, const int16_t *input, int16_t *output);
void kvz_hip_dct_32x32(int8_t bitdepth, const int16_t *input, int16_t *output);
void kvz_hip_fast_inverse_dst_4x4(int8_t bitdepth, const int16_t *input, int16_t *output);  /* :611-619 */
void kvz_hip_idct_4x4(int8_t bitdepth, const int16_t *input, int16_t *output);              /* :570-579 IDCT_NXN_GENERIC */
void kvz_hip_idct_8x8(int8_t bitdepth, const int16_t *input, int16_t *output);
void kvz_hip_idct_16x16(int8_t bitdepth, const int16_t *input, int16_t *output);
void kvz_hip_idct_32x32(int8_t bitdepth, const int16_t *input, int16_t *output);

/* ---- 1. typedef-exact: strategies-intra.h:45-63 (int_fast8_t == signed char on x86-64 glibc) ----------------------- */
void kvz_hip_angular_pred(const int_fast8_t log2_width, const int_fast8_t intra_mode, const uint8_t *in_ref_above,
                          const uint8_t *in_ref_left, uint8_t *dst);                         /* intra-generic.c:49-155 */
void kvz_hip_intra_pred_planar(const int_fast8_t log2_width, const uint8_t *ref_top, const uint8_t *ref_left, uint8_t *dst);       /* :165-201 */
void kvz_hip_intra_pred_filtered_dc(const int_fast8_t log2_width, const uint8_t *ref_top, const uint8_t *ref_left, uint8_t *dst);  /* :210-241 */

/* ---- 1. typedef-exact: strategies-quant.h:60-62 -------------------------------------------------------------------- */
/* array_checksum_func (strategies-nal.h:54-58): typedef-exact; writes the sum big-endian into checksum_out[0..3] */
void     kvz_hip_array_md5(const uint8_t *data, const int height, const int width, const int stride, unsigned char checksum_out[16],
                           const uint8_t bitdepth);  /* strategies-nal.h:54-58 "array_md5" (--hash md5) */
void     kvz_hip_array_checksum(const uint8_t *data, const int height, const int width, const int stride, unsigned char checksum_out[16],
                                const uint8_t bitdepth);
/* kvz_rdoq (rdo.c:661-1000; not a strategy pointer in the reference: quant-generic.c:234-244 calls it directly when --rdoq is on) for intra
 * blocks, flat scaling lists, sign hiding off: ctx_states = uc_state of state->cabac's contexts in KVZ_HIP_CX_* order (kvz_hip_types.h), lambda =
 * state->lambda, type 0 luma / 2 chroma, tr_depth as passed to kvz_rdoq.  dest is read and written (positions above the last significant
 * coefficient are zeroed, a block without any keeps the rest of the caller's values, as the reference does).  The 4th argument is unused (the
 * library prices with kvz_entropy_bits, rdo.c:69-80); it keeps the signature of the test checkers.  _blocks: `count` blocks of one shape back to back. */
void     kvz_hip_rdoq(int qp, double lambda, const uint8_t *ctx_states, const float *unused, const int16_t *coef, int16_t *dest, int width, int type, int scan_mode, int tr_depth);
/* kvz_quantize_residual (quant-generic.c:198-292) WITH rdoq (:234-244) for an intra block -- flat lists, no sign hiding, no transform skip, not lossless: residual,
 * forward transform, kvz_rdoq on the caller's context states (as kvz_hip_rdoq), dequantisation, inverse transform and reconstruction in one device round trip.
 * Arguments as kvz_hip_quantize_residual plus state->lambda, the contexts and kvz_rdoq's tr_depth. */
int      kvz_hip_quantize_residual_rdoq(const kvz_hip_quant_params *p, double lambda, const uint8_t *ctx_states, int tr_depth, int width, int color, int scan_order,
                                        int in_stride, int out_stride, const uint8_t *ref_in, const uint8_t *pred_in, uint8_t *rec_out, int16_t *coeff_out, int early_skip);
void     kvz_hip_rdoq_blocks(int qp, double lambda, const uint8_t *ctx_states, const int16_t *coef, int16_t *dest, int width, int type, int scan_mode, int tr_depth, int count);
void     kvz_hip_plane_md5(const uint8_t *data, int height, int width, int stride, uint8_t *out16);       /* nal-generic.c:41-55, the 16 digest bytes */
/* strategies-encode.h:49-65 kvz_encode_coeff_nxn (strategies/generic/encode_coding_tree-generic.c:40-283), the part that does not touch the coder's state: the residual
 * syntax of one width x width block (type 0 luma / 2 chroma; scan_mode 0 diagonal, 1 horizontal, 2 vertical; sign hiding, transform skip and encryption off) as bin
 * records -- context-coded bin: context index (KVZ_HIP_CX_* of kvz_hip_types.h) | value << 8; bypass run: 1 << 30 | bins << 16 | value (16 bins at most per record,
 * most significant first); -- in coding order.  The caller drives its cabac_data_t with them (integration/kvazaar/strategies/hip/encode-hip.c).  Returns the number of
 * records of the block; at most `capacity` are written. */
int kvz_hip_coeff_nxn_bins(const int16_t *coeff, int width, int type, int scan_mode, uint32_t *records, int capacity);
uint32_t kvz_hip_plane_checksum(const uint8_t *data, int height, int width, int stride);            /* nal-generic.c:57-82, the 32-bit sum */
uint32_t kvz_hip_coeff_abs_sum(const int16_t *coeffs, size_t length);                        /* quant-generic.c:342-349 */
double   kvz_hip_fast_coeff_cost(const int16_t *coeff, int32_t width, uint64_t weights);     /* :359-375 */

/* ---- 2. flat: functions whose reference signature carries host structs ------------------------------------------- */
/* All-sizes forms used by the tests (n in {4,8,16,32,64}); the typedef-exact names above forward to these. */
unsigned kvz_hip_sad_nxn(int n, const uint8_t *b1, const uint8_t *b2);
unsigned kvz_hip_satd_nxn(int n, const uint8_t *b1, const uint8_t *b2);
void     kvz_hip_sad_nxn_dual(int n, const uint8_t *preds, const uint8_t *orig, unsigned num_modes, unsigned *costs_out);
void     kvz_hip_satd_nxn_dual(int n, const uint8_t *preds, const uint8_t *orig, unsigned num_modes, unsigned *costs_out);
void     kvz_hip_transform(int kind /* enum kvz_hip_transform_kind */, int8_t bitdepth, const int16_t *in, int16_t *out);
/* image.c:407 kvz_image_calc_sad (the frame-edge glue around reg_sad/ver_sad/hor_sad, image.c:279-397) as ONE call:
 * SAD of the block of `pic` at (pic_x,pic_y) against `ref` at (ref_x,ref_y) with edge replication. */
unsigned kvz_hip_image_calc_sad(const uint8_t *pic, int pic_stride, const uint8_t *ref, int ref_w, int ref_h, int ref_stride,
                                int pic_x, int pic_y, int ref_x, int ref_y, int block_width, int block_height);
/* inter_recon_bipred_func (strategies-picture.h:136-148) takes lcu_t / yuv_t / yuv_im_t: one plane per call, exactly one of
 * (px, im) non-NULL per list (picture-generic.c:553-668) */
void kvz_hip_bipred_average_plane(uint8_t *dst, unsigned dst_stride, const uint8_t *px_L0, const int16_t *im_L0,
                                  const uint8_t *px_L1, const int16_t *im_L1, unsigned pu_w, unsigned pu_h);
/* quant_func / dequant_func / quant_residual_func (strategies-quant.h:49-59): encoder_state_t -> kvz_hip_quant_params.
 * quantize_residual covers the rdoq-off path (quant-generic.c:198-292); with rdoq on the reference calls the host
 * function kvz_rdoq (rdo.c:661), which stays on the host. */
void kvz_hip_quant(const kvz_hip_quant_params *p, const int16_t *coef, int16_t *q_coef, int32_t width, int32_t height,
                   int8_t type, int8_t scan_idx, int8_t block_type);
void kvz_hip_dequant(const kvz_hip_quant_params *p, const int16_t *q_coef, int16_t *coef, int32_t width, int32_t height,
                     int8_t type, int8_t block_type);
int  kvz_hip_quantize_residual(const kvz_hip_quant_params *p, int width, int color, int scan_order, int use_trskip,
                               int in_stride, int out_stride, const uint8_t *ref_in, const uint8_t *pred_in,
                               uint8_t *rec_out, int16_t *coeff_out, int early_skip);
/* find_last_scanpos_func (:64-65): struct kvz_sh_rates_t* -> its sig_coeff_inc array */
void kvz_hip_find_last_scanpos(const int16_t *coef, int16_t *dest_coeff, int8_t type, int32_t q_bits, const int16_t *quant_coeff,
                               int32_t *sig_coeff_inc, uint32_t cg_size, uint16_t *ctx_set, const uint32_t *scan,
                               int32_t *cg_last_scanpos, int32_t *last_scanpos, uint32_t cg_num, int32_t *cg_scanpos,
                               int32_t width, int8_t scan_mode);
int32_t kvz_hip_get_scaled_qp(int8_t type, int8_t qp, int8_t qp_offset);                     /* transform.c:141-155 */
/* strategies-ipol.h:95-120: the `const encoder_control_t *encoder` first argument is unused by the reference at 8 bit
 * and dropped here. */
void kvz_hip_sample_quarterpel_luma(const uint8_t *src, int16_t src_stride, int width, int height, uint8_t *dst, int16_t dst_stride,
                                    int8_t hor_flag, int8_t ver_flag, const int16_t mv[2]);  /* ipol-generic.c:134-178 */
void kvz_hip_sample_quarterpel_luma_hi(const uint8_t *src, int16_t src_stride, int width, int height, int16_t *dst, int16_t dst_stride,
                                       int8_t hor_flag, int8_t ver_flag, const int16_t mv[2]);  /* :180-211 */
void kvz_hip_sample_octpel_chroma(const uint8_t *src, int16_t src_stride, int width, int height, uint8_t *dst, int16_t dst_stride,
                                  int8_t hor_flag, int8_t ver_flag, const int16_t mv[2]);    /* :681-725 */
void kvz_hip_sample_octpel_chroma_hi(const uint8_t *src, int16_t src_stride, int width, int height, int16_t *dst, int16_t dst_stride,
                                     int8_t hor_flag, int8_t ver_flag, const int16_t mv[2]); /* :727-758 */
/* ipol_blocks_func (strategies-ipol.h:64-66): filtered = kvz_pixel[4][64*64]; hor_intermediate = int16[5][KVZ_HIP_IPOL_IM_PLANE];
 * hor_first_cols = int16[5][KVZ_HIP_IPOL_COL_LEN], all flattened.  The intermediates the reference leaves behind for the next
 * call are written back too, but every call recomputes what it needs from `src` (it is a pure function of the window). */
#define KVZ_HIP_IPOL_IM_PLANE ((64 + 7 + 1) * 64 + 1)
#define KVZ_HIP_IPOL_COL_LEN (64 + 7 + 1)
void kvz_hip_filter_hpel_blocks_hor_ver_luma(const uint8_t *src, int16_t src_stride, int width, int height, uint8_t *filtered,
                                             int16_t *hor_intermediate, int8_t fme_level, int16_t *hor_first_cols,
                                             int8_t hpel_off_x, int8_t hpel_off_y);         /* ipol-generic.c:213-326 */
void kvz_hip_filter_hpel_blocks_diag_luma(const uint8_t *src, int16_t src_stride, int width, int height, uint8_t *filtered,
                                          int16_t *hor_intermediate, int8_t fme_level, int16_t *hor_first_cols,
                                          int8_t hpel_off_x, int8_t hpel_off_y);            /* :328-407 */
void kvz_hip_filter_qpel_blocks_hor_ver_luma(const uint8_t *src, int16_t src_stride, int width, int height, uint8_t *filtered,
                                             int16_t *hor_intermediate, int8_t fme_level, int16_t *hor_first_cols,
                                             int8_t hpel_off_x, int8_t hpel_off_y);         /* :409-567 */
void kvz_hip_filter_qpel_blocks_diag_luma(const uint8_t *src, int16_t src_stride, int width, int height, uint8_t *filtered,
                                          int16_t *hor_intermediate, int8_t fme_level, int16_t *hor_first_cols,
                                          int8_t hpel_off_x, int8_t hpel_off_y);            /* :569-679 */
/* epol_func (strategies-ipol.h:95): returns 1 and fills `buf` (stride pad_l+blk_w+pad_r) when the window leaves the frame,
 * 0 when the reference would return a pointer into the frame (ipol-generic.c:761-814) */
int kvz_hip_get_extended_block(const kvz_hip_epol_params *args, const uint8_t *src, uint8_t *buf);
/* strategies-sao.h:49-70: encoder_control_t* -> bitdepth, sao_info_t* -> kvz_hip_sao_params, encoder_state_t* -> bitdepth */
int  kvz_hip_sao_edge_ddistortion(int bitdepth, const uint8_t *orig_data, const uint8_t *rec_data, int block_width, int block_height,
                                  int eo_class, const int offsets[5]);                       /* sao_shared_generics.h:52-91 */
void kvz_hip_calc_sao_edge_dir(int bitdepth, const uint8_t *orig_data, const uint8_t *rec_data, int eo_class, int block_width,
                               int block_height, int cat_sum_cnt[10] /* [2][5], accumulated */);  /* sao-generic.c:50-81 */
void kvz_hip_sao_reconstruct_color(const kvz_hip_sao_params *sao, const uint8_t *rec_data, uint8_t *new_rec_data, int stride,
                                   int new_stride, int block_width, int block_height, int color_i);  /* sao-generic.c:84-124 */
int  kvz_hip_sao_band_ddistortion(int bitdepth, const uint8_t *orig_data, const uint8_t *rec_data, int block_width, int block_height,
                                  int band_pos, const int sao_bands[4]);                     /* sao_shared_generics.h:93-130 */

#ifdef __cplusplus
}
#endif
#endif
