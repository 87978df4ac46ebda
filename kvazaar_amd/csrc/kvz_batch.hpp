// kvz_batch.hpp -- host side of the batched CTU pass: device buffers of a frame batch, the launches of the CTU kernels (kvz_ctu_kernels.hpp: persistent launch
// with an in-order ticket list; one launch per anti-diagonal kept for A/B), the cost model.  Included by kvz_hip.hip only.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

#include <vector>

#include "../../include/kvz_hip_batch.h"
#include "kvz_ctu.hpp"
#include "kvz_runtime.hpp"

#include "kvz_ctu_kernels.hpp"

namespace kvz { struct DevBuf { void *p = nullptr; size_t bytes = 0; }; }  // a grow-only device buffer (kvz_dev.hpp EntropyScratch::need)

struct kvz_hip_batch {
  kvz::CtuFrames F;
  int n_frames;
  hipStream_t stream;
  hipEvent_t ev0, ev1;
  hipEvent_t ev_up = nullptr;  // kvz_hip_batch_upload_all_async: what the next pass waits for
  int up_pending = 0;
  uint8_t *d_src, *d_rec, *d_depth, *d_mode;
  uint8_t *d_part, *d_mode4;  // search_nxn: NxN flag per 8x8 CU, luma mode per 4x4 unit (allocated with the first model that has it set)
  int16_t *d_coeff, *d_scratch;
  double *d_cost;
  uint8_t *d_border;
  unsigned long long *d_prof;
  float *d_entropy;  // the model's entropy_fbits [128 floats] followed by its ctx_init [160 bytes] of the run in flight
  uint32_t *d_items, *d_items_raster;  // ticket order with WPP (anti-diagonals) / without (raster order per picture)
  unsigned *d_ticket, *d_done, *d_error;
  unsigned *h_error = nullptr;  // pinned: the error word as the last pass left it, copied behind every pass on the batch's stream (batch_check reads it without a copy of its own)
  float last_entropy[128 + 40];  // the price table and initial contexts d_entropy holds (a launch with the same model skips the copy)
  int entropy_valid = 0;
  int entropy_deferred = 0;     // kvz_hip_batch_entropy_defer_download: the coder's calls return with the slice data's download queued, not finished
  kvz::DevBuf entropy_out;      // ... and compact into this buffer of the batch's own instead of the device's shared one
  unsigned total_items, epoch;
  int sched_ticket, grid_ticket;
  int slots_per_cu, cus;  // what the persistent pass may occupy at most (occupancy x CU count); grid_ticket = its share of that (kvz_hip_batch_set_device_share)
  // SAO (kvz_hip_batch_loop_filters with sao != 0; allocated on first use): the picture after the vertical edges and after all edges,
  // statistics / context-free candidates / packed parameter records per (LCU, plane), merge choice per LCU
  uint8_t *d_ver, *d_dbk, *d_sao_merge;
  void *d_sao_stats, *d_sao_cand;
  unsigned long long *d_sao_recs;
  float *d_sao_fbits;
  int device;   // the batch's buffers and stream live here; every entry point binds the calling thread to it
  int failed;   // sticky: a CTU hand-off wait of some run timed out, the results of that run are invalid
  unsigned long long wait_ticks;
};

namespace kvz {
// kvazaar worker threads other than the creating one call into a batch (search_lcu_hip.c): bind them to the batch's device
inline void batch_enter(const kvz_hip_batch *b) { KVZ_HIP_CHECK(hipSetDevice(b->device)); thread_state().bound = true; }
// after the stream has drained: did a hand-off wait time out?  0 ok, -1 invalid results (reported, never fatal: the embedding
// encoder decides what to do)
inline int batch_check(kvz_hip_batch *b)
{
  if (b->sched_ticket && !b->failed) {
    // a synchronous hipMemcpy here waited for everything the copy engine held -- 84 ms of another batch's pictures on their way up in the double-buffered chain
    const unsigned err = *(volatile unsigned *)b->h_error;
    if (err) {
      b->failed = 1;
      fprintf(stderr, "kvz_hip: a CTU hand-off wait timed out (KVZ_HIP_WAIT_MS to raise the bound) -- the results of this batch are invalid [first to give up waited for CTU x %u y %u of picture %u (%s), pass %u of the batch; that CTU's flag read at the memory side then: %u; tickets drawn then: %u of %u]\n",
              err & 0xffu, (err >> 8) & 0xffu, (err >> 16) & 0x1fffu, (err & 0x20000000u) ? "its above-right neighbour" : ((err & 0x40000000u) ? "the end of the row above" : "its left neighbour"), b->epoch, ((volatile unsigned *)b->h_error)[1], ((volatile unsigned *)b->h_error)[2], b->total_items);
    }
  }
  return b->failed ? -1 : 0;
}
}  // namespace kvz

namespace kvz {


// kvazaar's default fast-coefficient-cost weights per QP (fast_coeff_cost.h:48-101: four doubles per QP for |level| = 0, 1, 2, >= 3),
// packed as kvz_fast_coeff_use_default_table does (fast_coeff_cost.c:39-52, 76-82: Q8.8 each, |level| = 0 in the low word) --
// what kvz_fast_coeff_get_weights(state) returns unless --fastrd-learning-outdir / a custom table is used.  QP < 50
// (MAX_FAST_COEFF_COST_QP).  Constant data; tests/test_ctu_pipeline.py pins it against the reference build's values.
static const uint64_t kDefaultCoeffWeights[50] = {
  0x06f8038004200029ull, 0x06f8038004200029ull, 0x06f8038004200029ull, 0x06f8038004200029ull, 0x06f8038004200029ull,
  0x06f8038004200029ull, 0x06f8038004200029ull, 0x06f8038004200029ull, 0x06f8038004200029ull, 0x06f8038004200029ull,
  0x06f8038004200029ull, 0x06e5038f040a0028ull, 0x06f703eb044f0021ull, 0x06e603f2046c001cull, 0x06ce0429047b0018ull,
  0x06b9040b04a10013ull, 0x068f040004f6000dull, 0x067903f40522000aull, 0x066b03ce052f0009ull, 0x065a03c905340007ull,
  0x065903cc05510006ull, 0x065303d105390005ull, 0x065403f0052c0004ull, 0x064b042505170003ull, 0x0644043405040002ull,
  0x0635044e04f50002ull, 0x062d046704ea0001ull, 0x0622046304e80001ull, 0x0627048604df0001ull, 0x0627049704dd0001ull,
  0x0624049b04dd0001ull, 0x063b04cb04c60001ull, 0x064604ca04cb0000ull, 0x063c04ca04d80000ull, 0x064604ce04d90000ull,
  0x065904de04d70000ull, 0x067304f804c60000ull, 0x06aa050d04be0000ull, 0x069a050904cd0000ull, 0x06be051404cc0000ull,
  0x06df052504c80000ull, 0x073a053804ea0000ull, 0x0748052704f90000ull, 0x0696048305510000ull, 0x06c6048e054e0000ull,
  0x06f604b105430000ull, 0x071a04bc05310000ull, 0x075704d5052e0000ull, 0x075704d4052f0000ull, 0x0744048505640000ull,
};

// context.c:202-213 kvz_ctx_init
inline int ctx_state(int qp, int init_value)
{
  const int slope = (init_value >> 4) * 5 - 45, offset = ((init_value & 15) << 3) - 16;
  int st = ((slope * qp) >> 4) + offset;
  st = st < 1 ? 1 : (st > 126 ? 126 : st);
  return st >= 64 ? ((st - 64) << 1) + 1 : (63 - st) << 1;
}

// the struct versions this library knows (include/kvz_hip_types.h struct_size): the current one
inline bool cost_model_known(const kvz_hip_intra_cost_model *m, const char *who)
{
  if (m && m->struct_size == sizeof(kvz_hip_intra_cost_model)) return true;
  fprintf(stderr, "%s: kvz_hip_intra_cost_model.struct_size %u is not this library's %zu (caller built against other headers, or the struct was not set up by kvz_hip_intra_cost_model_init)\n",
          who, m ? m->struct_size : 0u, sizeof(kvz_hip_intra_cost_model));
  return false;
}

inline void cost_model_init(int qp, uint64_t coeff_weights, kvz_hip_intra_cost_model *m)
{
  // I-slice rows of context.c:96-193 (HEVC spec tables 9-5 ff.); 154 = CNU, never coded
  static const uint8_t init_split[3] = { 139, 141, 157 }, init_cbf_luma[2] = { 111, 141 }, init_cbf_chroma[2] = { 94, 138 };
  static const uint8_t init_sig_cg[4] = { 91, 171, 134, 141 };
  static const uint8_t init_sig[42] = { 111, 111, 125, 110, 110, 94, 124, 108, 124, 107, 125, 141, 179, 153, 125, 107, 125, 141, 179, 153, 125, 107, 125, 141, 179, 153, 125,
                                        140, 139, 182, 182, 152, 136, 152, 136, 153, 136, 139, 111, 136, 139, 111 };
  static const uint8_t init_last[30] = { 110, 110, 124, 125, 140, 153, 125, 127, 140, 109, 111, 143, 127, 111, 79, 108, 123, 63,
                                         154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154, 154 };
  static const uint8_t init_one[24] = { 140, 92, 137, 138, 140, 152, 138, 139, 153, 74, 149, 92, 139, 107, 122, 152, 140, 179, 166, 182, 140, 227, 122, 197 };
  static const uint8_t init_abs[6] = { 138, 153, 136, 167, 152, 152 };
  memset(m, 0, sizeof *m);
  m->struct_size = (uint32_t)sizeof *m;
  m->qp = qp;
  m->lambda = 0.57 * pow(2.0, (qp - 12) / 3.0);
  m->lambda_sqrt = sqrt(m->lambda);
  m->coeff_weights = coeff_weights;
  for (int i = 0; i < 128; i++) m->entropy_fbits[i] = (float)kEntropyBits[i] / 32768.0f;
  int n = 0;
  auto fill = [&](float dst[2], int init) {
    const int st = ctx_state(qp, init);
    m->ctx_init[n++] = (uint8_t)st;  // same order as the KVZ_HIP_CX_* indices
    dst[0] = m->entropy_fbits[st ^ 0];
    dst[1] = m->entropy_fbits[st ^ 1];
  };
  for (int i = 0; i < 3; i++) fill(m->split_flag[i], init_split[i]);
  fill(m->part_size, 184);
  fill(m->intra_mode, 184);
  fill(m->chroma_mode, 63);
  for (int i = 0; i < 2; i++) fill(m->cbf_luma[i], init_cbf_luma[i]);
  for (int i = 0; i < 2; i++) fill(m->cbf_chroma[i], init_cbf_chroma[i]);
  auto put = [&](int at, const uint8_t *init, int count) { for (int i = 0; i < count; i++) m->ctx_init[at + i] = (uint8_t)ctx_state(qp, init[i]); };
  put(KVZ_HIP_CX_SIG_CG, init_sig_cg, 4);
  put(KVZ_HIP_CX_SIG_LUMA, init_sig, 27);
  put(KVZ_HIP_CX_SIG_CHROMA, init_sig + 27, 15);
  put(KVZ_HIP_CX_LAST_Y_LUMA, init_last, 15);
  put(KVZ_HIP_CX_LAST_Y_CHROMA, init_last + 15, 15);
  put(KVZ_HIP_CX_LAST_X_LUMA, init_last, 15);
  put(KVZ_HIP_CX_LAST_X_CHROMA, init_last + 15, 15);
  put(KVZ_HIP_CX_ONE_LUMA, init_one, 16);
  put(KVZ_HIP_CX_ONE_CHROMA, init_one + 16, 8);
  put(KVZ_HIP_CX_ABS_LUMA, init_abs, 4);
  put(KVZ_HIP_CX_ABS_CHROMA, init_abs + 4, 2);
  {
    static const uint8_t init_cbf_chroma_deep[2] = { 182, 154 };  // INIT_QT_CBF[2][6..7] (context.c:130-134)
    put(KVZ_HIP_CX_CBF_CHROMA_DEEP, init_cbf_chroma_deep, 2);
  }
  m->ctx_init[KVZ_HIP_CX_SAO_MERGE] = (uint8_t)ctx_state(qp, 153);  // context.c:38-39 INIT_SAO_MERGE_FLAG / INIT_SAO_TYPE_IDX, I slice
  m->ctx_init[KVZ_HIP_CX_SAO_TYPE] = (uint8_t)ctx_state(qp, 200);
  m->adaptive = 1;
  m->coeff_cabac = qp >= 28;  // `ultrafast`: fast-residual-cost 28 (cfg.c:485-512), rdo.c:311-340
}

}  // namespace kvz

extern "C" {

void kvz_hip_intra_cost_model_init(int qp, uint64_t coeff_weights, kvz_hip_intra_cost_model *model) { kvz::cost_model_init(qp, coeff_weights, model); }
uint64_t kvz_hip_default_coeff_weights(int qp) { return qp >= 0 && qp < 50 ? kvz::kDefaultCoeffWeights[qp] : 0; }

kvz_hip_batch *kvz_hip_batch_create(int width, int height, int n_frames) { return kvz_hip_batch_create_on(-1, width, height, n_frames); }

kvz_hip_batch *kvz_hip_batch_create_on(int device, int width, int height, int n_frames)
{
  if (width <= 0 || height <= 0 || n_frames <= 0 || (width & 7) || (height & 7)) {
    fprintf(stderr, "kvz_hip_batch_create: width and height must be positive multiples of 8\n");
    return nullptr;
  }
  kvz::runtime_init(-1);
  if (device >= kvz::runtime().n_devices) {
    fprintf(stderr, "kvz_hip_batch_create_on: device %d of %d\n", device, kvz::runtime().n_devices);
    return nullptr;
  }
  if (device >= 0) KVZ_HIP_CHECK(hipSetDevice(device));  // (the calling thread stays on it, as after any call on the batch)
  kvz_hip_batch *b = new kvz_hip_batch();
  b->device = kvz::current_device();
  b->failed = 0;
  {
    const char *e = getenv("KVZ_HIP_WAIT_MS");  // bound of one hand-off wait (wall clock); a whole 4K picture without WPP is a 2.5 s chain
    const double ms = e && atof(e) > 0 ? atof(e) : 30000.0;
    b->wait_ticks = (unsigned long long)(ms * 1e5);
  }
  kvz::CtuFrames &F = b->F;
  F.W = width; F.H = height; F.wc = (width + 63) / 64; F.hc = (height + 63) / 64;
  F.frame_px = (long)width * height * 3 / 2;
  b->n_frames = n_frames;
  const size_t nctu = (size_t)F.wc * F.hc * n_frames, ncu = (size_t)(width / 8) * (height / 8) * n_frames;
  KVZ_HIP_CHECK(hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking));
  KVZ_HIP_CHECK(hipEventCreate(&b->ev0));
  KVZ_HIP_CHECK(hipEventCreate(&b->ev1));
  KVZ_HIP_CHECK(hipMalloc((void **)&b->d_src, F.frame_px * n_frames));
  KVZ_HIP_CHECK(hipMalloc((void **)&b->d_rec, F.frame_px * n_frames));
  KVZ_HIP_CHECK(hipMalloc((void **)&b->d_coeff, nctu * 6144 * sizeof(int16_t)));
  KVZ_HIP_CHECK(hipMalloc((void **)&b->d_scratch, nctu * 6144 * sizeof(int16_t)));
  KVZ_HIP_CHECK(hipMalloc((void **)&b->d_depth, ncu));
  KVZ_HIP_CHECK(hipMalloc((void **)&b->d_mode, ncu));
  KVZ_HIP_CHECK(hipMalloc((void **)&b->d_cost, nctu * sizeof(double)));
  KVZ_HIP_CHECK(hipMemsetAsync(b->d_rec, 0, F.frame_px * n_frames, b->stream));  // on the batch's stream: a non-blocking stream does not order against the null stream
#ifdef KVZ_CTU_TRACE
  KVZ_HIP_CHECK(hipMalloc((void **)&b->d_prof, (size_t)nctu * sizeof(unsigned long long)));  // a word per CTU: where it is (kvz_ctu_kernels.hpp KVZ_TRACE)
  KVZ_HIP_CHECK(hipMemsetAsync(b->d_prof, 0, (size_t)nctu * sizeof(unsigned long long), b->stream));
#else
  KVZ_HIP_CHECK(hipMalloc((void **)&b->d_prof, KVZ_PROF_WORDS * sizeof(unsigned long long)));  // + 8: the sections of rdoq_block_wave
  KVZ_HIP_CHECK(hipMemsetAsync(b->d_prof, 0, KVZ_PROF_WORDS * sizeof(unsigned long long), b->stream));
#endif
  F.prof = b->d_prof;
  KVZ_HIP_CHECK(hipMalloc((void **)&b->d_entropy, 128 * sizeof(float) + sizeof(((kvz_hip_intra_cost_model *)0)->ctx_init)));
  KVZ_HIP_CHECK(hipMalloc((void **)&b->d_border, nctu * KVZ_BORDER_BYTES));
  KVZ_HIP_CHECK(hipMemsetAsync(b->d_border, 0, nctu * KVZ_BORDER_BYTES, b->stream));
  F.border = b->d_border;
  {  // ticket schedule: items in dependency order (anti-diagonal, frame, row)
    std::vector<uint32_t> items;
    items.reserve(nctu);
    for (int wave = 0; wave <= (F.wc - 1) + 2 * (F.hc - 1); wave++)
      for (int f = 0; f < n_frames; f++)
        for (int y = 0; y < F.hc; y++) {
          const int x = wave - 2 * y;
          if (x >= 0 && x < F.wc) items.push_back((uint32_t)f << 16 | (uint32_t)y << 8 | (uint32_t)x);
        }
    b->total_items = (unsigned)items.size();
    b->epoch = 0;
    KVZ_HIP_CHECK(hipMalloc((void **)&b->d_items, items.size() * sizeof(uint32_t)));
    KVZ_HIP_CHECK(hipMemcpy(b->d_items, items.data(), items.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    items.clear();  // without WPP every CTU depends on its raster predecessor: CTU i of every picture, then CTU i + 1 of every picture
    for (int y = 0; y < F.hc; y++)
      for (int x = 0; x < F.wc; x++)
        for (int f = 0; f < n_frames; f++) items.push_back((uint32_t)f << 16 | (uint32_t)y << 8 | (uint32_t)x);
    KVZ_HIP_CHECK(hipMalloc((void **)&b->d_items_raster, items.size() * sizeof(uint32_t)));
    KVZ_HIP_CHECK(hipMemcpy(b->d_items_raster, items.data(), items.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    KVZ_HIP_CHECK(hipMalloc((void **)&b->d_done, nctu * sizeof(unsigned)));
    KVZ_HIP_CHECK(hipMemsetAsync(b->d_done, 0, nctu * sizeof(unsigned), b->stream));
    KVZ_HIP_CHECK(hipMalloc((void **)&b->d_ticket, 4 * sizeof(unsigned)));  // the ticket, the error word, two words of what the first CTU to give up saw
    KVZ_HIP_CHECK(hipMemsetAsync(b->d_ticket, 0, 4 * sizeof(unsigned), b->stream));
    b->d_error = b->d_ticket + 1;
    KVZ_HIP_CHECK(hipHostMalloc((void **)&b->h_error, 64, hipHostMallocDefault));
    *b->h_error = 0;
    const char *e = getenv("KVZ_HIP_SCHED");  // "wave": one launch per anti-diagonal (the simpler schedule, kept for A/B)
    b->sched_ticket = !(e && e[0] == 'w') && n_frames < 65536 && F.wc < 256 && F.hc < 256;
    int per_cu = 0, dev = 0, cus = 0;
    KVZ_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kvz::intra_ctu_ticket_kernel<true>, KVZ_CTU_THREADS, 0));
    KVZ_HIP_CHECK(hipGetDevice(&dev));
    KVZ_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    if (per_cu < 1) per_cu = 1;
    if (const char *lim = getenv("KVZ_HIP_WG_PER_CU")) { const int v = atoi(lim); if (v >= 1 && v < per_cu) per_cu = v; }  // occupancy experiments
    b->slots_per_cu = per_cu; b->cus = cus;
    b->grid_ticket = per_cu * cus;
    if ((unsigned)b->grid_ticket > b->total_items) b->grid_ticket = (int)b->total_items;
  }
  KVZ_HIP_CHECK(hipStreamSynchronize(b->stream));
  F.src = b->d_src; F.rec = b->d_rec; F.coeff = b->d_coeff; F.coeff_scratch = b->d_scratch;
  F.cu_depth = b->d_depth; F.cu_mode = b->d_mode; F.ctu_cost = b->d_cost;
  return b;
}

void kvz_hip_batch_destroy(kvz_hip_batch *b)
{
  if (!b) return;
  kvz::batch_enter(b);
  (void)hipStreamSynchronize(b->stream);
  (void)hipFree(b->d_ver); (void)hipFree(b->d_dbk); (void)hipFree(b->d_sao_merge); (void)hipFree(b->d_sao_stats); (void)hipFree(b->d_sao_cand); (void)hipFree(b->d_sao_recs); (void)hipFree(b->d_sao_fbits);
  (void)hipFree(b->d_border); (void)hipFree(b->d_items); (void)hipFree(b->d_items_raster); (void)hipFree(b->d_done); (void)hipFree(b->d_ticket); (void)hipFree(b->d_prof); (void)hipFree(b->d_entropy);
  (void)hipFree(b->d_src); (void)hipFree(b->d_rec); (void)hipFree(b->d_coeff); (void)hipFree(b->d_scratch); (void)hipFree(b->d_depth); (void)hipFree(b->d_mode); (void)hipFree(b->d_cost); (void)hipFree(b->d_part); (void)hipFree(b->d_mode4);
  (void)hipEventDestroy(b->ev0); (void)hipEventDestroy(b->ev1);
  if (b->h_error) (void)hipHostFree(b->h_error);
  if (b->entropy_out.p) (void)hipFree(b->entropy_out.p);
  if (b->ev_up) { (void)hipEventSynchronize(b->ev_up); (void)hipEventDestroy(b->ev_up); }
  (void)hipStreamDestroy(b->stream);
  delete b;
}

int kvz_hip_batch_ctus_per_frame(const kvz_hip_batch *b) { return b->F.wc * b->F.hc; }

void kvz_hip_batch_upload(kvz_hip_batch *b, int frame, const uint8_t *y, const uint8_t *u, const uint8_t *v)
{
  kvz::batch_enter(b);
  const long ys = (long)b->F.W * b->F.H, cs = ys / 4;
  uint8_t *dst = b->d_src + (long)frame * b->F.frame_px;
  KVZ_HIP_CHECK(hipMemcpyAsync(dst, y, ys, hipMemcpyHostToDevice, b->stream));
  KVZ_HIP_CHECK(hipMemcpyAsync(dst + ys, u, cs, hipMemcpyHostToDevice, b->stream));
  KVZ_HIP_CHECK(hipMemcpyAsync(dst + ys + cs, v, cs, hipMemcpyHostToDevice, b->stream));
  KVZ_HIP_CHECK(hipStreamSynchronize(b->stream));
}

namespace kvz {
// ONE upload queue per device, shared by its batches: the runtime multiplexes streams onto four hardware queues, and a stream that lands on the queue of an upload
// stream waits for the copy in front of it (a fifth stream -- two batches, the coder's side stream, an upload stream per batch -- made the coder's kernels wait 84 ms
// for another batch's pictures: tools/chain_h2d_trace.sh)
inline hipStream_t upload_stream(int device)
{
  static std::mutex lock;
  static hipStream_t streams[64] = {};
  std::lock_guard<std::mutex> guard(lock);
  hipStream_t &st = streams[device & 63];
  if (!st) KVZ_HIP_CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  return st;
}
}  // namespace kvz

void kvz_hip_batch_upload_all_async(kvz_hip_batch *b, const uint8_t *src)
{
  kvz::batch_enter(b);
  const hipStream_t up = kvz::upload_stream(b->device);
  if (!b->ev_up) KVZ_HIP_CHECK(hipEventCreateWithFlags(&b->ev_up, hipEventDisableTiming));
  // the source pictures are only read by the CTU pass (and by SAO's statistics): the copy may start as soon as the batch's last pass has ended (ev1; a no-op
  // before the first pass), whatever its stream still holds behind it -- deblocking, the entropy coder, downloads
  KVZ_HIP_CHECK(hipStreamWaitEvent(up, b->ev1, 0));
  KVZ_HIP_CHECK(hipMemcpyAsync(b->d_src, src, (size_t)b->n_frames * b->F.frame_px, hipMemcpyHostToDevice, up));
  KVZ_HIP_CHECK(hipEventRecord(b->ev_up, up));
  b->up_pending = 1;
}

int kvz_hip_batch_download(kvz_hip_batch *b, int frame, uint8_t *rec_y, uint8_t *rec_u, uint8_t *rec_v, int16_t *coeff, uint8_t *cu_depth,
                           uint8_t *cu_mode, double *ctu_cost)
{
  kvz::batch_enter(b);
  const kvz::CtuFrames &F = b->F;
  const long ys = (long)F.W * F.H, cs = ys / 4, nctu = (long)F.wc * F.hc, ncu = (long)(F.W / 8) * (F.H / 8);
  const uint8_t *src = b->d_rec + (long)frame * F.frame_px;
  KVZ_HIP_CHECK(hipStreamSynchronize(b->stream));
  if (rec_y) KVZ_HIP_CHECK(hipMemcpy(rec_y, src, ys, hipMemcpyDeviceToHost));
  if (rec_u) KVZ_HIP_CHECK(hipMemcpy(rec_u, src + ys, cs, hipMemcpyDeviceToHost));
  if (rec_v) KVZ_HIP_CHECK(hipMemcpy(rec_v, src + ys + cs, cs, hipMemcpyDeviceToHost));
  if (coeff) KVZ_HIP_CHECK(hipMemcpy(coeff, b->d_coeff + frame * nctu * 6144, nctu * 6144 * sizeof(int16_t), hipMemcpyDeviceToHost));
  if (cu_depth) KVZ_HIP_CHECK(hipMemcpy(cu_depth, b->d_depth + frame * ncu, ncu, hipMemcpyDeviceToHost));
  if (cu_mode) KVZ_HIP_CHECK(hipMemcpy(cu_mode, b->d_mode + frame * ncu, ncu, hipMemcpyDeviceToHost));
  if (ctu_cost) KVZ_HIP_CHECK(hipMemcpy(ctu_cost, b->d_cost + frame * nctu, nctu * sizeof(double), hipMemcpyDeviceToHost));
  return kvz::batch_check(b);
}

int kvz_hip_batch_download_partitions(kvz_hip_batch *b, int frame, uint8_t *cu_part, uint8_t *cu_mode4)
{
  kvz::batch_enter(b);
  const kvz::CtuFrames &F = b->F;
  const long ncu = (long)(F.W / 8) * (F.H / 8), n4 = (long)(F.W / 4) * (F.H / 4);
  KVZ_HIP_CHECK(hipStreamSynchronize(b->stream));
  if (!b->d_part) { fprintf(stderr, "kvz_hip_batch_download_partitions: no pass with model.search_nxn has run on this batch\n"); return -1; }
  if (cu_part) KVZ_HIP_CHECK(hipMemcpy(cu_part, b->d_part + frame * ncu, ncu, hipMemcpyDeviceToHost));
  if (cu_mode4) KVZ_HIP_CHECK(hipMemcpy(cu_mode4, b->d_mode4 + frame * n4, n4, hipMemcpyDeviceToHost));
  return kvz::batch_check(b);
}

void *kvz_hip_host_alloc(size_t bytes)
{
  kvz::runtime_init(-1);
  void *p = nullptr;
  KVZ_HIP_CHECK(hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault));
  return p;
}
void kvz_hip_host_free(void *p) { if (p) KVZ_HIP_CHECK(hipHostFree(p)); }

void kvz_hip_batch_download_all_async(kvz_hip_batch *b, uint8_t *rec, int16_t *coeff, uint8_t *cu_depth, uint8_t *cu_mode)
{
  kvz::batch_enter(b);
  const kvz::CtuFrames &F = b->F;
  const size_t n = (size_t)b->n_frames, nctu = (size_t)F.wc * F.hc * n, ncu = (size_t)(F.W / 8) * (F.H / 8) * n;
  if (rec) KVZ_HIP_CHECK(hipMemcpyAsync(rec, b->d_rec, (size_t)F.frame_px * n, hipMemcpyDeviceToHost, b->stream));
  if (coeff) KVZ_HIP_CHECK(hipMemcpyAsync(coeff, b->d_coeff, nctu * 6144 * sizeof(int16_t), hipMemcpyDeviceToHost, b->stream));
  if (cu_depth) KVZ_HIP_CHECK(hipMemcpyAsync(cu_depth, b->d_depth, ncu, hipMemcpyDeviceToHost, b->stream));
  if (cu_mode) KVZ_HIP_CHECK(hipMemcpyAsync(cu_mode, b->d_mode, ncu, hipMemcpyDeviceToHost, b->stream));
}

void kvz_hip_batch_set_device_share(kvz_hip_batch *b, int num, int den)
{
  if (!b || !b->sched_ticket || num < 1 || den < num) return;
  int per_cu = b->slots_per_cu * num / den;
  if (per_cu < 1) per_cu = 1;
  b->grid_ticket = per_cu * b->cus;
  if ((unsigned)b->grid_ticket > b->total_items) b->grid_ticket = (int)b->total_items;
}

void kvz_hip_batch_order_after(kvz_hip_batch *b, kvz_hip_batch *other)
{
  kvz::batch_enter(b);
  hipEvent_t ev;
  KVZ_HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  KVZ_HIP_CHECK(hipEventRecord(ev, other->stream));
  KVZ_HIP_CHECK(hipStreamWaitEvent(b->stream, ev, 0));
  KVZ_HIP_CHECK(hipEventDestroy(ev));  // released once the wait has been satisfied
}

int kvz_hip_intra_frames(kvz_hip_batch *b, const kvz_hip_intra_cost_model *model)
{
  if (!b || !kvz::cost_model_known(model, "kvz_hip_intra_frames")) return -1;
  kvz::batch_enter(b);
  const kvz::CtuFrames &F = b->F;
  int launches = 0;
  kvz::CtuModel cm;
  kvz::ctu_model_from(model, &cm);
  cm.entropy_fbits = b->d_entropy;
  cm.ctx_init = (const uint8_t *)(b->d_entropy + 128);
  if (b->up_pending) { KVZ_HIP_CHECK(hipStreamWaitEvent(b->stream, b->ev_up, 0)); b->up_pending = 0; }  // pictures on their way (kvz_hip_batch_upload_all_async)
  {  // the model's tables: copied when they differ from what the device holds (a small host-to-device copy waits behind whatever the copy engine is busy with)
    static_assert(sizeof model->ctx_init <= 40 * sizeof(float), "last_entropy");
    float now[128 + 40] = { 0 };
    memcpy(now, model->entropy_fbits, 128 * sizeof(float));
    memcpy(now + 128, model->ctx_init, sizeof model->ctx_init);
    if (!b->entropy_valid || memcmp(now, b->last_entropy, sizeof now) != 0) {
      KVZ_HIP_CHECK(hipMemcpyAsync(b->d_entropy, model->entropy_fbits, 128 * sizeof(float), hipMemcpyHostToDevice, b->stream));
      KVZ_HIP_CHECK(hipMemcpyAsync(b->d_entropy + 128, model->ctx_init, sizeof model->ctx_init, hipMemcpyHostToDevice, b->stream));
      KVZ_HIP_CHECK(hipStreamSynchronize(b->stream));  // the sources are the caller's: staged before this call returns
      memcpy(b->last_entropy, now, sizeof now);
      b->entropy_valid = 1;
    }
  }
  // argument errors are reported, not fatal (a HIP failure still aborts: there is no error channel for it and no CPU path to fall back to)
  if (!b->sched_ticket && (cm.search_32x32 || cm.rdoq || cm.search_nxn)) { fprintf(stderr, "kvz_hip_intra_frames: search_32x32 / rdoq / search_nxn need the ticket schedule\n"); return -1; }
  if (!b->sched_ticket && cm.no_wpp) { fprintf(stderr, "kvz_hip_intra_frames: the one-launch-per-diagonal schedule (KVZ_HIP_SCHED=wave) needs WPP\n"); return -1; }
  if (cm.rdoq && !cm.coeff_cabac) { fprintf(stderr, "kvz_hip_intra_frames: rdoq needs coeff_cabac (kvazaar's presets with --rdoq have --fast-residual-cost 0)\n"); return -1; }
  if (b->sched_ticket) {
    b->epoch++;
    KVZ_HIP_CHECK(hipMemsetAsync(b->d_ticket, 0, sizeof(unsigned), b->stream));  // the error word behind it stays: sticky across runs
    KVZ_HIP_CHECK(hipEventRecord(b->ev0, b->stream));
    kvz::CtuSched sc{ cm.no_wpp ? b->d_items_raster : b->d_items, b->d_ticket, b->d_done, b->d_error, b->total_items, b->epoch, cm.no_wpp, b->wait_ticks };
    // two instantiations: the one without the CABAC coefficient model carries none of its code, registers or context storage
    // (the instantiations that search 32x32 CUs, --pu-depth-intra 1-3, are separate ones too: the others stay as they were)
    if (cm.rdoq || cm.search_nxn) {  // --rdoq and / or NxN partitions (preset `medium`): their own instantiation (32x32 search and the coefficient cost model switched by the model)
      if (cm.search_nxn && !b->d_part) {
        KVZ_HIP_CHECK(hipMalloc((void **)&b->d_part, (size_t)(F.W / 8) * (F.H / 8) * b->n_frames));
        KVZ_HIP_CHECK(hipMalloc((void **)&b->d_mode4, (size_t)(F.W / 4) * (F.H / 4) * b->n_frames));
      }
      kvz::CtuFrames Fr = F;
      Fr.cu_part = b->d_part; Fr.cu_mode4 = b->d_mode4;
      hipLaunchKernelGGL(kvz::intra_ctu_ticket_kernel_rdoq, dim3(b->grid_ticket), dim3(KVZ_CTU_THREADS), 0, b->stream, Fr, cm, kvz::device_tables(), sc);
    } else if (cm.search_32x32) {
      if (cm.coeff_cabac) hipLaunchKernelGGL((kvz::intra_ctu_ticket_kernel<true, true>), dim3(b->grid_ticket), dim3(KVZ_CTU_THREADS), 0, b->stream, F, cm, kvz::device_tables(), sc);
      else hipLaunchKernelGGL((kvz::intra_ctu_ticket_kernel<false, true>), dim3(b->grid_ticket), dim3(KVZ_CTU_THREADS), 0, b->stream, F, cm, kvz::device_tables(), sc);
    } else if (cm.coeff_cabac) hipLaunchKernelGGL(kvz::intra_ctu_ticket_kernel<true>, dim3(b->grid_ticket), dim3(KVZ_CTU_THREADS), 0, b->stream, F, cm, kvz::device_tables(), sc);
    else hipLaunchKernelGGL(kvz::intra_ctu_ticket_kernel<false>, dim3(b->grid_ticket), dim3(KVZ_CTU_THREADS), 0, b->stream, F, cm, kvz::device_tables(), sc);
    KVZ_HIP_CHECK(hipGetLastError());
    KVZ_HIP_CHECK(hipEventRecord(b->ev1, b->stream));
    KVZ_HIP_CHECK(hipMemcpyAsync(b->h_error, b->d_error, 3 * sizeof(unsigned), hipMemcpyDeviceToHost, b->stream));  // what batch_check reads once the stream has drained
    return 1;
  }
  KVZ_HIP_CHECK(hipEventRecord(b->ev0, b->stream));
  // CTU (x, y) needs (x-1, y), (x, y-1), (x+1, y-1): all of them lie on earlier anti-diagonals x + 2y (the WPP order of
  // encoderstate.c:793-903), so one launch per diagonal needs no synchronisation inside the launch.
  for (int wave = 0; wave <= (F.wc - 1) + 2 * (F.hc - 1); wave++) {
    int y_min = (wave - (F.wc - 1) + 1) / 2;
    if (y_min < 0) y_min = 0;
    int y_max = wave / 2;
    if (y_max > F.hc - 1) y_max = F.hc - 1;
    const int n_diag = y_max - y_min + 1;
    if (n_diag <= 0) continue;
    if (cm.coeff_cabac) hipLaunchKernelGGL(kvz::intra_ctu_wave_kernel<true>, dim3(n_diag * b->n_frames), dim3(KVZ_CTU_THREADS), 0, b->stream, F, cm, kvz::device_tables(), wave,
                                           y_min, n_diag);
    else hipLaunchKernelGGL(kvz::intra_ctu_wave_kernel<false>, dim3(n_diag * b->n_frames), dim3(KVZ_CTU_THREADS), 0, b->stream, F, cm, kvz::device_tables(), wave,
                            y_min, n_diag);
    launches++;
  }
  KVZ_HIP_CHECK(hipGetLastError());
  KVZ_HIP_CHECK(hipEventRecord(b->ev1, b->stream));
  return launches;
}

int kvz_hip_batch_sync(kvz_hip_batch *b)
{
  kvz::batch_enter(b);
  KVZ_HIP_CHECK(hipStreamSynchronize(b->stream));
  return kvz::batch_check(b);
}

/* developer diagnostics (tools/chain_stress.py): the hand-off flags of every CTU as they stand in memory -- a CTU's flag holds the number of the last pass that completed
 * it -- into out [n_frames x ctus_per_frame]; returns the number of the batch's last pass */
#ifdef KVZ_CTU_TRACE
extern "C" void kvz_hip_batch_debug_trace(kvz_hip_batch *b, unsigned long long *out)
{
  kvz::batch_enter(b);
  KVZ_HIP_CHECK(hipStreamSynchronize(b->stream));
  KVZ_HIP_CHECK(hipMemcpy(out, b->d_prof, (size_t)b->n_frames * b->F.wc * b->F.hc * sizeof(unsigned long long), hipMemcpyDeviceToHost));
}
#endif
unsigned kvz_hip_batch_debug_flags(kvz_hip_batch *b, unsigned *out)
{
  kvz::batch_enter(b);
  KVZ_HIP_CHECK(hipStreamSynchronize(b->stream));
  KVZ_HIP_CHECK(hipMemcpy(out, b->d_done, (size_t)b->n_frames * b->F.wc * b->F.hc * sizeof(unsigned), hipMemcpyDeviceToHost));
  return b->epoch;
}

int kvz_hip_batch_reset(kvz_hip_batch *b)
{
  kvz::batch_enter(b);
  KVZ_HIP_CHECK(hipStreamSynchronize(b->stream));
  KVZ_HIP_CHECK(hipMemset(b->d_error, 0, 3 * sizeof(unsigned)));
  *b->h_error = 0;
  b->failed = 0;
  return 0;
}

/* cycle counters of a -DKVZ_CTU_PROFILE build (all zero otherwise); reading resets them */
int kvz_hip_batch_profile(kvz_hip_batch *b, unsigned long long *out, int n)
{
  kvz::batch_enter(b);
  KVZ_HIP_CHECK(hipStreamSynchronize(b->stream));
  if (n > KVZ_PROF_WORDS) n = KVZ_PROF_WORDS;  // cycles per category, marks per category, rdoq_block_wave's sections and sizes, the same two tables inside eval_pu (kvz_ctu.hpp KVZ_PROF_PU_AT)
  KVZ_HIP_CHECK(hipMemcpy(out, b->d_prof, n * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  KVZ_HIP_CHECK(hipMemsetAsync(b->d_prof, 0, KVZ_PROF_WORDS * sizeof(unsigned long long), b->stream));
  KVZ_HIP_CHECK(hipStreamSynchronize(b->stream));
  return kvz::KVZ_P_COUNT;
}

float kvz_hip_batch_last_kernel_ms(kvz_hip_batch *b)
{
  float ms = 0;
  kvz::batch_enter(b);
  KVZ_HIP_CHECK(hipEventSynchronize(b->ev1));
  KVZ_HIP_CHECK(hipEventElapsedTime(&ms, b->ev0, b->ev1));
  return ms;
}
}
