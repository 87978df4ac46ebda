// hostsim.cpp -- TEST INFRASTRUCTURE.  Compiles the device ops (kvazaar_amd/csrc/kvz_ops.hpp) and the per-call
// sequences (kvz_api_impl.hpp) for the HOST with every op executed by a plain loop, and exports them as
// kvz_hostsim_* with the flat API signatures.  It lets tests/test_hostsim.py check the index arithmetic of every
// kernel against the oracle on a machine without a GPU.  It is NOT part of the product: libkvz_hip.so has no host
// compute path and kvazaar_amd never loads this library.
#define KVZ_HOSTSIM 1
#include "../../kvazaar_amd/csrc/kvz_api_impl.hpp"
#include "../../kvazaar_amd/csrc/kvz_arena.hpp"

namespace {
struct HostBackend : kvz::ArenaBase {
  kvz::Tables tb;
  HostBackend()
  {
    cap = 4u << 20;
    h = d = (uint8_t *)malloc(cap);
    kvz::build_tables(&tb);
  }
  void begin() { reset(); }
  void upload() {}
  void download() {}
  const kvz::Tables *tables() { return &tb; }
  template <class Op> void run(const Op &op, int n) { for (int i = 0; i < n; i++) op(i); }
  template <class Op> void run_wave(const Op &op, int n) { for (int i = 0; i < n; i++) op.wave(i, 0); }
};
HostBackend &be() { static thread_local HostBackend b; return b; }
typedef kvz::Api<HostBackend> A;
}  // namespace

#define KVZ_API_PREFIX(name) kvz_hostsim_##name
#define KVZ_API_BACKEND be()
#include "../../kvazaar_amd/csrc/kvz_capi_exports.inc"

// ---- the batched CTU program (kvz_ctu.hpp) run on the host: CTUs in raster order, each one as 256 looped "threads" ----
#include "../../kvazaar_amd/csrc/kvz_ctu.hpp"
extern "C" void kvz_hostsim_intra_frame_nxn(const kvz_hip_intra_cost_model *m, int width, int height, const uint8_t *src /* Y|U|V */, uint8_t *rec,
                                            int16_t *coeff, uint8_t *cu_depth, uint8_t *cu_mode, double *ctu_cost, uint8_t *cu_part, uint8_t *cu_mode4)
{
  static kvz::Tables tb;
  kvz::build_tables(&tb);
  kvz::CtuFrames F;
  F.W = width; F.H = height; F.wc = (width + 63) / 64; F.hc = (height + 63) / 64; F.frame_px = (long)width * height * 3 / 2;
  F.src = src; F.rec = rec; F.coeff = coeff; F.cu_depth = cu_depth; F.cu_mode = cu_mode; F.ctu_cost = ctu_cost; F.prof = nullptr;
  F.cu_part = cu_part; F.cu_mode4 = cu_mode4;
  uint8_t *border = (uint8_t *)calloc((size_t)F.wc * F.hc, KVZ_BORDER_BYTES);
  F.border = border;
  int16_t *scratch = (int16_t *)calloc((size_t)F.wc * F.hc * 6144, sizeof(int16_t));
  F.coeff_scratch = scratch;
  // like the device, two instantiations of the program: with and without the CABAC coefficient model (kvz_batch.hpp picks by model)
  void *sh = calloc(1, sizeof(kvz::CtuSharedT<true>) > sizeof(kvz::CtuSharedT<false>) ? sizeof(kvz::CtuSharedT<true>) : sizeof(kvz::CtuSharedT<false>));
  kvz::CtuModel cm;
  kvz::ctu_model_from(m, &cm);
  for (int cy = 0; cy < F.hc; cy++)
    for (int cx = 0; cx < F.wc; cx++) {
      if (m->rdoq || m->search_nxn) {  // --rdoq / NxN partitions: the instantiation with kvz_rdoq in the quantisation stage and depth 4 of the search (coefficient cost model and 32x32 search switched by the model)
        static kvz::RdoqLds rdoq_lds;
        kvz::CtuProgramT<true, true, true> p;
        p.rl = &rdoq_lds;
        p.m = &cm; p.tb = &tb; p.F = F; p.s = (kvz::CtuSharedT<true> *)sh; p.frame = 0; p.cx = cx * 64; p.cy = cy * 64;
        p.run();
      } else if (m->search_32x32) {  // the instantiations that search 32x32 CUs
        if (m->coeff_cabac) {
          kvz::CtuProgramT<true, true> p;
          p.m = &cm; p.tb = &tb; p.F = F; p.s = (kvz::CtuSharedT<true> *)sh; p.frame = 0; p.cx = cx * 64; p.cy = cy * 64;
          p.run();
        } else {
          kvz::CtuProgramT<false, true> p;
          p.m = &cm; p.tb = &tb; p.F = F; p.s = (kvz::CtuSharedT<false> *)sh; p.frame = 0; p.cx = cx * 64; p.cy = cy * 64;
          p.run();
        }
      } else if (m->coeff_cabac) {
        kvz::CtuProgramT<true> p;
        p.m = &cm; p.tb = &tb; p.F = F; p.s = (kvz::CtuSharedT<true> *)sh; p.frame = 0; p.cx = cx * 64; p.cy = cy * 64;
        p.run();
      } else {
        kvz::CtuProgramT<false> p;
        p.m = &cm; p.tb = &tb; p.F = F; p.s = (kvz::CtuSharedT<false> *)sh; p.frame = 0; p.cx = cx * 64; p.cy = cy * 64;
        p.run();
      }
    }
  free(sh);
  free(scratch);
  free(border);
}
extern "C" void kvz_hostsim_intra_frame(const kvz_hip_intra_cost_model *m, int width, int height, const uint8_t *src, uint8_t *rec, int16_t *coeff, uint8_t *cu_depth,
                                        uint8_t *cu_mode, double *ctu_cost)
{
  kvz_hostsim_intra_frame_nxn(m, width, height, src, rec, coeff, cu_depth, cu_mode, ctu_cost, nullptr, nullptr);
}
extern "C" unsigned kvz_hostsim_ctu_shared_bytes(void) { return (unsigned)sizeof(kvz::CtuShared); }

// ---- the SAO parameter decision (kvz_sao.hpp) on the host: statistics by plain loops over the view, then the device's own candidate and
// chain code.  R / V / D: the reconstruction before deblocking, after the vertical edges, after all edges (Y|U|V tight). ----
#include "../../kvazaar_amd/csrc/kvz_sao.hpp"
#include <vector>
extern "C" void kvz_hostsim_sao_decide(const kvz_hip_intra_cost_model *m, int width, int height, const uint8_t *src, const uint8_t *R, const uint8_t *V, const uint8_t *D,
                                       kvz_hip_sao_params *luma, kvz_hip_sao_params *chroma, uint8_t *merge)
{
  using namespace kvz;
  static Tables tb;
  build_tables(&tb);
  const int wl = (width + 63) / 64, hl = (height + 63) / 64;
  std::vector<SaoStats> stats((size_t)wl * hl * 3);
  std::vector<SaoCand> cand((size_t)wl * hl * 3);
  std::vector<SaoRec> recs((size_t)wl * hl * 3);
  for (int lcu = 0; lcu < wl * hl; lcu++)
    for (int color = 0; color < 3; color++) {
      const int lx = lcu % wl, ly = lcu / wl, sh = color ? 1 : 0, n = 64 >> sh, fw = width >> sh, fh = height >> sh;
      const long plane = color == 0 ? 0 : (color == 1 ? (long)width * height : (long)width * height * 5 / 4);
      const int bw = imin(n, fw - lx * n), bh = imin(n, fh - ly * n);
      SaoView view{ R + plane, V + plane, D + plane, fw, n, lx * n, ly * n, lx == wl - 1, ly == hl - 1, color ? 1 : 3 };
      SaoStats &st = stats[(size_t)lcu * 3 + color];
      memset(&st, 0, sizeof st);
      for (int y = 0; y < bh; y++)
        for (int x = 0; x < bw; x++) {
          const int c = view.at(x, y), diff = (int)src[plane + (long)(ly * n + y) * fw + lx * n + x] - c;
          st.band_sum[c >> 3] += diff;
          st.band_cnt[c >> 3]++;
          if (x >= 1 && x < bw - 1 && y >= 1 && y < bh - 1)
            for (int ec = 0; ec < 4; ec++) {
              int ax, ay, bx, by;
              eo_offsets(ec, ax, ay, bx, by);
              const int cat = eo_cat(view.at(x + ax, y + ay), view.at(x + bx, y + by), c);
              st.edge_sum[ec][cat] += diff;
              st.edge_cnt[ec][cat]++;
            }
        }
      sao_candidates(st, cand[(size_t)lcu * 3 + color]);
    }
  sao_chain_picture(m->entropy_fbits, tb.ctx_next[0], tb.ctx_next[1], m->lambda, m->ctx_init[KVZ_HIP_CX_SAO_MERGE], m->ctx_init[KVZ_HIP_CX_SAO_TYPE], m->no_wpp, wl, hl, stats.data(),
                    cand.data(), recs.data(), merge);
  for (int i = 0; i < wl * hl; i++) {
    auto unpack = [&](kvz_hip_sao_params *o, int plane, int slot) {
      const SaoRec r = recs[(size_t)i * 3 + plane];
      if (slot == 0) { memset(o, 0, sizeof *o); o->bitdepth = 8; o->type = (int)(r & 0xff); o->eo_class = (int)((r >> 8) & 0xff); }
      o->band_position[slot] = (int)((r >> 16) & 0xff);
      for (int k = 0; k < 5; k++) o->offsets[5 * slot + k] = (int)(int8_t)(r >> (24 + 8 * k));
    };
    unpack(&luma[i], 0, 0);
    unpack(&chroma[i], 1, 0);
    unpack(&chroma[i], 2, 1);
  }
}

#ifdef KVZ_HOSTSIM_COUNT_SYNCS
extern "C" unsigned long long kvz_hostsim_syncs(void) { return g_kvz_syncs; }
#endif

// ---- the inter CTU pass (kvz_inter_ctu.hpp) run on the host: one B picture, CTUs in raster order, every phase a loop over its 256 "lanes" ----
#include "../../kvazaar_amd/csrc/kvz_inter_host.hpp"
// the pictures handed over are ONE TILE of a ref_w x ref_h frame at (tile_x, tile_y) (ref / ref_cu: that frame; all zero: the picture is the frame)
extern "C" void kvz_hostsim_inter_tile(int width, int height, int qp, int poc, uint64_t coeff_weights, const float *fbits, int mv_constraint, int sao, int deblock, int fme_level,
                                       int pu_depth_inter_max, int no_wpp, int fast_residual_cost, const uint8_t *src, const uint8_t *ref, const kvz_hip_cu_info *ref_cu, uint8_t *rec,
                                       kvz_hip_cu_info *cu, int ref_w, int ref_h, int tile_x, int tile_y, int no_tmvp);
extern "C" void kvz_hostsim_inter_frame(int width, int height, int qp, int poc, uint64_t coeff_weights, const float *fbits, int mv_constraint, int sao, int deblock, int fme_level,
                                        int pu_depth_inter_max, int no_wpp, int fast_residual_cost, const uint8_t *src, const uint8_t *ref, const kvz_hip_cu_info *ref_cu, uint8_t *rec,
                                        kvz_hip_cu_info *cu)
{
  kvz_hostsim_inter_tile(width, height, qp, poc, coeff_weights, fbits, mv_constraint, sao, deblock, fme_level, pu_depth_inter_max, no_wpp, fast_residual_cost, src, ref, ref_cu, rec, cu, 0, 0, 0, 0, 0);
}
extern "C" void kvz_hostsim_inter_tile(int width, int height, int qp, int poc, uint64_t coeff_weights, const float *fbits, int mv_constraint, int sao, int deblock, int fme_level,
                                       int pu_depth_inter_max, int no_wpp, int fast_residual_cost, const uint8_t *src, const uint8_t *ref, const kvz_hip_cu_info *ref_cu, uint8_t *rec,
                                       kvz_hip_cu_info *cu, int ref_w, int ref_h, int tile_x, int tile_y, int no_tmvp)
{
  static kvz::Tables tb;
  kvz::build_tables(&tb);
  kvz::InterModel m;
  kvz::inter_model_init(&m, qp, poc, coeff_weights, fbits, mv_constraint, sao, deblock, fme_level, pu_depth_inter_max, no_wpp, fast_residual_cost, width, height, ref_w, ref_h, tile_x, tile_y, no_tmvp);
  kvz::InterFrames F;
  memset(&F, 0, sizeof F);
  F.W = width; F.H = height; F.wc = (width + 63) / 64; F.hc = (height + 63) / 64; F.frame_px = (long)width * height * 3 / 2; F.cells = (long)(width / 4) * (height / 4);
  F.src = src; F.ref = ref; F.ref_cu = ref_cu; F.rec = rec; F.cu = cu; F.coeff = nullptr;
  F.ctx_out = (kvz::ICtx *)calloc((size_t)F.wc * F.hc, sizeof(kvz::ICtx));
  kvz::InterSlab *slab = (kvz::InterSlab *)calloc(1, sizeof(kvz::InterSlab));
  F.slabs = slab;
  memset(cu, 0, (size_t)F.cells * sizeof(kvz_hip_cu_info));
  kvz::InterCtu::begin_launch(F, &m, &tb, slab);
  for (int cy = 0; cy < F.hc; cy++)
    for (int cx = 0; cx < F.wc; cx++) {
      kvz::InterCtu::begin_ctu(0, cx * 64, cy * 64);
      kvz::InterCtu::run();
    }
  free(F.ctx_out); free(slab);
}
#ifdef KVZ_ICTU_COUNT_PHASES
extern "C" void kvz_hostsim_inter_phases(long *out) { for (int i = 0; i < 32; i++) { out[i] = kvz::g_ic_phases[i]; kvz::g_ic_phases[i] = 0; } }
#endif

// ---- the entropy coder's three stages (kvz_entropy.hpp) run on the host: every lane a loop iteration ----
#include "../../kvazaar_amd/csrc/kvz_entropy.hpp"
// stage 1 of the entropy coder as the device runs it by default (a lane = a call: the phased walk's votes are the lane's own), or the serial walk (KVZ_HIP_ENTROPY_BINS=serial)
static void hostsim_ctu_bins(const kvz::EntropyJob &J, const kvz::Tables *tb, long item)
{
  static const bool serial = [] { const char *e = getenv("KVZ_HIP_ENTROPY_BINS"); return e && !strcmp(e, "serial"); }();
  if (serial) { kvz::entropy_ctu_bins(J, tb, item); return; }
  uint32_t queue[kvz::DeferSink::QCAP];
  uint16_t stack[16];
  kvz::entropy_ctu_bins_phased(J, tb, item, true, queue, stack, 1);
}

// stage 3 as the device runs it: the wide coder (32-bit units; KVZ_HOSTSIM_WIDE_UNIT=8: byte units, where carries into all-ones units are common;
// KVZ_HOSTSIM_WIDE_EARLY: units leave as early as they may instead of as late as they must), then emulation prevention by the serial rule of bitstream.c:212-223.
// out null: count only.
static uint32_t hostsim_code_row(const kvz::EntropyJob &J, const kvz::Tables &tb, long item, uint8_t *out, size_t room)
{
  static const bool unit8 = [] { const char *e = getenv("KVZ_HOSTSIM_WIDE_UNIT"); return e && atoi(e) == 8; }();
  static unsigned long long tab[128];
  for (int i = 0; i < 128; i++) tab[i] = kvz::entropy_state_entry(&tb.ctx_next[0][0], i);
  uint8_t ctx[KVZ_ENTROPY_CTX_STRIDE];
  std::vector<uint8_t> raw(room + 64);
  const uint32_t n = unit8 ? kvz::entropy_code_row_wide<8>(J, tab, item, ctx, raw.data()) : kvz::entropy_code_row_wide<32>(J, tab, item, ctx, raw.data());
  uint32_t o = 0;
  int zerocount = 0;
  for (uint32_t i = 0; i < n; i++) {
    const uint8_t b = raw[i];
    if (zerocount == 2 && b < 4) { if (out) out[o] = 3; o++; zerocount = 0; }
    zerocount = b == 0 ? zerocount + 1 : 0;
    if (out) out[o] = b;
    o++;
  }
  return o;
}
// the device's emulation prevention pass (escape count by position, then 256 chunks written at their offsets) on n bytes; returns the size, or -1 if the two rules disagree
extern "C" long kvz_hostsim_escape(const uint8_t *raw, uint32_t n, uint8_t *out)
{
  uint32_t by_position = 0;
  for (uint32_t i = 0; i < n; i++) by_position += kvz::entropy_escape_at(raw, i) ? 1u : 0u;
  const uint32_t chunk = (n + 255) / 256;
  uint32_t before = 0;
  for (uint32_t t = 0; t < 256; t++) {
    const uint32_t lo = t * chunk < n ? t * chunk : n, hi = lo + chunk < n ? lo + chunk : n;
    const uint32_t counted = kvz::entropy_escape_chunk(raw, lo, hi, nullptr);
    if (kvz::entropy_escape_chunk(raw, lo, hi, out + lo + before) != counted) return -1;
    before += counted;
  }
  return before == by_position ? (long)n + before : -1;
}

extern "C" long kvz_hostsim_entropy_code(const kvz_hip_intra_cost_model *m, int width, int height, int n_frames, const uint8_t *cu_depth, const uint8_t *cu_mode, const uint8_t *part,
                                         const uint8_t *mode4, const int16_t *coeff, const unsigned long long *sao_recs, const uint8_t *sao_merge, uint32_t cap, uint8_t *out,
                                         uint32_t *substream_bytes, uint32_t *most_records)
{
  static kvz::Tables tb;
  kvz::build_tables(&tb);
  kvz::EntropyJob J;
  memset(&J, 0, sizeof J);
  J.W = width; J.H = height; J.wc = (width + 63) / 64; J.hc = (height + 63) / 64; J.n_frames = n_frames; J.no_wpp = m->no_wpp;
  J.depth = cu_depth; J.mode = cu_mode; J.part = part; J.mode4 = mode4; J.coeff = coeff; J.sao = sao_recs; J.sao_merge = sao_merge;
  const long items = (long)n_frames * J.wc * J.hc, streams = (long)n_frames * (m->no_wpp ? 1 : J.hc);
  cap = (cap + 15u) & ~15u;
  J.bins = (uint32_t *)aligned_alloc(64, (size_t)items * cap * sizeof(uint32_t)); J.nbins = (uint32_t *)malloc((size_t)items * sizeof(uint32_t)); J.nbits = (uint32_t *)malloc((size_t)items * sizeof(uint32_t)); J.cap = cap;
  J.row_ctx = (uint8_t *)malloc((size_t)n_frames * J.hc * KVZ_ENTROPY_CTXS);
  memcpy(J.ctx_init, m->ctx_init, sizeof m->ctx_init < sizeof J.ctx_init ? sizeof m->ctx_init : sizeof J.ctx_init);
  const kvz::EntropyTabs T{ &tb.ctx_next[0][0] };
  uint8_t ctx[KVZ_ENTROPY_CTXS];
  *most_records = 0;
  for (long i = 0; i < items; i++) { hostsim_ctu_bins(J, &tb, i); if (J.nbins[i] > *most_records) *most_records = J.nbins[i]; }
  long total = -1;
  if (*most_records <= cap) {
    if (!m->no_wpp) for (int f = 0; f < n_frames; f++) kvz::entropy_row_contexts(J, T, f, ctx);
    total = 0;
    for (long i = 0; i < streams; i++) {
      // the bound the device sizes its scratch with (kvz_hip_batch_entropy_code) must hold
      unsigned long long bits = 0;
      const long per_stream = m->no_wpp ? (long)J.wc * J.hc : J.wc;
      for (long k = 0; k < per_stream; k++) bits += J.nbits[i * per_stream + k];
      const size_t room = (size_t)(((bits + 7) / 8 + 16) * 3 / 2);
      const uint32_t counted = hostsim_code_row(J, tb, i, nullptr, room);
      substream_bytes[i] = hostsim_code_row(J, tb, i, out + total, room);
      if (counted != substream_bytes[i]) { total = -2; break; }
      if (substream_bytes[i] > ((bits + 7) / 8 + 16) * 3 / 2) { total = -3; break; }
      total += substream_bytes[i];
    }
  }
  free(J.bins); free(J.nbins); free(J.nbits); free(J.row_ctx);
  return total;
}

// ... and for B pictures: the CU records of a picture and of its reference picture, levels, packed SAO decisions
extern "C" long kvz_hostsim_entropy_code_inter(const uint8_t *ctx_init /* KVZ_ENTROPY_CTXS states */, int width, int height, int poc, int no_wpp, const kvz_hip_cu_info *cu,
                                               const kvz_hip_cu_info *ref_cu, const int16_t *coeff, const unsigned long long *sao_recs, const uint8_t *sao_merge, uint32_t cap,
                                               uint8_t *out, uint32_t *substream_bytes)
{
  static kvz::Tables tb;
  kvz::build_tables(&tb);
  kvz::EntropyJob J;
  memset(&J, 0, sizeof J);
  J.W = width; J.H = height; J.wc = (width + 63) / 64; J.hc = (height + 63) / 64; J.n_frames = 1; J.no_wpp = no_wpp;
  J.cu = cu; J.ref_cu = ref_cu; J.poc = poc; J.coeff = coeff; J.sao = sao_recs; J.sao_merge = sao_merge;
  const long items = (long)J.wc * J.hc, streams = no_wpp ? 1 : J.hc;
  cap = (cap + 15u) & ~15u;
  J.bins = (uint32_t *)aligned_alloc(64, (size_t)items * cap * sizeof(uint32_t)); J.nbins = (uint32_t *)malloc((size_t)items * sizeof(uint32_t));
  J.nbits = (uint32_t *)malloc((size_t)items * sizeof(uint32_t)); J.cap = cap;
  J.row_ctx = (uint8_t *)malloc((size_t)J.hc * KVZ_ENTROPY_CTXS);
  memcpy(J.ctx_init, ctx_init, KVZ_ENTROPY_CTXS);
  const kvz::EntropyTabs T{ &tb.ctx_next[0][0] };
  uint8_t ctx[KVZ_ENTROPY_CTXS];
  long total = 0;
  for (long i = 0; i < items; i++) { hostsim_ctu_bins(J, &tb, i); if (J.nbins[i] > cap) total = -1; }
  if (total == 0) {
    if (!no_wpp) kvz::entropy_row_contexts(J, T, 0, ctx);
    const long per_stream = no_wpp ? items : J.wc;
    for (long i = 0; i < streams; i++) {
      unsigned long long bits = 0;
      for (long k = 0; k < per_stream; k++) bits += J.nbits[i * per_stream + k];
      substream_bytes[i] = hostsim_code_row(J, tb, i, out + total, (size_t)(((bits + 7) / 8 + 16) * 3 / 2));
      total += substream_bytes[i];
    }
  }
  free(J.bins); free(J.nbins); free(J.nbits); free(J.row_ctx);
  return total;
}
