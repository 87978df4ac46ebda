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
};
HostBackend &be() { static thread_local HostBackend b; return b; }
typedef kvz::Api<HostBackend> A;
}  // namespace

#define KVZ_API_PREFIX(name) kvz_hostsim_##name
#define KVZ_API_BACKEND be()
#include "../../kvazaar_amd/csrc/kvz_capi_exports.inc"

// ---- the batched CTU program (kvz_ctu.hpp) run on the host: CTUs in raster order, each one as 256 looped "threads" ----
#include "../../kvazaar_amd/csrc/kvz_ctu.hpp"
extern "C" void kvz_hostsim_intra_frame(const kvz_hip_intra_cost_model *m, int width, int height, const uint8_t *src /* Y|U|V */, uint8_t *rec,
                                        int16_t *coeff, uint8_t *cu_depth, uint8_t *cu_mode, double *ctu_cost)
{
  static kvz::Tables tb;
  kvz::build_tables(&tb);
  kvz::CtuFrames F;
  F.W = width; F.H = height; F.wc = (width + 63) / 64; F.hc = (height + 63) / 64; F.frame_px = (long)width * height * 3 / 2;
  F.src = src; F.rec = rec; F.coeff = coeff; F.cu_depth = cu_depth; F.cu_mode = cu_mode; F.ctu_cost = ctu_cost; F.prof = nullptr;
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
      if (m->coeff_cabac) {
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
extern "C" unsigned kvz_hostsim_ctu_shared_bytes(void) { return (unsigned)sizeof(kvz::CtuShared); }

#ifdef KVZ_HOSTSIM_COUNT_SYNCS
extern "C" unsigned long long kvz_hostsim_syncs(void) { return g_kvz_syncs; }
#endif
