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
