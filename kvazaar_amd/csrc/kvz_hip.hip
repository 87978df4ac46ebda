// kvz_hip.hip -- libkvz_hip.so: HIP backend of the per-call strategy API (include/kvz_hip.h groups 1 and 2).
//
// One HipBackend per calling thread (thread_local, created on first use): a non-blocking HIP stream, a pinned
// host staging arena and a device arena of the same size.  A call = gather inputs into staging -> ONE
// hipMemcpyAsync H2D -> a few item-parallel kernels (kvz_ops.hpp) -> ONE hipMemcpyAsync D2H -> hipStreamSynchronize.
// kvazaar calls these from N pthread workers concurrently (threadqueue.c:275); nothing is shared between threads
// except the read-only constant tables.
//
// gfx950 only.  No CPU fallback: any HIP failure is fatal (message on stderr + abort()).
#include <hip/hip_runtime.h>

#include <mutex>

#include "../../include/kvz_hip.h"
#include "kvz_api_impl.hpp"
#include "kvz_arena.hpp"
#include "kvz_runtime.hpp"

namespace kvz {

template <class Op> __global__ void __launch_bounds__(256) item_kernel(const Op op, const int n)
{
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) op(i);
}

// one wavefront per item: op.wave(item, lane) with wavefront-uniform arguments (RdoqOp)
template <class Op> __global__ void __launch_bounds__(64) wave_item_kernel(const Op op) { op.wave((int)blockIdx.x, (int)threadIdx.x); }

struct HipBackend : ArenaBase {
  hipStream_t stream = nullptr;
  const Tables *tb = nullptr;
  HipBackend()  // on the calling thread's current device
  {
    cap = 1u << 20;
    KVZ_HIP_CHECK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    KVZ_HIP_CHECK(hipHostMalloc((void **)&h, cap, hipHostMallocDefault));
    KVZ_HIP_CHECK(hipMalloc((void **)&d, cap));
    tb = device_tables();
  }
  // released by the owning thread's ThreadState when a worker thread exits (be() below), never from a static destructor
  void release()
  {
    if (stream) { (void)hipStreamSynchronize(stream); (void)hipStreamDestroy(stream); }
    if (h) (void)hipHostFree(h);
    if (d) (void)hipFree(d);
    stream = nullptr; h = d = nullptr;
  }
  void begin() { reset(); runtime().calls.fetch_add(1, std::memory_order_relaxed); }
  void upload()
  {
    if (up_end) KVZ_HIP_CHECK(hipMemcpyAsync(d, h, up_end, hipMemcpyHostToDevice, stream));
  }
  void download()
  {
    if (dl_end > dl_begin) KVZ_HIP_CHECK(hipMemcpyAsync(h + dl_begin, d + dl_begin, dl_end - dl_begin, hipMemcpyDeviceToHost, stream));
    KVZ_HIP_CHECK(hipStreamSynchronize(stream));
  }
  const Tables *tables() { return tb; }
  template <class Op> void run(const Op &op, int n)
  {
    if (n <= 0) return;
    hipLaunchKernelGGL(item_kernel<Op>, dim3((n + 255) / 256), dim3(256), 0, stream, op, n);
    KVZ_HIP_CHECK(hipGetLastError());
  }
  template <class Op> void run_wave(const Op &op, int n)
  {
    if (n <= 0) return;
    hipLaunchKernelGGL(wave_item_kernel<Op>, dim3(n), dim3(64), 0, stream, op);
    KVZ_HIP_CHECK(hipGetLastError());
  }
};

// The calling thread's backend on its current device (one per thread and device: a thread that drives batches on several devices has a stream on each)
static HipBackend &be()
{
  static thread_local HipBackend *per_device[64] = {};
  runtime_init(-1);
  const int dev = current_device() & 63;
  HipBackend *&b = per_device[dev];
  if (!b) {
    b = new HipBackend();
    HipBackend **slot = &b;
    thread_state().cleanups.push_back([slot, dev]() { (void)hipSetDevice(dev); (*slot)->release(); delete *slot; *slot = nullptr; });
  }
  return *b;
}
typedef Api<HipBackend> A;

}  // namespace kvz

using kvz::A;
using kvz::be;

#define KVZ_API_PREFIX(name) kvz_hip_##name
#define KVZ_API_BACKEND be()
#include "kvz_capi_exports.inc"

// ---- typedef-exact entry points (include/kvz_hip.h group 1) ------------------------------------------------------
extern "C" {
#define KVZ_NXN(n)                                                                                                              \
  unsigned kvz_hip_sad_##n##x##n(const uint8_t *a, const uint8_t *b) { return kvz_hip_sad_nxn(n, a, b); }                       \
  unsigned kvz_hip_satd_##n##x##n(const uint8_t *a, const uint8_t *b) { return kvz_hip_satd_nxn(n, a, b); }                     \
  void kvz_hip_sad_##n##x##n##_dual(const uint8_t(*p)[1024], const uint8_t *o, unsigned m, unsigned *c) { kvz_hip_sad_nxn_dual(n, &p[0][0], o, m, c); } \
  void kvz_hip_satd_##n##x##n##_dual(const uint8_t(*p)[1024], const uint8_t *o, unsigned m, unsigned *c) { kvz_hip_satd_nxn_dual(n, &p[0][0], o, m, c); }
KVZ_NXN(4)
KVZ_NXN(8)
KVZ_NXN(16)
KVZ_NXN(32)
KVZ_NXN(64)

#define KVZ_TR(name, kind) \
  void kvz_hip_##name(int8_t bitdepth, const int16_t *in, int16_t *out) { kvz_hip_transform(kind, bitdepth, in, out); }
KVZ_TR(dct_4x4, KVZ_HIP_DCT_4)
KVZ_TR(dct_8x8, KVZ_HIP_DCT_8)
KVZ_TR(dct_16x16, KVZ_HIP_DCT_16)
KVZ_TR(dct_32x32, KVZ_HIP_DCT_32)
KVZ_TR(fast_forward_dst_4x4, KVZ_HIP_DST_4)
KVZ_TR(idct_4x4, KVZ_HIP_IDCT_4)
KVZ_TR(idct_8x8, KVZ_HIP_IDCT_8)
KVZ_TR(idct_16x16, KVZ_HIP_IDCT_16)
KVZ_TR(idct_32x32, KVZ_HIP_IDCT_32)
KVZ_TR(fast_inverse_dst_4x4, KVZ_HIP_IDST_4)

void kvz_hip_array_checksum(const uint8_t *data, const int height, const int width, const int stride, unsigned char checksum_out[16],
                            const uint8_t bitdepth)
{
  (void)bitdepth;  // 8-bit build (the registration shim does not register otherwise)
  const uint32_t v = kvz_hip_plane_checksum(data, height, width, stride);
  checksum_out[0] = (unsigned char)(v >> 24); checksum_out[1] = (unsigned char)(v >> 16);
  checksum_out[2] = (unsigned char)(v >> 8); checksum_out[3] = (unsigned char)v;
}

void kvz_hip_array_md5(const uint8_t *data, const int height, const int width, const int stride, unsigned char checksum_out[16], const uint8_t bitdepth)
{
  (void)bitdepth;
  kvz_hip_plane_md5(data, height, width, stride, checksum_out);
}

// get_optimized_sad (strategies-picture.h:128): the widths kvazaar's PUs can have (square, SMP and AMP partitions)
#define KVZ_OPT_SAD(w) \
  static uint32_t opt_sad_##w(const uint8_t *pic, const uint8_t *ref, int32_t height, uint32_t s1, uint32_t s2) { return kvz_hip_reg_sad(pic, ref, w, height, s1, s2); }
KVZ_OPT_SAD(4)
KVZ_OPT_SAD(8)
KVZ_OPT_SAD(12)
KVZ_OPT_SAD(16)
KVZ_OPT_SAD(24)
KVZ_OPT_SAD(32)
KVZ_OPT_SAD(48)
KVZ_OPT_SAD(64)
kvz_hip_optimized_sad_fn kvz_hip_get_optimized_sad(int32_t width)
{
  switch (width) {
    case 4: return opt_sad_4;
    case 8: return opt_sad_8;
    case 12: return opt_sad_12;
    case 16: return opt_sad_16;
    case 24: return opt_sad_24;
    case 32: return opt_sad_32;
    case 48: return opt_sad_48;
    case 64: return opt_sad_64;
    default: return nullptr;
  }
}
}

extern "C" {
int kvz_hip_device_count(void)
{
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}
int kvz_hip_init(int device)
{
  kvz::runtime_init(device);
  return 1;
}
int kvz_hip_set_thread_device(int device) { return kvz::thread_set_device(device); }
int kvz_hip_thread_device(void) { kvz::runtime_init(-1); return kvz::current_device(); }
const char *kvz_hip_version(void) { return "kvz_hip 0.1 (gfx950)"; }
unsigned long long kvz_hip_call_count(void) { return kvz::runtime().calls.load(); }
}

// ---- batched, device-resident path (include/kvz_hip_batch.h) ------------------------------------------------------
#include "kvz_batch.hpp"
#include "kvz_dev.hpp"
