// kvz_runtime.hpp -- process-wide HIP state of libkvz_hip.so: device selection, constant tables, error policy.
#pragma once
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <atomic>
#include <mutex>

#include "kvz_tables.hpp"

#define KVZ_HIP_CHECK(expr)                                                                              \
  do {                                                                                                   \
    hipError_t e_ = (expr);                                                                              \
    if (e_ != hipSuccess) {                                                                              \
      fprintf(stderr, "kvz_hip: %s failed: %s (%s:%d) -- no CPU fallback, aborting\n", #expr,            \
              hipGetErrorString(e_), __FILE__, __LINE__);                                                \
      abort();                                                                                           \
    }                                                                                                    \
  } while (0)

namespace kvz {

struct Runtime {
  int device = -1;
  Tables *d_tables = nullptr;
  Tables h_tables;
  std::atomic<unsigned long long> calls{0};  // per-call entry points served (kvz_hip_call_count)
};

inline Runtime &runtime()
{
  static Runtime r;
  return r;
}

// Binds the process to a device (first caller wins) and uploads the constant tables.  device < 0: $KVZ_HIP_DEVICE,
// else $LOCAL_RANK (one process per GPU under torch.distributed.run), else 0.  Every thread that touches HIP must
// also call hipSetDevice, which is why this runs at the top of every backend constructor.
inline void runtime_init(int device)
{
  static std::once_flag once;
  Runtime &r = runtime();
  std::call_once(once, [&]() {
    int n = 0;
    KVZ_HIP_CHECK(hipGetDeviceCount(&n));
    if (n <= 0) {
      fprintf(stderr, "kvz_hip: no HIP device visible -- the hip strategy has no CPU fallback, aborting\n");
      abort();
    }
    if (device < 0) {
      const char *e = getenv("KVZ_HIP_DEVICE");
      if (!e) e = getenv("LOCAL_RANK");
      device = e ? atoi(e) % n : 0;
    }
    r.device = device;
    KVZ_HIP_CHECK(hipSetDevice(device));
    build_tables(&r.h_tables);
    KVZ_HIP_CHECK(hipMalloc((void **)&r.d_tables, sizeof(Tables)));
    KVZ_HIP_CHECK(hipMemcpy(r.d_tables, &r.h_tables, sizeof(Tables), hipMemcpyHostToDevice));
    if (getenv("KVZ_HIP_STATS")) atexit([]() { fprintf(stderr, "kvz_hip: %llu strategy calls served on device %d\n", runtime().calls.load(), runtime().device); });
  });
  KVZ_HIP_CHECK(hipSetDevice(r.device));
}

inline const Tables *device_tables() { return runtime().d_tables; }

}  // namespace kvz
