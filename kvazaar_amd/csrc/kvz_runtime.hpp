// kvz_runtime.hpp -- process-wide HIP state of libkvz_hip.so: device selection, constant tables, error policy.
#pragma once
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <sys/syscall.h>
#include <unistd.h>

#include <atomic>
#include <functional>
#include <mutex>
#include <vector>

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

// Process-wide: the default device (first caller wins), the host copy of the constant tables and one device copy per device that has been used.
struct Runtime {
  int device = -1;      // the process default: kvz_hip_init's argument, $KVZ_HIP_DEVICE, $LOCAL_RANK, 0
  int n_devices = 0;
  Tables *d_tables[64] = {};
  std::mutex tables_mu;
  Tables h_tables;
  std::atomic<unsigned long long> calls{0};  // per-call entry points served (kvz_hip_call_count)
};

inline Runtime &runtime()
{
  static Runtime r;
  return r;
}

// Per calling thread: whether the thread has been bound to a device yet, and what it allocated lazily (per-device stream + staging arenas, the grow-only scratch of
// the kvz_hip_dev_* passes, timing events).  A WORKER thread that exits gives all of it back (kvazaar's threadqueue.c:275-355 workers live as long as the encoder,
// but an embedding application may churn threads); the main thread's set dies with the process -- its destructor would run inside exit(), next to the HIP runtime's own.
struct ThreadState {
  bool bound = false;
  std::vector<std::function<void()>> cleanups;  // run in reverse order of registration; HIP errors are ignored (nothing to report them to)
  ~ThreadState()
  {
    if ((long)syscall(SYS_gettid) == (long)getpid()) return;
    for (size_t i = cleanups.size(); i-- > 0;) cleanups[i]();
  }
};
inline ThreadState &thread_state()
{
  static thread_local ThreadState t;
  return t;
}

inline void runtime_once(int device)
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
    r.n_devices = n;
    r.device = device % n;
    build_tables(&r.h_tables);
    if (getenv("KVZ_HIP_STATS")) atexit([]() { fprintf(stderr, "kvz_hip: %llu strategy calls served on device %d\n", runtime().calls.load(), runtime().device); });
  });
}

// The HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues per device, four by default; streams that share one run in its order.  The
// double-buffered chain has five in flight (two batches, the coder's side stream, the upload queue, the thread's own): with four queues the coder's kernels sat behind
// 84 ms of another batch's pictures on their way up (tools/chain_h2d_trace.sh; with eight the upload costs nothing).  The runtime reads the variable when it initialises,
// at the first HIP call of the process: the library asks for eight when it is loaded, unless the host has set the variable itself.
struct HwQueuesDefault { HwQueuesDefault() { setenv("GPU_MAX_HW_QUEUES", "8", 0); } };
static HwQueuesDefault g_hw_queues_default;

// Every entry point that touches HIP starts here (directly or through be()): the first caller of the process picks the default device (device < 0: $KVZ_HIP_DEVICE,
// else $LOCAL_RANK -- one process per GPU under torch.distributed.run --, else 0); a thread that has not chosen a device yet (kvz_hip_set_thread_device, or a call
// on a batch, which binds the thread to the batch's device) is bound to the default.  A bound thread keeps the device it last selected.
inline void runtime_init(int device)
{
  runtime_once(device);
  ThreadState &t = thread_state();
  if (!t.bound) {
    KVZ_HIP_CHECK(hipSetDevice(runtime().device));
    t.bound = true;
  }
}

// include/kvz_hip.h kvz_hip_set_thread_device: the device of the calling thread's later calls
inline int thread_set_device(int device)
{
  runtime_once(-1);
  if (device < 0 || device >= runtime().n_devices) return 0;
  KVZ_HIP_CHECK(hipSetDevice(device));
  thread_state().bound = true;
  return 1;
}

inline int current_device()
{
  int d = 0;
  KVZ_HIP_CHECK(hipGetDevice(&d));
  return d;
}

// The constant tables on the calling thread's current device (uploaded the first time a device is used)
inline const Tables *device_tables()
{
  Runtime &r = runtime();
  const int d = current_device() & 63;
  Tables *t = __atomic_load_n(&r.d_tables[d], __ATOMIC_ACQUIRE);
  if (t) return t;
  std::lock_guard<std::mutex> lock(r.tables_mu);
  if (!r.d_tables[d]) {
    KVZ_HIP_CHECK(hipMalloc((void **)&t, sizeof(Tables)));
    KVZ_HIP_CHECK(hipMemcpy(t, &r.h_tables, sizeof(Tables), hipMemcpyHostToDevice));
    __atomic_store_n(&r.d_tables[d], t, __ATOMIC_RELEASE);
  }
  return r.d_tables[d];
}

}  // namespace kvz
