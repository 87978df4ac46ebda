// kvz_inter_kernels.hpp -- the __global__ entry point of the inter CTU pass (kvz_inter_ctu.hpp) under the in-order ticket schedule of kvz_ctu_kernels.hpp: one
// persistent workgroup per resident slot draws CTUs (picture, x, y) from a list in which every CTU follows the ones it depends on (its left and above-right
// neighbours in the same picture: CU info, reconstruction and the row coder's contexts), waits for their `done` flags and runs the program on the slab of its slot.
// Compiled in a translation unit of its own (kvz_inter_tu.hip); kvz_hip.hip only sees the declaration.
#pragma once
#include <hip/hip_runtime.h>

#include "kvz_inter_ctu.hpp"

namespace kvz {

struct InterSched {
  const uint32_t *items;  // [total]: picture << 16 | y << 8 | x
  unsigned *ticket;       // zeroed before the launch
  unsigned *done;         // [pictures * CTUs], zeroed before the launch
  unsigned *error;
  unsigned total;
  int no_wpp;
  unsigned long long wait_ticks;
};

// two builds of the kernel (kvz_inter_ctu.hpp KVZ_ICTU_CABAC): `_fast` for pictures whose coefficients are priced by kvz_fast_coeff_cost (small context sets: 20 KB of LDS,
// eight workgroups per CU), `_cabac` for those priced with the residual coder's contexts
#define KVZ_ICTU_KERNEL(name) __global__ void __launch_bounds__(KVZ_ICTU_THREADS) __attribute__((amdgpu_waves_per_eu(KVZ_ICTU_WAVES_PER_EU, KVZ_ICTU_WAVES_PER_EU))) name(const InterFrames F, const InterModel *model, const Tables *tb, const InterSched sched)
#ifndef KVZ_INTER_KERNEL_BODY
KVZ_ICTU_KERNEL(inter_ctu_ticket_kernel_fast);
KVZ_ICTU_KERNEL(inter_ctu_ticket_kernel_cabac);
#else
#if KVZ_ICTU_CABAC
KVZ_ICTU_KERNEL(inter_ctu_ticket_kernel_cabac)
#else
KVZ_ICTU_KERNEL(inter_ctu_ticket_kernel_fast)
#endif
{
  __shared__ int s_ticket;
  const int ctus = F.wc * F.hc;
  InterCtu::begin_launch(F, model, tb, F.slabs + blockIdx.x);  // the program's state and constants: workgroup-scope variables in LDS (kvz_inter_ctu.hpp)
  for (;;) {
    __syncthreads();
    if (threadIdx.x == 0) s_ticket = (int)atomicAdd(sched.ticket, 1u);
    __syncthreads();
    const unsigned t = (unsigned)s_ticket;
    if (t >= sched.total) break;
    const uint32_t item = sched.items[t];
    const int frame = item >> 16, y = (item >> 8) & 0xff, x = item & 0xff;
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned *done = sched.done + (long)frame * ctus;
      auto wait = [&](unsigned *flag) -> bool {
        unsigned long long t0 = 0;
        for (unsigned spins = 0;; ++spins) {
          if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 1u) return true;
          if ((spins & 1023u) == 1023u) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // (as in the intra pass: kvz_ctu_kernels.hpp wait_done)
            const unsigned long long now = __builtin_amdgcn_s_memrealtime();
            if (!t0) t0 = now;
            else if (now - t0 > sched.wait_ticks) { atomicExch(sched.error, 1u); return false; }
          }
          if (spins < 1024u) __builtin_amdgcn_s_sleep(16);  // (backs off as the intra pass does: kvz_ctu_kernels.hpp wait_done)
          else { __builtin_amdgcn_s_sleep(127); __builtin_amdgcn_s_sleep(127); __builtin_amdgcn_s_sleep(127); __builtin_amdgcn_s_sleep(127); }
        }
      };
      bool ok = __hip_atomic_load(sched.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0;
      if (ok && x > 0) ok = wait(&done[y * F.wc + x - 1]);
      if (ok && y > 0) ok = wait(&done[(y - 1) * F.wc + (x + 1 < F.wc ? x + 1 : x)]);
      if (ok && sched.no_wpp && x == 0 && y > 0) ok = wait(&done[(y - 1) * F.wc + F.wc - 1]);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      s_ticket = ok ? 1 : 0;
    }
    __syncthreads();
    if (s_ticket != 0) {
      InterCtu::begin_ctu(frame, x * 64, y * 64);
      InterCtu::run();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_store(&sched.done[(long)frame * ctus + y * F.wc + x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}
#endif

}  // namespace kvz
