// kvz_ctu_kernels.hpp -- the __global__ entry points of the batched CTU pass (the program itself is kvz_ctu.hpp) and their scheduling protocol.
//
// Build layout: the six instantiations are the bulk of the library's compile time, so kvazaar_amd/build.py compiles each of them in a translation unit of
// its own (kvz_ctu_tu.hip with -DKVZ_CTU_KERNEL_TU=<k>), in parallel, next to kvz_hip.hip compiled with -DKVZ_CTU_SEPARATE_TUS -- which then only sees
// DECLARATIONS here and launches the kernels through their host stubs.  Without that define kvz_hip.hip is self-contained as before (tools/build_variants.sh).
#pragma once
#include <hip/hip_runtime.h>

#include "kvz_ctu.hpp"

#if defined(KVZ_CTU_SEPARATE_TUS) && !defined(KVZ_CTU_KERNEL_TU)
#define KVZ_CTU_KERNEL_BODIES 0
#else
#define KVZ_CTU_KERNEL_BODIES 1
#endif

namespace kvz {

// One workgroup per CTU of the anti-diagonal `wave` (x + 2y == wave) of every frame.
#ifdef KVZ_CTU_NUM_VGPR  /* register-cap experiments */
#define KVZ_CTU_VGPR_ATTR __attribute__((amdgpu_num_vgpr(KVZ_CTU_NUM_VGPR)))
#else
#define KVZ_CTU_VGPR_ATTR
#endif
#ifndef KVZ_CTU_WAVES_PER_EU
#define KVZ_CTU_WAVES_PER_EU 4  /* 8 workgroups of 128 lanes per CU = 4 wavefronts per SIMD at 128 VGPRs (LDS 16.5 KB would allow 9, but 96 VGPRs cost more than the ninth workgroup gives: profiles/experiments/r01_ab13*) */
#endif
template <bool CABAC> __global__ void __launch_bounds__(KVZ_CTU_THREADS) __attribute__((amdgpu_waves_per_eu(KVZ_CTU_WAVES_PER_EU))) intra_ctu_wave_kernel(const CtuFrames F, const CtuModel model, const Tables *tb,
                                                                        const int wave, const int y_min, const int n_diag)
#if !KVZ_CTU_KERNEL_BODIES
;
#else
{
  __shared__ CtuSharedT<CABAC> shared;
  __shared__ CtuModel m;  // scalars in LDS; its price table stays in HBM (kvz_hip_batch::d_entropy)
  if (threadIdx.x == 0) m = model;
  __syncthreads();
  CtuProgramT<CABAC> p;
  p.m = &m; p.tb = tb; p.F = F; p.s = &shared;
  p.frame = blockIdx.x / n_diag;
  const int y = y_min + (int)(blockIdx.x % n_diag);
  p.cx = (wave - 2 * y) * 64;
  p.cy = y * 64;
  p.run();
}
#endif

// ---- single-launch schedule: in-order tickets --------------------------------------------------------------------
// Work items (one CTU each) are listed in an order in which every CTU comes after the CTUs it depends on (anti-diagonal
// x + 2y ascending, all frames interleaved).  Workgroups draw tickets from an atomic counter and process the item behind
// the ticket; before touching neighbour data they wait for the `done` flags of the left and the above-right CTU.
// Deadlock-free for ANY number of resident workgroups: an item only ever waits for items with smaller tickets, and every
// smaller ticket has been drawn by a workgroup that is running (induction over the ticket order) -- no co-residency of
// the whole grid is assumed.  Hand-off follows the agent-scope release/acquire recipe of the CDNA guide (G16): producer
// drains its stores, one lane releases at agent scope and stores the flag; consumer polls relaxed, one lane acquires,
// then the workgroup barrier.  Compared with one launch per diagonal this removes the per-launch tail (a diagonal of
// n CTUs x F frames rarely is a multiple of the resident workgroup count) and 61 of 62 launches.
struct CtuSched {
  const uint32_t *items;  // [total]: frame << 16 | y << 8 | x  (CTU coordinates)
  unsigned *ticket;       // atomic ticket counter, zeroed before every launch
  unsigned *done;         // [frames * ctus_per_frame]: epoch of the last call that completed the CTU
  unsigned *error;        // set when a wait exceeds its spin bound (never in a healthy run)
  unsigned total, epoch;
  int no_wpp;             // items in raster order per picture; a row's first CTU also waits for the last CTU of the row above
  unsigned long long wait_ticks;  // bound of one wait in ticks of the 100 MHz constant clock (s_memrealtime): wall-clock, so that counter
                                  // serialisation under rocprof, time slicing or preemption cannot turn a healthy run into a timeout
};

#if KVZ_CTU_KERNEL_BODIES
#ifdef KVZ_CTU_TRACE  /* developer build (tools/chain_stress.py): where every CTU of the pass is -- F.prof holds a word per CTU: epoch << 8 | stage */
#define KVZ_TRACE(stage) do { if (threadIdx.x == 0 && __hip_atomic_load(sched.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) __hip_atomic_store(&F.prof[(long)frame * ctus + y * F.wc + x], (unsigned long long)sched.epoch << 8 | (stage), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while (0)
#else
#define KVZ_TRACE(stage) do { } while (0)
#endif
__device__ __forceinline__ bool wait_done(unsigned *flag, unsigned epoch, unsigned *error, unsigned long long wait_ticks, unsigned tag)
{
  unsigned long long t0 = 0;
  for (unsigned spins = 0;; ++spins) {
    if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch) return true;
    if ((spins & 1023u) == 1023u) {  // bounded: a lost hand-off must not hang the GPU
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // (every half millisecond of waiting: whatever this XCD's L2 holds of the flags' lines goes -- insurance, the poll is an agent-scope load)
      const unsigned long long now = __builtin_amdgcn_s_memrealtime();
      if (!t0) t0 = now;
      else if (now - t0 > wait_ticks) {
        if (atomicCAS(error, 0u, 0x80000000u | tag) == 0u) {  // the first to give up leaves what it can see: the flag as a read-modify-write at the memory side returns it, and how far the tickets are
          error[1] = atomicAdd(flag, 0u);
          error[2] = __hip_atomic_load(error - 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return false;
      }  // the first CTU that gave up: which neighbour (bits 29-30), frame << 16 | y << 8 | x of the CTU waited FOR
    }
    // back off after the first half millisecond: a wait that long is on nobody's critical path (hand-offs arrive within microseconds), and where a pass has stalled --
    // tools/chain_stress.py, profiles/README.md round 6: an eighth of the workgroups standing still in their searches for seconds -- the other seven eighths
    // should not poll the memory system two thousand times a microsecond meanwhile
    if (spins < 1024u) __builtin_amdgcn_s_sleep(16);
    else { __builtin_amdgcn_s_sleep(127); __builtin_amdgcn_s_sleep(127); __builtin_amdgcn_s_sleep(127); __builtin_amdgcn_s_sleep(127); }
  }
}

// The persistent loop of the ticket schedule; the kernels below differ in their register budget only.
template <bool CABAC, bool S32, bool RDOQ> __device__ __forceinline__ void ticket_loop(const CtuFrames &F, const CtuModel &model, const Tables *tb, const CtuSched &sched)
{
  __shared__ CtuSharedT<CABAC> shared;
  __shared__ CtuModel m;  // scalars in LDS; its price table stays in HBM (kvz_hip_batch::d_entropy)
  // The ticket is broadcast through a field of `shared` that is dead between two CTUs: with the CABAC contexts the block is exactly 20 480 B,
  // and one more word would cost the eighth workgroup per CU (160 KB of LDS).
#ifndef KVZ_CTU_PROFILE
  static_assert(sizeof(CtuSharedT<CABAC>) + sizeof(CtuModel) <= 20480, "eight workgroups per CU");
#endif
  if (threadIdx.x == 0) m = model;
  const int ctus = F.wc * F.hc;
  for (;;) {
    __syncthreads();  // previous item fully retired (and m visible on the first trip)
    if (threadIdx.x == 0) shared.best_mode = (int)atomicAdd(sched.ticket, 1u);
    __syncthreads();
    const unsigned t = (unsigned)shared.best_mode;
    if (t >= sched.total) break;
    const uint32_t item = sched.items[t];
    const int frame = item >> 16, y = (item >> 8) & 0xff, x = item & 0xff;
    KVZ_TRACE(1);  // ticket drawn
    if (threadIdx.x == 0) {
      unsigned *done = sched.done + (long)frame * ctus;
      // a hand-off that timed out anywhere (this launch or an earlier one: the word is sticky until kvz_hip_batch_reset) poisons the pass: from then on
      // tickets are only drained -- no search on stale neighbour data, no further 30-second waits -- and every CTU still publishes its flag
      bool ok = __hip_atomic_load(sched.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0;
      if (ok && x > 0) ok = wait_done(&done[y * F.wc + x - 1], sched.epoch, sched.error, sched.wait_ticks, ((unsigned)frame << 16 | (unsigned)y << 8 | (unsigned)(x - 1)) & 0x1fffffffu);
      if (ok && y > 0) ok = wait_done(&done[(y - 1) * F.wc + (x + 1 < F.wc ? x + 1 : x)], sched.epoch, sched.error, sched.wait_ticks, 0x20000000u | (((unsigned)frame << 16 | (unsigned)(y - 1) << 8 | (unsigned)(x + 1 < F.wc ? x + 1 : x)) & 0x1fffffffu));  // above-right implies above and above-left
      if (ok && sched.no_wpp && x == 0 && y > 0) ok = wait_done(&done[(y - 1) * F.wc + F.wc - 1], sched.epoch, sched.error, sched.wait_ticks, 0x40000000u | (((unsigned)frame << 16 | (unsigned)(y - 1) << 8 | (unsigned)(F.wc - 1)) & 0x1fffffffu));  // its contexts come from there
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      shared.best_mode = ok ? 1 : 0;
    }
    __syncthreads();
    KVZ_TRACE(2);  // neighbours there
    const bool run_it = shared.best_mode != 0;  // uniform.  (run() first writes the field behind several barriers of its own: no lane can still be reading it here)
    if (run_it) {
      CtuProgramT<CABAC, S32, RDOQ> p;
      p.m = &m; p.tb = tb; p.F = F; p.s = &shared;
      if constexpr (RDOQ) { __shared__ RdoqLds rdoq_lds; p.rl = &rdoq_lds; }
      p.frame = frame; p.cx = x * 64; p.cy = y * 64;
      p.lane_rot = (t * 64) & (KVZ_CTU_THREADS - 1);
      p.run();
    }
    KVZ_TRACE(3);  // searched
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every wave drains its stores (reconstruction, CU info, coefficients)
    __syncthreads();
    KVZ_TRACE(4);  // stores drained, both wavefronts there
    if (threadIdx.x == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      KVZ_TRACE(5);  // released
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_store(&sched.done[(long)frame * ctus + y * F.wc + x], sched.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      KVZ_TRACE(6);  // flag stored
    }
  }
}

#endif  // KVZ_CTU_KERNEL_BODIES

template <bool CABAC, bool S32 = false, bool RDOQ = false> __global__ void __launch_bounds__(KVZ_CTU_THREADS) __attribute__((amdgpu_waves_per_eu(KVZ_CTU_WAVES_PER_EU))) KVZ_CTU_VGPR_ATTR intra_ctu_ticket_kernel(const CtuFrames F, const CtuModel model, const Tables *tb,
                                                                        const CtuSched sched)
#if !KVZ_CTU_KERNEL_BODIES
;
#else
{
  ticket_loop<CABAC, S32, RDOQ>(F, model, tb, sched);
}
#endif
// --rdoq / NxN partitions: its own register budget.  Round 3: kvz_rdoq runs as ONE out-of-line wavefront-cooperative routine (kvz_rdoq.hpp rdoq_block_wave), so the
// instantiation no longer needs the 168 registers the inlined one-lane routine did: 128 registers = 4 wavefronts per SIMD, and with 23.4 KB of LDS seven workgroups
// per CU.  Measured at 1080p QP 27 `medium-pu13`: 3 wavefronts per SIMD 130.4 k CTUs/s, 4: 135.4 k (141.4 k with 320 pictures in flight) -- the pass is still partly
// latency-bound (profiles/r03_*).  (Round 2, one lane per block: 1 wavefront per SIMD 22.5 k, 2: 41.0 k, 3: 54.3 k, 4 with thousands of spills: 17.5 k.)
#ifndef KVZ_RDOQ_WAVES_PER_EU
#define KVZ_RDOQ_WAVES_PER_EU 4
#endif
__global__ void __launch_bounds__(KVZ_CTU_THREADS) __attribute__((amdgpu_waves_per_eu(KVZ_RDOQ_WAVES_PER_EU, KVZ_RDOQ_WAVES_PER_EU))) intra_ctu_ticket_kernel_rdoq(const CtuFrames F, const CtuModel model, const Tables *tb, const CtuSched sched)
#if !KVZ_CTU_KERNEL_BODIES || (defined(KVZ_CTU_KERNEL_TU) && KVZ_CTU_KERNEL_TU != 5)
;
#else
{
  ticket_loop<true, true, true>(F, model, tb, sched);
}
#endif

}  // namespace kvz

