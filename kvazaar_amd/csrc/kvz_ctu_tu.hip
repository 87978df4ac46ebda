// kvz_ctu_tu.hip -- one CTU kernel instantiation per translation unit (see kvz_ctu_kernels.hpp): compiled six times by kvazaar_amd/build.py with
// -DKVZ_CTU_KERNEL_TU=0..5, in parallel with kvz_hip.hip.
#include <hip/hip_runtime.h>

#ifndef KVZ_CTU_KERNEL_TU
#error "compile with -DKVZ_CTU_KERNEL_TU=<0..5>"
#endif
#include "kvz_ctu_kernels.hpp"

namespace kvz {
#define KVZ_TICKET_ARGS const CtuFrames, const CtuModel, const Tables *, const CtuSched
#if KVZ_CTU_KERNEL_TU == 0    // no CABAC coefficient model (QP < 28 of `ultrafast`), both schedules
template __global__ void intra_ctu_ticket_kernel<false, false, false>(KVZ_TICKET_ARGS);
template __global__ void intra_ctu_wave_kernel<false>(const CtuFrames, const CtuModel, const Tables *, const int, const int, const int);
#elif KVZ_CTU_KERNEL_TU == 1  // CABAC coefficient model
template __global__ void intra_ctu_ticket_kernel<true, false, false>(KVZ_TICKET_ARGS);
#elif KVZ_CTU_KERNEL_TU == 2  // the older schedule with the CABAC coefficient model
template __global__ void intra_ctu_wave_kernel<true>(const CtuFrames, const CtuModel, const Tables *, const int, const int, const int);
#elif KVZ_CTU_KERNEL_TU == 3  // 32x32 CUs searched
template __global__ void intra_ctu_ticket_kernel<false, true, false>(KVZ_TICKET_ARGS);
#elif KVZ_CTU_KERNEL_TU == 4
template __global__ void intra_ctu_ticket_kernel<true, true, false>(KVZ_TICKET_ARGS);
#elif KVZ_CTU_KERNEL_TU == 5  // RDOQ / NxN: intra_ctu_ticket_kernel_rdoq's body (kvz_ctu_kernels.hpp)
#else
#error "KVZ_CTU_KERNEL_TU out of range"
#endif
}  // namespace kvz
