// valu_issue_bench.hip -- measures what one wave64 VALU instruction costs a gfx950 SIMD, the constant behind the
// `valu_issue` bound of the CTU kernel (bench.py roofline.limiter).
//
// For each instruction: every wave executes ITERS x 64 copies of it over 8 independent destination registers (so the
// dependent-issue latency does not limit) between two s_memtime reads (tick = shader cycle, MI355X_MICROARCH.md); the grid
// puts 1, 2, 4 or 8 waves on every SIMD (256-lane workgroups = one wave per SIMD of a CU, k workgroups per CU).  Reported per
// configuration: median over waves of   cycles / (instructions per wave x waves per SIMD)   = SIMD cycles per wave64
// instruction when the SIMD is saturated, plus the single-wave figure (issue interval of one wave).  A `chain` variant
// (one destination register) gives the dependent latency.  `lanes32` / `lanes16` run the same stream with only the low 32 / 16
// lanes enabled (EXEC mask), to see whether a partially filled wavefront is cheaper.
//
// Build: hipcc --offload-arch=gfx950 -O3 -o kvazaar_amd/lib/valu_issue_bench tools/valu_issue_bench.hip   (done by __graft_entry__.build)
// Run on the GPU box:  kvazaar_amd/lib/valu_issue_bench > gpurun_out/valu_issue.jsonl
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#define CHECK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(r_)); exit(1); } } while (0)

enum { OP_ADD_U32, OP_PK_ADD_I16, OP_MAD_U32_U24, OP_MUL_LO_U32, OP_FMA_F32, OP_FMA_F64, OP_SAD_U8, OP_LSHLREV, OP_CNDMASK, OP_DPP_MOV, OP_PERM, OP_PK_MAX_I16,
       OP_ALIGNBYTE, OP_MAD_I32_I24, OP_ADD3, OP_COUNT };
static const char *kNames[OP_COUNT] = { "v_add_u32", "v_pk_add_i16", "v_mad_u32_u24", "v_mul_lo_u32", "v_fma_f32", "v_fma_f64", "v_sad_u8", "v_lshlrev_b32",
                                        "v_cndmask_b32", "v_mov_b32_dpp(quad_perm)", "v_perm_b32", "v_pk_max_i16", "v_alignbyte_b32", "v_mad_i32_i24", "v_add3_u32" };

// one asm statement per loop trip (64 instructions): separate asm statements make the compiler put an s_nop between them
#define R8(T) T(0) T(1) T(2) T(3) T(4) T(5) T(6) T(7)
#define T64(T) R8(T) R8(T) R8(T) R8(T) R8(T) R8(T) R8(T) R8(T)
#define STREAM(T, TC)                                                                                                                       \
  if (CHAIN) asm volatile(T64(TC) : "+v"(r[0]) : "v"(s) : "vcc");                                                                            \
  else asm volatile(T64(T) : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(s) : "vcc");
#define STREAM64(T, TC)                                                                                                                     \
  if (CHAIN) asm volatile(T64(TC) : "+v"(d[0]));                                                                                             \
  else asm volatile(T64(T) : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7]));
// independent streams write register %i (i = 0..7) and read the shared source %8; chains only use %0 and the source %1
#define A_ADD(i) "v_add_u32 %" #i ", %" #i ", %8\n"
#define C_ADD(i) "v_add_u32 %0, %0, %1\n"
#define A_PK(i) "v_pk_add_i16 %" #i ", %" #i ", %8\n"
#define C_PK(i) "v_pk_add_i16 %0, %0, %1\n"
#define A_MAD(i) "v_mad_u32_u24 %" #i ", %" #i ", %8, %" #i "\n"
#define C_MAD(i) "v_mad_u32_u24 %0, %0, %1, %0\n"
#define A_MADI(i) "v_mad_i32_i24 %" #i ", %" #i ", %8, %" #i "\n"
#define C_MADI(i) "v_mad_i32_i24 %0, %0, %1, %0\n"
#define A_MUL(i) "v_mul_lo_u32 %" #i ", %" #i ", %8\n"
#define C_MUL(i) "v_mul_lo_u32 %0, %0, %1\n"
#define A_FMA(i) "v_fma_f32 %" #i ", %" #i ", %8, %" #i "\n"
#define C_FMA(i) "v_fma_f32 %0, %0, %1, %0\n"
#define A_FMA64(i) "v_fma_f64 %" #i ", %" #i ", %" #i ", %" #i "\n"
#define C_FMA64(i) "v_fma_f64 %0, %0, %0, %0\n"
#define A_SAD(i) "v_sad_u8 %" #i ", %" #i ", %8, %" #i "\n"
#define C_SAD(i) "v_sad_u8 %0, %0, %1, %0\n"
#define A_SHL(i) "v_lshlrev_b32 %" #i ", 1, %" #i "\n"
#define C_SHL(i) "v_lshlrev_b32 %0, 1, %0\n"
#define A_CND(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define C_CND(i) "v_cndmask_b32 %0, %0, %1, vcc\n"
#define A_DPP(i) "v_mov_b32_dpp %" #i ", %" #i " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define C_DPP(i) "v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define A_PERM(i) "v_perm_b32 %" #i ", %" #i ", %8, %8\n"
#define C_PERM(i) "v_perm_b32 %0, %0, %1, %1\n"
#define A_PKMAX(i) "v_pk_max_i16 %" #i ", %" #i ", %8\n"
#define C_PKMAX(i) "v_pk_max_i16 %0, %0, %1\n"
#define A_ALIGN(i) "v_alignbyte_b32 %" #i ", %" #i ", %8, 1\n"
#define C_ALIGN(i) "v_alignbyte_b32 %0, %0, %1, 1\n"
#define A_ADD3(i) "v_add3_u32 %" #i ", %" #i ", %8, %8\n"
#define C_ADD3(i) "v_add3_u32 %0, %0, %1, %1\n"

template <int OP, bool CHAIN> __device__ __forceinline__ void body(unsigned (&r)[8], double (&d)[8], unsigned s, int iters)
{
  for (int it = 0; it < iters; it++) {
    if (OP == OP_ADD_U32) { STREAM(A_ADD, C_ADD) }
    else if (OP == OP_PK_ADD_I16) { STREAM(A_PK, C_PK) }
    else if (OP == OP_MAD_U32_U24) { STREAM(A_MAD, C_MAD) }
    else if (OP == OP_MUL_LO_U32) { STREAM(A_MUL, C_MUL) }
    else if (OP == OP_FMA_F32) { STREAM(A_FMA, C_FMA) }
    else if (OP == OP_FMA_F64) { STREAM64(A_FMA64, C_FMA64) }
    else if (OP == OP_SAD_U8) { STREAM(A_SAD, C_SAD) }
    else if (OP == OP_LSHLREV) { STREAM(A_SHL, C_SHL) }
    else if (OP == OP_CNDMASK) { STREAM(A_CND, C_CND) }
    else if (OP == OP_DPP_MOV) { STREAM(A_DPP, C_DPP) }
    else if (OP == OP_PERM) { STREAM(A_PERM, C_PERM) }
    else if (OP == OP_PK_MAX_I16) { STREAM(A_PKMAX, C_PKMAX) }
    else if (OP == OP_ALIGNBYTE) { STREAM(A_ALIGN, C_ALIGN) }
    else if (OP == OP_MAD_I32_I24) { STREAM(A_MADI, C_MADI) }
    else if (OP == OP_ADD3) { STREAM(A_ADD3, C_ADD3) }
  }
}

template <int OP, bool CHAIN> __global__ void __launch_bounds__(256) bench_kernel(unsigned long long *cycles, unsigned *sink, int iters, int active_lanes)
{
  unsigned r[8];
  double d[8];
  for (int i = 0; i < 8; i++) { r[i] = threadIdx.x * 3u + i; d[i] = 1.0 + 1e-9 * (threadIdx.x + i); }
  const unsigned s = threadIdx.x | 1u;
  unsigned long long t0 = 0, t1 = 0;
  __syncthreads();
  if ((int)(threadIdx.x & 63) < active_lanes) {  // EXEC mask: the whole stream runs with `active_lanes` lanes enabled
    t0 = __builtin_amdgcn_s_memtime();
    body<OP, CHAIN>(r, d, s, iters);
    asm volatile("s_nop 0" ::: "memory");
    t1 = __builtin_amdgcn_s_memtime();
  }
  unsigned acc = 0;
  for (int i = 0; i < 8; i++) acc += r[i] + (unsigned)d[i];
  if (acc == 0x12345678u) sink[0] = acc;
  if ((threadIdx.x & 63) == 0) cycles[(size_t)blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int OP> static void run_op(unsigned long long *d_cycles, unsigned *d_sink, int cus)
{
  const int iters = 2000, insts = iters * 64;
  auto measure = [&](bool chain, int wg_per_cu, int lanes) {
    const int blocks = cus * wg_per_cu;
    for (int rep = 0; rep < 2; rep++) {  // first repetition warms the instruction cache and the clocks
      if (chain) hipLaunchKernelGGL((bench_kernel<OP, true>), dim3(blocks), dim3(256), 0, 0, d_cycles, d_sink, iters, lanes);
      else hipLaunchKernelGGL((bench_kernel<OP, false>), dim3(blocks), dim3(256), 0, 0, d_cycles, d_sink, iters, lanes);
      CHECK(hipDeviceSynchronize());
    }
    std::vector<unsigned long long> h((size_t)blocks * 4);
    CHECK(hipMemcpy(h.data(), d_cycles, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    return (double)h[h.size() / 2];
  };
  const double one = measure(false, 1, 64), chain = measure(true, 1, 64);
  printf("{\"instruction\": \"%s\", \"insts_per_wave\": %d, \"issue_interval_one_wave_cycles\": %.3f, \"dependent_latency_cycles\": %.3f", kNames[OP], insts, one / insts,
         chain / insts);
  for (int k : { 2, 4, 8 }) {
    const double c = measure(false, k, 64);
    printf(", \"simd_cycles_per_inst_%dwaves\": %.3f", k, c / ((double)insts * k));
  }
  for (int lanes : { 32, 16 }) {
    const double c = measure(false, 4, lanes);
    printf(", \"simd_cycles_per_inst_4waves_lanes%d\": %.3f", lanes, c / ((double)insts * 4));
  }
  printf("}\n");
  fflush(stdout);
}

int main()
{
  int cus = 0;
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  cus = prop.multiProcessorCount;
  unsigned long long *d_cycles;
  unsigned *d_sink;
  CHECK(hipMalloc((void **)&d_cycles, (size_t)cus * 8 * 4 * sizeof(unsigned long long)));
  CHECK(hipMalloc((void **)&d_sink, 4));
  printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d, \"note\": \"cycles = s_memtime ticks (shader cycles); k waves per SIMD = k 256-lane workgroups per CU\"}\n", prop.gcnArchName, cus,
         prop.clockRate / 1000);
  run_op<OP_ADD_U32>(d_cycles, d_sink, cus);
  run_op<OP_PK_ADD_I16>(d_cycles, d_sink, cus);
  run_op<OP_MAD_U32_U24>(d_cycles, d_sink, cus);
  run_op<OP_MAD_I32_I24>(d_cycles, d_sink, cus);
  run_op<OP_MUL_LO_U32>(d_cycles, d_sink, cus);
  run_op<OP_FMA_F32>(d_cycles, d_sink, cus);
  run_op<OP_FMA_F64>(d_cycles, d_sink, cus);
  run_op<OP_SAD_U8>(d_cycles, d_sink, cus);
  run_op<OP_LSHLREV>(d_cycles, d_sink, cus);
  run_op<OP_CNDMASK>(d_cycles, d_sink, cus);
  run_op<OP_DPP_MOV>(d_cycles, d_sink, cus);
  run_op<OP_PERM>(d_cycles, d_sink, cus);
  run_op<OP_PK_MAX_I16>(d_cycles, d_sink, cus);
  run_op<OP_ALIGNBYTE>(d_cycles, d_sink, cus);
  run_op<OP_ADD3>(d_cycles, d_sink, cus);
  return 0;
}
