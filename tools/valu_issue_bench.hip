// valu_issue_bench.hip -- measures what one wave64 VALU instruction costs a gfx950 SIMD, the constant behind the
// `valu_issue` bound of the CTU kernel (bench.py roofline.limiter).
//
// For each instruction: every wave executes ITERS x 64 copies of it over 8 independent destination registers (so the
// dependent-issue latency does not limit) between two s_memtime reads; the grid is cus x k workgroups of 256 lanes (k = 1, 2, 4, 8:
// with an even spread, k waves on every SIMD).  Two views per configuration: the median over waves of the s_memtime ticks per
// instruction of ONE wave (how often a wave gets to issue), and -- from the kernel's wall time by HIP events, independent of what a
// tick is and of where the dispatcher put the workgroups -- SIMD-nanoseconds per wave64 instruction = wall x SIMDs / wave-instructions:
// at saturation (k = 8) that is the issue cost of the instruction; x clock = cycles.  A `chain` variant (one destination register) gives
// the dependent latency.  `lanes32` / `lanes16` run the same stream with only the low 32 / 16 lanes enabled (EXEC mask), to see whether
// a partially filled wavefront is cheaper.
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
       OP_ALIGNBYTE, OP_MAD_I32_I24, OP_ADD3, OP_MOV, OP_AND, OP_LSHRREV, OP_SUB, OP_PK_SUB_I16, OP_MED3, OP_CVT_I32_F32, OP_SDWA_ADD, OP_ADD_F64, OP_MUL_F64,
       OP_CNDMASK_SGPR, OP_READLANE, OP_ADD_DPP_ROW_SHR, OP_COUNT };
static const char *kNames[OP_COUNT] = { "v_add_u32", "v_pk_add_i16", "v_mad_u32_u24", "v_mul_lo_u32", "v_fma_f32", "v_fma_f64", "v_sad_u8", "v_lshlrev_b32",
                                        "v_cndmask_b32(vcc)", "v_mov_b32_dpp(quad_perm)", "v_perm_b32", "v_pk_max_i16", "v_alignbyte_b32", "v_mad_i32_i24", "v_add3_u32",
                                        "v_mov_b32", "v_and_b32", "v_lshrrev_b32", "v_sub_u32", "v_pk_sub_i16", "v_med3_i32", "v_cvt_i32_f32", "v_add_u32_sdwa", "v_add_f64", "v_mul_f64",
                                        "v_cndmask_b32(sgpr pair)", "v_readlane_b32", "v_add_u32_dpp(row_shr)" };

// one asm statement per loop trip (64 instructions): separate asm statements make the compiler put an s_nop between them
#define R8(T) T(0) T(1) T(2) T(3) T(4) T(5) T(6) T(7)
#define T64(T) R8(T) R8(T) R8(T) R8(T) R8(T) R8(T) R8(T) R8(T)
#define STREAM(T, TC)                                                                                                                       \
  if (CHAIN) asm volatile(T64(TC) : "+v"(r[0]) : "v"(s) : "vcc", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47");                      \
  else asm volatile(T64(T) : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(s) : "vcc", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47");
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
#define A_MOV(i) "v_mov_b32 %" #i ", %8\n"
#define C_MOV(i) "v_mov_b32 %0, %0\n"
#define A_AND(i) "v_and_b32 %" #i ", %" #i ", %8\n"
#define C_AND(i) "v_and_b32 %0, %0, %1\n"
#define A_SHR(i) "v_lshrrev_b32 %" #i ", 1, %" #i "\n"
#define C_SHR(i) "v_lshrrev_b32 %0, 1, %0\n"
#define A_SUB(i) "v_sub_u32 %" #i ", %" #i ", %8\n"
#define C_SUB(i) "v_sub_u32 %0, %0, %1\n"
#define A_PKSUB(i) "v_pk_sub_i16 %" #i ", %" #i ", %8\n"
#define C_PKSUB(i) "v_pk_sub_i16 %0, %0, %1\n"
#define A_MED3(i) "v_med3_i32 %" #i ", %" #i ", %8, %8\n"
#define C_MED3(i) "v_med3_i32 %0, %0, %1, %1\n"
#define A_CVT(i) "v_cvt_i32_f32 %" #i ", %" #i "\n"
#define C_CVT(i) "v_cvt_i32_f32 %0, %0\n"
#define A_SDWA(i) "v_add_u32_sdwa %" #i ", %" #i ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD\n"
#define C_SDWA(i) "v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD\n"
#define A_ADD64(i) "v_add_f64 %" #i ", %" #i ", %" #i "\n"
#define C_ADD64(i) "v_add_f64 %0, %0, %0\n"
#define A_MUL64(i) "v_mul_f64 %" #i ", %" #i ", %" #i "\n"
#define C_MUL64(i) "v_mul_f64 %0, %0, %0\n"
#define A_CNDS(i) "v_cndmask_b32 %" #i ", %" #i ", %8, s[40:41]\n"
#define C_CNDS(i) "v_cndmask_b32 %0, %0, %1, s[40:41]\n"
#define A_RDL(i) "v_readlane_b32 s4" #i ", %" #i ", 3\n"
#define C_RDL(i) "v_readlane_b32 s40, %0, 3\n"
#define A_DPPADD(i) "v_add_u32_dpp %" #i ", %" #i ", %" #i " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define C_DPPADD(i) "v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n"
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
    else if (OP == OP_MOV) { STREAM(A_MOV, C_MOV) }
    else if (OP == OP_AND) { STREAM(A_AND, C_AND) }
    else if (OP == OP_LSHRREV) { STREAM(A_SHR, C_SHR) }
    else if (OP == OP_SUB) { STREAM(A_SUB, C_SUB) }
    else if (OP == OP_PK_SUB_I16) { STREAM(A_PKSUB, C_PKSUB) }
    else if (OP == OP_MED3) { STREAM(A_MED3, C_MED3) }
    else if (OP == OP_CVT_I32_F32) { STREAM(A_CVT, C_CVT) }
    else if (OP == OP_SDWA_ADD) { STREAM(A_SDWA, C_SDWA) }
    else if (OP == OP_ADD_F64) { STREAM64(A_ADD64, C_ADD64) }
    else if (OP == OP_MUL_F64) { STREAM64(A_MUL64, C_MUL64) }
    else if (OP == OP_CNDMASK_SGPR) { STREAM(A_CNDS, C_CNDS) }
    else if (OP == OP_READLANE) { STREAM(A_RDL, C_RDL) }
    else if (OP == OP_ADD_DPP_ROW_SHR) { STREAM(A_DPPADD, C_DPPADD) }
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
    asm volatile("s_mov_b64 s[40:41], 0x5555\ns_mov_b64 vcc, 0x3333" ::: "s40", "s41", "vcc");
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

struct Meas { double ticks_per_wave; double wall_us; };

template <int OP> static void run_op(unsigned long long *d_cycles, unsigned *d_sink, int cus)
{
  const int iters = 2000, insts = iters * 64;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  auto measure = [&](bool chain, int wg_per_cu, int lanes) {
    const int blocks = cus * wg_per_cu;
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {  // first repetition warms the instruction cache and the clocks
      CHECK(hipEventRecord(e0, 0));
      if (chain) hipLaunchKernelGGL((bench_kernel<OP, true>), dim3(blocks), dim3(256), 0, 0, d_cycles, d_sink, iters, lanes);
      else hipLaunchKernelGGL((bench_kernel<OP, false>), dim3(blocks), dim3(256), 0, 0, d_cycles, d_sink, iters, lanes);
      CHECK(hipEventRecord(e1, 0));
      CHECK(hipDeviceSynchronize());
      CHECK(hipEventElapsedTime(&ms, e0, e1));
    }
    std::vector<unsigned long long> h((size_t)blocks * 4);
    CHECK(hipMemcpy(h.data(), d_cycles, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    return Meas{ (double)h[h.size() / 2], ms * 1e3 };
  };
  // wall-clock view (independent of what an s_memtime tick is): wave-instructions per microsecond per SIMD when every SIMD holds k waves;
  // the kernel's wall time includes launch + drain (~10 us of ~250+), so the rate is a slight under-estimate
  const double simds = cus * 4.0;
  const Meas one = measure(false, 1, 64), chain = measure(true, 1, 64);
  printf("{\"instruction\": \"%s\", \"insts_per_wave\": %d, \"one_wave_ticks_per_inst\": %.3f, \"one_wave_wall_ns_per_inst\": %.3f, \"dependent_chain_ticks_per_inst\": %.3f, \"tick_mhz\": %.1f",
         kNames[OP], insts, one.ticks_per_wave / insts, one.wall_us * 1e3 / insts, chain.ticks_per_wave / insts, one.ticks_per_wave / one.wall_us);
  for (int k : { 2, 4, 8 }) {
    const Meas m = measure(false, k, 64);
    printf(", \"k%d\": {\"wave_ticks_per_inst\": %.3f, \"wall_us\": %.1f, \"simd_ns_per_wave_inst\": %.4f}", k, m.ticks_per_wave / insts, m.wall_us,
           m.wall_us * 1e3 * simds / ((double)insts * cus * k * 4));
  }
  for (int lanes : { 32, 16 }) {
    const Meas m = measure(false, 8, lanes);
    printf(", \"k8_lanes%d\": {\"wave_ticks_per_inst\": %.3f, \"simd_ns_per_wave_inst\": %.4f}", lanes, m.ticks_per_wave / insts, m.wall_us * 1e3 * simds / ((double)insts * cus * 8 * 4));
  }
  printf("}\n");
  fflush(stdout);
  CHECK(hipEventDestroy(e0));
  CHECK(hipEventDestroy(e1));
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
  run_op<OP_MOV>(d_cycles, d_sink, cus);
  run_op<OP_AND>(d_cycles, d_sink, cus);
  run_op<OP_LSHRREV>(d_cycles, d_sink, cus);
  run_op<OP_SUB>(d_cycles, d_sink, cus);
  run_op<OP_PK_SUB_I16>(d_cycles, d_sink, cus);
  run_op<OP_MED3>(d_cycles, d_sink, cus);
  run_op<OP_CVT_I32_F32>(d_cycles, d_sink, cus);
  run_op<OP_SDWA_ADD>(d_cycles, d_sink, cus);
  run_op<OP_ADD_F64>(d_cycles, d_sink, cus);
  run_op<OP_MUL_F64>(d_cycles, d_sink, cus);
  run_op<OP_CNDMASK_SGPR>(d_cycles, d_sink, cus);
  run_op<OP_READLANE>(d_cycles, d_sink, cus);
  run_op<OP_ADD_DPP_ROW_SHR>(d_cycles, d_sink, cus);
  return 0;
}
