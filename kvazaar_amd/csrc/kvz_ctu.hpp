// kvz_ctu.hpp -- the batched all-intra CTU pass: one workgroup (128 lanes, two wavefronts) searches and reconstructs one
// 64x64 CTU entirely out of LDS.
//
// What it computes: kvazaar's per-CTU flow for an I slice under `--preset ultrafast` (search.c:646 search_cu ->
// search_intra.c:391 search_intra_rough -> intra.c:623 kvz_intra_recon_cu -> quant-generic.c:198
// kvz_quantize_residual -> search.c:425 cu_rd_cost_tr_split_accurate), CU quadtree 64/32/16/8, priced on CABAC contexts
// that evolve as the encoder's do (kvz_hip_intra_cost_model; search-time updates with search_cu's save / restore points,
// syntax replay of the finished CTU, hand-off along the row and to the next row with or without WPP) and, from QP 28 on, with
// the residual coder in counting mode.  oracle/kvz_oracle_ctu.c is the function-by-function restatement this kernel is checked
// against bit for bit (modes, depths, coefficients, reconstruction, the double-precision RD costs); both reproduce the
// reference encoder's own reconstruction and CU tree (tests/test_encoder_parity.py).
//
// How it is laid out for CDNA4 (DESIGN.md 3.2 has the table and the measurements):
//   * LDS (exactly 20 480 B with the CABAC coefficient model, 19 792 B without: eight workgroups per CU): the source pixels of
//     the 32x32 quadrant being searched, ONE decided picture plus one candidate per depth sized to its CU (kvazaar keeps four
//     full lcu_t copies, search.c:103-122), a transform scratch that the small-CU search buffers share, the challenger levels
//     of 16x16 / 32x32 CUs, five context sets, the neighbour CTUs' borders.
//   * All 35 intra modes of a CU are scored at once (kvazaar tries 8..17 of them one pair at a time, search_intra.c:433-519):
//     32 angular ones by two lanes per (mode, 8x8 block) that predict and Hadamard-transform in registers (packed int16, one DPP
//     exchange), mode 34, planar and DC through LDS;
//     one wavefront then replays kvazaar's selection order on the cost table, which picks the same winner because every cost is
//     a pure function of (references, source).
//   * Y, U and V of a CU go through residual -> DCT -> quant -> dequant -> IDCT -> reconstruction together, one barrier per
//     stage; 16- and 32-point transforms on the matrix cores (kvz_mfma.hpp); an 8x8 CU with one lane per sample of any plane.
//   * CTUs depend on their left and above-right neighbours' border records (reconstructed pixels, CU info, contexts) in HBM;
//     one persistent launch draws CTUs from an in-order ticket list (kvz_batch.hpp).
//   * The kernel is bound by VALU instruction issue (77 % of all slots): what counts is the number of instructions per CTU.
//
// The program is a sequence of phases `KVZ_FOR_THREADS(tid) { ... } KVZ_SYNC();` with uniform control flow in
// between.  tests/hostsim compiles it with KVZ_HOSTSIM, where a phase is a loop over tid -- exact emulation as long
// as threads of one phase do not communicate, which is the discipline followed here (LDS atomics are integer adds; the few
// places that use cross-lane operations on the device have a plain form for the host next to them).
#pragma once
#include <type_traits>
#include "../../include/kvz_hip_types.h"
#include "kvz_ops.hpp"
#include "kvz_rdoq.hpp"

namespace kvz {

// Lanes per CTU.  Measured on MI355X (profiles/experiments, 1080p): two wavefronts per CTU beat four (twice the
// wavefronts that only skip through the lane-starved stages, half the registers each); with the LDS footprint just
// under 20 KB eight such workgroups fit a CU = four wavefronts per SIMD at 128 VGPRs.  Any multiple of 64 works.
#ifndef KVZ_CTU_THREADS
#define KVZ_CTU_THREADS 128
#endif
#ifdef KVZ_HOSTSIM
#define KVZ_FOR_THREADS(tid) for (int tid = 0; tid < KVZ_CTU_THREADS; ++tid)
#ifdef KVZ_HOSTSIM_COUNT_SYNCS
static unsigned long long g_kvz_syncs;
#define KVZ_SYNC() (++g_kvz_syncs)
#else
#define KVZ_SYNC()
#endif
#define KVZ_LDS_ADD(p, v) (*(p) += (v))
#else
// lane_rot (a multiple of 64, wavefront-uniform) rotates which wavefront plays "threads 0..63": the lane-starved stages of a
// small CU only occupy the first wavefront's worth of thread ids, and without the rotation that would always be the
// wavefront on SIMD 0 of every workgroup.
#define KVZ_FOR_THREADS(tid) for (int tid = (threadIdx.x + lane_rot) & (KVZ_CTU_THREADS - 1), once_ = 1; once_; once_ = 0)
#define KVZ_SYNC() __syncthreads()
#define KVZ_LDS_ADD(p, v) atomicAdd((p), (v))
#endif
// End of a phase whose results only the lanes of the SAME wavefront read next (the planes of the fused 8x8 CU: one wavefront is all luma, the other all chroma):
// a wavefront's LDS operations execute in order, so what is needed is only that the compiler keeps them in order -- no s_barrier, and neither wavefront
// waits for the other.  The host build runs a phase as a loop over the threads either way.
#ifdef KVZ_HOSTSIM
#define KVZ_WAVE_SYNC()
#else
#define KVZ_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
#endif

// Sum of `v` over the workgroup added to the LDS word *dst.  Must be reached by every lane of the wave (uniform control
// flow): wave64 DPP reduction, then one LDS atomic per wave.  Integer adds, so the result does not depend on the order.
KVZ_DEV void block_add(u32 *dst, u32 v)
{
#ifdef KVZ_HOSTSIM
  *dst += v;
#else
  // row_shr 8 / 4 / 2 / 1 within each row of 16 lanes (lanes shifted in from outside the row read 0): lane 15 of a row ends
  // up with the row's sum; the four row sums are then picked out by lane index.  No LDS traffic, no dependent shuffles.
  int x = (int)v;
  x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, true);
  x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, true);
  x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, true);
  x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, true);
  const u32 total = (u32)(__builtin_amdgcn_readlane(x, 15) + __builtin_amdgcn_readlane(x, 31) + __builtin_amdgcn_readlane(x, 47) + __builtin_amdgcn_readlane(x, 63));
  if ((threadIdx.x & 63) == 0 && total) atomicAdd(dst, total);
#endif
}

#define KVZ_MREF_STRIDE 36  // q in [-16, 17] for a 16x16 CU (odd number of dwords: modes fall in different banks)
#define KVZ_MREF_ORG 16
#define KVZ_MREF32_STRIDE 68  // the same for a 32x32 CU (S32 instantiations; lives in the idle 32-point transform scratch): q in [-32, 33]
#define KVZ_MREF32_ORG 32

// Two int16 lanes in one register (v_pk_add_i16 / v_pk_sub_i16 / v_pk_max_i16 on the device).  The 8x8 Hadamard of 9-bit
// differences stays within 15 bits + sign, so nothing here can wrap.
#ifdef KVZ_HOSTSIM
struct Pk16 { i16 lo, hi; };
KVZ_DEV Pk16 pk_make(int lo, int hi) { Pk16 r = { (i16)lo, (i16)hi }; return r; }
KVZ_DEV Pk16 pk_add(Pk16 a, Pk16 b) { return pk_make(a.lo + b.lo, a.hi + b.hi); }
KVZ_DEV Pk16 pk_sub(Pk16 a, Pk16 b) { return pk_make(a.lo - b.lo, a.hi - b.hi); }
KVZ_DEV u32 pk_absmax(Pk16 a) { return (u32)imax(iabs(a.lo), iabs(a.hi)); }
#else
typedef short Pk16 __attribute__((ext_vector_type(2)));
KVZ_DEV Pk16 pk_make(int lo, int hi) { Pk16 r; r.x = (short)lo; r.y = (short)hi; return r; }
KVZ_DEV Pk16 pk_add(Pk16 a, Pk16 b) { return a + b; }
KVZ_DEV Pk16 pk_sub(Pk16 a, Pk16 b) { return a - b; }
KVZ_DEV u32 pk_absmax(Pk16 a)
{
  const Pk16 m = __builtin_elementwise_max(a, -a);
  return (u32)imax((int)m.x, (int)m.y);
}
// acc + |lo + hi| + |lo - hi| of a packed pair whose LOW half carries the block's bias: 0x8000 added to ONE input sample of an 8x8 Hadamard transform -- the sample
// (0, 0), which enters every output with weight +1 or, after a stage that keeps a negated difference, -1, and -0x8000 == 0x8000 in 16 bits -- arrives in every output,
// and |x| of a 16-bit value then is the unsigned distance of x + 0x8000 from 0x8000: v_sad_u16 takes both halves' magnitudes and the running sum in one instruction,
// where sign flip, maximum, half extraction and addition were four.  The last butterfly stage (the one across the halves) is spelled out here instead of folded.
KVZ_DEV u32 pk_abs2_biased(Pk16 a, u32 acc)
{
  unsigned x;
  __builtin_memcpy(&x, &a, 4);
  const unsigned t = __builtin_amdgcn_alignbit(x, x, 16);  // [hi, lo]
  Pk16 sw, r;
  __builtin_memcpy(&sw, &t, 4);
  r = a * pk_make(1, -1) + sw;  // [lo + hi, lo - hi]
  unsigned rr;
  __builtin_memcpy(&rr, &r, 4);
  return __builtin_amdgcn_sad_u16(rr, 0x80008000u, acc);
}
#define KVZ_PK_BIAS pk_make(-32768, 0)
#endif

#ifndef KVZ_HOSTSIM
}  // namespace kvz
#include "kvz_mfma.hpp"  // 16- and 32-point transforms on the matrix cores (device only; the host build keeps scalar loops that compute the same integers)
namespace kvz {
#endif

// Optional in-kernel timeline (build with -DKVZ_CTU_PROFILE): lane 0 of every workgroup adds the shader-clock cycles
// spent since the previous mark to a per-category counter in HBM.  Categories are the KVZ_P_* constants.
#define KVZ_PROF_PU_AT (2 * KVZ_P_COUNT + 16)  /* F.prof: [0, 2 COUNT) stages, then 16 words of rdoq_block_wave (8 sections, 4 + 4 cycles / calls by block size), then the stages inside eval_pu */
#define KVZ_PROF_WORDS (4 * KVZ_P_COUNT + 16)
enum { KVZ_P_INIT = 0, KVZ_P_REFS, KVZ_P_PRED35, KVZ_P_SATD, KVZ_P_SELECT, KVZ_P_RPRED, KVZ_P_FDCT, KVZ_P_QUANT, KVZ_P_IDCT, KVZ_P_RECON,
       KVZ_P_COST, KVZ_P_COPY, KVZ_P_FINISH, KVZ_P_MISC, KVZ_P_COEFFBITS, KVZ_P_RDOQ, KVZ_P_COUNT };
#if defined(KVZ_CTU_PROFILE) && !defined(KVZ_HOSTSIM)
#define KVZ_PROF(cat) prof_mark(cat)
#define KVZ_PROF_SYNC(cat) do { KVZ_SYNC(); prof_mark(cat); } while (0)  /* a mark where the program has no barrier of its own */
#else
#define KVZ_PROF(cat)
#define KVZ_PROF_SYNC(cat)
#endif

// One byte of another workgroup's border record.  On the device the containing dword is read with an agent-scope
// atomic load (sc1: served from the coherent memory side, never from a stale non-coherent L2/L1 line).
KVZ_DEV u8 load_shared_byte(const u8 *p)
{
#ifdef KVZ_HOSTSIM
  return *p;
#else
  const unsigned long long a = (unsigned long long)p;
  const unsigned w = __hip_atomic_load((const unsigned *)(a & ~3ull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return (u8)(w >> (8 * (a & 3)));
#endif
}

// One per 8x8 (min CU), 4 bytes: 4 levels x 64 of them live in LDS.  tr_depth 4 marks an NxN CU (four 4x4 PUs, search.c:691): `mode` is then its first PU's,
// all four are in NxnLds::mode4, and `cbf` holds bits of its own: bit j (0..3) = PU j has luma levels, bits 5 / 10 = the 4x4 U / V block has (see nxn_attempt()).
struct CtuCu { u8 type : 1, depth : 2, tr_depth : 3; u8 mode; uint16_t cbf; };

// The CABAC contexts the all-intra search prices syntax with (cabac.h:63-100; indices: KVZ_HIP_CX_* of kvz_hip_types.h), each as
// kvazaar's uc_state = state << 1 | MPS: ten for the CU / transform-tree syntax, then the residual-coding ones, which only move
// when coefficients are priced with the CABAC model (CtuModel::coeff_cabac)
enum { KVZ_CX_SPLIT = KVZ_HIP_CX_SPLIT, KVZ_CX_PART = KVZ_HIP_CX_PART, KVZ_CX_INTRA = KVZ_HIP_CX_INTRA, KVZ_CX_CHROMA = KVZ_HIP_CX_CHROMA,
       KVZ_CX_CBF_LUMA = KVZ_HIP_CX_CBF_LUMA, KVZ_CX_CBF_CHROMA = KVZ_HIP_CX_CBF_CHROMA, KVZ_CX_SYNTAX_COUNT = KVZ_HIP_CX_SIG_CG, KVZ_CX_COUNT = KVZ_HIP_CX_COUNT };
// CABAC = false compiles a program that never prices coefficients with the CABAC model: the sets shrink to the syntax contexts and
// everything guarded by cabac_on() folds away (the kernel the QP < 28 runs launch)
template <bool CABAC> struct CtxSetT { alignas(4) u8 s[CABAC ? 148 : 12]; };

// What the CTU program reads of kvz_hip_intra_cost_model, compact (the device keeps it in LDS; the 128-entry price table stays
// behind a pointer: a copy in HBM on the device)
struct CtuModel {
  double lambda, lambda_sqrt;
  uint64_t coeff_weights;
  int qp;
  u8 adaptive, coeff_cabac, no_wpp, search_32x32, rdoq, search_nxn;  // switches (bytes: the struct lives in LDS next to a block that is sized to the last word)
  const float *entropy_fbits;  // [128]
  const u8 *ctx_init;          // [KVZ_CX_COUNT]
};
KVZ_HD void ctu_model_from(const kvz_hip_intra_cost_model *src, CtuModel *dst)
{
  dst->lambda = src->lambda; dst->lambda_sqrt = src->lambda_sqrt; dst->coeff_weights = src->coeff_weights; dst->qp = src->qp; dst->adaptive = src->adaptive != 0;
  dst->coeff_cabac = src->coeff_cabac != 0; dst->no_wpp = src->no_wpp != 0; dst->search_32x32 = src->search_32x32 != 0; dst->rdoq = src->rdoq != 0; dst->search_nxn = src->search_nxn != 0;
  dst->entropy_fbits = src->entropy_fbits;
  dst->ctx_init = src->ctx_init;
}

// Frame-level device buffers of one batch (all frames share the geometry).
struct CtuFrames {
  int W, H, wc, hc;          // luma size, CTU grid
  long frame_px;             // bytes per frame of one YUV420 image: W*H*3/2
  const u8 *src;             // [frames][Y|U|V]
  u8 *rec;                   // [frames][Y|U|V]
  i16 *coeff;                // [frames][ctu][6144]
  i16 *coeff_scratch;        // [frames][ctu][3 levels][6144]  (work-tree levels 1..3)
  u8 *cu_depth, *cu_mode;    // [frames][(H/8)*(W/8)]
  u8 *cu_part = nullptr;     // [frames][(H/8)*(W/8)]: 1 = the 8x8 CU is coded as four 4x4 PUs (model.search_nxn; may be null)
  u8 *cu_mode4 = nullptr;    // [frames][(H/4)*(W/4)]: luma mode per 4x4 (model.search_nxn; may be null); cu_mode keeps the first PU's
  double *ctu_cost;          // [frames][ctu]
  unsigned long long *prof;  // [KVZ_P_COUNT] cycle counters (KVZ_CTU_PROFILE builds only, else unused)
  // What a CTU hands to its right / lower neighbours: KVZ_BORDER_BYTES per CTU = four 128-byte lines with ONE producer each
  //   [0..127]   bottom row   Y 64 | U 32 | V 32        [128..255] right column Y 64 | U 32 | V 32
  //   [256..287] CU info: depth of the bottom 8x8 row [8], mode [8], depth of the right 8x8 column [8], mode [8]
  //   [288..435] the row's CABAC contexts after this CTU's syntax (KVZ_CX_*): what the CTU to the right starts from, and -- from
  //              the second CTU of a row -- the first CTU of the row below (WPP, encoderstate.c:763-771)
  //   [448..463] (search_nxn) luma mode of the sixteen 4x4 units of the right column: most probable modes are 4x4-granular once CUs can be NxN
  // Neighbour data is exchanged ONLY through these records: the frame-level rec / cu arrays share cache lines between CTUs
  // produced on different XCDs, and a line that is dirty in the reader's L2 cannot be invalidated by its acquire.
  u8 *border;                // [frames][ctu][KVZ_BORDER_BYTES]
};
#define KVZ_BORDER_BYTES 512

template <bool CABAC> struct CtuSharedT {
  alignas(8) u8 org[1536];   // source pixels of the 32x32 quadrant being searched: Y 32x32 | U 16x16 | V 16x16 (load_org())
  // Reconstruction.  kvazaar keeps one full lcu_t per depth (search.c:103-122); what those copies hold at any time is
  // (a) the pixels already decided, identical in every level that can see them, plus (b) one candidate per depth for the
  // CU being tried.  So: one decided picture + one candidate buffer per depth, sized to that depth's CU.
  u8 dec[6144];              // decided pixels, Y 64x64 | U 32x32 | V 32x32.  The depth-0 candidate (64x64 merge) reuses it:
                             // by then the split result has been written to the frame (run()).
#if defined(KVZ_CTU_PROFILE) && !defined(KVZ_HOSTSIM)
  unsigned long long prof_rq[16];   // rdoq_block_wave's sections and sizes (luma blocks, so one wavefront adds), flushed with prof_acc
  unsigned long long prof_acc[64];  // [category] cycles, [KVZ_P_COUNT + category] marks; [32 + ...] the same inside the 4x4 PUs of the NxN attempt (eval_pu)
  int prof_pu;
#endif
  CtuCu cu[4][64];
  u8 ref[3][2][68];          // [plane][0 top / 1 left][2w+1], w <= 32
  u8 fref[2][68];            // filtered luma refs
  // Transform scratch and everything whose lifetime fits around it, in one union:
  //  - a 32x32 transform set (depth-0 / depth-1 merges) works in place on ONE buffer of Y 1024 | U 256 | V 256 int16 -- both
  //    passes of a 16/32-point transform run chained through MFMA registers -- and its 32x32 candidate sits behind it;
  //  - smaller sets use two buffers of 384 (the 8/4-point planes ping-pong between them), and while one of those runs the
  //    buffers of the 16x16 / 8x8 search are live behind them.  The two regimes never overlap in time (tbuf()).
  union {
    struct {
      alignas(16) i16 tb_big[1536];
      u8 c1[1536];             // depth-1 candidate (32x32 merge): Y 1024 | U 256 | V 256
    };
    struct {
      alignas(16) i16 tb_small[2 * 384];
      alignas(8) u8 org_t[256];  // the CU's source block transposed (horizontal modes are predicted and scored transposed)
      u8 c2[384];              // depth-2 candidate (16x16 CU):     Y 256 | U 64 | V 64
      // (the depth-3 candidates -- the four 8x8 CUs of the current 16x16 -- are written straight into `dec`: nothing reads that region of the decided picture
      // before the 16x16 decision, which either keeps them or overwrites them with c2)
      // quantised levels of the four 8x8 CUs of the current 16x16, laid out like lv2_coeff (Y 4 x 64 | U 4 x 16 | V 4 x 16, children in z-order): like the larger
      // challengers' they only go to HBM if the split wins (commit()) -- every evaluated 8x8 CU storing its levels was three quarters of the pass's write traffic.
      // Also what the CABAC coefficient cost of an 8x8 CU reads (levels_lds()).
      alignas(8) i16 lv3_coeff[384];
      // Rough search, the 15 angular modes with a negative displacement (11..25): the main reference with its projected
      // extension (intra-generic.c:97-123), already picked from the filtered / unfiltered, top / left arrays.  Entry
      // [mode - 11][KVZ_MREF_ORG + q] is ref_main[q], q in [-w, w + 1] -- all such a mode can touch.  The other modes read
      // ref / fref directly.
      u8 mref[15][KVZ_MREF_STRIDE];
      u32 satd_raw[35][4];     // sum |Hadamard| per (mode, 8x8 block) before the per-block rounding
    };
  };
  alignas(8) i16 lv2_coeff[384];        // quantised levels of the 16x16 CU being tried (Y 256 | U 64 | V 64): they only go to HBM if it wins
  alignas(8) i16 lv1_coeff[1536];       // ... and of the 32x32 merge being tried (Y 1024 | U 256 | V 256)
  u32 acc[16];               // [0..2] ssd per plane, [3..5] coeff weight sums, [6..8] non-zero counts
  int8_t preds[3];
  int best_mode;
  // uniform scalars carried between phases (written by lane 0, read by everybody after the barrier)
  double cost[4], split_cost[4];  // per depth: the CU as one unit / split in four
  double res[4];                  // per depth: the cheaper of the two, what the parent adds up
  double child_bits[4];           // 64x64 attempt: CABAC-model bits of each 32x32 unit's coefficients (priced right after its reconstruction)
  u32 child_acc[4][9];
  int cbf_any;
  // neighbour CTUs' data, staged once per CTU: reconstructed border pixels and CU info of the left column / top row
  u8 bpx_left[3][66];        // x = ox-1, y = oy-1 .. oy+63   (index 0 = corner)
  u8 bpx_top[3][98];         // y = oy-1, x = ox-1 .. ox+95   (index 0 = corner)
  u8 nb_depth[2][8], nb_mode[2][8];  // [0 left / 1 top][8x8 index]
#ifdef KVZ_HOSTSIM
  i16 dct32[32 * 32];        // HEVC core transform matrix; the 16/8/4-point matrices are its rows 2k/4k/8k (dct-generic.c:46-120)
#else
  i16 dct_small[64 + 16];    // the 8- and 4-point matrices: larger transforms run on the matrix cores from Tables::dct_i8
#endif
  u8 dcval[3];               // DC value of the current references per plane
  int8_t mode_disp[35];      // angular parameters per mode (intra-generic.c:59-60, 70-76): signed sample displacement,
  int16_t mode_inv[35];      //   inverse angle, and whether the mode projects on the top reference
  QuantScalars qs[4][2];     // [log2w - 2][0 luma / 1 chroma] for this QP (quant-generic.c:57-66, 303-339)
  double mode_bits_cost[3];  // lambda_sqrt * kvz_luma_mode_bits for: not an MPM, MPM 0, MPM 1/2
  // CABAC contexts (lane 0 only).  cab = state->search_cabac while the CTU is searched; pre[d] = its value when search_cu entered
  // depth d (search.c:655; pre[0] is the row's state->cabac the search started from, search.c:1211, and what the CTU's real syntax
  // is replayed on afterwards); post2 = after the 16x16 CU was evaluated (search.c:956)
  CtxSetT<CABAC> cab, pre[3], post2;
  float entropy_fbits[128];  // the model's price table and the LPS transitions of Tables::ctx_next, staged per CTU: every lookup sits on lane 0's critical path
  u8 ctx_lps[64];            // packed state after a less probable symbol, for MPS = 0 (xor the MPS bit in; state 0 flips it: entry 0 is 1)
};

using CtuShared = CtuSharedT<true>;
static_assert(__builtin_offsetof(CtuSharedT<true>, fref) == __builtin_offsetof(CtuSharedT<true>, ref) + 408 && __builtin_offsetof(CtuSharedT<false>, fref) == __builtin_offsetof(CtuSharedT<false>, ref) + 408,
              "Tables::mref_tab addresses the filtered references 408 bytes behind the unfiltered ones");
static_assert(KVZ_MREF_STRIDE == kMrefStride && KVZ_MREF_ORG == kMrefOrg && sizeof(((CtuSharedT<true> *)0)->ref[0][0]) == kMrefRefRow && kMrefFiltered == 408,
              "Tables::mref_tab (kvz_tables.hpp) is built for this layout of the reference arrays");
static_assert(KVZ_CTU_THREADS >= 128, "rough_search requests four table entries per lane ahead of build_mref (mref_pre[4]): 510 entries need at least 128 lanes");

// Offset of plane c (0 Y, 1 U, 2 V) in a CTU's 6144-entry block: 0, 4096, 5120.  Arithmetic, not a table: indexed by a per-lane plane a constant array is a
// load from global memory on the critical path of the phase.
KVZ_HD int plane_off(int c) { return c ? 3072 + (c << 10) : 0; }

// What the RDOQ instantiation keeps in LDS on top of CtuSharedT: the price of both bins of every context at the row coder's states, fixed for the CTU.
// (kvz_rdoq's per-position cost arrays were tried here too, for blocks up to 16x16: no gain -- the routine is bound by its own serial instruction stream --
// and 12 KB less LDS means more CTUs in flight.)
struct RdoqLds {
  i32 ptab[2 * 148];
  u8 diag8[64];                   // Tables::diag8 (the group order of a 32x32 block) next to the lanes that index it per position
  // ---- the NxN partition of 8x8 CUs (kvz_hip_intra_cost_model::search_nxn, kvazaar's --pu-depth-intra ..-4): depth 4 of search_cu's recursion (search.c:691, 794,
  // 970-974).  kvazaar gives it a fifth lcu_t; here it is what that level can differ in: the candidate of the one 8x8 CU being tried, its levels, four modes and
  // coded-block flags.  CU info becomes 4x4-granular in one respect only -- the luma mode, which the most probable modes of a neighbour look at.
  u8 mode4[4][64][4];             // [work-tree level][8x8 cell][PU]: luma mode per 4x4 unit (four equal entries for a 2Nx2N CU)
  u8 nb_mode4_left[16];           // ... of the left CTU's right column (the row above is never asked: intra.c:107 takes DC across an LCU row boundary)
  u8 c4[96];                      // candidate pixels of the NxN attempt: Y 8x8 | U 4x4 | V 4x4
  alignas(8) i16 lv4_coeff[96];   // its levels: the four luma blocks in z-order (16 each), U, V
  u8 pu_mode[4], pu_cbf[4];       // per PU: mode, luma coded-block flag
  int pu_cbf_c[2];                // the CU's 4x4 U / V blocks (done with the first PU, transform.c:306-312)
  int a3x, a3y, n_pu;             // CTU-local origin of the 8x8 CU being tried as NxN; PUs evaluated so far (-1: no attempt running)
  double split_cost3;
  CtxSetT<true> pre3, post3;      // search contexts when search_cu entered the 8x8 CU / after its 2Nx2N evaluation (search.c:655, 956-959)
};
// RDOQ: the instantiation that quantises with kvz_rdoq (kvz_hip_intra_cost_model::rdoq, preset `medium`) and / or tries NxN partitions (search_nxn); the
// others carry none of that code
// S32: the instantiation that can also SEARCH 32x32 CUs (kvz_hip_intra_cost_model::search_32x32, --pu-depth-intra 1-3); the others carry none of its code
template <bool CABAC, bool S32 = false, bool RDOQ = false> struct CtuProgramT {
  using CtxSet = CtxSetT<CABAC>;
  const CtuModel *m;
  const Tables *tb;
  CtuFrames F;
  CtuSharedT<CABAC> *s;
  RdoqLds *rl = nullptr;  // RDOQ instantiation only
  static constexpr bool NXN = RDOQ;
  KVZ_DEV bool nxn_on() const { return NXN && m->search_nxn; }
  KVZ_DEV bool cabac_on() const { return CABAC && m->coeff_cabac; }  // coefficients priced with the CABAC model (rdo.c:311-340)
  int frame, cx, cy;  // CTU origin (luma px)
  int a1x, a1y, a2x, a2y;  // CTU-local luma origin of the depth-1 / depth-2 CU whose candidates are live (uniform)
  // The thread that runs a CU's scalar bookkeeping inside the first phase of its reference build (cu_header, price_modes: a serial chain of LDS lookups and
  // double-precision products): one of the OTHER wavefront than the one whose lanes fetch the reference samples of an 8x8 CU, so the two run side by side
  static constexpr int kHookThread = KVZ_CTU_THREADS > 64 ? 64 : 0;
  int a3q = 0;             // which 8x8 child of the depth-2 CU is being evaluated (z-order; uniform): its slot in CtuShared::lv3_coeff
  int lane_rot = 0;        // see KVZ_FOR_THREADS
#if defined(KVZ_CTU_PROFILE) && !defined(KVZ_HOSTSIM)
  unsigned long long t_last;
  __device__ void prof_mark(int cat)
  {
    if (threadIdx.x == 0) {
      const unsigned long long t = __builtin_amdgcn_s_memtime();
      const int at = s->prof_pu ? 32 : 0;
      s->prof_acc[at + cat] += t - t_last;  // LDS: a global atomic per mark would dominate what is being measured
      s->prof_acc[at + KVZ_P_COUNT + cat] += 1;
      t_last = __builtin_amdgcn_s_memtime();
    }
  }
#endif

  // ---------------------------------------------------------------- small uniform helpers
  KVZ_DEV const u8 *frame_rec(int c) const { return F.rec + (long)frame * F.frame_px + (c == 0 ? 0 : c == 1 ? (long)F.W * F.H : (long)F.W * F.H * 5 / 4); }
  KVZ_DEV const u8 *frame_src(int c) const { return F.src + (long)frame * F.frame_px + (c == 0 ? 0 : c == 1 ? (long)F.W * F.H : (long)F.W * F.H * 5 / 4); }
  KVZ_DEV int ctu_index() const { return (cy >> 6) * F.wc + (cx >> 6); }
  KVZ_DEV i16 *coeff_level(int level) const
  {
    // Level 3 is the CTU's output block; every CU is a challenger whose levels are copied over only when it wins (commit()):
    // 8x8 CUs, a 16x16 CU and a 32x32 merge keep them in LDS (CtuShared::lv3_coeff / lv2_coeff / lv1_coeff), the 64x64
    // merge writes the scratch block in HBM.  One scratch block is enough: a depth's challenger is dead by the time the next shallower depth writes
    // the same z-order range.
    const long ci = (long)frame * F.wc * F.hc + ctu_index();
    return level == 3 ? F.coeff + ci * 6144 : F.coeff_scratch + ci * 6144;
  }
  // Candidate buffer of depth `lv`, addressed with plane-local CTU coordinates: pixel (x, y) of plane c is
  // buf[bias[c] + (y << lw[c]) + x].  Everything in it is wavefront-uniform.
  struct CandView {
    u8 *buf;
    int bias[3], lw[3];
    // selects, not array indexing: with a per-lane plane index the arrays would have to live in scratch memory
    KVZ_DEV u8 &at(int c, int x, int y) const { return buf[(c == 0 ? bias[0] : (c == 1 ? bias[1] : bias[2])) + (y << (c == 0 ? lw[0] : lw[1])) + x]; }
  };
  KVZ_DEV CandView cand_view(int lv) const
  {
    CandView v;
    if (NXN && lv == 4) {  // the 8x8 CU being tried as four 4x4 PUs
      v.buf = rl->c4;
      v.lw[0] = 3; v.lw[1] = v.lw[2] = 2;
      v.bias[0] = -(rl->a3y * 8 + rl->a3x);
      v.bias[1] = 64 - ((rl->a3y >> 1) * 4 + (rl->a3x >> 1));
      v.bias[2] = v.bias[1] + 16;
    } else if (lv == 0 || lv == 3) {
      v.buf = s->dec;  // see CtuShared::dec, CtuShared::c2
      v.lw[0] = 6; v.lw[1] = v.lw[2] = 5;
      v.bias[0] = 0; v.bias[1] = 4096; v.bias[2] = 5120;
    } else if (lv == 1) {
      v.buf = s->c1;
      v.lw[0] = 5; v.lw[1] = v.lw[2] = 4;
      v.bias[0] = -(a1y * 32 + a1x);
      v.bias[1] = 1024 - ((a1y >> 1) * 16 + (a1x >> 1));
      v.bias[2] = v.bias[1] + 256;
    } else {
      v.buf = s->c2;
      v.lw[0] = 4; v.lw[1] = v.lw[2] = 3;
      v.bias[0] = -(a2y * 16 + a2x);
      v.bias[1] = 256 - ((a2y >> 1) * 8 + (a2x >> 1));
      v.bias[2] = v.bias[1] + 64;
    }
    return v;
  }
  // Source pixel at plane-local CTU coordinates (inside the quadrant load_org() staged); rows are 32 >> sh apart
  KVZ_DEV const u8 *org_at(int c, int pxl, int pyl) const
  {
    const int sh = c ? 1 : 0;
    return s->org + (c == 0 ? 0 : (c == 1 ? 1024 : 1280)) + (pyl - (a1y >> sh)) * (32 >> sh) + pxl - (a1x >> sh);
  }
  KVZ_DEV static unsigned zorder(int x, int y)  // cu.h:385-421
  {
    unsigned r = 0;
    for (int b = 0; b < 4; b++) r |= (((x >> (2 + b)) & 1) << (2 * b)) | (((y >> (2 + b)) & 1) << (2 * b + 1));
    return r * 16;
  }
  KVZ_DEV static int cbf_is_set(uint16_t cbf, int depth, int plane) { return (cbf & ((0x1f >> depth) << (5 * plane))) != 0; }
  KVZ_DEV static void cbf_set(uint16_t *cbf, int depth, int plane) { *cbf |= (0x10 >> depth) << (5 * plane); }
  KVZ_DEV static void cbf_clear(uint16_t *cbf, int depth, int plane) { *cbf &= ~((0x1f >> depth) << (5 * plane)); }
  // search_nxn: the 4x4-granular luma modes of cell `cell` at level lv, all four = mode (a 2Nx2N CU)
  KVZ_DEV void set_mode4(int lv, int cell, int mode) const
  {
    if constexpr (NXN) { if (m->search_nxn) { const u32 v = (u32)mode * 0x01010101u; __builtin_memcpy(rl->mode4[lv][cell], &v, 4); } }
  }

  // CU info at luma frame position (fx, fy) as seen from work-tree level lv; false = not available
  // CU info at luma frame position (fx, fy) as seen from work-tree level lv, as a VALUE: -1 = not available, else
  // type | depth << 1 | mode << 8 (KVZ_NB_*).  (Handing out a CtuCu through a pointer put the struct on the stack: every neighbour
  // lookup of the thread-0 blocks went through scratch memory.)
#define KVZ_NB_TYPE(v) ((v) & 1)
#define KVZ_NB_DEPTH(v) (((v) >> 1) & 3)
#define KVZ_NB_MODE(v) (((v) >> 8) & 0xff)
  KVZ_DEV int neighbour_cu(int lv, int fx, int fy) const
  {
    if (fx < 0 || fy < 0 || fx >= F.W || fy >= F.H) return -1;
    const int pu = ((fy >> 2) & 1) * 2 + ((fx >> 2) & 1);  // which 4x4 unit of its 8x8 cell (search_nxn: modes are 4x4-granular)
    if (fx >= cx && fx < cx + 64 && fy >= cy && fy < cy + 64) {
      const int cell = ((fy - cy) >> 3) * 8 + ((fx - cx) >> 3);
      if (NXN && lv == 4) {
        // level 4 of the work tree equals level 3 except inside the 8x8 CU being tried as NxN: there, the PUs evaluated so far (an untouched
        // cell of kvazaar's level is CU_NOTSET)
        if (cell == (rl->a3y >> 3) * 8 + (rl->a3x >> 3)) return pu < rl->n_pu ? (1 | (3 << 1) | ((int)rl->pu_mode[pu] << 8)) : 0;
        lv = 3;
      }
      const CtuCu c = s->cu[lv][cell];
      const int mode = nxn_on() ? (int)rl->mode4[lv][cell][pu] : (int)c.mode;
      return (int)c.type | ((int)c.depth << 1) | (mode << 8);
    }
    // outside the CTU only the left column (fx == cx-1) and the top row (fy == cy-1) are ever asked for
    const int side = fx < cx ? 0 : 1, i = side == 0 ? (fy - cy) >> 3 : (fx - cx) >> 3;
    const int mode = (nxn_on() && side == 0) ? (int)rl->nb_mode4_left[(fy - cy) >> 2] : (int)s->nb_mode[side][i];
    return 1 | ((int)s->nb_depth[side][i] << 1) | (mode << 8);
  }
  // Where the reconstructed sample (px, py) of plane c (plane coordinates of the frame) lives as work-tree level lv sees it, as a byte offset into CtuShared: the
  // decided picture (which also holds the 8x8 candidates of the 16x16 CU being split) or the left / top border of the neighbour CTUs.  The choice is made on the
  // offset and ONE load follows: lanes of a reference row disagree about the place all the time, and as branches every lane would walk through every
  // alternative.  *dv: distance from a U sample to the V sample at the same position.
#ifdef KVZ_HOSTSIM
  typedef long lds_off;  // the host build's "LDS" objects are ordinary allocations, any distance apart
#else
  typedef int lds_off;
#endif
  KVZ_DEV lds_off rec_off(int lv, int c, int px, int py, int *dv = nullptr) const
  {
    const int sh = c ? 1 : 0, w = 64 >> sh, pxl = px - (cx >> sh), pyl = py - (cy >> sh);
    const u8 *base = (const u8 *)s;
    const lds_off o_dec = (lds_off)(s->dec - base) + plane_off(c) + pyl * w + pxl;
    // neighbour CTUs: left column (pxl == -1) or top row (pyl == -1), staged in LDS by init()
    const lds_off o_left = (lds_off)(&s->bpx_left[0][0] - base) + c * 66 + pyl + 1, o_top = (lds_off)(&s->bpx_top[0][0] - base) + c * 98 + pxl + 1;
    const bool inside = (unsigned)pxl < (unsigned)w && (unsigned)pyl < (unsigned)w;
    lds_off off = inside ? o_dec : (pxl < 0 ? o_left : o_top);
    if (dv) *dv = inside ? 1024 : (pxl < 0 ? 66 : 98);
    if (NXN && lv == 4) {  // a 4x4 PU also sees the PUs of its own CU that came before it
      const int qx = rl->a3x >> sh, qy = rl->a3y >> sh, qw = 8 >> sh;
      if ((unsigned)(pxl - qx) < (unsigned)qw && (unsigned)(pyl - qy) < (unsigned)qw) {
        off = (lds_off)(rl->c4 - base) + (c == 0 ? 0 : (c == 1 ? 64 : 80)) + (pyl - qy) * qw + pxl - qx;
        if (dv) *dv = 16;
      }
    }
    return off;
  }

  // intra.c:84-126 kvz_intra_get_dir_luma_predictor
  KVZ_DEV static void mpm_candidates(int y, int left /* neighbour_cu() values */, int above, int8_t preds[3])
  {
    int l = 1, a = 1;
    if (left >= 0 && KVZ_NB_TYPE(left) == 1) l = KVZ_NB_MODE(left);
    if (above >= 0 && KVZ_NB_TYPE(above) == 1 && (y & 63) != 0) a = KVZ_NB_MODE(above);
    if (l == a) {
      if (l > 1) { preds[0] = (int8_t)l; preds[1] = (int8_t)(((l + 29) % 32) + 2); preds[2] = (int8_t)(((l - 1) % 32) + 2); }
      else { preds[0] = 0; preds[1] = 1; preds[2] = 26; }
    } else {
      preds[0] = (int8_t)l; preds[1] = (int8_t)a;
      if (l && a) preds[2] = 0; else preds[2] = (l + a) < 2 ? 26 : 1;
    }
  }
  // CABAC_FBITS_UPDATE (cabac.h:133-139) on context idx of `c`: the price of `bin` (CTX_ENTROPY_FBITS, cabac.h:131), then -- if
  // `update`, and unless the model freezes the contexts at their slice-start state -- the transition kvz_cabac_encode_bin
  // applies (cabac.c:104-132).  One lane.
  KVZ_DEV double ctx_price(CtxSet *c, int idx, int bin, bool update) const
  {
    const int st = c->s[idx];
    const double bits = (double)s->entropy_fbits[st ^ bin];
    if (update && m->adaptive) c->s[idx] = (u8)ctx_next(st, bin);
    return bits;
  }
  // kvz_g_auc_next_state_mps / _lps (cabac.c:40-62) on the packed state
  KVZ_DEV int ctx_next(int st, int bin) const
  {
    const int mps = st + ((st < 124) << 1), lps = (int)s->ctx_lps[st >> 1] ^ (st & 1);  // branch-free: both are a handful of ALU ops + one LDS byte
    return bin == (st & 1) ? mps : lps;
  }
  // Copy / exchange of context sets (search.c:655, 956-959, 1051): the residual-coding part only matters (and only moves) when coefficients are priced
  // with the CABAC model.  By the lanes that play threads 0..36 (one wavefront: thread ids rotate by whole wavefronts), a word each -- one lane doing
  // it was 74 LDS operations in a row.  Callers bring every thread; what thread 0 wrote into a set just before (same wavefront, program order) is seen.
  KVZ_DEV void ctx_copy_lanes(CtxSet *dst, const CtxSet *src, int tid) const
  {
    if (tid < (CABAC ? 37 : 3)) ((unsigned *)dst->s)[tid] = ((const unsigned *)src->s)[tid];
  }
  KVZ_DEV void ctx_swap_lanes(CtxSet *a, CtxSet *b, int tid) const
  {
    if (tid < (CABAC ? 37 : 3)) {
      const unsigned ta = ((const unsigned *)a->s)[tid], tb2 = ((const unsigned *)b->s)[tid];
      ((unsigned *)a->s)[tid] = tb2; ((unsigned *)b->s)[tid] = ta;
    }
  }
  // lambda_sqrt * kvz_luma_mode_bits of the three possible outcomes at the current state of the intra-mode context; the rough
  // search prices with it without touching the context (search_intra.c:524: search_cabac.update == 0 there).  One lane.
  KVZ_DEV void price_modes() const
  {
    const int st = s->cab.s[KVZ_CX_INTRA];
    const double f0 = (double)s->entropy_fbits[st ^ 0], f1 = (double)s->entropy_fbits[st ^ 1];
    s->mode_bits_cost[0] = m->lambda_sqrt * (f0 + 5);
    s->mode_bits_cost[1] = m->lambda_sqrt * (f1 + 1);
    s->mode_bits_cost[2] = m->lambda_sqrt * (f1 + 2);
  }
  // search_intra.c:641-676 kvz_luma_mode_bits
  KVZ_DEV double luma_mode_bits(CtxSet *c, int mode, const int8_t preds[3], bool update) const
  {
    double bits = 0;
    int in = 0;
    for (int i = 0; i < 3; i++) if (mode == preds[i]) in = 1;
    bits += ctx_price(c, KVZ_CX_INTRA, in, update);
    if (in) bits += (mode == preds[0]) ? 1 : 2; else bits += 5;
    return bits;
  }
  KVZ_DEV int split_model(int lv, int x, int y, int depth) const
  {
    int model = 0, n;
    if (x > 0 && (n = neighbour_cu(lv, x - 1, y)) >= 0 && KVZ_NB_DEPTH(n) > depth) model++;
    if (y > 0 && (n = neighbour_cu(lv, x, y - 1)) >= 0 && KVZ_NB_DEPTH(n) > depth) model++;
    return model;
  }
  // `known_preds`: the CU's most probable modes when the caller already has them (rough_search derives the same three from
  // the same neighbours: for the 8-aligned CU origins x >= 4 <=> x > 0 and yl > 0 <=> (y & 63) > 0).
  // `mock`: the mock encode looks its left neighbour up at LCU-local column SUB_SCU(x - 1) (encode_coding_tree.c:516), which
  // for a CU on the LCU's left edge is column 63 of the work tree -- a cell the z-order search has not reached yet (CU_NOTSET),
  // so the left candidate falls back to DC there; calc_mode_bits (search.c:566) and the real encode use the true neighbour.
  KVZ_DEV double intra_mode_syntax_bits(CtxSet *c, bool update, int lv, int x, int y, int mode, bool mock, const int8_t *known_preds = nullptr) const
  {
    int8_t preds[3];
    const bool no_left = mock && (x & 63) == 0;
    if (known_preds && !(no_left && x > 0)) { preds[0] = known_preds[0]; preds[1] = known_preds[1]; preds[2] = known_preds[2]; }
    else {
      const int left = (x > 0 && !no_left) ? neighbour_cu(lv, x - 1, y) : -1;
      const int above = ((y & 63) > 0 && y > 0) ? neighbour_cu(lv, x, y - 1) : -1;
      mpm_candidates(y, left, above, preds);
    }
    double bits = luma_mode_bits(c, mode, preds, update);
    bits += ctx_price(c, KVZ_CX_CHROMA, 0, update);
    return bits;
  }
  // encode_coding_tree.c:948-1049 kvz_mock_encode_coding_unit, intra 2Nx2N in an I slice; updates the search contexts
  KVZ_DEV double cu_bits(int lv, int x, int y, int depth, int mode, const int8_t *known_preds = nullptr) const
  {
    double bits = 0;
    const int w = 64 >> depth;
    if (depth != 3 && !(F.W < x + w || F.H < y + w)) bits += ctx_price(&s->cab, KVZ_CX_SPLIT + split_model(lv, x, y, depth), 0, true);
    if (depth == 3) bits += ctx_price(&s->cab, KVZ_CX_PART, 1, true);
    bits += intra_mode_syntax_bits(&s->cab, true, lv, x, y, mode, true, known_preds);
    return bits;
  }
  // cost of cu_split_flag = 1 at (x, y, depth) added to split_cost (search.c:962-971); updates the search contexts
  KVZ_DEV double split_flag_cost(int lv, int x, int y, int depth) const
  {
    double sb = 0;
    sb += ctx_price(&s->cab, KVZ_CX_SPLIT + split_model(lv, x, y, depth), 1, true);
    double sc = 0.0;
    sc += sb * m->lambda;
    return sc;
  }

  // ---------------------------------------------------------------- residual coding in counting mode
  // kvz_encode_coeff_nxn (strategies/generic/encode_coding_tree-generic.c:40-283) as get_coeff_cabac_cost runs it (rdo.c:220-263):
  // the bits of one transform block's residual syntax priced on context set *c, whose states move only with `update` (the coder works
  // on a copy of the search contexts, `update` flag included).  Sign data hiding, transform skip and encryption are off in this
  // configuration.  coeff: log2w x log2w levels, row-major, in LDS; type 0 luma / 2 chroma; scan 0 diagonal, 1 horizontal, 2 vertical.
  // One lane.  HEVC scans are hierarchical -- 4x4 groups in group order, the same 16-position pattern inside each group -- so both
  // orders come from three packed constants instead of the 1024-entry tables.
  KVZ_DEV static int scan_in_group(int scan, int k)  // raster index inside the 4x4 group of the k-th position (tables.c kvz_g_sig_last_scan, 4x4 entries)
  {
    const unsigned long long pat = scan == 0 ? 0xfbe7ad369c258140ull : (scan == 1 ? 0xfedcba9876543210ull : 0xfb73ea62d951c840ull);
    return (int)((pat >> (4 * k)) & 15);
  }
  KVZ_DEV int group_of(int log2w, int scan, int i) const  // raster index of the i-th group in group order (tables.h:45-89 g_sig_last_scan_cg)
  {
    if (log2w == 2) return 0;
    if (log2w == 3) return scan == 1 ? i : ((0x3120 >> (4 * i)) & 3);
    if (log2w == 4) return scan_in_group(0, i);
    if constexpr (RDOQ) return rl->diag8[i];  // staged per CTU (run()): a load from global memory in front of every group of a 32x32 block otherwise
    else return tb->diag8[i];
  }
  KVZ_DEV double coeff_bin(CtxSet *c, bool update, int idx, int bin) const
  {
    const int st = c->s[idx];
    const double bits = (double)s->entropy_fbits[st ^ bin];
    if (update) c->s[idx] = (u8)ctx_next(st, bin);
    return bits;
  }
  KVZ_DEV static int coeff_remain_bits(int symbol, int r_param)  // cabac.c:275-301 kvz_cabac_write_coeff_remain: number of bypass bins
  {
    if (symbol < (3 << r_param)) return (symbol >> r_param) + 1 + r_param;
    int length = r_param, code = symbol - (3 << r_param);
    while (code >= (1 << length)) { code -= 1 << length; length++; }
    return 3 + length + 1 - r_param + length;
  }
#ifndef KVZ_HOSTSIM
  // The same count without the loop: the escape length L satisfies 2^L <= symbol - 2 * 2^r < 2^(L + 1)
  KVZ_DEV static int coeff_remain_bits_flat(int symbol, int r_param)
  {
    const int length = 31 - __builtin_clz((unsigned)imax(symbol - (2 << r_param), 1));
    return symbol < (3 << r_param) ? (symbol >> r_param) + 1 + r_param : 4 + 2 * length - r_param;
  }
  // Escape codes of one coded group, every level on its own lane (k = scan position = lane < 16, q = levels coded before it): the Rice
  // parameter only ever moves up, by one after an escape-coded level above 3 << parameter (encode_coding_tree.c:224-246), so the up to
  // four positions where it moves are the first level above 3, the first one above 6 after that, ... -- four ballots and a few scalar
  // steps -- and each lane's parameter is the number of those positions coded before it.  Returns this lane's bypass bins.
  KVZ_DEV static unsigned escape_bins_lane(bool mine, int k, int q, int absval)
  {
    const unsigned ge2 = (unsigned)__ballot(mine && absval >= 2) & 0xffffu;
    const int base_level = q < 8 ? ((ge2 >> (k + 1)) ? 2 : 3) : 1;
    const bool esc = mine && absval >= base_level;
    int p = 16, rice = 0;
#pragma unroll
    for (int step = 0; step < 4; step++) {
      const unsigned above = (unsigned)__ballot(esc && absval > (3 << step)) & ((1u << p) - 1);
      p = above ? 31 - __builtin_clz(above) : 0;  // p == 0: nothing is coded after position 0, so it moves no lane's parameter
      rice += k < p ? 1 : 0;
    }
    return esc ? (unsigned)coeff_remain_bits_flat(absval - base_level, rice) : 0u;
  }
#endif
  KVZ_DEV static int sig_ctx_inc(int pattern, int scan, int px, int py, int log2w, int type)  // context.c:366-399 kvz_context_get_sig_ctx_inc
  {
    if (px + py == 0) return 0;
    if (log2w == 2) return (int)((0x8877886654325410ull >> (4 * (4 * py + px))) & 15);  // ctx_ind_map
    const int offset = log2w == 3 ? (scan == 0 ? 9 : 15) : (type == 0 ? 21 : 12), xs = px & 3, ys = py & 3;
    int cnt;
    if (pattern == 0) cnt = xs + ys <= 2 ? (xs + ys == 0 ? 2 : 1) : 0;
    else if (pattern == 1) cnt = ys <= 1 ? (ys == 0 ? 2 : 1) : 0;
    else if (pattern == 2) cnt = xs <= 1 ? (xs == 0 ? 2 : 1) : 0;
    else cnt = 2;
    return ((type == 0 && ((px >> 2) + (py >> 2)) > 0) ? 3 : 0) + offset + cnt;
  }
  KVZ_DEV double coeff_cabac_bits(CtxSet *c, bool update, const i16 *coeff, int log2w, int type, int scan) const
  {
    update = update && m->adaptive;
    const int w = 1 << log2w, side = w >> 2, ngroups = side * side;
    // which groups hold a level: four levels per 8-byte LDS read
    unsigned long long sig = 0;
    for (int g = 0; g < ngroups; g++) {
      const int gy = g >> (log2w - 2), gx = g & (side - 1);
      unsigned long long any = 0;
      for (int r = 0; r < 4; r++) { unsigned long long four; __builtin_memcpy(&four, coeff + ((gy * 4 + r) << log2w) + gx * 4, 8); any |= four; }
      if (any) sig |= 1ull << g;
    }
    if (!sig) return 0;  // get_coeff_cabac_cost: no coefficient, no bits
    int last_group = ngroups - 1;
    while (!((sig >> group_of(log2w, scan, last_group)) & 1)) last_group--;
    double bits = 0;
    int c1 = 1;
    bool first = true;
    for (int i = last_group; i >= 0; i--) {
      const int g = group_of(log2w, scan, i), gy = g >> (log2w - 2), gx = g & (side - 1);
      const i16 *base = coeff + ((gy * 4) << log2w) + gx * 4;
      int abs_coeff[16], num = 0, k = 15;
      if (first) {
        // the last significant position and its coding (encode_coding_tree.c:63-115 kvz_encode_last_significant_xy)
        while (!base[((scan_in_group(scan, k) >> 2) << log2w) + (scan_in_group(scan, k) & 3)]) k--;
        const int r = scan_in_group(scan, k);
        int lx = gx * 4 + (r & 3), ly = gy * 4 + (r >> 2);
        abs_coeff[num++] = iabs(base[((r >> 2) << log2w) + (r & 3)]);
        k--;
        if (scan == 2) { const int tmp = lx; lx = ly; ly = tmp; }
        const int index = log2w - 2, ctx_offset = type ? 0 : (index * 3 + (index + 1) / 4), shift = type ? index : (index + 3) / 4;
        const int bx = (type ? KVZ_HIP_CX_LAST_X_CHROMA : KVZ_HIP_CX_LAST_X_LUMA) + ctx_offset, by = (type ? KVZ_HIP_CX_LAST_Y_CHROMA : KVZ_HIP_CX_LAST_Y_LUMA) + ctx_offset;
        const unsigned long long gidx_lo = 0x7777666655443210ull;  // g_group_idx[0..15] (encoderstate.h:397); [16..23] = 8, [24..31] = 9
        const int gxi = lx < 16 ? (int)((gidx_lo >> (4 * lx)) & 15) : (lx < 24 ? 8 : 9), gyi = ly < 16 ? (int)((gidx_lo >> (4 * ly)) & 15) : (ly < 24 ? 8 : 9);
        const int gmax = w - 1 < 16 ? (int)((gidx_lo >> (4 * (w - 1))) & 15) : 9;
        for (int q = 0; q < gxi; q++) bits += coeff_bin(c, update, bx + (q >> shift), 1);
        if (gxi < gmax) bits += coeff_bin(c, update, bx + (gxi >> shift), 0);
        for (int q = 0; q < gyi; q++) bits += coeff_bin(c, update, by + (q >> shift), 1);
        if (gyi < gmax) bits += coeff_bin(c, update, by + (gyi >> shift), 0);
        if (gxi > 3) bits += (gxi - 2) / 2;  // suffixes: bypass bins
        if (gyi > 3) bits += (gyi - 2) / 2;
      }
      const bool right = gx < side - 1 && ((sig >> (g + 1)) & 1), lower = gy < side - 1 && ((sig >> (g + side)) & 1);
      bool coded = (sig >> g) & 1;
      if (i == last_group || i == 0) coded = true;  // inferred; the DC group is then scanned like a significant one
      else bits += coeff_bin(c, update, KVZ_HIP_CX_SIG_CG + type + (right || lower), coded);  // context.c:315-327
      if (coded) {
        const int pattern = log2w == 2 ? -1 : (int)right + ((int)lower << 1);  // context.c:339-351
        for (; k >= 0; k--) {
          const int r = scan_in_group(scan, k), px = gx * 4 + (r & 3), py = gy * 4 + (r >> 2);
          const int level = base[((r >> 2) << log2w) + (r & 3)];
          if (k > 0 || i == 0 || num) bits += coeff_bin(c, update, (type == 0 ? KVZ_HIP_CX_SIG_LUMA : KVZ_HIP_CX_SIG_CHROMA) + sig_ctx_inc(pattern, scan, px, py, log2w, type), level != 0);
          if (level) abs_coeff[num++] = iabs(level);
        }
      }
      first = false;
      if (num > 0) {
        int ctx_set = (i > 0 && type == 0) ? 2 : 0;
        if (c1 == 0) ctx_set++;
        c1 = 1;
        const int base_one = (type == 0 ? KVZ_HIP_CX_ONE_LUMA : KVZ_HIP_CX_ONE_CHROMA) + 4 * ctx_set, num_c1 = num < 8 ? num : 8;
        int first_c2 = -1;
        for (int q = 0; q < num_c1; q++) {
          const int symbol = abs_coeff[q] > 1;
          bits += coeff_bin(c, update, base_one + c1, symbol);
          if (symbol) { c1 = 0; if (first_c2 == -1) first_c2 = q; }
          else if (c1 < 3 && c1 > 0) c1++;
        }
        if (c1 == 0 && first_c2 != -1) bits += coeff_bin(c, update, (type == 0 ? KVZ_HIP_CX_ABS_LUMA : KVZ_HIP_CX_ABS_CHROMA) + ctx_set, abs_coeff[first_c2] > 2);
        bits += num;  // signs
        if (c1 == 0 || num > 8) {
          int first_coeff2 = 1, go_rice = 0;
          for (int q = 0; q < num; q++) {
            const int base_level = q < 8 ? 2 + first_coeff2 : 1;
            if (abs_coeff[q] >= base_level) {
              bits += coeff_remain_bits(abs_coeff[q] - base_level, go_rice);
              if (abs_coeff[q] > 3 * (1 << go_rice)) go_rice = go_rice + 1 < 4 ? go_rice + 1 : 4;
            }
            if (abs_coeff[q] >= 2) first_coeff2 = 0;
          }
        }
      }
    }
    return bits;
  }
#ifndef KVZ_HOSTSIM
  // The same count by a whole wavefront (all 64 lanes call it, converged; every argument wavefront-uniform).  What is parallel:
  // the group masks (one lane per 4x4 group, one ballot), and inside a coded group the sixteen scan positions -- level, context
  // increment of its significance flag -- on sixteen lanes.  What stays serial is only what the standard makes serial: with
  // `update` the chain of state changes (one table lookup per bin, operands fetched from the lanes with v_readlane); without it
  // even the significance flags are priced by their lanes and summed with DPP.  Prices are accumulated in Q15 integers -- every
  // table entry is a multiple of 2^-15 -- so the sum is exact and order-free, and equals the double sum of the one-lane version.
  KVZ_DEV static int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
  // The state machine of the residual contexts in REGISTERS for the duration of one block, one context per lane and register so that a bin is
  // v_readlane / v_writelane with a scalar lane number and a handful of scalar operations -- no LDS round trip, no byte insertion on the chain
  // from one bin's state to the next.  Three registers, every class of contexts inside one of them (r = index - KVZ_HIP_CX_SIG_CG):
  //   a: lanes 0..45 = r 0..45 (coded-group flags, significance luma / chroma), 46..51 = r 130..135 (greater-2 luma / chroma),
  //      52..59 = r 122..129 (greater-1 chroma);   b: lanes 0..59 = r 46..105 (last position y / x, luma / chroma);   c: lanes 0..15 = r 106..121 (greater-1 luma)
  // plus entry L of the LPS transition table and entries L / L + 64 of the price table (as Q15 integers).
  // The residual contexts of one block's class, one context per lane, in two registers: `a` the class's own contexts -- luma: coded-group
  // flags 0..1, significance 2..28, greater-1 29..44, greater-2 45..48; chroma: 0..1, 2..16, 17..24, 25..26 -- and `b` the last-position
  // prefixes of both classes (context r = 46 + lane, r = index - KVZ_HIP_CX_SIG_CG).
  struct WaveCtx { int a, b; };
  enum { KVZ_WL_SIG = 2, KVZ_WL_ONE_LUMA = 29, KVZ_WL_ABS_LUMA = 45, KVZ_WL_ONE_CHROMA = 17, KVZ_WL_ABS_CHROMA = 25 };
  KVZ_DEV static int wave_r_class(int type, int lane)  // the context a lane of register a holds, -1: none
  {
    if (type == 0) return lane < 2 ? lane : (lane < 29 ? lane + 2 : (lane < 45 ? 106 + lane - 29 : (lane < 49 ? 130 + lane - 45 : -1)));
    return lane < 2 ? 2 + lane : (lane < 17 ? 31 + lane - 2 : (lane < 25 ? 122 + lane - 17 : (lane < 27 ? 134 + lane - 25 : -1)));
  }
  KVZ_DEV static bool ctx_is_chroma(int r)  // r = index - KVZ_HIP_CX_SIG_CG
  {
    return (r >= 2 && r < 4) || (r >= 31 && r < 46) || (r >= 61 && r < 76) || (r >= 91 && r < 106) || (r >= 122 && r < 130) || r >= 134;
  }
  KVZ_DEV WaveCtx wave_ctx_load(const CtxSet *c, int lane, int type) const
  {
    WaveCtx w;
    const u8 *r = c->s + KVZ_HIP_CX_SIG_CG;
    const int ra = wave_r_class(type, lane);
    w.a = ra >= 0 ? (int)r[ra] : 0;
    w.b = lane < 60 ? (int)r[46 + lane] : 0;
    return w;
  }
  // Only the contexts of the block's own class -- luma (type 0) or chroma -- are written back: a block never moves the other class
  // (cabac.h:63-100: every residual context exists once per class), so a luma block on one wavefront and the chroma blocks of the same
  // unit on the other can price and update the same set concurrently.
  // part (coeff_cabac_bits_wave): 1 = only the coded-group and significance contexts moved, 2 = only the others
  KVZ_DEV void wave_ctx_store(CtxSet *c, const WaveCtx &w, int lane, int type, int part) const
  {
    u8 *r = c->s + KVZ_HIP_CX_SIG_CG;
    const int ra = wave_r_class(type, lane);
    const bool sig_half = lane < (type == 0 ? KVZ_WL_ONE_LUMA : KVZ_WL_ONE_CHROMA);
    if (ra >= 0 && (part == 0 || (part == 1) == sig_half)) r[ra] = (u8)w.a;
    if (part != 1 && lane < 60 && ctx_is_chroma(46 + lane) == (type != 0)) r[46 + lane] = (u8)w.b;
  }
  KVZ_DEV static int wlane_last(int idx) { return idx - KVZ_HIP_CX_SIG_CG - 46; }
  // The count with updates off (merge attempts: every bin is priced at the entry state, search.c:1005-1041): no bin depends on another, so
  // nothing is serial.  Per coded group the sixteen lanes of the scan positions price their own significance and greater-1 flags (the
  // greater-1 context of the q-th level is 0 once an earlier level exceeded 1, else min(q + 1, 3)), the prefix bins of the last position sit
  // on lanes 0.. / 16.., and every lane keeps its own Q15 sum -- one wavefront reduction per block.  Only a group with a level above 3 among
  // its escape-coded levels (the Rice parameter then moves) falls back to the serial bypass count.
  KVZ_DEV double coeff_cabac_bits_frozen(const CtxSet *c, const i16 *coeff, int log2w, int type, int scan) const
  {
    const int lane = threadIdx.x & 63;
    const int w = 1 << log2w, side = w >> 2, ngroups = side * side;
    bool any = false;
    if (lane < ngroups) {
      const int gy = lane >> (log2w - 2), gx = lane & (side - 1);
      unsigned long long acc4 = 0;
      for (int r = 0; r < 4; r++) { unsigned long long four; __builtin_memcpy(&four, coeff + ((gy * 4 + r) << log2w) + gx * 4, 8); acc4 |= four; }
      any = acc4 != 0;
    }
    const unsigned long long sig = __ballot(any);
    if (!sig) return 0;
    const unsigned long long ord = __ballot(lane < ngroups && ((sig >> group_of(log2w, scan, lane < ngroups ? lane : 0)) & 1));
    const int last_group = 63 - __builtin_clzll(ord);
    auto price = [&](int ctx, int bin) -> unsigned { return (unsigned)(s->entropy_fbits[c->s[ctx] ^ bin] * 32768.0f); };
    unsigned acc = 0, byp = 0;        // per lane: Q15 price sum, bypass bins
    unsigned long long bypass = 0;    // wavefront-uniform bypass bins
    int prev_c1_zero = 0;
    const int k = lane & 15;
    for (int i = last_group; i >= 0; i--) {
      const int g = uni(group_of(log2w, scan, i)), gy = g >> (log2w - 2), gx = g & (side - 1);
      const i16 *base = coeff + ((gy * 4) << log2w) + gx * 4;
      const bool right = gx < side - 1 && ((sig >> (g + 1)) & 1), lower = gy < side - 1 && ((sig >> (g + side)) & 1);
      bool coded = (sig >> g) & 1;
      if (i == last_group || i == 0) coded = true;
      else if (lane == 0) acc += price(KVZ_HIP_CX_SIG_CG + type + (right || lower), coded);
      if (!coded) continue;
      const int r = scan_in_group(scan, k), px = gx * 4 + (r & 3), py = gy * 4 + (r >> 2);
      const int level = lane < 16 ? base[((r >> 2) << log2w) + (r & 3)] : 0;
      const unsigned nzmask = (unsigned)__ballot(level != 0) & 0xffffu;
      const int absval = iabs(level);
      unsigned coded_mask;
      if (i == last_group) {
        const int k_last = 31 - __builtin_clz(nzmask), rl = scan_in_group(scan, k_last);
        int lx = gx * 4 + (rl & 3), ly = gy * 4 + (rl >> 2);
        if (scan == 2) { const int tmp = lx; lx = ly; ly = tmp; }
        const int index = log2w - 2, ctx_offset = type ? 0 : (index * 3 + (index + 1) / 4), shift = type ? index : (index + 3) / 4;
        const int bx = (type ? KVZ_HIP_CX_LAST_X_CHROMA : KVZ_HIP_CX_LAST_X_LUMA) + ctx_offset, by = (type ? KVZ_HIP_CX_LAST_Y_CHROMA : KVZ_HIP_CX_LAST_Y_LUMA) + ctx_offset;
        const unsigned long long gidx_lo = 0x7777666655443210ull;
        const int gxi = lx < 16 ? (int)((gidx_lo >> (4 * lx)) & 15) : (lx < 24 ? 8 : 9), gyi = ly < 16 ? (int)((gidx_lo >> (4 * ly)) & 15) : (ly < 24 ? 8 : 9);
        const int gmax = w - 1 < 16 ? (int)((gidx_lo >> (4 * (w - 1))) & 15) : 9;
        if (lane < gxi || (lane == gxi && gxi < gmax)) acc += price(bx + (lane >> shift), lane < gxi);
        { const int l2 = lane - 16; if (l2 >= 0 && (l2 < gyi || (l2 == gyi && gyi < gmax))) acc += price(by + (l2 >> shift), l2 < gyi); }
        if (gxi > 3) bypass += (unsigned long long)((gxi - 2) / 2);
        if (gyi > 3) bypass += (unsigned long long)((gyi - 2) / 2);
        coded_mask = (1u << k_last) - 1;
      } else {
        coded_mask = 0xffffu;
        if (i != 0 && !(nzmask & 0xfffeu)) coded_mask &= ~1u;
      }
      const int pattern = log2w == 2 ? -1 : (int)right + ((int)lower << 1);
      if (lane < 16 && ((coded_mask >> k) & 1))
        acc += price((type == 0 ? KVZ_HIP_CX_SIG_LUMA : KVZ_HIP_CX_SIG_CHROMA) + sig_ctx_inc(pattern, scan, px, py, log2w, type), level != 0);
      const int num = __builtin_popcount(nzmask);
      if (num > 0) {
        const int ctx_set = ((i > 0 && type == 0) ? 2 : 0) + prev_c1_zero;
        const int base_one = (type == 0 ? KVZ_HIP_CX_ONE_LUMA : KVZ_HIP_CX_ONE_CHROMA) + 4 * ctx_set;
        const int q = __builtin_popcount(nzmask >> (k + 1));                     // levels coded before this one (higher scan positions)
        const bool mine = lane < 16 && level != 0;
        const unsigned first8 = (unsigned)__ballot(mine && q < 8) & 0xffffu;     // the levels that carry a greater-1 flag
        const unsigned gt1 = (unsigned)__ballot(mine && q < 8 && absval > 1) & 0xffffu;
        if (mine && q < 8) acc += price(base_one + ((gt1 >> (k + 1)) ? 0 : (q + 1 < 3 ? q + 1 : 3)), absval > 1);
        (void)first8;
        if (gt1) {  // the first level above 1 carries the greater-2 flag
          const int kk = 31 - __builtin_clz(gt1);
          if (lane == kk) acc += price((type == 0 ? KVZ_HIP_CX_ABS_LUMA : KVZ_HIP_CX_ABS_CHROMA) + ctx_set, absval > 2);
        }
        bypass += (unsigned long long)num;  // signs
        if (gt1 || num > 8) byp += escape_bins_lane(mine, k, q, absval);
        prev_c1_zero = gt1 ? 1 : 0;
      }
    }
    // rows of 16 lanes first (each row's sum fits 32 bits), then the four row totals
    unsigned x = acc;
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, true);
    unsigned y = byp;
    y += __builtin_amdgcn_update_dpp(0, y, 0x118, 0xF, 0xF, true);
    y += __builtin_amdgcn_update_dpp(0, y, 0x114, 0xF, 0xF, true);
    y += __builtin_amdgcn_update_dpp(0, y, 0x112, 0xF, 0xF, true);
    y += __builtin_amdgcn_update_dpp(0, y, 0x111, 0xF, 0xF, true);
    unsigned long long q15 = 0;
    for (int row = 0; row < 4; row++) {
      q15 += (unsigned)__builtin_amdgcn_readlane((int)x, 16 * row + 15);
      bypass += (unsigned)__builtin_amdgcn_readlane((int)y, 16 * row + 15);
    }
    q15 += bypass << 15;
    return (double)q15 / 32768.0;
  }
  // part: the bins of a block fall into classes with disjoint contexts, so a block can be shared by two wavefronts -- 1: the coded-group
  // and significance flags (the longest chains), 2: last position, greater-1 / greater-2 flags, signs and escape codes; 0: everything.
  // The two shares add up to the block's bits, and each share writes back only the contexts it moved.
  KVZ_DEV double coeff_cabac_bits_wave(CtxSet *c, bool update, const i16 *coeff, int log2w, int type, int scan, int part = 0) const
  {
    update = update && m->adaptive;
    if (!update) return part == 2 ? 0.0 : coeff_cabac_bits_frozen(c, coeff, log2w, type, scan);
    const int lane = threadIdx.x & 63;
    const int w = 1 << log2w, side = w >> 2, ngroups = side * side;
    bool any = false;
    if (lane < ngroups) {
      const int gy = lane >> (log2w - 2), gx = lane & (side - 1);
      unsigned long long acc = 0;
      for (int r = 0; r < 4; r++) { unsigned long long four; __builtin_memcpy(&four, coeff + ((gy * 4 + r) << log2w) + gx * 4, 8); acc |= four; }
      any = acc != 0;
    }
    const unsigned long long sig = __ballot(any);  // bit g: group g (raster) holds a level
    if (!sig) return 0;
    WaveCtx wc = wave_ctx_load(c, lane, type);
    const unsigned long long ord = __ballot(lane < ngroups && ((sig >> group_of(log2w, scan, lane < ngroups ? lane : 0)) & 1));  // the same in group order
    const int last_group = 63 - __builtin_clzll(ord);
    unsigned long long q15 = 0;
    unsigned acc_par = 0;  // per lane: Q15 prices of the bins counted one context per lane
    unsigned byp_par = 0;  // per lane: bypass bins of the escape codes
    // One bin on this lane's own context: the price goes to the lane's sum (off the state chain), the state moves on
    auto step = [&](int &st, int bin) {
      const int lps = (int)s->ctx_lps[st >> 1] ^ (st & 1);
      acc_par += (unsigned)(s->entropy_fbits[st ^ bin] * 32768.0f);
      st = bin == (st & 1) ? st + ((st < 124) << 1) : lps;
    };
    bool prev_gt1 = false;  // the previous group with levels held one above 1 (c1 == 0 at its end)
    for (int i = last_group; i >= 0; i--) {
      const int g = uni(group_of(log2w, scan, i)), gy = g >> (log2w - 2), gx = g & (side - 1);
      const i16 *base = coeff + ((gy * 4) << log2w) + gx * 4;
      const bool right = gx < side - 1 && ((sig >> (g + 1)) & 1), lower = gy < side - 1 && ((sig >> (g + side)) & 1);
      bool coded = (sig >> g) & 1;
      const bool group_flag = i != last_group && i != 0;
      if (!group_flag) coded = true;
      if (!coded) { if (part != 2 && lane == (int)(right || lower)) step(wc.a, 0); continue; }  // the flag of a group without levels
      const int k = lane & 15, r = scan_in_group(scan, k), px = gx * 4 + (r & 3), py = gy * 4 + (r >> 2);
      const int level = lane < 16 ? base[((r >> 2) << log2w) + (r & 3)] : 0;
      const unsigned nzmask = (unsigned)__ballot(level != 0) & 0xffffu;
      const int absval = iabs(level);
      unsigned coded_mask;
      if (i == last_group) {
        // the last significant position and its coding (encode_coding_tree.c:63-115): the prefix bins of one context are a run of ones
        // and at most one zero, on the context's own lane of register b; the x and the y contexts run side by side
        const int k_last = 31 - __builtin_clz(nzmask), rl = scan_in_group(scan, k_last);
        int lx = gx * 4 + (rl & 3), ly = gy * 4 + (rl >> 2);
        if (scan == 2) { const int tmp = lx; lx = ly; ly = tmp; }
        const int index = log2w - 2, ctx_offset = type ? 0 : (index * 3 + (index + 1) / 4), shift = type ? index : (index + 3) / 4;
        const int bx = (type ? KVZ_HIP_CX_LAST_X_CHROMA : KVZ_HIP_CX_LAST_X_LUMA) + ctx_offset, by = (type ? KVZ_HIP_CX_LAST_Y_CHROMA : KVZ_HIP_CX_LAST_Y_LUMA) + ctx_offset;
        const unsigned long long gidx_lo = 0x7777666655443210ull;
        const int gxi = lx < 16 ? (int)((gidx_lo >> (4 * lx)) & 15) : (lx < 24 ? 8 : 9), gyi = ly < 16 ? (int)((gidx_lo >> (4 * ly)) & 15) : (ly < 24 ? 8 : 9);
        const int gmax = w - 1 < 16 ? (int)((gidx_lo >> (4 * (w - 1))) & 15) : 9;
        const int jx = lane - wlane_last(bx), jy = lane - wlane_last(by);
        int ones = 0;
        bool zero = false;
        if (part == 1) {}
        else if (jx >= 0) { ones = imax(0, imin(gxi - (jx << shift), 1 << shift)); zero = gxi < gmax && (gxi >> shift) == jx; }
        else if (jy >= 0) { ones = imax(0, imin(gyi - (jy << shift), 1 << shift)); zero = gyi < gmax && (gyi >> shift) == jy; }
        int st = wc.b;
        while (__ballot(ones > 0 || zero)) {
          if (ones > 0) { step(st, 1); ones--; }
          else if (zero) { step(st, 0); zero = false; }
        }
        wc.b = st;
        if (part != 1 && gxi > 3) q15 += (unsigned long long)((gxi - 2) / 2) << 15;
        if (part != 1 && gyi > 3) q15 += (unsigned long long)((gyi - 2) / 2) << 15;
        coded_mask = (1u << k_last) - 1;  // the positions below it; position 0 included (a level has been seen)
      } else {
        coded_mask = 0xffffu;
        if (i != 0 && !(nzmask & 0xfffeu)) coded_mask &= ~1u;  // position 0 of a coded group with no other level is inferred
      }
      const int pattern = log2w == 2 ? -1 : (int)right + ((int)lower << 1);
      const int ctxl = KVZ_WL_SIG + sig_ctx_inc(pattern, scan, px, py, log2w, type);
      // Every context-coded bin of the group, one context per lane: lane r of the class register IS a context (WaveCtx), so the bins of
      // one context are a serial chain on their own lane and the chains of different contexts run side by side -- the trip count is the
      // longest chain of the group, not the number of bins.  `mine`: the scan positions whose bin uses this lane's context (their order
      // is the coding order, from the highest position down; bit 16 = the group's own flag), `src`: the bin values by position.
      // Classes of bins never share a context, so the order between classes is free.
      unsigned mine = 0;
      for (unsigned rem = part == 2 ? 0u : coded_mask; rem;) {  // significance flags: one pass per distinct context of the group
        const int cc = __builtin_amdgcn_readlane(ctxl, uni(__builtin_ctz(rem)));
        const unsigned same = (unsigned)__ballot(lane < 16 && ctxl == cc) & coded_mask;
        mine = lane == cc ? same : mine;
        rem &= ~same;
      }
      if (part != 2 && group_flag && lane == (int)(right || lower)) mine = 1u << 16;
      const int num = __builtin_popcount(nzmask);
      const int ctx_set = ((i > 0 && type == 0) ? 2 : 0) + (prev_gt1 ? 1 : 0);
      // greater-1 flags of the first eight levels: context 0 once an earlier level exceeded 1, else min(levels before + 1, 3)
      const int q = __builtin_popcount(nzmask >> (k + 1));
      const bool has1 = lane < 16 && level != 0 && q < 8;
      const unsigned gt1 = (unsigned)__ballot(has1 && absval > 1) & 0xffffu;
      const int lane_one = type == 0 ? KVZ_WL_ONE_LUMA : KVZ_WL_ONE_CHROMA, lane_abs = type == 0 ? KVZ_WL_ABS_LUMA : KVZ_WL_ABS_CHROMA;
      unsigned src = nzmask | 0x10000u;
      if (part != 1 && num > 0) {
        const int c1v = (gt1 >> (k + 1)) ? 0 : (q + 1 < 3 ? q + 1 : 3);
        for (int cv = 0; cv < 4; cv++) {
          const unsigned mk = (unsigned)__ballot(has1 && c1v == cv) & 0xffffu;
          mine = lane == lane_one + 4 * ctx_set + cv ? mk : mine;
        }
        if (gt1) {  // the first level above 1 carries the greater-2 flag
          const int kk = 31 - __builtin_clz(gt1);
          const unsigned gt2 = __builtin_amdgcn_readlane(absval, kk) > 2 ? 1u << kk : 0u;
          mine = lane == lane_abs + ctx_set ? 1u << kk : mine;
          src = lane >= lane_abs ? gt2 : src;
        }
        src = lane >= lane_one && lane < lane_abs ? gt1 : src;
      }
      {
        int st = wc.a;
        while (__ballot(mine != 0)) {
          if (mine) { const int kk = 31 - __builtin_clz(mine); mine &= ~(1u << kk); step(st, (src >> kk) & 1); }
        }
        wc.a = st;
      }
      if (num > 0) prev_gt1 = gt1 != 0;
      if (part != 1 && num > 0) {
        q15 += (unsigned long long)num << 15;  // signs
        if (gt1 || num > 8) byp_par += escape_bins_lane(lane < 16 && level != 0, k, q, absval);
      }
    }
    {  // rows of 16 lanes first (each row's sum fits 32 bits), then the four row totals
      unsigned x = acc_par;
      x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, true);
      x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, true);
      x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, true);
      x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, true);
      for (int row = 0; row < 4; row++) q15 += (unsigned)__builtin_amdgcn_readlane((int)x, 16 * row + 15);
      unsigned y = byp_par;  // lanes 0..15 only
      y += __builtin_amdgcn_update_dpp(0, y, 0x118, 0xF, 0xF, true);
      y += __builtin_amdgcn_update_dpp(0, y, 0x114, 0xF, 0xF, true);
      y += __builtin_amdgcn_update_dpp(0, y, 0x112, 0xF, 0xF, true);
      y += __builtin_amdgcn_update_dpp(0, y, 0x111, 0xF, 0xF, true);
      q15 += (unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)y, 15) << 15;
    }
    wave_ctx_store(c, wc, lane, type, part);
    return (double)q15 / 32768.0;
  }
#endif
  // encoderstate.c:1761-1775 kvz_get_scan_order for an intra CU (the chroma mode is the luma mode here)
  KVZ_DEV static int scan_order(int mode, int depth)
  {
    if (depth >= 3) {
      if (mode >= 6 && mode <= 14) return 2;
      if (mode >= 22 && mode <= 30) return 1;
    }
    return 0;
  }
  // Levels of the unit being evaluated / priced, in LDS: 8x8 CUs (the child a3q of the current 16x16), 16x16 CUs and 32x32 merges keep theirs there anyway
  // (lv3_coeff / lv2_coeff / lv1_coeff); with the CABAC model the 32x32 units of the 64x64 attempt stage a copy in lv1_coeff (recon_tus stage 4)
  KVZ_DEV i16 *levels_lds(int lv, int c) const
  {
    if (NXN && lv == 4) return rl->lv4_coeff + (c == 0 ? 16 * rl->n_pu : (c == 1 ? 64 : 80));  // the PU being evaluated (n_pu counts the finished ones)
    if (lv == 3) return s->lv3_coeff + (c == 0 ? 64 * a3q : (c == 1 ? 256 : 320) + 16 * a3q);
    if (lv == 2) return s->lv2_coeff + (c == 0 ? 0 : (c == 1 ? 256 : 320));
    return s->lv1_coeff + (c == 0 ? 0 : (c == 1 ? 1024 : 1280));
  }

  // ---------------------------------------------------------------- phases
  // One reference sample of intra.c:305-425 kvz_intra_build_reference_any: where it comes from, as plane coordinates of the frame; false = no neighbour at all
  // (the sample is mid-grey).  side 0 = top, 1 = left; i in [0, 2w].
  // avail_top / avail_left: Tables::avail_* of the block origin in luma samples (uniform for the call, hoisted by build_refs)
  KVZ_DEV bool ref_coords(int log2w, int sh, int lx, int ly, int side, int i, int avail_top, int avail_left, int *qx_out, int *qy_out) const
  {
    const int w = 1 << log2w, px = lx >> sh, py = ly >> sh;
    const bool corner = i == 0 && lx > 0 && ly > 0;
    if (i == 0 && !corner) { i = 1; side = 1; }  // corner = left[1]
    const int k = i - 1;
    const int al = imin(avail_left >> sh, imin(2 * w, (F.H - ly) >> sh)), at = imin(avail_top >> sh, imin(2 * w, (F.W - lx) >> sh));
    int qx, qy;
    if (side == 1) {  // left: the column to the left if there is one, else the sample above, else mid-grey
      qx = lx > 0 ? px - 1 : px;
      qy = lx > 0 ? py + imin(k, al - 1) : py - 1;
    } else {          // top: the row above if there is one, else the sample to the left
      qx = ly > 0 ? px + imin(k, at - 1) : px - 1;
      qy = ly > 0 ? py - 1 : py;
    }
    if (corner) { qx = px - 1; qy = py - 1; }
    *qx_out = qx; *qy_out = qy;
    return !(lx <= 0 && ly <= 0);
  }

  // Builds the unfiltered references of the listed planes; second phase: the [1 2 1]-filtered luma references
  // (intra.c:176-204) and the DC value of every plane (intra-generic.c:219-225).
  struct NoHook { KVZ_DEV void operator()(int) const {} };
  // `first` runs inside the first phase (every thread calls it with its id; scalars are thread 0's business): bookkeeping of the caller that no lane reads before the next barrier
  template <class First = NoHook>
  KVZ_DEV void build_refs(int lv, int x, int y, int log2w_y, int log2w_c, bool luma, bool chroma, First first = First())
  {
    // intra.c:47-82 num_ref_pixels_top / _left (Tables::avail_*) in closed form -- a unit above-right / below-left is usable iff it comes earlier in z-order:
    // above, the units up to the end of the aligned group of 2^(t+1) columns, t = trailing zeros of the row; to the left, down to the end of the aligned group of
    // 2^t rows, t = trailing zeros of the column.  Scalar arithmetic instead of two loads from global memory in front of every reference build.
    const int ur = (y & 63) >> 2, uc = (x & 63) >> 2;
    const int gt = ur ? 2 * (ur & -ur) : 0, gl = uc ? (uc & -uc) : 0;
    const int avail_top = ur ? 4 * (gt - (uc & (gt - 1))) : 64, avail_left = uc ? 4 * (gl - (ur & (gl - 1))) : 64 - 4 * ur;
    KVZ_FOR_THREADS(tid) {
      first(tid);
      // one index space over the luma samples and the chroma POSITIONS (top then left inside a plane): U and V samples sit at the same coordinates, a fixed
      // distance apart wherever they live (rec_off), so one lane fetches both -- 52 lane tasks for an 8x8 CU (one wavefront), 100 for a 16x16 one
      const int ny = luma ? 2 * (2 * (1 << log2w_y) + 1) : 0, nc = chroma ? 2 * (2 * (1 << log2w_c) + 1) : 0;
      const u8 *base = (const u8 *)s;
      for (int g = tid; g < ny + nc; g += KVZ_CTU_THREADS) {
        const int c = g < ny ? 0 : 1, i = g - (c == 0 ? 0 : ny);
        const int l2 = c ? log2w_c : log2w_y, n = 2 * (1 << l2) + 1, side = i >= n, k = side ? i - n : i;
        int qx, qy, dv;
        const bool have = ref_coords(l2, c, x, y, side, k, avail_top, avail_left, &qx, &qy);
        const lds_off off = rec_off(lv, c, qx, qy, &dv);
        s->ref[c][side][k] = have ? base[off] : (u8)128;
        if (c) s->ref[2][side][k] = have ? base[off + dv] : (u8)128;
      }
    }
    KVZ_SYNC();
    KVZ_FOR_THREADS(tid) {
      if (luma) {
        const int n = 2 * (1 << log2w_y) + 1;
        for (int i = tid; i < 2 * n; i += KVZ_CTU_THREADS) {
          const int side = i >= n, k = side ? i - n : i;
          const u8 *r = s->ref[0][side];
          u8 v;
          if (k == 0) v = (u8)((s->ref[0][1][1] + 2 * s->ref[0][1][0] + s->ref[0][0][1] + 2) / 4);
          else if (k == n - 1) v = r[k];
          else v = (u8)((r[k - 1] + 2 * r[k] + r[k + 1] + 2) / 4);
          s->fref[side][k] = v;
        }
      }
      const int c = tid - (KVZ_CTU_THREADS - 3);  // the last three lanes: one DC value each
      if (c >= 0 && ((c == 0 && luma) || (c > 0 && chroma))) {
        const int l2 = c ? log2w_c : log2w_y, w = 1 << l2;
#ifdef KVZ_HOSTSIM
        s->dcval[c] = (u8)dc_value(l2, s->ref[c][0], s->ref[c][1]);
#else
        // intra-generic.c:219-225: four samples per v_sad_u8 (|x - 0| summed over the bytes of a dword)
        u32 sum = 0;
        for (int i = 0; i < w; i += 4) {
          u32 a, b;
          __builtin_memcpy(&a, &s->ref[c][0][1 + i], 4);
          __builtin_memcpy(&b, &s->ref[c][1][1 + i], 4);
          sum = __builtin_amdgcn_sad_u8(a, 0u, sum);
          sum = __builtin_amdgcn_sad_u8(b, 0u, sum);
        }
        s->dcval[c] = (u8)((sum + w) >> (l2 + 1));
#endif
      }
    }
    KVZ_SYNC();
    KVZ_PROF(KVZ_P_REFS);
  }

  // intra.c:252-301 kvz_intra_predict for one pixel (filter_boundary on for luma)
  KVZ_DEV u8 predict_pixel(int log2w, int mode, int c, int x, int y) const
  {
    const int w = 1 << log2w;
    const u8 *top = s->ref[c][0], *left = s->ref[c][1];
    if (c == 0 && mode != 1 && w != 4) {
      bool filt;
      if (mode == 0) filt = true;
      else {
        const int thres = log2w == 3 ? 7 : (log2w == 4 ? 1 : 0);
        filt = imin(iabs(mode - 26), iabs(mode - 10)) > thres;
      }
      if (filt) { top = s->fref[0]; left = s->fref[1]; }
    }
    if (mode == 0) return planar_pixel(log2w, x, y, top, left);
    if (mode == 1) {
      const int dc = s->dcval[c];
      return (c == 0 && w < 32) ? filtered_dc_pixel(dc, x, y, top, left) : (u8)dc;
    }
    int v;
    {  // intra-generic.c:49-155 with the per-mode parameters taken from the LDS table
      const int sample_disp = s->mode_disp[mode];
      const bool vertical = mode >= 18;
      const u8 *main_ref = vertical ? top : left, *side_ref = vertical ? left : top;
      const int px = vertical ? x : y, py = vertical ? y : x;
      if (sample_disp == 0) v = main_ref[px + 1];
      else {
        const int delta_pos = (py + 1) * sample_disp, di = delta_pos >> 5, df = delta_pos & 31, inv = s->mode_inv[mode];
        // df == 0 leaves (32 r1 + 16) >> 5 = r1 (intra-generic.c:131-148 copies in that case), so no branch on the fraction; the
        // extra sample read then lies inside the reference arrays
        const int r1 = angular_ref(main_ref, side_ref, px + di, inv), r2 = angular_ref(main_ref, side_ref, px + di + 1, inv);
        v = ((32 - df) * r1 + df * r2 + 16) >> 5;
      }
    }
    if (c == 0 && w < 32) {  // intra.c:207-219 intra_post_process_angular
      if (mode == 10 && y == 0) v = iclip(0, 255, v + ((top[x + 1] - top[0]) >> 1));
      else if (mode == 26 && x == 0) v = iclip(0, 255, v + ((left[y + 1] - left[0]) >> 1));
    }
    return (u8)v;
  }

  // Partial 8x8 SATD: column `col` of the Hadamard transform of (a - b).  Row responses are the branch of the butterfly
  // tree selected by the bits of `col` (picture-generic.c:252-340 computes all of them), then the 8-point butterfly down
  // the column and the sum of magnitudes.  The eight columns of a block add up to the block's sum |coefficients|.
  KVZ_DEV static u32 satd8_column(const u8 *a, int as, const u8 *b, int bs, int col)
  {
    int t[8];
    for (int r = 0; r < 8; r++) {
      const u8 *pa = a + r * as, *pb = b + r * bs;
      int d[8];
      for (int x = 0; x < 8; x++) d[x] = (int)pa[x] - (int)pb[x];
      int u0, u1, u2, u3;
      if (col & 4) { u0 = d[0] - d[4]; u1 = d[1] - d[5]; u2 = d[2] - d[6]; u3 = d[3] - d[7]; }
      else { u0 = d[0] + d[4]; u1 = d[1] + d[5]; u2 = d[2] + d[6]; u3 = d[3] + d[7]; }
      const int v0 = (col & 2) ? u0 - u2 : u0 + u2, v1 = (col & 2) ? u1 - u3 : u1 + u3;
      t[r] = (col & 1) ? v0 - v1 : v0 + v1;
    }
    KVZ_HAD8(t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7]);
    u32 sum = 0;
    for (int r = 0; r < 8; r++) sum += (u32)iabs(t[r]);
    return sum;
  }

  // One angular mode on one 8x8 block of the CU, predicted and Hadamard-scored by a single lane without leaving
  // registers.  Equals the block's sum |H D H^T| of picture-generic.c:252-340 (before the (s + 2) >> 2 rounding) for
  // D = intra-generic.c:49-155 prediction minus source:
  //  - horizontal modes run the same code on the transposed problem (main reference = left, source block transposed);
  //    the Hadamard magnitudes of D^T are those of D;
  //  - butterfly stages commute, so the one stage that would pair the two halves of a packed register is done last and
  //    folded into the magnitude sum: |a + b| + |a - b| = 2 max(|a|, |b|).
  // PAIR: the block is shared by two neighbouring lanes, `half` 0 / 1 taking rows 0..3 / 4..7 (across the main reference): each
  // predicts and row-transforms its four rows, the lanes swap them (DPP quad_perm) for the butterfly stage that pairs row r with
  // row r + 4 -- the even lane keeps the sums, the odd one the differences --, finish their halves alone and add up.  Both lanes
  // return the block's total.  Device only (the host simulation has no cross-lane operations and runs the one-lane form).
#ifndef KVZ_HOSTSIM
  // Two u16 per register for the row generators below (v_pk_mad_u16 / v_pk_lshrrev_b16; bytes widened by v_perm_b32, whose selector value 0x0c reads as a zero byte)
  typedef unsigned short U16x2 __attribute__((ext_vector_type(2)));
  KVZ_DEV static U16x2 u16x2_of(unsigned v) { U16x2 r; __builtin_memcpy(&r, &v, 4); return r; }
  KVZ_DEV static U16x2 splat16(int v) { U16x2 r; r.x = (unsigned short)v; r.y = (unsigned short)v; return r; }
  KVZ_DEV static U16x2 widen(unsigned hi, unsigned lo, unsigned sel) { return u16x2_of(__builtin_amdgcn_perm(hi, lo, sel)); }
  KVZ_DEV static Pk16 as_pk16(U16x2 v) { Pk16 r; __builtin_memcpy(&r, &v, 4); return r; }
  // The eight bytes at p (any alignment) as four u16 pairs
  KVZ_DEV static void widen8(const u8 *p, U16x2 out[4])
  {
    unsigned w[2];
    __builtin_memcpy(w, p, 8);
    out[0] = widen(0, w[0], 0x0c010c00u); out[1] = widen(0, w[0], 0x0c030c02u); out[2] = widen(0, w[1], 0x0c010c00u); out[3] = widen(0, w[1], 0x0c030c02u);
  }
#endif
  template <bool PAIR>
  KVZ_DEV u32 angular_block_satd(int log2w, int mode, int bx, int by, int xl, int yl, int half) const
  {
    const int w = 1 << log2w;
    const bool big = S32 && log2w == 5;  // a 32x32 CU: no edge filters (intra.c:207-219)
    const bool flat = mode < 2;          // planar and DC have row generators of their own, in the vertical orientation
    const bool vertical = mode >= 18 || flat;
    const int disp = s->mode_disp[mode];
    const int p0 = vertical ? bx : by, q0 = vertical ? by : bx;  // block origin along / across the main reference
    const u8 *mr;
    if (disp < 0) mr = big ? mref32() + (mode - 11) * KVZ_MREF32_STRIDE + KVZ_MREF32_ORG : s->mref[mode - 11] + KVZ_MREF_ORG;
    else {
      const bool filt = imin(iabs(mode - 26), iabs(mode - 10)) > (log2w == 3 ? 7 : (log2w == 4 ? 1 : 0));
      mr = filt ? s->fref[vertical ? 0 : 1] : s->ref[0][vertical ? 0 : 1];
    }
    mr += p0 + 1;
    const u8 *org = vertical ? org_at(0, xl + bx, yl + by) : (big ? org_t32() : s->org_t) + bx * w + by;
    const int ostride = vertical ? 32 : w;
    const u8 *side = vertical ? s->ref[0][1] : s->ref[0][0];  // intra.c:207-219: modes 10 / 26 use the unfiltered references
    const bool edge = !big && !flat && disp == 0 && p0 == 0;
    constexpr int NR = PAIR ? 4 : 8;
    const int r0 = PAIR ? 4 * half : 0;
    Pk16 d[NR][4];
#ifndef KVZ_HOSTSIM
    if (!big) {
      // Packed rows (8x8 and 16x16 CUs): the eight predicted samples of a row as four u16 pairs, minus the source row widened the same way.  Every intermediate
      // fits 16 bits ((32 - f) a + f b + 16 <= 8176; the planar sum <= 2 w 255 + w), so the arithmetic is that of intra-generic.c sample for sample.
      if (flat) {
        if (mode == 0) {  // planar (intra-generic.c:165-201) on the filtered references: (w-1-x) L + (x+1) TR + (w-1-y) T[x] + (y+1) BL + w
          const u8 *top = s->fref[0], *left = s->fref[1];
          const int tr = top[w + 1], bl = left[w + 1];
          U16x2 tp[4], xs[4];
          widen8(top + bx + 1, tp);
#pragma unroll
          for (int j = 0; j < 4; j++) { xs[j].x = (unsigned short)(bx + 2 * j); xs[j].y = (unsigned short)(bx + 2 * j + 1); }
#pragma unroll
          for (int i = 0; i < NR; i++) {
            const int y = by + r0 + i, l = left[y + 1];
            const U16x2 dl = splat16(tr - l), wy = splat16(w - 1 - y), hb = splat16((w - 1) * l + tr + (y + 1) * bl + w);
            U16x2 o[4];
            widen8(org + (r0 + i) * ostride, o);
#pragma unroll
            for (int j = 0; j < 4; j++) d[i][j] = as_pk16(((xs[j] * dl + (tp[j] * wy + hb)) >> (unsigned short)(log2w + 1)) - o[j]);
          }
        } else {  // DC with its edge smoothing (intra-generic.c:210-241) on the unfiltered references
          const u8 *top = s->ref[0][0], *left = s->ref[0][1];
          const int dc = s->dcval[0];
          U16x2 tp[4];
          widen8(top + bx + 1, tp);
#pragma unroll
          for (int i = 0; i < NR; i++) {
            const int y = by + r0 + i;
            U16x2 o[4], v[4];
            widen8(org + (r0 + i) * ostride, o);
#pragma unroll
            for (int j = 0; j < 4; j++) v[j] = y == 0 ? (U16x2)((tp[j] + splat16(3 * dc + 2)) >> (unsigned short)2) : splat16(dc);
            if (bx == 0) v[0].x = (unsigned short)(y == 0 ? (left[1] + 2 * dc + top[1] + 2) >> 2 : (left[y + 1] + 3 * dc + 2) >> 2);
#pragma unroll
            for (int j = 0; j < 4; j++) d[i][j] = as_pk16(v[j] - o[j]);
          }
        }
      } else {
        unsigned rnd_bits = 0x00100010u;
        asm("" : "+v"(rnd_bits));  // opaque, or the compiler pulls the constant out of the multiply-add chain: mul + mad + add where two mads do
#pragma unroll
        for (int i = 0; i < NR; i++) {
          const int qa = q0 + r0 + i + 1, delta = qa * disp, di = delta >> 5, df = delta & 31;
          const u8 *m = mr + di;
          unsigned mw[2];
          __builtin_memcpy(mw, m, 8);
          const unsigned m8 = m[8];
          const U16x2 c1 = splat16(df), c0 = splat16(32 - df), rnd = u16x2_of(rnd_bits);
          U16x2 pa[4], pb[4], o[4];
          pa[0] = widen(0, mw[0], 0x0c010c00u); pa[1] = widen(0, mw[0], 0x0c030c02u); pa[2] = widen(0, mw[1], 0x0c010c00u); pa[3] = widen(0, mw[1], 0x0c030c02u);
          pb[0] = widen(0, mw[0], 0x0c020c01u); pb[1] = widen(mw[1], mw[0], 0x0c040c03u); pb[2] = widen(0, mw[1], 0x0c020c01u); pb[3] = widen(m8, mw[1], 0x0c040c03u);
          widen8(org + (r0 + i) * ostride, o);
          U16x2 v[4];
#pragma unroll
          for (int j = 0; j < 4; j++) v[j] = (pa[j] * c0 + (pb[j] * c1 + rnd)) >> (unsigned short)5;
          if (edge) v[0].x = (unsigned short)iclip(0, 255, (int)v[0].x + (((int)side[qa] - (int)side[0]) >> 1));
#pragma unroll
          for (int j = 0; j < 4; j++) d[i][j] = as_pk16(v[j] - o[j]);
        }
      }
    } else
#endif
    {
#pragma unroll
    for (int i = 0; i < NR; i++) {
      const int r = r0 + i;
      const int qa = q0 + r + 1, delta = qa * disp, di = delta >> 5, df = delta & 31;
      const u8 *m = mr + di;
      // the eight source pixels of the row in one 8-byte load (block origins are multiples of 8 in 8-byte aligned arrays)
      unsigned long long ow;
      __builtin_memcpy(&ow, __builtin_assume_aligned(org + r * ostride, 8), 8);
      int v[8];
      if (flat) {
        // planar (intra-generic.c:165-201, on the filtered references) and DC (intra-generic.c:210-241 with its edge smoothing below 32x32, on the unfiltered ones)
        const int y = by + r;
        if (mode == 1) {
          const u8 *top = s->ref[0][0], *left = s->ref[0][1];
          const int dc = s->dcval[0];
          for (int k = 0; k < 8; k++) v[k] = big ? dc : (int)filtered_dc_pixel(dc, bx + k, y, top, left);
        } else {
          const u8 *top = s->fref[0], *left = s->fref[1];
          for (int k = 0; k < 8; k++) v[k] = planar_pixel(log2w, bx + k, y, top, left);
        }
      } else {
        int a = m[0];
        for (int k = 0; k < 8; k++) {
          const int b = m[k + 1];
          v[k] = ((32 - df) * a + df * b + 16) >> 5;
          a = b;
        }
      }
      if (edge) v[0] = iclip(0, 255, v[0] + (((int)side[qa] - (int)side[0]) >> 1));
      for (int j = 0; j < 4; j++) d[i][j] = pk_make(v[2 * j] - (int)((ow >> (16 * j)) & 0xff), v[2 * j + 1] - (int)((ow >> (16 * j + 8)) & 0xff));
    }
    }
#ifndef KVZ_HOSTSIM
    if (PAIR ? half == 0 : true) d[0][0] = pk_add(d[0][0], KVZ_PK_BIAS);  // sample (0, 0), in the lane that holds row 0: see pk_abs2_biased
#endif
#pragma unroll
    for (int i = 0; i < NR; i++) {  // along the rows: columns k and k + 4, then k and k + 2 (k and k + 1 is the last stage: folded into the sum / spelled out in pk_abs2_biased)
      const Pk16 a0 = pk_add(d[i][0], d[i][2]), a1 = pk_add(d[i][1], d[i][3]), a2 = pk_sub(d[i][0], d[i][2]), a3 = pk_sub(d[i][1], d[i][3]);
      d[i][0] = pk_add(a0, a1); d[i][1] = pk_sub(a0, a1); d[i][2] = pk_add(a2, a3); d[i][3] = pk_sub(a2, a3);
    }
    u32 sum = 0;
    if constexpr (PAIR) {
#ifndef KVZ_HOSTSIM
      // rows r and r + 4 live in the two lanes of the pair: the even lane keeps the sums d + o, the odd one d - o, which is the NEGATED difference of the
      // rows -- only magnitudes are summed below -- so both are one multiply-add with a per-lane sign
      const Pk16 sgn = pk_make(half ? -1 : 1, half ? -1 : 1);
#pragma unroll
      for (int j = 0; j < 4; j++) {  // down the columns, two columns per register
        Pk16 e[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
          int mine;
          __builtin_memcpy(&mine, &d[i][j], 4);
          const int theirs = __builtin_amdgcn_update_dpp(0, mine, 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, true);
          Pk16 o;
          __builtin_memcpy(&o, &theirs, 4);
          e[i] = o * sgn + d[i][j];
        }
        const Pk16 b0 = pk_add(e[0], e[2]), b1 = pk_add(e[1], e[3]), b2 = pk_sub(e[0], e[2]), b3 = pk_sub(e[1], e[3]);
        sum = pk_abs2_biased(pk_add(b0, b1), sum); sum = pk_abs2_biased(pk_sub(b0, b1), sum); sum = pk_abs2_biased(pk_add(b2, b3), sum); sum = pk_abs2_biased(pk_sub(b2, b3), sum);
      }
      sum += (u32)__builtin_amdgcn_update_dpp(0, (int)sum, 0xB1, 0xF, 0xF, true);
      return sum;
#endif
    } else {
      for (int j = 0; j < 4; j++) {  // down the columns, two columns per register
        const Pk16 a0 = pk_add(d[0][j], d[4 % NR][j]), a1 = pk_add(d[1][j], d[5 % NR][j]), a2 = pk_add(d[2][j], d[6 % NR][j]), a3 = pk_add(d[3][j], d[7 % NR][j]);
        const Pk16 a4 = pk_sub(d[0][j], d[4 % NR][j]), a5 = pk_sub(d[1][j], d[5 % NR][j]), a6 = pk_sub(d[2][j], d[6 % NR][j]), a7 = pk_sub(d[3][j], d[7 % NR][j]);
        const Pk16 b0 = pk_add(a0, a2), b1 = pk_add(a1, a3), b2 = pk_sub(a0, a2), b3 = pk_sub(a1, a3);
        const Pk16 b4 = pk_add(a4, a6), b5 = pk_add(a5, a7), b6 = pk_sub(a4, a6), b7 = pk_sub(a5, a7);
#ifdef KVZ_HOSTSIM
        sum += pk_absmax(pk_add(b0, b1)) + pk_absmax(pk_sub(b0, b1)) + pk_absmax(pk_add(b2, b3)) + pk_absmax(pk_sub(b2, b3));
        sum += pk_absmax(pk_add(b4, b5)) + pk_absmax(pk_sub(b4, b5)) + pk_absmax(pk_add(b6, b7)) + pk_absmax(pk_sub(b6, b7));
#else
        sum = pk_abs2_biased(pk_add(b0, b1), sum); sum = pk_abs2_biased(pk_sub(b0, b1), sum); sum = pk_abs2_biased(pk_add(b2, b3), sum); sum = pk_abs2_biased(pk_sub(b2, b3), sum);
        sum = pk_abs2_biased(pk_add(b4, b5), sum); sum = pk_abs2_biased(pk_sub(b4, b5), sum); sum = pk_abs2_biased(pk_add(b6, b7), sum); sum = pk_abs2_biased(pk_sub(b6, b7), sum);
#endif
      }
#ifndef KVZ_HOSTSIM
      return sum;
#endif
    }
    return 2 * sum;
  }

#ifndef KVZ_HOSTSIM
  // Planar, DC or mode 34 on one 8x8 block of an 8x8 / 16x16 CU by EIGHT neighbouring lanes, one row each (lane & 7 = the row): the three modes that do not fit
  // the wavefront of the 32 other ones (two lanes a block, angular_block_satd) would otherwise cost a second round of that routine's four-rows-a-lane code for a
  // handful of lanes.  Row generator (the same integers as angular_block_satd's), the row's butterflies in the lane, the column's through three DPP exchanges
  // (partner 7 - i, then i ^ 2, i ^ 1: pairing with the mirrored lane instead of i ^ 4 only swaps outputs between the two halves and flips signs, and magnitudes
  // are what is summed), the last row stage folded into the sum as there.  Every lane returns the block's total.
  KVZ_DEV u32 flat_block_satd8(int log2w, int mode, int bx, int by, int xl, int yl, int lane) const
  {
    const int w = 1 << log2w, r = lane & 7, y = by + r;
    U16x2 o[4], v[4];
    widen8(org_at(0, xl + bx, yl + y), o);
    if (mode == 0) {  // planar
      const u8 *top = s->fref[0], *left = s->fref[1];
      const int tr = top[w + 1], bl = left[w + 1], l = left[y + 1];
      U16x2 tp[4];
      widen8(top + bx + 1, tp);
      const U16x2 dl = splat16(tr - l), wy = splat16(w - 1 - y), hb = splat16((w - 1) * l + tr + (y + 1) * bl + w);
#pragma unroll
      for (int j = 0; j < 4; j++) {
        U16x2 xs;
        xs.x = (unsigned short)(bx + 2 * j); xs.y = (unsigned short)(bx + 2 * j + 1);
        v[j] = (xs * dl + (tp[j] * wy + hb)) >> (unsigned short)(log2w + 1);
      }
    } else if (mode == 1) {  // DC with its edge smoothing
      const u8 *top = s->ref[0][0], *left = s->ref[0][1];
      const int dc = s->dcval[0];
      U16x2 tp[4];
      widen8(top + bx + 1, tp);
#pragma unroll
      for (int j = 0; j < 4; j++) v[j] = y == 0 ? (U16x2)((tp[j] + splat16(3 * dc + 2)) >> (unsigned short)2) : splat16(dc);
      if (bx == 0) v[0].x = (unsigned short)(y == 0 ? (left[1] + 2 * dc + top[1] + 2) >> 2 : (left[y + 1] + 3 * dc + 2) >> 2);
    } else {  // mode 34: displacement 32 a row on the filtered top reference (thresholds 7 and 1 are both below its distance 8), no fraction
      widen8(s->fref[0] + bx + y + 2, v);
    }
    Pk16 d[4];
#pragma unroll
    for (int j = 0; j < 4; j++) d[j] = as_pk16(v[j] - o[j]);
    if (r == 0) d[0] = pk_add(d[0], KVZ_PK_BIAS);  // see pk_abs2_biased
    {
      const Pk16 a0 = pk_add(d[0], d[2]), a1 = pk_add(d[1], d[3]), a2 = pk_sub(d[0], d[2]), a3 = pk_sub(d[1], d[3]);
      d[0] = pk_add(a0, a1); d[1] = pk_sub(a0, a1); d[2] = pk_add(a2, a3); d[3] = pk_sub(a2, a3);
    }
#define KVZ_FLAT8_STAGE(ctrl, bit)                                                                          \
    {                                                                                                       \
      const Pk16 sgn = pk_make((lane & (bit)) ? -1 : 1, (lane & (bit)) ? -1 : 1);                           \
      _Pragma("unroll") for (int j = 0; j < 4; j++) {                                                       \
        int mine;                                                                                           \
        __builtin_memcpy(&mine, &d[j], 4);                                                                  \
        const int theirs = __builtin_amdgcn_update_dpp(0, mine, ctrl, 0xF, 0xF, true);                       \
        Pk16 ot;                                                                                            \
        __builtin_memcpy(&ot, &theirs, 4);                                                                  \
        d[j] = ot * sgn + d[j];                                                                             \
      }                                                                                                     \
    }
    KVZ_FLAT8_STAGE(0x141 /* row_half_mirror */, 4)
    KVZ_FLAT8_STAGE(0x4E /* quad_perm [2,3,0,1] */, 2)
    KVZ_FLAT8_STAGE(0xB1 /* quad_perm [1,0,3,2] */, 1)
#undef KVZ_FLAT8_STAGE
    u32 sum = pk_abs2_biased(d[3], pk_abs2_biased(d[2], pk_abs2_biased(d[1], pk_abs2_biased(d[0], 0))));
    sum += (u32)__builtin_amdgcn_update_dpp(0, (int)sum, 0xB1, 0xF, 0xF, true);
    sum += (u32)__builtin_amdgcn_update_dpp(0, (int)sum, 0x4E, 0xF, 0xF, true);
    sum += (u32)__builtin_amdgcn_update_dpp(0, (int)sum, 0x141, 0xF, 0xF, true);
    return sum;
  }
#endif

  // Extended main reference of the angular modes 11..25 (see CtuShared::mref) for a 2^L2 CU.  All reads first, then all
  // writes: the loop bounds are compile-time constants, so the LDS round trips of different entries overlap.
  // 8x8 and 16x16 CUs: which reference sample an entry copies is a constant of (mode, q, CU size) -- filtered or not, top or left, projected index -- so it comes
  // out of Tables::mref_tab together with where it goes: a table load, a byte load and a byte store per entry instead of forty instructions of index arithmetic.
  // `pre`: the lane's table entries, fetched by the caller before the reference build (device: they are loads from global memory, and two barriers lie between
  // the request and the use)
  template <int L2>
  KVZ_DEV void build_mref(int tid, const u32 *pre = nullptr)
  {
    constexpr int W = 1 << L2, NQ = 2 * W + 2, THRES = L2 == 3 ? 7 : (L2 == 4 ? 1 : 0), N = (15 * NQ + KVZ_CTU_THREADS - 1) / KVZ_CTU_THREADS;
    u8 vals[N];
    if constexpr (L2 != 5) {
      static_assert(N * KVZ_CTU_THREADS <= 512, "Tables::mref_tab rows");
      const u8 *src = &s->ref[0][0][0];
      u8 *dst = &s->mref[0][0];
      u32 e[N];
      for (int k = 0; k < N; k++) e[k] = pre ? pre[k] : tb->mref_tab[L2 - 3][tid + k * KVZ_CTU_THREADS];
      for (int k = 0; k < N; k++) vals[k] = src[e[k] & 0xffff];
      for (int k = 0; k < N; k++) dst[e[k] >> 16] = vals[k];
      return;
    }
    for (int k = 0; k < N; k++) {
      const int i = imin(tid + k * KVZ_CTU_THREADS, 15 * NQ - 1), mode = 11 + i / NQ, q = i % NQ - W;
      const bool vertical = mode >= 18, filt = imin(iabs(mode - 26), iabs(mode - 10)) > THRES;
      const u8 *top = filt ? s->fref[0] : s->ref[0][0], *left = filt ? s->fref[1] : s->ref[0][1];
      const u8 *main_ref = vertical ? top : left, *side_ref = vertical ? left : top;
      const int idx = q >= 0 ? q : (128 + (-q) * (int)s->mode_inv[mode]) >> 8;
      vals[k] = (q >= 0 ? main_ref : side_ref)[imin(idx, 2 * W)];
    }
    for (int k = 0; k < N; k++) {
      const int i = tid + k * KVZ_CTU_THREADS;
      if (i < 15 * NQ) mref32()[(i / NQ) * KVZ_MREF32_STRIDE + KVZ_MREF32_ORG + i % NQ - W] = vals[k];
    }
  }
  // 32x32 rough search (S32): the transposed source block and the extended references sit in the 32-point transform scratch, idle until the reconstruction
  KVZ_DEV u8 *org_t32() const { return reinterpret_cast<u8 *>(s->tb_big); }
  KVZ_DEV u8 *mref32() const { return reinterpret_cast<u8 *>(s->tb_big) + 1024; }

  KVZ_DEV u32 mode_satd(int mode, int nblk) const  // SATD_NxN: sum of (block sum + 2) >> 2 (strategies-picture.h:53-69)
  {
    if (S32 && nblk == 16) return s->satd_raw[mode][0];  // a 32x32 CU's sum over its sixteen blocks, already rounded per block (rough_search)
    if (NXN && nblk == 0) return s->satd_raw[mode][0];   // a 4x4 PU: the finished 4x4 SATD (eval_pu)
    u32 v = 0;
    for (int b = 0; b < nblk; b++) v += (s->satd_raw[mode][b] + 2) >> 2;
    return v;
  }

  // search_intra.c:391-530 search_intra_rough: all 35 modes predicted + SATD-scored, then the reference's selection
  // order replayed on the cost table by one lane.  Leaves the winner in s->best_mode and the CU's info entries filled.
  // Also builds the chroma references of the CU (they only depend on neighbouring chroma reconstruction).
#ifndef KVZ_HOSTSIM
  // Wavefront minima by DPP row shifts (a lane without a source keeps its own value) + the four row results by lane index
  KVZ_DEV static unsigned long long wave_min_u64(unsigned long long v)
  {
#define KVZ_MIN64_STEP(ctrl)                                                                                         \
    {                                                                                                                \
      const int lo_ = (int)(unsigned)v, hi_ = (int)(unsigned)(v >> 32);                                              \
      const unsigned olo = (unsigned)__builtin_amdgcn_update_dpp(lo_, lo_, ctrl, 0xF, 0xF, false);                    \
      const unsigned ohi = (unsigned)__builtin_amdgcn_update_dpp(hi_, hi_, ctrl, 0xF, 0xF, false);                    \
      const unsigned long long o = ((unsigned long long)ohi << 32) | olo;                                            \
      v = o < v ? o : v;                                                                                             \
    }
    KVZ_MIN64_STEP(0x111) KVZ_MIN64_STEP(0x112) KVZ_MIN64_STEP(0x114) KVZ_MIN64_STEP(0x118)
#undef KVZ_MIN64_STEP
    unsigned long long r = ~0ull;
    for (int row = 0; row < 4; row++) {
      const unsigned long long o = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), 16 * row + 15) << 32) |
                                   (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, 16 * row + 15);
      r = o < r ? o : r;
    }
    return r;
  }
  KVZ_DEV static unsigned wave_min_u32(unsigned v)
  {
    int x = (int)v;
#define KVZ_MIN32_STEP(ctrl) { const unsigned o = (unsigned)__builtin_amdgcn_update_dpp(x, x, ctrl, 0xF, 0xF, false); x = (int)(o < (unsigned)x ? o : (unsigned)x); }
    KVZ_MIN32_STEP(0x111) KVZ_MIN32_STEP(0x112) KVZ_MIN32_STEP(0x114) KVZ_MIN32_STEP(0x118)
#undef KVZ_MIN32_STEP
    const unsigned a = (unsigned)__builtin_amdgcn_readlane(x, 15), b = (unsigned)__builtin_amdgcn_readlane(x, 31), c = (unsigned)__builtin_amdgcn_readlane(x, 47), d = (unsigned)__builtin_amdgcn_readlane(x, 63);
    const unsigned ab = a < b ? a : b, cd = c < d ? c : d;
    return ab < cd ? ab : cd;
  }
#endif
  // The selection order of search_intra_rough (search_intra.c:433-530) replayed on the cost table of all 35 modes (s->satd_raw; nblk 8x8 blocks per mode, 0: a 4x4
  // PU whose entry is the finished SATD).  Returns the winner on the lanes that ran the replay -- the host build: thread 0; the device: the wavefront playing
  // threads 0..63, all its lanes together -- and -1 on the others.
  KVZ_DEV int replay_selection(int tid, int log2w, int nblk) const
  {
    // The replay below is serial and uniform.  On the device the first wavefront's worth of thread ids runs it together:
    // lane m first works out mode m's SATD and cost, the replay then picks values out of those registers by lane index
    // (v_readlane) instead of recomputing them from LDS.  The host build computes them on demand -- same formulas.
#ifdef KVZ_HOSTSIM
      if (tid != 0) return -1;
      {
#define KVZ_RAW(md) mode_satd((md), nblk)
#define KVZ_COST(md, raw) ((double)(raw) + s->mode_bits_cost[(md) == p0 ? 1 : (((md) == p1 || (md) == p2) ? 2 : 0)])
#else
      if (tid >= 64) return -1;
      {
        const int my_mode = tid < 35 ? tid : 0;
        const u32 my_raw = mode_satd(my_mode, nblk);
        const double my_cost = (double)my_raw + s->mode_bits_cost[my_mode == s->preds[0] ? 1 : ((my_mode == s->preds[1] || my_mode == s->preds[2]) ? 2 : 0)];
        const int my_cost_lo = __double2loint(my_cost), my_cost_hi = __double2hiint(my_cost);
#define KVZ_RAW(md) ((u32)__builtin_amdgcn_readlane((int)my_raw, __builtin_amdgcn_readfirstlane(md)))
        // On the device an append only records WHEN a mode was appended (in the lane that holds the mode); "first minimum in append
        // order" is then one wavefront minimum of the costs -- non-negative doubles order like their bit patterns -- and one of
        // the append positions among the lanes that reach it, instead of two v_readlane and a double compare per append.
        int my_pos = 0, n_app = 0;
#endif
        // The list kvazaar builds (modes[], costs[]) is only ever read back as "first minimum in append order", so it is
        // replayed with a visited mask and running minima instead of arrays.
        unsigned long long visited = 0;
        double final_cost = 0;
        int final_mode = -1;
        const int8_t p0 = s->preds[0], p1 = s->preds[1], p2 = s->preds[2];
#ifdef KVZ_HOSTSIM
#define KVZ_APPEND(md, raw)                                                                                         \
        {                                                                                                           \
          const int md_ = (md);                                                                                     \
          visited |= 1ull << md_;                                                                                   \
          const double c_ = KVZ_COST(md_, raw);                                                                     \
          if (final_mode < 0 || c_ < final_cost) { final_cost = c_; final_mode = md_; }                             \
        }
#else
#define KVZ_APPEND(md, raw)                                                                                         \
        {                                                                                                           \
          const int md_ = __builtin_amdgcn_readfirstlane(md);                                                       \
          visited |= 1ull << md_;                                                                                   \
          if (tid == md_) my_pos = n_app;                                                                           \
          n_app++;                                                                                                  \
        }
#endif
        int offset = log2w == 2 ? 2 : (log2w == 3 ? 4 : 8);
        int32_t min_cost = 0x7fffffff, max_cost = -0x7fffffff - 1;
        int best_mode = -1;
        u32 first_min = 0;
        for (int mode = 2; mode <= 34; mode += 2 * offset)
          for (int i = 0; i < 2; i++) if (mode + i * offset <= 34) {
            const u32 raw = KVZ_RAW(mode + i * offset);
            KVZ_APPEND(mode + i * offset, raw);
            if ((int32_t)raw < min_cost) min_cost = (int32_t)raw;
            if ((int32_t)raw > max_cost) max_cost = (int32_t)raw;
            if (best_mode < 0 || raw < first_min) { first_min = raw; best_mode = mode + i * offset; }
          }
        double best_cost = min_cost;
        if (min_cost != max_cost) {
          while (offset > 1) {
            offset >>= 1;
            const int tm[2] = { best_mode - offset, best_mode + offset };
            for (int i = 0; i < 2; i++) if (tm[i] >= 2 && tm[i] <= 34) {
              const u32 raw = KVZ_RAW(tm[i]);
              KVZ_APPEND(tm[i], raw);
              if ((double)raw < best_cost) { best_cost = (double)raw; best_mode = tm[i]; }
            }
          }
        }
        const int add_modes[5] = { p0, p1, p2, 0, 1 };
        for (int p = 0; p < 5; p++)
          if (!((visited >> add_modes[p]) & 1)) { const u32 raw = KVZ_RAW(add_modes[p]); KVZ_APPEND(add_modes[p], raw); }
#undef KVZ_APPEND
#undef KVZ_RAW
#ifdef KVZ_HOSTSIM
#undef KVZ_COST
#else
        {
          const bool mine = tid < 35 && ((visited >> tid) & 1);
          const unsigned long long key = mine ? (((unsigned long long)(unsigned)my_cost_hi << 32) | (unsigned)my_cost_lo) : ~0ull;
          const unsigned long long kmin = wave_min_u64(key);
          final_mode = (int)(wave_min_u32((mine && key == kmin) ? (unsigned)((my_pos << 6) | tid) : ~0u) & 63);
          (void)final_cost;
        }
#endif
        (void)p0; (void)p1; (void)p2;
        return final_mode;
      }
  }
  template <class First>
  KVZ_DEV void rough_search(int lv, int x, int y, int depth, First first)
  {
    const int log2w = 6 - depth, w = 1 << log2w, xl = x - cx, yl = y - cy, nblk = (w >> 3) * (w >> 3);
#ifndef KVZ_HOSTSIM
    u32 mref_pre[4] = { 0, 0, 0, 0 };  // build_mref's table entries of this lane, requested ahead of the reference build
    if (!(S32 && log2w == 5)) {
      const u32 *row = tb->mref_tab[log2w - 3] + ((threadIdx.x + lane_rot) & (KVZ_CTU_THREADS - 1));
      for (int k = 0; k < 4; k++) mref_pre[k] = row[k * KVZ_CTU_THREADS];
    }
#endif
    build_refs(lv, x, y, log2w, depth == 3 ? 2 : log2w - 1, true, true, first);
    if (S32 && log2w == 5) {
      // 32x32 CU: all 35 modes x 16 blocks predicted and Hadamard-scored in registers like the angular modes of the smaller CUs
      // (angular_block_satd; planar and DC have their own row generator there), each block's SATD rounded on its own
      // (strategies-picture.h:53-69: SATD_32x32 = sum of sixteen 8x8 SATDs) and added to the mode's total
      KVZ_FOR_THREADS(tid) {
        if constexpr (S32) build_mref<5>(tid);
        u8 *ot = org_t32();
        for (int e = tid; e < 1024; e += KVZ_CTU_THREADS) ot[e] = *org_at(0, xl + (e >> 5), yl + (e & 31));
        for (int v = tid; v < 35; v += KVZ_CTU_THREADS) s->satd_raw[v][0] = 0;
        if (tid == KVZ_CTU_THREADS - 1) {
          const int left = x >= 4 ? neighbour_cu(lv, x - 1, y) : -1, above = (y >= 4 && yl > 0) ? neighbour_cu(lv, x, y - 1) : -1;
          mpm_candidates(y, left, above, s->preds);
        }
      }
      KVZ_SYNC();
      KVZ_PROF(KVZ_P_PRED35);
      KVZ_FOR_THREADS(tid) {
#ifdef KVZ_HOSTSIM
        for (int t = tid; t < 35 * 16; t += KVZ_CTU_THREADS) {
          const int mode = t >> 4, b = t & 15;
          KVZ_LDS_ADD(&s->satd_raw[mode][0], (angular_block_satd<false>(5, mode, (b & 3) * 8, (b >> 2) * 8, xl, yl, 0) + 2) >> 2);
        }
#else
        for (int t = tid; t < 35 * 32; t += KVZ_CTU_THREADS) {  // two lanes per (mode, block); modes 0 and 1 fill one wavefront
          const int p = t >> 1, mode = p >> 4, b = p & 15;
          const u32 v = angular_block_satd<true>(5, mode, (b & 3) * 8, (b >> 2) * 8, xl, yl, t & 1);
          if (!(t & 1)) KVZ_LDS_ADD(&s->satd_raw[mode][0], (v + 2) >> 2);
        }
#endif
      }
    } else {
    KVZ_FOR_THREADS(tid) {
      // extended main reference per angular mode
#ifdef KVZ_HOSTSIM
      if (log2w == 3) build_mref<3>(tid); else build_mref<4>(tid);
#else
      if (log2w == 3) build_mref<3>(tid, mref_pre); else build_mref<4>(tid, mref_pre);
#endif
      for (int e = tid; e < w * w; e += KVZ_CTU_THREADS) {
        const int ex = e >> log2w, ey = e & (w - 1);
        s->org_t[e] = *org_at(0, xl + ex, yl + ey);
      }
      if (tid == KVZ_CTU_THREADS - 1) {
        const int left = x >= 4 ? neighbour_cu(lv, x - 1, y) : -1, above = (y >= 4 && yl > 0) ? neighbour_cu(lv, x, y - 1) : -1;
        mpm_candidates(y, left, above, s->preds);
      }
    }
    KVZ_SYNC();
    KVZ_PROF(KVZ_P_PRED35);
    KVZ_FOR_THREADS(tid) {
      const int lb = 2 * (log2w - 3);  // log2(nblk)
      // All 35 modes predicted and Hadamard-scored in registers (angular_block_satd), the angular ones first, then planar and DC (task index 33, 34): the lanes of
      // those two share the last round, where the wavefront walks through their row generators one after the other
#ifdef KVZ_HOSTSIM
      for (int t = tid; t < 35 * nblk; t += KVZ_CTU_THREADS) {  // one thread per (mode, block)
        const int mi = t >> lb, mode = mi < 33 ? mi + 2 : mi - 33, b = t & (nblk - 1), bx = (b & ((w >> 3) - 1)) * 8, by = (b >> (log2w - 3)) * 8;
        s->satd_raw[mode][b] = angular_block_satd<false>(log2w, mode, bx, by, xl, yl, 0);
      }
#else
      // modes 2..33: TWO lanes per (mode, block), rows 0..3 and 4..7 of the block (angular_block_satd<true>): 64 lane tasks for an 8x8 CU -- one wavefront --,
      // 256 for a 16x16 one
      if (log2w == 4 && KVZ_CTU_THREADS == 128) {
        // a 16x16 CU has 128 (mode, block) pairs: ONE lane each, all eight rows in the lane (angular_block_satd<false>) -- one round of the two wavefronts where the
        // paired form needs two, and no exchange stage
        const int mode = 2 + (tid >> 2), b = tid & 3;
        s->satd_raw[mode][b] = angular_block_satd<false>(4, mode, (b & 1) * 8, (b >> 1) * 8, xl, yl, 0);
      } else
      for (int t = tid; t < 64 * nblk; t += KVZ_CTU_THREADS) {
        const int p = t >> 1, mode = 2 + (p >> lb), b = p & (nblk - 1), bx = (b & ((w >> 3) - 1)) * 8, by = (b >> (log2w - 3)) * 8;
        const u32 v = angular_block_satd<true>(log2w, mode, bx, by, xl, yl, t & 1);
        if (!(t & 1)) s->satd_raw[mode][b] = v;
      }
      // mode 34, planar, DC: EIGHT lanes per (mode, block), one row each (flat_block_satd8), taken from the far end of the thread ids -- the other wavefront of an
      // 8x8 CU (24 lane tasks; 96 for a 16x16 one)
      {
        const int t = KVZ_CTU_THREADS - 1 - tid;  // groups of eight tasks are groups of eight lanes
        if (t < 24 * nblk) {
          const int mb = t >> 3, mi = mb >> lb, mode = mi == 0 ? 34 : mi - 1, b = mb & (nblk - 1), bx = (b & ((w >> 3) - 1)) * 8, by = (b >> (log2w - 3)) * 8;
          const u32 v = flat_block_satd8(log2w, mode, bx, by, xl, yl, tid);
          if ((tid & 7) == 0) s->satd_raw[mode][b] = v;
        }
      }
#endif
    }
    }
    KVZ_SYNC();
    KVZ_PROF(KVZ_P_SATD);
    KVZ_FOR_THREADS(tid) {
      const int final_mode = replay_selection(tid, log2w, nblk);
      if (final_mode >= 0) {
        if (tid == 0) s->best_mode = final_mode;
        // lcu_fill_cu_info (search.c:137-159) for the searched CU: at most 2x2 entries
        for (int i = 0; tid == 0 && i < (w >> 3) * (w >> 3); i++) {
          const int cell = ((yl >> 3) + i / (w >> 3)) * 8 + (xl >> 3) + i % (w >> 3);
          CtuCu *cu = &s->cu[lv][cell];
          cu->type = 1; cu->depth = (u8)depth; cu->mode = (u8)final_mode; cu->tr_depth = (u8)depth;
          set_mode4(lv, cell, final_mode);
        }
      }
    }
    KVZ_SYNC();
    KVZ_PROF(KVZ_P_SELECT);
  }

  // Transform-unit geometry of one reconstruction step: luma w x w at (x, y) and/or chroma cw x cw.
  struct TuSet { int x, y, lw /* log2 luma or 0 */, lc /* log2 chroma or 0 */; };
  KVZ_DEV static int tu_log2(const TuSet &t, int c) { return c ? t.lc : t.lw; }
  // Plane c of transform scratch buffer p for the TU set t (see the union in CtuShared)
  KVZ_DEV i16 *tbuf(const TuSet &t, int p, int c) const
  {
    if (t.lw == 5) return s->tb_big + (c == 0 ? 0 : (c == 1 ? 1024 : 1280));  // one buffer, every stage in place
    return s->tb_small + p * 384 + (c == 0 ? 0 : (c == 1 ? 256 : 320));
  }
  // Entry (k, i) of the 2^l2-point transform matrix
  KVZ_DEV int dct_at(int l2, int k, int i) const
  {
#ifdef KVZ_HOSTSIM
    return s->dct32[(k << (10 - l2)) + i];
#else
    return s->dct_small[(l2 == 3 ? 0 : 64) + (k << l2) + i];  // 16 and 32 points never come here on the device
#endif
  }

  // Both passes of a 16- or 32-point transform of plane c, in place on x.  Device: wavefront c (mod the wavefronts of the
  // workgroup) runs them chained through MFMA registers (kvz_mfma.hpp).  Host: one thread, scalar loops through a temporary --
  // the same integers (dct-generic.c:559-579: the forward intermediate wraps to int16, both inverse stages clip).
  KVZ_DEV void transform_big(int l2, i16 *x, bool inverse, int tid, int c) const
  {
#ifndef KVZ_HOSTSIM
    if ((tid >> 6) != c % (KVZ_CTU_THREADS / 64)) return;
    if (l2 == 5) mfma_transform_block<32>(x, x, inverse, tb, tid & 63);
    else mfma_transform_block<16>(x, x, inverse, tb, tid & 63);
#else
    if (tid != 0) return;
    const int n = 1 << l2;
    i16 tmp[32 * 32];
    for (int pass = 0; pass < 2; pass++) {
      const i16 *src = pass == 0 ? x : tmp;
      i16 *dst = pass == 0 ? tmp : x;
      const int shift = inverse ? (pass == 0 ? 7 : 12) : (pass == 0 ? l2 - 1 : l2 + 6), add = 1 << (shift - 1);
      for (int e = 0; e < n * n; e++) {
        int a = 0;
        if (!inverse) { const int k = e >> l2, j = e & (n - 1); for (int i = 0; i < n; i++) a += dct_at(l2, k, i) * (int)src[(j << l2) + i]; }
        else { const int j = e >> l2, i = e & (n - 1); for (int k = 0; k < n; k++) a += dct_at(l2, k, i) * (int)src[(k << l2) + j]; }
        dst[e] = inverse ? (i16)iclip(-32768, 32767, (a + add) >> shift) : (i16)((a + add) >> shift);
      }
    }
#endif
  }

  // intra_recon_tb_leaf (intra.c:561-608) + kvz_quantize_residual (quant-generic.c:198-292) for the planes of `t`,
  // written into work-tree level lv.  Sets the cbf bits of the CU's info entry.  One barrier per stage.
  // Per-plane sums of the fused 8x8-CU stages: lanes 0..63 carry luma, 64..79 U, 80..95 V (the rest contribute 0).  Device: DPP
  // row reduction, then the luma wavefront adds its four row sums to acc3[0], the other one rows 0 / 1 to acc3[1] / acc3[2].
  KVZ_DEV void plane_add(u32 *acc3, u32 v, int tid) const
  {
#ifdef KVZ_HOSTSIM
    if (tid < 96) acc3[tid < 64 ? 0 : (tid < 80 ? 1 : 2)] += v;
#else
    int x = (int)v;
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, true);
    const u32 r0 = (u32)__builtin_amdgcn_readlane(x, 15), r1 = (u32)__builtin_amdgcn_readlane(x, 31), r2 = (u32)__builtin_amdgcn_readlane(x, 47), r3 = (u32)__builtin_amdgcn_readlane(x, 63);
    if ((tid & 63) == 0) {  // one writer per slot and stage: the slots were zeroed in stage 1
      if (tid < 64) acc3[0] += r0 + r1 + r2 + r3;
      else { acc3[1] += r0; acc3[2] += r1; }
    }
#endif
  }
  // The 8x8 CU -- an 8x8 luma and two 4x4 chroma units, 96 samples -- with every stage of recon_tus() in ONE pass, one lane per
  // sample of any plane: three quarters of the CUs the search evaluates are these, and a loop per plane runs each stage's code
  // three times for a handful of lanes.  Wavefront-uniform by construction: the wavefront playing threads 0..63 is all luma
  // (8-point), the other one all chroma (4-point).  Same arithmetic, same order of the integer operations per sample.
  KVZ_DEV void recon_cu8(int lv, const TuSet &t, int depth, int mode)
  {
    const int xl = t.x - cx, yl = t.y - cy;
    const CandView cv = cand_view(lv);
    // A lane's role is fixed by its wavefront: threads 0..63 are the luma samples (8-point transform), 64..79 / 80..95 the U / V samples (4-point).  Every stage
    // below is written once and instantiated per role (`luma` a compile-time constant), so that inside a wavefront plane, block size, shifts and trip counts are
    // constants -- as one body with a per-lane plane they were selects and counted loops.
#define KVZ_CU8_STAGE(body) KVZ_FOR_THREADS(tid) { if (tid < 64) { body(std::true_type(), tid); } else if (tid < 96) { body(std::false_type(), tid); } }
#define KVZ_CU8_ROLE(tid)                                                                                               \
    constexpr bool LUMA = decltype(luma)::value;                                                                        \
    constexpr int l2 = LUMA ? 3 : 2, n = 1 << l2, sh = LUMA ? 0 : 1;                                                    \
    const int c = LUMA ? 0 : ((tid) < 80 ? 1 : 2), e = LUMA ? (tid) : ((tid) & 15);                                     \
    (void)sh; (void)n; (void)c; (void)e
    // The five stages below synchronise per wavefront only (KVZ_WAVE_SYNC): luma lives on one wavefront, U and V on the other, and nothing crosses between the
    // planes before the cost -- each wavefront clears and fills its own sums (acc[0, 3] luma, acc[1, 2, 4, 5] chroma).
    KVZ_FOR_THREADS(tid) {
      if (tid < 2) s->acc[3 * tid] = 0;
      if (tid >= 64 && tid < 68) s->acc[1 + (tid - 64) + ((tid - 64) >> 1)] = 0;  // 1, 2, 4, 5
    }
    auto stage1 = [&](auto luma, int tid) {
      KVZ_CU8_ROLE(tid);
      const int px = e & (n - 1), py = e >> l2;
      const u8 p = predict_pixel(l2, mode, c, px, py);
      cv.at(c, (xl >> sh) + px, (yl >> sh) + py) = p;
      tbuf(t, 0, c)[e] = (i16)((int)*org_at(c, (xl >> sh) + px, (yl >> sh) + py) - (int)p);
    };
    KVZ_CU8_STAGE(stage1)
    KVZ_WAVE_SYNC();
    KVZ_PROF(KVZ_P_RPRED);
    auto stage2 = [&](auto luma, int tid) {  // forward transform (dct-generic.c:559-568), first pass
      KVZ_CU8_ROLE(tid);
      constexpr int shift = l2 - 1, add = 1 << (shift - 1);
      const int k = e >> l2, j = e & (n - 1);
      const i16 *src = tbuf(t, 0, c);
      int a = 0;
#pragma unroll
      for (int i = 0; i < n; i++) a += dct_at(l2, k, i) * (int)src[(j << l2) + i];
      tbuf(t, 1, c)[e] = (i16)((a + add) >> shift);
    };
    KVZ_CU8_STAGE(stage2)
    KVZ_WAVE_SYNC();
    KVZ_PROF(KVZ_P_FDCT);
    // second pass, and -- the coefficient a lane produces is the one it quantises -- straight on: quantise (quant-generic.c:57-81)
    // -> coefficient store + cost sums; dequantise (:335-339)
    KVZ_FOR_THREADS(tid) {
      u32 packed = 0;
      auto stage3 = [&](auto luma, int tid_) {
        KVZ_CU8_ROLE(tid_);
        constexpr int shift = l2 + 6, add = 1 << (shift - 1);
        const int k = e >> l2, j = e & (n - 1);
        const i16 *src = tbuf(t, 1, c);
        int a = 0;
#pragma unroll
        for (int i = 0; i < n; i++) a += dct_at(l2, k, i) * (int)src[(j << l2) + i];
        const int cf = (i16)((a + add) >> shift);
        const QuantScalars q = s->qs[l2 - 2][LUMA ? 0 : 1];
        int level = (int)(((u32)iabs(cf) * (u32)q.flat_q + (u32)q.add) >> q.q_bits);
        if (cf < 0) level = -level;
        level = iclip(-32768, 32767, level);
        levels_lds(lv, c)[e] = (i16)level;  // lv == 3 here
        int al = iabs(level);
        const u32 nz = al != 0;
        if (al > 3) al = 3;
        const u32 wsum = (u32)((m->coeff_weights >> (16 * al)) & 0xffff);
        tbuf(t, 0, c)[e] = (i16)iclip(-32768, 32767, (level * q.dq_scale + (1 << (q.dq_shift - 1))) >> q.dq_shift);
        packed = wsum | (nz << 24);
      };
      if (tid < 64) stage3(std::true_type(), tid); else if (tid < 96) stage3(std::false_type(), tid);
      // the plane's weight sum (< 2^22) and its count of levels travel in ONE word: one reduction instead of two.  (Per-lane LDS atomics instead of the DPP
      // reduction -- profiles/experiments, r05_g -- take 7 k instructions per CTU off the vector pipe and cost 6 % throughput: 64 lanes on one address.)
      plane_add(&s->acc[3], packed, tid);
    }
    KVZ_WAVE_SYNC();
    KVZ_PROF(KVZ_P_QUANT);
    auto stage4 = [&](auto luma, int tid) {  // inverse transform (dct-generic.c:570-579), first pass; only observable when the plane has coefficients
      KVZ_CU8_ROLE(tid);
      if (s->acc[3 + c] >> 24) {
        constexpr int shift = 7, add = 1 << (shift - 1);
        const int j = e >> l2, i = e & (n - 1);
        const i16 *src = tbuf(t, 0, c);
        int a = 0;
#pragma unroll
        for (int k = 0; k < n; k++) a += dct_at(l2, k, i) * (int)src[(k << l2) + j];
        tbuf(t, 1, c)[e] = (i16)iclip(-32768, 32767, (a + add) >> shift);
      }
    };
    KVZ_CU8_STAGE(stage4)
    KVZ_WAVE_SYNC();
    KVZ_PROF(KVZ_P_IDCT);
    // second pass, and straight on with the sample it produces: reconstruction (quant-generic.c:266-277) + SSD against the source
    // (search.c:500-505, 512-523)
    KVZ_FOR_THREADS(tid) {
      u32 ssd = 0;
      auto stage5 = [&](auto luma, int tid_) {
        KVZ_CU8_ROLE(tid_);
        u8 *rp = &cv.at(c, (xl >> sh) + (e & (n - 1)), (yl >> sh) + (e >> l2));
        int v = *rp;
        if (s->acc[3 + c] >> 24) {
          constexpr int shift = 12, add = 1 << (shift - 1);
          const int j = e >> l2, i = e & (n - 1);
          const i16 *src = tbuf(t, 1, c);
          int a = 0;
#pragma unroll
          for (int k = 0; k < n; k++) a += dct_at(l2, k, i) * (int)src[(k << l2) + j];
          const i16 res = (i16)iclip(-32768, 32767, (a + add) >> shift);
          v = iclip(0, 255, (int)(i16)(res + v));
          *rp = (u8)v;
        }
        const int d = (int)*org_at(c, (xl >> sh) + (e & (n - 1)), (yl >> sh) + (e >> l2)) - v;
        ssd = (u32)(d * d);
      };
      if (tid < 64) stage5(std::true_type(), tid); else if (tid < 96) stage5(std::false_type(), tid);
      plane_add(&s->acc[0], ssd, tid);
    }
    KVZ_SYNC();
    KVZ_FOR_THREADS(tid) {
      if (tid == 0) {  // cbf bits of the TU's top-left CU entry (transform.c:314, 409-411): all three planes' counts, so behind the barrier; only thread 0 reads them
                       // next (eval_cu's cost phase, same thread: no barrier in between)
        CtuCu *cu = &s->cu[lv][(yl >> 3) * 8 + (xl >> 3)];
        for (int cc = 0; cc < 3; cc++) {
          const u32 packed = s->acc[3 + cc];  // unpacked for the cost: weight sum, count
          s->acc[3 + cc] = packed & 0xffffffu; s->acc[6 + cc] = packed >> 24;
          cbf_clear(&cu->cbf, depth, cc);
          if (packed >> 24) cbf_set(&cu->cbf, depth, cc);
        }
      }
    }
    if (cabac_on()) KVZ_SYNC();  // the counting-mode coder runs on every lane and asks acc[6..8] which planes have levels (price_unit_coeffs)
    KVZ_PROF(KVZ_P_RECON);
#undef KVZ_CU8_ROLE
#undef KVZ_CU8_STAGE
  }

  // Where the quantised levels of plane c of a transform unit at (xl, yl) go when work-tree level lv evaluates it (see coeff_level())
  KVZ_DEV i16 *coeff_dst(int lv, int c, int xl, int yl) const
  {
    const int sh = c ? 1 : 0;
    if ((NXN && lv == 4) || lv == 3) return levels_lds(lv, c);
    return lv == 2 ? s->lv2_coeff + (c == 0 ? 0 : (c == 1 ? 256 : 320))
         : lv == 1 ? s->lv1_coeff + (c == 0 ? 0 : (c == 1 ? 1024 : 1280)) : coeff_level(lv) + plane_off(c) + zorder(xl >> sh, yl >> sh);
  }
  // Entry (k, i) of the transform of a 2^l2 block of plane c: the DCT, except 4x4 intra luma (strategies-dct.c:82-86, 111-115: the DST), which only the PUs of
  // an NxN CU have
  KVZ_DEV int tmat(int l2, int c, int k, int i) const
  {
    if (NXN && l2 == 2 && c == 0) return tb->dst4[4 * k + i];
    return dct_at(l2, k, i);
  }

  KVZ_DEV void recon_tus(int lv, const TuSet &t, int depth, int mode, bool refs_ready = false)
  {
    const int xl = t.x - cx, yl = t.y - cy;
    const CandView cv = cand_view(lv);
    if (!refs_ready) build_refs(lv, t.x, t.y, t.lw, t.lc, t.lw != 0, t.lc != 0);
    if (!RDOQ && KVZ_CTU_THREADS == 128 && t.lw == 3 && t.lc == 2) { recon_cu8(lv, t, depth, mode); return; }
    // stage 1: prediction -> rec (as kvazaar blits it before quantising) and residual
    KVZ_FOR_THREADS(tid) {
      if (tid < 16) s->acc[tid] = 0;
      for (int c = 0; c < 3; c++) {
        const int l2 = tu_log2(t, c);
        if (!l2) continue;
        const int w = 1 << l2, sh = c ? 1 : 0;
        for (int e = tid; e < w * w; e += KVZ_CTU_THREADS) {
          const int px = e & (w - 1), py = e >> l2;
          const u8 p = predict_pixel(l2, mode, c, px, py);
          cv.at(c, (xl >> sh) + px, (yl >> sh) + py) = p;
          tbuf(t, 0, c)[e] = (i16)((int)*org_at(c, (xl >> sh) + px, (yl >> sh) + py) - (int)p);
        }
      }
    }
    KVZ_SYNC();
    KVZ_PROF(KVZ_P_RPRED);
    // stages 2-3: forward transform (dct-generic.c:559-568; 4x4 chroma uses the DCT, strategies-dct.c:82-86)
    for (int pass = 0; pass < 2; pass++) {
      KVZ_FOR_THREADS(tid) {
        for (int c = 0; c < 3; c++) {
          const int l2 = tu_log2(t, c);
          if (!l2) continue;
          const int n = 1 << l2, shift = pass == 0 ? l2 - 1 : l2 + 6, add = 1 << (shift - 1);
          const i16 *src = tbuf(t, pass, c);
          i16 *dst = tbuf(t, pass ^ 1, c);
          if (l2 >= 4) {  // 16 / 32 points: both passes at once (pass 0 only), in place on buffer 0, by one wavefront per plane
            if (pass == 0) transform_big(l2, tbuf(t, 0, c), false, tid, c);
            continue;
          }
          for (int e = tid; e < n * n; e += KVZ_CTU_THREADS) {
            const int k = e >> l2, j = e & (n - 1);
            int a = 0;
            for (int i = 0; i < n; i++) a += tmat(l2, c, k, i) * (int)src[(j << l2) + i];
            dst[e] = (i16)((a + add) >> shift);
          }
        }
      }
      KVZ_SYNC();
    }
    KVZ_PROF(KVZ_P_FDCT);
    // stage 4: quantise (quant-generic.c:57-81) -> coefficient store + cost sums; dequantise (:335-339) -> tb[1]
    // With RDOQ (quant-generic.c:234-244) the levels come from kvz_rdoq instead: serial per block, so one lane per plane runs it -- luma on the
    // first wavefront, U and V on two lanes of the other -- on the contexts of the row's coder as they stood when this CTU began (pre[0] =
    // state->cabac, rdo.c:665), and the loop below takes the levels from where it left them.
    if (RDOQ && m->rdoq) {
      KVZ_FOR_THREADS(tid) {
        // one wavefront per block (rdoq_block_wave): the one playing threads 0..63 takes the luma block, the next one U then V
#ifdef KVZ_HOSTSIM
        const int wv = tid == 0 ? 0 : (tid == 64 ? 1 : -1), lane = 0;
#else
        const int wv = tid >> 6, lane = tid & 63;
#endif
        for (int c = 0; c < 3; c++) {
          const int l2 = tu_log2(t, c);
          if (!l2 || wv != (c ? 1 : 0)) continue;
          // the levels go to LDS whatever the work-tree level: CUs whose levels live in HBM (8x8 CUs, the units of the 64x64 attempt) have a staging copy
          // for the coefficient cost anyway (levels_lds); the loop below moves them on
          i16 *cout = (lv == 3 || lv == 0) ? levels_lds(lv, c) : coeff_dst(lv, c, xl, yl);
          RdoqWaveArgs ra;
          ra.ptab = (KVZ_LDS_PTR(const i32))rl->ptab; ra.coef = (KVZ_LDS_PTR(const i16))tbuf(t, 0, c); ra.dest = (KVZ_LDS_PTR(i16))cout;
          ra.diag8 = (KVZ_LDS_PTR(const u8))rl->diag8; ra.lambda = m->lambda; ra.qp = m->qp; ra.log2w = l2; ra.type = c ? 2 : 0;
          ra.scan_mode = scan_order(mode, depth);
          // tr_depth = cu->tr_depth - cu->depth: 1 for the 32x32 units of the 64x64 attempt (level 0), 0 otherwise -- plus one for an NxN CU
          // (quant-generic.c:237-238): 2 for the blocks of its PUs (level 4)
          ra.tr_depth = lv == 4 ? 2 : (lv == 0 ? 1 : 0);
#if defined(KVZ_CTU_PROFILE) && !defined(KVZ_HOSTSIM)
          ra.prof = s->prof_rq;  // LDS: one global atomic per call and section would be the hottest cache line of the device
#endif
#if defined(KVZ_CTU_PROFILE) && !defined(KVZ_HOSTSIM)
          const unsigned long long tc0 = __builtin_amdgcn_s_memtime();
          rdoq_block_wave(ra, lane);
          if (c == 0 && lane == 0) ra.prof[7] += __builtin_amdgcn_s_memtime() - tc0;  // the call as the caller sees it: minus the routine's own clock = what calling it costs
#else
          rdoq_block_wave(ra, lane);
#endif
        }
      }
      KVZ_SYNC();
      KVZ_PROF(KVZ_P_RDOQ);
    }
    KVZ_FOR_THREADS(tid) {
      for (int c = 0; c < 3; c++) {
        const int l2 = tu_log2(t, c);
        if (!l2) continue;
        const int n2 = 1 << (2 * l2);
        const QuantScalars qf = s->qs[l2 - 2][c ? 1 : 0];  // forward and inverse share the plane's scaled QP (U and V alike)
        const QuantScalars qi = qf;
        i16 *cout = coeff_dst(lv, c, xl, yl);
        i16 *stage = (cabac_on() && lv == 0) ? levels_lds(lv, c) : nullptr;  // see levels_lds(); an 8x8 CU's destination IS its LDS slot
        const i16 *src = tbuf(t, 0, c);
        i16 *dq = tbuf(t, 1, c);
        u32 wsum = 0, nz = 0;
        for (int e = tid; e < n2; e += KVZ_CTU_THREADS) {
          const int cf = src[e];
          // |cf| * q + add < 2^31 for 8-bit flat lists (32767 * 26214 + (171 << 18)), so 32-bit arithmetic is exact
          int level;
          if (RDOQ && m->rdoq) { level = (stage ? stage : cout)[e]; if (stage) cout[e] = (i16)level; }  // kvz_rdoq left them in LDS (above)
          else {
            level = (int)(((u32)iabs(cf) * (u32)qf.flat_q + (u32)qf.add) >> qf.q_bits);
            if (cf < 0) level = -level;
            level = iclip(-32768, 32767, level);
            cout[e] = (i16)level;
          }
          if (stage && !(RDOQ && m->rdoq)) stage[e] = (i16)level;
          int a = iabs(level);
          nz += a != 0;
          if (a > 3) a = 3;
          wsum += (u32)((m->coeff_weights >> (16 * a)) & 0xffff);
          dq[e] = (i16)iclip(-32768, 32767, (level * qi.dq_scale + (1 << (qi.dq_shift - 1))) >> qi.dq_shift);
        }
        block_add(&s->acc[3 + c], wsum);
        block_add(&s->acc[6 + c], nz);
      }
    }
    KVZ_SYNC();
    KVZ_PROF(KVZ_P_QUANT);
    // stages 5-6: inverse transform (dct-generic.c:570-579), only observable when the plane has coefficients
    for (int pass = 0; pass < 2; pass++) {
      KVZ_FOR_THREADS(tid) {
        for (int c = 0; c < 3; c++) {
          const int l2 = tu_log2(t, c);
          if (!l2 || !s->acc[6 + c]) continue;
          const int n = 1 << l2, shift = pass == 0 ? 7 : 12, add = 1 << (shift - 1);
          const i16 *src = tbuf(t, pass ^ 1, c);
          i16 *dst = tbuf(t, pass, c);
          if (l2 >= 4) {  // both passes at once, in place on buffer 1 (where the dequantised coefficients are)
            if (pass == 0) transform_big(l2, tbuf(t, 1, c), true, tid, c);
            continue;
          }
          for (int e = tid; e < n * n; e += KVZ_CTU_THREADS) {
            const int j = e >> l2, i = e & (n - 1);
            int a = 0;
            for (int k = 0; k < n; k++) a += tmat(l2, c, k, i) * (int)src[(k << l2) + j];
            dst[e] = (i16)iclip(-32768, 32767, (a + add) >> shift);
          }
        }
      }
      KVZ_SYNC();
    }
    KVZ_PROF(KVZ_P_IDCT);
    // stage 7: reconstruction (quant-generic.c:266-277) + SSD against the source (search.c:500-505, 512-523)
    KVZ_FOR_THREADS(tid) {
      for (int c = 0; c < 3; c++) {
        const int l2 = tu_log2(t, c);
        if (!l2) continue;
        const int w = 1 << l2, sh = c ? 1 : 0;
        const bool has = s->acc[6 + c] != 0;
        u32 ssd = 0;
        for (int e = tid; e < w * w; e += KVZ_CTU_THREADS) {
          u8 *rp = &cv.at(c, (xl >> sh) + (e & (w - 1)), (yl >> sh) + (e >> l2));
          int v = *rp;
          if (has) { v = iclip(0, 255, (int)(i16)(tbuf(t, 1, c)[e] + v)); *rp = (u8)v; }
          const int d = (int)*org_at(c, (xl >> sh) + (e & (w - 1)), (yl >> sh) + (e >> l2)) - v;
          ssd += (u32)(d * d);
        }
        block_add(&s->acc[c], ssd);
      }
      if (tid == 0) {  // cbf bits of the TU's top-left CU entry (transform.c:314, 409-411)
        if (NXN && lv == 4) {  // a PU of the NxN attempt: its flags wait in RdoqLds until the partition wins (nxn_attempt)
          rl->pu_cbf[rl->n_pu] = s->acc[6] != 0;
          if (t.lc) { rl->pu_cbf_c[0] = s->acc[7] != 0; rl->pu_cbf_c[1] = s->acc[8] != 0; }
        } else {
          CtuCu *cu = &s->cu[lv][(yl >> 3) * 8 + (xl >> 3)];
          for (int c = 0; c < 3; c++) if (tu_log2(t, c)) { cbf_clear(&cu->cbf, depth, c); if (s->acc[6 + c]) cbf_set(&cu->cbf, depth, c); }
        }
      }
    }
    KVZ_SYNC();
    KVZ_PROF(KVZ_P_RECON);
  }

  KVZ_DEV QuantScalars quant_scalars_dev(int log2w, int type) const
  {
    // quant-generic.c:57-66, 303-339 with flat scaling lists, 8 bit, I slice (kvz_tables.hpp quant_scalars)
    const int quant_scales[6] = { 26214, 23302, 20560, 18396, 16384, 14564 }, inv_scales[6] = { 40, 45, 51, 57, 64, 72 };
    const u8 chroma_scale[58] = { 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 29, 30, 31, 32,
                                  33, 33, 34, 34, 35, 35, 36, 36, 37, 37, 38, 39, 40, 41, 42, 43, 44, 45, 46, 47, 48, 49, 50, 51 };
    int qps = m->qp;
    if (type != 0) { int q = iclip(0, 57, m->qp); qps = chroma_scale[q]; }
    QuantScalars q;
    const int transform_shift = 15 - 8 - log2w;
    q.q_bits = 14 + qps / 6 + transform_shift;
    q.add = 171 << (q.q_bits - 9);
    q.flat_q = quant_scales[qps % 6];
    q.dq_shift = 20 - 14 - transform_shift;
    q.dq_scale = inv_scales[qps % 6] << (qps / 6);
    q.dq_list = 0; q.dq_qp_per = qps / 6;
    return q;
  }

  // get_coeff_cabac_cost (rdo.c:220-263) of the planes of one transform unit that have levels, luma first (search.c:518-547).
  // On the device the callers bring the whole wavefront that plays threads 0..63 (KVZ_UNIT_COEFF_BITS), in the host simulation thread 0.
  KVZ_DEV double unit_coeff_bits(CtxSet *c, bool update, int lv, int depth, int mode, int cb_y, int cb_u, int cb_v, int tid) const
  {
    const int lw = 6 - depth, lc = depth >= 3 ? 2 : lw - 1, scan = scan_order(mode, depth);
    double bits = 0;
#ifdef KVZ_HOSTSIM
    (void)tid;
    if (cb_y) bits += coeff_cabac_bits(c, update, levels_lds(lv, 0), lw, 0, scan);
    if (cb_u) bits += coeff_cabac_bits(c, update, levels_lds(lv, 1), lc, 2, scan);
    if (cb_v) bits += coeff_cabac_bits(c, update, levels_lds(lv, 2), lc, 2, scan);
#else  // both wavefronts are here: the one playing threads 0..63 takes the luma block, the other one the two chroma blocks -- luma and
       // chroma contexts are disjoint, so the two chains (Y | U -> V) are independent even with updates on
    // ... and the luma block's own bins split again: its significance flags (the longest chains) stay on the first wavefront, the second one
    // takes the luma block's other classes after the chroma blocks
    if (tid < 64) {
      if (cb_y) bits += coeff_cabac_bits_wave(c, update, levels_lds(lv, 0), lw, 0, scan, 1);
    } else {
      if (cb_u) bits += coeff_cabac_bits_wave(c, update, levels_lds(lv, 1), lc, 2, scan);
      if (cb_v) bits += coeff_cabac_bits_wave(c, update, levels_lds(lv, 2), lc, 2, scan);
      if (cb_y) bits += coeff_cabac_bits_wave(c, update, levels_lds(lv, 0), lw, 0, scan, 2);
    }
#endif
    return bits;
  }
  // ... of the unit just reconstructed (levels in levels_lds(lv), non-zero counts in acc[6..8]) into *out, for thread 0 to pick up
  // in a later phase (same wavefront: no barrier needed in between)
#ifdef KVZ_HOSTSIM
#define KVZ_UNIT_COEFF_BITS(tid) ((tid) == 0)
#else
#define KVZ_UNIT_COEFF_BITS(tid) ((tid) < 64)
#endif
  KVZ_DEV void price_unit_coeffs(CtxSet *c, bool update, int lv, int depth, int mode, double *out) const
  {
#ifdef KVZ_HOSTSIM
    KVZ_FOR_THREADS(tid) {
      if (tid == 0) *out = unit_coeff_bits(c, update, lv, depth, mode, s->acc[6] != 0, s->acc[7] != 0, s->acc[8] != 0, tid);
    }
#else
    // the two wavefronts' shares meet in two doubles that alias child_acc[3][1..4]: dead here -- it is only written after the fourth
    // unit of the 64x64 attempt has been priced, and read right after that loop (try_merge)
    double *part = reinterpret_cast<double *>(&s->child_acc[3][1]);
    KVZ_FOR_THREADS(tid) {
      const double b = unit_coeff_bits(c, update, lv, depth, mode, s->acc[6] != 0, s->acc[7] != 0, s->acc[8] != 0, tid);
      if ((tid & 63) == 0) part[tid >> 6] = b;
    }
    KVZ_SYNC();
    KVZ_FOR_THREADS(tid) {
      if (tid == 0) *out = part[0] + part[1];  // multiples of 2^-15 far below 2^38: the sum is exact in any order
    }
#endif
  }
  // search.c:425-541 cu_rd_cost_tr_split_accurate for one leaf TU group whose sums sit in s->acc (lane 0 only)
  // `known_coeff_bits`: the units of the 64x64 attempt had their coefficients priced when their levels were staged (try_merge)
  KVZ_DEV double leaf_rd_cost(CtxSet *c, bool update, int lv, int xl, int yl, int depth, int cu_depth, bool code_cbf_u, bool code_cbf_v,
                              const double *known_coeff_bits = nullptr) const
  {
    const CtuCu *tr_cu = &s->cu[lv][(yl >> 3) * 8 + (xl >> 3)];
    double tr_tree_bits = 0, coeff_bits = 0;
    const int cb_u = cbf_is_set(tr_cu->cbf, depth, 1), cb_v = cbf_is_set(tr_cu->cbf, depth, 2), cb_y = cbf_is_set(tr_cu->cbf, depth, 0);
    if (code_cbf_u) tr_tree_bits += ctx_price(c, KVZ_CX_CBF_CHROMA + depth - cu_depth, cb_u, update);
    if (code_cbf_v) tr_tree_bits += ctx_price(c, KVZ_CX_CBF_CHROMA + depth - cu_depth, cb_v, update);
    tr_tree_bits += ctx_price(c, KVZ_CX_CBF_LUMA + (depth == cu_depth ? 1 : 0), cb_y, update);
    if (known_coeff_bits) coeff_bits += *known_coeff_bits;  // always the case with the CABAC model
    else {  // kvz_fast_coeff_cost (rdo.c:311-326): the weight sums of the quantisation stage
      if (cb_y) coeff_bits += (double)s->acc[3] / 256.0;
      if (cb_u) coeff_bits += (double)s->acc[4] / 256.0;
      if (cb_v) coeff_bits += (double)s->acc[5] / 256.0;
    }
    const unsigned luma_ssd = s->acc[0], chroma_ssd = s->acc[1] + s->acc[2];
    const double bits = tr_tree_bits + coeff_bits;
    return luma_ssd * 0.8 + chroma_ssd * 1.5 + bits * m->lambda;
  }

  // cu_bits() * lambda + leaf_rd_cost() of the CU just evaluated at its own depth (eval_cu: search.c:895-940 + 425-541), bin for bin the same
  // prices and transitions on the search contexts -- but the six bins sit on five different contexts (only U's and V's coded-block flags share
  // one), so all states are read first, all prices and successors looked up next, and everything is written back at the end: three LDS round
  // trips on thread 0's critical path instead of one read-price-write chain per bin.  The sums of prices are exact in any order (multiples
  // of 2^-15 below 2^10), the cost expressions are the callees' own.
  KVZ_DEV double cu_cost_batched(int lv, int x, int y, int depth, int mode, const int8_t *known_preds, const double *known_coeff_bits) const
  {
    const int xl = x - cx, yl = y - cy, w = 64 >> depth;
    u8 *cs = s->cab.s;
    const bool adaptive = m->adaptive != 0;
    // which contexts, which bins
    const bool has_sp = depth == 3 || !(F.W < x + w || F.H < y + w);
    const int i_sp = depth == 3 ? KVZ_CX_PART : KVZ_CX_SPLIT + (has_sp ? split_model(lv, x, y, depth) : 0), b_sp = depth == 3 ? 1 : 0;
    int8_t preds[3];
    const bool no_left = (x & 63) == 0;  // the mock encode's left neighbour (intra_mode_syntax_bits)
    if (known_preds && !(no_left && x > 0)) { preds[0] = known_preds[0]; preds[1] = known_preds[1]; preds[2] = known_preds[2]; }
    else {
      const int left = (x > 0 && !no_left) ? neighbour_cu(lv, x - 1, y) : -1;
      const int above = ((y & 63) > 0 && y > 0) ? neighbour_cu(lv, x, y - 1) : -1;
      mpm_candidates(y, left, above, preds);
    }
    const int b_in = (mode == preds[0] || mode == preds[1] || mode == preds[2]) ? 1 : 0;
    const CtuCu *tr_cu = &s->cu[lv][(yl >> 3) * 8 + (xl >> 3)];
    const int cb_u = cbf_is_set(tr_cu->cbf, depth, 1), cb_v = cbf_is_set(tr_cu->cbf, depth, 2), cb_y = cbf_is_set(tr_cu->cbf, depth, 0);
    const int i_cc = KVZ_CX_CBF_CHROMA, i_cl = KVZ_CX_CBF_LUMA + 1;
    // states, then prices and successors, then the second bin on the chroma flag's context
    const int s_sp = cs[i_sp], s_in = cs[KVZ_CX_INTRA], s_ch = cs[KVZ_CX_CHROMA], s_cu = cs[i_cc], s_cl = cs[i_cl];
    const float f_sp = s->entropy_fbits[s_sp ^ b_sp], f_in = s->entropy_fbits[s_in ^ b_in], f_ch = s->entropy_fbits[s_ch ^ 0];
    const float f_cu = s->entropy_fbits[s_cu ^ cb_u], f_cl = s->entropy_fbits[s_cl ^ cb_y];
    const int n_sp = ctx_next(s_sp, b_sp), n_in = ctx_next(s_in, b_in), n_ch = ctx_next(s_ch, 0), n_cu = ctx_next(s_cu, cb_u), n_cl = ctx_next(s_cl, cb_y);
    const int s_cv = adaptive ? n_cu : s_cu;
    const float f_cv = s->entropy_fbits[s_cv ^ cb_v];
    const int n_cv = ctx_next(s_cv, cb_v);
    if (adaptive) {
      if (has_sp) cs[i_sp] = (u8)n_sp;
      cs[KVZ_CX_INTRA] = (u8)n_in; cs[KVZ_CX_CHROMA] = (u8)n_ch; cs[i_cc] = (u8)n_cv; cs[i_cl] = (u8)n_cl;
    }
    // cu_bits
    double bits = 0;
    if (has_sp) bits += (double)f_sp;
    bits += ((double)f_in + (b_in ? ((mode == preds[0]) ? 1 : 2) : 5)) + (double)f_ch;
    double cost = bits * m->lambda;
    // leaf_rd_cost
    double tr_tree_bits = 0, coeff_bits = 0;
    tr_tree_bits += (double)f_cu;
    tr_tree_bits += (double)f_cv;
    tr_tree_bits += (double)f_cl;
    if (known_coeff_bits) coeff_bits += *known_coeff_bits;
    else {
      if (cb_y) coeff_bits += (double)s->acc[3] / 256.0;
      if (cb_u) coeff_bits += (double)s->acc[4] / 256.0;
      if (cb_v) coeff_bits += (double)s->acc[5] / 256.0;
    }
    const unsigned luma_ssd = s->acc[0], chroma_ssd = s->acc[1] + s->acc[2];
    const double lbits = tr_tree_bits + coeff_bits;
    cost += luma_ssd * 0.8 + chroma_ssd * 1.5 + lbits * m->lambda;
    return cost;
  }

  KVZ_DEV void fill_cu(int lv, int xl, int yl, int w, int type, int depth, int mode, int tr_depth)
  {
    KVZ_FOR_THREADS(tid) {
      const int n = w >> 3, ln = w == 8 ? 0 : (w == 16 ? 1 : (w == 32 ? 2 : 3));
      if (tid < n * n) {
        const int cell = ((yl >> 3) + (tid >> ln)) * 8 + (xl >> 3) + (tid & (n - 1));
        CtuCu *c = &s->cu[lv][cell];
        c->type = (u8)type; c->depth = (u8)depth; c->mode = (u8)mode; c->tr_depth = (u8)tr_depth;
        set_mode4(lv, cell, mode);
      }
    }
    KVZ_SYNC();
  }

  // The work-tree copies of search.c:55-122 for the region (xl, yl, w), in one phase:
  //   CU info      level cu_from -> levels cu_to_lo..cu_to_hi (copy_cu_info),
  //   pixels       candidate of depth pix_lv -> decided picture (copy_cu_pixels; -1: the decided picture already holds them),
  //   coefficients challenger block -> output block (copy_cu_coeffs) when `coeffs`.
  // Thread 0 also records the verdict of depth `res_depth`: res[d] = the cost search_cu returns for the CU (search.c:1046-1060).
  KVZ_DEV void commit(int cu_from, int cu_to_lo, int cu_to_hi, int pix_lv, bool coeffs, int xl, int yl, int w, int res_depth, bool split_won)
  {
    const CandView cv = cand_view(pix_lv < 0 ? 0 : pix_lv);
    KVZ_FOR_THREADS(tid) {
      if (tid == 0) {
        const double r = split_won ? s->split_cost[res_depth] : s->cost[res_depth];
        s->res[res_depth] = r;
        if (res_depth > 0) s->split_cost[res_depth - 1] += r;  // the parent's running sum (search.c:1005-1010)
        // search.c:1051: an unsplit CU below depth 0 continues from the contexts as they were after it was priced -- for a merge
        // those at entry, since the merge is priced with updates off (search.c:1005-1041)
      }
      if (!split_won && res_depth == 2) ctx_copy_lanes(&s->cab, &s->post2, tid);
      if (!split_won && res_depth == 1) ctx_copy_lanes(&s->cab, &s->pre[1], tid);
      const int n = w >> 3, lw = w == 16 ? 4 : (w == 32 ? 5 : 6), ln = lw - 3, cw = w >> 1;
      if (tid < n * n) {
        const int i = ((yl >> 3) + (tid >> ln)) * 8 + (xl >> 3) + (tid & (n - 1));
        for (int to = cu_to_lo; to <= cu_to_hi; to++) {
          s->cu[to][i] = s->cu[cu_from][i];
          if constexpr (NXN) { if (m->search_nxn) __builtin_memcpy(rl->mode4[to][i], rl->mode4[cu_from][i], 4); }
        }
      }
      if (pix_lv >= 0) {
        for (int e = tid; e < w * w; e += KVZ_CTU_THREADS) {
          const int px = xl + (e & (w - 1)), py = yl + (e >> lw);
          s->dec[py * 64 + px] = cv.at(0, px, py);
        }
        for (int e = tid; e < 2 * cw * cw; e += KVZ_CTU_THREADS) {
          const int c = 1 + (e >= cw * cw), k = e & (cw * cw - 1), px = (xl >> 1) + (k & (cw - 1)), py = (yl >> 1) + (k >> (lw - 1));
          s->dec[plane_off(c) + py * 32 + px] = cv.at(c, px, py);
        }
      }
      if (coeffs || (split_won && res_depth == 2)) {
        i16 *dst = coeff_level(3);
        const unsigned zy = zorder(xl, yl), zc = zorder(xl >> 1, yl >> 1);
        if (w == 16) {  // the challengers' levels never left LDS: the 16x16 CU's, or -- the split wins -- those of the 8x8 CUs that lie inside the picture
          const i16 *src = split_won ? s->lv3_coeff : s->lv2_coeff;
          for (int e = tid; e < 384; e += KVZ_CTU_THREADS) {
            const int q = e < 256 ? e >> 6 : ((e - 256) >> 4) & 3;
            if (split_won && (cx + xl + 8 * (q & 1) >= F.W || cy + yl + 8 * (q >> 1) >= F.H)) continue;  // never evaluated (search_d2): the output block keeps its zeros
            dst[e < 256 ? zy + e : plane_off(1 + ((e - 256) >> 6)) + zc + ((e - 256) & 63)] = src[e];
          }
        } else if (w == 32) {
          for (int e = tid; e < 1536; e += KVZ_CTU_THREADS) dst[e < 1024 ? zy + e : plane_off(1 + ((e - 1024) >> 8)) + zc + ((e - 1024) & 255)] = s->lv1_coeff[e];
        } else {
          const i16 *src = coeff_level(0);
          for (int e = tid; e < w * w; e += KVZ_CTU_THREADS) dst[zy + e] = src[zy + e];
          for (int e = tid; e < 2 * cw * cw; e += KVZ_CTU_THREADS) {
            const int c = e >= cw * cw, k = c ? e - cw * cw : e;
            dst[plane_off(1 + c) + zc + k] = src[plane_off(1 + c) + zc + k];
          }
        }
      }
    }
    KVZ_SYNC();
    KVZ_PROF(KVZ_P_COPY);
  }

  // search_cu at depth 2 or 3 (search.c:646-1063): searched CU.  Returns the cost through *out (LDS).
  // `first` / `last`: thread-0 bookkeeping of the caller folded into the first / last phase (see build_refs)
  template <class First, class Last>
  KVZ_DEV void eval_cu(int lv, int x, int y, int depth, double *out_cost, int *out_cbf, First first, Last last)
  {
    lane_rot = (lane_rot + 64) & (KVZ_CTU_THREADS - 1);
    const int log2w = 6 - depth, w = 1 << log2w, xl = x - cx, yl = y - cy;
    rough_search(lv, x, y, depth, first);
    const int mode = s->best_mode;
    (void)w;
    // kvz_intra_recon_cu luma, then chroma (search.c:807-827); chroma TUs are 4x4 for 8x8 CUs (transform.c:326).
    // All three planes' references were built by rough_search.
    TuSet t{ x, y, log2w, depth == 3 ? 2 : log2w - 1 };
    recon_tus(lv, t, depth, mode, true);
    if (cabac_on()) { KVZ_PROF_SYNC(KVZ_P_RECON); price_unit_coeffs(&s->cab, true, lv, depth, mode, &s->child_bits[0]); KVZ_PROF_SYNC(KVZ_P_COEFFBITS); }  // residual contexts; the syntax ones below are disjoint
    KVZ_FOR_THREADS(tid) {
      if (tid == 0) {
        // search.c:895-940: cabac->update = 1 around the mock encode and the transform tree's flags (= cu_bits() * lambda + leaf_rd_cost())
        *out_cost = cu_cost_batched(lv, x, y, depth, mode, s->preds, cabac_on() ? &s->child_bits[0] : nullptr);
        const CtuCu *cu = &s->cu[lv][(yl >> 3) * 8 + (xl >> 3)];
        *out_cbf = cbf_is_set(cu->cbf, depth, 0) || cbf_is_set(cu->cbf, depth, 1) || cbf_is_set(cu->cbf, depth, 2);
      }
      last(tid);
    }
    KVZ_SYNC();
    KVZ_PROF(KVZ_P_COST);
  }

  // ---------------------------------------------------------------- the NxN partition of an 8x8 CU (search_nxn)
  // SATD of mode `mode` on the 4x4 PU at CTU-local (xl, yl): prediction (intra.c:252-301 at width 4: unfiltered references, DC edge filter, the mode
  // 10 / 26 post-filter) minus source, 4x4 Hadamard, (sum + 1) >> 1 (picture-generic.c:117-208).  One lane, everything in registers.
  KVZ_DEV u32 pu_mode_satd(int mode, int xl, int yl) const
  {
    // One lane per mode, and what a lane does per sample is as little as the mode allows: the references it can touch come into registers once (an angular mode:
    // the thirteen samples ref_main[-4 .. 8] of intra-generic.c:82-104, packed into four words; a row of the block is then five consecutive bytes of them), the
    // horizontal modes run on the transposed problem (the sum of the Hadamard magnitudes of a block and of its transpose are the same).  The per-sample routine
    // (predict_pixel) decided the mode's class, its references and its projection again for each of the sixteen samples.
    const u8 *top = s->ref[0][0], *left = s->ref[0][1];
    u32 o[4];
    for (int r = 0; r < 4; r++) __builtin_memcpy(&o[r], org_at(0, xl, yl + r), 4);
    int p[4][4];
    if (mode == 0) {  // intra-generic.c:165-201
      const int tr = top[5], bl = left[5];
      int t4[4], l4[4];
      for (int i = 0; i < 4; i++) { t4[i] = top[i + 1]; l4[i] = left[i + 1]; }
#pragma unroll
      for (int y = 0; y < 4; y++)
#pragma unroll
        for (int x = 0; x < 4; x++) p[y][x] = ((3 - x) * l4[y] + (x + 1) * tr + (3 - y) * t4[x] + (y + 1) * bl + 4) >> 3;
    } else if (mode == 1) {  // intra-generic.c:210-241
      const int dc = s->dcval[0];
#pragma unroll
      for (int y = 0; y < 4; y++)
#pragma unroll
        for (int x = 0; x < 4; x++) p[y][x] = dc;
      p[0][0] = ((int)left[1] + 2 * dc + (int)top[1] + 2) >> 2;
#pragma unroll
      for (int i = 1; i < 4; i++) { p[0][i] = ((int)top[i + 1] + 3 * dc + 2) >> 2; p[i][0] = ((int)left[i + 1] + 3 * dc + 2) >> 2; }
    } else {  // intra-generic.c:49-155 (and the edge filter of the pure horizontal / vertical mode, intra.c:207-219)
      const bool vertical = mode >= 18;
      const u8 *main_ref = vertical ? top : left, *side_ref = vertical ? left : top;
      const int disp = s->mode_disp[mode], inv = s->mode_inv[mode];
      u32 m0, m1, m2, s0, s1;
      __builtin_memcpy(&m0, main_ref, 4); __builtin_memcpy(&m1, main_ref + 4, 4); __builtin_memcpy(&m2, main_ref + 8, 4);
      __builtin_memcpy(&s0, side_ref, 4); __builtin_memcpy(&s1, side_ref + 4, 4);
      // E[k] = ref_main[k - 4]: three projected samples (only reached by the modes whose projection stays inside the side reference), then main_ref[0 ..]
      const u32 e0 = side_ref[(128 + 3 * inv) >> 8], e1 = side_ref[(128 + 2 * inv) >> 8], e2 = side_ref[(128 + inv) >> 8];
      const u32 w0 = e0 | (e1 << 8) | (e2 << 16) | (m0 << 24), w1 = (m0 >> 8) | (m1 << 24), w2 = (m1 >> 8) | (m2 << 24), w3 = m2 >> 8;
      if (!vertical) {  // the transposed source
        u32 t[4];
#pragma unroll
        for (int r = 0; r < 4; r++) t[r] = ((o[0] >> (8 * r)) & 0xffu) | (((o[1] >> (8 * r)) & 0xffu) << 8) | (((o[2] >> (8 * r)) & 0xffu) << 16) | (((o[3] >> (8 * r)) & 0xffu) << 24);
#pragma unroll
        for (int r = 0; r < 4; r++) o[r] = t[r];
      }
      const unsigned long long side5 = ((unsigned long long)s1 << 32) | s0;
      const int side_0 = (int)(s0 & 0xffu);
#pragma unroll
      for (int py = 0; py < 4; py++) {
        const int dp = (py + 1) * disp, k0 = (dp >> 5) + 4, df = dp & 31, q = k0 >> 2;
        const u32 lo = q == 0 ? w0 : (q == 1 ? w1 : w2), hi = q == 0 ? w1 : (q == 1 ? w2 : w3);
        const unsigned long long win = (((unsigned long long)hi << 32) | lo) >> (8 * (k0 & 3));
        int bb[5];
#pragma unroll
        for (int i = 0; i < 5; i++) bb[i] = (int)((win >> (8 * i)) & 0xffu);
#pragma unroll
        for (int px = 0; px < 4; px++) p[py][px] = ((32 - df) * bb[px] + df * bb[px + 1] + 16) >> 5;
        if (disp == 0) p[py][0] = iclip(0, 255, p[py][0] + ((((int)((side5 >> (8 * (py + 1))) & 0xffu)) - side_0) >> 1));
      }
    }
    int t[4][4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int d0 = p[r][0] - (int)(o[r] & 0xffu), d1 = p[r][1] - (int)((o[r] >> 8) & 0xffu), d2 = p[r][2] - (int)((o[r] >> 16) & 0xffu), d3 = p[r][3] - (int)(o[r] >> 24);
      t[r][0] = d0 + d1 + d2 + d3; t[r][1] = d0 - d1 + d2 - d3; t[r][2] = d0 + d1 - d2 - d3; t[r][3] = d0 - d1 - d2 + d3;
    }
    int sum = 0;
#pragma unroll
    for (int c = 0; c < 4; c++) {
      const int pp = t[0][c], q = t[1][c], r = t[2][c], v = t[3][c];
      sum += iabs(pp + q + r + v) + iabs(pp - q + r - v) + iabs(pp + q - r - v) + iabs(pp - q - r + v);
    }
    return (u32)((sum + 1) >> 1);
  }
  // calc_mode_bits (search.c:557-581, updates on: search.c:906-913) + cu_rd_cost_tr_split_accurate at depth 4 (search.c:425-541) of the PU just
  // reconstructed; sums in s->acc, its most probable modes in s->preds.  One lane.
  KVZ_DEV double pu_cost(int j, int mode, const double *known_coeff_bits) const
  {
    CtxSet *c = &s->cab;
    double bits = luma_mode_bits(c, mode, s->preds, true);
    if (j == 0) bits += ctx_price(c, KVZ_CX_CHROMA, 0, true);  // the chroma mode is the CU's, coded once (with the PU at the CU's origin)
    double cost = bits * m->lambda;
    const int cb_y = s->acc[6] != 0, cb_u = j == 0 && s->acc[7] != 0, cb_v = j == 0 && s->acc[8] != 0;
    double tr_tree_bits = 0, coeff_bits = 0;
    // search.c:463-470 codes a chroma flag at this depth when the parent's is set, and reads the parent's from the same entry: cbf_is_set(cbf, depth - 1)
    // covers the bit the 4x4 block itself set.  So the flag is priced exactly when it is 1 -- on the depth-1 context (depth 4 - cu depth 3).
    if (cb_u) tr_tree_bits += ctx_price(c, KVZ_CX_CBF_CHROMA + 1, 1, true);
    if (cb_v) tr_tree_bits += ctx_price(c, KVZ_CX_CBF_CHROMA + 1, 1, true);
    tr_tree_bits += ctx_price(c, KVZ_CX_CBF_LUMA, cb_y, true);  // is_tr_split = depth - cu depth = 1 -> qt_cbf_model_luma[0]
    if (known_coeff_bits) coeff_bits += *known_coeff_bits;
    else {
      if (cb_y) coeff_bits += (double)s->acc[3] / 256.0;
      if (cb_u) coeff_bits += (double)s->acc[4] / 256.0;
      if (cb_v) coeff_bits += (double)s->acc[5] / 256.0;
    }
    const unsigned luma_ssd = s->acc[0], chroma_ssd = j == 0 ? s->acc[1] + s->acc[2] : 0u;
    const double tbits = tr_tree_bits + coeff_bits;
    cost += luma_ssd * 0.8 + chroma_ssd * 1.5 + tbits * m->lambda;
    return cost;
  }
  // Entry (k, i) of the 4x4 DST of intra luma (dct-generic.c:38-44, Tables::dst4) by arithmetic: the table lives in global memory, and a load from there in front of
  // every product is what a 4x4 PU's transform stages waited for
  KVZ_DEV static int dst4_at(int k, int i)
  {
    // { 29, 55, 74, 84 }, { 74, 74, 0, -74 }, { 84, -29, -74, 55 }, { 55, -84, 74, -29 }: one signed byte each, entry 0 lowest
    const unsigned long long lo = 0xb6004a4a544a371dull, hi = 0xe34aac3737b6e354ull;
    const int idx = 4 * k + i;
    return (int)(int8_t)(u8)((idx < 8 ? lo : hi) >> (8 * (idx & 7)));
  }
  // search_cu at depth 4 (search.c:646-1063 with depth > MAX_DEPTH: cu depth stays 3, search.c:691): PU j of the 8x8 CU at (rl->a3x, rl->a3y).
  // ONE pass per wavefront instead of the general stages of build_refs / recon_tus with a workgroup barrier each (a 4x4 block occupies sixteen lanes): the wavefront
  // playing threads 0..63 takes the luma block from its references to its reconstruction, the other one -- with the first PU -- the CU's 4x4 U and V blocks
  // (transform.c:306-312); the two meet when the luma mode is known (the chroma blocks are predicted with it), before the coefficient bits (whose luma part the two
  // share) and at the cost.  Same arithmetic, sample by sample, as the general stages.
  KVZ_DEV void eval_pu(int j)
  {
#if defined(KVZ_CTU_PROFILE) && !defined(KVZ_HOSTSIM)
    struct InPu { CtuShared *s; __device__ InPu(CtuShared *s_) : s(s_) { if (threadIdx.x == 0) s->prof_pu = 1; } __device__ ~InPu() { if (threadIdx.x == 0) s->prof_pu = 0; } } in_pu(s);
#endif
    lane_rot = (lane_rot + 64) & (KVZ_CTU_THREADS - 1);
    const int xl = rl->a3x + 4 * (j & 1), yl = rl->a3y + 4 * (j >> 1), x = cx + xl, y = cy + yl;
    const bool chroma = j == 0;
    const CandView cv = cand_view(4);
    const TuSet t{ x, y, 2, chroma ? 2 : 0 };
    // intra.c:47-82 in closed form (build_refs)
    const int ur = (y & 63) >> 2, uc = (x & 63) >> 2;
    const int gt = ur ? 2 * (ur & -ur) : 0, gl = uc ? (uc & -uc) : 0;
    const int avail_top = ur ? 4 * (gt - (uc & (gt - 1))) : 64, avail_left = uc ? 4 * (gl - (ur & (gl - 1))) : 64 - 4 * ur;
    // ---- references (unfiltered: a 4x4 block never reads the filtered ones, intra.c:252-301), the scalars of the search
    KVZ_FOR_THREADS(tid) {
      const u8 *base = (const u8 *)s;
      if (tid < 18) {
        const int side = tid >= 9, k = side ? tid - 9 : tid;
        int qx, qy;
        const bool have = ref_coords(2, 0, x, y, side, k, avail_top, avail_left, &qx, &qy);
        s->ref[0][side][k] = have ? base[rec_off(4, 0, qx, qy)] : (u8)128;
      }
      if (tid == 32) price_modes();
      if (tid == 33) {
        const int left = x >= 4 ? neighbour_cu(4, x - 1, y) : -1, above = (y >= 4 && yl > 0) ? neighbour_cu(4, x, y - 1) : -1;
        mpm_candidates(y, left, above, s->preds);
      }
      if (tid == 34) { s->acc[0] = 0; s->acc[3] = 0; s->acc[6] = 0; }
      if (tid >= 64 && tid < 70) s->acc[1 + (tid - 64) + (tid - 64) / 2] = 0;  // 1, 2, 4, 5, 7, 8
      if (chroma && tid >= 64 && tid < 82) {
        const int i = tid - 64, side = i >= 9, k = side ? i - 9 : i;
        int qx, qy, dv;
        const bool have = ref_coords(2, 1, x, y, side, k, avail_top, avail_left, &qx, &qy);
        const lds_off off = rec_off(4, 1, qx, qy, &dv);
        s->ref[1][side][k] = have ? base[off] : (u8)128;
        s->ref[2][side][k] = have ? base[off + dv] : (u8)128;
      }
    }
    KVZ_WAVE_SYNC();
    KVZ_FOR_THREADS(tid) {  // intra-generic.c:219-225: the DC value of each plane's references
      const int c = tid == 0 ? 0 : (tid == 64 ? 1 : (tid == 65 ? 2 : -1));
      if (c == 0 || (c > 0 && chroma)) s->dcval[c] = (u8)dc_value(2, s->ref[c][0], s->ref[c][1]);
    }
    KVZ_WAVE_SYNC();
    KVZ_PROF(KVZ_P_REFS);
    // ---- search_intra_rough at width 4: every mode's SATD, the reference's selection order on the table
    KVZ_FOR_THREADS(tid) { if (tid < 35) s->satd_raw[tid][0] = pu_mode_satd(tid, xl, yl); }
    KVZ_WAVE_SYNC();
    KVZ_FOR_THREADS(tid) {
      const int final_mode = replay_selection(tid, 2, 0);
      if (tid == 0) { s->best_mode = final_mode; rl->pu_mode[j] = (u8)final_mode; }
    }
    KVZ_SYNC();  // the chroma blocks are predicted with the luma mode
    KVZ_PROF(KVZ_P_SELECT);
    const int mode = s->best_mode;
    // a lane's role: threads 0..15 the luma samples, 64..79 / 80..95 the U / V samples
#define KVZ_PU_ROLE(tid) const int c = (tid) < 16 ? 0 : (((tid) >= 64 && (tid) < 96 && chroma) ? 1 + (((tid) - 64) >> 4) : -1), e = (tid) & 15, sh = c > 0 ? 1 : 0; (void)sh
    // ---- prediction -> candidate (as kvazaar blits it before quantising) and residual
    KVZ_FOR_THREADS(tid) {
      KVZ_PU_ROLE(tid);
      if (c >= 0) {
        const int px = e & 3, py = e >> 2;
        const u8 p = predict_pixel(2, mode, c, px, py);
        cv.at(c, (xl >> sh) + px, (yl >> sh) + py) = p;
        tbuf(t, 0, c)[e] = (i16)((int)*org_at(c, (xl >> sh) + px, (yl >> sh) + py) - (int)p);
      }
    }
    KVZ_WAVE_SYNC();
    KVZ_PROF(KVZ_P_RPRED);
    // ---- forward transform (dct-generic.c:559-568): the DST for luma (strategies-dct.c:82-86), the DCT for chroma
    for (int pass = 0; pass < 2; pass++) {
      KVZ_FOR_THREADS(tid) {
        KVZ_PU_ROLE(tid);
        if (c >= 0) {
          const int shift = pass == 0 ? 1 : 8, add = 1 << (shift - 1);
          const i16 *src = tbuf(t, pass, c);
          const int k = e >> 2, jj = e & 3;
          int a = 0;
#pragma unroll
          for (int i = 0; i < 4; i++) a += (c == 0 ? dst4_at(k, i) : dct_at(2, k, i)) * (int)src[(jj << 2) + i];
          tbuf(t, pass ^ 1, c)[e] = (i16)((a + add) >> shift);
        }
      }
      KVZ_WAVE_SYNC();
    }
    KVZ_PROF(KVZ_P_FDCT);
    // ---- kvz_rdoq (quant-generic.c:234-244), a block per wavefront: luma | U then V
    const bool rdoq = m->rdoq != 0;
    KVZ_FOR_THREADS(tid) {
#ifdef KVZ_HOSTSIM
      const int wv = tid == 0 ? 0 : (tid == 64 ? 1 : -1), lane = 0;
#else
      const int wv = tid >> 6, lane = tid & 63;
#endif
      for (int c = 0; c < 3; c++) {
        if (!rdoq || wv != (c ? 1 : 0) || (c && !chroma)) continue;
        RdoqWaveArgs ra;
        ra.ptab = (KVZ_LDS_PTR(const i32))rl->ptab; ra.coef = (KVZ_LDS_PTR(const i16))tbuf(t, 0, c); ra.dest = (KVZ_LDS_PTR(i16))levels_lds(4, c);
        ra.diag8 = (KVZ_LDS_PTR(const u8))rl->diag8; ra.lambda = m->lambda; ra.qp = m->qp; ra.log2w = 2; ra.type = c ? 2 : 0;
        ra.scan_mode = scan_order(mode, 4);
        ra.tr_depth = 2;  // cu->tr_depth - cu->depth + 1 for an NxN CU (quant-generic.c:237-238)
#if defined(KVZ_CTU_PROFILE) && !defined(KVZ_HOSTSIM)
        ra.prof = s->prof_rq;
        const unsigned long long tc0 = __builtin_amdgcn_s_memtime();
        rdoq_block_wave(ra, lane);
        if (c == 0 && lane == 0) ra.prof[7] += __builtin_amdgcn_s_memtime() - tc0;
#else
        rdoq_block_wave(ra, lane);
#endif
      }
    }
    KVZ_WAVE_SYNC();
    KVZ_PROF(KVZ_P_RDOQ);
    // ---- (without RDOQ: quantise;) cost sums of the levels; dequantise (quant-generic.c:335-339)
    KVZ_FOR_THREADS(tid) {
      KVZ_PU_ROLE(tid);
      u32 packed = 0;
      if (c >= 0) {
        const QuantScalars q = s->qs[0][c ? 1 : 0];
        int level;
        if (rdoq) level = levels_lds(4, c)[e];
        else {  // quant-generic.c:57-81
          const int cf = tbuf(t, 0, c)[e];
          level = (int)(((u32)iabs(cf) * (u32)q.flat_q + (u32)q.add) >> q.q_bits);
          if (cf < 0) level = -level;
          level = iclip(-32768, 32767, level);
          levels_lds(4, c)[e] = (i16)level;
        }
        int al = iabs(level);
        const u32 nz = al != 0;
        if (al > 3) al = 3;
        const u32 wsum = (u32)((m->coeff_weights >> (16 * al)) & 0xffff);
        tbuf(t, 1, c)[e] = (i16)iclip(-32768, 32767, (level * q.dq_scale + (1 << (q.dq_shift - 1))) >> q.dq_shift);
        packed = wsum | (nz << 24);
      }
      plane_add(&s->acc[3], packed, tid);  // weight sum (< 2^22) and count of levels in one word
    }
    KVZ_WAVE_SYNC();
    KVZ_PROF(KVZ_P_QUANT);
    // ---- inverse transform (dct-generic.c:570-579), only observable when the plane has levels; reconstruction (quant-generic.c:266-277) + SSD (search.c:500-505, 512-523)
    KVZ_FOR_THREADS(tid) {
      KVZ_PU_ROLE(tid);
      if (c >= 0 && (s->acc[3 + c] >> 24)) {
        const i16 *src = tbuf(t, 1, c);
        const int jj = e >> 2, i = e & 3;
        int a = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) a += (c == 0 ? dst4_at(k, i) : dct_at(2, k, i)) * (int)src[(k << 2) + jj];
        tbuf(t, 0, c)[e] = (i16)iclip(-32768, 32767, (a + 64) >> 7);
      }
    }
    KVZ_WAVE_SYNC();
    KVZ_PROF(KVZ_P_IDCT);
    KVZ_FOR_THREADS(tid) {
      KVZ_PU_ROLE(tid);
      u32 ssd = 0;
      if (c >= 0) {
        u8 *rp = &cv.at(c, (xl >> sh) + (e & 3), (yl >> sh) + (e >> 2));
        int v = *rp;
        if (s->acc[3 + c] >> 24) {
          const i16 *src = tbuf(t, 0, c);
          const int jj = e >> 2, i = e & 3;
          int a = 0;
#pragma unroll
          for (int k = 0; k < 4; k++) a += (c == 0 ? dst4_at(k, i) : dct_at(2, k, i)) * (int)src[(k << 2) + jj];
          const i16 res = (i16)iclip(-32768, 32767, (a + 2048) >> 12);
          v = iclip(0, 255, (int)(i16)(res + v));
          *rp = (u8)v;
        }
        const int d = (int)*org_at(c, (xl >> sh) + (e & 3), (yl >> sh) + (e >> 2)) - v;
        ssd = (u32)(d * d);
      }
      plane_add(&s->acc[0], ssd, tid);
    }
    KVZ_WAVE_SYNC();
    KVZ_FOR_THREADS(tid) {  // each wavefront unpacks its planes' sums; the coded-block flags wait in RdoqLds until the partition wins (nxn_attempt)
      if (tid == 0) {
        const u32 pk = s->acc[3];
        s->acc[3] = pk & 0xffffffu; s->acc[6] = pk >> 24;
        rl->pu_cbf[rl->n_pu] = (pk >> 24) != 0;
      }
      if (tid == 64 && chroma) {
        const u32 pu = s->acc[4], pv = s->acc[5];
        s->acc[4] = pu & 0xffffffu; s->acc[7] = pu >> 24; s->acc[5] = pv & 0xffffffu; s->acc[8] = pv >> 24;
        rl->pu_cbf_c[0] = (pu >> 24) != 0; rl->pu_cbf_c[1] = (pv >> 24) != 0;
      }
    }
    KVZ_SYNC();
    KVZ_PROF(KVZ_P_RECON);
#undef KVZ_PU_ROLE
    if (cabac_on()) { price_unit_coeffs(&s->cab, true, 4, 4, mode, &s->child_bits[0]); KVZ_PROF(KVZ_P_COEFFBITS); }
    KVZ_FOR_THREADS(tid) {
      if (tid == 0) {
        rl->split_cost3 += pu_cost(j, mode, cabac_on() ? &s->child_bits[0] : nullptr);
        rl->n_pu = j + 1;
      }
    }
    KVZ_SYNC();
    KVZ_PROF(KVZ_P_COST);
  }
  // The split alternative of an 8x8 CU (search.c:943-1063 at depth 3 with pu_depth_intra.max = 4).  On entry the CU has been evaluated as 2Nx2N at level 3
  // (cost in s->cost[3]; d3_last has saved the contexts after it, gone back to those at entry and priced part_size NxN into rl->split_cost3); on exit the cheaper
  // alternative is what level 3 holds and its cost has been added to the 16x16 CU's running split cost.
  KVZ_DEV void nxn_attempt(int x, int y)
  {
    const int xl = x - cx, yl = y - cy, cell = (yl >> 3) * 8 + (xl >> 3);
    for (int j = 0; j < 4; j++) {
      if (!(rl->split_cost3 < s->cost[3])) break;  // uniform: LDS scalars that only change in phases that end with a barrier
      eval_pu(j);
    }
    // every lane reads the verdict here; thread 0 does not touch its operands below (see search_d2)
    const bool nxn_wins = rl->split_cost3 < s->cost[3];
    const CandView c3v = cand_view(3), c4v = cand_view(4);
    KVZ_FOR_THREADS(tid) {
      if (nxn_wins) {  // work_tree_copy_up from level 4: pixels, levels, CU info
        if (tid < 96) {
          const int c = tid < 64 ? 0 : (tid < 80 ? 1 : 2), e = tid - (c == 0 ? 0 : (c == 1 ? 64 : 80)), sh = c ? 1 : 0, l2 = 3 - sh;
          const int px = (xl >> sh) + (e & ((1 << l2) - 1)), py = (yl >> sh) + (e >> l2);
          c3v.at(c, px, py) = c4v.at(c, px, py);
          levels_lds(3, c)[e] = rl->lv4_coeff[tid];
        }
        if (tid == 0) {
          CtuCu *cu = &s->cu[3][cell];
          cu->type = 1; cu->depth = 3; cu->tr_depth = 4; cu->mode = rl->pu_mode[0];
          cu->cbf = (uint16_t)(rl->pu_cbf[0] | (rl->pu_cbf[1] << 1) | (rl->pu_cbf[2] << 2) | (rl->pu_cbf[3] << 3) | (rl->pu_cbf_c[0] << 5) | (rl->pu_cbf_c[1] << 10));
          for (int k = 0; k < 4; k++) rl->mode4[3][cell][k] = rl->pu_mode[k];
        }
      } else ctx_copy_lanes(&s->cab, &rl->post3, tid);  // search.c:1051: the 2Nx2N CU stands; go on from the contexts after it
      if (tid == 0) {
        s->split_cost[2] += nxn_wins ? rl->split_cost3 : s->cost[3];
        rl->n_pu = -1;
      }
    }
    KVZ_SYNC();
  }

  // ---------------------------------------------------------------- CTU driver
  // Stages the source pixels of the 32x32 quadrant at (a1x, a1y), zero outside the picture (search.c:1084 FILL + :1151-1171).
  // One thread also runs `first` (bookkeeping of the caller that nobody reads before the barrier).
  template <class First = NoHook>
  KVZ_DEV void load_org(First first = First())
  {
    KVZ_FOR_THREADS(tid) {
      first(tid);
      for (int c = 0; c < 3; c++) {
        const int sh = c ? 1 : 0, l2 = 5 - sh, qw = 1 << l2, fw = F.W >> sh, fh = F.H >> sh, ox = (cx + a1x) >> sh, oy = (cy + a1y) >> sh;
        const u8 *src = frame_src(c);
        u8 *dst = s->org + (c == 0 ? 0 : (c == 1 ? 1024 : 1280));
        for (int e = tid; e < qw * qw; e += KVZ_CTU_THREADS) {
          const int px = ox + (e & (qw - 1)), py = oy + (e >> l2);
          dst[e] = (px < fw && py < fh) ? src[(long)py * fw + px] : 0;
        }
      }
    }
    KVZ_SYNC();
  }
  KVZ_DEV void init()
  {
    KVZ_FOR_THREADS(tid) {
      for (int e = tid; e < 6144; e += KVZ_CTU_THREADS) s->dec[e] = 0;
      for (int lv = 0; lv < 4; lv++)
        if (tid < 64) { CtuCu z = { 0, 0, 0, 0, 0 }; s->cu[lv][tid] = z; }
      // Coefficient buffers start zeroed like the lcu_t copies (search.c:1084).  Only observable for CTUs that stick out
      // of the picture: inside the picture every coefficient that reaches level 0 was written by a transform unit first.
      if (cx + 64 > F.W || cy + 64 > F.H)
        for (int lv = 0; lv < 4; lv += 3) { i16 *cf = coeff_level(lv); for (int e = tid; e < 6144; e += KVZ_CTU_THREADS) cf[e] = 0; }
#ifdef KVZ_HOSTSIM
      for (int e = tid; e < 1024; e += KVZ_CTU_THREADS) s->dct32[e] = tb->dct[3][e];
#else
      for (int e = tid; e < 80; e += KVZ_CTU_THREADS) s->dct_small[e] = e < 64 ? tb->dct[1][e] : tb->dct[0][e - 64];
#endif
      // neighbour CTUs (complete: they come earlier in the dependency order): border pixels and CU info from their records
      {
        const int ctx = cx >> 6, cty = cy >> 6;
        const u8 *base = F.border + (long)frame * F.wc * F.hc * KVZ_BORDER_BYTES;
        const u8 *r_left = ctx > 0 ? base + (long)(cty * F.wc + ctx - 1) * KVZ_BORDER_BYTES : nullptr;
        const u8 *r_top = cty > 0 ? base + (long)((cty - 1) * F.wc + ctx) * KVZ_BORDER_BYTES : nullptr;
        const u8 *r_tl = (ctx > 0 && cty > 0) ? base + (long)((cty - 1) * F.wc + ctx - 1) * KVZ_BORDER_BYTES : nullptr;
        const u8 *r_tr = (cty > 0 && ctx + 1 < F.wc) ? base + (long)((cty - 1) * F.wc + ctx + 1) * KVZ_BORDER_BYTES : nullptr;
        for (int c = 0; c < 3; c++) {
          const int lw = c ? 32 : 64, po = c == 0 ? 0 : (c == 1 ? 64 : 96);  // plane offset inside a 128-byte row / column record
          for (int i = tid; i < lw + 2; i += KVZ_CTU_THREADS) {  // left column incl. corner: index i <-> y = oy - 1 + i
            u8 v = 0;
            if (i == 0) { if (r_tl) v = load_shared_byte(r_tl + po + lw - 1); }
            else if (i <= lw) { if (r_left) v = load_shared_byte(r_left + 128 + po + i - 1); }
            s->bpx_left[c][i] = v;
          }
          for (int i = tid; i < lw + (lw >> 1) + 2; i += KVZ_CTU_THREADS) {  // top row incl. corner: index i <-> x = ox - 1 + i
            u8 v = 0;
            if (i == 0) { if (r_tl) v = load_shared_byte(r_tl + po + lw - 1); }
            else if (i <= lw) { if (r_top) v = load_shared_byte(r_top + po + i - 1); }
            else if (r_tr) v = load_shared_byte(r_tr + po + i - 1 - lw);
            s->bpx_top[c][i] = v;
          }
        }
        if (tid < 16) {
          const int side = tid >> 3, i = tid & 7;
          const u8 *r = side == 0 ? r_left : r_top;
          s->nb_depth[side][i] = r ? load_shared_byte(r + 256 + (side == 0 ? 16 : 0) + i) : 0;
          s->nb_mode[side][i] = r ? load_shared_byte(r + 256 + (side == 0 ? 24 : 8) + i) : 0;
        }
        if constexpr (NXN) {
          if (m->search_nxn) {
            if (tid < 16) rl->nb_mode4_left[tid] = r_left ? load_shared_byte(r_left + 448 + tid) : 0;
            for (int e = tid; e < 4 * 64; e += KVZ_CTU_THREADS) reinterpret_cast<u32 *>(rl->mode4)[e] = 0;
            if (tid == 0) rl->n_pu = -1;
          }
        }
      }
      for (int v = tid; v < 256; v += KVZ_CTU_THREADS) {
      if (v >= 64 && v < 64 + 35) {  // angular parameters (intra-generic.c:59-76)
        const int disp_tab[9] = { 0, 2, 5, 9, 13, 17, 21, 26, 32 }, inv_tab[9] = { 0, 4096, 1638, 910, 630, 482, 390, 315, 256 };
        const int mode = v - 64, md = mode >= 18 ? mode - 26 : 10 - mode, ad = iabs(md);
        s->mode_disp[mode] = (int8_t)(mode < 2 ? 0 : (md < 0 ? -disp_tab[ad] : disp_tab[ad]));
        s->mode_inv[mode] = (int16_t)(mode < 2 ? 0 : inv_tab[ad]);
      }
      if (v >= 128 && v < 136) s->qs[(v - 128) >> 1][v & 1] = quant_scalars_dev(2 + ((v - 128) >> 1), (v & 1) ? 2 : 0);
      if (v < 128) s->entropy_fbits[v] = m->entropy_fbits[v];
      if (v < 64) s->ctx_lps[v] = tb->ctx_next[1][2 * v];
      if (v < (cabac_on() ? KVZ_CX_COUNT : KVZ_CX_SYNTAX_COUNT)) {  // the residual contexts are only looked at with the CABAC coefficient cost
        // the row's contexts: from the CTU to the left; a row's first CTU from the second CTU of the row above (WPP; rows of a
        // one-CTU-wide picture and the first row start from the slice-start state, encoderstate.c:1218) -- or, without WPP, from
        // the last CTU of the row above (one coder runs through the picture in raster order)
        const int ctx = cx >> 6, cty = cy >> 6;
        const u8 *base = F.border + (long)frame * F.wc * F.hc * KVZ_BORDER_BYTES;
        const int seed = m->no_wpp ? F.wc - 1 : (F.wc > 1 ? 1 : -1);
        const u8 *r = ctx > 0 ? base + (long)(cty * F.wc + ctx - 1) * KVZ_BORDER_BYTES : ((cty > 0 && seed >= 0) ? base + (long)((cty - 1) * F.wc + seed) * KVZ_BORDER_BYTES : nullptr);
        const u8 st = (r && m->adaptive) ? load_shared_byte(r + 288 + v) : m->ctx_init[v];
        s->pre[0].s[v] = st; s->cab.s[v] = st;
      }
      }

    }
    KVZ_SYNC();
  }

  // copy_lcu_to_cu_data (search.c:1180-1207), pixel half: CtuShared::dec -> frame reconstruction and the pixel part of the
  // border record the right / lower neighbours read.  May run twice for a CTU (see run()): every lane rewrites the
  // addresses it wrote the first time, so the later values win.
  KVZ_DEV void write_rec()
  {
    const u8 *fin = s->dec;
    KVZ_FOR_THREADS(tid) {
      for (int c = 0; c < 3; c++) {
        const int sh = c ? 1 : 0, lw = 64 >> sh, fw = F.W >> sh, fh = F.H >> sh, ox = cx >> sh, oy = cy >> sh;
        u8 *dst = const_cast<u8 *>(frame_rec(c));
        for (int e = tid; e < lw * lw; e += KVZ_CTU_THREADS) {
          const int px = ox + e % lw, py = oy + e / lw;
          if (px < fw && py < fh) dst[(long)py * fw + px] = fin[plane_off(c) + e];
        }
      }
      u8 *r = F.border + ((long)frame * F.wc * F.hc + ctu_index()) * KVZ_BORDER_BYTES;
      for (int v = tid; v < 128; v += KVZ_CTU_THREADS) {
        const int c = v < 64 ? 0 : (v < 96 ? 1 : 2), i = v < 64 ? v : (v - 64) & 31, lw = c ? 32 : 64;
        r[v] = fin[plane_off(c) + (lw - 1) * lw + i];        // bottom row
        r[128 + v] = fin[plane_off(c) + i * lw + lw - 1];    // right column
      }
    }
    KVZ_SYNC();
  }
  // The syntax kvazaar writes for the finished CTU (kvz_encode_coding_tree, encode_coding_tree.c:745-940, with
  // encode_intra_coding_unit :467-652 and encode_transform_coeff :193-309), reduced to the bins that touch the ten contexts:
  // this is how the row's state->cabac moves from one CTU to the next.  Coefficient bins have contexts of their own that the
  // fast coefficient cost never prices with.  The recursion over the coding quadtree unrolled along the z-order of the 8x8
  // cells: at a cell, the nodes that start there and were not visited from an earlier cell are those of depth >= d0, d0 from
  // the cell's alignment.  One lane.
  KVZ_DEV void code_ctu_syntax(CtxSet *c) const
  {
    // the ten states travel in two registers (8 bits each): every bin then costs one table lookup instead of an LDS round trip
    // for the state as well -- this loop is serial, on one lane, and the row's next CTU waits for its result
    unsigned long long lo = 0, hi = 0;
    for (int k = 0; k < 8; k++) lo |= (unsigned long long)c->s[k] << (8 * k);
    hi = (unsigned long long)c->s[8] | ((unsigned long long)c->s[9] << 8);
    auto ctx_code = [&](int idx, int bin) {
      const int sh = (idx & 7) * 8;
      const int st = (int)(((idx < 8 ? lo : hi) >> sh) & 0xff);
      const unsigned long long nx = (unsigned long long)ctx_next(st, bin), keep = ~(0xffull << sh);
      if (idx < 8) lo = (lo & keep) | (nx << sh); else hi = (hi & keep) | (nx << sh);
    };
    int i = 0;
    while (i < 64) {
      const int xl = ((i & 1) | ((i >> 1) & 2) | ((i >> 2) & 4)) * 8, yl = (((i >> 1) & 1) | ((i >> 2) & 2) | ((i >> 3) & 4)) * 8, x = cx + xl, y = cy + yl;
      int d = i == 0 ? 0 : ((i & 15) == 0 ? 1 : ((i & 3) == 0 ? 2 : 3));
      if (x >= F.W || y >= F.H) { i += 1 << (2 * (3 - d)); continue; }  // the parent's border rule leaves this whole block out
      const CtuCu *cu = &s->cu[0][(yl >> 3) * 8 + (xl >> 3)];
      for (;; d++) {
        const int w = 64 >> d;
        if (d != 3) {
          const bool border = F.W < x + w || F.H < y + w;
          const bool split = cu->depth > d;  // GET_SPLITDATA
          if (!border) ctx_code(KVZ_CX_SPLIT + split_model(0, x, y, d), split);
          if (split || border) continue;
        }
        break;
      }
      const bool nxn = NXN && d == 3 && cu->tr_depth == 4;
      if (d == 3) ctx_code(KVZ_CX_PART, !nxn);  // part_mode at the minimum CU size: 2Nx2N = 1, NxN = 0
      if (nxn) {
        // encode_intra_coding_unit (encode_coding_tree.c:505-560): the prev_intra_luma_pred_flags of the four PUs first, each against the most probable modes at
        // its own position; then the chroma mode; the transform tree below codes cbf_cb / cbf_cr once (from the first PU's entry) and four luma flags on the
        // split-unit context (encode_coding_tree.c:148-163, 205-225)
        for (int j = 0; j < 4; j++) {
          const int px = x + 4 * (j & 1), py = y + 4 * (j >> 1), pm = rl->mode4[0][(yl >> 3) * 8 + (xl >> 3)][j];
          const int left = px > 0 ? neighbour_cu(0, px - 1, py) : -1, above = (py & 63) > 0 ? neighbour_cu(0, px, py - 1) : -1;
          int8_t preds[3];
          mpm_candidates(py, left, above, preds);
          ctx_code(KVZ_CX_INTRA, pm == preds[0] || pm == preds[1] || pm == preds[2]);
        }
        ctx_code(KVZ_CX_CHROMA, 0);
        ctx_code(KVZ_CX_CBF_CHROMA, (cu->cbf >> 5) & 1);
        ctx_code(KVZ_CX_CBF_CHROMA, (cu->cbf >> 10) & 1);
        for (int j = 0; j < 4; j++) ctx_code(KVZ_CX_CBF_LUMA, (cu->cbf >> j) & 1);
        i += 1;
        continue;
      }
      {
        const int left = x > 0 ? neighbour_cu(0, x - 1, y) : -1, above = (y & 63) > 0 ? neighbour_cu(0, x, y - 1) : -1;
        int8_t preds[3];
        mpm_candidates(y, left, above, preds);
        ctx_code(KVZ_CX_INTRA, cu->mode == preds[0] || cu->mode == preds[1] || cu->mode == preds[2]);  // prev_intra_luma_pred_flag; mpm_idx / rem mode are bypass
        ctx_code(KVZ_CX_CHROMA, 0);  // intra_chroma_pred_mode: derived from luma
      }
      // transform tree: split_transform_flag is never coded (tr_depth_intra = 0; the 64x64 split is inferred, encode_coding_tree.c:236-243)
      const int cb_u = cbf_is_set(cu->cbf, d, 1), cb_v = cbf_is_set(cu->cbf, d, 2);
      ctx_code(KVZ_CX_CBF_CHROMA, cb_u);
      ctx_code(KVZ_CX_CBF_CHROMA, cb_v);
      if (d == 0) {
        for (int q = 0; q < 4; q++) {
          const CtuCu *t = &s->cu[0][(q >> 1) * 32 + (q & 1) * 4];
          if (cb_u) ctx_code(KVZ_CX_CBF_CHROMA + 1, cbf_is_set(t->cbf, 1, 1));
          if (cb_v) ctx_code(KVZ_CX_CBF_CHROMA + 1, cbf_is_set(t->cbf, 1, 2));
          ctx_code(KVZ_CX_CBF_LUMA, cbf_is_set(t->cbf, 1, 0));
        }
      } else ctx_code(KVZ_CX_CBF_LUMA + 1, cbf_is_set(cu->cbf, d, 0));  // always present for intra (encode_coding_tree.c:276-279)
      i += 1 << (2 * (3 - d));
    }
    for (int k = 0; k < 8; k++) c->s[k] = (u8)(lo >> (8 * k));
    c->s[8] = (u8)hi; c->s[9] = (u8)(hi >> 8);
  }
  // ... and its residual part (encode_transform_unit, encode_coding_tree.c:117-190), only when coefficients are priced with the
  // CABAC model: the residual contexts of the row move with the levels of every coded block, transform units in z-order, Y U V
  // inside a unit.  The two sets of contexts are disjoint, so the order between this and code_ctu_syntax() is free.  The final
  // levels come back from the CTU's output block one 32x32 quadrant at a time, staged in lv1_coeff by all lanes.
  KVZ_DEV void code_ctu_residual()
  {
    const i16 *fin = coeff_level(3);
    for (int q = 0; q < 4; q++) {
      const int qxl = (q & 1) * 32, qyl = (q >> 1) * 32;
      if (cx + qxl >= F.W || cy + qyl >= F.H) continue;
      KVZ_FOR_THREADS(tid) {
        for (int e = tid; e < 1536; e += KVZ_CTU_THREADS)
          s->lv1_coeff[e] = fin[e < 1024 ? q * 1024 + e : plane_off(1 + ((e - 1024) >> 8)) + q * 256 + ((e - 1024) & 255)];
      }
      KVZ_SYNC();
      KVZ_FOR_THREADS(tid) {
#ifdef KVZ_HOSTSIM
        if (tid == 0) {
#else
        // independent chains on the two wavefronts: the significance flags of the luma blocks on one, the chroma blocks and the luma blocks' other
        // classes on the other (coeff_cabac_bits_wave's `part`)
        const bool luma_role = tid < 64, chroma_role = !luma_role;
        {  // both wavefronts walk the quadrant's units (everything they branch on is wavefront-uniform)
#endif
          CtxSet *c = &s->pre[0];
          int i = 16 * q;
          while (i < 16 * q + 16) {
            const int xl = ((i & 1) | ((i >> 1) & 2) | ((i >> 2) & 4)) * 8, yl = (((i >> 1) & 1) | ((i >> 2) & 2) | ((i >> 3) & 4)) * 8;
            if (cx + xl >= F.W || cy + yl >= F.H) { i++; continue; }
            const CtuCu *cu = &s->cu[0][(yl >> 3) * 8 + (xl >> 3)];
            if (NXN && cu->tr_depth == 4) {
              // four 4x4 luma blocks, each scanned by its own PU's mode, then the 4x4 chroma blocks under the first PU's (encode_coding_tree.c:148-163)
              const u8 *pm = rl->mode4[0][(yl >> 3) * 8 + (xl >> 3)];
              const i16 *y4 = s->lv1_coeff + (zorder(xl, yl) - q * 1024), *u4 = s->lv1_coeff + 1024 + (zorder(xl >> 1, yl >> 1) - q * 256);
              const int cscan = scan_order(pm[0], 4);
#ifdef KVZ_HOSTSIM
              for (int j = 0; j < 4; j++) if ((cu->cbf >> j) & 1) coeff_cabac_bits(c, true, y4 + 16 * j, 2, 0, scan_order(pm[j], 4));
              if ((cu->cbf >> 5) & 1) coeff_cabac_bits(c, true, u4, 2, 2, cscan);
              if ((cu->cbf >> 10) & 1) coeff_cabac_bits(c, true, u4 + 256, 2, 2, cscan);
#else
              for (int j = 0; j < 4; j++) if (luma_role && ((cu->cbf >> j) & 1)) coeff_cabac_bits_wave(c, true, y4 + 16 * j, 2, 0, scan_order(pm[j], 4), 1);
              if (chroma_role && ((cu->cbf >> 5) & 1)) coeff_cabac_bits_wave(c, true, u4, 2, 2, cscan);
              if (chroma_role && ((cu->cbf >> 10) & 1)) coeff_cabac_bits_wave(c, true, u4 + 256, 2, 2, cscan);
              for (int j = 0; j < 4; j++) if (chroma_role && ((cu->cbf >> j) & 1)) coeff_cabac_bits_wave(c, true, y4 + 16 * j, 2, 0, scan_order(pm[j], 4), 2);
#endif
              i += 1;
              continue;
            }
            const int td = cu->depth < 1 ? 1 : cu->depth;  // a 64x64 CU codes four 32x32 units, each read at its own origin
            const int lw = 6 - td, lc = td == 3 ? 2 : lw - 1, scan = scan_order(cu->mode, td);
            const i16 *y = s->lv1_coeff + (zorder(xl, yl) - q * 1024), *u = s->lv1_coeff + 1024 + (zorder(xl >> 1, yl >> 1) - q * 256);
#ifdef KVZ_HOSTSIM
            if (cbf_is_set(cu->cbf, td, 0)) coeff_cabac_bits(c, true, y, lw, 0, scan);
            if (cbf_is_set(cu->cbf, td, 1)) coeff_cabac_bits(c, true, u, lc, 2, scan);
            if (cbf_is_set(cu->cbf, td, 2)) coeff_cabac_bits(c, true, u + 256, lc, 2, scan);
#else
            if (luma_role && cbf_is_set(cu->cbf, td, 0)) coeff_cabac_bits_wave(c, true, y, lw, 0, scan, 1);
            if (chroma_role && cbf_is_set(cu->cbf, td, 1)) coeff_cabac_bits_wave(c, true, u, lc, 2, scan);
            if (chroma_role && cbf_is_set(cu->cbf, td, 2)) coeff_cabac_bits_wave(c, true, u + 256, lc, 2, scan);
            if (chroma_role && cbf_is_set(cu->cbf, td, 0)) coeff_cabac_bits_wave(c, true, y, lw, 0, scan, 2);
#endif
            i += 1 << (2 * (3 - td));
          }
        }
      }
      KVZ_SYNC();
    }
  }
  // ... and the CU-info half: level 0 -> cu arrays, CTU cost, CU part of the border record; coefficients already there
  KVZ_DEV void finish_info()
  {
    KVZ_FOR_THREADS(tid) {
      if (tid < 64) {
        const int fx = cx + (tid & 7) * 8, fy = cy + (tid >> 3) * 8;
        if (fx < F.W && fy < F.H) {
          const long i = (long)frame * (F.H >> 3) * (F.W >> 3) + (long)(fy >> 3) * (F.W >> 3) + (fx >> 3);
          F.cu_depth[i] = s->cu[0][tid].depth;
          F.cu_mode[i] = s->cu[0][tid].mode;
        }
      }
      u8 *r = F.border + ((long)frame * F.wc * F.hc + ctu_index()) * KVZ_BORDER_BYTES;
      if (tid == 0) {
        F.ctu_cost[(long)frame * F.wc * F.hc + ctu_index()] = s->res[0];
        if (m->adaptive) code_ctu_syntax(&s->pre[0]);  // moves the syntax contexts only
        for (int i = 0; i < KVZ_CX_SYNTAX_COUNT; i++) r[288 + i] = s->pre[0].s[i];
      }
      if (cabac_on())
        for (int v = KVZ_CX_SYNTAX_COUNT + tid; v < KVZ_CX_COUNT; v += KVZ_CTU_THREADS) r[288 + v] = s->pre[0].s[v];  // residual contexts: final since code_ctu_residual()
      for (int v = tid; v < 32; v += KVZ_CTU_THREADS) {
        const int i = v & 7;
        const CtuCu *cu = &s->cu[0][v < 16 ? 56 + i : i * 8 + 7];
        r[256 + v] = ((v >> 3) & 1) ? cu->mode : cu->depth;
      }
      if constexpr (NXN) {
        if (m->search_nxn) {
          if (tid < 16) r[448 + tid] = rl->mode4[0][(tid >> 1) * 8 + 7][(tid & 1) * 2 + 1];  // right column, 4x4 unit `tid` from the top
          if (tid < 64) {
            const int fx = cx + (tid & 7) * 8, fy = cy + (tid >> 3) * 8;
            if (fx < F.W && fy < F.H) {
              if (F.cu_part) F.cu_part[(long)frame * (F.H >> 3) * (F.W >> 3) + (long)(fy >> 3) * (F.W >> 3) + (fx >> 3)] = s->cu[0][tid].tr_depth == 4;
              if (F.cu_mode4) {
                u8 *m4 = F.cu_mode4 + (long)frame * (F.H >> 2) * (F.W >> 2) + (long)(fy >> 2) * (F.W >> 2) + (fx >> 2);
                m4[0] = rl->mode4[0][tid][0]; m4[1] = rl->mode4[0][tid][1]; m4[F.W >> 2] = rl->mode4[0][tid][2]; m4[(F.W >> 2) + 1] = rl->mode4[0][tid][3];
              }
            }
          }
        }
      }
    }
    KVZ_SYNC();
  }

  KVZ_DEV void cu_header(int lv, int xl, int yl, int depth) const  // search.c:694-700, one thread
  {
    CtuCu *c = &s->cu[lv][(yl >> 3) * 8 + (xl >> 3)];
    c->depth = (u8)(depth > 3 ? 3 : depth); c->tr_depth = (u8)(depth > 0 ? depth : 1); c->type = 0;
  }
  KVZ_DEV void set_cu_header(int lv, int xl, int yl, int depth)
  {
    KVZ_FOR_THREADS(tid) { if (tid == 0) cu_header(lv, xl, yl, depth); }
    KVZ_SYNC();
  }

  // The combine_intra_cus attempt of search.c:996-1044 at depth 0 or 1: reconstruct the whole CU with the mode of its
  // top-left child and price it.  Result in s->cost[depth].
  KVZ_DEV void try_merge(int x, int y, int depth)
  {
    const int w = 64 >> depth, xl = x - cx, yl = y - cy, lv = depth;
    const CtuCu d1 = s->cu[depth + 1][(yl >> 3) * 8 + (xl >> 3)];  // uniform read
    if (!(d1.type == 1 && d1.depth == depth + 1)) return;
    const int mode = d1.mode;
    fill_cu(lv, xl, yl, w, 1, depth, mode, depth > 0 ? depth : 1);
    if (depth == 1) {
      TuSet t{ x, y, 5, 4 };
      recon_tus(lv, t, 1, mode);
      if (cabac_on()) price_unit_coeffs(&s->pre[1], false, 1, 1, mode, &s->child_bits[0]);
      KVZ_FOR_THREADS(tid) {
        if (tid == 0) {
          // search.c:1005-1041: priced from the contexts at entry; pre_search_cabac carries update == 0 (it was copied while the
          // caller had updates off), so nothing here moves a context
          CtxSet *pc = &s->pre[1];
          double bits = 0;
          bits += ctx_price(pc, KVZ_CX_SPLIT + split_model(lv, x, y, depth), 0, false);
          const double mode_bits = intra_mode_syntax_bits(pc, false, lv, x, y, mode, false) + bits;  // calc_mode_bits search.c:557-581
          double cost = 0;
          cost += mode_bits * m->lambda;
          cost += leaf_rd_cost(pc, false, lv, xl, yl, 1, 1, true, true, cabac_on() ? &s->child_bits[0] : nullptr);
          s->cost[1] = cost;
        }
      }
      KVZ_SYNC();
      return;
    }
    // depth 0: four 32x32 transform units in z-order (intra.c:656-679), each predicted from the previous ones
    for (int q = 0; q < 4; q++) {
      const int qx = x + (q & 1) * 32, qy = y + (q >> 1) * 32;
      TuSet t{ qx, qy, 5, 4 };
      a1x = qx - cx; a1y = qy - cy;
      load_org();
      recon_tus(lv, t, 1, mode);
      // priced from the contexts at entry with updates off (search.c:1005-1041), so each unit's bits stand alone
      if (cabac_on()) price_unit_coeffs(&s->pre[0], false, 0, 1, mode, &s->child_bits[q]);
      KVZ_FOR_THREADS(tid) { if (tid < 9) s->child_acc[q][tid] = s->acc[tid]; }
      KVZ_SYNC();
    }
    KVZ_FOR_THREADS(tid) {
      if (tid == 0) {
        CtuCu *cu = &s->cu[lv][0];
        // cbf_set_conditionally (intra.c:681-696): depth-0 bit if any of the other three children has a depth-1 bit
        for (int c = 0; c < 3; c++) {
          cbf_clear(&cu->cbf, 0, c);  // intra.c:641-647 cleared levels >= 0 before the recursion; the TL child then set its depth-1 bit
        }
        // re-apply the TL child's own depth-1 bits (they live in the same entry and were set during q = 0)
        for (int c = 0; c < 3; c++) if (s->child_acc[0][6 + c]) cbf_set(&cu->cbf, 1, c);
        for (int c = 0; c < 3; c++) {
          const bool any_other = cbf_is_set(s->cu[lv][4].cbf, 1, c) || cbf_is_set(s->cu[lv][32].cbf, 1, c) || cbf_is_set(s->cu[lv][36].cbf, 1, c);
          if (any_other) cbf_set(&cu->cbf, 0, c);
        }
        // cu_rd_cost_tr_split_accurate at depth 0 (search.c:425-541): chroma cbf at depth 0, then the four children
        CtxSet *pc = &s->pre[0];
        double tr_tree_bits = 0;
        tr_tree_bits += ctx_price(pc, KVZ_CX_CBF_CHROMA, cbf_is_set(cu->cbf, 0, 1), false);
        tr_tree_bits += ctx_price(pc, KVZ_CX_CBF_CHROMA, cbf_is_set(cu->cbf, 0, 2), false);
        double sum = 0;
        for (int q = 0; q < 4; q++) {
          const int qxl = (q & 1) * 32, qyl = (q >> 1) * 32;
          const CtuCu *tr_cu = &s->cu[lv][(qyl >> 3) * 8 + (qxl >> 3)];
          for (int i = 0; i < 9; i++) s->acc[i] = s->child_acc[q][i];
          // search.c:466-471: child cbf_cb/cbf_cr are coded when the entry has any chroma bit at depth >= 0
          sum += leaf_rd_cost(pc, false, lv, qxl, qyl, 1, 0, cbf_is_set(tr_cu->cbf, 0, 1), cbf_is_set(tr_cu->cbf, 0, 2), cabac_on() ? &s->child_bits[q] : nullptr);
        }
        const double rd = sum + tr_tree_bits * m->lambda;
        double bits = 0;
        bits += ctx_price(pc, KVZ_CX_SPLIT + split_model(lv, x, y, 0), 0, false);
        const double mode_bits = intra_mode_syntax_bits(pc, false, lv, x, y, mode, false) + bits;
        double cost = 0;
        cost += mode_bits * m->lambda;
        cost += rd;
        s->cost[0] = cost;
      }
    }
    KVZ_SYNC();
  }

  // search_cu for depth 2 (16x16) including its depth-3 children (search.c:646-1063)
  KVZ_DEV void search_d2(int x, int y)
  {
    const int xl = x - cx, yl = y - cy;
    if (x >= F.W || y >= F.H) return;  // search_cu returns 0 outside the picture: nothing to add to the parent's sum
    a2x = xl; a2y = yl;
    const bool inside = x + 16 <= F.W && y + 16 <= F.H;
    // thread-0 bookkeeping around the 16x16 CU: header + cost initialisation before, split cost after
    auto d2_first = [&](int tid) {
      if (tid == kHookThread) { cu_header(2, xl, yl, 2); s->cost[2] = 1.7e+308; s->cbf_any = 0; price_modes(); }
      ctx_copy_lanes(&s->pre[2], &s->cab, tid);
    };
    auto d2_last = [&](int tid) {
      ctx_copy_lanes(&s->post2, &s->cab, tid);  // search.c:956-959: the split alternative starts again from the contexts at entry
      ctx_copy_lanes(&s->cab, &s->pre[2], tid);
      if (tid == 0) {  // prices the split flag on the contexts at entry: after the copy above (same wavefront, program order)
        double sc = split_flag_cost(2, x, y, 2);
        if (inside && !s->cbf_any) sc = 2147483647;  // cu_split_termination = zero (search.c:975-984)
        s->split_cost[2] = sc;
      }
    };
    if (inside) eval_cu(2, x, y, 2, &s->cost[2], &s->cbf_any, d2_first, d2_last);
    else {
      KVZ_FOR_THREADS(tid) { d2_first(tid); d2_last(tid); }
      KVZ_SYNC();
    }
    if (!inside || s->cbf_any) {
      for (int q = 0; q < 4; q++) {
        if (!(s->split_cost[2] < s->cost[2])) break;  // uniform: both are LDS scalars
        const int qx = x + (q & 1) * 8, qy = y + (q >> 1) * 8;
        if (qx >= F.W || qy >= F.H) continue;  // child outside the picture costs 0
        a3q = q;
        if (nxn_on()) {
          if constexpr (NXN) {
            // search_cu at depth 3 with pu_depth_intra.max = 4: the 8x8 CU as 2Nx2N, then -- if it has coefficients (cu-split-termination zero, search.c:975-984)
            // -- as four 4x4 PUs starting again from the contexts at entry with part_size NxN priced (search.c:956-974)
            eval_cu(3, qx, qy, 3, &s->cost[3], &s->cbf_any,
                    [&](int tid) {
                      if (tid == 0) { cu_header(3, qx - cx, qy - cy, 3); price_modes(); rl->a3x = qx - cx; rl->a3y = qy - cy; rl->n_pu = 0; }
                      ctx_copy_lanes(&rl->pre3, &s->cab, tid);
                    },
                    [&](int tid) {
                      ctx_copy_lanes(&rl->post3, &s->cab, tid);
                      ctx_copy_lanes(&s->cab, &rl->pre3, tid);
                      if (tid == 0) {  // after the copy above (same wavefront, program order)
                        double sb = 0;
                        sb += ctx_price(&s->cab, KVZ_CX_PART, 0, true);
                        double sc = 0.0;
                        sc += sb * m->lambda;
                        if (!s->cbf_any) sc = 2147483647;
                        rl->split_cost3 = sc;
                      }
                    });
            nxn_attempt(qx, qy);
          }
        } else
        eval_cu(3, qx, qy, 3, &s->cost[3], &s->cbf_any, [&](int tid) { if (tid == kHookThread) { cu_header(3, qx - cx, qy - cy, 3); price_modes(); } }, [&](int tid) { if (tid == 0) s->split_cost[2] += s->cost[3]; });
      }
    }
    // Every lane reads the verdict here; thread 0 only touches its operands again after the barrier that ends commit()
    // (the result goes to res[2]).  Without that a lagging wavefront could read an updated cost and take the other branch
    // (the host simulation cannot show such a race).
    const bool split_wins2 = s->split_cost[2] < s->cost[2];
    if (split_wins2) {
      commit(3, 2, 2, -1, false, xl, yl, 16, 2, true);  // work_tree_copy_up: the 8x8 CUs win (their pixels and coefficients are already in place)
    } else {
      commit(2, 3, 3, 2, true, xl, yl, 16, 2, false);   // work_tree_copy_down: the 16x16 CU wins
    }
  }

  // search_cu for depth 1 when 32x32 CUs are searched (--pu-depth-intra 1-3; search.c:646-1063 one level above search_d2): the 32x32 CU is
  // evaluated first, its four 16x16 children only if it has coefficients (cu-split-termination zero) and while the running split cost stays
  // below its cost.  Differences to the merge flow of the other presets: the 32x32 candidate exists BEFORE the children are searched, and the
  // children's search buffers share storage with its pixels (CtuShared's union) -- so the candidate's 1 536 bytes wait in the CTU's
  // scratch block in HBM (free until the 64x64 attempt at the very end) and come back if the 32x32 CU wins; its levels stay in lv1_coeff,
  // which the children do not use.  The contexts after the 32x32 CU (search.c:956-959 post_search_cabac) are kept by SWAPPING them with the
  // entry state in pre[1]: nothing reads this depth's entry state again, and commit() restores from pre[1] when the CU stays unsplit.
  KVZ_DEV void search_d1(int x1, int y1)
  {
    const int xl = x1 - cx, yl = y1 - cy;
    const bool inside = x1 + 32 <= F.W && y1 + 32 <= F.H;
    int *cbf1 = reinterpret_cast<int *>(&s->child_acc[0][0]);  // "the 32x32 CU has coefficients": a scalar of its own (cbf_any is reused by the children); child_acc is dead until the 64x64 attempt
    auto d1_first = [&](int tid) { if (tid == 0) { *cbf1 = 0; price_modes(); } };
    auto d1_last = [&](int tid) {
      ctx_swap_lanes(&s->cab, &s->pre[1], tid);
      if (tid == 0) {
        double sc = split_flag_cost(1, x1, y1, 1);
        if (inside && !*cbf1) sc = 2147483647;  // search.c:975-984
        s->split_cost[1] = sc;
      }
    };
    if (inside) eval_cu(1, x1, y1, 1, &s->cost[1], cbf1, d1_first, d1_last);
    else {
      KVZ_FOR_THREADS(tid) { d1_first(tid); d1_last(tid); }
      KVZ_SYNC();
    }
    const bool descend = !inside || *cbf1;
    if (!descend) return;
    u8 *stash = reinterpret_cast<u8 *>(coeff_level(0));
    if (inside) {
      KVZ_FOR_THREADS(tid) { for (int e = tid; e < 1536 / 4; e += KVZ_CTU_THREADS) reinterpret_cast<u32 *>(stash)[e] = reinterpret_cast<const u32 *>(s->c1)[e]; }
      KVZ_SYNC();
    }
    for (int q2 = 0; q2 < 4; q2++) {
      if (!(s->split_cost[1] < s->cost[1])) break;  // uniform: LDS scalars that only change inside commit(), which ends with a barrier
      search_d2(x1 + (q2 & 1) * 16, y1 + (q2 >> 1) * 16);
    }
    if (inside && !(s->split_cost[1] < s->cost[1])) {  // the 32x32 CU wins: its pixels back into the candidate buffer for commit()
      KVZ_SYNC();
      KVZ_FOR_THREADS(tid) { for (int e = tid; e < 1536 / 4; e += KVZ_CTU_THREADS) reinterpret_cast<u32 *>(s->c1)[e] = reinterpret_cast<const u32 *>(stash)[e]; }
      KVZ_SYNC();
    }
  }

  KVZ_DEV void run()
  {
#if defined(KVZ_CTU_PROFILE) && !defined(KVZ_HOSTSIM)
    if (threadIdx.x == 0) { for (int i = 0; i < 64; i++) s->prof_acc[i] = 0; for (int i = 0; i < 16; i++) s->prof_rq[i] = 0; s->prof_pu = 0; }
    t_last = __builtin_amdgcn_s_memtime();
#endif
    init();
    if constexpr (RDOQ) {  // kvz_rdoq prices on state->cabac's contexts as they stand now (pre[0]): both bins of every context, once per CTU
      KVZ_FOR_THREADS(tid) {
        if (m->rdoq) for (int v = tid; v < 2 * KVZ_CX_COUNT; v += KVZ_CTU_THREADS) rl->ptab[v] = (i32)(s->entropy_fbits[s->pre[0].s[v >> 1] ^ (v & 1)] * 32768.0f);
        if (tid < 64) rl->diag8[tid] = tb->diag8[tid];
      }
      KVZ_SYNC();
    }
    KVZ_PROF(KVZ_P_INIT);
    KVZ_FOR_THREADS(tid) {
      if (tid == 0) { cu_header(0, 0, 0, 0); s->cost[0] = 1.7e+308; s->split_cost[0] = split_flag_cost(0, cx, cy, 0); }
    }
    KVZ_SYNC();
    for (int q1 = 0; q1 < 4; q1++) {
      const int x1 = cx + (q1 & 1) * 32, y1 = cy + (q1 >> 1) * 32;
      if (x1 >= F.W || y1 >= F.H) continue;  // search_cu returns 0 outside the picture
      a1x = x1 - cx; a1y = y1 - cy;
      const bool search32 = S32 && m->search_32x32;
      load_org([&](int tid) {
        ctx_copy_lanes(&s->pre[1], &s->cab, tid);  // before the split flag below is priced (it moves its context)
        if (tid == 0) { cu_header(1, x1 - cx, y1 - cy, 1); s->cost[1] = 1.7e+308; if (!search32) s->split_cost[1] = split_flag_cost(1, x1, y1, 1); }
      });
      if (search32) search_d1(x1, y1);
      else {
        for (int q2 = 0; q2 < 4; q2++) search_d2(x1 + (q2 & 1) * 16, y1 + (q2 >> 1) * 16);
        if (x1 + 32 <= F.W && y1 + 32 <= F.H) try_merge(x1, y1, 1);
      }
      const bool split_wins1 = s->split_cost[1] < s->cost[1];  // operands stay put until the barrier that ends commit()
      if (split_wins1) commit(2, 1, 1, -1, false, x1 - cx, y1 - cy, 32, 1, true);
      else commit(1, 2, 3, 1, true, x1 - cx, y1 - cy, 32, 1, false);  // the 32x32 merge wins
    }
    // The 64x64 merge builds its candidate in the storage of the decided picture, so the split result goes out first
    bool attempt0 = false;
    if (cx + 64 <= F.W && cy + 64 <= F.H) { const CtuCu d1 = s->cu[1][0]; attempt0 = d1.type == 1 && d1.depth == 1; }  // try_merge's own test
    if (attempt0) { write_rec(); try_merge(cx, cy, 0); }
    const bool split_wins0 = s->split_cost[0] < s->cost[0];
    if (split_wins0) {
      commit(1, 0, 0, -1, false, 0, 0, 64, 0, true);
      if (!attempt0) write_rec();
    } else {  // the 64x64 merge wins: its pixels replace the ones written above, its coefficients move to the output block
      commit(0, 0, -1, -1, true, 0, 0, 64, 0, false);
      write_rec();
    }
    KVZ_PROF(KVZ_P_MISC);
    if (m->adaptive && cabac_on()) { code_ctu_residual(); KVZ_PROF_SYNC(KVZ_P_PRED35); }  // category reused: the residual replay
    finish_info();
    KVZ_PROF(KVZ_P_FINISH);
#if defined(KVZ_CTU_PROFILE) && !defined(KVZ_HOSTSIM)
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int i = 0; i < 2 * KVZ_P_COUNT; i++) { atomicAdd(&F.prof[i], s->prof_acc[i]); atomicAdd(&F.prof[KVZ_PROF_PU_AT + i], s->prof_acc[32 + i]); }
      for (int i = 0; i < 16; i++) atomicAdd(&F.prof[2 * KVZ_P_COUNT + i], s->prof_rq[i]);
    }
#endif
  }
};
using CtuProgram = CtuProgramT<true>;

}  // namespace kvz
