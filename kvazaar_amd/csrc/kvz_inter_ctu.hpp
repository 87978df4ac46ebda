// kvz_inter_ctu.hpp -- the CTU pass of pictures with inter prediction (BASELINE config 4: `--preset veryfast --gop lp-g4d3t1`, B slices whose two lists hold the
// previous picture; `ultrafast` .. `faster` with that GOP): one workgroup searches and reconstructs one 64x64 CTU -- search_cu of a P / B slice (search.c:646-1063) with everything below it:
// kvz_search_cu_inter (search_inter.c:2202: merge candidates, early skip, the motion search of kvz_me.hpp's shape, half-pel refinement), the intra alternative
// (search_intra.c:812, rd 0), motion compensation (inter.c:374-660), the transform tree of either CU type, zero-coefficient RDO, kvz_mock_encode_coding_unit and
// cu_rd_cost_tr_split_accurate on CABAC contexts that live as the encoder's do, the recursion with its work-tree copies, and the finished CTU's syntax for the
// next CTU's contexts.  oracle/kvz_oracle_inter.inc is the function-by-function CPU restatement it is checked against (itself equal to the reference encoder CU
// for CU, tests/test_inter_oracle.py); function names below are the oracle's, which cites the reference lines.
//
// One wavefront per CTU.  The decisions are the reference's, taken uniformly by all lanes on scalar state that lives once in LDS; every loop over samples is a phase
// `IC_FOR(tid) { ... } IC_SYNC();` spread over the lanes.
//
// EVERYTHING THE PROGRAM TOUCHES WHILE IT DECIDES IS IN LDS AND ADDRESSED AS LDS (round 4).  Rounds 1-3 kept search.c:1220-1225's lcu_t x 5 as a 126 KB slab in HBM per
// workgroup, the program object behind a generic `this`, and the tables / per-picture model behind generic pointers: 65 % of a wavefront's time was s_waitcnt
// (profiles/r04_a_inter_pmc_before.json: 94 GB moved per launch for 1.2 GB of pictures; every LDS access a flat_load, every table look-up a trip to L2).  Now:
//  * the work tree is what its five levels can differ in: one DECIDED picture (every finished CU, whatever its depth: level 3's view) plus ONE candidate per depth,
//    the CU under evaluation there -- 32x32 (level 1), 16x16 (level 2); an 8x8 CU is evaluated in place.  Copying a level down (search.c:943-1063) is candidate ->
//    decided picture, copying up is nothing at all; CU records likewise (one record per 8x8 of the decided picture + the CU under evaluation per depth); the source
//    samples are staged per 32x32 quadrant; the candidates' quantised levels wait in a small HBM scratch of the workgroup (written, never read on the way) and go
//    to the output block when their CU wins;
//  * the program's state (g_ic), the LDS block (g_il) and in it copies of the per-picture model and of the small tables (transform matrix, filters, the CABAC state
//    machine, reference availability) are workgroup-scope variables: every access is a ds_read / ds_write at a constant offset, none goes through a generic pointer;
//    pointers into LDS that cross a call are typed as such (KVZ_LDS), pointers to pictures as global (KVZ_GLB).
// CTUs of a picture run in WPP order under the ticket schedule of kvz_ctu_kernels.hpp; pictures of one sequence are launches in order, the launch carries picture k
// of many independent sequences.  tests/hostsim compiles this file for the host (a phase = a loop over tid).
// Coefficients are priced as kvz_get_coeff_cost does (rdo.c:311-340): kvz_fast_coeff_cost while the picture QP lies below fast-residual-cost 28 (fused with the
// quantisation), the residual coder in counting mode on the search contexts from there on (coeff_bits_cabac: kvz_residual.hpp's syntax walk into a price sink).
// Restrictions of this version: square PUs, one reference picture.
#pragma once
#include <stddef.h>

#include "../../include/kvz_hip_types.h"
#include "../../include/kvz_hip_dev.h"
#include "kvz_ops.hpp"
#include "kvz_tables.hpp"
#include "kvz_residual.hpp"

namespace kvz {

#ifndef KVZ_ICTU_THREADS
#define KVZ_ICTU_THREADS 64  // lanes per CTU.  Measured on the MI355X with 256 sequences in flight (416x240): 64 lanes 20.9 k CTUs/s, 128: 18.8 k, 256: 11.9 k
                             // -- the program is a chain of short phases, and with one wavefront per CTU a barrier costs nothing and more CTUs share a CU
#endif
#ifdef KVZ_HOSTSIM
#define IC_FOR(tid) for (int tid = 0; tid < KVZ_ICTU_THREADS; ++tid)
#ifdef KVZ_ICTU_COUNT_PHASES  // developer build of the host simulation: barriers per stage (tools/inter_phase_count.py)
static long g_ic_phases[32]; static int g_ic_cat = 31;
#define IC_SYNC() (++g_ic_phases[g_ic_cat])
#define IC_COUNT(slot) (++g_ic_phases[slot])  /* events per CTU, slots 16 .. 30 (tools/inter_phase_count.py names them) */
#else
#define IC_SYNC()
#endif
#ifndef IC_COUNT
#define IC_COUNT(slot)
#endif
#define IC_LDS_ADD(p, v) (*(p) += (v))
#define KVZ_LDS
#define KVZ_GLB
#define IC_WGVAR static
static inline int mul24(int a, int b) { return a * b; }
static inline unsigned umul24(unsigned a, unsigned b) { return a * b; }
static inline int mul24v(int a, int b) { return a * b; }
#else
// a 32-bit multiply is a quarter-rate instruction, the 24-bit one a full-rate one: rows, strides, cell indices and the div_by products all fit
__device__ __forceinline__ int mul24(int a, int b) { return __mul24(a, b); }
__device__ __forceinline__ unsigned umul24(unsigned a, unsigned b) { return __umul24(a, b); }
// ... and where the compiler would not take it (one factor in a scalar register: it falls back to the 32-bit multiply) the instruction by name
__device__ __forceinline__ int mul24v(int a, int b) { int r; asm("v_mul_i32_i24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
#define IC_FOR(tid) for (int tid = threadIdx.x, once_ = 1; once_; once_ = 0)
#define IC_SYNC() __syncthreads()  /* (one wavefront: the compiler drops the s_barrier; a wavefront-scope fence in its place measured the same, profiles/experiments) */
#define IC_COUNT(slot)
#define IC_LDS_ADD(p, v) atomicAdd((p), (v))
#define KVZ_LDS __attribute__((address_space(3)))  // a pointer into the workgroup's LDS block: ds_read / ds_write, not flat
#define KVZ_GLB __attribute__((address_space(1)))  // a pointer into HBM (pictures, CU records, levels): global_load / global_store, not flat
#define IC_WGVAR __shared__
#endif
// IC_FN: the program's larger functions are real calls on the device (one copy each; inlined, the four depths of search_cu_b would each carry a copy of everything).
// Arguments are values and indices: an LDS object is named, not passed.
#ifdef KVZ_HOSTSIM
#define IC_FN static inline
#define IC_FN_CALL static inline
#define IC_FN_INTER static inline
#define IC_FN_INTRA static inline
#else
#ifndef KVZ_ICTU_WAVES_PER_EU
#define KVZ_ICTU_WAVES_PER_EU 3  /* 13.3 KB of LDS: 12 workgroups (wavefronts) per CU = 3 on a SIMD, at <= 168 VGPRs */
#endif
// KVZ_ICTU_INLINE: 0 -- every IC_FN a call; 1 -- the compiler decides; 2 -- everything inlined but the recursion (IC_FN_CALL).  A call whose callee needs more than the
// caller-saved registers spills to scratch (HBM) in its prologue: calls are round trips to memory
#ifndef KVZ_ICTU_INLINE
#define KVZ_ICTU_INLINE 2
#endif
#if KVZ_ICTU_INLINE == 2
#define IC_FN static __device__ __forceinline__
#elif KVZ_ICTU_INLINE == 1
#define IC_FN static __device__ inline
#else
#define IC_FN static __device__ __noinline__
#endif
// the two halves of a CU's evaluation that every depth shares: as calls (one copy) or inlined into each depth's search_cu_b
#ifdef KVZ_ICTU_INTER_CALL
#define IC_FN_INTER static __device__ __noinline__
#else
#define IC_FN_INTER IC_FN
#endif
#ifdef KVZ_ICTU_INTRA_CALL
#define IC_FN_INTRA static __device__ __noinline__
#else
#define IC_FN_INTRA IC_FN
#endif
#ifdef KVZ_ICTU_RECURSION_INLINE
#define IC_FN_CALL static __device__ __forceinline__
#else
#define IC_FN_CALL static __device__ __noinline__
#endif
#endif
#define IC_DEV static KVZ_DEV
// stage profile (developer builds, -DKVZ_ICTU_PROFILE): ticks of the 100 MHz clock per category, lane 0 of every workgroup adds into F.prof[]
#if defined(KVZ_ICTU_PROFILE) && !defined(KVZ_HOSTSIM)
#define IC_PROF(cat, stmt) do { const unsigned long long t0_ = __builtin_amdgcn_s_memrealtime(); stmt; if (threadIdx.x == 0 && F.prof) atomicAdd(&F.prof[cat], __builtin_amdgcn_s_memrealtime() - t0_); } while (0)
#elif defined(KVZ_ICTU_COUNT_PHASES)
#define IC_PROF(cat, stmt) do { const int c0_ = g_ic_cat; g_ic_cat = (cat); stmt; g_ic_cat = c0_; } while (0)
#else
#define IC_PROF(cat, stmt) do { stmt; } while (0)
#endif
enum { IP_MERGE = 0, IP_EARLY_SKIP, IP_ME, IP_FME, IP_CAND, IP_INTRA_SEARCH, IP_INTRA_RECON, IP_INTER_RECON, IP_COST, IP_COPY, IP_IO, IP_TOTAL, IP_WINNER_MC, IP_CTX, IP_ZERO_COEFF, IP_SYNTAX, IP_COUNT };

typedef kvz_hip_cu_info CuInfo;  // one 4x4 unit of the frame's CU info (include/kvz_hip_dev.h)
typedef KVZ_LDS u8 lu8;
typedef KVZ_LDS i16 li16;
typedef KVZ_GLB u8 gu8;
typedef KVZ_GLB i16 gi16;

// compact context numbering of a B slice (cabac.h:63-100): the CU / transform-tree syntax in the first 32 bytes -- all that moves while coefficients are priced with the
// fast estimate --, then the residual coder's contexts, KVZ_HIP_CX_SIG_CG .. KVZ_HIP_CX_ABS_CHROMA + 1 of include/kvz_hip_types.h in that order (picture QP >= 28)
enum { IX_SPLIT = 0 /* 3 */, IX_SKIP = 3 /* 3 */, IX_MERGE_FLAG = 6, IX_MERGE_IDX = 7, IX_PRED_MODE = 8, IX_PART = 9, IX_INTRA = 10, IX_CHROMA = 11, IX_CBF_LUMA = 12 /* 2 */,
       IX_CBF_CHROMA = 14 /* 2 */, IX_MVD = 16 /* 2 */, IX_MVP_IDX = 18, IX_INTER_DIR = 19 /* 5 */, IX_ROOT_CBF = 24, IX_SYNTAX = 32,
       IX_RES = 32 /* 136 */, IX_COUNT = IX_RES + (KVZ_HIP_CX_ABS_CHROMA + 2 - KVZ_HIP_CX_SIG_CG) };
struct alignas(8) ICtx { u8 s[IX_COUNT]; };
static_assert(IX_COUNT == 168 && sizeof(ICtx) == 168, "context sets are copied as 32-bit words");
// The kernel comes in two builds (kvz_inter_tu.hip, -DKVZ_ICTU_CABAC=0 / 1; the host picks by the picture's coeff_cabac): the context sets in LDS -- the search's, two
// copies per depth, the row coder's -- carry the residual coder's 136 states only where coefficients are priced with them; without, 32 bytes each and 1.3 KB of LDS less
#ifndef KVZ_ICTU_CABAC
#define KVZ_ICTU_CABAC 1
#endif
struct alignas(8) ICtxL { u8 s[KVZ_ICTU_CABAC ? IX_COUNT : IX_SYNTAX]; };

struct InterModel {  // per picture
  double lambda, lambda_sqrt;
  uint64_t coeff_weights;
  int qp, poc, mv_constraint, sao, deblock, fme_level, pu_depth_inter_max, no_wpp;
  int coeff_cabac;  // qp >= fast-residual-cost (28 `ultrafast` .. `veryfast`, 0 `faster`: cfg.c:509-593): get_coeff_cabac_cost instead of kvz_fast_coeff_cost
  int ref_w, ref_h, tile_x, tile_y;  // the reference FRAME's size and this picture's (tile's) origin in it (include/kvz_hip_dev.h kvz_hip_inter_params); not tiled: the picture's size, 0, 0
  int no_tmvp;
  alignas(8) u8 ctx_init[IX_COUNT];  // an ICtx: the slice's initial states
  QuantScalars qf[2][4], qi[2][4];  // forward / inverse scalars, [luma, chroma][log2 size - 2]
  float fbits[128];                 // kvz_f_entropy_bits
};

struct InterSlab {  // HBM scratch of one resident workgroup: the quantised levels of the candidates of depth 1 and 2 (Y | U | V, raster inside each plane's block), written
                    // when the CU is quantised and copied to the output block if it wins; `out` stands in for the output block when the caller wants no coefficients
  i16 cand1[32 * 32 + 2 * 16 * 16];
  i16 cand2[16 * 16 + 2 * 8 * 8];
  i16 out[64 * 64 + 2 * 32 * 32];
};

struct InterFrames {
  int W, H, wc, hc;
  long frame_px, cells;  // bytes of a picture (Y|U|V), CU records of a picture
  const u8 *src;         // [n] pictures to encode
  const u8 *ref;         // [n] their reference pictures (the previous picture of each sequence after its loop filters)
  const CuInfo *ref_cu;  // [n] the reference pictures' CU info
  u8 *rec;               // [n] out: reconstruction before the loop filters
  CuInfo *cu;            // [n] out: CU info
  i16 *coeff;            // [n] out: KVZ_HIP_CTU_COEFFS per CTU (raster CTU order), z-order inside as lcu_coeff_t
  ICtx *ctx_out;         // [n][CTUs]: the row coder's contexts after each CTU
  InterSlab *slabs;      // one per resident workgroup
  const int *tile_xy;    // NULL, or [n][2]: every picture's own origin in its reference frame (tiles of one size from different places of the grid in one launch)
  int ref_count;         // 0, or the number of reference frames: picture p predicts from frame p % ref_count (several tiles of one frame in the launch)
  unsigned long long *prof;  // [IP_COUNT] or NULL (KVZ_ICTU_PROFILE)
};

#define IC_WS 28  /* window stride: 3 bytes of alignment + up to 24 samples (a 16x16 tile + 8), rounded to dwords */
#define IC_GS 17  /* stride of the horizontal intermediates: up to 16 + 1 columns */
#define IC_MREF_STRIDE 36  /* q in [-16, 17] for a 16x16 CU (an odd number of dwords: the modes fall into different banks) */
#define IC_MREF_ORG 16
struct MCand { i16 mv[2][2]; u8 ref[2], dir; };                         // inter_merge_cand_t
struct PuSearch { int x, y, w; i16 mv_cand[2][2]; MCand merge[5]; int num_merge, merge_dup; };  // merge_dup: bit k = merge[k] repeats an earlier entry
struct UMap { CuInfo unit[5]; double cost[5], bits[5]; int8_t keys[5]; int size; };

struct PView { lu8 *p; int s; };  // a plane of a block in LDS: sample (x, y) at p[y * s + x]

// Per-picture constants and the small tables, copied into LDS once per workgroup (the kernel is persistent): a table look-up on the decision path is an LDS read, not a
// round trip to L2 / HBM.
struct QScal { int flat_q, add, q_bits, dq_scale, dq_shift; };  // kvz_quant / kvz_dequant with flat lists (quant-generic.c:57-81, 335-339)
struct InterConst {
  double lambda, lambda_sqrt;
  uint64_t coeff_weights;
  int qp, poc, mv_constraint, sao, deblock, fme_level, pu_depth_inter_max, no_wpp, coeff_cabac;
  int ref_w, ref_h, tile_x, tile_y;  // the reference frame and the picture's origin in it (tiles)
  int no_tmvp;
  QScal q[2][4];               // [luma, chroma][log2 size - 2]
  float fbits[128];            // kvz_f_entropy_bits
  int8_t dct32[32 * 32];       // kvz_g_dct_32 (dct-generic.c:83-120) as signed bytes; the N-point matrix is its rows 0, 32 / N, 2 * 32 / N .. and first N columns
  int8_t dst4[16];             // dct-generic.c:38-44
  u8 ctx_next[2][128];         // the CABAC state machine (kvz_tables.hpp)
  int8_t luma_filter[4][8];    // filter.c:66-72
  int8_t chroma_filter[8][4];  // filter.c:74-84
  u8 avail_top[16][16], avail_left[16][16];  // intra.c:47-82 as regenerated by kvz_tables.hpp
  u32 div_magic[32];           // 2^20 / d + 1 (div_by's multiplier, kvz_inter_ctu_pix.inc): there is no integer division in hardware, and one per staged window was ~25 instructions
};

// A 32x32 block is interpolated, compared and transformed in 16x16 TILES (its four quadrants; smaller blocks are one tile): the sample buffers below are sized for a tile.
struct InterLds {
  // ---- the work tree ----
  // (the decided picture itself is the frame's reconstruction in HBM: a finished CU is written there once, commit_down, and read back one reference sample per lane)
  alignas(8) u8 orgq[32 * 32 + 2 * 16 * 16]; // source samples of the 32x32 quadrant being searched: Y | U | V (every sample loop runs inside one depth-1 CU)
  alignas(8) u8 C1[32 * 32 + 2 * 16 * 16];   // the depth-1 CU under evaluation
  alignas(8) u8 C2[16 * 16 + 2 * 8 * 8];     // the depth-2 CU under evaluation
  alignas(8) u8 C3[8 * 8 + 2 * 4 * 4];       // the depth-3 CU under evaluation
  alignas(8) u8 Z3[8 * 8 + 2 * 4 * 4];       // cu_zero_coeff_cost's copy of a depth-3 CU's prediction (search.c:222 puts it into level 4)
  struct alignas(8) DCell { CuInfo c; uint16_t pad; };  // 24 bytes: a record of the decided picture is read whole with three 8-byte LDS reads (cell_at), not as eleven half-words
  DCell Dcu[64];                             // CU records of the decided picture, one per 8x8 (the smallest CU)
  union {                               // the inter side's tile buffers | the parked prediction | the intra side's references and scores
    struct {
      alignas(8) u8 win[24 * IC_WS + 16];  // reference window of a tile (motion compensation, fractional search): rows staged from a dword-aligned column, sample (r, c) at
                                           // win[r * IC_WS + win_xo + c]; 16 bytes of slack behind the last row for the horizontal pass's whole-dword reads
      alignas(8) i16 g[8 * 72];            // 14-bit horizontal intermediates of a tile (24 rows, stride IC_GS); the SATD's eight slots of 72
    };
    alignas(8) u8 park[32 * 32 + 2 * 16 * 16];  // cu_zero_coeff_cost's copy of a depth-1 / depth-2 CU's prediction (search.c:222): between parking and the decision
                                                // the CU is quantised and priced, nothing is predicted
    struct {                               // (behind `win` only: the intra scores run the SATD, whose host form goes through g)
      u32 mcost[36];                       // SATD of every intra mode of the CU under evaluation
      u8 top[65], left[65], ftop[65], fleft[65];
      int8_t modes[36];
      int8_t todo[36];
    };
  };
  int win_xo;
  union {                          // the sample buffers of stages that never overlap in time
    struct {
      alignas(8) u8 pred[4][16 * 16];  // the candidate planes of a fractional step (tile)
      i16 im[2][16 * 16];              // 14-bit predictions of the two lists (tile)
    };
    alignas(8) i16 tb[32 * 32];        // the transform path: one block, every pass in place
    alignas(8) u8 planes[8 * 256];     // intra predictions being scored: eight 16x16 blocks or thirty-two 8x8
    struct {                           // intra_all_mode_costs: all 35 modes of a CU scored at once
      alignas(8) u8 flat[3][256];      // the predictions of planar, DC and mode 34 (scored through satd_tiles)
      alignas(8) u8 org_t[256];        // the CU's source block transposed (horizontal modes are predicted and scored on the transposed problem)
      alignas(4) u8 mref[15][IC_MREF_STRIDE];  // the extended main reference of the modes with a negative displacement, 11 .. 25: [mode - 11][IC_MREF_ORG + q] = ref_main[q], q in [-w, w + 1]
    };
  };
  u32 tsum[8];                     // satd_tiles: the eight tiles of a round
  // Scalar work memory.  Every lane runs the same control flow on the same values, and with one wavefront per CTU the lanes are in lockstep: small arrays that are
  // indexed at run time live here once instead of 64 times in private (scratch) memory
  union {
    UMap amvp[2];      // search_pu_inter's candidates of the two lists (dead once search_cu_inter has returned)
    double costs[36];  // search_cu_intra's sort
  };
  UMap merge;
  PuSearch pu;
  unsigned long long intra_done;  // the modes L->mcost holds for the CU under evaluation
  int mvc_key[4];      // the PU and list L->mvc holds the AMVP predictors of ({x, y, w, list}; w = 0: none) -- the search asks for the same pair up to three times
  i16 mvc[2][2];
  int level_holds;     // after search_pu_inter: bit 0 / 1 = the level's luma / chroma samples are the prediction of the best merge candidate (merge.keys[0])
  i16 px[8], py[8];   // the probes of a round: integer displacements
  u32 sad[8];
  i16 pbits[8];        // per probe: MVD bits against the cheaper predictor, -1 where the vector is not allowed
  u32 ssd[2];          // ssd_cu's result: luma, U + V
  ICtxL ctx;  // state->search_cabac's contexts (indexed at run time on every priced bin)
  ICtxL pre[4], post[3];  // search_cu's copies of them, per depth (search.c:655, 956; a depth-3 CU is never split)
  struct { double cost, split_cost; } fr[3];  // search_cu's locals that live across its children (depth 0 .. 2): nothing of a depth is held in registers while its
  u8 child[4];                               // children run; a depth's position follows from the child indices above it
  ICtxL row;              // the row coder's contexts at the start of the CTU (the finished CTU's syntax runs on them)
  CuInfo cur_cu[4];      // the CU under evaluation at each depth of the recursion
  struct { int mvx, mvy; double cost, bits; } best;  // check_mv_cost's best so far
  struct { int mv[2]; double cost, bits; } frac;     // me_fractional's result
  double inter_cost, inter_bitcost, intra_cost;      // results of search_cu_inter / search_cu_intra
  double ccost[3];  // kvz_fast_coeff_cost of the transform units the CU under evaluation was last quantised into (one per plane: its transform tree is one unit); unused when coeff_cabac
  InterConst k;
};
static_assert(KVZ_ICTU_THREADS == 64, "the scalar work memory in LDS relies on one wavefront per CTU");

// the program's state: picture geometry and pointers (the kernel arguments), the CTU at hand
static_assert(sizeof(u32) * 36 + 4 * 65 + 72 <= 24 * IC_WS + 16, "the intra side's scratch must fit behind the window");
struct InterState {
  InterFrames F;
  const InterModel *model;  // in HBM: the slice's initial context states are read from it where a coder starts (two copies per CTU row)
  const Tables *tb;   // the large tables that stay in HBM: the coefficient scans (the residual coder's walk, picture QP >= 28 only)
  InterSlab *S;
  int frame, cx, cy;
  const CuInfo *cu_frame;  // F.cu + frame * F.cells: the picture's CU records (neighbours in finished CTUs), once per CTU
  int ref_idx;  // frame % F.ref_count (several tiles of one reference frame), once per CTU: every address into the reference went through a division otherwise
  // the luma planes of the picture's reference frame, source and reconstruction, once per CTU: (long) frame x plane-size products behind every sample address otherwise
  const uint8_t *ref_base, *src_base;
  uint8_t *rec_base;
  const CuInfo *ref_cu_base;  // the reference frame's CU records
  int16_t *coef_out;  // the CTU's 6144 levels: F.coeff's slot of it, or the workgroup's scratch
  int acc_slot;
};

IC_WGVAR InterLds g_il;
IC_WGVAR InterState g_ic;

#define IC_MAX_COST 1.7e+308
#define IC_MAX_INT 2147483647.0

// Names the program text uses (they end with this file): L-> the LDS block, M-> the picture's model, K-> the tables, F. the frames, S-> the HBM scratch
#define L (&g_il)
#define M (&g_il.k)
#define K (&g_il.k)
#define F (g_ic.F)
#define S (g_ic.S)
#define cab (g_il.ctx)  /* the search contexts */
#define frame (g_ic.frame)
#define cx (g_ic.cx)
#define cy (g_ic.cy)

struct InterCtu {
  // ---- small things ----
  // Level lv's samples of plane c of the CU whose luma origin inside the LCU is (xl, yl) -- the CU under evaluation at depth lv (levels 1, 2: its candidate buffer;
  // level 3: the 8x8 candidate buffer; level 4: the 8x8 side buffer)
  IC_DEV PView lvl(int lv, int c, int xl, int yl)
  {
    const int sh = c ? 1 : 0;
    if (lv == 1) return PView{ (lu8 *)L->C1 + (c == 0 ? 0 : (c == 1 ? 1024 : 1280)), 32 >> sh };
    if (lv == 2) return PView{ (lu8 *)L->C2 + (c == 0 ? 0 : (c == 1 ? 256 : 320)), 16 >> sh };
    if (lv == 4) return PView{ (lu8 *)L->Z3 + (c == 0 ? 0 : (c == 1 ? 64 : 80)), 8 >> sh };
    return PView{ (lu8 *)L->C3 + (c == 0 ? 0 : (c == 1 ? 64 : 80)), 8 >> sh };
  }
  // where cu_zero_coeff_cost parks the prediction of the depth-lv CU (search.c:222: the next level)
  IC_DEV PView parked(int lv, int c, int xl, int yl)
  {
    if (lv == 3) return lvl(4, c, xl, yl);
    const int w = 64 >> lv, sh = c ? 1 : 0;
    return PView{ (lu8 *)L->park + (c == 0 ? 0 : (c == 1 ? w * w : w * w + (w * w >> 2))), w >> sh };
  }
  // the luma prediction of the best merge candidate so far (search_pu_inter): the fractional search's candidate planes, which nothing uses before the motion search
  IC_DEV PView best_luma(int lv) { return PView{ (lu8 *)&L->pred[0][0], 64 >> lv }; }
  // the source samples at LCU position (xl, yl) [luma coordinates] of plane c: inside the quadrant staged by load_org_quadrant
  IC_DEV PView orgv(int c, int xl, int yl)
  {
    const int sh = c ? 1 : 0;
    return PView{ (lu8 *)L->orgq + (c == 0 ? 0 : (c == 1 ? 1024 : 1280)) + ((yl & 31) >> sh) * (32 >> sh) + ((xl & 31) >> sh), 32 >> sh };
  }
  IC_DEV unsigned zorder(int x, int y)
  {
    unsigned r = 0;
    for (int b = 0; b < 4; b++) r |= (((unsigned)(x >> (2 + b)) & 1u) << (2 * b)) | (((unsigned)(y >> (2 + b)) & 1u) << (2 * b + 1));
    return r * 16;
  }
  // the quantised levels of the depth-lv CU at (xl, yl): candidates of depth 1 and 2 in the workgroup's scratch, a depth-3 CU's straight in the output block
  IC_DEV gi16 *out_coef() { return (gi16 *)g_ic.coef_out; }
  IC_DEV gi16 *coef(int lv, int c, int xl, int yl)
  {
    const int sh = c ? 1 : 0;
    if (lv == 1) return (gi16 *)S->cand1 + (c == 0 ? 0 : (c == 1 ? 1024 : 1280));
    if (lv == 2) return (gi16 *)S->cand2 + (c == 0 ? 0 : (c == 1 ? 256 : 320));
    return out_coef() + (c == 0 ? 0 : (c == 1 ? 4096 : 5120)) + zorder(xl >> sh, yl >> sh);
  }
  IC_DEV int plane_off(int c) { const int n = F.W * F.H; return c == 0 ? 0 : (c == 1 ? n : n + (n >> 2)); }  // (W, H even: n * 5 / 4 exactly)
  // plane c of the picture's reference FRAME (K->ref_w x K->ref_h; the picture lies at (K->tile_x, K->tile_y) in it)
  IC_DEV int ref_index() { return g_ic.ref_idx; }
  IC_DEV const gu8 *refp(int c) { const int n = K->ref_w * K->ref_h; return (const gu8 *)g_ic.ref_base + (c == 0 ? 0 : (c == 1 ? n : n + (n >> 2))); }
  IC_DEV const CuInfo *ref_cu_frame() { return g_ic.ref_cu_base; }
  IC_DEV const gu8 *srcp(int c) { return (const gu8 *)g_ic.src_base + plane_off(c); }
  IC_DEV gu8 *recp(int c) { return (gu8 *)g_ic.rec_base + plane_off(c); }
  IC_DEV CuInfo *dcell(int xl, int yl) { return &L->Dcu[((yl >> 3) & 7) * 8 + ((xl >> 3) & 7)].c; }  // (& 7: nothing for a position inside the CTU; the index's range lets the record size multiply at full rate)
  IC_DEV CuInfo load_dcell(int xl, int yl)  // *dcell(xl, yl), the whole record
  {
    union { CuInfo c; unsigned long long q[3]; } u;
    const KVZ_LDS unsigned long long *p = (const KVZ_LDS unsigned long long *)&L->Dcu[((yl >> 3) & 7) * 8 + ((xl >> 3) & 7)];
    u.q[0] = p[0]; u.q[1] = p[1]; u.q[2] = p[2];
    return u.c;
  }
  IC_DEV bool cbf_is_set(unsigned cbf, int depth, int plane) { return (cbf & ((0x1fu >> depth) << (5 * plane))) != 0; }
  IC_DEV bool cbf_any(unsigned cbf, int depth) { return cbf_is_set(cbf, depth, 0) || cbf_is_set(cbf, depth, 1) || cbf_is_set(cbf, depth, 2); }
  IC_DEV uint16_t cbf_set(unsigned cbf, int depth, int plane) { return (uint16_t)(cbf | ((0x10u >> depth) << (5 * plane))); }
  IC_DEV uint16_t cbf_clear(unsigned cbf, int depth, int plane) { return (uint16_t)(cbf & ~((0x1fu >> depth) << (5 * plane))); }

  // Sum over the lanes of a value accumulated inside a phase.  The idiom is
  //   u32 part = 0;  IC_FOR(tid) { ... part += ...; }  const u32 total = lanes_sum(part);
  // on the host the phase is a loop over tid around ONE `part`, which therefore already holds the total; on the device every lane has its own.
  IC_DEV u32 lanes_sum(u32 part)
  {
#ifdef KVZ_HOSTSIM
    return part;
#else
    int x = (int)part;  // wave64: row_shr 8 / 4 / 2 / 1 inside rows of 16 lanes, then the four row totals
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, true);
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, true);
    return (u32)(__builtin_amdgcn_readlane(x, 15) + __builtin_amdgcn_readlane(x, 31) + __builtin_amdgcn_readlane(x, 47) + __builtin_amdgcn_readlane(x, 63));
#endif
  }

  // four samples of HBM at any alignment; acc + the sum of absolute differences of two groups of four samples (v_sad_u8)
  IC_DEV u32 glb_read32u(const gu8 *p)
  {
#ifdef KVZ_HOSTSIM
    return (u32)p[0] | (u32)p[1] << 8 | (u32)p[2] << 16 | (u32)p[3] << 24;
#else
    typedef u32 __attribute__((aligned(1))) u32_any;
    return *(const KVZ_GLB u32_any *)p;
#endif
  }
  IC_DEV u32 sad4(u32 a, u32 b, u32 acc)
  {
#ifdef KVZ_HOSTSIM
    for (int j = 0; j < 4; j++) { const int d = (int)((a >> (8 * j)) & 255u) - (int)((b >> (8 * j)) & 255u); acc += (u32)(d < 0 ? -d : d); }
    return acc;
#else
    return __builtin_amdgcn_sad_u8(a, b, acc);
#endif
  }

  // byte-string helpers of the horizontal interpolation pass (v_alignbyte_b32, v_dot4_i32_i8; plain C++ for the host simulation)
#ifdef KVZ_HOSTSIM
  IC_DEV u32 alignbyte(u32 hi, u32 lo, u32 n) { return (u32)(((((unsigned long long)hi) << 32) | lo) >> (8 * (n & 3))); }
  IC_DEV int dot4(u32 a, u32 b, int c)
  {
    for (int k = 0; k < 4; k++) c += (int)(int8_t)(a >> (8 * k)) * (int)(int8_t)(b >> (8 * k));
    return c;
  }
#else
  IC_DEV u32 alignbyte(u32 hi, u32 lo, u32 n) { return __builtin_amdgcn_alignbyte(hi, lo, n); }
  IC_DEV int dot4(u32 a, u32 b, int c) { return __builtin_amdgcn_sdot4((int)a, (int)b, c, false); }
#endif

  // a CU record of a picture in HBM (22 bytes, 2-byte aligned), as eleven global 16-bit loads
  IC_DEV CuInfo load_cu(const CuInfo *p)
  {
    static_assert(sizeof(CuInfo) == 22, "CU records are copied as five words and a half-word");
#ifdef KVZ_HOSTSIM
    return *p;
#else
    // 2-byte aligned: five unaligned 32-bit loads and a 16-bit one (global memory takes unaligned words), not eleven half-words and their packing
    struct __attribute__((packed, aligned(2))) W { u32 v; };
    union { CuInfo c; u32 w[6]; } u;
    const KVZ_GLB W *q = (const KVZ_GLB W *)p;
    for (int i = 0; i < 5; i++) u.w[i] = q[i].v;
    u.w[5] = ((const KVZ_GLB uint16_t *)p)[10];
    return u.c;
#endif
  }
  IC_DEV void store_cu(CuInfo *p, const CuInfo &c)
  {
    union { CuInfo c; uint16_t h[11]; } u;
    u.c = c;
    for (int i = 0; i < 11; i++) ((KVZ_GLB uint16_t *)p)[i] = u.h[i];
  }
  // the CU info of luma position (fx, fy), which lies outside the CU under evaluation (a neighbour: every caller asks for one): inside this CTU the decided picture's
  // record -- a finished CU looks the same from every level --, else the frame's (finished CTUs)
  IC_DEV CuInfo cell_at(int fx, int fy)
  {
    IC_COUNT(17);
    if (fx >= cx && fx < cx + 64 && fy >= cy && fy < cy + 64) return load_dcell(fx - cx, fy - cy);
    return load_cu((const CuInfo *)((const uint8_t *)g_ic.cu_frame + mul24(mul24(fy >> 2, F.W >> 2) + (fx >> 2), (int)sizeof(CuInfo))));  // (a 32-bit index: a picture has fewer than 2^31 4x4 units)
  }

  // CABAC_FBITS_UPDATE on the search contexts (cabac.h:133-139)
  IC_DEV double price(int idx, int bin, bool update)
  {
    IC_COUNT(16);
    const u8 st = cab.s[idx];
    const double bits = M->fbits[st ^ bin];
    if (update) cab.s[idx] = K->ctx_next[bin != (st & 1)][st];
    return bits;
  }

  // A context set copied by the lanes, a word each: the syntax contexts always, the residual coder's only when they can have moved.  (Inlined: the address spaces of
  // the two sets are the call site's.)
  template <class PD, class PS> IC_DEV void ctx_copy_words(PD dst, PS src)
  {
    const int words = M->coeff_cabac ? IX_COUNT / 4 : IX_SYNTAX / 4;
    IC_FOR(tid) { for (int i = tid; i < words; i += KVZ_ICTU_THREADS) dst[i] = src[i]; }
    IC_SYNC();
  }
#define IC_CTX_L(set) ((KVZ_LDS u32 *)(set).s)
#define IC_CTX_G(ptr) ((KVZ_GLB u32 *)(ptr)->s)
  // get_coeff_cabac_cost (rdo.c:220-263): kvz_encode_coeff_nxn in counting mode on the search contexts -- every context-coded bin at CABAC_FBITS_UPDATE's price
  // (cabac.h:133-139; the state moves only while the search has updates on), every bypass bin one bit.  The prices are multiples of 2^-15 below 2^20, so their
  // sum in a double does not depend on the order the reference adds them in.
  struct PriceSink {
    bool update; double bits;
    KVZ_DEV void ctx(int c, int v)
    {
      const int idx = IX_RES + c - KVZ_HIP_CX_SIG_CG;
      const u8 st = cab.s[idx];
      bits += M->fbits[st ^ (v ? 1 : 0)];
      if (update) cab.s[idx] = K->ctx_next[(v ? 1 : 0) != (st & 1)][st];
    }
    KVZ_DEV void ep(u32, int n) { bits += n; }
  };
  IC_FN double coeff_bits_cabac(const gi16 *coeff, int log2w, int type, int scan_mode, bool update)
  {
    if (!KVZ_ICTU_CABAC) return 0.0;  // (this build's context sets do not hold the residual coder's states; the host never sends it a picture priced with them)
    PriceSink s{ update, 0.0 };
    entropy_coeff_nxn(s, g_ic.tb, (const i16 *)coeff, log2w, type, scan_mode);
    return s.bits;
  }

#include "kvz_inter_ctu_cand.inc"
#include "kvz_inter_ctu_pix.inc"
#include "kvz_inter_ctu_search.inc"

  // the workgroup's constants, once per launch: the picture's model and the small tables into LDS
  IC_FN void load_constants(const InterModel *model, const Tables *tb)
  {
    IC_FOR(tid) {
      if (tid == 0) {
        K->lambda = model->lambda; K->lambda_sqrt = model->lambda_sqrt; K->coeff_weights = model->coeff_weights;
        K->qp = model->qp; K->poc = model->poc; K->mv_constraint = model->mv_constraint; K->sao = model->sao; K->deblock = model->deblock; K->fme_level = model->fme_level;
        K->pu_depth_inter_max = model->pu_depth_inter_max; K->no_wpp = model->no_wpp; K->coeff_cabac = model->coeff_cabac;
        K->ref_w = model->ref_w; K->ref_h = model->ref_h; K->tile_x = model->tile_x; K->tile_y = model->tile_y; K->no_tmvp = model->no_tmvp;
      }
      if (tid < 8) {
        const QuantScalars f = model->qf[tid >> 2][tid & 3], iv = model->qi[tid >> 2][tid & 3];
        K->q[tid >> 2][tid & 3] = QScal{ f.flat_q, f.add, f.q_bits, iv.dq_scale, iv.dq_shift };
      }
      for (int i = tid; i < 128; i += KVZ_ICTU_THREADS) K->fbits[i] = ((const KVZ_GLB float *)model->fbits)[i];
      for (int i = tid; i < 1024; i += KVZ_ICTU_THREADS) K->dct32[i] = (int8_t)((const KVZ_GLB i16 *)tb->dct[3])[i];
      for (int i = tid; i < 16; i += KVZ_ICTU_THREADS) K->dst4[i] = (int8_t)((const KVZ_GLB i16 *)tb->dst4)[i];
      for (int i = tid; i < 256; i += KVZ_ICTU_THREADS) {
        (&K->ctx_next[0][0])[i] = ((const KVZ_GLB u8 *)&tb->ctx_next[0][0])[i];
        (&K->avail_top[0][0])[i] = ((const KVZ_GLB u8 *)&tb->avail_top[0][0])[i];
        (&K->avail_left[0][0])[i] = ((const KVZ_GLB u8 *)&tb->avail_left[0][0])[i];
      }
      for (int i = tid; i < 32; i += KVZ_ICTU_THREADS) K->div_magic[i] = i ? (1u << 20) / (unsigned)i + 1u : 0u;
      for (int i = tid; i < 32; i += KVZ_ICTU_THREADS) {
        (&K->luma_filter[0][0])[i] = ((const KVZ_GLB int8_t *)&tb->luma_filter[0][0])[i];
        (&K->chroma_filter[0][0])[i] = ((const KVZ_GLB int8_t *)&tb->chroma_filter[0][0])[i];
      }
    }
    IC_SYNC();
  }
  IC_DEV void begin_launch(const InterFrames &frames, const InterModel *model, const Tables *tb, InterSlab *slab)
  {
    IC_FOR(tid) { if (tid == 0) { F = frames; g_ic.model = model; g_ic.tb = tb; S = slab; g_ic.acc_slot = 0; } }
    IC_SYNC();
    load_constants(model, tb);
  }
  IC_DEV void begin_ctu(int frame_, int cx_, int cy_)
  {
    IC_FOR(tid) { if (tid == 0) { frame = frame_; cx = cx_; cy = cy_; g_ic.ref_idx = F.ref_count ? frame_ % F.ref_count : frame_; g_ic.cu_frame = (const CuInfo *)F.cu + (long)frame_ * F.cells;
      g_ic.ref_base = (const uint8_t *)F.ref + g_ic.ref_idx * ((long)K->ref_w * K->ref_h * 3 / 2); g_ic.ref_cu_base = F.ref_cu + g_ic.ref_idx * ((long)(K->ref_w >> 2) * (K->ref_h >> 2)); g_ic.src_base = (const uint8_t *)F.src + frame_ * F.frame_px; g_ic.rec_base = (uint8_t *)F.rec + frame_ * F.frame_px;
      g_ic.coef_out = F.coeff ? (int16_t *)F.coeff + ((long)frame_ * F.wc * F.hc + (cy_ >> 6) * F.wc + (cx_ >> 6)) * 6144 : (int16_t *)S->out;
      if (F.tile_xy) {
      // the origin is a DEVICE-side input nobody validated: brought inside the reference frame here (multiples of 8, the tile inside the frame), so that no read of
      // the reference picture or of its CU records can leave them whatever the table holds; a valid table is unchanged
      const int tx = ((const KVZ_GLB int *)F.tile_xy)[2 * frame_], ty = ((const KVZ_GLB int *)F.tile_xy)[2 * frame_ + 1];
      K->tile_x = imax(0, imin(tx, K->ref_w - F.W)) & ~7; K->tile_y = imax(0, imin(ty, K->ref_h - F.H)) & ~7;
    } } }
    IC_SYNC();
  }
};

#undef L
#undef M
#undef K
#undef F
#undef S
#undef cab
#undef frame
#undef cx
#undef cy

}  // namespace kvz
