// spgnn_gemm.hip — fp32-accurate projection GEMMs on the fp16 matrix cores of gfx950.
//
// The GNN step is bounded by its dense projections (0.98 TFLOP per step at 512 trees), and gfx950 has no
// reduced-precision fast path for fp32 inputs: the fp32 MFMA runs at 1/16 of the fp16/bf16 rate
// (157 vs 2500 TFLOP/s dense).  These kernels read fp32 operands, split every value on the fly into two
// fp16 terms x = (hi + lo) / s  (s a per-tensor power of two that centres the tensor in the fp16 range;
// hi = fp16(s x), lo = fp16(s x - hi): 22 significant bits), and accumulate the three products
//     hi_a*hi_b + hi_a*lo_b + lo_a*hi_b
// in fp32 on v_mfma_f32_32x32x16_f16.  The dropped lo*lo term is 2^-22 relative, so the result matches an
// fp32 GEMM (measured against fp64: same 1e-6 error as rocBLAS fp32; logits of the full model 8.0e-7 vs 7.2e-7),
// at 3 MFMAs per fp32-equivalent product: a 5.3x higher ceiling than the fp32 MFMA.
//
//   spgnn_gemm_nt : C[M,N] = A[M,K] * B[N,K]^T     (forward projections; input gradients with W^T as B)
//
// Tiling: 128x128 block tile, 4 waves (2x2), each wave 64x64 = 2x2 MFMA tiles of 32x32, BK = 32 fp32
// elements per stage (two k16 MFMA steps).  Global fp32 tiles are fetched into registers one stage ahead
// (16-byte loads, 128-byte row segments), converted and written to LDS as fp16 hi/lo images with an 80-byte
// row pitch (conflict-free ds_read_b128 fragment reads: 5 is coprime with the 16 slots of a bank row).
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>

#include "spgnn_hip.h"
#include "spgnn_internal.h"

namespace gemm {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int PITCH = 40;                       // halves per LDS row (32 + 8 pad) = 80 bytes
constexpr int TILE_HALVES = BM * PITCH;         // one fp16 image of a 128 x 32 tile
constexpr int kThreads = 256;

struct Args {
  const float* A; int64_t lda;
  const float* B; int64_t ldb;
  float* C; int64_t ldc;
  int M, N, K;
  const float* sA; const float* sB;             // scale blocks (spgnn_internal.h) or null (= 1)
  int nbm, nbn;
  // optional exact fp32 rank-J update applied in the epilogue: C += U[M,J] * V[J,N]  (J <= 16; the score
  // gradient term g_S * W_lr of the input gradient, which otherwise costs a read-modify-write pass over C)
  const float* U; int64_t ldu; const float* V; int64_t ldv; int J;
  const float* bias; int act;                   // optional epilogue C = act(C + bias[col]) (pipelined kernel only)
  // optional score partials (GATConv's el / er straight from the projection): for every row and every 64-column
  // block b of the first sc_cols output columns, sc_out[(row * (sc_cols/64) + b) * 2 + {0,1}] =
  // <C[row, 64b : 64b+64], sc_l / sc_r[64b : 64b+64]>  (raw product, before rank-J / bias / activation)
  const float* sc_l; const float* sc_r; float* sc_out; int sc_cols;
  // optional head mean of a two-head layer (pipelined kernels): mean_out[row, col] = 0.5 * (this tile's final value +
  // mean_other[row, col]) - the second head's product also writes the mean over heads (replaces a pass over both heads).
  // mean_other WITHOUT mean_out is an ADDEND: C = act(A B^T + bias + mean_other) - the second of two products that share an
  // output (SAGEConv: fc_self(h) + fc_neigh(neigh), reference models.py:668-679) adds the first one's result in its epilogue
  const float* mean_other; int64_t ld_mo; float* mean_out; int64_t ld_mn;
  // optional scale block (spgnn_internal.h): max |stored value| of every tile is folded into its slots - the result's GEMM
  // operand scale for the next product, without a pass over it
  float* absmax;
  // optional feature dropout of the stored result (after bias / activation): spgnn_cat_dropout's mask over an N-wide row,
  // counter row * N + column of the float4 group; drop_p == 0: none.  N % 4 == 0 (host check).
  float drop_p, drop_inv; uint64_t drop_seed; const uint64_t* drop_seed_off;
};
// B may arrive PRE-SPLIT (spgnn_presplit): every group of four fp32 values replaced, in place, by its packed fp16 pairs
// [hi01, hi23, lo01, lo23] of s*x (the 16 bytes split4_pk would produce), rows zero padded to a multiple of four columns.
// Weights are split once per step instead of once per row tile of every product that reads them (299 times at 512 trees):
// the B half of the in-kernel conversion disappears (template flag BPS: staged words go to LDS as they are).

// -------------------------------------------------------------------------------------------------
// NT kernel, second generation: (64*WM) x 128 block tile, 2*WM waves, double-buffered LDS, ONE barrier per
// stage, and the fp32 -> fp16 hi/lo conversion of stage t+1 issued between the MFMAs of stage t (an MFMA
// occupies the vector issue port for 8 of its 32 cycles, so ~5 VALU instructions per MFMA are free).
// Global fp32 data is prefetched two stages ahead into two register sets.  Conversion is packed:
//   hi = cvt_pkrtz(x),  float(hi) = x & 0xFFFFE000 (round-toward-zero keeps the top 11 significand bits),
//   lo = cvt_pkrtz(x - float(hi))          -> 16 VALU ops per float4 instead of ~24.
// LDS stores pair rows r and r+4 in each 16-lane group so the two 64-byte row pieces of a ds_write_b64
// fall in different halves of the 128-byte store bank window (pitch 80 B: 4*80 = 64 mod 128).
// -------------------------------------------------------------------------------------------------
typedef __fp16 pk2 __attribute__((ext_vector_type(2)));

// the same function without data-dependent branches: __expf expands to a guarded sequence that hipcc wraps in
// exec-mask branches per element; v_exp_f32 (2^x) directly is one instruction and both sides become selects.
// exp(x) - 1 is only taken for x <= -0.5, where exp(x) < 0.61: no denormal input, no cancellation issue.
__device__ __forceinline__ float elu_fwd_nb(float x) {
  const float p = x * (1.f + x * (0.5f + x * (1.f / 6 + x * (1.f / 24 + x * (1.f / 120 + x * (1.f / 720 + x * (1.f / 5040 +
                  x * (1.f / 40320 + x * (1.f / 362880)))))))));
  const float e = __builtin_amdgcn_exp2f(fmaxf(x, -126.f) * 1.44269504088896341f) - 1.f;
  const float n = x > -0.5f ? p : e;
  return x > 0.f ? x : n;
}

// hi = fp16_rtz(s x), lo = fp16_rtz(s x - hi) in TEN vector instructions per four values: two packed multiplies, two packed
// conversions, one v_fma_mix_f32 per value for the residual - it reads the fp16 hi half IN PLACE as its addend and folds the
// scale in as the multiplier (s is a power of two: s x is exact, and so is s x - hi) - and two packed conversions.  The
// 128 x 128-tile kernels convert eight float4 per thread and stage beside 24 MFMAs: at sixteen instructions per float4 (the
// first form: hi recovered as `x & 0xFFFFE000`, four multiplies, four ANDs, four subtractions) the conversions alone held the
// SIMD's issue port for an estimated ~21 of every MFMA's 32 cycles.  Measured in round 4 (two libraries in one process,
// tools/step_ab.py): 5.187 vs 5.188 ms per step at 512 trees, 1.072 vs 1.070 at 64, st_gin_3 3.727 vs 3.733 - NEUTRAL: the
// conversions' issue slots are not what holds these kernels.  Kept for the second property: the residual is taken from the
// hi that was actually stored - the AND form assumed a normal fp16 hi and was off by up to one fp16 subnormal quantum
// (2^-39 of the tensor maximum) for |s x| < 2^-14.
//
// WIDE (round 4, the wide-range arithmetic of SPGNN_GEMM_WIDE): the residual is stored as lo' = 2^11 lo.  lo <= 2^-11 |s x|, so lo'
// sits in the binade range of hi and is a NORMAL fp16 wherever hi is - the plain form loses lo to fp16's subnormals once |s x|
// < 2^-3, i.e. 2^18 below the tensor maximum (the envelope of DESIGN.md section 4.2).  The kernels then keep the two cross
// products hi lo' + lo' hi in a second accumulator set and add it, times 2^-11, in the epilogue: the same three MFMAs per
// product, 22 bits for every value within ~2^28 of the maximum.  Operands pre-split in one form must be consumed in that form.
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr float kWideLo = 2048.f, kWideLoInv = 1.f / 2048.f;
template <bool WIDE = false>
__device__ __forceinline__ void split4_pk(float4 v, float s, uint2& hi, uint2& lo) {
  const f32x2 ss = {s, s};
  const f32x2 a = f32x2{v.x, v.y} * ss, b = f32x2{v.z, v.w} * ss;
  union { pk2 h; unsigned u; } h01, h23, l01, l23;
  h01.h = __builtin_amdgcn_cvt_pkrtz(a.x, a.y);
  h23.h = __builtin_amdgcn_cvt_pkrtz(b.x, b.y);
  float r0 = __builtin_fmaf(v.x, s, -(float)h01.h[0]), r1 = __builtin_fmaf(v.y, s, -(float)h01.h[1]);
  float r2 = __builtin_fmaf(v.z, s, -(float)h23.h[0]), r3 = __builtin_fmaf(v.w, s, -(float)h23.h[1]);
  if constexpr (WIDE) { r0 *= kWideLo; r1 *= kWideLo; r2 *= kWideLo; r3 *= kWideLo; }
  l01.h = __builtin_amdgcn_cvt_pkrtz(r0, r1);
  l23.h = __builtin_amdgcn_cvt_pkrtz(r2, r3);
  hi = make_uint2(h01.u, h23.u);
  lo = make_uint2(l01.u, l23.u);
}

// natural row q (0..ROWS-1) of a tile -> stored/loaded row: inside each block of 8 rows interleave (0,4,1,5,2,6,3,7)
__device__ __forceinline__ int pair_row(int q) { const int s_ = q & 7; return (q & ~7) | ((s_ >> 1) + 4 * (s_ & 1)); }

template <int ROWS, int NT>    // ROWS x 32 fp32 tile, NT threads: NL = ROWS*8/NT float4 per thread
struct TileIO {
  static constexpr int NL = ROWS * 8 / NT;
  // The loads of the K loop are UNCONDITIONAL straight-line code.  A branch around a load (bounds test, "is there a
  // next stage") makes hipcc's s_waitcnt insertion merge two histories of the VM queue at the join and wait for the
  // worse one: the old loop waited vmcnt(3..0) at the top of every stage, i.e. it drained the refill issued one
  // barrier earlier and the two-stage prefetch hid nothing (timing-only build without the loop's loads: -31 %).
  //   load_full : a stage that lies wholly below K (k0 + 32 <= K).  Rows past the end are clamped to the last row:
  //               they only feed output rows/columns that the epilogue never stores.
  //   load_any  : any stage, existing or not: a float4 that starts at or beyond K is fetched from column 0 instead
  //               (valid memory; rows are 16-byte multiples, ld % 4 == 0, so a float4 that starts below K lies
  //               inside the row's stride) and mask() zeroes what lies beyond K before the set is converted.
  static __device__ __forceinline__ void load_full(const float* __restrict__ base, int64_t ld, int row0, int nrows, int k0,
                                                   float4 (&r)[NL]) {
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int idx = threadIdx.x + NT * i;
      int row = row0 + pair_row(idx >> 3);
      row = row < nrows ? row : nrows - 1;
      r[i] = *reinterpret_cast<const float4*>(base + (int64_t)row * ld + k0 + (idx & 7) * 4);
    }
  }
  static __device__ __forceinline__ void load_any(const float* __restrict__ base, int64_t ld, int row0, int nrows, int k0, int K,
                                                  float4 (&r)[NL]) {
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int idx = threadIdx.x + NT * i;
      int row = row0 + pair_row(idx >> 3);
      row = row < nrows ? row : nrows - 1;
      const int k = k0 + (idx & 7) * 4;
      r[i] = *reinterpret_cast<const float4*>(base + (int64_t)row * ld + (k < K ? k : 0));
    }
  }
  static __device__ __forceinline__ float4 load_full1(const float* __restrict__ base, int64_t ld, int row0, int nrows, int k0, int i) {
    const int idx = threadIdx.x + NT * i;
    int row = row0 + pair_row(idx >> 3);
    row = row < nrows ? row : nrows - 1;
    return *reinterpret_cast<const float4*>(base + (int64_t)row * ld + k0 + (idx & 7) * 4);
  }
  static __device__ __forceinline__ float4 load_any1(const float* __restrict__ base, int64_t ld, int row0, int nrows, int k0, int K, int i) {
    const int idx = threadIdx.x + NT * i;
    int row = row0 + pair_row(idx >> 3);
    row = row < nrows ? row : nrows - 1;
    const int k = k0 + (idx & 7) * 4;
    return *reinterpret_cast<const float4*>(base + (int64_t)row * ld + (k < K ? k : 0));
  }
  static __device__ __forceinline__ void mask(float4 (&r)[NL], int k0, int K) {
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int k = k0 + ((threadIdx.x + NT * i) & 7) * 4;
      r[i].x = k + 0 < K ? r[i].x : 0.f;
      r[i].y = k + 1 < K ? r[i].y : 0.f;
      r[i].z = k + 2 < K ? r[i].z : 0.f;
      r[i].w = k + 3 < K ? r[i].w : 0.f;
    }
  }
  // pre-split operand (see Args): groups at or beyond K were fetched from column 0 and are dropped whole; a group that
  // straddles K is already zero padded in memory
  static __device__ __forceinline__ void mask_groups(float4 (&r)[NL], int k0, int K) {
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int k = k0 + ((threadIdx.x + NT * i) & 7) * 4;
      if (!(k < K)) r[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  static __device__ __forceinline__ void store_raw(_Float16* hi_img, _Float16* lo_img, const float4& v, int i) {
    const int idx = threadIdx.x + NT * i;
    const int off = pair_row(idx >> 3) * PITCH + (idx & 7) * 4;
    *reinterpret_cast<uint2*>(hi_img + off) = make_uint2(__float_as_uint(v.x), __float_as_uint(v.y));
    *reinterpret_cast<uint2*>(lo_img + off) = make_uint2(__float_as_uint(v.z), __float_as_uint(v.w));
  }
  // convert + store float4 #i (called between MFMAs)
  template <bool WIDE = false>
  static __device__ __forceinline__ void store_one(_Float16* hi_img, _Float16* lo_img, const float4& v, int i, float s) {
    const int idx = threadIdx.x + NT * i;
    const int off = pair_row(idx >> 3) * PITCH + (idx & 7) * 4;
    uint2 h, l;
    split4_pk<WIDE>(v, s, h, l);
    *reinterpret_cast<uint2*>(hi_img + off) = h;
    *reinterpret_cast<uint2*>(lo_img + off) = l;
  }
  template <bool RAW, bool WIDE = false>
  static __device__ __forceinline__ void put(_Float16* hi_img, _Float16* lo_img, const float4& v, int i, float s) {
    if constexpr (RAW) store_raw(hi_img, lo_img, v, i); else store_one<WIDE>(hi_img, lo_img, v, i, s);
  }
  template <bool RAW>
  static __device__ __forceinline__ void mask_as(float4 (&r)[NL], int k0, int K) {
    if constexpr (RAW) mask_groups(r, k0, K); else mask(r, k0, K);
  }
};

// Epilogue through LDS: an accumulator register holds one element of 32 different... columns of one row
// (32 lanes x 4 bytes), which stores as 128-byte pieces.  Each wave parks a 32 x 64 half of its tile in its
// own LDS slab (pitch 68 floats) and writes it out as whole 256-byte row segments with 16-byte stores.
// (All waves passed the last barrier of the K loop, so the stage buffers are free; slabs are wave-private.)
// Optional terms, in this order: exact fp32 rank-J update, bias, activation.
// sum over the 16 lanes of a DPP row, result in every lane: xor 1, xor 2 (quad permutes), then the two mirrors.
// The empty asm statements keep hipcc from pairing two such chains into v_pk_add_f32 / v_pk_fma_f32: a DPP (or
// ds_bpermute) read of a register written by a PACKED fp32 op got only the two wait states of a single-pass op and
// lanes 48-63 - written in the op's last pass - were read early: a few wrong sums per million, different every run
// (ROCm 7.2, gfx950; seen with the 256-thread kernel).
__device__ __forceinline__ float row16_sum(float x) {
  asm volatile("" : "+v"(x));
  x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
  asm volatile("" : "+v"(x));
  x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
  asm volatile("" : "+v"(x));
  x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x141, 0xF, 0xF, true));   // row_half_mirror
  asm volatile("" : "+v"(x));
  x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x140, 0xF, 0xF, true));   // row_mirror
  asm volatile("" : "+v"(x));
  return x;
}

template <bool DROP, class ARGS, int MI>   // DROP: the dropout fields of ARGS are honoured (single-product kernels only: the
                                           // pair kernels hold two argument sets in SGPRs and have none to spare)
__device__ __forceinline__ void store_tile_through_lds(const ARGS& a, f32x16 (&acc)[MI][2], _Float16* smem, int row0, int col0,
                                                       int wave, int lane, int wm, int wn, float alpha) {
  const int fr = lane & 31, fh = lane >> 5;
  constexpr int EP = 68;
  float* slab = reinterpret_cast<float*>(smem) + wave * (32 * EP);
  const bool vec_ok = (a.ldc & 3) == 0 && (reinterpret_cast<uintptr_t>(a.C) & 15) == 0;
  // Everything that depends only on the lane's column group is fetched once, and the rank-J row factors of a
  // 32-row half are fetched together: as loads inside the store loop they cost one dependent round trip per
  // iteration (bias alone: 55 us of a 330 us product).
  const int r_in = lane >> 4, c4 = (lane & 15) * 4;
  const int col = col0 + wn * 64 + c4;
  float4 bq = make_float4(0.f, 0.f, 0.f, 0.f);
  if (a.bias) {
    bq.x = col + 0 < a.N ? a.bias[col + 0] : 0.f; bq.y = col + 1 < a.N ? a.bias[col + 1] : 0.f;
    bq.z = col + 2 < a.N ? a.bias[col + 2] : 0.f; bq.w = col + 3 < a.N ? a.bias[col + 3] : 0.f;
  }
  const bool use_sc = a.sc_out != nullptr && col < a.sc_cols;          // wave-uniform: a wave's 64 columns are one block
  float4 sl = make_float4(0.f, 0.f, 0.f, 0.f), sr = sl;
  if (use_sc) { sl = *reinterpret_cast<const float4*>(a.sc_l + col); sr = *reinterpret_cast<const float4*>(a.sc_r + col); }
  const bool use_j = a.J > 0 && col + 3 < a.ldv;          // exact fp32 rank-J term (V rows are zero padded)
  const bool small_j = a.J <= 4;
  float4 wv[4];
#pragma unroll
  for (int jj = 0; jj < 4; ++jj)
    wv[jj] = (use_j && small_j && jj < a.J) ? *reinterpret_cast<const float4*>(a.V + (int64_t)jj * a.ldv + col)
                                            : make_float4(0.f, 0.f, 0.f, 0.f);
  float amx = 0.f;                                       // max |stored value| of this lane (a.absmax)
  bool use_drop = false;                                 // block-uniform
  uint64_t dseed = 0; float dp = 0.f, dinv = 1.f;
  if constexpr (DROP) {
    use_drop = a.drop_p > 0.f;
    if (use_drop) { dseed = a.drop_seed + (a.drop_seed_off ? a.drop_seed_off[0] : 0); dp = a.drop_p; dinv = a.drop_inv; }
  }
  // one 32-row half of the wave's tile; called with a literal i per half (a loop over i is not unrolled once its body has
  // two large paths, and acc[i] with a run-time i puts the accumulators in scratch memory: 7x slower)
  auto do_half = [&](const int i) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e)
        slab[((e & 3) + 8 * (e >> 2) + 4 * fh) * EP + j * 32 + fr] = acc[i][j][e] * alpha;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // the slab is exchanged between the lanes of this wave
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // Interior halves (every row and column of the wave's 32 x 64 piece exists, 16-byte stores possible) take a
    // straight-line path: no per-row / per-column tests, the activation chosen once.  The general path below compiled
    // to ~17 000 instructions of exec-mask branches per kernel and cost ~17 us per 256 x 256 tile (30 % of a
    // 12-stage tile: timing-only build without the stores).
    const bool interior = vec_ok && row0 + wm * (32 * MI) + i * 32 + 32 <= a.M && col0 + wn * 64 + 64 <= a.N;
    if (interior && !(use_j && !small_j)) {
      float4 vv[8];
#pragma unroll
      for (int it = 0; it < 8; ++it) vv[it] = *reinterpret_cast<const float4*>(slab + (it * 4 + r_in) * EP + c4);
      if (use_j) {
#pragma unroll
        for (int it = 0; it < 8; ++it) {
          const int row = row0 + wm * (32 * MI) + i * 32 + it * 4 + r_in;
          float u4[4];
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) u4[jj] = a.U[(int64_t)row * a.ldu + (jj < a.J ? jj : 0)];
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            vv[it].x = fmaf(u4[jj], wv[jj].x, vv[it].x); vv[it].y = fmaf(u4[jj], wv[jj].y, vv[it].y);
            vv[it].z = fmaf(u4[jj], wv[jj].z, vv[it].z); vv[it].w = fmaf(u4[jj], wv[jj].w, vv[it].w);
          }
        }
      }
      if (use_sc) {                                   // scores from the raw product (before bias / activation)
#pragma unroll
        for (int it = 0; it < 8; ++it) {
          const int row = row0 + wm * (32 * MI) + i * 32 + it * 4 + r_in;
          const float4 v = *reinterpret_cast<const float4*>(slab + (it * 4 + r_in) * EP + c4);
          float pl = v.x * sl.x + v.y * sl.y + v.z * sl.z + v.w * sl.w;
          float pr = v.x * sr.x + v.y * sr.y + v.z * sr.z + v.w * sr.w;
          pl = row16_sum(pl); pr = row16_sum(pr);
          if ((lane & 15) == 0)
            *reinterpret_cast<float2*>(a.sc_out + ((int64_t)row * (a.sc_cols >> 6) + (col >> 6)) * 2) = make_float2(pl, pr);
        }
      }
#pragma unroll
      for (int it = 0; it < 8; ++it) { vv[it].x += bq.x; vv[it].y += bq.y; vv[it].z += bq.z; vv[it].w += bq.w; }
      if (a.mean_other && !a.mean_out) {                // addend: the other product of the pair
        const int64_t r0_ = row0 + wm * (32 * MI) + i * 32 + r_in;
        float4 oo[8];
#pragma unroll
        for (int it = 0; it < 8; ++it) oo[it] = *reinterpret_cast<const float4*>(a.mean_other + (r0_ + it * 4) * a.ld_mo + col);
#pragma unroll
        for (int it = 0; it < 8; ++it) { vv[it].x += oo[it].x; vv[it].y += oo[it].y; vv[it].z += oo[it].z; vv[it].w += oo[it].w; }
      }
      if (a.act == SPGNN_ACT_ELU) {
#pragma unroll
        for (int it = 0; it < 8; ++it) {
          vv[it].x = elu_fwd_nb(vv[it].x); vv[it].y = elu_fwd_nb(vv[it].y); vv[it].z = elu_fwd_nb(vv[it].z); vv[it].w = elu_fwd_nb(vv[it].w);
        }
      } else if (a.act == SPGNN_ACT_TANH) {
#pragma unroll
        for (int it = 0; it < 8; ++it) {
          vv[it].x = tanhf(vv[it].x); vv[it].y = tanhf(vv[it].y); vv[it].z = tanhf(vv[it].z); vv[it].w = tanhf(vv[it].w);
        }
      } else if (a.act == SPGNN_ACT_RELU) {
#pragma unroll
        for (int it = 0; it < 8; ++it) {
          vv[it].x = fmaxf(vv[it].x, 0.f); vv[it].y = fmaxf(vv[it].y, 0.f); vv[it].z = fmaxf(vv[it].z, 0.f); vv[it].w = fmaxf(vv[it].w, 0.f);
        }
      } else if (a.act == SPGNN_ACT_LRELU) {
#pragma unroll
        for (int it = 0; it < 8; ++it) {
          vv[it].x = vv[it].x > 0.f ? vv[it].x : 0.01f * vv[it].x; vv[it].y = vv[it].y > 0.f ? vv[it].y : 0.01f * vv[it].y;
          vv[it].z = vv[it].z > 0.f ? vv[it].z : 0.01f * vv[it].z; vv[it].w = vv[it].w > 0.f ? vv[it].w : 0.01f * vv[it].w;
        }
      }
      if (use_drop) {
        const int64_t r0_ = row0 + wm * (32 * MI) + i * 32 + r_in;
#pragma unroll
        for (int it = 0; it < 8; ++it) {
          const float4 k = spgnn_detail::feat_keep4(dseed, (r0_ + it * 4) * a.N + col, dp, dinv);
          vv[it].x *= k.x; vv[it].y *= k.y; vv[it].z *= k.z; vv[it].w *= k.w;
        }
      }
      if (a.mean_out) {
        const int64_t r0_ = row0 + wm * (32 * MI) + i * 32 + r_in;
        float4 oo[8];
#pragma unroll
        for (int it = 0; it < 8; ++it) oo[it] = *reinterpret_cast<const float4*>(a.mean_other + (r0_ + it * 4) * a.ld_mo + col);
#pragma unroll
        for (int it = 0; it < 8; ++it) {
          const float4 m = make_float4(0.5f * (vv[it].x + oo[it].x), 0.5f * (vv[it].y + oo[it].y), 0.5f * (vv[it].z + oo[it].z),
                                       0.5f * (vv[it].w + oo[it].w));
          *reinterpret_cast<float4*>(a.mean_out + (r0_ + it * 4) * a.ld_mn + col) = m;
        }
      }
      float* dst0 = a.C + (int64_t)(row0 + wm * (32 * MI) + i * 32 + r_in) * a.ldc + col;
#pragma unroll
      for (int it = 0; it < 8; ++it)
        amx = fmaxf(amx, fmaxf(fmaxf(fabsf(vv[it].x), fabsf(vv[it].y)), fmaxf(fabsf(vv[it].z), fabsf(vv[it].w))));
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        // streaming store: the (M, N) result is far larger than the caches and is next read by another kernel; without the
        // temporal hint the store bursts at the end of every round of tiles are 2-5 % of a product (timing-only build
        // without the stores: 515 -> 483 us at K = 1063, 217 -> 186 at K = 384; with this hint 504 / 206; the consumers'
        // times do not move: tools/bench_variants.sh)
        typedef float f4v __attribute__((ext_vector_type(4)));
        __builtin_nontemporal_store(f4v{vv[it].x, vv[it].y, vv[it].z, vv[it].w}, reinterpret_cast<f4v*>(dst0 + (int64_t)(it * 4) * a.ldc));
      }
    } else {
    float uu[8][4];
    if (use_j && small_j) {
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        int row = row0 + wm * (32 * MI) + i * 32 + it * 4 + r_in;
        row = row < a.M ? row : a.M - 1;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) uu[it][jj] = a.U[(int64_t)row * a.ldu + (jj < a.J ? jj : 0)];
      }
    }
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int lr = it * 4 + r_in;
      const int row = row0 + wm * (32 * MI) + i * 32 + lr;
      float4 v = *reinterpret_cast<const float4*>(slab + lr * EP + c4);
      if (row < a.M) {
        if (use_j) {
          if (small_j) {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
              const float u = uu[it][jj];
              v.x = fmaf(u, wv[jj].x, v.x); v.y = fmaf(u, wv[jj].y, v.y); v.z = fmaf(u, wv[jj].z, v.z); v.w = fmaf(u, wv[jj].w, v.w);
            }
          } else {
            const float* urow = a.U + (int64_t)row * a.ldu;
            for (int jj = 0; jj < a.J; ++jj) {
              const float u = urow[jj];
              const float4 w = *reinterpret_cast<const float4*>(a.V + (int64_t)jj * a.ldv + col);
              v.x = fmaf(u, w.x, v.x); v.y = fmaf(u, w.y, v.y); v.z = fmaf(u, w.z, v.z); v.w = fmaf(u, w.w, v.w);
            }
          }
        }
        v.x += bq.x; v.y += bq.y; v.z += bq.z; v.w += bq.w;
        if (a.mean_other && !a.mean_out) {
          const float* op = a.mean_other + (int64_t)row * a.ld_mo + col;
          if (col < a.N) v.x += op[0];
          if (col + 1 < a.N) v.y += op[1];
          if (col + 2 < a.N) v.z += op[2];
          if (col + 3 < a.N) v.w += op[3];
        }
        if (a.act == SPGNN_ACT_ELU) {
          v.x = elu_fwd_nb(v.x); v.y = elu_fwd_nb(v.y); v.z = elu_fwd_nb(v.z); v.w = elu_fwd_nb(v.w);
        } else if (a.act == SPGNN_ACT_TANH) {
          v.x = tanhf(v.x); v.y = tanhf(v.y); v.z = tanhf(v.z); v.w = tanhf(v.w);
        } else if (a.act == SPGNN_ACT_RELU) {
          v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        } else if (a.act == SPGNN_ACT_LRELU) {
          v.x = v.x > 0.f ? v.x : 0.01f * v.x; v.y = v.y > 0.f ? v.y : 0.01f * v.y;
          v.z = v.z > 0.f ? v.z : 0.01f * v.z; v.w = v.w > 0.f ? v.w : 0.01f * v.w;
        }
        if (use_drop) {
          const float4 k = spgnn_detail::feat_keep4(dseed, (int64_t)row * a.N + col, dp, dinv);
          v.x *= k.x; v.y *= k.y; v.z *= k.z; v.w *= k.w;
        }
        if (a.mean_out) {
          const float vq[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int q_ = 0; q_ < 4; ++q_)
            if (col + q_ < a.N)
              a.mean_out[(int64_t)row * a.ld_mn + col + q_] = 0.5f * (vq[q_] + a.mean_other[(int64_t)row * a.ld_mo + col + q_]);
        }
        float* dst = a.C + (int64_t)row * a.ldc + col;
        amx = fmaxf(amx, fmaxf(fmaxf(col < a.N ? fabsf(v.x) : 0.f, col + 1 < a.N ? fabsf(v.y) : 0.f),
                               fmaxf(col + 2 < a.N ? fabsf(v.z) : 0.f, col + 3 < a.N ? fabsf(v.w) : 0.f)));
        if (vec_ok && col + 3 < a.N) *reinterpret_cast<float4*>(dst) = v;
        else {
          if (col < a.N) dst[0] = v.x;
          if (col + 1 < a.N) dst[1] = v.y;
          if (col + 2 < a.N) dst[2] = v.z;
          if (col + 3 < a.N) dst[3] = v.w;
        }
      }
    }
    if (use_sc) {                                     // 16 lanes hold one row's 64 columns: reduce across them
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int lr = it * 4 + r_in;
        const int row = row0 + wm * (32 * MI) + i * 32 + lr;
        const float4 v = *reinterpret_cast<const float4*>(slab + lr * EP + c4);
        float pl = v.x * sl.x + v.y * sl.y + v.z * sl.z + v.w * sl.w;
        float pr = v.x * sr.x + v.y * sr.y + v.z * sr.z + v.w * sr.w;
        pl = row16_sum(pl); pr = row16_sum(pr);
        if ((lane & 15) == 0 && row < a.M)
          *reinterpret_cast<float2*>(a.sc_out + ((int64_t)row * (a.sc_cols >> 6) + (col >> 6)) * 2) = make_float2(pl, pr);
      }
    }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // all reads of this half done before it is overwritten
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  do_half(0);
  do_half(1);
  if constexpr (MI == 4) { do_half(2); do_half(3); }
  static_assert(MI == 2 || MI == 4, "wave tile is 64 or 128 rows");
  if (a.absmax) {                                        // block-uniform
    for (int off = 32; off > 0; off >>= 1) amx = fmaxf(amx, __shfl_xor(amx, off, 64));
    if (lane == 0) spgnn_detail::slots_max(a.absmax, amx, (unsigned)((row0 >> 5) + (col0 >> 6) + wave));
  }
}

template <int WM, bool APS, bool BPS, bool DROP, bool WIDE>   // APS / BPS: the A / B operand arrives pre-split (see Args); WIDE: split4_pk
__device__ __forceinline__ void nt_v2_body(const Args& a, const unsigned b, const unsigned nb) {   // workgroup b of the nb this product owns
  constexpr int TBM = 64 * WM, NT = 128 * WM;
  constexpr int A_IMG = TBM * PITCH, B_IMG = BN * PITCH;
  constexpr int STAGE = 2 * A_IMG + 2 * B_IMG;                 // halves per stage: Ah | Al | Bh | Bl
  extern __shared__ __attribute__((aligned(16))) _Float16 smem[];
  using AIO = TileIO<TBM, NT>;
  using BIO = TileIO<BN, NT>;
  constexpr int NLA = AIO::NL, NLB = BIO::NL;

  const unsigned tile = (b & 7u) * (nb >> 3) + (b >> 3);
  if (tile >= (unsigned)(a.nbm * a.nbn)) return;
  const int bm = tile / a.nbn, bn = tile % a.nbn;
  const int row0 = bm * TBM, col0 = bn * BN;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 31, fh = lane >> 5;
  const float sA = spgnn_detail::load_scale_monitored(a.sA), sB = spgnn_detail::load_scale_monitored(a.sB);

  f32x16 acc[2][2], acc2[2][2];                                // acc2 (WIDE only): the cross products, carrying 2^11
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) { acc[i][j][e] = 0.f; acc2[i][j][e] = 0.f; }

  const int nk = (a.K + BK - 1) / BK;
  float4 ra0[NLA], rb0[NLB], ra1[NLA], rb1[NLB];             // two prefetch sets (stage parity)

  // prologue: stage 0 -> LDS buffer 0, stage 1 -> register set 1 (unconditional loads, see TileIO)
  AIO::load_any(a.A, a.lda, row0, a.M, 0, a.K, ra0);
  BIO::load_any(a.B, a.ldb, col0, a.N, 0, a.K, rb0);
  AIO::load_any(a.A, a.lda, row0, a.M, BK, a.K, ra1);
  BIO::load_any(a.B, a.ldb, col0, a.N, BK, a.K, rb1);
  {
    _Float16* st = smem;
    AIO::template mask_as<APS>(ra0, 0, a.K);
    BIO::template mask_as<BPS>(rb0, 0, a.K);
#pragma unroll
    for (int i = 0; i < NLA; ++i) AIO::template put<APS, WIDE>(st, st + A_IMG, ra0[i], i, sA);
#pragma unroll
    for (int i = 0; i < NLB; ++i) BIO::template put<BPS, WIDE>(st + 2 * A_IMG, st + 2 * A_IMG + B_IMG, rb0[i], i, sB);
  }
  __syncthreads();

  // one stage: MFMAs on buffer PAR_ (= T_ & 1); convert register set (RA, RB) = stage T_+1 into the other buffer;
  // then refill that set with stage T_+3 (stage T_+2 lives in the other set).
  // STEADY_ = 1: stages T_+1 and T_+3 exist and lie wholly below K: no tests, no masks.
  // STEADY_ = 0: the last stages: the set is masked beyond K first, the refill is a harmless load_any.
#define SPGNN_STAGE(T_, PAR_, RA, RB, STEADY_)                                                               \
  {                                                                                                          \
    const _Float16* cb = smem + (PAR_) * STAGE;                                                              \
    _Float16* nbuf = smem + (1 - (PAR_)) * STAGE;                                                            \
    const bool has_next = (STEADY_) || (T_) + 1 < nk;                                                        \
    if (!(STEADY_) && has_next) {                                                                            \
      AIO::template mask_as<APS>(RA, ((T_) + 1) * BK, a.K);                                                  \
      BIO::template mask_as<BPS>(RB, ((T_) + 1) * BK, a.K);                                                                  \
    }                                                                                                        \
    _Pragma("unroll") for (int ks = 0; ks < BK / 16; ++ks) {                                                 \
      half8 ah[2], al[2], bh[2], bl[2];                                                                      \
      _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                        \
        const int off = (wm * 64 + i * 32 + fr) * PITCH + ks * 16 + fh * 8;                                  \
        ah[i] = *reinterpret_cast<const half8*>(cb + off);                                                   \
        al[i] = *reinterpret_cast<const half8*>(cb + A_IMG + off);                                           \
      }                                                                                                      \
      _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                        \
        const int off = (wn * 64 + j * 32 + fr) * PITCH + ks * 16 + fh * 8;                                  \
        bh[j] = *reinterpret_cast<const half8*>(cb + 2 * A_IMG + off);                                       \
        bl[j] = *reinterpret_cast<const half8*>(cb + 2 * A_IMG + B_IMG + off);                               \
      }                                                                                                      \
      /* product-major order: the four accumulators take turns, so no MFMA waits on the one issued just before it */ \
      _Pragma("unroll") for (int c = 0; c < 12; ++c) {                                                       \
        const int pr = c >> 2, ij = c & 3;                                                                   \
        const int i = ij >> 1, j = ij & 1;                                                                   \
        if (WIDE && pr != 2)                                                                                 \
          acc2[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pr == 0 ? al[i] : ah[i], pr == 1 ? bl[j] : bh[j], acc2[i][j], 0, 0, 0); \
        else                                                                                                 \
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pr == 0 ? al[i] : ah[i], pr == 1 ? bl[j] : bh[j], acc[i][j], 0, 0, 0); \
        if (has_next && c % 3 == 2) { /* a slice of next stage's conversion after every third MFMA */        \
          const int slot = ks * 4 + c / 3;                                                                   \
          _Pragma("unroll") for (int q = 0; q < NLA; ++q)                                                    \
            if (q * 8 / NLA == slot || (NLA > 8 && q % 8 == slot))                                           \
              AIO::template put<APS, WIDE>(nbuf, nbuf + A_IMG, RA[q], q, sA);                                \
          _Pragma("unroll") for (int q = 0; q < NLB; ++q)                                                    \
            if (q * 8 / NLB == slot || (NLB > 8 && q % 8 == slot))                                           \
              BIO::template put<BPS, WIDE>(nbuf + 2 * A_IMG, nbuf + 2 * A_IMG + B_IMG, RB[q], q, sB);              \
        }                                                                                                    \
      }                                                                                                      \
    }                                                                                                        \
    if (STEADY_) {                                                                                           \
      AIO::load_full(a.A, a.lda, row0, a.M, ((T_) + 3) * BK, RA);                                            \
      BIO::load_full(a.B, a.ldb, col0, a.N, ((T_) + 3) * BK, RB);                                            \
    } else {                                                                                                 \
      AIO::load_any(a.A, a.lda, row0, a.M, ((T_) + 3) * BK, a.K, RA);                                        \
      BIO::load_any(a.B, a.ldb, col0, a.N, ((T_) + 3) * BK, a.K, RB);                                        \
    }                                                                                                        \
    __syncthreads();                                                                                         \
    __builtin_amdgcn_sched_barrier(0);   /* keep the next stage's conversion arithmetic (and its vmcnt) behind the barrier */ \
  }

  // register set 1 holds stage 1 (odd stages), set 0 will hold stage 2 (even stages)
  AIO::load_any(a.A, a.lda, row0, a.M, 2 * BK, a.K, ra0);
  BIO::load_any(a.B, a.ldb, col0, a.N, 2 * BK, a.K, rb0);
  const int nfull = a.K / BK;                                   // stages that lie wholly below K
  int t = 0;
  for (; t + 4 < nfull; t += 2) {                               // refill targets t+3, t+4 are full stages
    SPGNN_STAGE(t, 0, ra1, rb1, 1)     // computes stage t (even), converts stage t+1 from set 1, refills set 1 with t+3
    SPGNN_STAGE(t + 1, 1, ra0, rb0, 1) // computes stage t+1 (odd), converts stage t+2 from set 0, refills set 0 with t+4
  }
  for (; t + 1 < nk; t += 2) {
    SPGNN_STAGE(t, 0, ra1, rb1, 0)
    SPGNN_STAGE(t + 1, 1, ra0, rb0, 0)
  }
  if (t < nk) SPGNN_STAGE(t, 0, ra1, rb1, 0)
#undef SPGNN_STAGE

  if constexpr (WIDE) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = fmaf(acc2[i][j][e], kWideLoInv, acc[i][j][e]);
  }
  store_tile_through_lds<DROP>(a, acc, smem, row0, col0, wave, lane, wm, wn, 1.f / (sA * sB));
}

// Two independent products in ONE launch (spgnn_gemm_nt_pair): workgroups [0, nb0) run product 0, the rest product 1,
// each with the tile order it has alone.  A GATConv level's structure and position projections (reference
// models.py:472-484) are such a pair: the small one's memory-bound tiles run in the CUs the large one's last round
// leaves idle instead of in a launch of their own.
struct PairArgs { Args p[2]; unsigned nb0; };

// (WIDE: two accumulator sets - 128 registers - beside the staging sets: one workgroup per CU, so that the accumulators can live in AGPRs)
template <int WM, bool APS, bool BPS, bool WIDE>
__global__ __launch_bounds__(128 * WM, (WM == 2 && !WIDE) ? 2 : 1) void gemm_nt_f16x3_v2(Args a) {
  nt_v2_body<WM, APS, BPS, true, WIDE>(a, blockIdx.x, gridDim.x);
}
template <int WM, bool APS, bool BPS, bool WIDE>
__global__ __launch_bounds__(128 * WM, (WM == 2 && !WIDE) ? 2 : 1) void gemm_nt_pair_v2(PairArgs pa) {
  const bool second = blockIdx.x >= pa.nb0;                 // block-uniform: the arguments are read from one half of the kernarg
  nt_v2_body<WM, APS, BPS, false, WIDE>(pa.p[second ? 1 : 0], second ? blockIdx.x - pa.nb0 : blockIdx.x, second ? gridDim.x - pa.nb0 : pa.nb0);
}

// -------------------------------------------------------------------------------------------------
// NT kernel, third generation: 256 x 256 block tile, 8 waves as 2 (M) x 4 (N), each wave 128 x 64 = 4 x 2 MFMA
// tiles (128 accumulator registers), one workgroup per CU.  Per MFMA it moves 2/3 of the operand bytes, converts
// 2/3 of the values, reads 3/4 of the LDS fragments and meets half the barriers of the 256 x 128 kernel above.
// ONE register set: a staging register is refilled (stage t+2) right after it has been converted (stage t+1) between
// the MFMAs of stage t, so every load has a whole stage to land; fragments are read per k16 step in two halves
// (B once, A for two of the four row tiles at a time) to stay inside 256 registers.  LDS: 2 stages x 4 images of
// 256 rows x 80 bytes = 160 KB.
// -------------------------------------------------------------------------------------------------
template <bool APS, bool BPS, bool DROP>
__device__ __forceinline__ void nt_v3_body(const Args& a, const unsigned b, const unsigned nb) {
  constexpr int TB = 256;
  constexpr int IMG = TB * PITCH;                               // halves per image
  constexpr int STAGE = 4 * IMG;                                // Ah | Al | Bh | Bl
  extern __shared__ __attribute__((aligned(16))) _Float16 smem[];

  const unsigned tile = (b & 7u) * (nb >> 3) + (b >> 3);
  if (tile >= (unsigned)(a.nbm * a.nbn)) return;
  const int bm = tile / a.nbn, bn = tile % a.nbn;
  const int row0 = bm * TB, col0 = bn * TB;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  const int fr = lane & 31, fh = lane >> 5;
  const float sA = spgnn_detail::load_scale_monitored(a.sA), sB = spgnn_detail::load_scale_monitored(a.sB);

  // Operand tiles come through buffer descriptors: one 32-bit per-thread offset per operand, everything else (tile
  // row, staging register, stage) in the scalar offset; rows past the end read as zero (no clamps, no 64-bit address
  // arithmetic: 14 fewer registers than eight flat pointers).  The host guarantees the byte extents fit 31 bits.
  // Thread t's float4 #i of a 256 x 32 tile: row pair_row(t >> 3) + 64 i, columns 4 (t & 7) .. + 3.
  const int kq = (threadIdx.x & 7) * 4, rq = pair_row(threadIdx.x >> 3);
  const auto rsA = __builtin_amdgcn_make_buffer_rsrc((void*)a.A, (short)0, (int)((((int64_t)a.M - 1) * a.lda + ((a.K + 3) & ~3)) * 4), 0x00020000);
  const auto rsB = __builtin_amdgcn_make_buffer_rsrc((void*)a.B, (short)0, (int)((((int64_t)a.N - 1) * a.ldb + ((a.K + 3) & ~3)) * 4), 0x00020000);
  const int vA = (rq * (int)a.lda + kq) * 4, vB = (rq * (int)a.ldb + kq) * 4;
  const int sA0 = row0 * (int)a.lda * 4, sB0 = col0 * (int)a.ldb * 4;
  const int stepA = 64 * (int)a.lda * 4, stepB = 64 * (int)a.ldb * 4;
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define SPGNN_LDA(I_, K0_) __builtin_amdgcn_raw_buffer_load_b128(rsA, vA, sA0 + (I_) * stepA + (K0_) * 4, 0)
#define SPGNN_LDB(I_, K0_) __builtin_amdgcn_raw_buffer_load_b128(rsB, vB, sB0 + (I_) * stepB + (K0_) * 4, 0)
  const int woff = rq * PITCH + kq;                            // LDS store offset (halves) of staging register 0
#define SPGNN_CVT(IMG0_, R_, I_, S_)                                                                         \
  {                                                                                                          \
    uint2 h_, l_;                                                                                            \
    split4_pk(make_float4(__uint_as_float(R_[0]), __uint_as_float(R_[1]), __uint_as_float(R_[2]), __uint_as_float(R_[3])), S_, h_, l_); \
    *reinterpret_cast<uint2*>((IMG0_) + woff + (I_) * 64 * PITCH) = h_;                                      \
    *reinterpret_cast<uint2*>((IMG0_) + IMG + woff + (I_) * 64 * PITCH) = l_;                                \
  }
#define SPGNN_RAW(IMG0_, R_, I_)                                                                             \
  {                                                                                                          \
    *reinterpret_cast<uint2*>((IMG0_) + woff + (I_) * 64 * PITCH) = make_uint2(R_[0], R_[1]);                \
    *reinterpret_cast<uint2*>((IMG0_) + IMG + woff + (I_) * 64 * PITCH) = make_uint2(R_[2], R_[3]);          \
  }
#define SPGNN_PUTB(IMG0_, R_, I_)                                                                            \
  { if constexpr (BPS) SPGNN_RAW(IMG0_, R_, I_) else SPGNN_CVT(IMG0_, R_, I_, sB) }
#define SPGNN_PUTA(IMG0_, R_, I_)                                                                            \
  { if constexpr (APS) SPGNN_RAW(IMG0_, R_, I_) else SPGNN_CVT(IMG0_, R_, I_, sA) }
#define SPGNN_MASKG(R_, K0_)                                                                                 \
  {                                                                                                          \
    if (!((K0_) + kq < a.K)) { _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) R_[q_] = u32x4{0u, 0u, 0u, 0u}; } \
  }
#define SPGNN_MASK(R_, K0_)                                                                                  \
  {                                                                                                          \
    const int k_ = (K0_) + kq;                                                                               \
    _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) {                                                       \
      R_[q_][0] = k_ + 0 < a.K ? R_[q_][0] : 0u; R_[q_][1] = k_ + 1 < a.K ? R_[q_][1] : 0u;                  \
      R_[q_][2] = k_ + 2 < a.K ? R_[q_][2] : 0u; R_[q_][3] = k_ + 3 < a.K ? R_[q_][3] : 0u;                  \
    }                                                                                                        \
  }

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int nk = (a.K + BK - 1) / BK;
  u32x4 ra[4], rb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) ra[i] = SPGNN_LDA(i, 0);
#pragma unroll
  for (int i = 0; i < 4; ++i) rb[i] = SPGNN_LDB(i, 0);
  if constexpr (APS) SPGNN_MASKG(ra, 0) else SPGNN_MASK(ra, 0)
  if constexpr (BPS) SPGNN_MASKG(rb, 0) else SPGNN_MASK(rb, 0)
#pragma unroll
  for (int i = 0; i < 4; ++i) SPGNN_PUTA(smem, ra[i], i)
#pragma unroll
  for (int i = 0; i < 4; ++i) SPGNN_PUTB(smem + 2 * IMG, rb[i], i)
#pragma unroll
  for (int i = 0; i < 4; ++i) ra[i] = SPGNN_LDA(i, BK);
#pragma unroll
  for (int i = 0; i < 4; ++i) rb[i] = SPGNN_LDB(i, BK);
  __syncthreads();
  __builtin_amdgcn_sched_barrier(0);

  // stage T_: MFMAs on buffer PAR_; the register set (stage T_+1) is converted into the other buffer, one float4 per
  // six MFMAs, and each register is refilled with stage T_+2 as soon as it is free (loads past K or past the last
  // row return in-bounds garbage or zeros; the tail stages mask beyond K).  STEADY_ as in the kernel above.
#define SPGNN_STAGE3(T_, PAR_, STEADY_)                                                                      \
  {                                                                                                          \
    const _Float16* cb = smem + (PAR_) * STAGE;                                                              \
    _Float16* nbuf = smem + (1 - (PAR_)) * STAGE;                                                            \
    const bool has_next = (STEADY_) || (T_) + 1 < nk;                                                        \
    if (!(STEADY_) && has_next) {                                                                            \
      if constexpr (APS) SPGNN_MASKG(ra, ((T_) + 1) * BK) else SPGNN_MASK(ra, ((T_) + 1) * BK)               \
      if constexpr (BPS) SPGNN_MASKG(rb, ((T_) + 1) * BK) else SPGNN_MASK(rb, ((T_) + 1) * BK)               \
    }                                                                                                        \
    _Pragma("unroll") for (int ks = 0; ks < BK / 16; ++ks) {                                                 \
      half8 bh[2], bl[2];                                                                                    \
      _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                        \
        const int off = (wn * 64 + j * 32 + fr) * PITCH + ks * 16 + fh * 8;                                  \
        bh[j] = *reinterpret_cast<const half8*>(cb + 2 * IMG + off);                                         \
        bl[j] = *reinterpret_cast<const half8*>(cb + 3 * IMG + off);                                         \
      }                                                                                                      \
      _Pragma("unroll") for (int h = 0; h < 2; ++h) {                                                        \
        half8 ah[2], al[2];                                                                                  \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                      \
          const int off = (wm * 128 + (2 * h + i) * 32 + fr) * PITCH + ks * 16 + fh * 8;                     \
          ah[i] = *reinterpret_cast<const half8*>(cb + off);                                                 \
          al[i] = *reinterpret_cast<const half8*>(cb + IMG + off);                                           \
        }                                                                                                    \
        _Pragma("unroll") for (int c = 0; c < 12; ++c) {                   /* product-major over four tiles */ \
          const int pr = c >> 2, i = (c >> 1) & 1, j = c & 1;                                                \
          acc[2 * h + i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pr == 0 ? al[i] : ah[i], pr == 1 ? bl[j] : bh[j], \
                                                                     acc[2 * h + i][j], 0, 0, 0);            \
          if (has_next && c % 6 == 5) {                                                                      \
            /* pin the slot: hipcc otherwise hoists the first multiply of every staging register to the top of */ \
            /* the stage and waits vmcnt(0) there - for loads that were issued moments ago                      */ \
            __builtin_amdgcn_sched_barrier(0);                                                               \
            const int slot = ks * 4 + h * 2 + c / 6;                       /* 8 slots = 4 A + 4 B registers */ \
            const int k2 = ((T_) + 2) * BK;                                                                  \
            if (slot < 4) {                                                                                  \
              SPGNN_PUTA(nbuf, ra[slot], slot)                                                               \
              ra[slot] = SPGNN_LDA(slot, k2);                                                                \
            } else {                                                                                         \
              SPGNN_PUTB(nbuf + 2 * IMG, rb[slot - 4], slot - 4)                                             \
              rb[slot - 4] = SPGNN_LDB(slot - 4, k2);                                                        \
            }                                                                                                \
          }                                                                                                  \
        }                                                                                                    \
      }                                                                                                      \
    }                                                                                                        \
    __syncthreads();                                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
  }
  const int nfull = a.K / BK;
  int t = 0;
  for (; t + 3 < nfull; t += 2) {                                 // stages t+1 .. t+3 lie wholly below K
    SPGNN_STAGE3(t, 0, 1)
    SPGNN_STAGE3(t + 1, 1, 1)
  }
  for (; t + 1 < nk; t += 2) {
    SPGNN_STAGE3(t, 0, 0)
    SPGNN_STAGE3(t + 1, 1, 0)
  }
  if (t < nk) SPGNN_STAGE3(t, 0, 0)
#undef SPGNN_STAGE3
#undef SPGNN_LDA
#undef SPGNN_LDB
#undef SPGNN_CVT
#undef SPGNN_MASK
#undef SPGNN_MASKG
#undef SPGNN_PUTB
#undef SPGNN_PUTA
#undef SPGNN_RAW

  store_tile_through_lds<DROP>(a, acc, smem, row0, col0, wave, lane, wm, wn, 1.f / (sA * sB));
}
template <bool APS, bool BPS>
__global__ __launch_bounds__(512) void gemm_nt_f16x3_v3(Args a) {
  nt_v3_body<APS, BPS, true>(a, blockIdx.x, gridDim.x);
}
template <bool APS, bool BPS>
__global__ __launch_bounds__(512) void gemm_nt_pair_v3(PairArgs pa) {
  const bool second = blockIdx.x >= pa.nb0;
  nt_v3_body<APS, BPS, false>(pa.p[second ? 1 : 0], second ? blockIdx.x - pa.nb0 : blockIdx.x, second ? gridDim.x - pa.nb0 : pa.nb0);
}

// A phase-skewed form of this kernel (the two waves of every SIMD one phase apart - R: fragment reads + wait, M: twelve
// MFMAs with the conversions between them - so that one of them is always in its MFMA cluster; eight barriers per stage,
// s_setprio / sched_barrier fences to keep hipcc from moving MFMAs across them) was written, verified bit-identical and
// measured in one process against this one: 578 vs 572 us on the 1063 -> 1024 product, equal everywhere else.  The
// barrier drain is not what holds the kernel at ~45 % MFMA-busy; removed again.

// -------------------------------------------------------------------------------------------------
//   spgnn_gemm_tn : C[M,N] = A[R,M]^T * B[R,N]     (weight gradients: A = g_Y, B = X, R = node count)
//
// The reduction runs over the ROW index of both operands, so global tiles arrive k-major (32 rows x 128
// columns, 512-byte coalesced row segments).  They are stored to LDS as they come ([k][m] fp16 images) and
// the MFMA fragments (8 consecutive k per lane) are produced by the gfx950 transposing LDS read
// ds_read_b64_tr_b16: a 16-lane group reads a 4(k) x 16(m) block and each lane receives one column.
// Row pitch 160 halves (320 B = 64 mod 256) puts the four k-rows of a block in different quarters of the
// bank row: conflict-free.  The row range is split across blockIdx.y (split-K); every split writes its own
// partial tile and the caller sums them (deterministic, no atomics).
// -------------------------------------------------------------------------------------------------
typedef short s16x4 __attribute__((__vector_size__(4 * sizeof(short))));
constexpr int TBK = 32;                          // rows (reduction) per stage

struct ArgsTN {
  const float* A; int64_t lda;                   // (R, M)
  const float* B; int64_t ldb;                   // (R, N)
  float* C; int64_t ldc; int64_t split_stride;   // partials: C + split * split_stride
  int64_t R; int M, N;
  int64_t rows_per_split;
  const float* sA; const float* sB;
  int nbm, nbn;
  float* colsum;                                  // optional per-split column sums of A (bias gradient):
  int64_t cs_stride, cs_split_stride;             //   element (split, m) at colsum[split * cs_split_stride + m * cs_stride]
  int splits;
};

// fragment of the 32 (m) x 16 (k) operand block whose first column is m0 and first k-row is k0 (PITCH_ halves per k-row)
template <int PITCH_>
__device__ __forceinline__ half8 tr_frag(const _Float16* img, int m0, int k0, int lane) {
  const int g = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
  const int col = m0 + (g & 1) * 16 + 4 * pp;
  const int krow = k0 + (g >> 1) * 8 + q;
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + krow * PITCH_ + col));
  const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + (krow + 4) * PITCH_ + col));
  union { s16x4 v[2]; half8 h; } u;
  u.v[0] = lo4; u.v[1] = hi4;
  return u.h;
}

// Tile geometry of the TN kernels: (64 WM) x 128 block tile, 2 WM waves of 64 x 64, 32 reduction rows per stage.
//   WM = 2: 128 x 128, 256 threads, two workgroups per CU (round 1)
//   WM = 4: 256 x 128, 512 threads, one workgroup per CU (round 4): per MFMA 3/4 of the operand bytes come through the
//           vector memory pipe and L2, 3/4 of the values are converted and half the barriers are met - the step the NT
//           kernels took from 256 x 128 to 256 x 256.  Same arithmetic in the same order per output element: bit-identical.
// fp32 tiles are prefetched two stages ahead into two register sets, the conversion + LDS store of stage t+1 is
// interleaved between the MFMAs of stage t (double-buffered LDS, one barrier per stage) - the schedule of
// gemm_nt_f16x3_v2.  An operand tile is 32 rows x COLS columns, k-major as it arrives; thread t's float4 #i sits in row
// (t + NT i) / (COLS / 4), column group (t + NT i) % (COLS / 4).
template <int WM>
struct TnGeo {
  static constexpr int BMt = 64 * WM, NT = 128 * WM;
  static constexpr int PA = BMt + 32, PB = BN + 32;            // halves per LDS k-row (pitch = 64 mod 256 bytes: conflict-free tr reads)
  static constexpr int A_IMG = TBK * PA, B_IMG = TBK * PB;
  static constexpr int STAGE = 2 * A_IMG + 2 * B_IMG;          // Ah | Al | Bh | Bl
  static constexpr int CA4 = BMt / 4, CB4 = BN / 4;            // float4 per tile row
  static constexpr int NLA = TBK * CA4 / NT, NLB = TBK * CB4 / NT;      // float4 per thread and stage: 4 and 4 (WM = 2) / 4 and 2 (WM = 4)
  static constexpr int RSA = NT / CA4, RSB = NT / CB4;         // rows between a thread's consecutive float4
  static constexpr size_t lds_bytes = 2 * (size_t)STAGE * sizeof(_Float16);
};
template <int COLS4, int PITCH_, int NT_, bool WIDE>
__device__ __forceinline__ void store_one_t(_Float16* hi_img, _Float16* lo_img, const float4& v, int i, float s) {
  const int idx = threadIdx.x + NT_ * i;
  const int off = (idx / COLS4) * PITCH_ + (idx % COLS4) * 4;
  uint2 h, l;
  split4_pk<WIDE>(v, s, h, l);
  *reinterpret_cast<uint2*>(hi_img + off) = h;
  *reinterpret_cast<uint2*>(lo_img + off) = l;
}
// pre-split operand (see Args): the staged 16 bytes ARE the hi / lo halves of four consecutive columns of one k-row
template <int COLS4, int PITCH_, int NT_>
__device__ __forceinline__ void store_raw_t(_Float16* hi_img, _Float16* lo_img, const float4& v, int i) {
  const int idx = threadIdx.x + NT_ * i;
  const int off = (idx / COLS4) * PITCH_ + (idx % COLS4) * 4;
  *reinterpret_cast<uint2*>(hi_img + off) = make_uint2(__float_as_uint(v.x), __float_as_uint(v.y));
  *reinterpret_cast<uint2*>(lo_img + off) = make_uint2(__float_as_uint(v.z), __float_as_uint(v.w));
}

template <int WM, bool BPS, bool WIDE>      // BPS: B (= X, the layer input; constant node data for a model's first layer) arrives pre-split
__device__ __forceinline__ void tn_body(const ArgsTN& a, const unsigned bid, const unsigned nblocks) {
  using G = TnGeo<WM>;
  constexpr int BMt = G::BMt, NT = G::NT, PA = G::PA, PB = G::PB, A_IMG = G::A_IMG, B_IMG = G::B_IMG, STAGE = G::STAGE;
  constexpr int NLA = G::NLA, NLB = G::NLB;
  extern __shared__ __attribute__((aligned(16))) _Float16 smem_t[];          // 2 stages x (Ah | Al | Bh | Bl)

  // Placement.  Every tile of one split streams the same row range of both operands, so the (split, tile) work items,
  // split-major, are dealt to the XCDs in eight CONTIGUOUS ranges (workgroups go to XCDs round-robin by linear id): an
  // XCD works on one or two splits at a time, their row ranges are fetched into that XCD's L2 once and the tiles'
  // re-reads (8-9x per operand at 128x128 tiles) are L2 hits instead of fabric traffic.  Any split count works, so the
  // caller picks the one whose workgroup count fills whole rounds of the chip.  The grid is padded to a
  // multiple of 8; the pad workgroups leave at once.
  const unsigned tiles_ = (unsigned)(a.nbm * a.nbn);
  const unsigned w_ = (bid & 7u) * (nblocks >> 3) + (bid >> 3);
  if (w_ >= tiles_ * (unsigned)a.splits) return;
  const unsigned split = w_ / tiles_, tile = w_ % tiles_;
  const int bm = tile / a.nbn, bn = tile % a.nbn;
  const int m0 = bm * BMt, n0 = bn * BN;
  const int64_t r_beg = (int64_t)split * a.rows_per_split;
  const int64_t r_end = r_beg + a.rows_per_split < a.R ? r_beg + a.rows_per_split : a.R;
  const int nk = r_beg < r_end ? (int)((r_end - r_beg + TBK - 1) / TBK) : 0;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 31, fh = lane >> 5;
  const float sA = spgnn_detail::load_scale_monitored(a.sA), sB = spgnn_detail::load_scale_monitored(a.sB);

  f32x16 acc[2][2], acc2[2][2];                                // acc2 (WIDE only): the cross products, carrying 2^11
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) { acc[i][j][e] = 0.f; acc2[i][j][e] = 0.f; }

  float4 ra0[NLA], rb0[NLB], ra1[NLA], rb1[NLB];
  float4 csum = make_float4(0.f, 0.f, 0.f, 0.f);
  const bool do_colsum = a.colsum != nullptr && bn == 0;

  // Loads are unconditional straight-line code (see TileIO in the NT kernel: a branch around a load makes hipcc drain
  // the VM queue at every stage).  A thread's float4 of a tile share one column group and sit RSA / RSB rows apart.  A
  // float4 that starts at or beyond the width is fetched from column 0 instead, rows past the split's range from its last
  // row; SPGNN_TN_MASK zeroes both before the set is summed / converted.  In the steady loop all rows exist, so only tiles
  // that hang over the width (edge_a / edge_b, block-uniform) are masked there.
  const int tcolA = ((int)threadIdx.x % G::CA4) * 4, trowA = (int)threadIdx.x / G::CA4;
  const int tcolB = ((int)threadIdx.x % G::CB4) * 4, trowB = (int)threadIdx.x / G::CB4;
  const int ca = m0 + tcolA, cbn = n0 + tcolB;
  const float* pA = a.A + (ca < a.M ? ca : 0);
  const float* pB = a.B + (cbn < a.N ? cbn : 0);
  const bool edge_a = m0 + BMt > a.M, edge_b = n0 + BN > a.N;
  const int64_t r_last = r_end - 1;
#define SPGNN_TN_LOAD_FULL(T_, RA, RB)                                                                       \
  {                                                                                                          \
    const int64_t rr = r_beg + (int64_t)(T_) * TBK;                                                          \
    _Pragma("unroll") for (int q = 0; q < NLA; ++q) RA[q] = *reinterpret_cast<const float4*>(pA + (rr + trowA + G::RSA * q) * a.lda); \
    _Pragma("unroll") for (int q = 0; q < NLB; ++q) RB[q] = *reinterpret_cast<const float4*>(pB + (rr + trowB + G::RSB * q) * a.ldb); \
  }
#define SPGNN_TN_LOAD_ANY(T_, RA, RB)                                                                        \
  {                                                                                                          \
    const int64_t rr = r_beg + (int64_t)(T_) * TBK;                                                          \
    _Pragma("unroll") for (int q = 0; q < NLA; ++q) {                                                        \
      const int64_t r_ = rr + trowA + G::RSA * q < r_end ? rr + trowA + G::RSA * q : r_last;                 \
      RA[q] = *reinterpret_cast<const float4*>(pA + r_ * a.lda);                                             \
    }                                                                                                        \
    _Pragma("unroll") for (int q = 0; q < NLB; ++q) {                                                        \
      const int64_t r_ = rr + trowB + G::RSB * q < r_end ? rr + trowB + G::RSB * q : r_last;                 \
      RB[q] = *reinterpret_cast<const float4*>(pB + r_ * a.ldb);                                             \
    }                                                                                                        \
  }
#define SPGNN_TN_MASK1(R_, NL_, TROW_, RS_, C_, W_, ROWS_TOO, T_)                                            \
  _Pragma("unroll") for (int q = 0; q < NL_; ++q) {                                                          \
    const bool rv = !(ROWS_TOO) || r_beg + (int64_t)(T_) * TBK + TROW_ + RS_ * q < r_end;                    \
    R_[q].x = rv && (C_) + 0 < (W_) ? R_[q].x : 0.f;                                                         \
    R_[q].y = rv && (C_) + 1 < (W_) ? R_[q].y : 0.f;                                                         \
    R_[q].z = rv && (C_) + 2 < (W_) ? R_[q].z : 0.f;                                                         \
    R_[q].w = rv && (C_) + 3 < (W_) ? R_[q].w : 0.f;                                                         \
  }
  // pre-split rows: a group that straddles the width is zero padded in memory; one at or beyond it came from column 0
#define SPGNN_TN_MASKG(R_, NL_, TROW_, RS_, C_, W_, ROWS_TOO, T_)                                            \
  _Pragma("unroll") for (int q = 0; q < NL_; ++q) {                                                          \
    const bool rv = !(ROWS_TOO) || r_beg + (int64_t)(T_) * TBK + TROW_ + RS_ * q < r_end;                    \
    if (!(rv && (C_) < (W_))) R_[q] = make_float4(0.f, 0.f, 0.f, 0.f);                                       \
  }
#define SPGNN_TN_MASK(T_, RA, RB, ROWS_TOO)                                                                  \
  {                                                                                                          \
    if ((ROWS_TOO) || edge_a) SPGNN_TN_MASK1(RA, NLA, trowA, G::RSA, ca, a.M, ROWS_TOO, T_)                  \
    if ((ROWS_TOO) || edge_b) { if constexpr (BPS) SPGNN_TN_MASKG(RB, NLB, trowB, G::RSB, cbn, a.N, ROWS_TOO, T_) \
                                else SPGNN_TN_MASK1(RB, NLB, trowB, G::RSB, cbn, a.N, ROWS_TOO, T_) }         \
  }
#define SPGNN_TN_PUTA(HI_, LO_, R_, I_) store_one_t<G::CA4, PA, NT, WIDE>(HI_, LO_, R_, I_, sA);
#define SPGNN_TN_PUTB(HI_, LO_, R_, I_)                                                                      \
  { if constexpr (BPS) store_raw_t<G::CB4, PB, NT>(HI_, LO_, R_, I_); else store_one_t<G::CB4, PB, NT, WIDE>(HI_, LO_, R_, I_, sB); }
#define SPGNN_TN_CSUM(RA)                                                            \
  if (do_colsum) {                                                                   \
    _Pragma("unroll") for (int q = 0; q < NLA; ++q) { csum.x += RA[q].x; csum.y += RA[q].y; csum.z += RA[q].z; csum.w += RA[q].w; } \
  }
  if (nk > 0) {                                          // block-uniform; an empty split only writes zeros
    SPGNN_TN_LOAD_ANY(0, ra0, rb0)
    SPGNN_TN_LOAD_ANY(1, ra1, rb1)
    SPGNN_TN_MASK(0, ra0, rb0, true)
    SPGNN_TN_CSUM(ra0)
    _Float16* st = smem_t;
#pragma unroll
    for (int q = 0; q < NLA; ++q) SPGNN_TN_PUTA(st, st + A_IMG, ra0[q], q)
#pragma unroll
    for (int q = 0; q < NLB; ++q) SPGNN_TN_PUTB(st + 2 * A_IMG, st + 2 * A_IMG + B_IMG, rb0[q], q)
    __syncthreads();
    SPGNN_TN_LOAD_ANY(2, ra0, rb0)

  // stage T_: MFMAs on buffer PAR_ (= T_ & 1); register set (RA, RB) = stage T_+1 is converted into the other buffer,
  // one float4 per three MFMAs; then the set is refilled with stage T_+3.  STEADY_ = 1: stages T_+1 and
  // T_+3 exist with all their rows (no tests); STEADY_ = 0: the last stages (set masked first, harmless refill).
#define SPGNN_TN_STAGE(T_, PAR_, RA, RB, STEADY_)                                                            \
  {                                                                                                          \
    const _Float16* cb = smem_t + (PAR_) * STAGE;                                                            \
    _Float16* nbuf = smem_t + (1 - (PAR_)) * STAGE;                                                          \
    const bool has_next = (STEADY_) || (T_) + 1 < nk;                                                        \
    if (has_next) {                                                                                          \
      SPGNN_TN_MASK((T_) + 1, RA, RB, !(STEADY_))                                                            \
      SPGNN_TN_CSUM(RA)                                                                                      \
    }                                                                                                        \
    _Pragma("unroll") for (int ks = 0; ks < TBK / 16; ++ks) {                                                \
      half8 ah[2], al[2], bh[2], bl[2];                                                                      \
      _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                        \
        ah[i] = tr_frag<PA>(cb, wm * 64 + i * 32, ks * 16, lane);                                            \
        al[i] = tr_frag<PA>(cb + A_IMG, wm * 64 + i * 32, ks * 16, lane);                                    \
      }                                                                                                      \
      _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                        \
        bh[j] = tr_frag<PB>(cb + 2 * A_IMG, wn * 64 + j * 32, ks * 16, lane);                                \
        bl[j] = tr_frag<PB>(cb + 2 * A_IMG + B_IMG, wn * 64 + j * 32, ks * 16, lane);                        \
      }                                                                                                      \
      _Pragma("unroll") for (int c = 0; c < 12; ++c) {                   /* product-major, as the NT kernel */ \
        const int pr = c >> 2, ij = c & 3;                                                                   \
        const int i = ij >> 1, j = ij & 1;                                                                   \
        if (WIDE && pr != 2)                                                                                 \
          acc2[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pr == 0 ? al[i] : ah[i], pr == 1 ? bl[j] : bh[j], acc2[i][j], 0, 0, 0); \
        else                                                                                                 \
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pr == 0 ? al[i] : ah[i], pr == 1 ? bl[j] : bh[j], acc[i][j], 0, 0, 0); \
        if (has_next && c % 3 == 2) {                                                                        \
          const int slot = ks * 4 + c / 3;                              /* 8 slots, NLA + NLB float4 to convert */ \
          if (slot < NLA) SPGNN_TN_PUTA(nbuf, nbuf + A_IMG, RA[slot], slot)                                  \
          else if (slot - NLA < NLB) SPGNN_TN_PUTB(nbuf + 2 * A_IMG, nbuf + 2 * A_IMG + B_IMG, RB[slot - NLA < NLB ? slot - NLA : 0], slot - NLA) \
        }                                                                                                    \
      }                                                                                                      \
    }                                                                                                        \
    if (STEADY_) SPGNN_TN_LOAD_FULL((T_) + 3, RA, RB) else SPGNN_TN_LOAD_ANY((T_) + 3, RA, RB)               \
    __syncthreads();                                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
  }
    const int nfull = (int)((r_end - r_beg) / TBK);            // stages with all 32 rows
    int t = 0;
    for (; t + 4 < nfull; t += 2) {
      SPGNN_TN_STAGE(t, 0, ra1, rb1, 1)
      SPGNN_TN_STAGE(t + 1, 1, ra0, rb0, 1)
    }
    for (; t + 1 < nk; t += 2) {
      SPGNN_TN_STAGE(t, 0, ra1, rb1, 0)
      SPGNN_TN_STAGE(t + 1, 1, ra0, rb0, 0)
    }
    if (t < nk) SPGNN_TN_STAGE(t, 0, ra1, rb1, 0)
  }
#undef SPGNN_TN_STAGE
#undef SPGNN_TN_LOAD_FULL
#undef SPGNN_TN_LOAD_ANY
#undef SPGNN_TN_MASK
#undef SPGNN_TN_MASK1
#undef SPGNN_TN_MASKG
#undef SPGNN_TN_PUTA
#undef SPGNN_TN_PUTB
#undef SPGNN_TN_CSUM

  if (do_colsum) {                                 // fold the 8 row groups that share a column chunk
    float* red = reinterpret_cast<float*>(smem_t);
    *reinterpret_cast<float4*>(red + trowA * BMt + tcolA) = csum;
    __syncthreads();
    if ((int)threadIdx.x < BMt && m0 + (int)threadIdx.x < a.M) {
      float t_ = 0.f;
#pragma unroll
      for (int g_ = 0; g_ < G::RSA; ++g_) t_ += red[g_ * BMt + threadIdx.x];
      a.colsum[(int64_t)split * a.cs_split_stride + (int64_t)(m0 + threadIdx.x) * a.cs_stride] = t_;
    }
  }
  if constexpr (WIDE) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = fmaf(acc2[i][j][e], kWideLoInv, acc[i][j][e]);
  }
  const float alpha = 1.f / (sA * sB);
  float* Cp = a.C + (int64_t)split * a.split_stride;
  if (m0 + BMt <= a.M && n0 + BN <= a.N) {             // interior tile (block-uniform): no per-element tests
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        float* dst = Cp + (int64_t)(m0 + wm * 64 + i * 32 + 4 * fh) * a.ldc + n0 + wn * 64 + j * 32 + fr;
#pragma unroll
        for (int e = 0; e < 16; ++e) dst[(int64_t)((e & 3) + 8 * (e >> 2)) * a.ldc] = acc[i][j][e] * alpha;
      }
    return;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wn * 64 + j * 32 + fr;
      if (col >= a.N) continue;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
        if (row < a.M) Cp[(int64_t)row * a.ldc + col] = acc[i][j][e] * alpha;
      }
    }
}
template <int WM, bool BPS, bool WIDE>
__global__ __launch_bounds__(128 * WM, (WM == 2 && !WIDE) ? 2 : 1) void gemm_tn_f16x3(ArgsTN a) { tn_body<WM, BPS, WIDE>(a, blockIdx.x, gridDim.x); }
struct PairArgsTN { ArgsTN p[2]; unsigned nb0; };
template <int WM, bool BPS, bool WIDE>
__global__ __launch_bounds__(128 * WM, (WM == 2 && !WIDE) ? 2 : 1) void gemm_tn_pair(PairArgsTN pa) {          // see gemm_nt_pair_v2
  const bool second = blockIdx.x >= pa.nb0;
  tn_body<WM, BPS, WIDE>(pa.p[second ? 1 : 0], second ? blockIdx.x - pa.nb0 : blockIdx.x, second ? gridDim.x - pa.nb0 : pa.nb0);
}

// out[i] = sum_s part[s * stride + i]: the deterministic reduction of split-K partial tiles (weight gradients, skinny
// score-weight gradients).  One float4 per thread; SL threads share a float4 and take every SL-th split (fixed order,
// folded through LDS), so a reduction over thousands of small partials still fills the chip.
// (the three reductions are device functions of a block index, so that spgnn_sum_partials_multi can run several of them in
// one launch with exactly the arithmetic of the single-job kernels)
template <int SL>
__device__ __forceinline__ void sum_partials_body(const float* __restrict__ part, int64_t stride4, int S, int64_t n4,
                                                  float* __restrict__ out, unsigned bid, float4* red) {
  constexpr int QB = 256 / SL;
  const int q = threadIdx.x % QB, l = threadIdx.x / QB;
  const int64_t i = (int64_t)bid * QB + q;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < n4) {
    const float4* p = reinterpret_cast<const float4*>(part) + i;
    int s = l;
    for (; s + 3 * SL < S; s += 4 * SL) {
      const float4 a = p[(int64_t)s * stride4], b = p[(int64_t)(s + SL) * stride4], c = p[(int64_t)(s + 2 * SL) * stride4],
                   d = p[(int64_t)(s + 3 * SL) * stride4];
      acc.x += a.x; acc.y += a.y; acc.z += a.z; acc.w += a.w;
      acc.x += b.x; acc.y += b.y; acc.z += b.z; acc.w += b.w;
      acc.x += c.x; acc.y += c.y; acc.z += c.z; acc.w += c.w;
      acc.x += d.x; acc.y += d.y; acc.z += d.z; acc.w += d.w;
    }
    for (; s < S; s += SL) { const float4 a = p[(int64_t)s * stride4]; acc.x += a.x; acc.y += a.y; acc.z += a.z; acc.w += a.w; }
  }
  if (SL > 1) {
    red[threadIdx.x] = acc;
    __syncthreads();
    if (l == 0 && i < n4) {
#pragma unroll
      for (int k = 1; k < SL; ++k) { const float4 a = red[k * QB + q]; acc.x += a.x; acc.y += a.y; acc.z += a.z; acc.w += a.w; }
      reinterpret_cast<float4*>(out)[i] = acc;
    }
  } else if (i < n4) {
    reinterpret_cast<float4*>(out)[i] = acc;
  }
}
template <int SL>
__global__ __launch_bounds__(256) void sum_partials_kernel(const float* __restrict__ part, int64_t stride4, int S, int64_t n4,
                                                           float* __restrict__ out) {
  __shared__ float4 red[256];
  sum_partials_body<SL>(part, stride4, S, n4, out, blockIdx.x, red);
}

// Block diagonal of the summed (2H, H*D) score-weight gradient, as (2, H, D): out[(w*H + h)*D + d] = sum_s part[s][w*H + h][h*D + d]
// - the attention vectors' gradients g_attn_l (w = 0) and g_attn_r (w = 1) of a GATConv whose scores come from ft; the
// off-diagonal blocks (other heads' columns) are never needed, so they are neither summed nor copied.
__device__ __forceinline__ void sum_partials_blockdiag_body(const float* __restrict__ part, int64_t stride, int S, int H, int D, int ld,
                                                            float* __restrict__ out, unsigned bid, float* red) {
  // 32 lanes per output element, eight loads in flight per lane: with 16 lanes and a rolled loop (one load at a time, 32 trips
  // at 512 splits) this launch took 19 us, six times per step
  constexpr int SL = 32, QB = 256 / SL;
  const int q = threadIdx.x % QB, l = threadIdx.x / QB;
  const int i = (int)bid * QB + q, n = 2 * H * D;
  float acc = 0.f;
  if (i < n) {
    const int row = i / D, d = i - row * D, h = row % H;
    const float* p = part + (int64_t)row * ld + h * D + d;
    int s = l;
    for (; s + 7 * SL < S; s += 8 * SL) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = p[(int64_t)(s + u * SL) * stride];
      acc += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    }
    for (; s < S; s += SL) acc += p[(int64_t)s * stride];
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  if (l == 0 && i < n) {
#pragma unroll
    for (int k = 1; k < SL; ++k) acc += red[k * QB + q];
    out[i] = acc;
  }
}
__global__ __launch_bounds__(256) void sum_partials_blockdiag_kernel(const float* __restrict__ part, int64_t stride, int S, int H,
                                                                     int D, int ld, float* __restrict__ out) {
  __shared__ float red[256];
  sum_partials_blockdiag_body(part, stride, S, H, D, ld, out, blockIdx.x, red);
}

// The same reduction for the weight-gradient tiles, written COMPACT: out (M, N) contiguous, plus one extra column of the
// partial rows (the bias column sums that ride in a spare column) as its own vector.  Contiguous gradients are taken
// over by autograd's accumulation as they are; row-strided views (N + 4 floats per row) were cloned once per parameter.
__device__ __forceinline__ void sum_partials_compact_body(const float* __restrict__ part, int64_t stride4, int S, int M, int N, int ldi4,
                                                          float* __restrict__ out, int64_t ldo, float* __restrict__ out2, int64_t ldo2,
                                                          int split_col, float* __restrict__ extra, int extra_col, unsigned bid,
                                                          float4* red) {
  // one float4 of a partial row per four threads (each takes every fourth split, folded through LDS in a fixed order),
  // as sum_partials_kernel<4>; the destination of each of its four columns is looked up afterwards
  constexpr int SL = 4, QB = 256 / SL;
  const int q = threadIdx.x % QB, l = threadIdx.x / QB;
  const int64_t i = (int64_t)bid * QB + q, n4 = (int64_t)M * ldi4;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < n4) {
    const float4* p = reinterpret_cast<const float4*>(part) + i;
    int s = l;
    for (; s + 3 * SL < S; s += 4 * SL) {
      const float4 a = p[(int64_t)s * stride4], b = p[(int64_t)(s + SL) * stride4], c = p[(int64_t)(s + 2 * SL) * stride4],
                   d = p[(int64_t)(s + 3 * SL) * stride4];
      acc.x += a.x; acc.y += a.y; acc.z += a.z; acc.w += a.w;
      acc.x += b.x; acc.y += b.y; acc.z += b.z; acc.w += b.w;
      acc.x += c.x; acc.y += c.y; acc.z += c.z; acc.w += c.w;
      acc.x += d.x; acc.y += d.y; acc.z += d.z; acc.w += d.w;
    }
    for (; s < S; s += SL) { const float4 a = p[(int64_t)s * stride4]; acc.x += a.x; acc.y += a.y; acc.z += a.z; acc.w += a.w; }
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  if (l != 0 || i >= n4) return;
#pragma unroll
  for (int k = 1; k < SL; ++k) { const float4 a = red[k * QB + q]; acc.x += a.x; acc.y += a.y; acc.z += a.z; acc.w += a.w; }
  const int64_t r = i / ldi4; const int c = (int)(i - r * ldi4) * 4;
  const float v[4] = {acc.x, acc.y, acc.z, acc.w};
  float* dst = (out2 && c >= split_col) ? out2 + r * ldo2 + (c - split_col) : out + r * ldo + c;
  const bool whole = c + 3 < N && (!out2 || c + 3 < split_col || c >= split_col) && ((reinterpret_cast<uintptr_t>(dst) & 15) == 0);
  if (whole) { *reinterpret_cast<float4*>(dst) = acc; return; }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int cc = c + j;
    if (cc < N) {
      if (out2 && cc >= split_col) out2[r * ldo2 + (cc - split_col)] = v[j];
      else out[r * ldo + cc] = v[j];
    } else if (extra && cc == extra_col) extra[r] = v[j];
  }
}
__global__ __launch_bounds__(256) void sum_partials_compact_kernel(const float* __restrict__ part, int64_t stride4, int S, int M,
                                                                   int N, int ldi4, float* __restrict__ out, int64_t ldo,
                                                                   float* __restrict__ out2, int64_t ldo2, int split_col,
                                                                   float* __restrict__ extra, int extra_col) {
  __shared__ float4 red[256];
  sum_partials_compact_body(part, stride4, S, M, N, ldi4, out, ldo, out2, ldo2, split_col, extra, extra_col, blockIdx.x, red);
}

// several reductions of the three kinds above in ONE launch (spgnn_sum_partials_multi): the job is found from the block index
constexpr int kMaxSumJobs = 24;          // 24 x 112 bytes of jobs + offsets: 2.8 KB of kernel arguments
struct SumJobs { spgnn_sum_job j[kMaxSumJobs]; unsigned first[kMaxSumJobs + 1]; int n; };
__global__ __launch_bounds__(256) void sum_jobs_kernel(SumJobs a) {
  __shared__ float4 red[256];
  int k = 0;
  while (k + 1 < a.n && blockIdx.x >= a.first[k + 1]) ++k;
  const spgnn_sum_job& j = a.j[k];
  const unsigned bid = blockIdx.x - a.first[k];
  if (j.kind == 2) {
    sum_partials_compact_body(j.partials, j.split_stride / 4, j.splits, j.M, j.N, (int)(j.ld_in / 4), j.out, j.out_stride, j.out2,
                              j.out2_stride, j.split_col, j.extra, j.extra_col, bid, red);
  } else if (j.kind == 1) {
    sum_partials_blockdiag_body(j.partials, j.split_stride, j.splits, j.H, j.D, j.ld, j.out, bid, reinterpret_cast<float*>(red));
  } else {
    const int64_t n4 = j.n / 4;
    if (j.splits <= 8) sum_partials_body<1>(j.partials, j.split_stride / 4, j.splits, n4, j.out, bid, red);
    else if (j.splits <= 128) sum_partials_body<4>(j.partials, j.split_stride / 4, j.splits, n4, j.out, bid, red);
    else sum_partials_body<16>(j.partials, j.split_stride / 4, j.splits, n4, j.out, bid, red);
  }
}

// Weight preparation for a projection layer in ONE pass: the rows of A (ra x K) then B (rb x K) -> dst (ra + rb rows,
// row stride ldd >= K, pad columns zeroed: 16-byte rows for the GEMMs), optionally the transpose dst_t (K rows, row
// stride ldt >= ra + rb, pad columns zeroed: the B operand of the input-gradient product), and one |max| partial per
// block for the tensor's GEMM scale.  32 x 32 tiles through LDS so both images are written in whole row segments.
// (Replaces torch.cat + F.pad + slice + absmax per layer and step, and the transposing copy in the backward pass.)
__global__ __launch_bounds__(256) void weight_cat_kernel(const float* __restrict__ A, int64_t lda, int ra,
                                                         const float* __restrict__ B, int64_t ldb, int rb, int K,
                                                         float* __restrict__ dst, int64_t ldd, float* __restrict__ dst_t,
                                                         int64_t ldt, float* __restrict__ partial) {
  __shared__ float tile[32][33];
  __shared__ float red[4];
  const int R = ra + rb;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;           // 32 x 8
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  float m = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + ty + 8 * i, c = c0 + tx;
    float v = 0.f;
    if (r < R && c < K) v = r < ra ? A[(int64_t)r * lda + c] : B[(int64_t)(r - ra) * ldb + c];
    tile[ty + 8 * i][tx] = v;
    if (r < R && c < ldd) dst[(int64_t)r * ldd + c] = v;
    m = fmaxf(m, fabsf(v));
  }
  __syncthreads();
  if (dst_t) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = c0 + ty + 8 * i, r = r0 + tx;                   // dst_t[c][r] = tile[r - r0][c - c0]
      if (c < K && r < ldt) dst_t[(int64_t)c * ldt + r] = tile[tx][ty + 8 * i];
    }
  }
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.y * gridDim.x + blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// absmax -> power-of-two scale that puts the largest magnitude at 2^14 (fp16 max is 2^16): scale[0] = 2^(14 - ceil(log2 max))
// one partial maximum per block (no atomics: deterministic, no contention); rows must be 16-byte aligned
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, int64_t ld, int64_t rows, int cols,
                                                     float* __restrict__ partial) {
  __shared__ float red[4];
  float m = 0.f;
  const int c4n = cols >> 2;                                   // float4 chunks per row
  const int64_t total4 = rows * c4n;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / c4n; const int c = (int)(i - r * c4n) * 4;
    const float4 v = *reinterpret_cast<const float4*>(x + r * ld + c);
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
  }
  const int tail = cols & 3;
  if (tail) {
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (int64_t)gridDim.x * blockDim.x)
      for (int t = 0; t < tail; ++t) m = fmaxf(m, fabsf(x[r * ld + (cols - tail) + t]));
  }
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}
// scale[0] = 2^(14 - e) with max <= 2^e ; optional multiplicative bound factor (e.g. 1/(1-p) for dropout)
using spgnn_detail::pow2_scale_of;
// small inputs: one block
__global__ void scale_from_partials(const float* __restrict__ partial, int n, float factor, float* __restrict__ scale) {
  float m = 0.f;
  for (int i = threadIdx.x; i < n; i += 64) m = fmaxf(m, partial[i]);
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  if (threadIdx.x == 0) scale[0] = pow2_scale_of(m * factor);
}
// large inputs (per-node maxima, ~150k floats): many blocks fold into ws[0] with one atomicMax each; the block
// that draws the last ticket (ws[1]) publishes the scale and re-zeroes both words for the next call on the
// stream (ws is zero-initialised once by the caller).  Non-negative floats order like their bit patterns.
__global__ __launch_bounds__(256) void scale_from_partials_mb(const float* __restrict__ partial, int n, float factor,
                                                              float* __restrict__ scale, unsigned* __restrict__ ws) {
  __shared__ float red[4];
  float m = 0.f;
  const int n4 = n >> 2;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n4; i += gridDim.x * 256) {
    const float4 v = reinterpret_cast<const float4*>(partial)[i];
    m = fmaxf(fmaxf(m, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) m = fmaxf(m, partial[n4 * 4 + threadIdx.x]);
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    atomicMax(&ws[0], __float_as_uint(m));
    __threadfence();
    if (atomicAdd(&ws[1], 1u) == gridDim.x - 1) {            // last block: every atomicMax above is visible
      const float all = __uint_as_float(atomicMax(&ws[0], 0u));
      scale[0] = pow2_scale_of(all * factor);
      ws[0] = 0u; ws[1] = 0u;
    }
  }
}

// spgnn_presplit: up to two row-major fp32 matrices (a weight operand and its transpose) -> their pre-split form (see
// Args), with the operand's power-of-two scale either given (scale_in) or derived here from the block maxima a producer
// left (partial[0..n), as scale_from_partials would): every block reduces the short array itself, block 0 publishes the
// scale.  One launch replaces the scale reduction of spgnn_weight_cat's operand and adds the split.
__global__ __launch_bounds__(256) void presplit_kernel(const float* __restrict__ partial, int n, const float* __restrict__ scale_in,
                                                       float* __restrict__ scale_out,
                                                       const float* __restrict__ s0, int64_t ld0, int R0, int K0, float* __restrict__ d0,
                                                       const float* __restrict__ s1, int64_t ld1, int R1, int K1, float* __restrict__ d1,
                                                       int wide) {
  __shared__ float red[4];
  float sc;
  if (scale_in) {
    sc = spgnn_detail::load_scale(scale_in);
  } else {
    float m = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) m = fmaxf(m, partial[i]);
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    sc = pow2_scale_of(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])));
  }
  if (scale_out && blockIdx.x == 0 && threadIdx.x == 0) scale_out[0] = sc;
  const int g0 = (K0 + 3) >> 2, g1 = s1 ? (K1 + 3) >> 2 : 0;           // float4 groups per row
  const int64_t n0 = (int64_t)R0 * g0, tot = n0 + (int64_t)R1 * g1;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < tot; i += (int64_t)gridDim.x * 256) {
    const bool first = i < n0;
    const int64_t j = first ? i : i - n0;
    const int g = first ? g0 : g1, K = first ? K0 : K1;
    const int64_t ld = first ? ld0 : ld1;
    const int r = (int)(j / g), c = (int)(j % g) * 4;
    const float* src = (first ? s0 : s1) + (int64_t)r * ld + c;
    float4 v = *reinterpret_cast<const float4*>(src);
    v.y = c + 1 < K ? v.y : 0.f; v.z = c + 2 < K ? v.z : 0.f; v.w = c + 3 < K ? v.w : 0.f;
    uint2 h, l;
    if (wide) split4_pk<true>(v, sc, h, l); else split4_pk<false>(v, sc, h, l);
    *reinterpret_cast<uint4*>((first ? d0 : d1) + (int64_t)r * ld + c) = make_uint4(h.x, h.y, l.x, l.y);
  }
}

// -------------------------------------------------------------------------------------------------
// spgnn_weight_prep: the weight operands of EVERY projection layer of a model in two launches per step instead of two per
// layer (spgnn_weight_cat + spgnn_presplit each): a table of layers, one 32 x 32 tile per workgroup, the table entry found
// from the block index.  Pass 0 leaves max |.| of every tile in one word per workgroup (plain stores: 2 400 atomicMax on a
// dozen addresses serialised to 24 us, three times the copy itself); pass 1 folds its layer's words (a maximum: order-
// independent, so the scale is deterministic), derives the power-of-two scale and writes the four images the GEMMs take:
// [A; B] with 16-byte rows, its transpose, and both in pre-split form.
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void weight_prep_kernel(const spgnn_weight_prep_layer* __restrict__ tab, int n_layers,
                                                          float* __restrict__ blockmax, int pass) {
  __shared__ float tile[32][33];
  __shared__ float red[4];
  int l = 0;
  while (l + 1 < n_layers && (int64_t)blockIdx.x >= tab[l + 1].first_block) ++l;
  const spgnn_weight_prep_layer L = tab[l];
  const bool cols = (L.mode & 1) != 0;                   // [a | b]: rows_b = the columns a contributes
  const bool wide = (L.mode & 2) != 0;                   // pre-split images in the wide-range form (split4_pk<true>)
  const int R = cols ? L.rows_a : L.rows_a + L.rows_b, K = L.K;
  const int tiles_x = (int)((L.dst_stride + 31) / 32);
  const int b = (int)(blockIdx.x - L.first_block);
  const int r0 = (b / tiles_x) * 32, c0 = (b % tiles_x) * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;           // 32 x 8
  float m = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + ty + 8 * i, c = c0 + tx;
    float v = 0.f;
    if (r < R && c < K) {
      if (cols) v = c < L.rows_b ? L.a[(int64_t)r * L.a_stride + c] : L.b[(int64_t)r * L.b_stride + c - L.rows_b];
      else v = r < L.rows_a ? L.a[(int64_t)r * L.a_stride + c] : L.b[(int64_t)(r - L.rows_a) * L.b_stride + c];
    }
    tile[ty + 8 * i][tx] = v;
    m = fmaxf(m, fabsf(v));
  }
  if (pass == 0) {
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) blockmax[blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    return;
  }
  const int64_t lb_end = l + 1 < n_layers ? tab[l + 1].first_block : (int64_t)gridDim.x;
  float lm = 0.f;
  for (int64_t q = L.first_block + threadIdx.x; q < lb_end; q += 256) lm = fmaxf(lm, blockmax[q]);
  for (int off = 32; off > 0; off >>= 1) lm = fmaxf(lm, __shfl_xor(lm, off, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = lm;
  __syncthreads();
  const float sc = pow2_scale_of(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])));
  if (b == 0 && threadIdx.x == 0) L.scale[0] = sc;
  // one group of four per thread: rows of the tile for dst / ps, columns of the tile (rows of the transpose) for dst_t / ps_t
  const int gr = threadIdx.x >> 3, gc = (threadIdx.x & 7) * 4;
  {
    const int r = r0 + gr, c = c0 + gc;
    if (r < R && c < L.dst_stride) {
      const float4 v = make_float4(tile[gr][gc], tile[gr][gc + 1], tile[gr][gc + 2], tile[gr][gc + 3]);     // beyond K: zeros
      *reinterpret_cast<float4*>(L.dst + (int64_t)r * L.dst_stride + c) = v;
      uint2 h, lo;
      if (wide) split4_pk<true>(v, sc, h, lo); else split4_pk<false>(v, sc, h, lo);
      *reinterpret_cast<uint4*>(L.ps + (int64_t)r * L.dst_stride + c) = make_uint4(h.x, h.y, lo.x, lo.y);
    }
  }
  if (L.dst_t) {
    const int c = c0 + gr, r = r0 + gc;                           // row c of the transpose, its columns r .. r + 3
    if (c < K && r < L.dst_t_stride) {
      const float4 v = make_float4(tile[gc][gr], tile[gc + 1][gr], tile[gc + 2][gr], tile[gc + 3][gr]);     // beyond R: zeros
      *reinterpret_cast<float4*>(L.dst_t + (int64_t)c * L.dst_t_stride + r) = v;
      uint2 h, lo;
      if (wide) split4_pk<true>(v, sc, h, lo); else split4_pk<false>(v, sc, h, lo);
      *reinterpret_cast<uint4*>(L.ps_t + (int64_t)c * L.dst_t_stride + r) = make_uint4(h.x, h.y, lo.x, lo.y);
    }
  }
}

// -------------------------------------------------------------------------------------------------
// The weight-space half of "output GATConv without activation, heads averaged, classifier folded through"
// (ops.gat_layer_linear_mean; reference models.py:320-327 with 921-933) - ~28 tiny torch / rocBLAS launches per step before:
//   W_comb[d, h F + f] = W_fc[h D + d, f] / H,   W_comb[d, H F + f] = sum_h W_res[h D + d, f] / H,   b_mean = sum_h bias[h] / H
//   P = Wc W_comb  (J, (H+1) F),   c0 = Wc b_mean + bc
// and, backward, from M1 = g_logits^T Zx (J, Kc) and cs = colsum(g_logits):
//   g_W_comb = Wc^T M1 -> g_W_fc, g_W_res (/ H);   g_bias = Wc^T cs / H;   g_Wc = M1 W_comb^T + cs b_mean^T
// Forward: one workgroup of 1024 threads per 32 columns of W_comb (all D rows: no cross-workgroup reduction, fixed summation
// order) + one for b_mean / c0.  With `wbf` the combined weight is rounded to bf16 first (bf16-storage path): the image the
// bf16 product reads is written too, and P / the fp32 image hold the rounded values - the function as evaluated.
// -------------------------------------------------------------------------------------------------
__device__ __forceinline__ float bf16_rne(float x, unsigned short& bits) {
  unsigned u = __float_as_uint(x);
  u += 0x7FFFu + ((u >> 16) & 1u);
  bits = (unsigned short)(u >> 16);
  return __uint_as_float(u & 0xFFFF0000u);
}

struct FoldFwd {
  const float* w_fc; int64_t ld_fc; const float* w_res; int64_t ld_res; const float* bias;
  const float* wc; int64_t ld_wc; const float* bc;
  float* w_comb; int64_t ld_w; unsigned short* wbf; int64_t ld_wbf; float* b_mean; float* P; int64_t ld_p; float* c0; float* absmax;
  float* part; unsigned* tickets;                            // (column chunk, row quarter, JP, 32) partial sums of P; one ticket per column chunk
  int H, D, F, J, Kc, Kp;
};
constexpr int kFoldQ = 4;                                   // row quarters of D per column chunk: 4 x more workgroups share the work
constexpr int kFoldJ = 32;                                  // classifier outputs at most

// sum over the 32 row groups of `v[j]` (32 values per thread) -> out(j, column) for the threads of group j % 8, eight j at a time
template <int JP, class STORE>
__device__ __forceinline__ void fold_reduce32(float (&v)[JP], float (&red)[32][8][33], int tg, int tk, STORE store) {
#pragma unroll
  for (int r = 0; r < JP / 8; ++r) {
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) red[tg][jj][tk] = v[r * 8 + jj];
    __syncthreads();
    if (tg < 8) {
      float s_ = 0.f;
#pragma unroll
      for (int g = 0; g < 32; ++g) s_ += red[g][tg][tk];
      store(r * 8 + tg, s_);
    }
    __syncthreads();
  }
}

// JP: J rounded up to 8 (compile time): the classifier-row loops are unrolled with UNCONDITIONAL loads (rows beyond J re-read
// row J - 1 and feed sums that are never stored) - with a test around them hipcc drained the memory queue after every load and
// the 22 x 32 loads of a thread ran one after the other: 193 us instead of 12.
template <int JP>
__global__ __launch_bounds__(1024) void linear_mean_fold_fwd_kernel(FoldFwd a) {
  __shared__ float red[32][8][33];
  __shared__ float mxs[16];
  const int tk = threadIdx.x & 31, tg = threadIdx.x >> 5;
  const int nkc = (a.Kp + 31) / 32;
  const float invH = 1.f / (float)a.H;
  __shared__ bool last;
  float v[JP];
#pragma unroll
  for (int j = 0; j < JP; ++j) v[j] = 0.f;
  const float* wcj[JP];                                    // row pointers of the classifier weight, clamped to the last row
#pragma unroll
  for (int j = 0; j < JP; ++j) wcj[j] = a.wc + (int64_t)(j < a.J ? j : a.J - 1) * a.ld_wc;
  if ((int)blockIdx.x < nkc * kFoldQ) {
    const int kc = blockIdx.x / kFoldQ, dq = blockIdx.x - kc * kFoldQ;
    const int dchunk = ((a.D + kFoldQ - 1) / kFoldQ + 31) / 32 * 32;
    const int d_beg = dq * dchunk, d_end = d_beg + dchunk < a.D ? d_beg + dchunk : a.D;
    const int k = kc * 32 + tk;
    const bool kin = k < a.Kc, kst = k < a.Kp;
    const int HF = a.H * a.F;
    const bool fc = k < HF;
    const int h = fc ? k / a.F : 0, f = fc ? k - h * a.F : k - HF;
    const float* rsrc = a.w_res ? a.w_res : a.w_fc;        // (no residual: read W_fc instead, the value is dropped)
    const int64_t rld = a.w_res ? a.ld_res : a.ld_fc;
    float mx = 0.f;
    for (int d = d_beg + tg; d < d_end; d += 32) {
      // (loads without a test around them: a column that does not exist reads element (d, 0) of the same matrix)
      const float* src = fc ? a.w_fc : rsrc;
      const int64_t ld_ = fc ? a.ld_fc : rld;
      const int fq = kin ? f : 0;
      float w = src[(int64_t)((fc ? h : 0) * a.D + d) * ld_ + fq];
      if (!fc) {
        for (int hh = 1; hh < a.H; ++hh) w += src[(int64_t)(hh * a.D + d) * ld_ + fq];
      }
      w = (kin && (fc || a.w_res)) ? w * invH : 0.f;
      unsigned short bits = 0;
      if (a.wbf) w = bf16_rne(w, bits);
      if (kst) {
        a.w_comb[(int64_t)d * a.ld_w + k] = w;
        if (a.wbf) a.wbf[(int64_t)d * a.ld_wbf + k] = bits;
      }
      mx = fmaxf(mx, fabsf(w));
      float c[JP];
#pragma unroll
      for (int j = 0; j < JP; ++j) c[j] = wcj[j][d];
#pragma unroll
      for (int j = 0; j < JP; ++j) v[j] = fmaf(c[j], w, v[j]);
    }
    // this quarter's sums -> device-visible partials; the workgroup that arrives last for the column chunk adds the four
    // quarters in order (see masked_ce_kernel for the store / ticket protocol)
    float* mine = a.part + ((int64_t)blockIdx.x * JP) * 32;
    fold_reduce32<JP>(v, red, tg, tk, [&](int j, float s_) {
      __hip_atomic_store(mine + j * 32 + tk, s_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    });
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) last = __hip_atomic_fetch_add(a.tickets + kc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == kFoldQ - 1;
    __syncthreads();
    if (last) {
      if (threadIdx.x < JP * 32) {
        const int j = threadIdx.x >> 5;
        float s_ = 0.f;
#pragma unroll
        for (int q = 0; q < kFoldQ; ++q)
          s_ += __hip_atomic_load(a.part + (((int64_t)(kc * kFoldQ + q) * JP + j) * 32 + tk), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (j < a.J && kst) a.P[(int64_t)j * a.ld_p + k] = s_;
      }
      if (threadIdx.x == 0) a.tickets[kc] = 0u;
    }
    if (a.absmax) {
      for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
      if ((threadIdx.x & 63) == 0) mxs[threadIdx.x >> 6] = mx;
      __syncthreads();
      if (threadIdx.x == 0) {
        float m = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) m = fmaxf(m, mxs[q]);
        spgnn_detail::slots_max(a.absmax, m, blockIdx.x);
      }
    }
    return;
  }
  // the last workgroup: b_mean and c0 = Wc b_mean + bc
  for (int d = threadIdx.x; d < a.D; d += 1024) {
    float b = 0.f;
    if (a.bias) {
      for (int hh = 0; hh < a.H; ++hh) b += a.bias[hh * a.D + d];
      b *= invH;
    }
    if (a.b_mean) a.b_mean[d] = b;
    float c[JP];
#pragma unroll
    for (int j = 0; j < JP; ++j) c[j] = wcj[j][d];
#pragma unroll
    for (int j = 0; j < JP; ++j) v[j] = fmaf(c[j], b, v[j]);
  }
  fold_reduce32<JP>(v, red, tg, tk, [&](int j, float s_) {
    for (int off = 16; off > 0; off >>= 1) s_ += __shfl_xor(s_, off, 32);     // the 32 columns of group j (lanes of one half-wave)
    if (tk == 0 && j < a.J) a.c0[j] = s_ + (a.bc ? a.bc[j] : 0.f);
  });
}

struct FoldBwd {
  const float* M1; int64_t ld_m; const float* cs; const float* wc; int64_t ld_wc; const float* w_comb; int64_t ld_w; const float* b_mean;
  float* g_fc; int64_t ld_gfc; float* g_res; int64_t ld_gres; float* g_bias; float* g_wc; int64_t ld_gwc;
  int H, D, F, J, Kc; unsigned blocksA;
};

template <int JP>
__global__ __launch_bounds__(256) void linear_mean_fold_bwd_kernel(FoldBwd a) {
  const float invH = 1.f / (float)a.H;
  if (blockIdx.x < a.blocksA) {                              // g_W_comb = Wc^T M1, scattered into the two parameters' gradients
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (int64_t)a.D * a.Kc) return;
    const int d = (int)(idx / a.Kc), k = (int)(idx - (int64_t)d * a.Kc);
    float cw[JP], cm[JP];                                    // unconditional loads (see the forward kernel); rows beyond J weigh 0
#pragma unroll
    for (int j = 0; j < JP; ++j) {
      const int jc = j < a.J ? j : a.J - 1;
      cw[j] = a.wc[(int64_t)jc * a.ld_wc + d];
      cm[j] = a.M1[(int64_t)jc * a.ld_m + k];
    }
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < JP; ++j) acc = fmaf(j < a.J ? cw[j] : 0.f, cm[j], acc);
    acc *= invH;
    const int HF = a.H * a.F;
    if (k < HF) {
      const int h = k / a.F, f = k - h * a.F;
      a.g_fc[(int64_t)(h * a.D + d) * a.ld_gfc + f] = acc;
    } else if (a.g_res) {
      const int f = k - HF;
      for (int hh = 0; hh < a.H; ++hh) a.g_res[(int64_t)(hh * a.D + d) * a.ld_gres + f] = acc;
    }
    return;
  }
  // g_Wc[j, d] = sum_k M1[j, k] W_comb[d, k] + cs[j] b_mean[d] for 32 rows d of W_comb; g_bias = Wc^T cs / H
  __shared__ __attribute__((aligned(16))) float wt[32][68], m1s[32][68];     // pitch 68: rows stay 16-byte aligned for ds_read_b128
  const int d0 = (int)(blockIdx.x - a.blocksA) * 32;
  const int dl = threadIdx.x & 31, jg = threadIdx.x >> 5;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < a.Kc; k0 += 64) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int idx = threadIdx.x + 256 * i, r = idx >> 6, c = idx & 63;
      const bool cin = k0 + c < a.Kc;
      const int cc = cin ? k0 + c : 0;
      const float wv = a.w_comb[(int64_t)(d0 + r < a.D ? d0 + r : a.D - 1) * a.ld_w + cc];     // unconditional, clamped
      const float mv = a.M1[(int64_t)(r < a.J ? r : a.J - 1) * a.ld_m + cc];
      wt[r][c] = (d0 + r < a.D && cin) ? wv : 0.f;
      m1s[r][c] = (r < a.J && cin) ? mv : 0.f;
    }
    __syncthreads();
#pragma unroll 4
    for (int c = 0; c < 64; c += 4) {
      const float4 w = *reinterpret_cast<const float4*>(&wt[dl][c]);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 m = *reinterpret_cast<const float4*>(&m1s[jg + 8 * q][c]);
        acc[q] = fmaf(w.x, m.x, acc[q]); acc[q] = fmaf(w.y, m.y, acc[q]); acc[q] = fmaf(w.z, m.z, acc[q]); acc[q] = fmaf(w.w, m.w, acc[q]);
      }
    }
    __syncthreads();
  }
  const int d = d0 + dl;
  if (d >= a.D) return;
  const float bm = a.b_mean ? a.b_mean[d] : 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int j = jg + 8 * q;
    if (j < a.J && a.g_wc) a.g_wc[(int64_t)j * a.ld_gwc + d] = fmaf(a.cs[j], bm, acc[q]);
  }
  if (jg == 0 && a.g_bias) {
    float s_ = 0.f;
#pragma unroll
    for (int j = 0; j < JP; ++j) {
      const int jc = j < a.J ? j : a.J - 1;
      s_ = fmaf(j < a.J ? a.wc[(int64_t)jc * a.ld_wc + d] : 0.f, a.cs[jc], s_);
    }
    s_ *= invH;
    for (int hh = 0; hh < a.H; ++hh) a.g_bias[hh * a.D + d] = s_;
  }
}

}  // namespace gemm

extern "C" {

// `tile`: 0 = chosen from the shape (below); 2 = 128 x 128, 4 = 256 x 128, 5 = 256 x 256 block tiles.  Every tile shape
// performs the same arithmetic in the same order per output element: results are bit-identical (tests/test_hip_gemm.py).
// One NT product as the kernels take it, after validation.  variant: 5 = 256 x 256 (gemm_nt_f16x3_v3), 4 = 256 x 128, 2 = 128 x 128.
struct NtPlan { gemm::Args a; int variant; int64_t blocks; };

static int gemm_nt_plan(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int64_t M,
                        int64_t N, int64_t K, const float* scale_a, const float* scale_b, const float* upd_u,
                        int64_t upd_u_stride, const float* upd_v, int64_t upd_v_stride, int32_t upd_j, const float* bias,
                        int32_t activation, const float* score_l, const float* score_r, float* score_out, int32_t score_cols,
                        const float* mean_other, int64_t mean_other_stride, float* mean_out, int64_t mean_out_stride,
                        float* absmax_out, int32_t tile, NtPlan* plan) {
  plan->blocks = 0;
  if (tile != 0 && tile != 2 && tile != 4 && tile != 5) return spgnn_detail::fail_at(SPGNN_ERR_ENUM, __func__, __LINE__);
  if (mean_out) {
    if (!mean_other) return spgnn_detail::fail_at(SPGNN_ERR_NULLPTR, __func__, __LINE__);
    if (mean_other_stride < N || mean_out_stride < N || (mean_other_stride & 3) || (mean_out_stride & 3) ||
        (reinterpret_cast<uintptr_t>(mean_other) & 15) || (reinterpret_cast<uintptr_t>(mean_out) & 15))
      return spgnn_detail::fail_at(SPGNN_ERR_STRIDE, __func__, __LINE__);
  } else if (mean_other) {                                  // addend (spgnn_gemm_nt_add)
    if (mean_other_stride < N || (mean_other_stride & 3) || (reinterpret_cast<uintptr_t>(mean_other) & 15))
      return spgnn_detail::fail_at(SPGNN_ERR_STRIDE, __func__, __LINE__);
  }
  if (M < 0 || N < 0 || K <= 0 || M > INT32_MAX || N > INT32_MAX || K > INT32_MAX) return spgnn_detail::fail_at(SPGNN_ERR_SHAPE, __func__, __LINE__);
  if (score_out) {
    if (!score_l || !score_r) return spgnn_detail::fail_at(SPGNN_ERR_NULLPTR, __func__, __LINE__);
    if (score_cols <= 0 || (score_cols & 63) || score_cols > N ||
        (reinterpret_cast<uintptr_t>(score_l) & 15) || (reinterpret_cast<uintptr_t>(score_r) & 15) ||
        (reinterpret_cast<uintptr_t>(score_out) & 7))
      return spgnn_detail::fail_at(SPGNN_ERR_SHAPE, __func__, __LINE__);
  }
  if (activation < SPGNN_ACT_NONE || activation > SPGNN_ACT_LRELU) return spgnn_detail::fail_at(SPGNN_ERR_ENUM, __func__, __LINE__);
  if (upd_j < 0 || upd_j > 32) return spgnn_detail::fail_at(SPGNN_ERR_SHAPE, __func__, __LINE__);
  if (upd_j > 0) {
    if (!upd_u || !upd_v) return spgnn_detail::fail_at(SPGNN_ERR_NULLPTR, __func__, __LINE__);
    if (upd_u_stride < upd_j || upd_v_stride < ((N + 3) & ~int64_t(3)) || (upd_v_stride & 3) ||
        (reinterpret_cast<uintptr_t>(upd_v) & 15))
      return spgnn_detail::fail_at(SPGNN_ERR_STRIDE, __func__, __LINE__);        // V rows: 16-byte aligned, zero padded to a multiple of 4 columns
  }
  if (M == 0 || N == 0) return SPGNN_OK;                    // plan->blocks stays 0: nothing to launch
  if (!A || !B || !C) return spgnn_detail::fail_at(SPGNN_ERR_NULLPTR, __func__, __LINE__);
  if (lda < K || ldb < K || ldc < N || (lda & 3) || (ldb & 3) || (reinterpret_cast<uintptr_t>(A) & 15) ||
      (reinterpret_cast<uintptr_t>(B) & 15))
    return spgnn_detail::fail_at(SPGNN_ERR_STRIDE, __func__, __LINE__);
  // 256-row tiles (8 waves, 1 block/CU) pay off only for deep, wide products; otherwise 128-row tiles, 2 blocks/CU.
  // 256 x 256 tiles (gemm_nt_f16x3_v3) run ~13 % faster per flop than 256 x 128 ones, but quantise coarser: compare whole
  // rounds of tiles over the 256 CUs, a 256 x 128 tile costing half a 256 x 256 one (tools/gemm_tiles.py, one process,
  // M = 76 410: N = 1024 K = 1063 4.67 -> 5 rounds vs 9.34 -> 10 halves: 481 vs 571 us; N = 768 K = 512 3.50 -> 4 vs
  // 7.01 -> 8: 193 vs 214; N = 512 K = 768 2.34 -> 3 vs 4.67 -> 5: 199 vs 194; M = 9 641: N = 1024 one partial round
  // vs 1.19 -> 2: 85 vs 98 us; N = 768 one round of either: 39 vs 29).  Offsets inside that kernel are 32-bit.
  const bool fits31 = M * lda * 4 < (int64_t(1) << 31) && N * ldb * 4 < (int64_t(1) << 31);
  const int64_t rounds3 = (((M + 255) / 256) * ((N + 255) / 256) + 255) / 256;
  const int64_t rounds4 = (((M + 255) / 256) * ((N + 127) / 128) + 255) / 256;
  const bool v3_wins = M >= 4096 && K >= 256 && N >= 512 && (double)rounds3 / 1.13 <= 0.5 * (double)rounds4;
  int variant;
  if (tile == 5 || (tile == 0 && v3_wins)) variant = fits31 ? 5 : 4;
  else variant = tile == 4 ? 4 : (tile == 2 || M < 4096 || K < 512 || N < 512) ? 2 : 4;
  const int TBM = variant == 2 ? 128 : 256, TBN = variant == 5 ? 256 : gemm::BN;
  plan->a = gemm::Args{A, lda, B, ldb, C, ldc, (int)M, (int)N, (int)K, scale_a, scale_b,
                       (int)((M + TBM - 1) / TBM), (int)((N + TBN - 1) / TBN), upd_u, upd_u_stride, upd_v, upd_v_stride,
                       (int)upd_j, bias, (int)activation, score_l, score_r, score_out, score_out ? (int)score_cols : 0,
                       mean_other, mean_other_stride, mean_out, mean_out_stride, absmax_out};
  plan->variant = variant;
  plan->blocks = ((int64_t)plan->a.nbm * plan->a.nbn + 7) & ~int64_t(7);
  return SPGNN_OK;
}

// the tile counts of `a` for another variant (a pair runs both products in the first one's kernel)
static void gemm_nt_retile(NtPlan* p, int variant) {
  const int TBM = variant == 2 ? 128 : 256, TBN = variant == 5 ? 256 : gemm::BN;
  p->a.nbm = (p->a.M + TBM - 1) / TBM;
  p->a.nbn = (p->a.N + TBN - 1) / TBN;
  p->variant = variant;
  p->blocks = ((int64_t)p->a.nbm * p->a.nbn + 7) & ~int64_t(7);
}

static size_t gemm_nt_lds(int variant) {
  const int TBM = variant == 2 ? 128 : 256;
  return variant == 5 ? 2 * 4 * 256 * gemm::PITCH * sizeof(_Float16)                       // 160 KB
                      : 2 * (2 * TBM + 2 * gemm::BN) * gemm::PITCH * sizeof(_Float16);
}

// `presplit`: SPGNN_PRESPLIT_B (1) = B pre-split, SPGNN_PRESPLIT_A | SPGNN_PRESPLIT_B (3) = both operands pre-split (a
// constant A - node data of a model's first layer - is split once per loader batch, next to weights split once per step)
static bool presplit_mask_ok(int32_t presplit) {
  const int32_t ps = presplit & ~SPGNN_GEMM_WIDE;
  return !(presplit & ~(SPGNN_PRESPLIT_A | SPGNN_PRESPLIT_B | SPGNN_GEMM_WIDE)) && (ps == 0 || ps == SPGNN_PRESPLIT_B || ps == (SPGNN_PRESPLIT_A | SPGNN_PRESPLIT_B));
}

static int gemm_nt_launch(const NtPlan& p0, const NtPlan* p1, int32_t presplit, hipStream_t st) {
  const int variant = p0.variant;
  const size_t lds_bytes = gemm_nt_lds(variant);
  const int threads = variant == 2 ? 256 : 512;
#define SPGNN_LAUNCH_NT(KERNEL_, BLOCKS_, ARG_)                                                                \
  { const int rc_ = spgnn_detail::ensure_dynamic_lds((const void*)(KERNEL_), (int)lds_bytes); if (rc_ != SPGNN_OK) return rc_; \
    hipLaunchKernelGGL((KERNEL_), dim3((unsigned)(BLOCKS_)), dim3(threads), lds_bytes, st, ARG_); }
#define SPGNN_LAUNCH_NT_PS(KERNEL_, BLOCKS_, ARG_, ...)                                                        \
  { if (ps == 3) SPGNN_LAUNCH_NT((KERNEL_<__VA_ARGS__ true, true>), BLOCKS_, ARG_)                              \
    else if (ps == 1) SPGNN_LAUNCH_NT((KERNEL_<__VA_ARGS__ false, true>), BLOCKS_, ARG_)                        \
    else SPGNN_LAUNCH_NT((KERNEL_<__VA_ARGS__ false, false>), BLOCKS_, ARG_) }
#define SPGNN_LAUNCH_NT_V2(KERNEL_, BLOCKS_, ARG_, WM_, WIDE_)                                                 \
  { if (ps == 3) SPGNN_LAUNCH_NT((KERNEL_<WM_, true, true, WIDE_>), BLOCKS_, ARG_)                              \
    else if (ps == 1) SPGNN_LAUNCH_NT((KERNEL_<WM_, false, true, WIDE_>), BLOCKS_, ARG_)                        \
    else SPGNN_LAUNCH_NT((KERNEL_<WM_, false, false, WIDE_>), BLOCKS_, ARG_) }
  const int32_t ps = presplit & ~SPGNN_GEMM_WIDE;
  const bool wide = (presplit & SPGNN_GEMM_WIDE) != 0;
  // the wide-range form exists in 128 x 128 tiles only: its two accumulator sets (128 registers) fit beside the staging sets
  // when a wave may use the AGPRs, i.e. with one 4-wave workgroup per CU; the 8-wave tiles would spill
  if (wide && variant != 2) return spgnn_detail::fail(SPGNN_ERR_ENUM, "spgnn_gemm_nt: the wide-range form runs in 128 x 128 tiles only (tile 0 or 2)");
  if (!p1) {
    if (variant == 5) SPGNN_LAUNCH_NT_PS(gemm::gemm_nt_f16x3_v3, p0.blocks, p0.a, )
    else if (variant == 4) SPGNN_LAUNCH_NT_V2(gemm::gemm_nt_f16x3_v2, p0.blocks, p0.a, 4, false)
    else if (wide) SPGNN_LAUNCH_NT_V2(gemm::gemm_nt_f16x3_v2, p0.blocks, p0.a, 2, true)
    else SPGNN_LAUNCH_NT_V2(gemm::gemm_nt_f16x3_v2, p0.blocks, p0.a, 2, false)
  } else {
    gemm::PairArgs pa{{p0.a, p1->a}, (unsigned)p0.blocks};
    const int64_t blocks = p0.blocks + p1->blocks;
    if (variant == 5) SPGNN_LAUNCH_NT_PS(gemm::gemm_nt_pair_v3, blocks, pa, )
    else if (variant == 4) SPGNN_LAUNCH_NT_V2(gemm::gemm_nt_pair_v2, blocks, pa, 4, false)
    else if (wide) SPGNN_LAUNCH_NT_V2(gemm::gemm_nt_pair_v2, blocks, pa, 2, true)
    else SPGNN_LAUNCH_NT_V2(gemm::gemm_nt_pair_v2, blocks, pa, 2, false)
  }
#undef SPGNN_LAUNCH_NT_V2
#undef SPGNN_LAUNCH_NT_PS
#undef SPGNN_LAUNCH_NT
  return spgnn_detail::check_launch("spgnn_gemm");
}

static int gemm_nt_impl(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int64_t M,
                        int64_t N, int64_t K, const float* scale_a, const float* scale_b, const float* upd_u,
                        int64_t upd_u_stride, const float* upd_v, int64_t upd_v_stride, int32_t upd_j, const float* bias,
                        int32_t activation, const float* score_l, const float* score_r, float* score_out, int32_t score_cols,
                        const float* mean_other, int64_t mean_other_stride, float* mean_out, int64_t mean_out_stride,
                        int32_t tile, int32_t b_presplit, spgnn_stream_t stream) {
  if (!presplit_mask_ok(b_presplit)) return spgnn_detail::fail_at(SPGNN_ERR_ENUM, __func__, __LINE__);
  NtPlan p;
  const int rc = gemm_nt_plan(A, lda, B, ldb, C, ldc, M, N, K, scale_a, scale_b, upd_u, upd_u_stride, upd_v, upd_v_stride, upd_j, bias,
                              activation, score_l, score_r, score_out, score_cols, mean_other, mean_other_stride, mean_out,
                              mean_out_stride, nullptr, tile, &p);
  if (rc != SPGNN_OK || p.blocks == 0) return rc;
  if ((b_presplit & SPGNN_GEMM_WIDE) && tile == 0 && p.variant != 2) gemm_nt_retile(&p, 2);       // the wide-range form: 128 x 128 tiles
  return gemm_nt_launch(p, nullptr, b_presplit, (hipStream_t)stream);
}

static int nt_plan_of(const spgnn_gemm_nt_problem* q, NtPlan* p) {
  if (!(q->drop_p >= 0.f && q->drop_p < 1.f) || (q->drop_p > 0.f && (q->N & 3)))
    return spgnn_detail::fail(SPGNN_ERR_SHAPE, "spgnn_gemm_nt_problem: drop_p outside [0, 1), or dropout with N % 4 != 0");
  const int rc = gemm_nt_plan(q->A, q->lda, q->B, q->ldb, q->C, q->ldc, q->M, q->N, q->K, q->scale_a, q->scale_b, q->upd_u,
                              q->upd_u_stride, q->upd_v, q->upd_v_stride, q->upd_j, q->bias, q->activation, q->score_l, q->score_r,
                              q->score_out, q->score_cols, q->addend, q->addend_stride, nullptr, 0, q->absmax_out, 0, p);
  if (rc == SPGNN_OK && q->drop_p > 0.f) {
    p->a.drop_p = q->drop_p; p->a.drop_inv = 1.f / (1.f - q->drop_p);
    p->a.drop_seed = q->drop_seed; p->a.drop_seed_off = q->drop_seed_offset;
  }
  return rc;
}

int spgnn_gemm_nt_problem_run(const spgnn_gemm_nt_problem* problem, int32_t b_presplit, spgnn_stream_t stream) {
  if (!problem) return spgnn_detail::fail_at(SPGNN_ERR_NULLPTR, __func__, __LINE__);
  if (!presplit_mask_ok(b_presplit)) return spgnn_detail::fail_at(SPGNN_ERR_ENUM, __func__, __LINE__);
  NtPlan p;
  const int rc = nt_plan_of(problem, &p);
  if (rc != SPGNN_OK || p.blocks == 0) return rc;
  if ((b_presplit & SPGNN_GEMM_WIDE) && p.variant != 2) gemm_nt_retile(&p, 2);
  return gemm_nt_launch(p, nullptr, b_presplit, (hipStream_t)stream);
}

int spgnn_gemm_nt_pair(const spgnn_gemm_nt_problem* first, const spgnn_gemm_nt_problem* second, int32_t b_presplit,
                       spgnn_stream_t stream) {
  if (!first || !second) return spgnn_detail::fail_at(SPGNN_ERR_NULLPTR, __func__, __LINE__);
  if (!presplit_mask_ok(b_presplit)) return spgnn_detail::fail_at(SPGNN_ERR_ENUM, __func__, __LINE__);
  if (first->drop_p > 0.f || second->drop_p > 0.f)                 // the pair kernels carry no dropout epilogue
    return spgnn_detail::fail(SPGNN_ERR_SHAPE, "spgnn_gemm_nt_pair: drop_p is a single-product option (spgnn_gemm_nt_problem_run)");
  NtPlan p0, p1;
  int rc = nt_plan_of(first, &p0);
  if (rc != SPGNN_OK) return rc;
  rc = nt_plan_of(second, &p1);
  if (rc != SPGNN_OK) return rc;
  if (b_presplit & SPGNN_GEMM_WIDE) {
    if (p0.variant != 2) gemm_nt_retile(&p0, 2);
    if (p1.variant != 2) gemm_nt_retile(&p1, 2);
  }
  if (p0.blocks == 0 && p1.blocks == 0) return SPGNN_OK;
  if (p0.blocks == 0) return gemm_nt_launch(p1, nullptr, b_presplit, (hipStream_t)stream);
  if (p1.blocks == 0) return gemm_nt_launch(p0, nullptr, b_presplit, (hipStream_t)stream);
#ifndef SPGNN_NT_PAIR_NO_ROUND_RULE
  // The first product chose 256 x 128 tiles from its own shape; count the PAIR's tiles before taking that over.  One workgroup
  // per CU means a started round costs a whole tile time however few tiles it holds (M = 76 410: 1 196 + 598 = 1 794 = 7.008
  // rounds; M = 9 641: 228 + 76 = 304 = 1.19).  128 x 128 tiles run two per CU - a round of 512 takes about the same time, 4 % more
  // per flop - and a last round of at most 256 has every CU to itself (~0.55 of a round).  Results are bit-identical in every
  // tile shape, so this only moves time.
  if (p0.variant == 4 && !(b_presplit & SPGNN_GEMM_WIDE)) {
    NtPlan q0 = p0, q1 = p1;
    gemm_nt_retile(&q1, 4);
    const int64_t t4 = (int64_t)q0.a.nbm * q0.a.nbn + (int64_t)q1.a.nbm * q1.a.nbn;
    gemm_nt_retile(&q0, 2); gemm_nt_retile(&q1, 2);
    const int64_t t2 = (int64_t)q0.a.nbm * q0.a.nbn + (int64_t)q1.a.nbm * q1.a.nbn;
    const double est4 = (double)((t4 + 255) / 256);
    const int64_t rem2 = t2 % 512;
    const double est2 = 1.04 * ((double)(t2 / 512) + (rem2 == 0 ? 0.0 : rem2 <= 256 ? 0.55 : 1.0));
    if (est2 < est4 - 0.01) { p0 = q0; p1 = q1; }
  }
#endif
  if (p1.variant != p0.variant) {                              // both in the first product's kernel
    if (p0.variant == 5 && !((int64_t)second->M * second->lda * 4 < (int64_t(1) << 31) && (int64_t)second->N * second->ldb * 4 < (int64_t(1) << 31)))
      return spgnn_detail::fail_at(SPGNN_ERR_SHAPE, __func__, __LINE__);
    gemm_nt_retile(&p1, p0.variant);
  }
  return gemm_nt_launch(p0, &p1, b_presplit, (hipStream_t)stream);
}

int spgnn_gemm_nt(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int64_t M,
                  int64_t N, int64_t K, const float* scale_a, const float* scale_b, const float* upd_u,
                  int64_t upd_u_stride, const float* upd_v, int64_t upd_v_stride, int32_t upd_j, const float* bias,
                  int32_t activation, const float* score_l, const float* score_r, float* score_out, int32_t score_cols,
                  int32_t b_presplit, spgnn_stream_t stream) {
  return gemm_nt_impl(A, lda, B, ldb, C, ldc, M, N, K, scale_a, scale_b, upd_u, upd_u_stride, upd_v, upd_v_stride, upd_j, bias,
                      activation, score_l, score_r, score_out, score_cols, nullptr, 0, nullptr, 0, 0, b_presplit, stream);
}

int spgnn_gemm_nt_tile(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int64_t M,
                       int64_t N, int64_t K, const float* scale_a, const float* scale_b, const float* upd_u,
                       int64_t upd_u_stride, const float* upd_v, int64_t upd_v_stride, int32_t upd_j, const float* bias,
                       int32_t activation, const float* score_l, const float* score_r, float* score_out, int32_t score_cols,
                       int32_t tile, int32_t b_presplit, spgnn_stream_t stream) {
  return gemm_nt_impl(A, lda, B, ldb, C, ldc, M, N, K, scale_a, scale_b, upd_u, upd_u_stride, upd_v, upd_v_stride, upd_j, bias,
                      activation, score_l, score_r, score_out, score_cols, nullptr, 0, nullptr, 0, tile, b_presplit, stream);
}

int spgnn_gemm_nt_headmean(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int64_t M,
                           int64_t N, int64_t K, const float* scale_a, const float* scale_b, const float* bias,
                           int32_t activation, const float* other_head, int64_t other_head_stride, float* mean_out,
                           int64_t mean_out_stride, int32_t b_presplit, spgnn_stream_t stream) {
  if (!other_head || !mean_out) return spgnn_detail::fail_at(SPGNN_ERR_NULLPTR, __func__, __LINE__);
  return gemm_nt_impl(A, lda, B, ldb, C, ldc, M, N, K, scale_a, scale_b, nullptr, 0, nullptr, 0, 0, bias, activation, nullptr,
                      nullptr, nullptr, 0, other_head, other_head_stride, mean_out, mean_out_stride, 0, b_presplit, stream);
}

int spgnn_gemm_nt_add(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int64_t M, int64_t N,
                      int64_t K, const float* scale_a, const float* scale_b, const float* bias, int32_t activation,
                      const float* addend, int64_t addend_stride, int32_t b_presplit, spgnn_stream_t stream) {
  if (!addend) return spgnn_detail::fail_at(SPGNN_ERR_NULLPTR, __func__, __LINE__);
  return gemm_nt_impl(A, lda, B, ldb, C, ldc, M, N, K, scale_a, scale_b, nullptr, 0, nullptr, 0, 0, bias, activation, nullptr,
                      nullptr, nullptr, 0, addend, addend_stride, nullptr, 0, 0, b_presplit, stream);
}

int spgnn_presplit(const float* partials, int64_t n_partials, const float* scale_in, float* scale_out,
                   const float* src0, int64_t ld0, int64_t R0, int64_t K0, float* dst0,
                   const float* src1, int64_t ld1, int64_t R1, int64_t K1, float* dst1, int32_t wide, spgnn_stream_t stream) {
  if (wide != 0 && wide != 1) return spgnn_detail::fail_at(SPGNN_ERR_ENUM, __func__, __LINE__);
  if ((!partials || n_partials <= 0) == (scale_in == nullptr)) return spgnn_detail::fail_at(SPGNN_ERR_NULLPTR, __func__, __LINE__);   // exactly one source of the scale
  if (n_partials > INT32_MAX || R0 < 0 || K0 <= 0 || R0 > INT32_MAX || K0 > INT32_MAX || (src1 && (R1 < 0 || K1 <= 0 || R1 > INT32_MAX || K1 > INT32_MAX)))
    return spgnn_detail::fail_at(SPGNN_ERR_SHAPE, __func__, __LINE__);
  if (!src0 || !dst0 || (src1 && !dst1)) return spgnn_detail::fail_at(SPGNN_ERR_NULLPTR, __func__, __LINE__);
  if (ld0 < ((K0 + 3) & ~int64_t(3)) || (ld0 & 3) || (reinterpret_cast<uintptr_t>(src0) & 15) || (reinterpret_cast<uintptr_t>(dst0) & 15) ||
      (src1 && (ld1 < ((K1 + 3) & ~int64_t(3)) || (ld1 & 3) || (reinterpret_cast<uintptr_t>(src1) & 15) || (reinterpret_cast<uintptr_t>(dst1) & 15))))
    return spgnn_detail::fail_at(SPGNN_ERR_STRIDE, __func__, __LINE__);        // rows 16-byte aligned, stride >= the width rounded up to 4; dst uses the same strides
  const int64_t groups = R0 * ((K0 + 3) / 4) + (src1 ? R1 * ((K1 + 3) / 4) : 0);
  if (groups == 0) return SPGNN_OK;
  int64_t blocks = (groups + 1023) / 1024;                         // four groups per thread
  if (blocks > 1024) blocks = 1024;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(gemm::presplit_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, partials, (int)n_partials, scale_in,
                     scale_out, src0, ld0, (int)R0, (int)K0, dst0, src1, ld1, (int)R1, (int)(src1 ? K1 : 0), dst1, (int)wide);
  return spgnn_detail::check_launch("spgnn_presplit");
}

// Tile rows of the weight-gradient kernel: 256 (gemm_tn<4>: 8 waves, one workgroup per CU) when the result has whole 256-row
// tiles and the reduction is long enough to fill the chip with them, else 128 (gemm_tn<2>).  `flags` may pin it.
static int gemm_tn_rows(int64_t R, int64_t M, int64_t N, int32_t flags) {
  if (flags & SPGNN_TN_TILE_256) return 256;
  if ((flags & SPGNN_TN_TILE_128) || (flags & SPGNN_TN_WIDE)) return 128;        // (the wide-range form: 128-row tiles, see gemm_nt_launch)
  // tools/tn_tiles.py (MI355X, one process; best split count of each form): R = 76 410: 1024 x 1063 623 -> 579 us, 1024 x 384
  // 223 -> 211, 512 x 768 224 -> 210, 256 x 384 71 -> 69, 256 x 256 53 = 53; R = 9 641: 99 -> 93, 47 -> 45, 46 -> 45, 28 = 28
  return (M % 256 == 0 && R >= 4096 && M * N >= 384 * 1024) ? 256 : 128;
}

static int gemm_tn_plan(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int64_t split_stride,
                        int32_t splits, int64_t R, int64_t M, int64_t N, const float* scale_a, const float* scale_b,
                        float* colsum_a, int64_t colsum_stride, int64_t colsum_split_stride, int tile_rows, gemm::ArgsTN* a, int64_t* blocks) {
  *blocks = 0;
  if (R < 0 || M <= 0 || N <= 0 || splits <= 0 || M > INT32_MAX || N > INT32_MAX) return spgnn_detail::fail_at(SPGNN_ERR_SHAPE, __func__, __LINE__);
  if (colsum_a && (colsum_stride < 1 || (splits > 1 && colsum_split_stride < 1))) return spgnn_detail::fail_at(SPGNN_ERR_STRIDE, __func__, __LINE__);
  if (!A || !B || !C) return spgnn_detail::fail_at(SPGNN_ERR_NULLPTR, __func__, __LINE__);
  if (lda < M || ldb < N || ldc < N || (lda & 3) || (ldb & 3) || (reinterpret_cast<uintptr_t>(A) & 15) ||
      (reinterpret_cast<uintptr_t>(B) & 15) || (splits > 1 && split_stride < M * ldc))
    return spgnn_detail::fail_at(SPGNN_ERR_STRIDE, __func__, __LINE__);
  int64_t rps = (R + splits - 1) / splits;
  rps = (rps + gemm::TBK - 1) / gemm::TBK * gemm::TBK;
  if (rps == 0) rps = gemm::TBK;
  *a = gemm::ArgsTN{A, lda, B, ldb, C, ldc, split_stride, R, (int)M, (int)N, rps, scale_a, scale_b,
                    (int)((M + tile_rows - 1) / tile_rows), (int)((N + gemm::BN - 1) / gemm::BN), colsum_a, colsum_stride, colsum_split_stride, (int)splits};
  *blocks = ((int64_t)a->nbm * a->nbn * splits + 7) & ~int64_t(7);
  if (*blocks > INT32_MAX) return spgnn_detail::fail_at(SPGNN_ERR_SHAPE, __func__, __LINE__);
  return SPGNN_OK;
}

static bool tn_flags_ok(int32_t f) {
  return !(f & ~(SPGNN_TN_B_PRESPLIT | SPGNN_TN_TILE_128 | SPGNN_TN_TILE_256 | SPGNN_TN_WIDE)) && (f & (SPGNN_TN_TILE_128 | SPGNN_TN_TILE_256)) != (SPGNN_TN_TILE_128 | SPGNN_TN_TILE_256);
}

// one product (p1 null) or a pair, in the kernel of `tile_rows` x 128 tiles
static int gemm_tn_launch(const gemm::ArgsTN& a0, int64_t b0, const gemm::ArgsTN* a1, int64_t b1, int tile_rows, bool bps, bool wide, hipStream_t st) {
  const size_t lds_bytes = tile_rows == 256 ? gemm::TnGeo<4>::lds_bytes : gemm::TnGeo<2>::lds_bytes;
  const int threads = tile_rows == 256 ? 512 : 256;
#define SPGNN_LAUNCH_TN(KERNEL_, BLOCKS_, ARG_)                                                                \
  { const int rc_ = spgnn_detail::ensure_dynamic_lds((const void*)(KERNEL_), (int)lds_bytes); if (rc_ != SPGNN_OK) return rc_; \
    hipLaunchKernelGGL((KERNEL_), dim3((unsigned)(BLOCKS_)), dim3(threads), lds_bytes, st, ARG_); }
#define SPGNN_LAUNCH_TN_W(KERNEL_, BLOCKS_, ARG_, WM_, WIDE_)                                                  \
  { if (bps) SPGNN_LAUNCH_TN((KERNEL_<WM_, true, WIDE_>), BLOCKS_, ARG_) else SPGNN_LAUNCH_TN((KERNEL_<WM_, false, WIDE_>), BLOCKS_, ARG_) }
#define SPGNN_LAUNCH_TN_V(KERNEL_, BLOCKS_, ARG_)                                                              \
  { if (tile_rows == 256) SPGNN_LAUNCH_TN_W(KERNEL_, BLOCKS_, ARG_, 4, false)                                  \
    else if (wide) SPGNN_LAUNCH_TN_W(KERNEL_, BLOCKS_, ARG_, 2, true)                                          \
    else SPGNN_LAUNCH_TN_W(KERNEL_, BLOCKS_, ARG_, 2, false) }
  if (wide && tile_rows != 128) return spgnn_detail::fail(SPGNN_ERR_ENUM, "spgnn_gemm_tn: the wide-range form runs in 128-row tiles only");
  if (!a1) SPGNN_LAUNCH_TN_V(gemm::gemm_tn_f16x3, b0, a0)
  else {
    gemm::PairArgsTN pa{{a0, *a1}, (unsigned)b0};
    SPGNN_LAUNCH_TN_V(gemm::gemm_tn_pair, b0 + b1, pa)
  }
#undef SPGNN_LAUNCH_TN_V
#undef SPGNN_LAUNCH_TN_W
#undef SPGNN_LAUNCH_TN
  return spgnn_detail::check_launch("spgnn_gemm");
}

int32_t spgnn_gemm_tn_tile_rows(int64_t R, int64_t M, int64_t N, int32_t flags) { return gemm_tn_rows(R, M, N, flags); }

int spgnn_gemm_tn(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int64_t split_stride,
                  int32_t splits, int64_t R, int64_t M, int64_t N, const float* scale_a, const float* scale_b,
                  float* colsum_a, int64_t colsum_stride, int64_t colsum_split_stride, spgnn_stream_t stream) {
  gemm::ArgsTN a; int64_t blocks;
  const int rows = gemm_tn_rows(R, M, N, 0);
  const int rc = gemm_tn_plan(A, lda, B, ldb, C, ldc, split_stride, splits, R, M, N, scale_a, scale_b, colsum_a, colsum_stride,
                              colsum_split_stride, rows, &a, &blocks);
  if (rc != SPGNN_OK) return rc;
  return gemm_tn_launch(a, blocks, nullptr, 0, rows, false, false, (hipStream_t)stream);
}

int spgnn_gemm_tn_problem_run(const spgnn_gemm_tn_problem* q, spgnn_stream_t stream) {
  if (!q) return spgnn_detail::fail_at(SPGNN_ERR_NULLPTR, __func__, __LINE__);
  if (!tn_flags_ok(q->flags)) return spgnn_detail::fail_at(SPGNN_ERR_ENUM, __func__, __LINE__);
  gemm::ArgsTN a; int64_t blocks;
  const int rows = gemm_tn_rows(q->R, q->M, q->N, q->flags);
  const int rc = gemm_tn_plan(q->A, q->lda, q->B, q->ldb, q->C, q->ldc, q->split_stride, q->splits, q->R, q->M, q->N, q->scale_a,
                              q->scale_b, q->colsum_a, q->colsum_stride, q->colsum_split_stride, rows, &a, &blocks);
  if (rc != SPGNN_OK) return rc;
  return gemm_tn_launch(a, blocks, nullptr, 0, rows, (q->flags & SPGNN_TN_B_PRESPLIT) != 0, (q->flags & SPGNN_TN_WIDE) != 0, (hipStream_t)stream);
}

int spgnn_gemm_tn_pair(const spgnn_gemm_tn_problem* first, const spgnn_gemm_tn_problem* second, spgnn_stream_t stream) {
  if (!first || !second) return spgnn_detail::fail_at(SPGNN_ERR_NULLPTR, __func__, __LINE__);
  if (!tn_flags_ok(first->flags) || !tn_flags_ok(second->flags) ||
      (first->flags & (SPGNN_TN_B_PRESPLIT | SPGNN_TN_WIDE)) != (second->flags & (SPGNN_TN_B_PRESPLIT | SPGNN_TN_WIDE)))
    return spgnn_detail::fail(SPGNN_ERR_ENUM, "spgnn_gemm_tn_pair: bad flags, or SPGNN_TN_B_PRESPLIT / SPGNN_TN_WIDE differ between the products (one kernel runs both)");
  // both products run in the FIRST one's tile shape (pass the larger one first), as the NT pair does
  const int rows = gemm_tn_rows(first->R, first->M, first->N, first->flags);
  gemm::ArgsTN a[2]; int64_t bl[2];
  const spgnn_gemm_tn_problem* q[2] = {first, second};
  for (int i = 0; i < 2; ++i) {
    const int rc = gemm_tn_plan(q[i]->A, q[i]->lda, q[i]->B, q[i]->ldb, q[i]->C, q[i]->ldc, q[i]->split_stride, q[i]->splits, q[i]->R,
                                q[i]->M, q[i]->N, q[i]->scale_a, q[i]->scale_b, q[i]->colsum_a, q[i]->colsum_stride,
                                q[i]->colsum_split_stride, rows, &a[i], &bl[i]);
    if (rc != SPGNN_OK) return rc;
  }
  if (bl[0] + bl[1] > INT32_MAX) return spgnn_detail::fail_at(SPGNN_ERR_SHAPE, __func__, __LINE__);
  return gemm_tn_launch(a[0], bl[0], &a[1], bl[1], rows, (first->flags & SPGNN_TN_B_PRESPLIT) != 0, (first->flags & SPGNN_TN_WIDE) != 0, (hipStream_t)stream);
}

int spgnn_sum_partials(const float* partials, int64_t split_stride, int32_t splits, int64_t n, float* out, spgnn_stream_t stream) {
  if (splits <= 0 || n < 0 || (n & 3) || (split_stride & 3) || split_stride < n) return spgnn_detail::fail_at(SPGNN_ERR_SHAPE, __func__, __LINE__);
  if (n == 0) return SPGNN_OK;
  if (!partials || !out) return spgnn_detail::fail_at(SPGNN_ERR_NULLPTR, __func__, __LINE__);
  if ((reinterpret_cast<uintptr_t>(partials) & 15) || (reinterpret_cast<uintptr_t>(out) & 15)) return spgnn_detail::fail_at(SPGNN_ERR_STRIDE, __func__, __LINE__);
  const int64_t n4 = n / 4;
  hipStream_t st = (hipStream_t)stream;
  if (splits <= 8)
    hipLaunchKernelGGL(gemm::sum_partials_kernel<1>, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, partials, split_stride / 4, (int)splits, n4, out);
  else if (splits <= 128)
    hipLaunchKernelGGL(gemm::sum_partials_kernel<4>, dim3((unsigned)((n4 + 63) / 64)), dim3(256), 0, st, partials, split_stride / 4, (int)splits, n4, out);
  else if (splits <= 512 || n4 >= 4096)
    hipLaunchKernelGGL(gemm::sum_partials_kernel<16>, dim3((unsigned)((n4 + 15) / 16)), dim3(256), 0, st, partials, split_stride / 4, (int)splits, n4, out);
  else      // many splits of a short row (the per-block column sums of spgnn_act_bwd_colsum: 2048 x 64..1024 floats): 64 lanes per
            // float4, 32 loads each instead of 128 in a dependent chain (12.7 -> ~5 us)
    hipLaunchKernelGGL(gemm::sum_partials_kernel<64>, dim3((unsigned)((n4 + 3) / 4)), dim3(256), 0, st, partials, split_stride / 4, (int)splits, n4, out);
  return spgnn_detail::check_launch("spgnn_gemm");
}

int spgnn_sum_partials_blockdiag(const float* partials, int64_t split_stride, int32_t splits, int32_t H, int32_t D, int32_t ld,
                                 float* out, spgnn_stream_t stream) {
  if (splits <= 0 || H <= 0 || D <= 0 || ld < H * D || split_stride < (int64_t)2 * H * ld) return spgnn_detail::fail_at(SPGNN_ERR_SHAPE, __func__, __LINE__);
  if (!partials || !out) return spgnn_detail::fail_at(SPGNN_ERR_NULLPTR, __func__, __LINE__);
  const int n = 2 * H * D;
  hipLaunchKernelGGL(gemm::sum_partials_blockdiag_kernel, dim3((unsigned)((n + 7) / 8)), dim3(256), 0, (hipStream_t)stream,
                     partials, split_stride, (int)splits, (int)H, (int)D, (int)ld, out);
  return spgnn_detail::check_launch("spgnn_gemm");
}

int spgnn_sum_partials_compact(const float* partials, int64_t split_stride, int32_t splits, int32_t M, int32_t N, int64_t ld_in,
                               float* out, int64_t out_stride, float* out2, int64_t out2_stride, int32_t split_col, float* extra,
                               int32_t extra_col, spgnn_stream_t stream) {
  if (splits <= 0 || M <= 0 || N <= 0 || ld_in < N || split_stride < (int64_t)M * ld_in || (extra && (extra_col < 0 || extra_col >= ld_in)))
    return spgnn_detail::fail_at(SPGNN_ERR_SHAPE, __func__, __LINE__);
  if (!partials || !out) return spgnn_detail::fail_at(SPGNN_ERR_NULLPTR, __func__, __LINE__);
  if (out2 ? (split_col <= 0 || split_col >= N || out_stride < split_col || out2_stride < N - split_col) : out_stride < N)
    return spgnn_detail::fail_at(SPGNN_ERR_STRIDE, __func__, __LINE__);
  if ((ld_in & 3) || (split_stride & 3) || (reinterpret_cast<uintptr_t>(partials) & 15)) return spgnn_detail::fail_at(SPGNN_ERR_STRIDE, __func__, __LINE__);   // float4 reads
  const int64_t n4 = (int64_t)M * (ld_in / 4);
  hipLaunchKernelGGL(gemm::sum_partials_compact_kernel, dim3((unsigned)((n4 + 63) / 64)), dim3(256), 0, (hipStream_t)stream,
                     partials, split_stride / 4, (int)splits, (int)M, (int)N, (int)(ld_in / 4), out, out_stride, out2, out2_stride,
                     (int)split_col, extra, (int)extra_col);
  return spgnn_detail::check_launch("spgnn_gemm");
}

int spgnn_weight_cat(const float* a, int64_t a_stride, int32_t rows_a, const float* b, int64_t b_stride, int32_t rows_b, int32_t K,
                     float* dst, int64_t dst_stride, float* dst_t, int64_t dst_t_stride, float* absmax_partials,
                     spgnn_stream_t stream) {
  const int64_t R = (int64_t)rows_a + rows_b;
  if (rows_a <= 0 || rows_b < 0 || K <= 0) return spgnn_detail::fail_at(SPGNN_ERR_SHAPE, __func__, __LINE__);
  if (!a || !dst || !absmax_partials || (rows_b > 0 && !b)) return spgnn_detail::fail_at(SPGNN_ERR_NULLPTR, __func__, __LINE__);
  if (a_stride < K || (rows_b > 0 && b_stride < K) || dst_stride < K || (dst_t && dst_t_stride < R)) return spgnn_detail::fail_at(SPGNN_ERR_STRIDE, __func__, __LINE__);
  const int64_t wide = dst_stride > K ? dst_stride : K;             // tiles cover dst's pad columns and dst_t's pad columns
  const int64_t tall = dst_t && dst_t_stride > R ? dst_t_stride : R;
  const dim3 grid((unsigned)((wide + 31) / 32), (unsigned)((tall + 31) / 32));
  hipLaunchKernelGGL(gemm::weight_cat_kernel, grid, dim3(256), 0, (hipStream_t)stream, a, a_stride, (int)rows_a, b, b_stride,
                     (int)rows_b, (int)K, dst, dst_stride, dst_t, dst_t_stride, absmax_partials);
  return spgnn_detail::check_launch("spgnn_gemm");
}

int64_t spgnn_weight_cat_partials(int32_t rows, int32_t K, int64_t dst_stride, int64_t dst_t_stride) {
  const int64_t wide = dst_stride > K ? dst_stride : K;
  const int64_t tall = dst_t_stride > rows ? dst_t_stride : rows;
  return ((wide + 31) / 32) * ((tall + 31) / 32);
}

int spgnn_pow2_scale(const float* x, int64_t x_stride, int64_t rows, int64_t cols, float* scale, float* workspace,
                     int32_t workspace_floats, spgnn_stream_t stream) {
  if (rows < 0 || cols <= 0 || cols > INT32_MAX || workspace_floats < 1) return spgnn_detail::fail_at(SPGNN_ERR_SHAPE, __func__, __LINE__);
  if (!scale || !workspace || (rows > 0 && !x)) return spgnn_detail::fail_at(SPGNN_ERR_NULLPTR, __func__, __LINE__);
  if (rows > 0 && ((x_stride & 3) || (reinterpret_cast<uintptr_t>(x) & 15))) return spgnn_detail::fail_at(SPGNN_ERR_STRIDE, __func__, __LINE__);
  hipStream_t st = (hipStream_t)stream;
  int64_t blocks = (rows * ((cols + 3) / 4) + 255) / 256;
  if (blocks > workspace_floats) blocks = workspace_floats;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(gemm::absmax_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, x_stride, rows, (int)cols, workspace);
  hipLaunchKernelGGL(gemm::scale_from_partials, dim3(1), dim3(64), 0, st, workspace, (int)blocks, 1.f, scale);
  return spgnn_detail::check_launch("spgnn_gemm");
}

int spgnn_scale_from_partials(const float* partials, int64_t n, float factor, float* scale, uint32_t* workspace,
                              spgnn_stream_t stream) {
  if (n < 0 || n > INT32_MAX || !(factor > 0.f)) return spgnn_detail::fail_at(SPGNN_ERR_SHAPE, __func__, __LINE__);
  if (!scale || (n > 0 && !partials)) return spgnn_detail::fail_at(SPGNN_ERR_NULLPTR, __func__, __LINE__);
  hipStream_t st = (hipStream_t)stream;
  if (n > 1024 && workspace && (reinterpret_cast<uintptr_t>(partials) & 15) == 0) {   // one 64-thread block: 13 us at n = 4776
    int blocks = (int)((n / 4 + 255) / 256);
    if (blocks > 64) blocks = 64;
    hipLaunchKernelGGL(gemm::scale_from_partials_mb, dim3(blocks), dim3(256), 0, st, partials, (int)n, factor, scale, workspace);
  } else {
    hipLaunchKernelGGL(gemm::scale_from_partials, dim3(1), dim3(64), 0, st, partials, (int)n, factor, scale);
  }
  return spgnn_detail::check_launch("spgnn_gemm");
}

int spgnn_sum_partials_multi(const spgnn_sum_job* jobs, int32_t n_jobs, spgnn_stream_t stream) {
  if (n_jobs < 0 || n_jobs > gemm::kMaxSumJobs) return spgnn_detail::fail_at(SPGNN_ERR_SHAPE, __func__, __LINE__);
  if (n_jobs == 0) return SPGNN_OK;
  if (!jobs) return spgnn_detail::fail_at(SPGNN_ERR_NULLPTR, __func__, __LINE__);
  gemm::SumJobs a{};
  unsigned total = 0;
  for (int i = 0; i < n_jobs; ++i) {
    const spgnn_sum_job& j = jobs[i];
    if (!j.partials || !j.out || j.splits <= 0) return spgnn_detail::fail_at(SPGNN_ERR_NULLPTR, __func__, __LINE__);
    int64_t blocks;
    if (j.kind == 0) {
      if (j.n <= 0 || (j.n & 3) || (j.split_stride & 3) || j.split_stride < j.n || (reinterpret_cast<uintptr_t>(j.partials) & 15) ||
          (reinterpret_cast<uintptr_t>(j.out) & 15)) return spgnn_detail::fail_at(SPGNN_ERR_SHAPE, __func__, __LINE__);
      const int64_t n4 = j.n / 4;
      blocks = j.splits <= 8 ? (n4 + 255) / 256 : j.splits <= 128 ? (n4 + 63) / 64 : (n4 + 15) / 16;
    } else if (j.kind == 1) {
      if (j.H <= 0 || j.D <= 0 || j.ld < j.H * j.D || j.split_stride < (int64_t)2 * j.H * j.ld) return spgnn_detail::fail_at(SPGNN_ERR_SHAPE, __func__, __LINE__);
      blocks = (2 * j.H * j.D + 7) / 8;
    } else if (j.kind == 2) {
      if (j.M <= 0 || j.N <= 0 || j.ld_in < j.N || j.split_stride < (int64_t)j.M * j.ld_in || (j.extra && (j.extra_col < 0 || j.extra_col >= j.ld_in)))
        return spgnn_detail::fail_at(SPGNN_ERR_SHAPE, __func__, __LINE__);
      if (j.out2 ? (j.split_col <= 0 || j.split_col >= j.N || j.out_stride < j.split_col || j.out2_stride < j.N - j.split_col) : j.out_stride < j.N)
        return spgnn_detail::fail_at(SPGNN_ERR_STRIDE, __func__, __LINE__);
      if ((j.ld_in & 3) || (j.split_stride & 3) || (reinterpret_cast<uintptr_t>(j.partials) & 15)) return spgnn_detail::fail_at(SPGNN_ERR_STRIDE, __func__, __LINE__);
      blocks = ((int64_t)j.M * (j.ld_in / 4) + 63) / 64;
    } else {
      return spgnn_detail::fail_at(SPGNN_ERR_ENUM, __func__, __LINE__);
    }
    if (blocks <= 0 || total + blocks > (1u << 30)) return spgnn_detail::fail_at(SPGNN_ERR_SHAPE, __func__, __LINE__);
    a.j[i] = j; a.first[i] = total; total += (unsigned)blocks;
  }
  a.first[n_jobs] = total; a.n = n_jobs;
  hipLaunchKernelGGL(gemm::sum_jobs_kernel, dim3(total), dim3(256), 0, (hipStream_t)stream, a);
  return spgnn_detail::check_launch("spgnn_sum_partials_multi");
}

int64_t spgnn_weight_prep_blocks(int32_t rows, int64_t dst_stride, int64_t dst_t_stride) {
  const int64_t tall = dst_t_stride > rows ? dst_t_stride : rows;
  return ((dst_stride + 31) / 32) * ((tall + 31) / 32);
}

int spgnn_weight_prep(const spgnn_weight_prep_layer* table, int32_t n_layers, int64_t total_blocks, float* workspace,
                      spgnn_stream_t stream) {
  if (n_layers < 0 || total_blocks < 0 || total_blocks > (1ll << 30)) return spgnn_detail::fail_at(SPGNN_ERR_SHAPE, __func__, __LINE__);
  if (n_layers == 0 || total_blocks == 0) return SPGNN_OK;
  if (!table || !workspace) return spgnn_detail::fail_at(SPGNN_ERR_NULLPTR, __func__, __LINE__);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(gemm::weight_prep_kernel, dim3((unsigned)total_blocks), dim3(256), 0, st, table, (int)n_layers, workspace, 0);
  hipLaunchKernelGGL(gemm::weight_prep_kernel, dim3((unsigned)total_blocks), dim3(256), 0, st, table, (int)n_layers, workspace, 1);
  return spgnn_detail::check_launch("spgnn_weight_prep");
}

int spgnn_linear_mean_fold_fwd(const float* w_fc, int64_t w_fc_stride, const float* w_res, int64_t w_res_stride, const float* bias,
                               const float* w_cls, int64_t w_cls_stride, const float* b_cls, int32_t H, int32_t D, int32_t F, int32_t J,
                               float* w_comb, int64_t w_comb_stride, uint16_t* w_comb_bf16, int64_t w_comb_bf16_stride,
                               float* b_mean, float* P, int64_t P_stride, float* c0, float* absmax_out, float* workspace,
                               uint32_t* tickets, int32_t x_block, spgnn_stream_t stream) {
  if (H <= 0 || D <= 0 || F <= 0 || J <= 0 || J > gemm::kFoldJ || (x_block != 0 && x_block != 1) || (!x_block && w_res))
    return spgnn_detail::fail_at(SPGNN_ERR_SHAPE, __func__, __LINE__);
  if (!w_fc || !w_cls || !w_comb || !P || !c0 || !workspace || !tickets) return spgnn_detail::fail_at(SPGNN_ERR_NULLPTR, __func__, __LINE__);
  const int64_t Kc = (int64_t)(H + x_block) * F;
  const int64_t Kp = (Kc + 15) / 16 * 16;                   // every image is written (zero padded) up to a multiple of 16 columns
  if (Kp > INT32_MAX || w_fc_stride < F || (w_res && w_res_stride < F) || w_cls_stride < D || w_comb_stride < Kp || P_stride < Kp ||
      (w_comb_bf16 && w_comb_bf16_stride < Kp))
    return spgnn_detail::fail_at(SPGNN_ERR_STRIDE, __func__, __LINE__);
  gemm::FoldFwd a{w_fc, w_fc_stride, w_res, w_res_stride, bias, w_cls, w_cls_stride, b_cls, w_comb, w_comb_stride,
                  w_comb_bf16, w_comb_bf16_stride, b_mean, P, P_stride, c0, absmax_out, workspace, tickets, (int)H, (int)D, (int)F, (int)J,
                  (int)Kc, (int)Kp};
  const dim3 grid((unsigned)(((Kp + 31) / 32) * gemm::kFoldQ + 1));
  if (J <= 8) hipLaunchKernelGGL(gemm::linear_mean_fold_fwd_kernel<8>, grid, dim3(1024), 0, (hipStream_t)stream, a);
  else if (J <= 16) hipLaunchKernelGGL(gemm::linear_mean_fold_fwd_kernel<16>, grid, dim3(1024), 0, (hipStream_t)stream, a);
  else if (J <= 24) hipLaunchKernelGGL(gemm::linear_mean_fold_fwd_kernel<24>, grid, dim3(1024), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(gemm::linear_mean_fold_fwd_kernel<32>, grid, dim3(1024), 0, (hipStream_t)stream, a);
  return spgnn_detail::check_launch("spgnn_linear_mean_fold_fwd");
}

int64_t spgnn_linear_mean_fold_workspace(int32_t H, int32_t F) {       // floats of `workspace` (either x_block); `tickets`: one uint32 per 32 columns of Kp, zero before the first call
  const int64_t Kp = (((int64_t)(H + 1) * F) + 15) / 16 * 16;
  return (Kp / 32 + 1) * gemm::kFoldQ * gemm::kFoldJ * 32;
}

int spgnn_linear_mean_fold_bwd(const float* M1, int64_t M1_stride, const float* cs, const float* w_cls, int64_t w_cls_stride,
                               const float* w_comb, int64_t w_comb_stride, const float* b_mean, int32_t H, int32_t D, int32_t F,
                               int32_t J, float* g_w_fc, int64_t g_w_fc_stride, float* g_w_res, int64_t g_w_res_stride, float* g_bias,
                               float* g_w_cls, int64_t g_w_cls_stride, int32_t x_block, spgnn_stream_t stream) {
  if (H <= 0 || D <= 0 || F <= 0 || J <= 0 || J > gemm::kFoldJ || (x_block != 0 && x_block != 1) || (!x_block && g_w_res))
    return spgnn_detail::fail_at(SPGNN_ERR_SHAPE, __func__, __LINE__);
  if (!M1 || !cs || !w_cls || !w_comb || !g_w_fc) return spgnn_detail::fail_at(SPGNN_ERR_NULLPTR, __func__, __LINE__);
  const int64_t Kc = (int64_t)(H + x_block) * F;
  if (M1_stride < Kc || w_cls_stride < D || w_comb_stride < Kc || g_w_fc_stride < F || (g_w_res && g_w_res_stride < F) ||
      (g_w_cls && g_w_cls_stride < D))
    return spgnn_detail::fail_at(SPGNN_ERR_STRIDE, __func__, __LINE__);
  const int64_t blocksA = ((int64_t)D * Kc + 255) / 256, blocksB = (D + 31) / 32;
  if (blocksA + blocksB > INT32_MAX) return spgnn_detail::fail_at(SPGNN_ERR_SHAPE, __func__, __LINE__);
  gemm::FoldBwd a{M1, M1_stride, cs, w_cls, w_cls_stride, w_comb, w_comb_stride, b_mean, g_w_fc, g_w_fc_stride, g_w_res, g_w_res_stride,
                  g_bias, g_w_cls, g_w_cls_stride, (int)H, (int)D, (int)F, (int)J, (int)Kc, (unsigned)blocksA};
  const dim3 grid((unsigned)(blocksA + blocksB));
  if (J <= 8) hipLaunchKernelGGL(gemm::linear_mean_fold_bwd_kernel<8>, grid, dim3(256), 0, (hipStream_t)stream, a);
  else if (J <= 16) hipLaunchKernelGGL(gemm::linear_mean_fold_bwd_kernel<16>, grid, dim3(256), 0, (hipStream_t)stream, a);
  else if (J <= 24) hipLaunchKernelGGL(gemm::linear_mean_fold_bwd_kernel<24>, grid, dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(gemm::linear_mean_fold_bwd_kernel<32>, grid, dim3(256), 0, (hipStream_t)stream, a);
  return spgnn_detail::check_launch("spgnn_linear_mean_fold_bwd");
}

}  // extern "C"
