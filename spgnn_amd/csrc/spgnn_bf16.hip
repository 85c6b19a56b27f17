// spgnn_bf16.hip — projection GEMMs of the bf16-STORAGE path (BASELINE config "st_gat_6 ... bf16": node-feature rows,
// projected rows and their gradients live in HBM as bfloat16; every sum is accumulated in fp32; parameters, optimizer
// state, attention scores and attention weights stay fp32).  gfx950 only.
//
//   spgnn_gemm_nt_bf16 : C[M,N] = A[M,K] * B[N,K]^T        (forward projections Y = X [W_fc;W_res]^T, input gradients)
//   spgnn_gemm_tn_bf16 : C[M,N] = A[R,M]^T * B[R,N]        (weight gradients, fp32 split-K partials)
//   spgnn_weight_cat_bf16, spgnn_cast_rows_bf16            (fp32 parameters / node data -> bf16 GEMM operands)
//
// With 16-bit operands the single-product v_mfma_f32_32x32x16_bf16 needs no conversion work at all, so the NT kernel
// stages its tiles global -> LDS by DMA (global_load_lds_dwordx4: no staging registers, no ds_write) with the next
// stage in flight across a raw s_barrier under a counted s_waitcnt (cdna_hip_programming.md, "Pipelining across
// barriers").  At the widths of the GNN (K = 128 ... 1064) these products are bound by HBM (read X once, write Y once),
// not by the matrix pipe: the tile order keeps an A row panel in one XCD's L2 for all of its column tiles.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "spgnn_hip.h"
#include "spgnn_internal.h"

namespace bfg {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x4 __attribute__((__vector_size__(4 * sizeof(short))));
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glb_void;

using spgnn_detail::check_launch;
using spgnn_detail::fail;

// 16 bytes of zeros in device memory: the DMA source of every k-chunk that lies beyond K (a DMA cannot be masked after
// the fact, so the ragged last stage is handled by redirecting the SOURCE address of whole 8-element chunks).
__device__ __attribute__((aligned(16))) uint16_t g_zero_chunk[8];

constexpr int BN = 128;          // tile columns (2 waves x 64)
constexpr int SUB = 32;          // elements per LDS sub-image row (64 bytes), two k16 MFMA steps
// One 32-element sub-image per stage (with two per stage and two buffers, round 2, the 1024 -> 1024 product took 183 us
// against 174 with three buffers of one) and NBUF stage buffers, a template parameter of the NT kernel: every buffer but the
// one being consumed holds a stage of DMA in flight.  Results do not depend on either: the k order does not change.
constexpr int KSUB = 1;          // sub-images per stage: BK = 32 KSUB elements per barrier

struct ArgsNT {
  const uint16_t* A; int64_t lda;
  const uint16_t* B; int64_t ldb;
  void* C; int64_t ldc;           // bf16 (c_f32 == 0) or fp32
  int M, N, K, K8;                // K8 = K rounded up to 8: columns [K, K8) of both operands hold zeros
  int nbm, nbn;
  const float* bias; int act;
  const float* sc_l; const float* sc_r; float* sc_out; int sc_cols;
  int sc_direct;                  // 1: one 64-column block per head - the dots go straight into the (M, 2H) [el | er] rows
};

__device__ __forceinline__ float elu_nb(float x) {          // same function as spgnn_gemm.hip's elu_fwd_nb
  const float p = x * (1.f + x * (0.5f + x * (1.f / 6 + x * (1.f / 24 + x * (1.f / 120 + x * (1.f / 720 + x * (1.f / 5040 +
                  x * (1.f / 40320 + x * (1.f / 362880)))))))));
  const float e = __builtin_amdgcn_exp2f(fmaxf(x, -126.f) * 1.44269504088896341f) - 1.f;
  const float n = x > -0.5f ? p : e;
  return x > 0.f ? x : n;
}

// sum over the 16 lanes of a DPP row (see spgnn_gemm.hip row16_sum for the packed-op hazard the empty asm statements avoid)
__device__ __forceinline__ float row16_sum(float x) {
  asm volatile("" : "+v"(x));
  x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xF, 0xF, true));
  asm volatile("" : "+v"(x));
  x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x4E, 0xF, 0xF, true));
  asm volatile("" : "+v"(x));
  x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x141, 0xF, 0xF, true));
  asm volatile("" : "+v"(x));
  x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x140, 0xF, 0xF, true));
  asm volatile("" : "+v"(x));
  return x;
}

__device__ __forceinline__ float4 round_bf16(float4 v, uint2& packed) {
  const f32x4 f = {v.x, v.y, v.z, v.w};
  union { bf16x4 h; uint2 u; } q;
  q.h = __builtin_convertvector(f, bf16x4);
  packed = q.u;
  return make_float4(__uint_as_float(q.u.x << 16), __uint_as_float(q.u.x & 0xFFFF0000u), __uint_as_float(q.u.y << 16),
                     __uint_as_float(q.u.y & 0xFFFF0000u));
}

// LDS sub-image of a ROWS x 32-element tile: 64-byte rows, lane-linear as the DMA writes them (piece q = 16 bytes: row
// q >> 2, position q & 3); position = chunk XOR ((row >> 2) & 3), applied on the global SOURCE address here and again
// on the fragment read (conflict-free ds_read_b128; same image as spgnn_gemm.hip's planes kernel).
template <int ROWS, int NT>
__device__ __forceinline__ void stage_image(const uint16_t* __restrict__ g, int64_t ld, int row0, int nrows, int k0, int K8,
                                            uint16_t* img) {
  constexpr int NP = ROWS * 4 / NT;
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const int q = threadIdx.x + NT * i;
    const int row = q >> 2;
    const int c = (q & 3) ^ ((row >> 2) & 3);
    int grow = row0 + row;
    grow = grow < nrows ? grow : nrows - 1;
    const int k = k0 + c * 8;
    // ONE DMA instruction per piece, whatever the lanes' sources: the per-lane address is chosen by a select and then
    // hidden from the optimiser (empty asm on its two halves).  Left visible, hipcc turns the select into two DMAs under
    // complementary exec masks (one with the zero chunk as a scalar base): the LDS base M0 is then taken from the first
    // ACTIVE lane while the hardware still adds lane * 16 - misplaced pieces - and the number of DMAs a wave issues
    // varies, which breaks the counted s_waitcnt vmcnt.  The LDS pointer handed over is the wave's piece 0 (uniform).
    const uint64_t pa = reinterpret_cast<uint64_t>(g + (int64_t)grow * ld + k), pz = reinterpret_cast<uint64_t>(g_zero_chunk);
    const uint64_t ps = k < K8 ? pa : pz;
    uint32_t lo = (uint32_t)ps, hi = (uint32_t)(ps >> 32);
    asm volatile("" : "+v"(lo), "+v"(hi));
    const void* src = reinterpret_cast<const void*>(((uint64_t)hi << 32) | lo);
    __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)(img + (q - lane) * 8), 16, 0, 0);
  }
}

__device__ __forceinline__ bf16x8 frag_swz(const uint16_t* img, int row, int chunk) {
  return *reinterpret_cast<const bf16x8*>(img + row * 32 + ((chunk ^ ((row >> 2) & 3)) << 3));
}

template <int WM, int WN, int MI, int NB> constexpr int nt_lds_bytes() {
  const int stages = NB * KSUB * (32 * MI * WM * SUB + 64 * WN * SUB) * 2, slabs = WM * WN * 32 * 68 * 4;
  return stages > slabs ? stages : slabs;
}

// WM row waves x WN column waves, each wave (32 MI) x 64 = MI x 2 MFMA tiles of 32 x 32: block tile (32 MI WM) x (64 WN).
//   <2, 2, 2> 128 x 128 (4 waves, two blocks per CU), <4, 2, 2> 256 x 128 (8 waves), <2, 4, 4> 256 x 256 (8 waves of
//   128 x 64: per MFMA 3/4 of the LDS fragment bytes of the 64 x 64 wave tile - with 64 x 64 wave tiles the fragment reads
//   of a stage take as many LDS cycles as its MFMAs take matrix-pipe cycles - and half the DMA bytes of the 256 x 128 tile).
template <int WM, int WN, int MI, bool F32OUT, int NBUF>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN == 4) ? 2 : 1) void gemm_nt_bf16(ArgsNT a) {
  constexpr int TBM = 32 * MI * WM, TBN = 64 * WN, NT = 64 * WM * WN;
  constexpr int A_IMG = TBM * SUB, B_IMG = TBN * SUB;             // elements per sub-image
  constexpr int STAGE = KSUB * (A_IMG + B_IMG);
  constexpr int LOADS = KSUB * (TBM * 4 / NT + TBN * 4 / NT);     // DMA instructions per thread and stage
  constexpr int EP = 68;                                          // epilogue slab pitch (floats)
  extern __shared__ __attribute__((aligned(16))) uint16_t smem[];      // ONE LDS object: stages, then the epilogue slabs

  const unsigned nb = gridDim.x, b = blockIdx.x;
  const unsigned tile = (b & 7u) * (nb >> 3) + (b >> 3);          // XCD-aware: an XCD walks consecutive tiles, column fastest
  if (tile >= (unsigned)(a.nbm * a.nbn)) return;
  const int bm = tile / a.nbn, bn = tile % a.nbn;
  const int row0 = bm * TBM, col0 = bn * TBN;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int fr = lane & 31, fh = lane >> 5;

  f32x16 acc[MI][2];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int nk = (a.K + KSUB * SUB - 1) / (KSUB * SUB);
#define SPGNN_STAGE_IN(T_)                                                                                   \
  {                                                                                                          \
    uint16_t* sb_ = smem + ((T_) % NBUF) * STAGE;                                                            \
    _Pragma("unroll") for (int u_ = 0; u_ < KSUB; ++u_) {                                                    \
      stage_image<TBM, NT>(a.A, a.lda, row0, a.M, ((T_) * KSUB + u_) * SUB, a.K8, sb_ + u_ * (A_IMG + B_IMG)); \
      stage_image<TBN, NT>(a.B, a.ldb, col0, a.N, ((T_) * KSUB + u_) * SUB, a.K8, sb_ + u_ * (A_IMG + B_IMG) + A_IMG); \
    }                                                                                                        \
  }
  // ONE barrier per stage, fragments read one k16 step ahead of the MFMAs that consume them: the reads of a stage's second
  // step are issued before the first step's MFMAs, the reads of the NEXT stage's first step (behind the barrier that makes
  // it visible) before the second step's - no MFMA waits on an LDS read issued right before it, and the barrier that
  // publishes stage t + 1 is also the one after which stage t's buffer may be refilled (every wave's reads of it have
  // returned: lgkmcnt(0) before the barrier).
  static_assert(KSUB == 1 && SUB == 32 && NBUF >= 3, "the pipelined loop walks two k16 steps per stage");
  auto read_frags = [&](const uint16_t* cb, const int ks, bf16x8 (&af)[MI], bf16x8 (&bf)[2]) {
#pragma unroll
    for (int i = 0; i < MI; ++i) af[i] = frag_swz(cb, wm * (32 * MI) + i * 32 + fr, ks * 2 + fh);
#pragma unroll
    for (int j = 0; j < 2; ++j) bf[j] = frag_swz(cb + A_IMG, wn * 64 + j * 32 + fr, ks * 2 + fh);
  };
  // the MFMAs of one k16 step in two parts: the first one alone (it waits for the step's fragments, read a whole step
  // ago), then - behind the reads of the NEXT step, pinned there by scheduling barriers: left to itself the compiler hoists
  // those reads above the first MFMA and then waits for all of them with one lgkmcnt(0) - the other 2 MI - 1
  auto mma_first = [&](const bf16x8 (&af)[MI], const bf16x8 (&bf)[2]) {
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[0], acc[0][0], 0, 0, 0);
  };
  auto mma_rest = [&](const bf16x8 (&af)[MI], const bf16x8 (&bf)[2]) {
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        if (i + j > 0) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
  };
  constexpr int P = NBUF;                                   // stages issued before the first one is consumed: every buffer
#pragma unroll
  for (int p_ = 0; p_ < P; ++p_)
    if (p_ < nk) SPGNN_STAGE_IN(p_)
  // stage 0 has landed when at most the P - 1 stages behind it are outstanding
  if (nk >= P) { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS * (P - 1)) : "memory"); }
  else { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
  __builtin_amdgcn_s_barrier();
  bf16x8 a0[MI], b0[2], a1[MI], b1[2];
  read_frags(smem, 0, a0, b0);
  for (int t = 0; t + 1 < nk; ++t) {                        // (the last stage is peeled: no branch around an MFMA)
    mma_first(a0, b0);
    __builtin_amdgcn_sched_barrier(0);
    read_frags(smem + (t % NBUF) * STAGE, 1, a1, b1);
    __builtin_amdgcn_sched_barrier(0);
    mma_rest(a0, b0);
    __builtin_amdgcn_sched_barrier(0);
    // issued so far: stages 0 .. t + P - 1; stage t + 1 has landed when at most the P - 2 behind it are outstanding
    if (t + P - 1 < nk) { asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(LOADS * (P - 2)) : "memory"); }
    else { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); }
    __builtin_amdgcn_s_barrier();                           // stage t + 1 visible; nobody reads stage t's buffer any more
    mma_first(a1, b1);
    __builtin_amdgcn_sched_barrier(0);
    if (t + P < nk) SPGNN_STAGE_IN(t + P)                   // into the buffer of stage t (P == NBUF)
    read_frags(smem + ((t + 1) % NBUF) * STAGE, 0, a0, b0);
    __builtin_amdgcn_sched_barrier(0);
    mma_rest(a1, b1);
    __builtin_amdgcn_sched_barrier(0);
  }
  mma_first(a0, b0);
  __builtin_amdgcn_sched_barrier(0);
  read_frags(smem + ((nk - 1) % NBUF) * STAGE, 1, a1, b1);
  __builtin_amdgcn_sched_barrier(0);
  mma_rest(a0, b0);
  mma_first(a1, b1);
  mma_rest(a1, b1);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                             // the epilogue slabs overlay the stage buffers
#undef SPGNN_STAGE_IN

  // Epilogue through LDS: an accumulator register holds one element of 32 different columns of one row, which would
  // store as 64-byte pieces.  Each wave parks a 32 x 64 half of its tile in its own slab and writes whole 128-byte
  // (bf16) / 256-byte (fp32) row segments.  (All waves passed the last barrier: the stage buffers are free.)
  float* slab = reinterpret_cast<float*>(smem) + wave * (32 * EP);
  const int r_in = lane >> 4, c4 = (lane & 15) * 4;
  const int col = col0 + wn * 64 + c4;
  const bool col_ok = col + 3 < a.N;                        // N % 4 == 0 (host check)
  float4 bq = make_float4(0.f, 0.f, 0.f, 0.f);
  if (a.bias && col_ok) bq = make_float4(a.bias[col], a.bias[col + 1], a.bias[col + 2], a.bias[col + 3]);
  const bool use_sc = a.sc_out != nullptr && col < a.sc_cols;   // wave-uniform: a wave's 64 columns are one block
  float4 sl = make_float4(0.f, 0.f, 0.f, 0.f), sr = sl;
  if (use_sc) {
    sl = make_float4(a.sc_l[col], a.sc_l[col + 1], a.sc_l[col + 2], a.sc_l[col + 3]);
    sr = make_float4(a.sc_r[col], a.sc_r[col + 1], a.sc_r[col + 2], a.sc_r[col + 3]);
  }
  // every 32-row half is a literal call: as a loop over i the body (two large paths) is no longer unrolled, and acc[i]
  // with a run-time i puts the accumulators in scratch memory
  auto do_half = [&](const int i) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) slab[((e & 3) + 8 * (e >> 2) + 4 * fh) * EP + j * 32 + fr] = acc[i][j][e];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float4 vv[8];
#pragma unroll
    for (int it = 0; it < 8; ++it) vv[it] = *reinterpret_cast<const float4*>(slab + (it * 4 + r_in) * EP + c4);
    // Interior halves (all 32 rows and 64 columns exist) take a path without per-row / per-column tests, the activation
    // chosen once, and streaming stores (the result is far larger than the caches and is read next by another kernel) -
    // the two changes that took 30 % off short products in the fp32 kernel (spgnn_gemm.hip, store_tile_through_lds).
    if (row0 + wm * (32 * MI) + i * 32 + 32 <= a.M && col0 + wn * 64 + 64 <= a.N) {
      if (!use_sc) {
#pragma unroll
        for (int it = 0; it < 8; ++it) { vv[it].x += bq.x; vv[it].y += bq.y; vv[it].z += bq.z; vv[it].w += bq.w; }
        if (a.act == SPGNN_ACT_ELU) {
#pragma unroll
          for (int it = 0; it < 8; ++it) { vv[it].x = elu_nb(vv[it].x); vv[it].y = elu_nb(vv[it].y); vv[it].z = elu_nb(vv[it].z); vv[it].w = elu_nb(vv[it].w); }
        } else if (a.act == SPGNN_ACT_TANH) {
#pragma unroll
          for (int it = 0; it < 8; ++it) { vv[it].x = tanhf(vv[it].x); vv[it].y = tanhf(vv[it].y); vv[it].z = tanhf(vv[it].z); vv[it].w = tanhf(vv[it].w); }
        } else if (a.act == SPGNN_ACT_RELU) {
#pragma unroll
          for (int it = 0; it < 8; ++it) { vv[it].x = fmaxf(vv[it].x, 0.f); vv[it].y = fmaxf(vv[it].y, 0.f); vv[it].z = fmaxf(vv[it].z, 0.f); vv[it].w = fmaxf(vv[it].w, 0.f); }
        }
      }
      const int64_t rbase = row0 + wm * (32 * MI) + i * 32 + r_in;
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int64_t row = rbase + it * 4;
        uint2 pk;
        const float4 vr = round_bf16(vv[it], pk);
        if (use_sc) {
          const float4 s_ = F32OUT ? vv[it] : vr;
          float pl = s_.x * sl.x + s_.y * sl.y + s_.z * sl.z + s_.w * sl.w;
          float pr = s_.x * sr.x + s_.y * sr.y + s_.z * sr.z + s_.w * sr.w;
          pl = row16_sum(pl); pr = row16_sum(pr);
          if ((lane & 15) == 0)
{
            const int nb_ = a.sc_cols >> 6, b_ = col >> 6;
            if (a.sc_direct) { float* p_ = a.sc_out + (int64_t)row * (2 * nb_); p_[b_] = pl; p_[nb_ + b_] = pr; }
            else *reinterpret_cast<float2*>(a.sc_out + ((int64_t)row * nb_ + b_) * 2) = make_float2(pl, pr);
          }
        }
        if (F32OUT) {
          typedef float f4v __attribute__((ext_vector_type(4)));
          __builtin_nontemporal_store(f4v{vv[it].x, vv[it].y, vv[it].z, vv[it].w},
                                      reinterpret_cast<f4v*>(reinterpret_cast<float*>(a.C) + row * a.ldc + col));
        } else {
          typedef unsigned u2v __attribute__((ext_vector_type(2)));
          __builtin_nontemporal_store(u2v{pk.x, pk.y}, reinterpret_cast<u2v*>(reinterpret_cast<uint16_t*>(a.C) + row * a.ldc + col));
        }
      }
    } else
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      const int row = row0 + wm * (32 * MI) + i * 32 + it * 4 + r_in;
      float4 v = vv[it];
      if (!use_sc) {                                        // act(C + bias); never combined with the score partials
        v.x += bq.x; v.y += bq.y; v.z += bq.z; v.w += bq.w;
        if (a.act == SPGNN_ACT_ELU) { v.x = elu_nb(v.x); v.y = elu_nb(v.y); v.z = elu_nb(v.z); v.w = elu_nb(v.w); }
        else if (a.act == SPGNN_ACT_TANH) { v.x = tanhf(v.x); v.y = tanhf(v.y); v.z = tanhf(v.z); v.w = tanhf(v.w); }
        else if (a.act == SPGNN_ACT_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      }
      uint2 pk;
      const float4 vr = round_bf16(v, pk);                  // what the consumers will read back
      if (use_sc) {                                         // el / er of DGL's GATConv: (ft * attn).sum(-1) over the STORED ft
        const float4 s_ = F32OUT ? v : vr;
        float pl = s_.x * sl.x + s_.y * sl.y + s_.z * sl.z + s_.w * sl.w;
        float pr = s_.x * sr.x + s_.y * sr.y + s_.z * sr.z + s_.w * sr.w;
        pl = row16_sum(pl); pr = row16_sum(pr);
        if ((lane & 15) == 0 && row < a.M)
{
          const int nb_ = a.sc_cols >> 6, b_ = col >> 6;
          if (a.sc_direct) { float* p_ = a.sc_out + (int64_t)row * (2 * nb_); p_[b_] = pl; p_[nb_ + b_] = pr; }
          else *reinterpret_cast<float2*>(a.sc_out + ((int64_t)row * nb_ + b_) * 2) = make_float2(pl, pr);
        }
      }
      if (row < a.M && col_ok) {
        if (F32OUT) *reinterpret_cast<float4*>(reinterpret_cast<float*>(a.C) + (int64_t)row * a.ldc + col) = v;
        else *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(a.C) + (int64_t)row * a.ldc + col) = pk;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  do_half(0);
  do_half(1);
  if constexpr (MI == 4) { do_half(2); do_half(3); }
  static_assert(MI == 2 || MI == 4, "wave tile is 64 or 128 rows");
}

// -------------------------------------------------------------------------------------------------
//   spgnn_gemm_tn_bf16 : C[M,N] = A[R,M]^T * B[R,N], reduction over the ROW index of both operands (weight
//   gradients: A = g_Y, B = X, R = node count).  Tiles arrive k-major (32 rows x 128 columns, 256-byte row segments),
//   go to LDS as they come ([k][m] images, pitch 160 elements = 320 B: conflict-free for the transposing read) and the
//   MFMA fragments are produced by ds_read_b64_tr_b16.  Register staging here (a padded image cannot be written by
//   DMA): the tile of stage t+1 is in flight during the MFMAs of stage t and stored after them, one barrier per stage.
//   Split over row ranges (one split per XCD when splits % 8 == 0), fp32 partial tiles, summed by the caller.
// -------------------------------------------------------------------------------------------------
constexpr int TBK = 32, TPITCH = 160, TTILE = TBK * TPITCH, TM = 128;

struct ArgsTN {
  const uint16_t* A; int64_t lda;                // (R, M)
  const uint16_t* B; int64_t ldb;                // (R, N)
  float* C; int64_t ldc; int64_t split_stride;
  int64_t R; int M, N;
  int64_t rows_per_split;
  int nbm, nbn;
  float* colsum; int64_t cs_stride, cs_split_stride;
  int by_xcd;
};

__device__ __forceinline__ bf16x8 tr_frag(const uint16_t* img, int m0, int k0, int lane) {
  const int g = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
  const int col = m0 + (g & 1) * 16 + 4 * pp;
  const int krow = k0 + (g >> 1) * 8 + q;
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + krow * TPITCH + col));
  const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + (krow + 4) * TPITCH + col));
  union { s16x4 v[2]; bf16x8 h; } u;
  u.v[0] = lo4; u.v[1] = hi4;
  return u.h;
}

__global__ __launch_bounds__(256, 2) void gemm_tn_bf16(ArgsTN a) {
  __shared__ __attribute__((aligned(16))) uint16_t smem[2 * 2 * TTILE];     // 2 stages x (A | B) = 40 KB

  unsigned tile, split;
  if (a.by_xcd) {                                 // a split's tiles all stream the same row range: give it to one XCD
    const unsigned L = blockIdx.x, tiles = (unsigned)(a.nbm * a.nbn);
    const unsigned xcd = L & 7u, q = L >> 3;
    split = xcd + 8u * (q / tiles);
    tile = q % tiles;
  } else {
    const unsigned nb = gridDim.x, b = blockIdx.x;
    tile = (b & 7u) * (nb >> 3) + (b >> 3);
    split = blockIdx.y;
    if (tile >= (unsigned)(a.nbm * a.nbn)) return;
  }
  const int bm = tile / a.nbn, bn = tile % a.nbn;
  const int m0 = bm * TM, n0 = bn * TM;
  const int64_t r_beg = (int64_t)split * a.rows_per_split;
  const int64_t r_end = r_beg + a.rows_per_split < a.R ? r_beg + a.rows_per_split : a.R;
  const int nk = r_beg < r_end ? (int)((r_end - r_beg + TBK - 1) / TBK) : 0;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 31, fh = lane >> 5;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // thread t: 8-column chunk (t & 15), rows (t >> 4) and (t >> 4) + 16 of the 32 x 128 tile.  Loads are unconditional
  // (rows past the range clamp to its last row, chunks past the width read column 0); both are zeroed at store time.
  const int tcol = (threadIdx.x & 15) * 8, trow = threadIdx.x >> 4;
  const int ca = m0 + tcol, cb_ = n0 + tcol;
  const bool va = ca < a.M, vb = cb_ < a.N;      // widths are multiples of 8 up to zero padding (host check)
  const uint16_t* pA = a.A + (va ? ca : 0);
  const uint16_t* pB = a.B + (vb ? cb_ : 0);
  const int64_t r_last = r_end - 1;
  const bool do_colsum = a.colsum != nullptr && bn == 0;
  float csum[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) csum[c] = 0.f;
  uint4 ra[2], rb[2];
#define SPGNN_TNB_LOAD(T_)                                                                                   \
  {                                                                                                          \
    const int64_t rr = r_beg + (int64_t)(T_) * TBK + trow;                                                   \
    _Pragma("unroll") for (int q = 0; q < 2; ++q) {                                                          \
      const int64_t r_ = rr + 16 * q < r_end ? rr + 16 * q : r_last;                                         \
      ra[q] = *reinterpret_cast<const uint4*>(pA + r_ * a.lda);                                              \
      rb[q] = *reinterpret_cast<const uint4*>(pB + r_ * a.ldb);                                              \
    }                                                                                                        \
  }
#define SPGNN_TNB_STORE(T_, BUF_)                                                                            \
  {                                                                                                          \
    uint16_t* ia = smem + (BUF_) * 2 * TTILE;                                                                \
    uint16_t* ib = ia + TTILE;                                                                               \
    _Pragma("unroll") for (int q = 0; q < 2; ++q) {                                                          \
      const bool rv = r_beg + (int64_t)(T_) * TBK + trow + 16 * q < r_end;                                   \
      const uint4 z = make_uint4(0u, 0u, 0u, 0u);                                                            \
      const uint4 xa = (rv && va) ? ra[q] : z, xb = (rv && vb) ? rb[q] : z;                                  \
      if (do_colsum) {                                                                                       \
        csum[0] += __uint_as_float(xa.x << 16); csum[1] += __uint_as_float(xa.x & 0xFFFF0000u);              \
        csum[2] += __uint_as_float(xa.y << 16); csum[3] += __uint_as_float(xa.y & 0xFFFF0000u);              \
        csum[4] += __uint_as_float(xa.z << 16); csum[5] += __uint_as_float(xa.z & 0xFFFF0000u);              \
        csum[6] += __uint_as_float(xa.w << 16); csum[7] += __uint_as_float(xa.w & 0xFFFF0000u);              \
      }                                                                                                      \
      *reinterpret_cast<uint4*>(ia + (trow + 16 * q) * TPITCH + tcol) = xa;                                  \
      *reinterpret_cast<uint4*>(ib + (trow + 16 * q) * TPITCH + tcol) = xb;                                  \
    }                                                                                                        \
  }
  if (nk > 0) {                                   // block-uniform; an empty split only writes zeros
    SPGNN_TNB_LOAD(0)
    SPGNN_TNB_STORE(0, 0)
    SPGNN_TNB_LOAD(1)                             // harmless when nk == 1 (clamped rows, never stored)
    __syncthreads();
    for (int t = 0; t < nk; ++t) {
      const uint16_t* ia = smem + (t & 1) * 2 * TTILE;
      const uint16_t* ib = ia + TTILE;
#pragma unroll
      for (int ks = 0; ks < TBK / 16; ++ks) {
        bf16x8 af[2], bf[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) af[i] = tr_frag(ia, wm * 64 + i * 32, ks * 16, lane);
#pragma unroll
        for (int j = 0; j < 2; ++j) bf[j] = tr_frag(ib, wn * 64 + j * 32, ks * 16, lane);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
      }
      if (t + 1 < nk) SPGNN_TNB_STORE(t + 1, (t + 1) & 1)  // the other buffer: last read in stage t - 1, one barrier ago
      SPGNN_TNB_LOAD(t + 2)
      __syncthreads();
    }
  }
#undef SPGNN_TNB_LOAD
#undef SPGNN_TNB_STORE

  if (do_colsum) {                                // fold the 16 row groups (tid >> 4) that share a column chunk
    float* red = reinterpret_cast<float*>(smem);
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 8; ++c) red[(threadIdx.x >> 4) * 128 + tcol + c] = csum[c];
    __syncthreads();
    if (threadIdx.x < 128 && m0 + (int)threadIdx.x < a.M) {
      float t_ = 0.f;
#pragma unroll
      for (int g_ = 0; g_ < 16; ++g_) t_ += red[g_ * 128 + threadIdx.x];
      a.colsum[(int64_t)split * a.cs_split_stride + (int64_t)(m0 + threadIdx.x) * a.cs_stride] = t_;
    }
  }
  float* Cp = a.C + (int64_t)split * a.split_stride;
  if (m0 + TM <= a.M && n0 + BN <= a.N) {         // interior tile (block-uniform): straight-line stores, no per-element tests
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        float* cp = Cp + (int64_t)(m0 + wm * 64 + i * 32 + 4 * fh) * a.ldc + n0 + wn * 64 + j * 32 + fr;
#pragma unroll
        for (int e = 0; e < 16; ++e) cp[(int64_t)((e & 3) + 8 * (e >> 2)) * a.ldc] = acc[i][j][e];
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
        if (row < a.M) Cp[(int64_t)row * a.ldc + col] = acc[i][j][e];
      }
    }
}

// [w_a ; w_b] (fp32 parameters, row blocks) -> bf16 operand W (R, ldw) and, optionally, its transpose (K, ldt); pad
// columns are written as zeros.  One 32 x 32 tile per block through LDS.
__device__ __forceinline__ void weight_cat_bf16_tile(const float* __restrict__ A, int64_t lda, int ra,
                                                     const float* __restrict__ B, int64_t ldb, int rb, int K,
                                                     uint16_t* __restrict__ W, int64_t ldw,
                                                     uint16_t* __restrict__ Wt, int64_t ldt, int r0, int k0, float (*tile)[33]) {
  const int R = ra + rb;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int r = r0 + ty + 8 * q, k = k0 + tx;
    float v = 0.f;
    if (r < R && k < K) v = r < ra ? A[(int64_t)r * lda + k] : B[(int64_t)(r - ra) * ldb + k];
    tile[ty + 8 * q][tx] = v;
    if (r < R && k < ldw) {
      const __bf16 h = (__bf16)v;
      W[(int64_t)r * ldw + k] = *reinterpret_cast<const uint16_t*>(&h);
    }
  }
  if (!Wt) return;
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int k = k0 + ty + 8 * q, r = r0 + tx;
    if (k < K && r < ldt) {
      const __bf16 h = (__bf16)(r < R ? tile[tx][ty + 8 * q] : 0.f);
      Wt[(int64_t)k * ldt + r] = *reinterpret_cast<const uint16_t*>(&h);
    }
  }
}

__global__ __launch_bounds__(256) void weight_cat_bf16_kernel(const float* __restrict__ A, int64_t lda, int ra,
                                                              const float* __restrict__ B, int64_t ldb, int rb, int K,
                                                              uint16_t* __restrict__ W, int64_t ldw,
                                                              uint16_t* __restrict__ Wt, int64_t ldt) {
  __shared__ float tile[32][33];
  weight_cat_bf16_tile(A, lda, ra, B, ldb, rb, K, W, ldw, Wt, ldt, blockIdx.y * 32, blockIdx.x * 32, tile);
}

// every projection layer's operands in ONE launch: block b belongs to the job whose [first_block, next first_block) holds it
// (a handful of jobs: linear scan), and is tile (b' % tiles_x, b' / tiles_x) of that job - the arithmetic of the single kernel
__global__ __launch_bounds__(256) void weight_cat_bf16_multi_kernel(const spgnn_weight_cat_bf16_job* __restrict__ jobs, int n_jobs) {
  __shared__ float tile[32][33];
  int j = 0;
  while (j + 1 < n_jobs && (int)blockIdx.x >= jobs[j + 1].first_block) ++j;
  const spgnn_weight_cat_bf16_job q = jobs[j];
  const int local = (int)blockIdx.x - q.first_block;
  weight_cat_bf16_tile(q.a, q.a_stride, q.rows_a, q.b, q.b_stride, q.rows_b, q.K, q.w, q.w_stride, q.w_t, q.w_t_stride,
                       (local / q.tiles_x) * 32, (local % q.tiles_x) * 32, tile);
}

// x (N, K) fp32 -> y (N, ldy) bf16, columns [K, ldy) zero
__global__ __launch_bounds__(256) void cast_rows_bf16_kernel(const float* __restrict__ x, int64_t ldx, int64_t N, int K,
                                                             uint16_t* __restrict__ y, int64_t ldy) {
  const int c4 = (int)(ldy >> 2);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N * c4; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / c4; const int c = (int)(i % c4) * 4;
    const float* s = x + row * ldx + c;
    float4 v = make_float4(c < K ? s[0] : 0.f, c + 1 < K ? s[1] : 0.f, c + 2 < K ? s[2] : 0.f, c + 3 < K ? s[3] : 0.f);
    uint2 pk;
    (void)round_bf16(v, pk);
    *reinterpret_cast<uint2*>(y + row * ldy + c) = pk;
  }
}

}  // namespace bfg

extern "C" {

static int gemm_nt_bf16_impl(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, void* C, int64_t ldc, int32_t c_is_f32,
                             int64_t M, int64_t N, int64_t K, const float* bias, int32_t act, const float* score_l,
                             const float* score_r, float* score_out, int32_t score_cols, int32_t tile, int32_t score_layout,
                             spgnn_stream_t stream) {
  using namespace bfg;
  if (score_layout != 0 && score_layout != 1) return fail(SPGNN_ERR_ENUM, "spgnn_gemm_nt_bf16: score_layout must be 0 or 1");
  if (tile != 0 && tile != 2 && tile != 4 && tile != 5) return fail(SPGNN_ERR_ENUM, "spgnn_gemm_nt_bf16: tile must be 0, 2, 4 or 5");
  if (M < 0 || N <= 0 || K <= 0 || N > (1 << 24) || K > (1 << 24)) return fail(SPGNN_ERR_SHAPE, "spgnn_gemm_nt_bf16: bad M/N/K");
  if (M == 0) return SPGNN_OK;
  if (!A || !B || !C) return fail(SPGNN_ERR_NULLPTR, "spgnn_gemm_nt_bf16: null pointer");
  const int64_t K8 = (K + 7) / 8 * 8;
  if (lda < K8 || ldb < K8 || (lda & 7) || (ldb & 7) || (reinterpret_cast<uintptr_t>(A) & 15) || (reinterpret_cast<uintptr_t>(B) & 15))
    return fail(SPGNN_ERR_STRIDE, "spgnn_gemm_nt_bf16: operand rows must be 16-byte aligned with stride >= K rounded up to 8 "
                                  "(columns [K, K8) zero)");
  if ((N & 3) || ldc < N || (ldc & 3) || (reinterpret_cast<uintptr_t>(C) & (c_is_f32 ? 15 : 7)))
    return fail(SPGNN_ERR_STRIDE, "spgnn_gemm_nt_bf16: N and the C row stride must be multiples of 4, C rows vector aligned");
  if (act < SPGNN_ACT_NONE || act > SPGNN_ACT_RELU) return fail(SPGNN_ERR_ENUM, "spgnn_gemm_nt_bf16: activation");
  if (score_out) {
    if (!score_l || !score_r || score_cols <= 0 || (score_cols & 63) || score_cols > N || bias || act != SPGNN_ACT_NONE)
      return fail(SPGNN_ERR_SHAPE, "spgnn_gemm_nt_bf16: score partials need score_l/r, score_cols % 64 == 0 <= N, no bias/act");
  }
  // 256 x 256 tiles for deep products whose rounds over the 256 CUs come out nearly full (<= 8 % of the last round idle, as
  // the fp32 kernel's rule; measured at M = 76 410: 1024 x 1024 255 -> 237 us, but 384 -> 1024 124 -> 130 us: 12 stages do
  // not repay the larger tile's prologue and epilogue); else 256-row tiles when they still give every CU two rounds of
  // work; else 128-row tiles (two blocks per CU)
  const double r3 = (double)(((M + 255) / 256) * ((N + 255) / 256)) / 256.0;
  const bool sq = tile == 5 || (tile == 0 && M >= 4096 && K >= 512 && N >= 512 && (double)(int64_t)(r3 + 0.999999) <= 1.08 * r3);
  const int tbn = sq ? 256 : BN;
  const int64_t nbn = (N + tbn - 1) / tbn;
  const bool big = sq || tile == 4 || (tile == 0 && ((M + 255) / 256) * nbn >= 1024);
  const int tbm = big ? 256 : 128;
  const int64_t nbm = (M + tbm - 1) / tbm;
  if (nbm * nbn > (1ll << 30)) return fail(SPGNN_ERR_SHAPE, "spgnn_gemm_nt_bf16: too many tiles");
  ArgsNT a{A, lda, B, ldb, C, ldc, (int)M, (int)N, (int)K, (int)K8, (int)nbm, (int)nbn, bias, act, score_l, score_r, score_out,
           score_out ? score_cols : 0, score_layout};
  const unsigned grid = (unsigned)((nbm * nbn + 7) / 8 * 8);
  hipStream_t st = (hipStream_t)stream;
#define SPGNN_NT_LAUNCH(WM_, WN_, MI_, F32_, NB_)                                                                  \
  {                                                                                                               \
    constexpr int lds_ = nt_lds_bytes<WM_, WN_, MI_, NB_>();                                                      \
    const int rc_ = spgnn_detail::ensure_dynamic_lds((const void*)gemm_nt_bf16<WM_, WN_, MI_, F32_, NB_>, lds_);  \
    if (rc_ != SPGNN_OK) return rc_;                                                                              \
    hipLaunchKernelGGL((gemm_nt_bf16<WM_, WN_, MI_, F32_, NB_>), dim3(grid), dim3(64 * WM_ * WN_), lds_, st, a);  \
  }
  // stage buffers (tools/gemm_bf16_ab.py, M = 76 410, us with 3 / 4 buffers): 256 x 256 tiles 1024 -> 1024 178 / 174; 256 x 128
  // tiles 1024 -> 512 114 / 102, 512 -> 512 65 / 59; 128 x 128 tiles (two blocks per CU) 512 -> 256 34 / 36, 256 -> 128 16 / 18
  if (sq) {
    if (c_is_f32) SPGNN_NT_LAUNCH(2, 4, 4, true, 4) else SPGNN_NT_LAUNCH(2, 4, 4, false, 4)
  } else if (big) {
    if (c_is_f32) SPGNN_NT_LAUNCH(4, 2, 2, true, 4) else SPGNN_NT_LAUNCH(4, 2, 2, false, 4)
  } else {
    if (c_is_f32) SPGNN_NT_LAUNCH(2, 2, 2, true, 3) else SPGNN_NT_LAUNCH(2, 2, 2, false, 3)
  }
#undef SPGNN_NT_LAUNCH
  return check_launch("spgnn_gemm_nt_bf16");
}

int spgnn_gemm_nt_bf16(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, void* C, int64_t ldc, int32_t c_is_f32,
                       int64_t M, int64_t N, int64_t K, const float* bias, int32_t act, const float* score_l,
                       const float* score_r, float* score_out, int32_t score_cols, spgnn_stream_t stream) {
  return gemm_nt_bf16_impl(A, lda, B, ldb, C, ldc, c_is_f32, M, N, K, bias, act, score_l, score_r, score_out, score_cols, 0, 0, stream);
}

int spgnn_gemm_nt_bf16_tile(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, void* C, int64_t ldc, int32_t c_is_f32,
                            int64_t M, int64_t N, int64_t K, const float* bias, int32_t act, const float* score_l,
                            const float* score_r, float* score_out, int32_t score_cols, int32_t tile, spgnn_stream_t stream) {
  return gemm_nt_bf16_impl(A, lda, B, ldb, C, ldc, c_is_f32, M, N, K, bias, act, score_l, score_r, score_out, score_cols, tile, 0, stream);
}

int spgnn_gemm_nt_bf16_scores(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, void* C, int64_t ldc, int32_t c_is_f32,
                              int64_t M, int64_t N, int64_t K, const float* score_l, const float* score_r, float* score_out,
                              int32_t score_cols, int32_t score_layout, spgnn_stream_t stream) {
  return gemm_nt_bf16_impl(A, lda, B, ldb, C, ldc, c_is_f32, M, N, K, nullptr, SPGNN_ACT_NONE, score_l, score_r, score_out, score_cols, 0,
                           score_layout, stream);
}

int spgnn_gemm_tn_bf16(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, float* C, int64_t ldc,
                       int64_t split_stride, int32_t splits, int64_t R, int64_t M, int64_t N, float* colsum,
                       int64_t colsum_stride, int64_t colsum_split_stride, spgnn_stream_t stream) {
  using namespace bfg;
  if (R < 0 || M <= 0 || N <= 0 || splits <= 0 || M > (1 << 24) || N > (1 << 24)) return fail(SPGNN_ERR_SHAPE, "spgnn_gemm_tn_bf16: bad R/M/N/splits");
  if (!A || !B || !C) return fail(SPGNN_ERR_NULLPTR, "spgnn_gemm_tn_bf16: null pointer");
  const int64_t M8 = (M + 7) / 8 * 8, N8 = (N + 7) / 8 * 8;
  if (lda < M8 || ldb < N8 || (lda & 7) || (ldb & 7) || (reinterpret_cast<uintptr_t>(A) & 15) || (reinterpret_cast<uintptr_t>(B) & 15))
    return fail(SPGNN_ERR_STRIDE, "spgnn_gemm_tn_bf16: operand rows must be 16-byte aligned with stride >= width rounded up to 8 "
                                  "(pad columns zero)");
  if (ldc < N || split_stride < M * ldc) return fail(SPGNN_ERR_STRIDE, "spgnn_gemm_tn_bf16: C strides");
  const int64_t nbm = (M + TM - 1) / TM, nbn = (N + TM - 1) / TM;
  const int64_t rps = ((R + splits - 1) / splits + TBK - 1) / TBK * TBK;
  const int by_xcd = (splits % 8 == 0) ? 1 : 0;
  ArgsTN a{A, lda, B, ldb, C, ldc, split_stride, R, (int)M, (int)N, rps > 0 ? rps : TBK, (int)nbm, (int)nbn, colsum, colsum_stride,
           colsum_split_stride, by_xcd};
  hipStream_t st = (hipStream_t)stream;
  if (by_xcd) hipLaunchKernelGGL(gemm_tn_bf16, dim3((unsigned)(nbm * nbn * splits)), dim3(256), 0, st, a);
  else hipLaunchKernelGGL(gemm_tn_bf16, dim3((unsigned)((nbm * nbn + 7) / 8 * 8), (unsigned)splits), dim3(256), 0, st, a);
  return check_launch("spgnn_gemm_tn_bf16");
}

int spgnn_weight_cat_bf16(const float* a, int64_t a_stride, int32_t rows_a, const float* b, int64_t b_stride, int32_t rows_b,
                          int32_t K, uint16_t* w, int64_t w_stride, uint16_t* w_t, int64_t w_t_stride, spgnn_stream_t stream) {
  using namespace bfg;
  if (rows_a <= 0 || rows_b < 0 || K <= 0) return fail(SPGNN_ERR_SHAPE, "spgnn_weight_cat_bf16: bad rows/K");
  if (!a || !w || (rows_b > 0 && !b)) return fail(SPGNN_ERR_NULLPTR, "spgnn_weight_cat_bf16: null pointer");
  const int R = rows_a + rows_b;
  if (a_stride < K || (rows_b > 0 && b_stride < K) || w_stride < K || (w_t && w_t_stride < R))
    return fail(SPGNN_ERR_STRIDE, "spgnn_weight_cat_bf16: row stride smaller than row");
  const int64_t kmax = w_stride > K ? w_stride : K, rmax = (w_t && w_t_stride > R) ? w_t_stride : R;
  hipLaunchKernelGGL(weight_cat_bf16_kernel, dim3((unsigned)((kmax + 31) / 32), (unsigned)((rmax + 31) / 32)), dim3(256), 0,
                     (hipStream_t)stream, a, a_stride, rows_a, b, b_stride, rows_b, K, w, w_stride, w_t, w_t_stride);
  return check_launch("spgnn_weight_cat_bf16");
}

int32_t spgnn_weight_cat_bf16_blocks(int32_t rows, int32_t K, int64_t w_stride, int64_t w_t_stride, int32_t* tiles_x) {
  const int64_t kmax = w_stride > K ? w_stride : K, rmax = w_t_stride > rows ? w_t_stride : rows;
  const int32_t tx = (int32_t)((kmax + 31) / 32), ty = (int32_t)((rmax + 31) / 32);
  if (tiles_x) *tiles_x = tx;
  return tx * ty;
}

int spgnn_weight_cat_bf16_multi(const spgnn_weight_cat_bf16_job* jobs, int32_t n_jobs, int32_t total_blocks, spgnn_stream_t stream) {
  using namespace bfg;
  if (n_jobs <= 0 || total_blocks <= 0) return fail(SPGNN_ERR_SHAPE, "spgnn_weight_cat_bf16_multi: bad n_jobs / total_blocks");
  if (!jobs) return fail(SPGNN_ERR_NULLPTR, "spgnn_weight_cat_bf16_multi: null pointer");
  hipLaunchKernelGGL(weight_cat_bf16_multi_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, jobs, (int)n_jobs);
  return check_launch("spgnn_weight_cat_bf16_multi");
}

int spgnn_cast_rows_bf16(const float* x, int64_t x_stride, int64_t N, int32_t K, uint16_t* y, int64_t y_stride,
                         spgnn_stream_t stream) {
  using namespace bfg;
  if (N < 0 || K <= 0) return fail(SPGNN_ERR_SHAPE, "spgnn_cast_rows_bf16: bad N/K");
  if (N == 0) return SPGNN_OK;
  if (!x || !y) return fail(SPGNN_ERR_NULLPTR, "spgnn_cast_rows_bf16: null pointer");
  if (x_stride < K || y_stride < K || (y_stride & 3) || (reinterpret_cast<uintptr_t>(y) & 7))
    return fail(SPGNN_ERR_STRIDE, "spgnn_cast_rows_bf16: y rows must be 8-byte aligned with stride % 4 == 0");
  int64_t blocks = (N * (y_stride / 4) + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  hipLaunchKernelGGL(cast_rows_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, x_stride, N, K, y, y_stride);
  return check_launch("spgnn_cast_rows_bf16");
}

}  // extern "C"
