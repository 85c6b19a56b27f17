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
  const float* sA; const float* sB;             // device scalars (power-of-two scales) or null (= 1)
  int nbm, nbn;
};

__device__ __forceinline__ void split4(float4 v, float s, half4& hi, half4& lo) {
  const float x0 = v.x * s, x1 = v.y * s, x2 = v.z * s, x3 = v.w * s;
  hi = half4{(_Float16)x0, (_Float16)x1, (_Float16)x2, (_Float16)x3};
  lo = half4{(_Float16)(x0 - (float)hi[0]), (_Float16)(x1 - (float)hi[1]), (_Float16)(x2 - (float)hi[2]),
             (_Float16)(x3 - (float)hi[3])};
}

// one 128 x 32 fp32 tile = 1024 float4; thread t loads float4 #(t + 256 i), i = 0..3: row = idx / 8, k4 = idx % 8
__device__ __forceinline__ void load_tile(const float* __restrict__ base, int64_t ld, int row0, int nrows, int k0, int K,
                                          float4 (&r)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = threadIdx.x + kThreads * i;
    const int row = row0 + (idx >> 3), k = k0 + (idx & 7) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < nrows) {
      const float* p = base + (int64_t)row * ld + k;
      if (k + 3 < K) v = *reinterpret_cast<const float4*>(p);
      else {
        if (k < K) v.x = p[0];
        if (k + 1 < K) v.y = p[1];
        if (k + 2 < K) v.z = p[2];
      }
    }
    r[i] = v;
  }
}

__device__ __forceinline__ void store_tile(_Float16* hi_img, _Float16* lo_img, const float4 (&r)[4], float s) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = threadIdx.x + kThreads * i;
    const int off = (idx >> 3) * PITCH + (idx & 7) * 4;
    half4 h, l;
    split4(r[i], s, h, l);
    *reinterpret_cast<half4*>(hi_img + off) = h;
    *reinterpret_cast<half4*>(lo_img + off) = l;
  }
}

__global__ __launch_bounds__(kThreads, 2) void gemm_nt_f16x3(Args a) {
  __shared__ __attribute__((aligned(16))) _Float16 lds[4 * TILE_HALVES];   // Ah | Al | Bh | Bl  (40 KB)
  _Float16* Ah = lds;
  _Float16* Al = lds + TILE_HALVES;
  _Float16* Bh = lds + 2 * TILE_HALVES;
  _Float16* Bl = lds + 3 * TILE_HALVES;

  // XCD-aware tile order: blocks that share an XCD (b % 8) walk consecutive tiles; tiles are numbered with
  // the n-block fastest, so the A row panel of a tile row stays in that XCD's L2 for all its column tiles.
  const unsigned nb = gridDim.x, b = blockIdx.x;
  const unsigned tile = (b & 7u) * (nb >> 3) + (b >> 3);
  if (tile >= (unsigned)(a.nbm * a.nbn)) return;
  const int bm = tile / a.nbn, bn = tile % a.nbn;
  const int row0 = bm * BM, col0 = bn * BN;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;                 // 2 x 2 waves, 64 x 64 each
  const int fr = lane & 31, fh = lane >> 5;                // fragment row / k-half

  const float sA = a.sA ? a.sA[0] : 1.f, sB = a.sB ? a.sB[0] : 1.f;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  float4 ra[4], rb[4];
  load_tile(a.A, a.lda, row0, a.M, 0, a.K, ra);
  load_tile(a.B, a.ldb, col0, a.N, 0, a.K, rb);
  // row index inside load_tile is absolute; make store offsets tile-relative by construction (row0 folded below)

  const int nk = (a.K + BK - 1) / BK;
  for (int kt = 0; kt < nk; ++kt) {
    __syncthreads();                                       // previous stage's fragment reads are done
    store_tile(Ah, Al, ra, sA);
    store_tile(Bh, Bl, rb, sB);
    __syncthreads();
    if (kt + 1 < nk) {                                     // next stage in flight while this one computes
      load_tile(a.A, a.lda, row0, a.M, (kt + 1) * BK, a.K, ra);
      load_tile(a.B, a.ldb, col0, a.N, (kt + 1) * BK, a.K, rb);
    }
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      half8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int off = (wm * 64 + i * 32 + fr) * PITCH + ks * 16 + fh * 8;
        ah[i] = *reinterpret_cast<const half8*>(Ah + off);
        al[i] = *reinterpret_cast<const half8*>(Al + off);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int off = (wn * 64 + j * 32 + fr) * PITCH + ks * 16 + fh * 8;
        bh[j] = *reinterpret_cast<const half8*>(Bh + off);
        bl[j] = *reinterpret_cast<const half8*>(Bl + off);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
    }
  }

  // epilogue: C/D layout of 32x32: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
  const float alpha = 1.f / (sA * sB);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = col0 + wn * 64 + j * 32 + fr;
      if (col >= a.N) continue;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = row0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
        if (row < a.M) a.C[(int64_t)row * a.ldc + col] = acc[i][j][e] * alpha;
      }
    }
}


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
constexpr int TPITCH = 160;                      // halves per LDS row: 128 + 32
constexpr int TTILE = TBK * TPITCH;              // one fp16 image

struct ArgsTN {
  const float* A; int64_t lda;                   // (R, M)
  const float* B; int64_t ldb;                   // (R, N)
  float* C; int64_t ldc; int64_t split_stride;   // partials: C + split * split_stride
  int64_t R; int M, N;
  int64_t rows_per_split;
  const float* sA; const float* sB;
  int nbm, nbn;
};

// one 32 x 128 fp32 tile = 1024 float4; thread t takes float4 #(t + 256 i): row = idx / 32, c4 = idx % 32
__device__ __forceinline__ void load_tile_t(const float* __restrict__ base, int64_t ld, int64_t r0, int64_t rend, int c0,
                                            int ncols, float4 (&r)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = threadIdx.x + kThreads * i;
    const int64_t row = r0 + (idx >> 5);
    const int c = c0 + (idx & 31) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < rend && c < ncols) {
      const float* p = base + row * ld + c;
      if (c + 3 < ncols) v = *reinterpret_cast<const float4*>(p);
      else { v.x = p[0]; if (c + 1 < ncols) v.y = p[1]; if (c + 2 < ncols) v.z = p[2]; }
    }
    r[i] = v;
  }
}

__device__ __forceinline__ void store_tile_t(_Float16* hi_img, _Float16* lo_img, const float4 (&r)[4], float s) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = threadIdx.x + kThreads * i;
    const int off = (idx >> 5) * TPITCH + (idx & 31) * 4;
    half4 h, l;
    split4(r[i], s, h, l);
    *reinterpret_cast<half4*>(hi_img + off) = h;
    *reinterpret_cast<half4*>(lo_img + off) = l;
  }
}

// fragment of the 32 (m) x 16 (k) operand block whose first column is m0 and first k-row is k0
__device__ __forceinline__ half8 tr_frag(const _Float16* img, int m0, int k0, int lane) {
  const int g = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
  const int col = m0 + (g & 1) * 16 + 4 * pp;
  const int krow = k0 + (g >> 1) * 8 + q;
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + krow * TPITCH + col));
  const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + (krow + 4) * TPITCH + col));
  union { s16x4 v[2]; half8 h; } u;
  u.v[0] = lo4; u.v[1] = hi4;
  return u.h;
}

__global__ __launch_bounds__(kThreads, 2) void gemm_tn_f16x3(ArgsTN a) {
  __shared__ __attribute__((aligned(16))) _Float16 lds[4 * TTILE];        // Ah | Al | Bh | Bl  (40 KB)
  _Float16* Ah = lds;
  _Float16* Al = lds + TTILE;
  _Float16* Bh = lds + 2 * TTILE;
  _Float16* Bl = lds + 3 * TTILE;

  const unsigned nb = gridDim.x, b = blockIdx.x;
  const unsigned tile = (b & 7u) * (nb >> 3) + (b >> 3);
  if (tile >= (unsigned)(a.nbm * a.nbn)) return;
  const int bm = tile / a.nbn, bn = tile % a.nbn;
  const int m0 = bm * BM, n0 = bn * BN;
  const int64_t r_beg = (int64_t)blockIdx.y * a.rows_per_split;
  const int64_t r_end = r_beg + a.rows_per_split < a.R ? r_beg + a.rows_per_split : a.R;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 31, fh = lane >> 5;
  const float sA = a.sA ? a.sA[0] : 1.f, sB = a.sB ? a.sB[0] : 1.f;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  float4 ra[4], rb[4];
  load_tile_t(a.A, a.lda, r_beg, r_end, m0, a.M, ra);
  load_tile_t(a.B, a.ldb, r_beg, r_end, n0, a.N, rb);
  for (int64_t r0 = r_beg; r0 < r_end; r0 += TBK) {
    __syncthreads();
    store_tile_t(Ah, Al, ra, sA);
    store_tile_t(Bh, Bl, rb, sB);
    __syncthreads();
    if (r0 + TBK < r_end) {
      load_tile_t(a.A, a.lda, r0 + TBK, r_end, m0, a.M, ra);
      load_tile_t(a.B, a.ldb, r0 + TBK, r_end, n0, a.N, rb);
    }
#pragma unroll
    for (int ks = 0; ks < TBK / 16; ++ks) {
      half8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        ah[i] = tr_frag(Ah, wm * 64 + i * 32, ks * 16, lane);
        al[i] = tr_frag(Al, wm * 64 + i * 32, ks * 16, lane);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        bh[j] = tr_frag(Bh, wn * 64 + j * 32, ks * 16, lane);
        bl[j] = tr_frag(Bl, wn * 64 + j * 32, ks * 16, lane);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
    }
  }

  const float alpha = 1.f / (sA * sB);
  float* Cp = a.C + (int64_t)blockIdx.y * a.split_stride;
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
__global__ void scale_from_partials(const float* __restrict__ partial, int n, float factor, float* __restrict__ scale) {
  float m = 0.f;
  for (int i = threadIdx.x; i < n; i += 64) m = fmaxf(m, partial[i]);
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  if (threadIdx.x == 0) {
    m *= factor;
    float s = 1.f;
    if (m > 0.f && m < INFINITY) { int e; frexpf(m, &e); s = ldexpf(1.f, 14 - e); }
    scale[0] = s;
  }
}
}  // namespace gemm

extern "C" {

int spgnn_gemm_nt(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int64_t M,
                  int64_t N, int64_t K, const float* scale_a, const float* scale_b, spgnn_stream_t stream) {
  if (M < 0 || N < 0 || K <= 0 || M > INT32_MAX || N > INT32_MAX || K > INT32_MAX) return SPGNN_ERR_SHAPE;
  if (M == 0 || N == 0) return SPGNN_OK;
  if (!A || !B || !C) return SPGNN_ERR_NULLPTR;
  if (lda < K || ldb < K || ldc < N || (lda & 3) || (ldb & 3) || (reinterpret_cast<uintptr_t>(A) & 15) ||
      (reinterpret_cast<uintptr_t>(B) & 15))
    return SPGNN_ERR_STRIDE;
  gemm::Args a{A, lda, B, ldb, C, ldc, (int)M, (int)N, (int)K, scale_a, scale_b,
               (int)((M + gemm::BM - 1) / gemm::BM), (int)((N + gemm::BN - 1) / gemm::BN)};
  int64_t tiles = (int64_t)a.nbm * a.nbn;
  tiles = (tiles + 7) & ~int64_t(7);
  hipLaunchKernelGGL(gemm::gemm_nt_f16x3, dim3((unsigned)tiles), dim3(gemm::kThreads), 0, (hipStream_t)stream, a);
  return hipGetLastError() == hipSuccess ? SPGNN_OK : -1000;
}

int spgnn_gemm_tn(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int64_t split_stride,
                  int32_t splits, int64_t R, int64_t M, int64_t N, const float* scale_a, const float* scale_b,
                  spgnn_stream_t stream) {
  if (R < 0 || M <= 0 || N <= 0 || splits <= 0 || M > INT32_MAX || N > INT32_MAX) return SPGNN_ERR_SHAPE;
  if (!A || !B || !C) return SPGNN_ERR_NULLPTR;
  if (lda < M || ldb < N || ldc < N || (lda & 3) || (ldb & 3) || (reinterpret_cast<uintptr_t>(A) & 15) ||
      (reinterpret_cast<uintptr_t>(B) & 15) || (splits > 1 && split_stride < M * ldc))
    return SPGNN_ERR_STRIDE;
  int64_t rps = (R + splits - 1) / splits;
  rps = (rps + gemm::TBK - 1) / gemm::TBK * gemm::TBK;
  if (rps == 0) rps = gemm::TBK;
  gemm::ArgsTN a{A, lda, B, ldb, C, ldc, split_stride, R, (int)M, (int)N, rps, scale_a, scale_b,
                 (int)((M + gemm::BM - 1) / gemm::BM), (int)((N + gemm::BN - 1) / gemm::BN)};
  int64_t tiles = (int64_t)a.nbm * a.nbn;
  tiles = (tiles + 7) & ~int64_t(7);
  hipLaunchKernelGGL(gemm::gemm_tn_f16x3, dim3((unsigned)tiles, (unsigned)splits), dim3(gemm::kThreads), 0,
                     (hipStream_t)stream, a);
  return hipGetLastError() == hipSuccess ? SPGNN_OK : -1000;
}

int spgnn_pow2_scale(const float* x, int64_t x_stride, int64_t rows, int64_t cols, float* scale, float* workspace,
                     int32_t workspace_floats, spgnn_stream_t stream) {
  if (rows < 0 || cols <= 0 || cols > INT32_MAX || workspace_floats < 1) return SPGNN_ERR_SHAPE;
  if (!scale || !workspace || (rows > 0 && !x)) return SPGNN_ERR_NULLPTR;
  if (rows > 0 && ((x_stride & 3) || (reinterpret_cast<uintptr_t>(x) & 15))) return SPGNN_ERR_STRIDE;
  hipStream_t st = (hipStream_t)stream;
  int64_t blocks = (rows * ((cols + 3) / 4) + 255) / 256;
  if (blocks > workspace_floats) blocks = workspace_floats;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(gemm::absmax_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, x_stride, rows, (int)cols, workspace);
  hipLaunchKernelGGL(gemm::scale_from_partials, dim3(1), dim3(64), 0, st, workspace, (int)blocks, 1.f, scale);
  return hipGetLastError() == hipSuccess ? SPGNN_OK : -1000;
}

}  // extern "C"
