// Internal helpers shared by the translation units of libspgnn_hip.so (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace spgnn_detail {
// record the message spgnn_last_error() returns (thread-local) and hand `code` back
int fail(int code, const char* msg);
// hipGetLastError() after a launch -> SPGNN_OK or -(1000 + hipError_t) with the runtime's message recorded
int check_launch(const char* what);
// argument check failed inside `func` (source line `line`): records "<func>: <kind of error> (line N)"
int fail_at(int code, const char* func, int line);
// hipFuncAttributeMaxDynamicSharedMemorySize for kernels that need more than 64 KB of LDS; the attribute is per
// device, so it is remembered per (function, device)
int ensure_dynamic_lds(const void* func, int bytes);
}  // namespace spgnn_detail

// ---- scale blocks ----------------------------------------------------------------------------------------------------
// The power-of-two operand scales of the split GEMMs (spgnn_gemm.hip) travel as device pointers to a SCALE BLOCK:
//   {s}                          one positive float: the scale itself, or
//   {-n, 0, 0, 0, m_1 ... m_n}   n = 256 maxima ("slots"): the operand's largest magnitude is max_i m_i and its scale
//                                2^(14 - e) with max <= 2^e is derived by the consuming kernel.
// Producers of an activation tensor fold the maxima of their rows into the slots with one result-free atomicMax per node
// team (non-negative floats order like their bit patterns; max is order-independent, so the scale is deterministic), which
// replaces a reduction launch per operand.  The slots must be zero when the first producer runs (ops.ScalePool).
namespace spgnn_detail {
constexpr int kScaleSlots = 256, kScaleHeader = 4;
__device__ __forceinline__ float pow2_scale_of(float m) {
  float s = 1.f;
  if (m > 0.f && m < INFINITY) { int e; frexpf(m, &e); s = ldexpf(1.f, 14 - e); }
  return s;
}
// call with all 64 lanes of the wave active (top of a kernel)
__device__ __forceinline__ float load_scale(const float* __restrict__ p) {
  if (!p) return 1.f;
  const float h = p[0];
  if (h > 0.f) return h;
  const int lane = threadIdx.x & 63;
  float m = 0.f;
#pragma unroll
  for (int i = 0; i < kScaleSlots / 64; ++i) m = fmaxf(m, p[kScaleHeader + lane + 64 * i]);
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  return pow2_scale_of(m);
}
// load_scale for the operand of a split GEMM, with the RANGE MONITOR (round 4): hi + lo keeps 22 bits only for values within
// 2^18 of the tensor's maximum (below that lo falls into fp16's subnormals: DESIGN.md section 4.2's envelope).  The 256 slots
// are maxima over disjoint pieces of the operand (whole rows - slot = node index mod 256 - for the row kernels, a wave's
// 32 x 64 piece of a tile for a product's epilogue), so a NON-ZERO slot more than 2^18 below the largest one means whole rows
// or blocks of the operand sit outside the envelope - a row-structured heavy tail, e.g. a few exploding nodes - and their
// results lose relative accuracy.  The consuming kernel then sets header word 1 of the block (every workgroup stores
// the same value: no atomics); spgnn_step_begin adds the set flags of a step's pool to a device counter before it re-arms
// the blocks.  Detection only: the arithmetic is unchanged (recovery: ops.GEMM_MODE = "fp32").  {s} blocks carry no slots.
constexpr float kRangeEnvelope = 262144.f;             // 2^18
__device__ __forceinline__ float load_scale_monitored(const float* __restrict__ p) {
  if (!p) return 1.f;
  const float h = p[0];
  if (h > 0.f) return h;
  const int lane = threadIdx.x & 63;
  float m = 0.f, lo = INFINITY;
#pragma unroll
  for (int i = 0; i < kScaleSlots / 64; ++i) {
    const float v = p[kScaleHeader + lane + 64 * i];
    m = fmaxf(m, v);
    lo = fminf(lo, v > 0.f ? v : INFINITY);
  }
  for (int off = 32; off > 0; off >>= 1) { m = fmaxf(m, __shfl_xor(m, off, 64)); lo = fminf(lo, __shfl_xor(lo, off, 64)); }
  if (lo * kRangeEnvelope < m && lane == 0) const_cast<float*>(p)[1] = 1.f;
  return pow2_scale_of(m);
}
// one lane per team / block: fold m >= 0 into slot idx of a scale block
__device__ __forceinline__ void slots_max(float* block, float m, unsigned idx) {
  unsigned* w = reinterpret_cast<unsigned*>(block + kScaleHeader) + (idx & (kScaleSlots - 1));
  // unconditional and result-free: a fire-and-forget atomic costs the wave ~5 us less than first reading the slot to skip it
  // (the read is a dependent memory round trip at the very end of the wave's life; measured on the three lspe kernels)
  atomicMax(w, __float_as_uint(m));
}
// Counter hash of the dropout masks (splitmix64 finaliser over seed + golden * (idx + 1)).
__device__ __forceinline__ uint64_t mix64(uint64_t seed, int64_t idx) {
  uint64_t z = seed + 0x9E3779B97F4A7C15ull * (uint64_t)(idx + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
// Feature-dropout keep factors of four neighbouring elements (spgnn_cat_dropout's mask: one 64-bit hash per group of four
// columns, 16 bits per element, counter = row * total_width + column of the group's first element).
__device__ __forceinline__ float4 feat_keep4(uint64_t seed, int64_t counter, float p, float inv_keep) {
  const uint64_t z = mix64(seed, counter);
  const unsigned thr = (unsigned)(p * 65536.f);
  return make_float4(((unsigned)(z) & 0xFFFFu) >= thr ? inv_keep : 0.f, ((unsigned)(z >> 16) & 0xFFFFu) >= thr ? inv_keep : 0.f,
                     ((unsigned)(z >> 32) & 0xFFFFu) >= thr ? inv_keep : 0.f, ((unsigned)(z >> 48) & 0xFFFFu) >= thr ? inv_keep : 0.f);
}
}  // namespace spgnn_detail
