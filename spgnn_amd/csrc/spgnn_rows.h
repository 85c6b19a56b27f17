// Device helpers shared by the row kernels of libspgnn_hip.so (spgnn_kernels.hip, spgnn_lspe.hip): team geometry, 16-byte row
// access for fp32 / bf16 storage, activations, the counter-hash dropout masks and the cross-lane reductions.  Every including
// translation unit must be compiled with -fno-slp-vectorize -fno-vectorize -DSPGNN_NO_SLP_VECTORIZE (csrc/build.py): see the
// note on packed fp32 ops below.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

#include "spgnn_hip.h"
#include "spgnn_internal.h"

// With hipcc's vectorizers on (ROCm 7.2, gfx950) per-edge dots next to cross-lane reads are computed by packed fp32 ops
// (v_pk_fma_f32 / v_pk_mul_f32 fed by v_pk_mov_b32 op_sel shuffles), and when a second process shares the GPU a few launches
// per hundred produced a wrong dot for ALL 16 lanes of one team - one 16-lane pass of one instruction - with every operand in
// memory and in registers verified correct.  Without the vectorizers: 0 of 400 repetitions in either process, and the row
// kernels run as fast.  The mechanism is NOT confirmed; csrc/build.py fails the build if a packed fp32 op shows up in these
// objects and tests/test_two_process.py repeats the stress inside the GPU suite.
#if !defined(SPGNN_NO_SLP_VECTORIZE)
#error "the row kernels must be compiled with -fno-slp-vectorize -fno-vectorize -DSPGNN_NO_SLP_VECTORIZE (csrc/build.py)"
#endif

namespace {

constexpr int kBlock = 256;
static_assert(kBlock % 64 == 0 && kBlock >= 64 && kBlock <= 1024, "whole waves per block; every grid below is derived from kBlock");

// ---- team geometry -----------------------------------------------------------------------------
// H*D floats per node = 4 * T * R.  T = lanes per node, R = float4 chunks per lane.
// `narrow`: rows of <= 64 float4 go to 16-lane teams (four nodes per wave, R = q/16 chunks per lane).  Such rows make
// the kernels latency-bound - a wave walks one dependent index -> score -> row chain per node - and four chains per
// wave hide more of it than the SGPR savings of a whole-wave team are worth: measured for single-head layers
// (1x256: fwd 51 -> 42, bwd 68 -> 59 / 35 -> 27 us; 1x128: 33 -> 25, 39 -> 30, 20 -> 17 us), not for two-head ones.
bool pick_team(int64_t width, int& T, int& R, bool narrow = false) {
  if (width <= 0 || (width & 3)) return false;
  int64_t q = width >> 2;
  if (narrow && q <= 64 && q % 16 == 0) {
    const int64_t r = q / 16;
    if (r == 1 || r == 2 || r == 4) { T = 16; R = (int)r; return true; }
  }
  const int ts[3] = {64, 32, 16};
  for (int t : ts) {
    if (q % t) continue;
    int64_t r = q / t;
    if (r == 1 || r == 2 || r == 4 || r == 8) { T = t; R = (int)r; return true; }
  }
  return false;
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
bool vec_ok(const void* p, int64_t stride) { return p == nullptr || (aligned16(p) && (stride & 3) == 0); }

inline unsigned grid_for(int64_t n_nodes, int nodes_per_block) {
  int64_t nb = (n_nodes + nodes_per_block - 1) / nodes_per_block;
  nb = (nb + 7) & ~int64_t(7);            // multiple of 8 so the XCD remap is a bijection
  return (unsigned)nb;
}

// Blocks b and b+8 share an XCD (round-robin dispatch, MI355X_MICROARCH.md "Workgroup dispatch").
// Give each XCD a contiguous run of node blocks.  Pure speed: any placement is correct.
__device__ __forceinline__ int64_t xcd_block(void) {
  const unsigned nb = gridDim.x, b = blockIdx.x;
  return (int64_t)(b & 7u) * (nb >> 3) + (b >> 3);
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
// Row storage types.  Every row kernel computes in fp32; `ST` is what the node-feature rows are STORED as in HBM:
// float, or bf16s = bfloat16 (BASELINE config "st_gat_6 ... bf16": bf16 storage, fp32 accumulate).  A lane's chunk is four
// consecutive elements either way (16-byte or 8-byte vector access), so the team geometry is the same for both.
struct bf16s { uint16_t bits; };
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ldv(const float* p) { return ld4(p); }
__device__ __forceinline__ float4 ldv(const bf16s* p) {
  const uint2 u = *reinterpret_cast<const uint2*>(p);
  return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xFFFF0000u), __uint_as_float(u.y << 16),
                     __uint_as_float(u.y & 0xFFFF0000u));
}
__device__ __forceinline__ void stv(float* p, float4 v) { st4(p, v); }
__device__ __forceinline__ void stv(bf16s* p, float4 v) {       // round to nearest even (v_cvt_pk_bf16_f32)
  const f32x4_t f = {v.x, v.y, v.z, v.w};
  union { bf16x4_t h; uint2 u; } q;
  q.h = __builtin_convertvector(f, bf16x4_t);
  *reinterpret_cast<uint2*>(p) = q.u;
}
template <typename ST> struct is_f32 { static constexpr bool value = false; };
template <> struct is_f32<float> { static constexpr bool value = true; };
__device__ __forceinline__ float dot4(float4 a, float4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
__device__ __forceinline__ void fma4(float4& acc, float s, float4 x) {
  acc.x = fmaf(s, x.x, acc.x); acc.y = fmaf(s, x.y, acc.y); acc.z = fmaf(s, x.z, acc.z); acc.w = fmaf(s, x.w, acc.w);
}

__device__ __forceinline__ float lrelu(float x, float slope) { return x > 0.f ? x : x * slope; }

// exp(x) for x <= -0.5 as ONE v_exp_f32 (2^(x log2 e)): __expf expands to a guarded sequence that hipcc wraps in
// exec-mask branches per element (the same ELU in the GEMM epilogue: ~17 000 instructions of them).  Used for ELU
// only (result exp(x) - 1 in (-1, -0.39]: the exponent's rounding stays below one ulp of it).  The softmax keeps expf:
// the one-instruction form there was neutral for the step (6.78 ms either way) and cost 3e-7 of logits parity.
__device__ __forceinline__ float exp_nb(float x) { return __builtin_amdgcn_exp2f(fmaxf(x * 1.44269504088896341f, -127.f)); }

// expm1 for x <= 0, branch-free and ~4x cheaper than libm's expm1f (the ELU epilogues are VALU-heavy: 120 us of a
// 400 us GEMM went into it): Taylor polynomial of degree 9 on [-0.5, 0] (truncation 5e-9), exp(x) - 1 below
// (result in (-1, -0.39]: 2e-7 relative).  Within 2 ulp of expm1f on x <= 0.
__device__ __forceinline__ float expm1_neg(float x) {
  const float p = x * (1.f + x * (0.5f + x * (1.f / 6 + x * (1.f / 24 + x * (1.f / 120 + x * (1.f / 720 + x * (1.f / 5040 +
                  x * (1.f / 40320 + x * (1.f / 362880)))))))));
  const float e = exp_nb(x) - 1.f;
  return x > -0.5f ? p : e;
}
__device__ __forceinline__ float elu_fwd(float x) { return x > 0.f ? x : expm1_neg(x); }

__device__ __forceinline__ float act_fwd(float x, int act) {
  switch (act) {
    case SPGNN_ACT_ELU:  return elu_fwd(x);
    case SPGNN_ACT_TANH: return tanhf(x);
    case SPGNN_ACT_RELU: return x > 0.f ? x : 0.f;
    case SPGNN_ACT_LRELU: return x > 0.f ? x : 0.01f * x;
    default:             return x;
  }
}
// whole register rows at once: one wave-uniform switch, then straight-line element code
template <int R> __device__ __forceinline__ void act_fwd_rows(float4 (&o)[R], int act) {
#define SPGNN_ROWS(EXPR) _Pragma("unroll") for (int r = 0; r < R; ++r) { \
    { float x = o[r].x; o[r].x = (EXPR); } { float x = o[r].y; o[r].y = (EXPR); } \
    { float x = o[r].z; o[r].z = (EXPR); } { float x = o[r].w; o[r].w = (EXPR); } }
  if (act == SPGNN_ACT_ELU) { SPGNN_ROWS(elu_fwd(x)) }
  else if (act == SPGNN_ACT_TANH) { SPGNN_ROWS(tanhf(x)) }
  else if (act == SPGNN_ACT_RELU) { SPGNN_ROWS(x > 0.f ? x : 0.f) }
  else if (act == SPGNN_ACT_LRELU) { SPGNN_ROWS(x > 0.f ? x : 0.01f * x) }
#undef SPGNN_ROWS
}
// derivative expressed through the OUTPUT y = act(x) (what the forward saved)
__device__ __forceinline__ float act_bwd_from_out(float y, int act) {
  switch (act) {
    case SPGNN_ACT_ELU:  return y > 0.f ? 1.f : y + 1.f;
    case SPGNN_ACT_TANH: return 1.f - y * y;
    case SPGNN_ACT_RELU: return y > 0.f ? 1.f : 0.f;
    case SPGNN_ACT_LRELU: return y > 0.f ? 1.f : 0.01f;
    default:             return 1.f;
  }
}

template <int R> __device__ __forceinline__ void act_bwd_rows(float4 (&g)[R], const float4 (&o)[R], int act) {
#define SPGNN_ROWS(EXPR) _Pragma("unroll") for (int r = 0; r < R; ++r) { \
    { float y = o[r].x; g[r].x *= (EXPR); } { float y = o[r].y; g[r].y *= (EXPR); } \
    { float y = o[r].z; g[r].z *= (EXPR); } { float y = o[r].w; g[r].w *= (EXPR); } }
  if (act == SPGNN_ACT_ELU) { SPGNN_ROWS(y > 0.f ? 1.f : y + 1.f) }
  else if (act == SPGNN_ACT_TANH) { SPGNN_ROWS(1.f - y * y) }
  else if (act == SPGNN_ACT_RELU) { SPGNN_ROWS(y > 0.f ? 1.f : 0.f) }
  else if (act == SPGNN_ACT_LRELU) { SPGNN_ROWS(y > 0.f ? 1.f : 0.01f) }
#undef SPGNN_ROWS
}

// Counter-based keep mask for attention dropout: one 64-bit mix of (seed, slot*H + h).  The
// backward kernels regenerate it instead of storing E*H bytes.
__device__ __forceinline__ float keep_scale(uint64_t seed, int64_t idx, float p, float inv_keep) {
  uint64_t z = seed + 0x9E3779B97F4A7C15ull * (uint64_t)(idx + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  const float u = (float)(uint32_t)(z >> 40) * (1.0f / 16777216.0f);   // 24 bits -> [0,1)
  return u >= p ? inv_keep : 0.f;
}

using spgnn_detail::mix64;          // the counter hash and the four-element feature-dropout mask live in spgnn_internal.h
using spgnn_detail::feat_keep4;     // (the GEMM epilogue applies the same mask)

// wave-uniform value -> SGPR (no-op when the template flag is off)
template <bool ON> __device__ __forceinline__ int uni(int x) { return ON ? __builtin_amdgcn_readfirstlane(x) : x; }
template <bool ON> __device__ __forceinline__ float uni(float x) {
  return ON ? __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x))) : x;
}
// Cross-lane reductions.  `single_pass` re-writes the value with a plain v_mov first: hipcc pairs neighbouring fp32
// chains into packed ops (v_pk_fma_f32 / v_pk_add_f32, two passes), and a ds_bpermute / DPP read of a register such an
// op has just written was observed to see lanes 48-63 - written in the last pass - too early (a few wrong sums per
// million, different every run; ROCm 7.2, gfx950; found in the GEMM's score-partial epilogue).  A VALU -> VALU
// dependency is fully interlocked, so the one extra move makes the cross-lane read safe.
__device__ __forceinline__ float single_pass(float x) {
  asm volatile("v_mov_b32 %0, %0" : "+v"(x));
  return x;
}
// Row-local lane permutations as DPP modifiers (VALU: no trip through the LDS crossbar that ds_bpermute takes): quad_perm
// [1,0,3,2] = xor 1, [2,3,0,1] = xor 2, row_half_mirror = i <-> 7 - i inside 8 lanes, row_mirror = i <-> 15 - i.  Applied in
// this order they fold pairs, quads, groups of 8 and the 16-lane row: after each step the partial result is uniform in the
// group it covers (fp addition and max are commutative), so the mirrors combine two uniform halves exactly as xor 4 / xor 8 do.
template <int CTRL> __device__ __forceinline__ float dpp_row(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, true));
}
// sum / max over aligned groups of 8 lanes, result in every lane (bit-identical to the xor 1, 2, 4 butterfly)
__device__ __forceinline__ float group8_sum(float x) {
  x = single_pass(x);
  x = single_pass(x + dpp_row<0xB1>(x));
  x = single_pass(x + dpp_row<0x4E>(x));
  x = single_pass(x + dpp_row<0x141>(x));
  return x;
}
__device__ __forceinline__ float group8_max(float x) {
  x = single_pass(x);
  x = single_pass(fmaxf(x, dpp_row<0xB1>(x)));
  x = single_pass(fmaxf(x, dpp_row<0x4E>(x)));
  x = single_pass(fmaxf(x, dpp_row<0x141>(x)));
  return x;
}
template <int W> __device__ __forceinline__ float team_sum_fixed(float x) {
  x = single_pass(x);
#pragma unroll
  for (int off = W >> 1; off > 0; off >>= 1) x = single_pass(x + __shfl_xor(x, off, 64));
  return x;
}
__device__ __forceinline__ float team_max(float x, int width) {
  x = single_pass(x);
  if (width == 16) {
    x = single_pass(fmaxf(x, dpp_row<0xB1>(x)));
    x = single_pass(fmaxf(x, dpp_row<0x4E>(x)));
    x = single_pass(fmaxf(x, dpp_row<0x141>(x)));
    return single_pass(fmaxf(x, dpp_row<0x140>(x)));
  }
  for (int off = width >> 1; off > 0; off >>= 1) x = single_pass(fmaxf(x, __shfl_xor(x, off, 64)));
  return x;
}
__device__ __forceinline__ float absmax4(float m, float4 v) {
  return fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
}
__device__ __forceinline__ float team_sum(float x, int width) {
  x = single_pass(x);
  if (width == 16) {
    x = single_pass(x + dpp_row<0xB1>(x));
    x = single_pass(x + dpp_row<0x4E>(x));
    x = single_pass(x + dpp_row<0x141>(x));
    return single_pass(x + dpp_row<0x140>(x));
  }
  for (int off = width >> 1; off > 0; off >>= 1) x = single_pass(x + __shfl_xor(x, off, 64));
  return x;
}

// keep / send of one reduce-scatter exchange between lanes l and l ^ half (`up` = this lane's bit `half`; lo / hi = the two
// table entries that differ in that bit).  The two values are made opaque first: left visible, hipcc rewrites
// `up ? pd[i + half] : pd[i]` as pd[i + (lane & half)] - a DYNAMIC index into the register array - and lowers every such
// access to a compare-and-select chain over all entries (450 extra VALU instructions for a 16-entry table, 1,900 for 32:
// the dst-major kernels spent more time there than on their rows).
__device__ __forceinline__ void rs_pair(bool up, float lo, float hi, float& keep, float& send) {
  asm volatile("" : "+v"(lo), "+v"(hi));
  keep = up ? hi : lo;
  send = single_pass(up ? lo : hi);
}

// Nodes with at most this many edges (every airway node: in-degree <= 5) take a path that loads all edge indices and scores
// up front (independent loads) and keeps per-edge weights in registers.
constexpr int kMaxFast = 8;

}  // namespace
