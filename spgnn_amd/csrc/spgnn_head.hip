// spgnn_head.hip - the tail of a training step behind the output layer (ABI 61): classifier + masked cross entropy + the
// classifier's weight gradient in one pass (spgnn_classifier_ce).  A file of its own because it is built WITH the vectorizers:
// its weight-gradient loop runs on packed fp32 FMAs (v_pk_fma_f32, two per lane and issue slot), which the row-kernel files
// exclude wholesale (csrc/build.py: no packed fp32 next to their cross-lane reads).  Here the only cross-lane reads are the
// 16-lane DPP reductions of the softmax, fed by single-pass scalar ops; build.py checks per instruction that none is fed by a
// packed op.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

#include "spgnn_hip.h"
#include "spgnn_internal.h"

namespace {
using spgnn_detail::check_launch;
using spgnn_detail::fail;
using spgnn_detail::mix64;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
// Cross-lane reductions as in spgnn_rows.h: the value is re-written by a plain v_mov before every cross-lane read (a DPP read of
// a register a packed two-pass op has just written was observed to see the last lanes too early; a VALU -> VALU dependency
// is interlocked).  Row-local DPP: quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror fold 2, 4, 8, 16 lanes.
__device__ __forceinline__ float single_pass(float x) { asm volatile("v_mov_b32 %0, %0" : "+v"(x)); return x; }
template <int CTRL> __device__ __forceinline__ float dpp_row(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float row16_sum(float x) {
  x = single_pass(x);
  x = single_pass(x + dpp_row<0xB1>(x)); x = single_pass(x + dpp_row<0x4E>(x)); x = single_pass(x + dpp_row<0x141>(x));
  return single_pass(x + dpp_row<0x140>(x));
}
__device__ __forceinline__ float row16_max(float x) {
  x = single_pass(x);
  x = single_pass(fmaxf(x, dpp_row<0xB1>(x))); x = single_pass(fmaxf(x, dpp_row<0x4E>(x))); x = single_pass(fmaxf(x, dpp_row<0x141>(x)));
  return single_pass(fmaxf(x, dpp_row<0x140>(x)));
}
__device__ __forceinline__ float wave_sum(float x) {
  x = single_pass(x);
  for (int off = 32; off > 0; off >>= 1) x = single_pass(x + __shfl_xor(x, off, 64));
  return x;
}
// acc (4 columns as two packed pairs) += g * x: two v_pk_fma_f32, the scalar broadcast into both halves
struct Acc4 { f32x2 lo, hi; };
__device__ __forceinline__ void pk_fma4(Acc4& acc, float g, f32x2 xlo, f32x2 xhi) {
  const f32x2 gg = {g, g};
  acc.lo = __builtin_elementwise_fma(gg, xlo, acc.lo);
  acc.hi = __builtin_elementwise_fma(gg, xhi, acc.hi);
}

// =================================================================================================
// Skinny classifier + masked cross entropy + the classifier's weight gradient from ONE read of the rows (ABI 61).
// The tail of a training step behind the output layer (reference models.py:1125, 1167-1170: n_out = gnn_out(n_embed);
// job_runner.py:1896-1900: loss = F.cross_entropy(pre[mask], y[mask], weight=w)):
//   logits[n, :] = x[n, :] W^T + b                                    (was spgnn_scores_fwd: one pass over x)
//   m_n, nll_n, g_logits[n, :] = m_n w[y_n] (softmax(logits[n]) - e_y)   (was spgnn_masked_ce_step: a launch of its own)
//   gW = g_logits^T x, g_b = colsum(g_logits)                           (was spgnn_scores_bwd_w: a second pass over x)
// g_logits is the gradient of the loss NUMERATOR and depends on the row alone (the step divides by the global weight sum in
// its optimizer kernel), so everything is row-local and the weight gradient can be accumulated while the rows are on the CU.
// A workgroup (512 threads, one per CU) owns a row range and walks it in chunks of 16 rows with two roles side by side - see
// the kernel.  The block's J x K partial goes out once, at the end (spgnn_sum_partials_multi adds the blocks in order); loss
// sums and bias gradient as in masked_ce_kernel (last workgroup, block order).  Deterministic: no atomics on data.
// =================================================================================================
struct bf16s { uint16_t bits; };             // bfloat16 storage of the rows (BASELINE config 4); every product is formed in fp32
__device__ __forceinline__ float4 ldrow(const float* p) { return ld4(p); }
__device__ __forceinline__ float4 ldrow(const bf16s* p) {       // four bf16 -> fp32, exact
  const uint2 u = *reinterpret_cast<const uint2*>(p);
  return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xFFFF0000u), __uint_as_float(u.y << 16),
                     __uint_as_float(u.y & 0xFFFF0000u));
}

struct ClsCe {
  const void* x; int64_t ldx; const float* w; int Kp; const float* bias; const int64_t* labels; const float* draws;
  uint64_t draw_seed; const int64_t* seed_off; const float* sampling_p; const float* class_w; const int32_t* flag;
  float* logits; int64_t ldl; float* g_logits; int64_t ldg; float* wpart; float* partial; float* sums; unsigned* ticket;
  float* colpart; float* colsum; int64_t N; int K; int J; int64_t rps;
};

constexpr int kCeThreads = 512;             // waves 0-3: logits on the matrix pipe + softmax / loss; waves 4-7: the rows' loads and the weight gradient

// Workgroup barrier that orders LDS traffic only.  __syncthreads() is a workgroup-scope fence + s_barrier, and the fence makes
// hipcc wait for EVERY outstanding memory operation (s_waitcnt vmcnt(0)): the global loads issued a step ahead would be waited
// for at the very next barrier and the prefetch would hide nothing (measured: 245 -> see profiles/r06_classifier_ce.md).
// Here only LDS operations (lgkmcnt) are drained; global loads stay in flight and are waited for where their registers are
// first read; global STORES of one role are never read by the other inside the kernel.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ST: storage type of the rows of x (float, or bf16s: two-byte rows, fp32 arithmetic).  RH (1, 2, 4): row groups of the C role -
// with K <= 512 only K / 4 of the 256 C-threads have columns, so RH groups of them share a chunk's 16 rows (16 / RH rows each,
// own accumulators, own partial slice): the folded classifier of the GAT heads reads 384-wide rows.
template <typename ST, int NG, int JA, int RH>
__global__ __launch_bounds__(kCeThreads, 1) void classifier_ce_kernel(ClsCe a) {
  constexpr int RPT = 16 / RH;               // rows of a chunk per C-thread
  // One workgroup per CU, two ROLES that work on neighbouring 16-row chunks at the same time (each SIMD hosts one wave of
  // either role, so the matrix pipe and the vector ALUs run side by side):
  //   A-waves (0-3): keep their W fragments in registers for the whole kernel (wave w: k range [w K/4, (w+1) K/4), 16 x J logits of
  //            chunk c on v_mfma_f32_16x16x4_f32 with the rows read from LDS), then 16 lanes per row: softmax, loss terms,
  //            logit gradient -> gl (LDS) and global;
  //   C-waves (4-7): thread t owns columns 4t .. 4t+3: loads the rows of chunk c + 2 (in flight under a whole step), writes chunk
  //            c + 1 to the staging buffer, and adds g_logits[c - 1]^T x[c - 1] into its persistent J x 4 accumulators
  //            (RH > 1: thread (column group, row group h) does that for rows [h RPT, (h + 1) RPT) of every chunk).
  // Step c:   H1  A: logits(c) from S[c % 2]            | C: weight gradient of chunk c - 1 (S[(c-1) % 2], gl[(c-1) % 2])
  //           barrier
  //           H2  A: softmax / loss (c) -> gl[c % 2]    | C: staged rows of chunk c + 1 -> S[(c+1) % 2]; issue the loads of c + 2
  //           barrier
  constexpr int kPitch = 1024 + 4;           // floats per staged row: 16 consecutive rows start 4 banks apart
  __shared__ float stage[2][16 * kPitch];
  __shared__ float part[4][16][32];          // partial logits per A-wave
  __shared__ float gl[2][16][32];            // logit gradients of a chunk, zero where (row, j) does not exist
  __shared__ float cwl[32];
  __shared__ float red[2][4];
  __shared__ bool last;
  float (*csred)[32] = part[0];              // (16, 32) scratch of the epilogue: `part` is free by then
  const int tid = threadIdx.x;
  const bool is_a = tid < 256;
  const int lane = tid & 63, wv = (tid >> 6) & 3;
  const int r = lane & 15, q = lane >> 4;
  const int64_t n0 = (int64_t)blockIdx.x * a.rps;
  const int64_t n1 = n0 + a.rps < a.N ? n0 + a.rps : a.N;
  const int nc = (int)((n1 - n0 + 15) >> 4);
  const int K = a.K, J = a.J;
  if (tid < 32) cwl[tid] = tid < J ? a.class_w[tid] : 0.f;
  // ---- A role state -------------------------------------------------------------------------------------------------
  const int kq = ((K / 16 + 3) / 4) * 16;    // K % 128 == 0 (the host checks): every wave's range is an even number of 16-steps
  const int kbeg = wv * kq < K ? wv * kq : K;
  const int kend = kbeg + kq < K ? kbeg + kq : K;
  const int nsteps = (kend - kbeg) >> 4;     // wave-uniform, <= 16
  // ONE register array for both roles (a wave has one role for good; declared separately the compiler keeps both sets live
  // and spills): A: st[16 g + s] = this lane's B fragment W[r + 16 g][kbeg + 16 s + 4 q ..], zero past the range / past J;
  // C: st[j] = the J x 4 accumulators, st[JA + rr] = the 16 rows in flight
  constexpr int kSt = (JA + RPT > 16 * NG) ? JA + RPT : 16 * NG;
  union Regs { float4 st[kSt]; Acc4 acc[JA]; __device__ Regs() {} } u_;
  float4 (&st)[kSt] = u_.st;                 // A: fragments; C: st[JA + rr] = rows in flight
  Acc4 (&acc)[JA] = u_.acc;                  // C: the accumulators, as packed pairs (same registers as st[0 .. JA))
  const int j0 = r, j1 = r + 16;
  const bool v0ok = j0 < J, v1ok = NG > 1 && j1 < J;
  float b0 = 0.f, b1 = 0.f, num = 0.f, den = 0.f, cs0 = 0.f, cs1 = 0.f;
  const bool poisoned = a.flag && a.flag[1] != 0;
  int64_t yl_n = 0; float rn_n = 1.f, sp_n = 0.f;          // label / draw / sampling probability of this lane's row in the NEXT chunk
  const uint64_t sd = a.draw_seed + (a.seed_off ? 0xD1B54A32D192ED03ull * (uint64_t)a.seed_off[0] : 0ull);
  auto fetch_row = [&](int c) {              // A: the per-row scalars of chunk c, a step ahead of their use
    const int64_t i = n0 + 16 * (int64_t)c + 4 * wv + q;
    const int64_t ii = i < n1 ? i : n1 - 1;
    yl_n = a.labels[ii];
    sp_n = a.sampling_p[ii];
    rn_n = a.draws ? a.draws[ii] : (float)(uint32_t)(mix64(sd, ii) >> 40) * (1.0f / 16777216.0f);
  };
  // ---- C role state -------------------------------------------------------------------------------------------------
  const int ct = tid - 256;
  const int cthreads = K >> 2;               // C-threads per row group
  const int half = RH > 1 ? (is_a ? 0 : ct / cthreads) : 0;
  const int kc = 4 * (RH > 1 ? ct - half * cthreads : ct);
  const int r0 = half * RPT;                 // this thread's rows of a chunk: r0 .. r0 + RPT - 1
  const bool cols = !is_a && (RH > 1 ? half < RH : kc < K);
  auto issue = [&](int c) {                  // C: rows of chunk c, columns 4 ct ..: 16 loads in flight, straight-line, unconditional
    const int64_t c0 = n0 + 16 * (int64_t)c;
#pragma unroll
    for (int rr = 0; rr < RPT; ++rr) {
      const int64_t row = c0 + r0 + rr < n1 ? c0 + r0 + rr : n1 - 1;   // rows past the range re-read the last one (their gradient rows are zero)
      st[JA + rr] = ldrow(reinterpret_cast<const ST*>(a.x) + row * a.ldx + kc);
    }
  };
  auto put = [&](int c) {                    // C: the loaded rows -> S[c % 2]
    float* sp = stage[c & 1] + r0 * kPitch + kc;
#pragma unroll
    for (int rr = 0; rr < RPT; ++rr) *reinterpret_cast<float4*>(sp + rr * kPitch) = st[JA + rr];
  };
  // ---- prologue -----------------------------------------------------------------------------------------------------
  if (is_a) {
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const bool jv = r + 16 * g < J;
      const float* wp = a.w + (int64_t)(jv ? r + 16 * g : 0) * a.Kp + kbeg + 4 * q;
#pragma unroll
      for (int s_ = 0; s_ < 16; ++s_) {
        const bool live = jv && s_ < nsteps;
        const float4 t_ = ld4(wp + 16 * (s_ < nsteps ? s_ : 0));
        const float z_ = live ? 1.f : 0.f;
        st[16 * g + s_] = make_float4(t_.x * z_, t_.y * z_, t_.z * z_, t_.w * z_);
      }
    }
    if (a.bias) { b0 = v0ok ? a.bias[j0] : 0.f; b1 = v1ok ? a.bias[j1] : 0.f; }
    fetch_row(0);
  } else {
#pragma unroll
    for (int j = 0; j < JA; ++j) { acc[j].lo = f32x2{0.f, 0.f}; acc[j].hi = f32x2{0.f, 0.f}; }
    if (cols) {
      issue(0);
      put(0);
      if (nc > 1) issue(1);
    }
  }
  lds_barrier();
  for (int c = 0; c < nc; ++c) {
    const int64_t c0 = n0 + 16 * (int64_t)c;
    // ================= H1 =================
    if (is_a) {
      const float* sp = stage[c & 1] + r * kPitch + kbeg + 4 * q;
      f32x4 la[NG];
#pragma unroll
      for (int g = 0; g < NG; ++g) la[g] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t2 = 0; t2 < 16; t2 += 2) {
        if (t2 >= nsteps) break;             // wave-uniform
        const float4 x0 = *reinterpret_cast<const float4*>(sp + 16 * t2), x1 = *reinterpret_cast<const float4*>(sp + 16 * t2 + 16);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          la[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(x0.x, st[16 * g + t2].x, la[g], 0, 0, 0);
          la[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(x0.y, st[16 * g + t2].y, la[g], 0, 0, 0);
          la[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(x0.z, st[16 * g + t2].z, la[g], 0, 0, 0);
          la[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(x0.w, st[16 * g + t2].w, la[g], 0, 0, 0);
          la[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(x1.x, st[16 * g + t2 + 1].x, la[g], 0, 0, 0);
          la[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(x1.y, st[16 * g + t2 + 1].y, la[g], 0, 0, 0);
          la[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(x1.z, st[16 * g + t2 + 1].z, la[g], 0, 0, 0);
          la[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(x1.w, st[16 * g + t2 + 1].w, la[g], 0, 0, 0);
        }
      }
      // C/D layout of 16x16x4: column = lane & 15, row = (lane >> 4) * 4 + register
#pragma unroll
      for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int i = 0; i < 4; ++i) part[wv][4 * q + i][r + 16 * g] = la[g][i];
    } else if (cols && c > 0) {
      // gW[:, 4t .. 4t+3] += g_logits[c - 1]^T x[c - 1], rows and gradients from LDS (rows past the range carry zero gradients)
      const float* sc = stage[(c - 1) & 1] + r0 * kPitch + kc;
      const float (*gp)[32] = gl[(c - 1) & 1] + r0;
#pragma unroll
      for (int rr = 0; rr < RPT; ++rr) {
        const float4 x4 = *reinterpret_cast<const float4*>(sc + rr * kPitch);
        const f32x2 xlo = {x4.x, x4.y}, xhi = {x4.z, x4.w};
#pragma unroll
        for (int j4 = 0; j4 < JA; j4 += 4) {
          const float4 g = *reinterpret_cast<const float4*>(&gp[rr][j4]);   // same address in every lane of a row group: a broadcast read
          pk_fma4(acc[j4], g.x, xlo, xhi); pk_fma4(acc[j4 + 1], g.y, xlo, xhi); pk_fma4(acc[j4 + 2], g.z, xlo, xhi); pk_fma4(acc[j4 + 3], g.w, xlo, xhi);
        }
      }
    }
    lds_barrier();
    // ================= H2 =================
    if (is_a) {
      // softmax, loss terms and logit gradient of row c0 + 4 wv + q, 16 lanes per row
      const int rr = 4 * wv + q;
      const int64_t i = c0 + rr;
      const bool rowok = i < n1;
      const int64_t yl = yl_n; const float rn = rn_n, spv = sp_n;
      if (c + 1 < nc) fetch_row(c + 1);      // the next chunk's scalars: in flight under the next H1
      float v0 = v0ok ? b0 + part[0][rr][j0] + part[1][rr][j0] + part[2][rr][j0] + part[3][rr][j0] : -INFINITY;
      float v1 = -INFINITY;
      if (NG > 1) v1 = v1ok ? b1 + part[0][rr][j1] + part[1][rr][j1] + part[2][rr][j1] + part[3][rr][j1] : -INFINITY;
      const bool y_ok = yl >= 0 && yl < J && !poisoned;          // F.cross_entropy raises for such a label: here a NaN weight
      const int y = y_ok ? (int)yl : 0;
      const float m = (rowok && rn < spv) ? 1.f : 0.f;
      const float w = y_ok ? m * cwl[y] : (rowok ? NAN : 0.f);
      const float mx = row16_max(fmaxf(v0, v1));
      const float e0 = v0ok ? expf(v0 - mx) : 0.f, e1 = v1ok ? expf(v1 - mx) : 0.f;
      const float se = row16_sum(e0 + e1);
      const float vy = row16_sum((v0ok && j0 == y ? v0 : 0.f) + (v1ok && j1 == y ? v1 : 0.f));
      const float inv = 1.f / se;
      const float g0 = (v0ok && rowok) ? w * (e0 * inv - (j0 == y ? 1.f : 0.f)) : 0.f;
      const float g1 = (v1ok && rowok) ? w * (e1 * inv - (j1 == y ? 1.f : 0.f)) : 0.f;
      gl[c & 1][rr][j0] = g0; gl[c & 1][rr][j1] = g1;            // (NG == 1: j1 = 16 .. 31 are written as zeros, never read past JA)
      if (rowok) {
        if (v0ok) { a.logits[i * a.ldl + j0] = v0; a.g_logits[i * a.ldg + j0] = g0; }
        if (v1ok) { a.logits[i * a.ldl + j1] = v1; a.g_logits[i * a.ldg + j1] = g1; }
        if (r == 0) { num += w * (mx + logf(se) - vy); den += w; }
      }
      cs0 += g0; cs1 += g1;
    } else if (cols) {
      if (c + 1 < nc) put(c + 1);            // (the loads issued a step ago have had all of H1 to land)
      if (c + 2 < nc) issue(c + 2);
    }
    lds_barrier();
  }
  if (cols && nc > 0) {                      // the last chunk's weight-gradient contribution
    const float* sc = stage[(nc - 1) & 1] + r0 * kPitch + kc;
    const float (*gp)[32] = gl[(nc - 1) & 1] + r0;
#pragma unroll
    for (int rr = 0; rr < RPT; ++rr) {
      const float4 x4 = *reinterpret_cast<const float4*>(sc + rr * kPitch);
      const f32x2 xlo = {x4.x, x4.y}, xhi = {x4.z, x4.w};
#pragma unroll
      for (int j4 = 0; j4 < JA; j4 += 4) {
        const float4 g = *reinterpret_cast<const float4*>(&gp[rr][j4]);
        pk_fma4(acc[j4], g.x, xlo, xhi); pk_fma4(acc[j4 + 1], g.y, xlo, xhi); pk_fma4(acc[j4 + 2], g.z, xlo, xhi); pk_fma4(acc[j4 + 3], g.w, xlo, xhi);
      }
    }
  }
  if (cols) {
#pragma unroll
    for (int j = 0; j < JA; ++j)
      if (j < J) st4(a.wpart + (((int64_t)blockIdx.x * RH + half) * J + j) * a.Kp + kc, make_float4(acc[j].lo.x, acc[j].lo.y, acc[j].hi.x, acc[j].hi.y));
  }
  // bias gradient: this block's column sums of g_logits (lanes (r, q) of A-wave wv hold rows 4 wv + q of every chunk)
  if (is_a) { csred[4 * wv + q][j0] = cs0; csred[4 * wv + q][j1] = cs1; }
  num = wave_sum(num); den = wave_sum(den);
  if (is_a && lane == 0) { red[0][wv] = num; red[1][wv] = den; }
  __syncthreads();
  if (tid < 32 && a.colpart) {
    float t_ = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) t_ += csred[g][tid];
    __hip_atomic_store(a.colpart + (int64_t)blockIdx.x * 32 + tid, t_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (tid == 0) {
    const float s0 = red[0][0] + red[0][1] + red[0][2] + red[0][3], s1 = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    __hip_atomic_store(a.partial + 2 * blockIdx.x, s0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(a.partial + 2 * blockIdx.x + 1, s1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // this thread's device-scope stores are out ...
  __syncthreads();                                                // ... and so are every thread's, before the ticket is taken
  if (tid == 0) last = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
  __syncthreads();
  if (!last) return;
  // the workgroup that arrives last adds the per-block pairs and column sums in block order (bitwise independent of arrival)
  if (a.colpart && a.colsum) {
    if (tid < 256) {
      const int c = tid & 31, g8 = tid >> 5;
      float t_ = 0.f;
      for (unsigned b = g8; b < gridDim.x; b += 8) t_ += __hip_atomic_load(a.colpart + (int64_t)b * 32 + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      csred[g8][c] = t_;
    }
    __syncthreads();
    if (tid < J) {
      float u_ = 0.f;
#pragma unroll
      for (int g = 0; g < 8; ++g) u_ += csred[g][tid];
      a.colsum[tid] = u_;
    }
  }
  float sa = 0.f, sb = 0.f;
  if (tid < 256)
    for (unsigned b = tid; b < gridDim.x; b += 256) {
      sa += __hip_atomic_load(a.partial + 2 * b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      sb += __hip_atomic_load(a.partial + 2 * b + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  sa = wave_sum(sa); sb = wave_sum(sb);
  __syncthreads();
  if (is_a && lane == 0) { red[0][wv] = sa; red[1][wv] = sb; }
  __syncthreads();
  if (tid == 0) {
    a.sums[0] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    a.sums[1] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    *a.ticket = 0u;                                               // re-armed for the next launch on the stream
  }
}

}  // namespace

extern "C" {

static int classifier_ce_row_groups(int32_t K, int32_t J) {
  // the 22-class airway heads (J in 17 .. 24) on rows of at most 512 columns: 2 (K > 256) or 4 row groups of C-threads
  if (J < 17 || J > 24 || K > 512) return 1;
  return K > 256 ? 2 : 4;
}

int64_t spgnn_classifier_ce_partial_slices(int64_t N, int32_t K, int32_t J) {
  if (N <= 0) return 0;
  const int64_t rps = spgnn_classifier_ce_rows_per_block(N);
  return (N + rps - 1) / rps * classifier_ce_row_groups(K, J);
}

int spgnn_classifier_ce_rows_per_block(int64_t N) {
  // one workgroup per CU (256 of them) at the 512-tree batch, never fewer than 64 rows each: the J x K partial a workgroup
  // writes at its end (90 KB at 22 x 1024) must stay small against the rows it read (4 KB each)
  int64_t rps = (N + 255) / 256;
  rps = (rps + 15) / 16 * 16;
  return (int)(rps < 64 ? 64 : rps);
}

static int classifier_ce_launch(const void* x, bool bf16, int64_t x_stride, const float* w, int32_t Kp, const float* bias,
                                const int64_t* labels, const float* draws, uint64_t draw_seed, const int64_t* seed_offset,
                                const float* sampling_p, const float* class_weight, const int32_t* flag, float* logits,
                                int64_t logits_stride, float* g_logits, int64_t g_stride, float* w_partials, float* partials, float* sums,
                                uint32_t* ticket, float* colsum_partials, float* g_colsum, int64_t N, int32_t K, int32_t J,
                                spgnn_stream_t stream) {
  if (N <= 0 || K <= 0 || (K & 127) || K > 1024 || J <= 0 || J > 32 || Kp < K || (Kp & 15))
    return fail(SPGNN_ERR_SHAPE, "spgnn_classifier_ce: need N > 0, K % 128 == 0, K <= 1024, J <= 32, Kp >= K, Kp % 16 == 0");
  if (!x || !w || !labels || !sampling_p || !class_weight || !logits || !g_logits || !w_partials || !partials || !sums || !ticket)
    return fail(SPGNN_ERR_NULLPTR, "spgnn_classifier_ce: null pointer");
  if ((g_colsum != nullptr) != (colsum_partials != nullptr)) return fail(SPGNN_ERR_NULLPTR, "spgnn_classifier_ce: colsum needs its partials");
  if (x_stride < K || (x_stride & 3) || (reinterpret_cast<uintptr_t>(x) & (bf16 ? 7 : 15)) || !aligned16(w) || !aligned16(w_partials) ||
      logits_stride < J || g_stride < J)
    return fail(SPGNN_ERR_STRIDE, "spgnn_classifier_ce: x rows must be 16-byte (bf16: 8-byte) aligned with stride % 4 == 0, w / partial rows 16-byte aligned, logit rows at least J wide");
  const int64_t rps = spgnn_classifier_ce_rows_per_block(N);
  const unsigned nb = (unsigned)((N + rps - 1) / rps);
  ClsCe a{x, x_stride, w, Kp, bias, labels, draws, draw_seed, seed_offset, sampling_p, class_weight, flag, logits, logits_stride,
          g_logits, g_stride, w_partials, partials, sums, ticket, colsum_partials, g_colsum, N, K, J, rps};
  hipStream_t st = (hipStream_t)stream;
#define X(ST_, NG_, JA_, RH_) hipLaunchKernelGGL((classifier_ce_kernel<ST_, NG_, JA_, RH_>), dim3(nb), dim3(kCeThreads), 0, st, a)
  const int rh = classifier_ce_row_groups(K, J);
#define Y(ST_) { if (J <= 4) X(ST_, 1, 4, 1); else if (J <= 8) X(ST_, 1, 8, 1); else if (J <= 16) X(ST_, 1, 16, 1); \
                 else if (J <= 24) { if (rh == 4) X(ST_, 2, 24, 4); else if (rh == 2) X(ST_, 2, 24, 2); else X(ST_, 2, 24, 1); } else X(ST_, 2, 32, 1); }
  if (bf16) Y(bf16s) else Y(float)
#undef Y
#undef X
  return check_launch("spgnn_classifier_ce");
}

int spgnn_classifier_ce(const float* x, int64_t x_stride, const float* w, int32_t Kp, const float* bias, const int64_t* labels,
                        const float* draws, uint64_t draw_seed, const int64_t* seed_offset, const float* sampling_p,
                        const float* class_weight, const int32_t* flag, float* logits, int64_t logits_stride, float* g_logits,
                        int64_t g_stride, float* w_partials, float* partials, float* sums, uint32_t* ticket, float* colsum_partials,
                        float* g_colsum, int64_t N, int32_t K, int32_t J, spgnn_stream_t stream) {
  return classifier_ce_launch(x, false, x_stride, w, Kp, bias, labels, draws, draw_seed, seed_offset, sampling_p, class_weight, flag, logits,
                              logits_stride, g_logits, g_stride, w_partials, partials, sums, ticket, colsum_partials, g_colsum, N, K, J, stream);
}

int spgnn_classifier_ce_bf16(const uint16_t* x, int64_t x_stride, const float* w, int32_t Kp, const float* bias, const int64_t* labels,
                             const float* draws, uint64_t draw_seed, const int64_t* seed_offset, const float* sampling_p,
                             const float* class_weight, const int32_t* flag, float* logits, int64_t logits_stride, float* g_logits,
                             int64_t g_stride, float* w_partials, float* partials, float* sums, uint32_t* ticket, float* colsum_partials,
                             float* g_colsum, int64_t N, int32_t K, int32_t J, spgnn_stream_t stream) {
  return classifier_ce_launch(x, true, x_stride, w, Kp, bias, labels, draws, draw_seed, seed_offset, sampling_p, class_weight, flag, logits,
                              logits_stride, g_logits, g_stride, w_partials, partials, sums, ticket, colsum_partials, g_colsum, N, K, J, stream);
}

}  // extern "C"
