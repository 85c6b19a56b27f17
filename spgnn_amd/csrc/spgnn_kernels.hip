// spgnn_kernels.hip — gfx950 (MI355X / CDNA4) message-passing kernels behind include/spgnn_hip.h.
//
// Workload shape: batched airway trees, in-degree 2..5, E = 3N - 2B.  The
// parallelism is in N x H x D, not in the neighbour reduction, so every kernel maps a TEAM of
// T in {16,32,64} lanes to one node, each lane owning R float4 chunks of the node's H*D row
// (H*D = 4*T*R), and walks the node's <=5 edges serially.  All row accesses are 16-byte vector
// loads, contiguous across the team (up to 1 KiB per wave instruction).  Blocks are remapped so
// that each XCD (private 4 MiB L2) sweeps a contiguous node range: a tree's rows (its neighbours
// live within ~150 rows) are then re-read from that XCD's L2 rather than from HBM.
//
// No atomics anywhere: forward and the dst-major backward half reduce over in-edges (CSC), the
// src-major backward half over out-edges (CSR + slot map), so results are run-to-run bitwise
// reproducible.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>

#include <mutex>

#include "spgnn_hip.h"
#include "spgnn_internal.h"
#include "spgnn_rows.h"

namespace spgnn_detail {

thread_local char g_err[512] = "";

int fail(int code, const char* msg) {
  snprintf(g_err, sizeof(g_err), "%s", msg);
  return code;
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    snprintf(g_err, sizeof(g_err), "%s: launch failed: %s", what, hipGetErrorString(e));
    return -(1000 + (int)e);
  }
  return SPGNN_OK;
}

int fail_at(int code, const char* func, int line) {
  const char* kind = code == SPGNN_ERR_NULLPTR ? "null pointer" : code == SPGNN_ERR_SHAPE ? "bad shape / size argument"
                   : code == SPGNN_ERR_STRIDE ? "bad row stride or alignment" : code == SPGNN_ERR_ENUM ? "bad enum / option value"
                   : "argument error";
  snprintf(g_err, sizeof(g_err), "%s: %s (line %d)", func, kind, line);
  return code;
}

int ensure_dynamic_lds(const void* func, int bytes) {
  constexpr int kMaxDev = 64, kMaxFn = 64;
  static const void* fns[kMaxFn];
  static int sizes[kMaxFn][kMaxDev];
  static std::mutex mu;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) dev = 0;
  std::lock_guard<std::mutex> lock(mu);
  int slot = -1;
  for (int i = 0; i < kMaxFn; ++i) {
    if (fns[i] == func) { slot = i; break; }
    if (fns[i] == nullptr) { fns[i] = func; slot = i; break; }
  }
  if (slot >= 0 && sizes[slot][dev] >= bytes) return SPGNN_OK;
  hipError_t e = hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) {
    snprintf(g_err, sizeof(g_err), "hipFuncSetAttribute(MaxDynamicSharedMemorySize = %d): %s", bytes, hipGetErrorString(e));
    (void)hipGetLastError();
    return -(1000 + (int)e);
  }
  if (slot >= 0) sizes[slot][dev] = bytes;
  return SPGNN_OK;
}

}  // namespace spgnn_detail

namespace {

using spgnn_detail::check_launch;
using spgnn_detail::fail;
using spgnn_detail::g_err;

constexpr int kAbpRows = 4;       // rows per trip of act_bwd_proj (2 measured slower: see DESIGN.md)

// =================================================================================================
// GAT kernels.  Template parameters of the vector kernels:
//   R   float4 chunks per lane (H*D = 4*T*R)
//   CH  chunks of one lane that belong to one head when a head is at least a team wide
//       (D = 4*T*CH, the lane then sees NS = R/CH heads, head index = chunk / CH, uniform per team);
//       CH = 0 when a head is narrower than the team (4*T % D == 0): every chunk is its own slot,
//       head index = column / D varies across lanes, reductions run over W = D/4 lanes.
// kMaxFast: nodes with at most this many edges (every airway node: in-degree <= 5) take a path that
// loads all edge indices and scores up front (independent loads) and keeps per-edge weights in
// registers; larger degrees fall back to multi-pass loops of identical arithmetic.
// =================================================================================================
// Code shape (measured with tools/ab_kernels.py, MI355X, 512 trees; the losing forms are not in the tree).  With exec-masked
// per-edge branches (one s_waitcnt per load) the up-front path only paid off for the forward; written as unconditional
// clamped loads + wave-uniform scalars it wins everywhere: fwd 1213 -> 692 us, bwd_dst 1162 -> 779 us, bwd_src 540 -> 451 us
// summed over the seven st_pgat_spgnn_3 layer shapes (2x1024 forward alone 770 -> 385 us = 5.7 TB/s algorithmic).  When a
// head is at least a team wide (CH >= 1) the softmax runs one (edge slot, head) ENTRY per lane with the weights broadcast
// back, instead of the whole table in every lane (7 % off the GAT kernels' time).
//
// The dst-major backward half in that entry-per-lane form (reduce-scatter of the per-edge dots, below) is why this file is
// built without hipcc's vectorizers: see the note on packed fp32 ops in spgnn_rows.h.

template <int R, int CH> struct Slots {
  static constexpr int NS = (CH == 0) ? R : R / (CH == 0 ? 1 : CH);
  static __device__ __forceinline__ constexpr int of(int r) { return (CH == 0) ? r : r / (CH == 0 ? 1 : CH); }
};

// -------------------------------------------------------------------------------------------------
// forward
// -------------------------------------------------------------------------------------------------
template <typename ST> struct GatFwdT {
  const int32_t* indptr; const int32_t* indices;
  const int32_t* nbr8;                   // optional (N, 8): in-neighbours of v, slots >= deg repeat the last one (see spgnn_gat_fwd)
  const ST* ft; int64_t ft_ld;
  const float* el; const float* er; int64_t s_ld;
  const ST* res; int64_t res_ld;
  const float* bias;
  ST* out; int64_t out_ld;               // per-head output (N, H*D); may be null when out_mean is set
  float* out_mean; int64_t out_mean_ld;  // optional head-mean output (N, D)
  float* attn;
  int64_t N; int H; int D; int T;
  float slope; int act; float p; float inv_keep; uint64_t seed;
  const uint64_t* seed_off;              // optional device word added to `seed` (fresh masks under graph replay)
  // optional feature dropout of the CONSUMER applied to the stored per-head output (spgnn_gat_fwd: out_drop_*): the
  // rows land in the next layer's input buffer already dropped, mask = spgnn_cat_dropout's for (fseed, ftotal, foff)
  float fp; float finv; uint64_t fseed; int ftotal; int foff;
  float* absmax;                         // optional scale block (spgnn_internal.h): max |stored out| is folded into its slots
};
using GatFwd = GatFwdT<float>;

// TT = 64: the team is the whole wave (one node per wave).  Node id, degree, edge endpoints and - when a head is
// at least a wave wide (CH >= 1) - the attention weights are then wave-uniform: they are moved to SGPRs
// (v_readfirstlane), row bases become scalar, the degree tests become scalar branches, and about 30 VGPRs per
// lane are freed (2x1024 forward: 159 -> ~90 VGPRs).  TT = 0: team width a.T < 64 at run time (several nodes
// per wave; only R = 1 geometries get there), everything stays per lane.
template <typename ST, int TT, int R, int CH, bool MEAN>
__global__ __launch_bounds__(kBlock) void gat_fwd_vec(GatFwdT<ST> a) {
  if (a.seed_off) { a.seed += a.seed_off[0]; a.fseed += a.seed_off[0]; }
  using SL = Slots<R, CH>;
  constexpr int NS = SL::NS;
  constexpr bool WAVE = TT == 64;
  constexpr bool UW = WAVE && CH >= 1;          // per-edge weights are wave-uniform too
  const int T = WAVE ? 64 : a.T;
  const int lane = threadIdx.x % T;
  const int64_t v = xcd_block() * (kBlock / T) + uni<WAVE>((int)(threadIdx.x / T));
  if (v >= a.N) return;
  const int beg = uni<WAVE>(a.indptr[v]), end = uni<WAVE>(a.indptr[v + 1]), deg = end - beg;

  int hs[NS]; bool wr[NS]; float erv[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int c0 = ((CH == 0 ? s : s * CH) * T + lane) * 4;
    hs[s] = (CH == 0) ? c0 / a.D : s;
    wr[s] = (CH == 0) ? (c0 % a.D == 0) : (lane == 0);
    erv[s] = a.er[v * a.s_ld + hs[s]];
  }
  // The accumulator starts from the residual row (+ bias): those loads depend on nothing, so they are in flight
  // while the index -> score -> neighbour-row chain resolves (as an epilogue they cost a serial HBM round trip).
  float4 acc[R];
#pragma unroll
  for (int r = 0; r < R; ++r) acc[r] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (a.res) {
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = ldv(a.res + v * a.res_ld + (r * T + lane) * 4);
  }
  if (a.bias) {
    float4 q[R];
#pragma unroll
    for (int r = 0; r < R; ++r) q[r] = ld4(a.bias + (r * T + lane) * 4);
#pragma unroll
    for (int r = 0; r < R; ++r) { acc[r].x += q[r].x; acc[r].y += q[r].y; acc[r].z += q[r].z; acc[r].w += q[r].w; }
  }

  // Neighbour ids.  With the padded (N, 8) neighbour rows they depend on v only, so they are fetched together with
  // indptr[v] (one round trip) instead of after it: the per-node dependent chain is two memory round trips
  // (ids + degree, then scores + rows) instead of three.
  int u[kMaxFast];
  const bool ell = a.nbr8 != nullptr;
  if (ell) {
#pragma unroll
    for (int k = 0; k < kMaxFast; ++k) u[k] = uni<WAVE>(a.nbr8[v * 8 + k]);
  }
  if (deg > 0 && deg <= kMaxFast) {
    // Straight-line loads: every index / score load is unconditional (slot k >= deg re-reads the last edge and
    // gets weight 0), so they issue back to back instead of one exec-masked branch and one s_waitcnt per edge;
    // neighbour rows then arrive in batches of kGather edges behind a wave-uniform test.
    if (!ell) {
#pragma unroll
      for (int k = 0; k < kMaxFast; ++k) u[k] = uni<WAVE>(a.indices[beg + (k < deg ? k : deg - 1)]);
    }
    float w[kMaxFast][NS];
    if constexpr (CH >= 1) {
      // One (edge slot, head) ENTRY per lane instead of the whole 8 x NS table in every lane.  With a head at least a
      // team wide (CH >= 1) every lane of the team needs the same NS x 8 attention weights; computing the table
      // redundantly per lane made the narrow layers (<= 128 columns, 16-lane teams) VALU-bound: 16 expf, 16 divisions
      // and - with attention dropout - 16 64-bit hashes per lane against 64 FMAs of payload.  Entry e = s * 8 + k lives
      // in lane e % T (register e / T): leaky-relu, the segment max / sum over the 8 lanes of a head (xor shuffles),
      // exp, the division, the attention store and the dropout hash are done once per entry, then the NS x 8 weights
      // are broadcast to the lanes (v_readlane into SGPRs when the team is the wave).
      constexpr int NENT = NS * 8;
      constexpr int NREG = WAVE ? 1 : (NENT + 15) / 16;       // registers per lane (narrowest team: 16 lanes; a wave holds all 64)
      const int wl = threadIdx.x & 63, tbase = wl & ~(T - 1);
      float al[NREG];
#pragma unroll
      for (int i = 0; i < NREG; ++i) {
        const int e = lane + i * T;                           // this lane's entry (teams wider than NENT: idle lanes)
        const int k = e & 7, s_ = e >> 3;
        const bool own = e < NENT, valid = own && k < deg;
        const int kk = k < deg ? k : deg - 1;
        const int hh = own ? s_ : 0;
        const int ue = ell ? a.nbr8[v * 8 + k] : a.indices[beg + kk];
        float x = a.el[(int64_t)ue * a.s_ld + hh] + a.er[v * a.s_ld + hh];
        x = valid ? lrelu(x, a.slope) : -INFINITY;
        float mx = group8_max(x);
        const float ex = valid ? expf(x - mx) : 0.f;
        float sm = group8_sum(ex);
        float a_ = ex / sm;
        if (valid) a.attn[(int64_t)(beg + k) * a.H + hh] = a_;
        if (a.p > 0.f) a_ *= keep_scale(a.seed, (int64_t)(beg + kk) * a.H + hh, a.p, a.inv_keep);
        al[i] = valid ? a_ : 0.f;
      }
#pragma unroll
      for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int k = 0; k < kMaxFast; ++k) {
          // entry e = s * 8 + k sits in lane e % T, register e / T (T = 64: always register 0)
          if constexpr (WAVE) {
            w[k][s] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(al[0]), s * 8 + k));
          } else {
            const int e = s * 8 + k;
            float v0 = __shfl(al[0], tbase + (e & (T - 1)), 64);
            if constexpr (NREG > 1) { const float v1 = __shfl(al[NREG - 1], tbase + (e & (T - 1)), 64); v0 = (e / T) ? v1 : v0; }
            w[k][s] = v0;
          }
        }
    } else {
#pragma unroll
    for (int k = 0; k < kMaxFast; ++k)
#pragma unroll
      for (int s = 0; s < NS; ++s) w[k][s] = a.el[(int64_t)u[k] * a.s_ld + hs[s]];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      float mx = -INFINITY;
#pragma unroll
      for (int k = 0; k < kMaxFast; ++k) {
        w[k][s] = k < deg ? lrelu(w[k][s] + erv[s], a.slope) : -INFINITY;
        mx = fmaxf(mx, w[k][s]);
      }
      float sm = 0.f;
#pragma unroll
      for (int k = 0; k < kMaxFast; ++k) {
        w[k][s] = k < deg ? expf(w[k][s] - mx) : 0.f;
        sm += w[k][s];
      }
#pragma unroll
      for (int k = 0; k < kMaxFast; ++k) {
        const float al = w[k][s] / sm;
        if (wr[s] && k < deg) a.attn[(int64_t)(beg + k) * a.H + hs[s]] = al;
        w[k][s] = al;
      }
    }
    if (a.p > 0.f) {
#pragma unroll
      for (int k = 0; k < kMaxFast; ++k)
#pragma unroll
        for (int s = 0; s < NS; ++s)
          w[k][s] *= keep_scale(a.seed, (int64_t)(beg + (k < deg ? k : deg - 1)) * a.H + hs[s], a.p, a.inv_keep);
    }
#pragma unroll
    for (int k = 0; k < kMaxFast; ++k)
#pragma unroll
      for (int s = 0; s < NS; ++s) w[k][s] = uni<UW>(w[k][s]);
    }
    constexpr int kGather = (R >= 8 ? 1 : R == 4 ? 2 : 4);   // edges per batch (8 float4 in flight for R >= 2)
#pragma unroll
    for (int k0 = 0; k0 < kMaxFast; k0 += kGather) {
      if (WAVE ? !(k0 < deg) : !__any(k0 < deg)) break;
      float4 x[kGather][R];
#pragma unroll
      for (int q = 0; q < kGather; ++q)
#pragma unroll
        for (int r = 0; r < R; ++r) x[q][r] = ldv(a.ft + (int64_t)u[k0 + q] * a.ft_ld + (r * T + lane) * 4);
#pragma unroll
      for (int q = 0; q < kGather; ++q)
#pragma unroll
        for (int r = 0; r < R; ++r) fma4(acc[r], w[k0 + q][SL::of(r)], x[q][r]);
    }
  } else {
    float mx[NS], sm[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      float m_ = -INFINITY;
      for (int j = beg; j < end; ++j)
        m_ = fmaxf(m_, lrelu(a.el[(int64_t)a.indices[j] * a.s_ld + hs[s]] + erv[s], a.slope));
      float s_ = 0.f;
      for (int j = beg; j < end; ++j)
        s_ += expf(lrelu(a.el[(int64_t)a.indices[j] * a.s_ld + hs[s]] + erv[s], a.slope) - m_);
      mx[s] = m_; sm[s] = s_;
    }
    for (int j = beg; j < end; ++j) {
      const int64_t u = a.indices[j];
      const ST* row = a.ft + u * a.ft_ld;
      float w[NS];
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const float al = expf(lrelu(a.el[u * a.s_ld + hs[s]] + erv[s], a.slope) - mx[s]) / sm[s];
        const int64_t eidx = (int64_t)j * a.H + hs[s];
        if (wr[s]) a.attn[eidx] = al;
        w[s] = a.p > 0.f ? al * keep_scale(a.seed, eidx, a.p, a.inv_keep) : al;
      }
#pragma unroll
      for (int r = 0; r < R; ++r) fma4(acc[r], w[SL::of(r)], ldv(row + (r * T + lane) * 4));
    }
  }

  act_fwd_rows<R>(acc, a.act);
  if (a.out) {
    float amx = 0.f;
    if (a.fp > 0.f) {                     // store dropout(out) for the consumer: the un-dropped rows are never needed again
#pragma unroll                            // (where an element is dropped its gradient is zero whatever act' was)
      for (int r = 0; r < R; ++r) {
        const int c = (r * T + lane) * 4;
        const float4 kf = feat_keep4(a.fseed, v * a.ftotal + a.foff + c, a.fp, a.finv);
        const float4 d = make_float4(acc[r].x * kf.x, acc[r].y * kf.y, acc[r].z * kf.z, acc[r].w * kf.w);
        stv(a.out + v * a.out_ld + c, d);
        amx = absmax4(amx, d);
      }
    } else {
#pragma unroll
      for (int r = 0; r < R; ++r) { stv(a.out + v * a.out_ld + (r * T + lane) * 4, acc[r]); amx = absmax4(amx, acc[r]); }
    }
    if (a.absmax) {
      amx = team_max(amx, T);
      if (lane == 0) spgnn_detail::slots_max(a.absmax, amx, (unsigned)v);
    }
  }
  if (MEAN) {   // heads live in chunks r, r+CH, r+2CH, ... of the same lane (CH >= 1): mean is lane-local
    constexpr int CHs = CH == 0 ? 1 : CH;
    const float inv_h = 1.f / (float)a.H;
#pragma unroll
    for (int rr = 0; rr < CHs; ++rr) {
      float4 m = acc[rr];
#pragma unroll
      for (int s = 1; s < NS; ++s) { const float4 q = acc[s * CHs + rr]; m.x += q.x; m.y += q.y; m.z += q.z; m.w += q.w; }
      m.x *= inv_h; m.y *= inv_h; m.z *= inv_h; m.w *= inv_h;
      st4(a.out_mean + v * a.out_mean_ld + (rr * T + lane) * 4, m);
    }
  }
}

// teams narrower than a wave exist only for R = 1 (pick_team tries 64 lanes first): keep the other instances out
template <typename ST, int R, int CH, bool MEAN> static void launch_small_team(dim3 grid, dim3 block, hipStream_t st, const GatFwdT<ST>& a) {
  if constexpr (R <= 4) hipLaunchKernelGGL((gat_fwd_vec<ST, 0, R, CH, MEAN>), grid, block, 0, st, a);
}

// scalar fallback: one thread per (node, column); any H, D, stride, alignment
__global__ void gat_fwd_scalar(GatFwd a) {
  if (a.seed_off) a.seed += a.seed_off[0];
  const int HD = a.H * a.D;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= a.N * HD) return;
  const int64_t v = gid / HD; const int col = (int)(gid % HD); const int h = col / a.D;
  const int beg = a.indptr[v], end = a.indptr[v + 1];
  const float erv = a.er[v * a.s_ld + h];
  float mx = -INFINITY;
  for (int j = beg; j < end; ++j) mx = fmaxf(mx, lrelu(a.el[(int64_t)a.indices[j] * a.s_ld + h] + erv, a.slope));
  float sm = 0.f;
  for (int j = beg; j < end; ++j) sm += expf(lrelu(a.el[(int64_t)a.indices[j] * a.s_ld + h] + erv, a.slope) - mx);
  float acc = 0.f;
  for (int j = beg; j < end; ++j) {
    const int64_t u = a.indices[j];
    const float al = expf(lrelu(a.el[u * a.s_ld + h] + erv, a.slope) - mx) / sm;
    if (col % a.D == 0) a.attn[(int64_t)j * a.H + h] = al;
    const float w = a.p > 0.f ? al * keep_scale(a.seed, (int64_t)j * a.H + h, a.p, a.inv_keep) : al;
    acc = fmaf(w, a.ft[u * a.ft_ld + col], acc);
  }
  if (a.res) acc += a.res[v * a.res_ld + col];
  if (a.bias) acc += a.bias[col];
  acc = act_fwd(acc, a.act);
  a.out[v * a.out_ld + col] = acc;
  if (a.absmax) spgnn_detail::slots_max(a.absmax, fabsf(acc), (unsigned)v);
}

// head mean for shapes the vector kernel cannot fuse: out_mean[v,d] = mean_h out[v,h,d]
__global__ void head_mean_scalar(const float* out, int64_t out_ld, float* om, int64_t om_ld, int64_t N, int H, int D) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= N * D) return;
  const int64_t v = gid / D; const int d = (int)(gid % D);
  float s = 0.f;
  for (int h = 0; h < H; ++h) s += out[v * out_ld + h * D + d];
  om[v * om_ld + d] = s / (float)H;
}

// -------------------------------------------------------------------------------------------------
// backward, dst-major half
// -------------------------------------------------------------------------------------------------
template <typename ST> struct GatBwdDstT {
  const int32_t* indptr; const int32_t* indices;
  const int32_t* nbr8;
  const ST* ft; int64_t ft_ld;
  const float* el; const float* er; int64_t s_ld;
  const float* attn;
  const void* g_out; int64_t g_out_ld;   // (N, H*D) of ST, or the head mean's gradient (N, D), always fp32, when mean != 0
  const ST* out; int64_t out_ld;
  ST* g_pre; int64_t g_pre_ld;
  float* g_e;
  float* g_er; int64_t gs_ld;
  float* absmax;                         // optional scale block: max |g_pre| is folded into its slots
  int64_t N; int H; int D; int T; int W; int mean;
  float slope; int act; float p; float inv_keep; uint64_t seed;
  const uint64_t* seed_off;
  float fp; float finv; uint64_t fseed; int ftotal; int foff;     // `out` was stored dropped (see GatFwdT): g_out gets the same mask
};
using GatBwdDst = GatBwdDstT<float>;

template <typename ST, int TT, int R, int CH>
__global__ __launch_bounds__(kBlock) void gat_bwd_dst_vec(GatBwdDstT<ST> a) {
  if (a.seed_off) { a.seed += a.seed_off[0]; a.fseed += a.seed_off[0]; }
  using SL = Slots<R, CH>;
  constexpr int NS = SL::NS;
  constexpr bool WAVE = TT == 64;               // see gat_fwd_vec
  constexpr bool UW = WAVE && CH >= 1;
  const int T = WAVE ? 64 : a.T;
  const int lane = threadIdx.x % T;
  const int64_t v = xcd_block() * (kBlock / T) + uni<WAVE>((int)(threadIdx.x / T));
  if (v >= a.N) return;
  const int beg = uni<WAVE>(a.indptr[v]), end = uni<WAVE>(a.indptr[v + 1]), deg = end - beg;
  const int width = (CH == 0) ? a.W : T;
  const float gscale = a.mean ? 1.f / (float)a.H : 1.f;

  // g = dL/d(pre-activation row): all row loads first, then the activation derivative with one uniform switch
  float4 g[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int c = (r * T + lane) * 4;
    if (is_f32<ST>::value || a.mean)
      g[r] = ld4(reinterpret_cast<const float*>(a.g_out) + v * a.g_out_ld +
                 (a.mean ? (CH >= 1 ? ((r % (CH ? CH : 1)) * T + lane) * 4 : c % a.D) : c));
    else
      g[r] = ldv(reinterpret_cast<const ST*>(a.g_out) + v * a.g_out_ld + c);
  }
  if (a.fp > 0.f) {                       // the forward pass stored dropout(out): the same mask on the incoming gradient
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const float4 kf = feat_keep4(a.fseed, v * a.ftotal + a.foff + (r * T + lane) * 4, a.fp, a.finv);
      g[r].x *= kf.x; g[r].y *= kf.y; g[r].z *= kf.z; g[r].w *= kf.w;
    }
  }
  if (a.act != SPGNN_ACT_NONE) {
    float4 o[R];
#pragma unroll
    for (int r = 0; r < R; ++r) o[r] = ldv(a.out + v * a.out_ld + (r * T + lane) * 4);
    if (a.fp > 0.f) {                     // kept elements: out = stored / inv_keep; dropped ones are irrelevant (g is 0 there)
      const float un = 1.f - a.fp;
#pragma unroll
      for (int r = 0; r < R; ++r) { o[r].x *= un; o[r].y *= un; o[r].z *= un; o[r].w *= un; }
    }
    if (a.mean) {
#pragma unroll
      for (int r = 0; r < R; ++r) { g[r].x *= gscale; g[r].y *= gscale; g[r].z *= gscale; g[r].w *= gscale; }
    }
    act_bwd_rows<R>(g, o, a.act);
  } else if (a.mean) {
#pragma unroll
    for (int r = 0; r < R; ++r) { g[r].x *= gscale; g[r].y *= gscale; g[r].z *= gscale; g[r].w *= gscale; }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) stv(a.g_pre + v * a.g_pre_ld + (r * T + lane) * 4, g[r]);
  if (!is_f32<ST>::value) {      // the dots below must see what the src-major half and the GEMMs will read: the ROUNDED g_pre
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const f32x4_t f = {g[r].x, g[r].y, g[r].z, g[r].w};
      const bf16x4_t h = __builtin_convertvector(f, bf16x4_t);
      const f32x4_t b = __builtin_convertvector(h, f32x4_t);
      g[r] = make_float4(b[0], b[1], b[2], b[3]);
    }
  }
  if (a.absmax) {
    float mx = 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r) mx = absmax4(mx, g[r]);
    mx = team_max(mx, T);
    if (lane == 0) spgnn_detail::slots_max(a.absmax, mx, (unsigned)v);
  }
  int hs[NS]; bool wr[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int c0 = ((CH == 0 ? s : s * CH) * T + lane) * 4;
    hs[s] = (CH == 0) ? c0 / a.D : s;
    wr[s] = (CH == 0) ? (c0 % a.D == 0) : (lane == 0);
  }

  if (deg > 0 && deg <= kMaxFast) {
    // unconditional, batched loads (slot k >= deg repeats the last edge with attention 0): see gat_fwd_vec
    int u[kMaxFast];
    if (a.nbr8) {                        // ids from the padded neighbour rows: they depend on v only (fetched with indptr[v])
#pragma unroll
      for (int k = 0; k < kMaxFast; ++k) u[k] = uni<WAVE>(a.nbr8[v * 8 + k]);
    } else {
#pragma unroll
      for (int k = 0; k < kMaxFast; ++k) u[k] = uni<WAVE>(a.indices[beg + (k < deg ? k : deg - 1)]);
    }
    if constexpr (CH >= 1 && (NS & (NS - 1)) == 0 && (WAVE || NS * 8 <= 16)) {   // the table must fit the team
      // One (edge slot, head) entry per lane (see gat_fwd_vec).  The NS x 8 per-edge dots <ft[u], g_pre[v]> are summed
      // over the team by a reduce-scatter - each round halves the values a lane carries, 8 NS - 1 shuffles in all instead
      // of log2(T) per value - which leaves entry e = s * 8 + k complete in lane e; the softmax / LeakyReLU backward, the
      // dropout hash and the stores then run once per entry.
      constexpr int NENT = NS * 8;
      float pd[NENT];
#pragma unroll
      for (int e = 0; e < NENT; ++e) pd[e] = 0.f;
      constexpr int kGatherD = (R >= 8 ? 1 : R == 4 ? 2 : 4);
#pragma unroll
      for (int k0 = 0; k0 < kMaxFast; k0 += kGatherD) {
        if (WAVE ? !(k0 < deg) : !__any(k0 < deg)) break;
        float4 x[kGatherD][R];
#pragma unroll
        for (int q = 0; q < kGatherD; ++q)
#pragma unroll
          for (int r = 0; r < R; ++r) x[q][r] = ldv(a.ft + (int64_t)u[k0 + q] * a.ft_ld + (r * T + lane) * 4);
#pragma unroll
        for (int q = 0; q < kGatherD; ++q)
#pragma unroll
          for (int r = 0; r < R; ++r) pd[SL::of(r) * 8 + k0 + q] += dot4(x[q][r], g[r]);
      }
#pragma unroll
      for (int half = NENT / 2; half >= 1; half >>= 1) {        // entry bit `half` pairs with lane bit `half`
        const bool up = (lane & half) != 0;
#pragma unroll
        for (int i = 0; i < half; ++i) {
          float keep, send;
          rs_pair(up, pd[i], pd[i + half], keep, send);
          pd[i] = keep + __shfl_xor(send, half, 64);
        }
      }
      float ga = single_pass(pd[0]);
      for (int off = NENT; off < T; off <<= 1) ga = single_pass(ga + __shfl_xor(ga, off, 64));   // teams wider than the table
      const int e = lane & (NENT - 1), k = e & 7, hh = e >> 3;
      const bool valid = k < deg;
      const int kk = valid ? k : deg - 1;
      const int64_t slot = (int64_t)(beg + kk) * a.H + hh;
      float al = a.attn[slot];
      const int ue = a.nbr8 ? a.nbr8[v * 8 + k] : a.indices[beg + kk];
      const float epre = a.el[(int64_t)ue * a.s_ld + hh] + a.er[v * a.s_ld + hh];
      al = valid ? al : 0.f;
      if (a.p > 0.f) ga *= keep_scale(a.seed, slot, a.p, a.inv_keep);
      float S = group8_sum(valid ? al * ga : 0.f);
      float ge = al * ga - al * S;
      ge = epre > 0.f ? ge : ge * a.slope;
      ge = valid ? ge : 0.f;
      const bool writer = lane < NENT;                         // wider teams hold identical copies of the table
      if (writer && valid) a.g_e[slot] = ge;
      float ger = group8_sum(ge);
      if (writer && k == 0) a.g_er[v * a.gs_ld + hh] = ger;
      return;
    }
    float al[kMaxFast][NS], ep[kMaxFast][NS], erv[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) erv[s] = a.er[v * a.s_ld + hs[s]];
#pragma unroll
    for (int k = 0; k < kMaxFast; ++k)
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        al[k][s] = a.attn[(int64_t)(beg + (k < deg ? k : deg - 1)) * a.H + hs[s]];
        ep[k][s] = a.el[(int64_t)u[k] * a.s_ld + hs[s]];
      }
    float ga[kMaxFast][NS];
#pragma unroll
    for (int k = 0; k < kMaxFast; ++k)
#pragma unroll
      for (int s = 0; s < NS; ++s) ga[k][s] = 0.f;
    constexpr int kGather = (R >= 8 ? 1 : R == 4 ? 2 : 4);
#pragma unroll
    for (int k0 = 0; k0 < kMaxFast; k0 += kGather) {
      if (WAVE ? !(k0 < deg) : !__any(k0 < deg)) break;
      float4 x[kGather][R];
#pragma unroll
      for (int q = 0; q < kGather; ++q)
#pragma unroll
        for (int r = 0; r < R; ++r) x[q][r] = ldv(a.ft + (int64_t)u[k0 + q] * a.ft_ld + (r * T + lane) * 4);
#pragma unroll
      for (int q = 0; q < kGather; ++q) {
#pragma unroll
        for (int r = 0; r < R; ++r) ga[k0 + q][SL::of(r)] += dot4(x[q][r], g[r]);
#pragma unroll
        for (int s = 0; s < NS; ++s) ga[k0 + q][s] = uni<UW>(team_sum(ga[k0 + q][s], width));
      }
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      float S = 0.f;
#pragma unroll
      for (int k = 0; k < kMaxFast; ++k) {
        al[k][s] = k < deg ? al[k][s] : 0.f;
        if (a.p > 0.f) ga[k][s] *= keep_scale(a.seed, (int64_t)(beg + (k < deg ? k : deg - 1)) * a.H + hs[s], a.p, a.inv_keep);
        S = k < deg ? fmaf(al[k][s], ga[k][s], S) : S;
      }
      float ger = 0.f;
#pragma unroll
      for (int k = 0; k < kMaxFast; ++k) {
        float ge = al[k][s] * ga[k][s] - al[k][s] * S;
        ge = ep[k][s] + erv[s] > 0.f ? ge : ge * a.slope;
        if (wr[s] && k < deg) a.g_e[(int64_t)(beg + k) * a.H + hs[s]] = ge;
        ger += k < deg ? ge : 0.f;
      }
      if (wr[s]) a.g_er[v * a.gs_ld + hs[s]] = ger;
    }
    return;
  }

  // general degree: pass 1 keeps g_a in g_e (written and re-read by the same writer lane), pass 2 finishes
  float S[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) S[s] = 0.f;
  for (int j = beg; j < end; ++j) {
    const int64_t u = a.indices[j];
    const ST* row = a.ft + u * a.ft_ld;
    float pd[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) pd[s] = 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r) pd[SL::of(r)] += dot4(ldv(row + (r * T + lane) * 4), g[r]);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      float x = team_sum(pd[s], width);
      const int64_t eidx = (int64_t)j * a.H + hs[s];
      if (a.p > 0.f) x *= keep_scale(a.seed, eidx, a.p, a.inv_keep);
      S[s] = fmaf(a.attn[eidx], x, S[s]);
      if (wr[s]) a.g_e[eidx] = x;
    }
  }
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    if (!wr[s]) continue;
    const float erv = a.er[v * a.s_ld + hs[s]];
    float ger = 0.f;
    for (int j = beg; j < end; ++j) {
      const int64_t eidx = (int64_t)j * a.H + hs[s];
      const float al = a.attn[eidx];
      float ge = al * a.g_e[eidx] - al * S[s];
      const float epre = a.el[(int64_t)a.indices[j] * a.s_ld + hs[s]] + erv;
      ge = epre > 0.f ? ge : ge * a.slope;
      a.g_e[eidx] = ge;
      ger += ge;
    }
    a.g_er[v * a.gs_ld + hs[s]] = ger;
  }
}

template <typename ST, int R, int CH> static void launch_small_team(dim3 grid, dim3 block, hipStream_t st, const GatBwdDstT<ST>& a) {
  if constexpr (R <= 4) hipLaunchKernelGGL((gat_bwd_dst_vec<ST, 0, R, CH>), grid, block, 0, st, a);
}

// scalar fallback: one thread per (node, head)
__global__ void gat_bwd_dst_scalar(GatBwdDst a) {
  if (a.seed_off) a.seed += a.seed_off[0];
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= a.N * a.H) return;
  const int64_t v = gid / a.H; const int h = (int)(gid % a.H);
  const int beg = a.indptr[v], end = a.indptr[v + 1];
  const int base = h * a.D;
  const float gscale = a.mean ? 1.f / (float)a.H : 1.f;
  float amx = 0.f;
  for (int d = 0; d < a.D; ++d) {
    float q = reinterpret_cast<const float*>(a.g_out)[v * a.g_out_ld + (a.mean ? d : base + d)] * gscale;
    if (a.act != SPGNN_ACT_NONE) q *= act_bwd_from_out(a.out[v * a.out_ld + base + d], a.act);
    a.g_pre[v * a.g_pre_ld + base + d] = q;
    amx = fmaxf(amx, fabsf(q));
  }
  if (a.absmax) spgnn_detail::slots_max(a.absmax, amx, (unsigned)v);
  float S = 0.f;
  for (int j = beg; j < end; ++j) {
    const int64_t u = a.indices[j];
    float ga = 0.f;
    for (int d = 0; d < a.D; ++d) ga = fmaf(a.ft[u * a.ft_ld + base + d], a.g_pre[v * a.g_pre_ld + base + d], ga);
    const int64_t eidx = (int64_t)j * a.H + h;
    if (a.p > 0.f) ga *= keep_scale(a.seed, eidx, a.p, a.inv_keep);
    S = fmaf(a.attn[eidx], ga, S);
    a.g_e[eidx] = ga;
  }
  const float erv = a.er[v * a.s_ld + h];
  float ger = 0.f;
  for (int j = beg; j < end; ++j) {
    const int64_t eidx = (int64_t)j * a.H + h;
    const float al = a.attn[eidx];
    float ge = al * a.g_e[eidx] - al * S;
    const float epre = a.el[(int64_t)a.indices[j] * a.s_ld + h] + erv;
    ge = epre > 0.f ? ge : ge * a.slope;
    a.g_e[eidx] = ge;
    ger += ge;
  }
  a.g_er[v * a.gs_ld + h] = ger;
}

// -------------------------------------------------------------------------------------------------
// backward, src-major half
// -------------------------------------------------------------------------------------------------
template <typename ST> struct GatBwdSrcT {
  const int32_t* out_indptr; const int32_t* out_indices; const int32_t* out_pos;
  const int32_t* out_nbr8; const int32_t* out_pos8;   // optional (N, 8) rows of out_indices / out_pos, padded like nbr8
  const float* attn; const float* g_e;
  const ST* g_pre; int64_t g_pre_ld;
  ST* g_ft; int64_t g_ft_ld;
  float* g_el; int64_t gs_ld;
  float* absmax;                         // optional scale block: max |g_ft| is folded into its slots
  int64_t N; int H; int D; int T;
  float p; float inv_keep; uint64_t seed;
  const uint64_t* seed_off;
  // optional score term (el / er computed FROM ft, DGL's own form): g_ft[u,h,:] += g_el[u,h] * sc_l[h,:] + g_er[u,h] * sc_r[h,:]
  const float* sc_l; const float* sc_r; const float* g_er;
};
using GatBwdSrc = GatBwdSrcT<float>;

template <typename ST, int TT, int R, int CH>
__global__ __launch_bounds__(kBlock) void gat_bwd_src_vec(GatBwdSrcT<ST> a) {
  if (a.seed_off) a.seed += a.seed_off[0];
  using SL = Slots<R, CH>;
  constexpr int NS = SL::NS;
  constexpr bool WAVE = TT == 64;               // see gat_fwd_vec
  constexpr bool UW = WAVE && CH >= 1;
  const int T = WAVE ? 64 : a.T;
  const int lane = threadIdx.x % T;
  const int64_t u = xcd_block() * (kBlock / T) + uni<WAVE>((int)(threadIdx.x / T));
  if (u >= a.N) return;
  const int beg = uni<WAVE>(a.out_indptr[u]), end = uni<WAVE>(a.out_indptr[u + 1]), deg = end - beg;
  int hs[NS]; bool wr[NS]; float gel[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const int c0 = ((CH == 0 ? s : s * CH) * T + lane) * 4;
    hs[s] = (CH == 0) ? c0 / a.D : s;
    wr[s] = (CH == 0) ? (c0 % a.D == 0) : (lane == 0);
    gel[s] = 0.f;
  }
  float4 acc[R];
#pragma unroll
  for (int r = 0; r < R; ++r) acc[r] = make_float4(0.f, 0.f, 0.f, 0.f);

  if (deg > 0 && deg <= kMaxFast) {
    // unconditional, batched loads (slot k >= deg repeats the last edge with weight 0): see gat_fwd_vec
    int vv[kMaxFast], pp[kMaxFast];
    if (a.out_nbr8) {                    // padded out-neighbour rows: ids and slots depend on u only
#pragma unroll
      for (int k = 0; k < kMaxFast; ++k) {
        vv[k] = uni<WAVE>(a.out_nbr8[u * 8 + k]);
        pp[k] = uni<WAVE>(a.out_pos8[u * 8 + k]);
      }
    } else {
#pragma unroll
      for (int k = 0; k < kMaxFast; ++k) {
        vv[k] = uni<WAVE>(a.out_indices[beg + (k < deg ? k : deg - 1)]);
        pp[k] = uni<WAVE>(a.out_pos[beg + (k < deg ? k : deg - 1)]);
      }
    }
    float w[kMaxFast][NS];
    if constexpr (CH >= 1) {
      // one (edge slot, head) entry per lane (see gat_fwd_vec): attention weight, dropout hash and score gradient are read /
      // formed once per entry, g_el is an 8-lane sum, the NS x 8 weights are broadcast for the row phase
      constexpr int NENT = NS * 8;
      constexpr int NREG = WAVE ? 1 : (NENT + 15) / 16;
      const int wl = threadIdx.x & 63, tbase = wl & ~(T - 1);
      float wv[NREG], gsum[NREG];
#pragma unroll
      for (int i = 0; i < NREG; ++i) {
        const int e = lane + i * T;
        const int k = e & 7, s_ = e >> 3;
        const bool own = e < NENT, valid = own && k < deg;
        const int hh = own ? s_ : 0;
        const int kk = k < deg ? k : deg - 1;
        const int pe = a.out_nbr8 ? a.out_pos8[u * 8 + k] : a.out_pos[beg + kk];
        const int64_t slot = (int64_t)pe * a.H + hh;
        float x = a.attn[slot];
        const float gq = a.g_e[slot];
        if (a.p > 0.f) x *= keep_scale(a.seed, slot, a.p, a.inv_keep);
        wv[i] = valid ? x : 0.f;
        float gs = group8_sum(valid ? gq : 0.f);
        gsum[i] = gs;
      }
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const int e0 = s * 8;                                   // entry (k = 0, head s): its lane holds the head's g_el sum
        if constexpr (WAVE) {
          gel[s] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gsum[0]), e0));
        } else {
          float v0 = __shfl(gsum[0], tbase + (e0 & (T - 1)), 64);
          if constexpr (NREG > 1) { const float v1 = __shfl(gsum[NREG - 1], tbase + (e0 & (T - 1)), 64); v0 = (e0 / T) ? v1 : v0; }
          gel[s] = v0;
        }
#pragma unroll
        for (int k = 0; k < kMaxFast; ++k) {
          const int e = s * 8 + k;
          if constexpr (WAVE) {
            w[k][s] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wv[0]), e));
          } else {
            float v0 = __shfl(wv[0], tbase + (e & (T - 1)), 64);
            if constexpr (NREG > 1) { const float v1 = __shfl(wv[NREG - 1], tbase + (e & (T - 1)), 64); v0 = (e / T) ? v1 : v0; }
            w[k][s] = v0;
          }
        }
      }
    } else {
    float ge[kMaxFast][NS];
#pragma unroll
    for (int k = 0; k < kMaxFast; ++k)
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        w[k][s] = a.attn[(int64_t)pp[k] * a.H + hs[s]];
        ge[k][s] = a.g_e[(int64_t)pp[k] * a.H + hs[s]];
      }
#pragma unroll
    for (int k = 0; k < kMaxFast; ++k)
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        if (a.p > 0.f) w[k][s] *= keep_scale(a.seed, (int64_t)pp[k] * a.H + hs[s], a.p, a.inv_keep);
        w[k][s] = uni<UW>(k < deg ? w[k][s] : 0.f);
        gel[s] += k < deg ? ge[k][s] : 0.f;
      }
    }
    constexpr int kGather = (R >= 8 ? 1 : R == 4 ? 2 : 4);
#pragma unroll
    for (int k0 = 0; k0 < kMaxFast; k0 += kGather) {
      if (WAVE ? !(k0 < deg) : !__any(k0 < deg)) break;
      float4 x[kGather][R];
#pragma unroll
      for (int q = 0; q < kGather; ++q)
#pragma unroll
        for (int r = 0; r < R; ++r) x[q][r] = ldv(a.g_pre + (int64_t)vv[k0 + q] * a.g_pre_ld + (r * T + lane) * 4);
#pragma unroll
      for (int q = 0; q < kGather; ++q)
#pragma unroll
        for (int r = 0; r < R; ++r) fma4(acc[r], w[k0 + q][SL::of(r)], x[q][r]);
    }
  } else {
    for (int k = beg; k < end; ++k) {
      const int64_t v = a.out_indices[k], pos = a.out_pos[k];
      const ST* row = a.g_pre + v * a.g_pre_ld;
      float w[NS];
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const int64_t eidx = pos * a.H + hs[s];
        w[s] = a.attn[eidx];
        if (a.p > 0.f) w[s] *= keep_scale(a.seed, eidx, a.p, a.inv_keep);
        gel[s] += a.g_e[eidx];
      }
#pragma unroll
      for (int r = 0; r < R; ++r) fma4(acc[r], w[SL::of(r)], ldv(row + (r * T + lane) * 4));
    }
  }
  if (a.sc_l) {
    float ger[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) ger[s] = a.g_er[u * a.gs_ld + hs[s]];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int c = (r * T + lane) * 4;
      fma4(acc[r], gel[SL::of(r)], ld4(a.sc_l + c));
      fma4(acc[r], ger[SL::of(r)], ld4(a.sc_r + c));
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) stv(a.g_ft + u * a.g_ft_ld + (r * T + lane) * 4, acc[r]);
#pragma unroll
  for (int s = 0; s < NS; ++s)
    if (wr[s]) a.g_el[u * a.gs_ld + hs[s]] = gel[s];
  if (a.absmax) {
    float mx = 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r) mx = absmax4(mx, acc[r]);
    mx = team_max(mx, T);
    if (lane == 0) spgnn_detail::slots_max(a.absmax, mx, (unsigned)u);
  }
}

template <typename ST, int R, int CH> static void launch_small_team(dim3 grid, dim3 block, hipStream_t st, const GatBwdSrcT<ST>& a) {
  if constexpr (R <= 4) hipLaunchKernelGGL((gat_bwd_src_vec<ST, 0, R, CH>), grid, block, 0, st, a);
}

__global__ void gat_bwd_src_scalar(GatBwdSrc a) {
  if (a.seed_off) a.seed += a.seed_off[0];
  const int HD = a.H * a.D;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= a.N * HD) return;
  const int64_t u = gid / HD; const int col = (int)(gid % HD); const int h = col / a.D;
  const int beg = a.out_indptr[u], end = a.out_indptr[u + 1];
  float acc = 0.f, gel = 0.f;
  for (int k = beg; k < end; ++k) {
    const int64_t v = a.out_indices[k];
    const int64_t eidx = (int64_t)a.out_pos[k] * a.H + h;
    float w = a.attn[eidx];
    if (a.p > 0.f) w *= keep_scale(a.seed, eidx, a.p, a.inv_keep);
    acc = fmaf(w, a.g_pre[v * a.g_pre_ld + col], acc);
    gel += a.g_e[eidx];
  }
  if (a.sc_l) acc = fmaf(gel, a.sc_l[col], fmaf(a.g_er[u * a.gs_ld + h], a.sc_r[col], acc));
  a.g_ft[u * a.g_ft_ld + col] = acc;
  if (col % a.D == 0) a.g_el[u * a.gs_ld + h] = gel;
  if (a.absmax) spgnn_detail::slots_max(a.absmax, fabsf(acc), (unsigned)u);
}

// =================================================================================================
// Aggregate-first GAT (layers whose input is narrower than one head's output, e.g. the 192 -> 2 x 1024 output
// layer of st_pgat_spgnn_3).  sum_u alpha_h(u,v) (W_h x_u) = W_h (sum_u alpha_h(u,v) x_u): the attention-weighted
// sums run over the F-wide INPUT rows, once per head, and the projection follows as a GEMM on [z_h | x] (the
// residual projection rides in the same product).  The H*D-wide projected rows are then never gathered.
// One wave per node; lane l owns float4 chunks l, l+64, ... of the F-wide row (R chunks, F <= 256*R); lanes
// past the row end re-read chunk 0 and are masked at the stores and in the reductions.
//   z      (N, H*zs): head h's block starts at column h*zs; [0,F) = z_h, [xoff, xoff+F) = copy of x (xoff < 0: none)
// =================================================================================================
// The three aggregate-first kernels take a team of T lanes per node: T = 64 (one node per wave, wave-uniform scalars) for
// wide inputs, T = 16 (four nodes per wave, R <= 3 float4 per lane) for F <= 192: these kernels are latency-bound per node
// (index -> score -> row chains), so four nodes in flight per wave is what fills the memory pipe.
template <typename ST> struct GatAggFwdT {
  const int32_t* indptr; const int32_t* indices;
  const ST* x; int64_t x_ld;
  const float* el; const float* er; int64_t s_ld;
  float* attn;
  ST* z; int64_t z_ld; int zs; int xoff;
  float* absmax;                         // optional scale block: max |z| is folded into its slots
  int64_t N; int F;
  float slope; float p; float inv_keep; uint64_t seed; const uint64_t* seed_off;
};

template <typename ST, int H, int R, int T>
__global__ __launch_bounds__(kBlock) void gat_agg_fwd(GatAggFwdT<ST> a) {
  if (a.seed_off) a.seed += a.seed_off[0];
  const int lane = threadIdx.x % T;
  const int64_t v = xcd_block() * (kBlock / T) + uni<T == 64>((int)(threadIdx.x / T));
  if (v >= a.N) return;
  const int beg = uni<T == 64>(a.indptr[v]), end = uni<T == 64>(a.indptr[v + 1]), deg = end - beg;
  int col[R]; bool ok[R];
#pragma unroll
  for (int r = 0; r < R; ++r) { const int c = (r * T + lane) * 4; ok[r] = c < a.F; col[r] = ok[r] ? c : 0; }
  float erv[H];
#pragma unroll
  for (int h = 0; h < H; ++h) erv[h] = a.er[v * a.s_ld + h];
  float4 xs[R];                                     // the node's own row (residual operand)
#pragma unroll
  for (int r = 0; r < R; ++r) xs[r] = ldv(a.x + v * a.x_ld + col[r]);
  float4 acc[H][R];
#pragma unroll
  for (int h = 0; h < H; ++h)
#pragma unroll
    for (int r = 0; r < R; ++r) acc[h][r] = make_float4(0.f, 0.f, 0.f, 0.f);

  if (deg > 0 && deg <= kMaxFast) {
    int u[kMaxFast];
#pragma unroll
    for (int k = 0; k < kMaxFast; ++k) u[k] = uni<T == 64>(a.indices[beg + (k < deg ? k : deg - 1)]);
    float w[kMaxFast][H];
#pragma unroll
    for (int k = 0; k < kMaxFast; ++k)
#pragma unroll
      for (int h = 0; h < H; ++h) w[k][h] = a.el[(int64_t)u[k] * a.s_ld + h];
#pragma unroll
    for (int h = 0; h < H; ++h) {
      float mx = -INFINITY;
#pragma unroll
      for (int k = 0; k < kMaxFast; ++k) {
        w[k][h] = k < deg ? lrelu(w[k][h] + erv[h], a.slope) : -INFINITY;
        mx = fmaxf(mx, w[k][h]);
      }
      float sm = 0.f;
#pragma unroll
      for (int k = 0; k < kMaxFast; ++k) {
        w[k][h] = k < deg ? expf(w[k][h] - mx) : 0.f;
        sm += w[k][h];
      }
#pragma unroll
      for (int k = 0; k < kMaxFast; ++k) {
        const float al = w[k][h] / sm;
        if (lane == 0 && k < deg) a.attn[(int64_t)(beg + k) * H + h] = al;
        w[k][h] = al;
      }
    }
    if (a.p > 0.f) {
#pragma unroll
      for (int k = 0; k < kMaxFast; ++k)
#pragma unroll
        for (int h = 0; h < H; ++h)
          w[k][h] *= keep_scale(a.seed, (int64_t)(beg + (k < deg ? k : deg - 1)) * H + h, a.p, a.inv_keep);
    }
#pragma unroll
    for (int k = 0; k < kMaxFast; ++k)
#pragma unroll
      for (int h = 0; h < H; ++h) w[k][h] = uni<T == 64>(w[k][h]);
    constexpr int G = R >= 4 ? 2 : 4;
#pragma unroll
    for (int k0 = 0; k0 < kMaxFast; k0 += G) {
      if (!(k0 < deg)) break;
      float4 xr[G][R];
#pragma unroll
      for (int q = 0; q < G; ++q)
#pragma unroll
        for (int r = 0; r < R; ++r) xr[q][r] = ldv(a.x + (int64_t)u[k0 + q] * a.x_ld + col[r]);
#pragma unroll
      for (int q = 0; q < G; ++q)
#pragma unroll
        for (int h = 0; h < H; ++h)
#pragma unroll
          for (int r = 0; r < R; ++r) fma4(acc[h][r], w[k0 + q][h], xr[q][r]);
    }
  } else {
    float mx[H], sm[H];
#pragma unroll
    for (int h = 0; h < H; ++h) {
      float m_ = -INFINITY;
      for (int j = beg; j < end; ++j) m_ = fmaxf(m_, lrelu(a.el[(int64_t)a.indices[j] * a.s_ld + h] + erv[h], a.slope));
      float s_ = 0.f;
      for (int j = beg; j < end; ++j) s_ += expf(lrelu(a.el[(int64_t)a.indices[j] * a.s_ld + h] + erv[h], a.slope) - m_);
      mx[h] = m_; sm[h] = s_;
    }
    for (int j = beg; j < end; ++j) {
      const int64_t u = a.indices[j];
      float w[H];
#pragma unroll
      for (int h = 0; h < H; ++h) {
        const float al = expf(lrelu(a.el[u * a.s_ld + h] + erv[h], a.slope) - mx[h]) / sm[h];
        const int64_t eidx = (int64_t)j * H + h;
        if (lane == 0) a.attn[eidx] = al;
        w[h] = a.p > 0.f ? al * keep_scale(a.seed, eidx, a.p, a.inv_keep) : al;
      }
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const float4 xr = ldv(a.x + u * a.x_ld + col[r]);
#pragma unroll
        for (int h = 0; h < H; ++h) fma4(acc[h][r], w[h], xr);
      }
    }
  }
  float mxv = 0.f;
#pragma unroll
  for (int h = 0; h < H; ++h)
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (!ok[r]) continue;
      ST* dst = a.z + v * a.z_ld + (int64_t)h * a.zs + col[r];
      stv(dst, acc[h][r]);
      mxv = absmax4(mxv, acc[h][r]);
      if (a.xoff >= 0) { stv(dst + a.xoff, xs[r]); mxv = absmax4(mxv, xs[r]); }
      else if (a.xoff == -2 && h == H - 1) { stv(dst + a.zs, xs[r]); mxv = absmax4(mxv, xs[r]); }   // ONE copy of x behind the last block
    }
  if (a.absmax) {
    mxv = team_max(mxv, T);
    if (lane == 0) spgnn_detail::slots_max(a.absmax, mxv, (unsigned)v);
  }
}

template <typename ST> struct GatAggBwdDstT {
  const int32_t* indptr; const int32_t* indices;
  const ST* x; int64_t x_ld;
  const float* el; const float* er; int64_t s_ld;
  const float* attn;
  const ST* gz; int64_t gz_ld; int zs;             // gradient of the z blocks (same layout as z)
  float* g_e; float* g_er; int64_t gs_ld;
  int64_t N; int F;
  float slope; float p; float inv_keep; uint64_t seed; const uint64_t* seed_off;
  const int32_t* inv;                              // nullable (spgnn_gat_agg_bwd_dst_rows): gz holds one row per LISTED node; node v's
};                                                 // row is gz[inv[v]], or all zeros when inv[v] < 0 (its g_e, g_er are then zero)

template <typename ST, int H, int R, int T>
__global__ __launch_bounds__(kBlock) void gat_agg_bwd_dst(GatAggBwdDstT<ST> a) {
  if (a.seed_off) a.seed += a.seed_off[0];
  const int lane = threadIdx.x % T;
  const int64_t v = xcd_block() * (kBlock / T) + uni<T == 64>((int)(threadIdx.x / T));
  if (v >= a.N) return;
  const int beg = uni<T == 64>(a.indptr[v]), end = uni<T == 64>(a.indptr[v + 1]), deg = end - beg;
  int64_t gv_row = v;
  if (a.inv) {
    gv_row = uni<T == 64>(a.inv[v]);
    if (gv_row < 0) {                               // a zero gradient row: every dot product, g_e and g_er of this node is zero
      if (lane == 0) {
        for (int j = beg; j < end; ++j)
#pragma unroll
          for (int h = 0; h < H; ++h) a.g_e[(int64_t)j * H + h] = 0.f;
#pragma unroll
        for (int h = 0; h < H; ++h) a.g_er[v * a.gs_ld + h] = 0.f;
      }
      return;
    }
  }
  int col[R]; bool ok[R];
#pragma unroll
  for (int r = 0; r < R; ++r) { const int c = (r * T + lane) * 4; ok[r] = c < a.F; col[r] = ok[r] ? c : 0; }
  float4 g[H][R];
#pragma unroll
  for (int h = 0; h < H; ++h)
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const float4 q = ldv(a.gz + gv_row * a.gz_ld + (int64_t)h * a.zs + col[r]);
      g[h][r] = ok[r] ? q : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  float erv[H];
#pragma unroll
  for (int h = 0; h < H; ++h) erv[h] = a.er[v * a.s_ld + h];

  if (deg > 0 && deg <= kMaxFast) {
    int u[kMaxFast];
#pragma unroll
    for (int k = 0; k < kMaxFast; ++k) u[k] = uni<T == 64>(a.indices[beg + (k < deg ? k : deg - 1)]);
    float al[kMaxFast][H], ep[kMaxFast][H], ga[kMaxFast][H];
#pragma unroll
    for (int k = 0; k < kMaxFast; ++k)
#pragma unroll
      for (int h = 0; h < H; ++h) {
        al[k][h] = a.attn[(int64_t)(beg + (k < deg ? k : deg - 1)) * H + h];
        ep[k][h] = a.el[(int64_t)u[k] * a.s_ld + h];
        ga[k][h] = 0.f;
      }
    constexpr int G = R >= 4 ? 2 : 4;
#pragma unroll
    for (int k0 = 0; k0 < kMaxFast; k0 += G) {
      if (!(k0 < deg)) break;
      float4 xr[G][R];
#pragma unroll
      for (int q = 0; q < G; ++q)
#pragma unroll
        for (int r = 0; r < R; ++r) xr[q][r] = ldv(a.x + (int64_t)u[k0 + q] * a.x_ld + col[r]);
#pragma unroll
      for (int q = 0; q < G; ++q)
#pragma unroll
        for (int h = 0; h < H; ++h) {
          float pd = 0.f;
#pragma unroll
          for (int r = 0; r < R; ++r) pd += dot4(xr[q][r], g[h][r]);
          ga[k0 + q][h] = uni<T == 64>(team_sum(pd, T));
        }
    }
#pragma unroll
    for (int h = 0; h < H; ++h) {
      float S = 0.f;
#pragma unroll
      for (int k = 0; k < kMaxFast; ++k) {
        al[k][h] = k < deg ? al[k][h] : 0.f;
        if (a.p > 0.f) ga[k][h] *= keep_scale(a.seed, (int64_t)(beg + (k < deg ? k : deg - 1)) * H + h, a.p, a.inv_keep);
        S = k < deg ? fmaf(al[k][h], ga[k][h], S) : S;
      }
      float ger = 0.f;
#pragma unroll
      for (int k = 0; k < kMaxFast; ++k) {
        float ge = al[k][h] * ga[k][h] - al[k][h] * S;
        ge = ep[k][h] + erv[h] > 0.f ? ge : ge * a.slope;
        if (lane == 0 && k < deg) a.g_e[(int64_t)(beg + k) * H + h] = ge;
        ger += k < deg ? ge : 0.f;
      }
      if (lane == 0) a.g_er[v * a.gs_ld + h] = ger;
    }
    return;
  }
  // general degree: pass 1 parks g_alpha in g_e (written and re-read by lane 0), pass 2 finishes
  float S[H];
#pragma unroll
  for (int h = 0; h < H; ++h) S[h] = 0.f;
  for (int j = beg; j < end; ++j) {
    const int64_t u = a.indices[j];
    float4 xr[R];
#pragma unroll
    for (int r = 0; r < R; ++r) xr[r] = ldv(a.x + u * a.x_ld + col[r]);
#pragma unroll
    for (int h = 0; h < H; ++h) {
      float pd = 0.f;
#pragma unroll
      for (int r = 0; r < R; ++r) pd += dot4(xr[r], g[h][r]);
      float xg = team_sum(pd, T);
      const int64_t eidx = (int64_t)j * H + h;
      if (a.p > 0.f) xg *= keep_scale(a.seed, eidx, a.p, a.inv_keep);
      S[h] = fmaf(a.attn[eidx], xg, S[h]);
      if (lane == 0) a.g_e[eidx] = xg;
    }
  }
  if (lane != 0) return;
#pragma unroll
  for (int h = 0; h < H; ++h) {
    float ger = 0.f;
    for (int j = beg; j < end; ++j) {
      const int64_t eidx = (int64_t)j * H + h;
      const float al = a.attn[eidx];
      float ge = al * a.g_e[eidx] - al * S[h];
      ge = a.el[(int64_t)a.indices[j] * a.s_ld + h] + erv[h] > 0.f ? ge : ge * a.slope;
      a.g_e[eidx] = ge;
      ger += ge;
    }
    a.g_er[v * a.gs_ld + h] = ger;
  }
}

template <typename ST> struct GatAggBwdSrcT {
  const int32_t* out_indptr; const int32_t* out_indices; const int32_t* out_pos;
  const float* attn; const float* g_e;
  const ST* gz; int64_t gz_ld; int zs; int xoff;
  const float* g_er;                               // (N, H) at stride gs_ld, written by the dst-major half
  const float* w_lr; int64_t wlr_ld;               // (2H, F): g_x += [g_el | g_er] @ w_lr fused here
  ST* g_x; int64_t gx_ld;
  float* g_el; int64_t gs_ld;
  int64_t N; int F;
  float p; float inv_keep; uint64_t seed; const uint64_t* seed_off;
  const int32_t* inv;                              // nullable (spgnn_gat_agg_bwd_src_rows): as in GatAggBwdDstT
};

template <typename ST, int H, int R, int T>
__global__ __launch_bounds__(kBlock) void gat_agg_bwd_src(GatAggBwdSrcT<ST> a) {
  if (a.seed_off) a.seed += a.seed_off[0];
  const int lane = threadIdx.x % T;
  const int64_t u = xcd_block() * (kBlock / T) + uni<T == 64>((int)(threadIdx.x / T));
  if (u >= a.N) return;
  const int beg = uni<T == 64>(a.out_indptr[u]), end = uni<T == 64>(a.out_indptr[u + 1]), deg = end - beg;
  int col[R]; bool ok[R];
#pragma unroll
  for (int r = 0; r < R; ++r) { const int c = (r * T + lane) * 4; ok[r] = c < a.F; col[r] = ok[r] ? c : 0; }
  float4 acc[R];
#pragma unroll
  for (int r = 0; r < R; ++r) acc[r] = make_float4(0.f, 0.f, 0.f, 0.f);
  const int64_t gu_row = a.inv ? (int64_t)uni<T == 64>(a.inv[u]) : u;
  if (a.xoff >= 0 && gu_row >= 0) {                 // gradient of the residual operand copies
#pragma unroll
    for (int h = 0; h < H; ++h)
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const float4 q = ldv(a.gz + gu_row * a.gz_ld + (int64_t)h * a.zs + a.xoff + col[r]);
        acc[r].x += q.x; acc[r].y += q.y; acc[r].z += q.z; acc[r].w += q.w;
      }
  }
  float gel[H];
#pragma unroll
  for (int h = 0; h < H; ++h) gel[h] = 0.f;
  if (deg > 0 && deg <= kMaxFast) {
    int vv[kMaxFast], pp[kMaxFast];
#pragma unroll
    for (int k = 0; k < kMaxFast; ++k) {
      vv[k] = uni<T == 64>(a.out_indices[beg + (k < deg ? k : deg - 1)]);
      pp[k] = uni<T == 64>(a.out_pos[beg + (k < deg ? k : deg - 1)]);
    }
    if (a.inv) {                                    // the destinations' rows in the list (-1: a zero row, skipped below)
#pragma unroll
      for (int k = 0; k < kMaxFast; ++k) vv[k] = uni<T == 64>(a.inv[vv[k]]);
    }
    float w[kMaxFast][H];
#pragma unroll
    for (int k = 0; k < kMaxFast; ++k)
#pragma unroll
      for (int h = 0; h < H; ++h) {
        w[k][h] = a.attn[(int64_t)pp[k] * H + h];
        const float ge = a.g_e[(int64_t)pp[k] * H + h];
        if (a.p > 0.f) w[k][h] *= keep_scale(a.seed, (int64_t)pp[k] * H + h, a.p, a.inv_keep);
        w[k][h] = uni<T == 64>(k < deg ? w[k][h] : 0.f);
        gel[h] += k < deg ? ge : 0.f;
      }
    constexpr int G = H * R >= 8 ? 1 : H * R >= 4 ? 2 : 4;
#pragma unroll
    for (int k0 = 0; k0 < kMaxFast; k0 += G) {
      if (!(k0 < deg)) break;
      float4 gr[G][H][R];
#pragma unroll
      for (int q = 0; q < G; ++q)
#pragma unroll
        for (int h = 0; h < H; ++h)
#pragma unroll
          for (int r = 0; r < R; ++r)
            gr[q][h][r] = vv[k0 + q] >= 0 ? ldv(a.gz + (int64_t)vv[k0 + q] * a.gz_ld + (int64_t)h * a.zs + col[r])
                                          : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int q = 0; q < G; ++q)
#pragma unroll
        for (int h = 0; h < H; ++h)
#pragma unroll
          for (int r = 0; r < R; ++r) fma4(acc[r], w[k0 + q][h], gr[q][h][r]);
    }
  } else {
    for (int k = beg; k < end; ++k) {
      const int64_t pos = a.out_pos[k];
      const int64_t v = a.inv ? (int64_t)a.inv[a.out_indices[k]] : (int64_t)a.out_indices[k];
#pragma unroll
      for (int h = 0; h < H; ++h) {
        const int64_t eidx = pos * H + h;
        float w = a.attn[eidx];
        if (a.p > 0.f) w *= keep_scale(a.seed, eidx, a.p, a.inv_keep);
        gel[h] += a.g_e[eidx];
        if (v >= 0) {
#pragma unroll
          for (int r = 0; r < R; ++r) fma4(acc[r], w, ldv(a.gz + v * a.gz_ld + (int64_t)h * a.zs + col[r]));
        }
      }
    }
  }
  if (lane == 0) {
#pragma unroll
    for (int h = 0; h < H; ++h) a.g_el[u * a.gs_ld + h] = gel[h];
  }
  if (a.w_lr) {                                     // score-projection backward: g_x += g_el @ w_lr[:H] + g_er @ w_lr[H:]
#pragma unroll
    for (int h = 0; h < H; ++h) {
      const float ger = a.g_er[u * a.gs_ld + h];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        fma4(acc[r], gel[h], ld4(a.w_lr + (int64_t)h * a.wlr_ld + col[r]));
        fma4(acc[r], ger, ld4(a.w_lr + (int64_t)(H + h) * a.wlr_ld + col[r]));
      }
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r)
    if (ok[r]) stv(a.g_x + u * a.gx_ld + col[r], acc[r]);
}

// out_mean[v, d] = mean_h out[v, h*D + d]  (vector form of head_mean_scalar; D % 4 == 0, 16-byte rows)
__global__ __launch_bounds__(kBlock) void head_mean_vec(const float* __restrict__ out, int64_t out_ld, float* __restrict__ om,
                                                        int64_t om_ld, int64_t N, int H, int D) {
  const int d4 = D >> 2;
  const float inv_h = 1.f / (float)H;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N * d4; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t v = i / d4; const int c = (int)(i % d4) * 4;
    float4 m = ld4(out + v * out_ld + c);
    for (int h = 1; h < H; ++h) { const float4 q = ld4(out + v * out_ld + (int64_t)h * D + c); m.x += q.x; m.y += q.y; m.z += q.z; m.w += q.w; }
    m.x *= inv_h; m.y *= inv_h; m.z *= inv_h; m.w *= inv_h;
    st4(om + v * om_ld + c, m);
  }
}

// g_pre[v, c] = g[v, mean ? c % D : c] * (mean ? 1/H : 1) * act'(out[v, c]);  absmax[v] = max_c |g_pre[v, c]|
// (the dst-major backward's first phase as a kernel of its own, for layers whose projection follows the aggregation)
__global__ __launch_bounds__(kBlock) void act_bwd_kernel(const float* __restrict__ g, int64_t g_ld, int mean,
                                                         const float* __restrict__ out, int64_t out_ld,
                                                         float* __restrict__ g_pre, int64_t gp_ld, float* __restrict__ absmax,
                                                         int64_t N, int H, int D, int act) {
  const int lane = threadIdx.x & 63;
  const int64_t v = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (v >= N) return;
  const float gscale = mean ? 1.f / (float)H : 1.f;
  float mx = 0.f;
  for (int c = lane * 4; c < D; c += 256) {             // the mean's gradient row is read once and shared by the heads
    float4 gm = make_float4(0.f, 0.f, 0.f, 0.f);
    if (mean) { gm = ld4(g + v * g_ld + c); gm.x *= gscale; gm.y *= gscale; gm.z *= gscale; gm.w *= gscale; }
    for (int h = 0; h < H; ++h) {
      const int idx = h * D + c;
      float4 q = mean ? gm : ld4(g + v * g_ld + idx);
      if (act != SPGNN_ACT_NONE) {
        const float4 o = ld4(out + v * out_ld + idx);
        q.x *= act_bwd_from_out(o.x, act); q.y *= act_bwd_from_out(o.y, act);
        q.z *= act_bwd_from_out(o.z, act); q.w *= act_bwd_from_out(o.w, act);
      }
      st4(g_pre + v * gp_ld + idx, q);
      mx = absmax4(mx, q);
    }
  }
  if (absmax) {
    mx = team_max(mx, 64);
    if (lane == 0) spgnn_detail::slots_max(absmax, mx, (unsigned)v);
  }
}

// The same without the head mean, as a flat pass over 16-byte chunks (one wave per node left 3/4 of the lanes idle on rows
// of 64 columns and made the kernel as slow for (N, 64) as for (N, 1024): 100 us): grid-stride over N * W / 4 chunks, the
// wave's maximum folded into the scale block once per wave.
// drop_p > 0: `g` is the gradient of dropout(act(pre)) under spgnn_cat_dropout's mask (seed, row * W + column): the mask is
// regenerated and applied first - the dropout's backward pass and the activation's in one (the GIN MLP's Linear, Dropout,
// LeakyReLU: reference models.py:236-246).
__global__ __launch_bounds__(kBlock) void act_bwd_flat_kernel(const float* __restrict__ g, int64_t g_ld, const float* __restrict__ out,
                                                              int64_t out_ld, float* __restrict__ g_pre, int64_t gp_ld,
                                                              float* __restrict__ absmax, int64_t N, int W, int act, float drop_p,
                                                              uint64_t seed, const uint64_t* __restrict__ seed_off, float o_scale) {
  // o_scale: 1, or 1 - p when `out` holds the DROPPED activation output (kept elements were multiplied by 1 / (1 - p); the
  // derivative of a dropped element is irrelevant: its mask is 0)
  const int w4 = W >> 2;
  const int64_t total = N * w4;
  float mx = 0.f;
  if (seed_off) seed += seed_off[0];
  const float inv_keep = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
    const int64_t v = i / w4; const int c = (int)(i - v * w4) * 4;
    float4 q = ld4(g + v * g_ld + c);
    if (drop_p > 0.f) {
      const float4 k = feat_keep4(seed, v * W + c, drop_p, inv_keep);
      q.x *= k.x; q.y *= k.y; q.z *= k.z; q.w *= k.w;
    }
    if (act != SPGNN_ACT_NONE) {
      const float4 o = ld4(out + v * out_ld + c);
      q.x *= act_bwd_from_out(o.x * o_scale, act); q.y *= act_bwd_from_out(o.y * o_scale, act);
      q.z *= act_bwd_from_out(o.z * o_scale, act); q.w *= act_bwd_from_out(o.w * o_scale, act);
    }
    st4(g_pre + v * gp_ld + c, q);
    mx = absmax4(mx, q);
  }
  if (absmax) {                                       // (every lane of the wave is here: no early return above)
    mx = team_max(mx, 64);
    if ((threadIdx.x & 63) == 0) spgnn_detail::slots_max(absmax, mx, blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6));
  }
}

// The same pass that also leaves the COLUMN SUMS of g_pre (the bias gradient of a layer whose bias sits in the aggregation's
// epilogue: GraphConv, reference models.py:172-182, and GINConv's first Linear applied before the aggregation) as per-block
// partials colpart[block][W], summed afterwards in block order (spgnn_sum_partials) - deterministic, and no pass of a
// reduction kernel over g_pre.  W / 4 must divide the block size: a thread then keeps ONE column group for all its rows
// (grid stride = a multiple of W / 4), accumulates it in registers, and the block folds its 256 / (W / 4) row lanes through LDS
// in a fixed order.  Two rows per trip (independent loads) since the grid is capped at spgnn_act_bwd_colsum_blocks().
__global__ __launch_bounds__(kBlock) void act_bwd_colsum_kernel(const float* __restrict__ g, int64_t g_ld, const float* __restrict__ out,
                                                                int64_t out_ld, float* __restrict__ g_pre, int64_t gp_ld,
                                                                float* __restrict__ absmax, float* __restrict__ colpart, int64_t N,
                                                                int W, int act, float drop_p, uint64_t seed,
                                                                const uint64_t* __restrict__ seed_off,
                                                                const float* __restrict__ dot_x, int64_t dx_ld) {
  // dot_x (nullable): also sum_v <g_pre[v, :], dot_x[v, :]> (GINConv's eps gradient, reference models.py:358-383, with dot_x
  // = the aggregation's input rows) - the block's share goes into float W of its partial row, whose pitch is then W + 4
  __shared__ float4 red[kBlock];
  __shared__ float dred[kBlock / 64];
  const int pitch = dot_x ? W + 4 : W;
  float dsum = 0.f;
  const int w4 = W >> 2;
  const int rows_per_block = kBlock / w4;                       // w4 divides kBlock (host check)
  const int c = (threadIdx.x % w4) * 4;
  const int64_t rstep = (int64_t)gridDim.x * rows_per_block;
  float mx = 0.f;
  float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
  if (seed_off) seed += seed_off[0];
  const float inv_keep = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
  auto one = [&](int64_t v, float4 q, float4 o) __attribute__((always_inline)) {
    if (drop_p > 0.f) {
      const float4 k = feat_keep4(seed, v * W + c, drop_p, inv_keep);
      q.x *= k.x; q.y *= k.y; q.z *= k.z; q.w *= k.w;
    }
    if (act != SPGNN_ACT_NONE) {
      q.x *= act_bwd_from_out(o.x, act); q.y *= act_bwd_from_out(o.y, act);
      q.z *= act_bwd_from_out(o.z, act); q.w *= act_bwd_from_out(o.w, act);
    }
    st4(g_pre + v * gp_ld + c, q);
    mx = absmax4(mx, q);
    cs.x += q.x; cs.y += q.y; cs.z += q.z; cs.w += q.w;
    if (dot_x) { const float4 u = ld4(dot_x + v * dx_ld + c); dsum += (q.x * u.x + q.y * u.y) + (q.z * u.z + q.w * u.w); }
  };
  int64_t v = (int64_t)blockIdx.x * rows_per_block + threadIdx.x / w4;
  for (; v + rstep < N; v += 2 * rstep) {
    const float4 q0 = ld4(g + v * g_ld + c), q1 = ld4(g + (v + rstep) * g_ld + c);
    float4 o0 = make_float4(0.f, 0.f, 0.f, 0.f), o1 = o0;
    if (act != SPGNN_ACT_NONE) { o0 = ld4(out + v * out_ld + c); o1 = ld4(out + (v + rstep) * out_ld + c); }
    one(v, q0, o0);
    one(v + rstep, q1, o1);
  }
  if (v < N) {
    const float4 q0 = ld4(g + v * g_ld + c);
    float4 o0 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (act != SPGNN_ACT_NONE) o0 = ld4(out + v * out_ld + c);
    one(v, q0, o0);
  }
  red[threadIdx.x] = cs;
  __syncthreads();
  if ((int)threadIdx.x < w4) {
    float4 t = red[threadIdx.x];
    for (int r = 1; r < rows_per_block; ++r) { const float4 u = red[r * w4 + threadIdx.x]; t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
    st4(colpart + (int64_t)blockIdx.x * pitch + c, t);
  }
  if (dot_x) {                                              // wave sums (fixed butterfly), then the four waves in order
    dsum = team_sum(dsum, 64);
    if ((threadIdx.x & 63) == 0) dred[threadIdx.x >> 6] = dsum;
    __syncthreads();
    if (threadIdx.x == 0) {
      float d = dred[0];
      for (int w = 1; w < kBlock / 64; ++w) d += dred[w];
      st4(colpart + (int64_t)blockIdx.x * pitch + W, make_float4(d, 0.f, 0.f, 0.f));
    }
  }
  if (absmax) {
    mx = team_max(mx, 64);
    if ((threadIdx.x & 63) == 0) spgnn_detail::slots_max(absmax, mx, blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6));
  }
}

// act_bwd with the classifier's input gradient formed on the fly (mean-over-heads output layer followed by a skinny
// Linear, reference models.py:1125 `gnn_out`):
//   g_pre[v, h*D + c] = (1/H) * (sum_j gS[v, j] * W[j, c]) * act'(out[v, h*D + c])
// i.e. spgnn_scores_bwd_x (g_mean = g_logits W, written) + spgnn_act_bwd (g_mean re-read) in one pass: g_mean never
// exists in memory.  A thread owns four columns and keeps their W entries in registers (J float4); a block walks a row
// range, two rows per trip with all loads issued first; gS rows are wave-uniform (scalar loads); one |max| per block.
// WG (ABI 55): the pass ALSO forms the skinny Linear's WEIGHT gradient g_W[j, c] = sum_v gS[v, j] * mean_h out[v, h*D + c] -
// the classifier's (reference gnn_out, models.py:1125) - from the rows it holds anyway: a thread owns its four columns for
// every row of the block's range, so the J x 4 accumulators need no cross-lane step; per-block partials
// wgrad[block][j][D], summed by the caller in block order.  Replaces a second pass over the (N, D) head mean
// (spgnn_scores_bwd_w: 313 MB read at 512 trees).
template <int JP, int HT, int RB, bool WG>       // HT: heads at compile time (0: run-time H <= 4); RB rows per trip
__global__ __launch_bounds__(256) void act_bwd_proj_kernel(const float* __restrict__ gS, int64_t ldg, int J,
                                                           const float* __restrict__ W, int64_t ldw,
                                                           const float* __restrict__ out, int64_t out_ld,
                                                           float* __restrict__ g_pre, int64_t gp_ld, float* __restrict__ absmax,
                                                           int64_t N, int64_t rows_per_block, int Hrt, int D, int act,
                                                           float* __restrict__ wgrad, const int32_t* __restrict__ row_list,
                                                           const int32_t* __restrict__ rows_cnt) {
  // `row_list` (spgnn_act_bwd_proj_rows): g_pre has one row per LISTED node (N = the list's capacity) - row q is formed from
  // rows row_list[q] of gS and out when q < rows_cnt[0], and is zero otherwise
  __shared__ float red[4];
  constexpr int HMAX = HT ? HT : 4;
  const int H = HT ? HT : Hrt;
  const int c = threadIdx.x * 4;
  const bool cv = c < D;
  const int cc = cv ? c : 0;
  float4 w[JP];
  float4 gw[WG ? JP : 1];
#pragma unroll
  for (int j = 0; j < JP; ++j) {
    const float4 q = ld4(W + (int64_t)(j < J ? j : 0) * ldw + cc);
    w[j] = j < J ? q : make_float4(0.f, 0.f, 0.f, 0.f);
    if (WG) gw[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const float inv_h = 1.f / (float)H;
  const int64_t n0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t n1 = n0 + rows_per_block < N ? n0 + rows_per_block : N;
  float mx = 0.f;
  // the row of gS is fetched by ONE vector load (lane j holds gS[row, j]) issued with the row's other loads and
  // broadcast by v_readlane: scalar loads would each expose their latency (and spilled 88 SGPRs with two rows in flight)
  const int jl = (threadIdx.x & 63) < J ? (threadIdx.x & 63) : 0;
#define SPGNN_ABP_GM(GM, GV, EM)                                                                             \
  {                                                                                                          \
    GM = make_float4(0.f, 0.f, 0.f, 0.f);                                                                    \
    _Pragma("unroll") for (int j = 0; j < JP; ++j)                                                           \
      if (j < J) {                                                                                           \
        const float gj = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(GV), j));                   \
        fma4(GM, gj, w[j]);                                                                                  \
        if (WG) fma4(gw[j], gj, EM);                                                                         \
      }                                                                                                      \
    GM.x *= inv_h; GM.y *= inv_h; GM.z *= inv_h; GM.w *= inv_h;                                              \
  }
#define SPGNN_ABP_OUT(GM, O, ROW, HH)                                                                        \
  {                                                                                                          \
    float4 q = GM;                                                                                           \
    if (act != SPGNN_ACT_NONE) {                                                                             \
      q.x *= act_bwd_from_out(O.x, act); q.y *= act_bwd_from_out(O.y, act);                                  \
      q.z *= act_bwd_from_out(O.z, act); q.w *= act_bwd_from_out(O.w, act);                                  \
    }                                                                                                        \
    if (cv) { st4(g_pre + (ROW) * gp_ld + (int64_t)(HH) * D + c, q); mx = absmax4(mx, q); }                  \
  }
  const int64_t listed = row_list ? (int64_t)rows_cnt[0] : 0;
  const bool overflowed = row_list && rows_cnt[1] != 0;      // the list lost rows: poison the gradient instead of dropping them
  int64_t n = n0;
  for (; n + RB <= n1; n += RB) {      // RB rows per trip, every load of the trip issued before the first use
    float4 o[RB][HMAX];
    float gv[RB];
    int64_t sr[RB];
#pragma unroll
    for (int r = 0; r < RB; ++r) sr[r] = row_list ? (n + r < listed ? (int64_t)row_list[n + r] : 0) : n + r;
    if (act != SPGNN_ACT_NONE) {
#pragma unroll
      for (int r = 0; r < RB; ++r)
#pragma unroll
        for (int h = 0; h < HMAX; ++h)
          if (HT || h < H) o[r][h] = ld4(out + sr[r] * out_ld + (int64_t)h * D + cc);
    }
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      gv[r] = gS[sr[r] * ldg + jl];
      if (row_list && n + r >= listed) gv[r] = 0.f;
      if (overflowed) gv[r] = NAN;
    }
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      float4 gm, em = make_float4(0.f, 0.f, 0.f, 0.f);
      if (WG) {                                          // the head mean of this row's four columns (WG needs `out`: host-checked)
#pragma unroll
        for (int h = 0; h < HMAX; ++h)
          if (HT || h < H) { em.x += o[r][h].x; em.y += o[r][h].y; em.z += o[r][h].z; em.w += o[r][h].w; }
        em.x *= inv_h; em.y *= inv_h; em.z *= inv_h; em.w *= inv_h;
      }
      SPGNN_ABP_GM(gm, gv[r], em)
#pragma unroll
      for (int h = 0; h < HMAX; ++h)
        if (HT || h < H) SPGNN_ABP_OUT(gm, o[r][h], n + r, h)
    }
  }
  for (; n < n1; ++n) {
    const int64_t sr = row_list ? (n < listed ? (int64_t)row_list[n] : 0) : n;
    float gv = gS[sr * ldg + jl];
    if (row_list && n >= listed) gv = 0.f;
    if (overflowed) gv = NAN;
    float4 ot[HMAX];
    float4 em = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int h = 0; h < HMAX; ++h) {
      ot[h] = make_float4(0.f, 0.f, 0.f, 0.f);
      if ((HT || h < H) && act != SPGNN_ACT_NONE) ot[h] = ld4(out + sr * out_ld + (int64_t)h * D + cc);
      if (WG && (HT || h < H)) { em.x += ot[h].x; em.y += ot[h].y; em.z += ot[h].z; em.w += ot[h].w; }
    }
    em.x *= inv_h; em.y *= inv_h; em.z *= inv_h; em.w *= inv_h;
    float4 gm;
    SPGNN_ABP_GM(gm, gv, em)
#pragma unroll
    for (int h = 0; h < HMAX; ++h)
      if (HT || h < H) SPGNN_ABP_OUT(gm, ot[h], n, h)
  }
#undef SPGNN_ABP_GM
#undef SPGNN_ABP_OUT
  if (WG && cv) {
#pragma unroll
    for (int j = 0; j < JP; ++j)
      if (j < J) st4(wgrad + ((int64_t)blockIdx.x * J + j) * D + c, gw[j]);
  }
  mx = team_max(mx, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) spgnn_detail::slots_max(absmax, fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])), blockIdx.x);
}

// =================================================================================================
// Score-vector folding: w_lr[h,:] = sum_d attn_l[h,d] * W[h*D+d,:], w_lr[H+h,:] likewise with attn_r - the (2H, K)
// weights of the skinny score projection (and their gradients), as two small kernels instead of a dozen
// batched-GEMM / cat / add launches per layer and step.
// =================================================================================================
__global__ __launch_bounds__(1024) void fold_scores_fwd(const float* __restrict__ W, int64_t ldw, const float* __restrict__ al,
                                                        const float* __restrict__ ar, float* __restrict__ out, int Kp, int H,
                                                        int D, int K) {
  // fp64 accumulation: these few MFLOP cost nothing, and the scores' gradients downstream are near-total
  // cancellations (softmax is shift-invariant in er up to the LeakyReLU kink), so every ulp here shows there.
  // 16 columns x 64 row groups per block (was 64 columns x 16: 6 blocks for the 192-column output layer, 45 us at any
  // batch size - a quarter of the launches' time at 64 trees); the weights are L2-resident, 64-byte row pieces do no harm.
  __shared__ double red[2][64][16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = blockIdx.y;
  const int c = lane & 15, rg = wave * 4 + (lane >> 4);
  const int k = blockIdx.x * 16 + c;
  double sl = 0.0, sr = 0.0;
  if (k < K) {
    for (int d0 = rg; d0 < D; d0 += 64 * 8) {
      float w[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) { const int d = d0 + 64 * i; w[i] = d < D ? W[(int64_t)(h * D + d) * ldw + k] : 0.f; }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int d = d0 + 64 * i;
        if (d < D) { sl = fma((double)al[h * D + d], (double)w[i], sl); sr = fma((double)ar[h * D + d], (double)w[i], sr); }
      }
    }
  }
  red[0][rg][c] = sl; red[1][rg][c] = sr;
  __syncthreads();
  if (threadIdx.x < 32) {                         // fixed summation order: reproducible
    const int which = threadIdx.x >> 4, c2 = threadIdx.x & 15, k2 = blockIdx.x * 16 + c2;
    if (k2 < Kp) {
      double s = 0.0;
#pragma unroll 8
      for (int i = 0; i < 64; ++i) s += red[which][i][c2];
      out[(int64_t)(which * H + h) * Kp + k2] = k2 < K ? (float)s : 0.f;
    }
  }
}

// one wave per weight row (h,d): g_al[h,d] = <g_wlr[h,:], W[row,:]>, g_ar likewise; g_W[row,:] = al*g_wlr[h,:] + ar*g_wlr[H+h,:]
__global__ __launch_bounds__(kBlock) void fold_scores_bwd(const float* __restrict__ W, int64_t ldw, const float* __restrict__ al,
                                                          const float* __restrict__ ar, const float* __restrict__ g_wlr, int Kp,
                                                          float* __restrict__ g_W, int64_t ldg, float* __restrict__ g_al,
                                                          float* __restrict__ g_ar, int H, int D, int K) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  if (row >= H * D) return;
  const int h = row / D;
  const float a_l = al[row], a_r = ar[row];
  double dl = 0.0, dr = 0.0;
  for (int k = lane; k < K; k += 64) {
    const float w = W[(int64_t)row * ldw + k];
    const float gl = g_wlr[(int64_t)h * Kp + k], gr = g_wlr[(int64_t)(H + h) * Kp + k];
    dl = fma((double)gl, (double)w, dl); dr = fma((double)gr, (double)w, dr);
    g_W[(int64_t)row * ldg + k] = a_l * gl + a_r * gr;
  }
  for (int off = 32; off > 0; off >>= 1) {
    asm volatile("s_nop 7" ::: "memory");         // multi-pass fp64 results settle before the cross-lane read (see single_pass)
    dl += __shfl_xor(dl, off, 64); dr += __shfl_xor(dr, off, 64);
  }
  if (lane == 0) { g_al[row] = (float)dl; g_ar[row] = (float)dr; }
}

// =================================================================================================
// cat + feature dropout in one pass (the layer input of every hidden SPGNN layer is dropout(cat[h_s, h_p]),
// reference models.py:477-481 with GATConv's feat_drop): out[:, off:off+w] = src * keep/(1-p), keep from the same
// counter hash as the attention dropout, regenerated (not stored) by the backward pass
//   g_src = g_out[:, off:off+w] * keep/(1-p).
// One launch per source; element (row, off + c) uses counter row * width_total + off + c.
// =================================================================================================
// One 64-bit hash serves four neighbouring elements (16 bits each: keep iff bits >= p * 65536), counter = index of the
// group's first element.

template <typename ST, bool VEC>
__global__ __launch_bounds__(kBlock) void cat_dropout_kernel(const ST* __restrict__ src, int64_t src_ld, ST* __restrict__ dst,
                                                             int64_t dst_ld, int64_t N, int w, int off, int total, float p,
                                                             float inv_keep, uint64_t seed, const uint64_t* __restrict__ seed_off,
                                                             int backward, float* __restrict__ absmax) {
  if (seed_off) seed += seed_off[0];
  const int w4 = (w + 3) >> 2;
  const unsigned thr = (unsigned)(p * 65536.f);
  float amx = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N * w4; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / w4; const int c = (int)(i % w4) * 4;
    // forward: src is a source tensor (its own column 0 = output column off); backward: src is the gradient of the
    // concatenation (read at off + c) and dst the gradient of the source
    const ST* sp = src + row * src_ld + (backward ? off : 0) + c;
    ST* dp = dst + row * dst_ld + (backward ? 0 : off) + c;
    float v[4];
    if (VEC) { const float4 q = ldv(sp); v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w; }
    else if constexpr (is_f32<ST>::value) {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = c + j < w ? sp[j] : 0.f;
    }
    if (p > 0.f) {
      const uint64_t z = mix64(seed, row * total + off + c);
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] *= ((unsigned)(z >> (16 * j)) & 0xFFFFu) >= thr ? inv_keep : 0.f;
    }
    if (VEC) stv(dp, make_float4(v[0], v[1], v[2], v[3]));
    else if constexpr (is_f32<ST>::value) {
#pragma unroll
      for (int j = 0; j < 4; ++j) if (c + j < w) dp[j] = v[j];
    }
    amx = fmaxf(fmaxf(amx, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));   // padded lanes hold 0
  }
  if (absmax) {                                     // one partial maximum per block (the consumer GEMM's operand scale)
    __shared__ float red[kBlock / 64];
    amx = team_max(amx, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = amx;
    __syncthreads();
    if (threadIdx.x == 0) {
      float m = red[0];
#pragma unroll
      for (int q = 1; q < kBlock / 64; ++q) m = fmaxf(m, red[q]);
      spgnn_detail::slots_max(absmax, m, blockIdx.x);
    }
  }
}

// el / er from the projection GEMM's score partials: s[v, h] = sum of head h's 64-column blocks of parts[v, :, 0],
// s[v, H + h] likewise of parts[v, :, 1]  (parts: (N, H*D/64, 2))
__global__ __launch_bounds__(kBlock) void scores_from_parts_kernel(const float* __restrict__ parts, float* __restrict__ s, int64_t s_ld,
                                                                   int64_t N, int H, int bph) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= N * H) return;
  const int64_t v = i / H; const int h = (int)(i % H);
  const float2* p = reinterpret_cast<const float2*>(parts) + (v * H + h) * bph;
  float l = 0.f, r = 0.f;
  for (int b = 0; b < bph; ++b) { const float2 q = p[b]; l += q.x; r += q.y; }
  s[v * s_ld + h] = l; s[v * s_ld + H + h] = r;
}

// =================================================================================================
// SpMM sum / max
// =================================================================================================
struct SpmmSum {
  const int32_t* indptr; const int32_t* indices;
  const float* x; int64_t x_ld;
  const float* w_src; const float* w_dst; const float* self_eps;
  float* out; int64_t out_ld;
  int64_t N; int F; int T;
  const float* bias; int act;            // optional epilogue out = act(... + bias[col]) (GraphConv: reference models.py:172-182)
  float* absmax;                         // optional scale block: max |out| folded into its slots (the result as a GEMM operand)
  // optional feature dropout of the stored rows (after bias / activation): spgnn_cat_dropout's mask for an F-wide row
  float drop_p, drop_inv; uint64_t drop_seed; const uint64_t* drop_seed_off;
};

// TT = 64: one node per wave with wave-uniform index / weight values in SGPRs; TT = 0: run-time team width
// (see gat_fwd_vec).  Nodes with 1..8 in-edges load all indices and source weights unconditionally and fetch
// the neighbour rows in batches; larger degrees loop.
template <int TT, int R>
__global__ __launch_bounds__(kBlock) void spmm_sum_vec(SpmmSum a) {
  constexpr bool WAVE = TT == 64;
  const int T = WAVE ? 64 : a.T;
  const int64_t v = xcd_block() * (kBlock / T) + uni<WAVE>((int)(threadIdx.x / T));
  if (v >= a.N) return;
  const int lane = threadIdx.x % T;
  const int beg = uni<WAVE>(a.indptr[v]), end = uni<WAVE>(a.indptr[v + 1]), deg = end - beg;
  float4 acc[R];
#pragma unroll
  for (int r = 0; r < R; ++r) acc[r] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (deg > 0 && deg <= kMaxFast) {
    int u[kMaxFast]; float w[kMaxFast];
#pragma unroll
    for (int k = 0; k < kMaxFast; ++k) u[k] = uni<WAVE>(a.indices[beg + (k < deg ? k : deg - 1)]);
#pragma unroll
    for (int k = 0; k < kMaxFast; ++k) {
      const float ws = a.w_src ? a.w_src[u[k]] : 1.f;
      w[k] = uni<WAVE>(k < deg ? ws : 0.f);
    }
    constexpr int kGather = R >= 8 ? 1 : R == 4 ? 2 : 4;
#pragma unroll
    for (int k0 = 0; k0 < kMaxFast; k0 += kGather) {
      if (WAVE ? !(k0 < deg) : !__any(k0 < deg)) break;
      float4 x[kGather][R];
#pragma unroll
      for (int q = 0; q < kGather; ++q)
#pragma unroll
        for (int r = 0; r < R; ++r) x[q][r] = ld4(a.x + (int64_t)u[k0 + q] * a.x_ld + (r * T + lane) * 4);
#pragma unroll
      for (int q = 0; q < kGather; ++q)
#pragma unroll
        for (int r = 0; r < R; ++r) fma4(acc[r], w[k0 + q], x[q][r]);
    }
  } else {
    for (int j = beg; j < end; ++j) {
      const int64_t u = a.indices[j];
      const float w = a.w_src ? a.w_src[u] : 1.f;
      const float* row = a.x + u * a.x_ld;
#pragma unroll
      for (int r = 0; r < R; ++r) fma4(acc[r], w, ld4(row + (r * T + lane) * 4));
    }
  }
  const float wd = a.w_dst ? a.w_dst[v] : 1.f;
  const float sc = a.self_eps ? 1.f + a.self_eps[0] : 0.f;
  const uint64_t dseed = a.drop_p > 0.f ? a.drop_seed + (a.drop_seed_off ? a.drop_seed_off[0] : 0) : 0;
  float amx = 0.f;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int c = (r * T + lane) * 4;
    float4 o = make_float4(acc[r].x * wd, acc[r].y * wd, acc[r].z * wd, acc[r].w * wd);
    if (a.self_eps) fma4(o, sc, ld4(a.x + v * a.x_ld + c));
    if (a.bias) { const float4 b = ld4(a.bias + c); o.x += b.x; o.y += b.y; o.z += b.z; o.w += b.w; }
    if (a.act != SPGNN_ACT_NONE) { o.x = act_fwd(o.x, a.act); o.y = act_fwd(o.y, a.act); o.z = act_fwd(o.z, a.act); o.w = act_fwd(o.w, a.act); }
    if (a.drop_p > 0.f) {
      const float4 k = feat_keep4(dseed, v * a.F + c, a.drop_p, a.drop_inv);
      o.x *= k.x; o.y *= k.y; o.z *= k.z; o.w *= k.w;
    }
    st4(a.out + v * a.out_ld + c, o);
    amx = absmax4(amx, o);
  }
  if (a.absmax) {
    amx = team_max(amx, T);
    if (lane == 0) spgnn_detail::slots_max(a.absmax, amx, (unsigned)v);
  }
}

__global__ void spmm_sum_scalar(SpmmSum a) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= a.N * a.F) return;
  const int64_t v = gid / a.F; const int col = (int)(gid % a.F);
  float acc = 0.f;
  for (int j = a.indptr[v]; j < a.indptr[v + 1]; ++j) {
    const int64_t u = a.indices[j];
    acc = fmaf(a.w_src ? a.w_src[u] : 1.f, a.x[u * a.x_ld + col], acc);
  }
  acc *= a.w_dst ? a.w_dst[v] : 1.f;
  if (a.self_eps) acc = fmaf(1.f + a.self_eps[0], a.x[v * a.x_ld + col], acc);
  if (a.bias) acc += a.bias[col];
  a.out[v * a.out_ld + col] = act_fwd(acc, a.act);
}

struct SpmmMaxFwd {
  const int32_t* indptr; const int32_t* indices;
  const float* x; int64_t x_ld;
  float* out; int64_t out_ld;
  int32_t* arg; int64_t arg_ld;
  int64_t N; int F; int T;
  uint8_t* arg8;        // compact form (spgnn_spmm_max_fwd_u8): the winner's position INSIDE v's in-edge list, 255 = none
};

// U8: the argmax goes out as one byte per element (position inside the destination's in-edge list, in-degree <= 254) instead
// of the 32-bit CSC slot: the backward pass gathers the arg rows of every out-neighbour, so at 1024 columns the slot form
// moves 4 KB per edge there (and 315 MB per pass at 76k nodes) where one KB does.
template <int TT, int R, bool U8>
__device__ __forceinline__ void spmm_max_fwd_body(const SpmmMaxFwd& a) {
  constexpr bool WAVE = TT == 64;
  const int T = WAVE ? 64 : a.T;
  const int64_t v = xcd_block() * (kBlock / T) + uni<WAVE>((int)(threadIdx.x / T));
  if (v >= a.N) return;
  const int lane = threadIdx.x % T;
  const int beg = uni<WAVE>(a.indptr[v]), end = uni<WAVE>(a.indptr[v + 1]), deg = end - beg;
  float4 best[R]; int4 arg[R];
#pragma unroll
  for (int r = 0; r < R; ++r) { best[r] = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY); arg[r] = make_int4(-1, -1, -1, -1); }
  if (deg > 0 && deg <= kMaxFast) {
    int u[kMaxFast];
#pragma unroll
    for (int k = 0; k < kMaxFast; ++k) u[k] = uni<WAVE>(a.indices[beg + (k < deg ? k : deg - 1)]);
    constexpr int kGather = R >= 8 ? 1 : R == 4 ? 2 : 4;
#pragma unroll
    for (int k0 = 0; k0 < kMaxFast; k0 += kGather) {
      if (WAVE ? !(k0 < deg) : !__any(k0 < deg)) break;
      float4 x[kGather][R];
#pragma unroll
      for (int q = 0; q < kGather; ++q)
#pragma unroll
        for (int r = 0; r < R; ++r) x[q][r] = ld4(a.x + (int64_t)u[k0 + q] * a.x_ld + (r * T + lane) * 4);
#pragma unroll
      for (int q = 0; q < kGather; ++q) {
        const bool live = k0 + q < deg;               // clamped slots repeat the last edge: they must not win a tie
        const int j = beg + k0 + q;
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const float4 t = x[q][r];
          if (live && t.x > best[r].x) { best[r].x = t.x; arg[r].x = j; }
          if (live && t.y > best[r].y) { best[r].y = t.y; arg[r].y = j; }
          if (live && t.z > best[r].z) { best[r].z = t.z; arg[r].z = j; }
          if (live && t.w > best[r].w) { best[r].w = t.w; arg[r].w = j; }
        }
      }
    }
  } else {
    for (int j = beg; j < end; ++j) {
      const float* row = a.x + (int64_t)a.indices[j] * a.x_ld;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const float4 q = ld4(row + (r * T + lane) * 4);
        if (q.x > best[r].x) { best[r].x = q.x; arg[r].x = j; }
        if (q.y > best[r].y) { best[r].y = q.y; arg[r].y = j; }
        if (q.z > best[r].z) { best[r].z = q.z; arg[r].z = j; }
        if (q.w > best[r].w) { best[r].w = q.w; arg[r].w = j; }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int c = (r * T + lane) * 4;
    float4 o = best[r];
    if (arg[r].x < 0) o.x = 0.f;
    if (arg[r].y < 0) o.y = 0.f;
    if (arg[r].z < 0) o.z = 0.f;
    if (arg[r].w < 0) o.w = 0.f;
    st4(a.out + v * a.out_ld + c, o);
    if constexpr (U8) {
      const unsigned b0 = arg[r].x < 0 ? 255u : (unsigned)(arg[r].x - beg), b1 = arg[r].y < 0 ? 255u : (unsigned)(arg[r].y - beg);
      const unsigned b2 = arg[r].z < 0 ? 255u : (unsigned)(arg[r].z - beg), b3 = arg[r].w < 0 ? 255u : (unsigned)(arg[r].w - beg);
      *reinterpret_cast<uint32_t*>(a.arg8 + v * a.arg_ld + c) = b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
    } else {
      *reinterpret_cast<int4*>(a.arg + v * a.arg_ld + c) = arg[r];
    }
  }
}
template <int TT, int R> __global__ __launch_bounds__(kBlock) void spmm_max_fwd_vec(SpmmMaxFwd a) { spmm_max_fwd_body<TT, R, false>(a); }
template <int TT, int R> __global__ __launch_bounds__(kBlock) void spmm_max_fwd_vec_u8(SpmmMaxFwd a) { spmm_max_fwd_body<TT, R, true>(a); }

__global__ void spmm_max_fwd_scalar(SpmmMaxFwd a) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= a.N * a.F) return;
  const int64_t v = gid / a.F; const int col = (int)(gid % a.F);
  float best = -INFINITY; int arg = -1;
  for (int j = a.indptr[v]; j < a.indptr[v + 1]; ++j) {
    const float q = a.x[(int64_t)a.indices[j] * a.x_ld + col];
    if (q > best) { best = q; arg = j; }
  }
  a.out[v * a.out_ld + col] = arg < 0 ? 0.f : best;
  a.arg[v * a.arg_ld + col] = arg;
}

struct SpmmMaxBwd {
  const int32_t* out_indptr; const int32_t* out_indices; const int32_t* out_pos;
  const float* g_out; int64_t g_out_ld;
  const int32_t* arg; int64_t arg_ld;
  float* g_x; int64_t g_x_ld;
  int64_t N; int F; int T;
  const uint8_t* arg8; const int32_t* indptr;     // compact form: positions inside the in-edge lists + the CSC offsets
  // optional: x was a ReLU output (SAGEConv's fc_pool, reference models.py:668-679) - g_x is masked by relu_of > 0 right here
  // (the activation's backward pass over g_x is not needed) and max |g_x| goes to the scale block `absmax`
  const float* relu_of; int64_t relu_ld; float* absmax;
};

template <int TT, int R, bool U8>
__device__ __forceinline__ void spmm_max_bwd_body(const SpmmMaxBwd& a) {
  constexpr bool WAVE = TT == 64;
  const int T = WAVE ? 64 : a.T;
  const int64_t u = xcd_block() * (kBlock / T) + uni<WAVE>((int)(threadIdx.x / T));
  if (u >= a.N) return;
  const int lane = threadIdx.x % T;
  const int beg = uni<WAVE>(a.out_indptr[u]), end = uni<WAVE>(a.out_indptr[u + 1]), deg = end - beg;
  float4 acc[R];
#pragma unroll
  for (int r = 0; r < R; ++r) acc[r] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (deg > 0 && deg <= kMaxFast) {
    int vv[kMaxFast], pp[kMaxFast];
#pragma unroll
    for (int k = 0; k < kMaxFast; ++k) {
      vv[k] = uni<WAVE>(a.out_indices[beg + (k < deg ? k : deg - 1)]);
      pp[k] = uni<WAVE>(k < deg ? a.out_pos[beg + (k < deg ? k : deg - 1)] : -2);      // -2 never equals an arg
    }
    if constexpr (U8) {                              // CSC slot -> position inside the destination's list
#pragma unroll
      for (int k = 0; k < kMaxFast; ++k) {
        const int b = uni<WAVE>(a.indptr[vv[k]]);
        pp[k] = k < deg ? pp[k] - b : 256;           // 256 never equals a byte
      }
    }
    constexpr int kGather = R >= 4 ? 1 : 2;          // two row streams (arg, g_out) per edge
#pragma unroll
    for (int k0 = 0; k0 < kMaxFast; k0 += kGather) {
      if (WAVE ? !(k0 < deg) : !__any(k0 < deg)) break;
      int4 ar[kGather][R]; float4 g[kGather][R];
#pragma unroll
      for (int q = 0; q < kGather; ++q)
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const int c = (r * T + lane) * 4;
          if constexpr (U8) {
            const uint32_t w = *reinterpret_cast<const uint32_t*>(a.arg8 + (int64_t)vv[k0 + q] * a.arg_ld + c);
            ar[q][r] = make_int4((int)(w & 255u), (int)((w >> 8) & 255u), (int)((w >> 16) & 255u), (int)(w >> 24));
          } else {
            ar[q][r] = *reinterpret_cast<const int4*>(a.arg + (int64_t)vv[k0 + q] * a.arg_ld + c);
          }
          g[q][r] = ld4(a.g_out + (int64_t)vv[k0 + q] * a.g_out_ld + c);
        }
#pragma unroll
      for (int q = 0; q < kGather; ++q)
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const int pos = pp[k0 + q];
          if (ar[q][r].x == pos) acc[r].x += g[q][r].x;
          if (ar[q][r].y == pos) acc[r].y += g[q][r].y;
          if (ar[q][r].z == pos) acc[r].z += g[q][r].z;
          if (ar[q][r].w == pos) acc[r].w += g[q][r].w;
        }
    }
  } else {
    for (int k = beg; k < end; ++k) {
      const int64_t v = a.out_indices[k];
      const int pos = U8 ? a.out_pos[k] - a.indptr[v] : a.out_pos[k];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int c = (r * T + lane) * 4;
        int4 ar;
        if constexpr (U8) {
          const uint32_t w = *reinterpret_cast<const uint32_t*>(a.arg8 + v * a.arg_ld + c);
          ar = make_int4((int)(w & 255u), (int)((w >> 8) & 255u), (int)((w >> 16) & 255u), (int)(w >> 24));
        } else {
          ar = *reinterpret_cast<const int4*>(a.arg + v * a.arg_ld + c);
        }
        const float4 g = ld4(a.g_out + v * a.g_out_ld + c);
        if (ar.x == pos) acc[r].x += g.x;
        if (ar.y == pos) acc[r].y += g.y;
        if (ar.z == pos) acc[r].z += g.z;
        if (ar.w == pos) acc[r].w += g.w;
      }
    }
  }
  float amx = 0.f;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int c = (r * T + lane) * 4;
    float4 q = acc[r];
    if (a.relu_of) {
      const float4 o = ld4(a.relu_of + u * a.relu_ld + c);
      q.x = o.x > 0.f ? q.x : 0.f; q.y = o.y > 0.f ? q.y : 0.f; q.z = o.z > 0.f ? q.z : 0.f; q.w = o.w > 0.f ? q.w : 0.f;
    }
    st4(a.g_x + u * a.g_x_ld + c, q);
    amx = absmax4(amx, q);
  }
  if (a.absmax) {
    amx = team_max(amx, T);
    if (lane == 0) spgnn_detail::slots_max(a.absmax, amx, (unsigned)u);
  }
}
template <int TT, int R> __global__ __launch_bounds__(kBlock) void spmm_max_bwd_vec(SpmmMaxBwd a) { spmm_max_bwd_body<TT, R, false>(a); }
template <int TT, int R> __global__ __launch_bounds__(kBlock) void spmm_max_bwd_vec_u8(SpmmMaxBwd a) { spmm_max_bwd_body<TT, R, true>(a); }

__global__ void spmm_max_bwd_scalar(SpmmMaxBwd a) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= a.N * a.F) return;
  const int64_t u = gid / a.F; const int col = (int)(gid % a.F);
  float acc = 0.f;
  for (int k = a.out_indptr[u]; k < a.out_indptr[u + 1]; ++k) {
    const int64_t v = a.out_indices[k];
    if (a.arg[v * a.arg_ld + col] == a.out_pos[k]) acc += a.g_out[v * a.g_out_ld + col];
  }
  a.g_x[u * a.g_x_ld + col] = acc;
}

// =================================================================================================
// Attention-score projections: the skinny side of GATConv (J = 2H <= 16 output columns).
//   fwd : S[n][j]   = sum_k X[n][k] * W[j][k]            (el | er for every node)
//   dW  : gW[j][k]  = sum_n gS[n][j] * X[n][k]
//   dX  : gX[n][k] += sum_j gS[n][j] * W[j][k]
// rocBLAS runs these shapes 2-4x off the byte bound (0.10-0.24 ms each at N = 76k); all three are pure
// streaming passes over X / gX.  The forward uses the fp32 MFMA (16x16x4) so that the k-reduction needs no
// cross-lane traffic: 16 rows x 16 (zero-padded) columns per wave.  W is passed zero-padded to Kp = 16*ceil(K/16).
// =================================================================================================
typedef float f32x4 __attribute__((ext_vector_type(4)));

// NG 16-column groups (J <= 16 * NG), RG 16-row groups per wave; ST: storage type of the rows of X (W, S fp32).
// A wave keeps its W fragments in registers across its RG row groups: with one group per wave the 22 x 1024 classifier
// weights were re-read from L1 for every 16 rows - twice the bytes of x (1024 -> 22: 2.7 TB/s of x).
// KS = 4 (deep products, K >= 512): the four waves of a block share ONE set of 16 RG rows and each takes a quarter of the
// k range; the partial accumulators meet in LDS and wave 0 adds them in wave order (deterministic).  Four times the waves
// for the same rows: the 1024 -> 22 classifier product ran with ~2 waves per SIMD, each alone with its load latency.
template <typename ST, int NG, int RG, int KS>
__global__ __launch_bounds__(kBlock) void scores_fwd_mfma(const ST* __restrict__ X, int64_t ldx,
                                                          const float* __restrict__ W, int Kp,
                                                          float* __restrict__ S, int64_t lds_, int64_t N, int K, int J,
                                                          float* __restrict__ absmax, const float* __restrict__ bias) {
  static_assert(KS == 1 || KS == kBlock / 64, "k split = the waves of a block");
  const int ks = KS > 1 ? (int)(threadIdx.x >> 6) : 0;
  const int64_t wave = KS > 1 ? (int64_t)blockIdx.x : ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 6;   // index of the row set
  const int lane = threadIdx.x & 63;
  const int64_t row0 = wave * (16 * RG);
  if (row0 >= N) return;                // (block-uniform with KS > 1)
  const int r = lane & 15, q = lane >> 4;
  bool rv[RG]; const ST* xp[RG]; float rvf[RG];
#pragma unroll
  for (int t = 0; t < RG; ++t) {
    rv[t] = row0 + 16 * t + r < N;
    xp[t] = X + (rv[t] ? row0 + 16 * t + r : 0) * ldx + 4 * q;
    rvf[t] = rv[t] ? 1.f : 0.f;
  }
  bool wv[NG]; const float* wp[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    wv[g] = r + 16 * g < J;
    wp[g] = W + (int64_t)(wv[g] ? r + 16 * g : 0) * Kp + 4 * q;
  }
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  f32x4 acc[RG][NG];
#pragma unroll
  for (int t = 0; t < RG; ++t)
#pragma unroll
    for (int g = 0; g < NG; ++g) acc[t][g] = f32x4{0.f, 0.f, 0.f, 0.f};
  float amx[RG];                      // every element of x passes through this kernel: its absmax is free
#pragma unroll
  for (int t = 0; t < RG; ++t) amx[t] = 0.f;
  const int kfull_all = K & ~15;
  // this wave's k range [kbeg, kfull): whole 16-element steps
  const int kq = KS > 1 ? ((kfull_all / 16 + KS - 1) / KS) * 16 : kfull_all;
  const int kbeg = ks * kq < kfull_all ? ks * kq : kfull_all;
  const int kfull = kbeg + kq < kfull_all ? kbeg + kq : kfull_all;
  // main loop: straight-line body, unrolled so that all row loads of a trip are in flight before the first MFMA.
  // Loads are unconditional: rows past N and columns past J read row 0 / column 0 (valid memory) and only feed
  // outputs that are never stored.  A per-lane test around a load (or around the absmax update) is a branch to hipcc:
  // the loop was not unrolled and every trip waited vmcnt(0) for its own loads - one load in flight per wave.
  constexpr int UK = RG >= 4 ? 2 : 4;  // k16 steps per trip (4 RG UK / 4 ... row loads + NG UK weight loads in flight)
  int k0 = kbeg;
  for (; k0 + 16 * UK <= kfull; k0 += 16 * UK) {
    float4 xa[RG][UK], wb[NG][UK];
#pragma unroll
    for (int t = 0; t < RG; ++t)
#pragma unroll
      for (int u = 0; u < UK; ++u) xa[t][u] = ldv(xp[t] + k0 + 16 * u);
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
      for (int u = 0; u < UK; ++u) wb[g][u] = ld4(wp[g] + k0 + 16 * u);
#pragma unroll
    for (int u = 0; u < UK; ++u)
#pragma unroll
      for (int t = 0; t < RG; ++t) {
        amx[t] = fmaxf(amx[t], rvf[t] * fmaxf(fmaxf(fabsf(xa[t][u].x), fabsf(xa[t][u].y)), fmaxf(fabsf(xa[t][u].z), fabsf(xa[t][u].w))));
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          acc[t][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[t][u].x, wb[g][u].x, acc[t][g], 0, 0, 0);
          acc[t][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[t][u].y, wb[g][u].y, acc[t][g], 0, 0, 0);
          acc[t][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[t][u].z, wb[g][u].z, acc[t][g], 0, 0, 0);
          acc[t][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[t][u].w, wb[g][u].w, acc[t][g], 0, 0, 0);
        }
      }
  }
  for (; k0 < kfull; k0 += 16) {
    float4 wb[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) wb[g] = ld4(wp[g] + k0);
#pragma unroll
    for (int t = 0; t < RG; ++t) {
      const float4 xa = ldv(xp[t] + k0);
      amx[t] = fmaxf(amx[t], rvf[t] * fmaxf(fmaxf(fabsf(xa.x), fabsf(xa.y)), fmaxf(fabsf(xa.z), fabsf(xa.w))));
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        acc[t][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa.x, wb[g].x, acc[t][g], 0, 0, 0);
        acc[t][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa.y, wb[g].y, acc[t][g], 0, 0, 0);
        acc[t][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa.z, wb[g].z, acc[t][g], 0, 0, 0);
        acc[t][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa.w, wb[g].w, acc[t][g], 0, 0, 0);
      }
    }
  }
  if (kfull_all < K && ks == KS - 1) {  // ragged tail (the last wave's): element-wise guards on X (W is zero padded)
    const int kt = kfull_all, k = kt + 4 * q;
    float4 wb[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) wb[g] = wv[g] ? ld4(wp[g] + kt) : z4;
#pragma unroll
    for (int t = 0; t < RG; ++t) {
      float4 xa = z4;
      if (rv[t] && k < K) {             // the row stride is a multiple of 4 >= K: the whole chunk at k < K is inside the row
        const float4 t_ = ldv(xp[t] + kt);
        if (k + 0 < K) xa.x = t_.x;
        if (k + 1 < K) xa.y = t_.y;
        if (k + 2 < K) xa.z = t_.z;
        if (k + 3 < K) xa.w = t_.w;
      }
      amx[t] = absmax4(amx[t], xa);
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        acc[t][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa.x, wb[g].x, acc[t][g], 0, 0, 0);
        acc[t][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa.y, wb[g].y, acc[t][g], 0, 0, 0);
        acc[t][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa.z, wb[g].z, acc[t][g], 0, 0, 0);
        acc[t][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa.w, wb[g].w, acc[t][g], 0, 0, 0);
      }
    }
  }
  if (absmax) {                         // one entry per 16 rows, as the callers size the array
#pragma unroll
    for (int t = 0; t < RG; ++t) {
      float m = amx[t];
      for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
      if (lane == 0 && row0 + 16 * t < N) spgnn_detail::slots_max(absmax, m, (unsigned)(wave * RG + t));   // max: any order
    }
  }
  if constexpr (KS > 1) {               // partial accumulators of waves 1 .. KS-1 -> LDS; wave 0 adds them in wave order
    __shared__ f32x4 red[(KS - 1) * RG * NG * 64];
    if (ks > 0) {
#pragma unroll
      for (int t = 0; t < RG; ++t)
#pragma unroll
        for (int g = 0; g < NG; ++g) red[(((ks - 1) * RG + t) * NG + g) * 64 + lane] = acc[t][g];
    }
    __syncthreads();
    if (ks > 0) return;
#pragma unroll
    for (int w = 1; w < KS; ++w)
#pragma unroll
      for (int t = 0; t < RG; ++t)
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          const f32x4 p_ = red[(((w - 1) * RG + t) * NG + g) * 64 + lane];
          acc[t][g][0] += p_[0]; acc[t][g][1] += p_[1]; acc[t][g][2] += p_[2]; acc[t][g][3] += p_[3];     // (no vector add: packed fp32 ops are fenced out of this file)
        }
  }
  // C/D layout of 16x16x4: col = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
  for (int t = 0; t < RG; ++t)
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      if (r + 16 * g < J) {
        const float bcol = bias ? bias[r + 16 * g] : 0.f;           // optional S += bias[column] (a skinny Linear's bias)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int64_t orow = row0 + 16 * t + q * 4 + j;
          if (orow < N) S[orow * lds_ + r + 16 * g] = acc[t][g][j] + bcol;
        }
      }
    }
}

// =================================================================================================
// Skinny product for SMALL batches (ABI 56): C = act(A B^T + bias), A (M, K), B (N, K), M up to a few hundred rows - the
// per-scan inference forward of the reference (ONE tree per model.forward, job_runner.py:2046-2052: M = 100-300) and the
// 2-5-tree batches of tests.  The split-fp16 kernels of spgnn_gemm.hip tile for M in the tens of thousands: at M = 150 a
// 1063 -> 1024 projection is 16 workgroups walking 34 K stages one after the other (35 us, 4 x rocBLAS).  Here a
// workgroup owns 16 rows x 64 columns, its four waves split K, the fp32 matrix pipe (v_mfma_f32_16x16x4_f32: exact
// fp32 products, fp32 accumulation) does the arithmetic and the partial accumulators meet in LDS in wave order
// (deterministic): M / 16 x N / 64 workgroups, every one of them short.  Optional epilogue as the large kernels':
// bias, activation, and GATConv's per-64-column score partials <C[row, 64 b : 64 b + 64], attn[...]> (spgnn_gemm_nt).
// =================================================================================================
template <int KS>
__global__ __launch_bounds__(KS * 64) void gemm_nt_skinny(const float* __restrict__ A, int64_t lda, const float* __restrict__ B, int64_t ldb,
                                                         float* __restrict__ C, int64_t ldc, int64_t M, int N, int K,
                                                         const float* __restrict__ bias, int act,
                                                         const float* __restrict__ score_l, const float* __restrict__ score_r,
                                                         float* __restrict__ score_out, int score_cols) {
  constexpr int NG = 4;                                  // 16-column groups per workgroup: one 64-column block
  const int ks = (int)(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int r = lane & 15, q = lane >> 4;
  const int64_t row0 = (int64_t)blockIdx.x * 16;
  const int col0 = (int)blockIdx.y * 64;
  const bool rv = row0 + r < M;
  const float* ap = A + (rv ? row0 + r : 0) * lda + 4 * q;
  const float* bp[NG]; bool cvv[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    cvv[g] = col0 + 16 * g + r < N;
    bp[g] = B + (int64_t)(cvv[g] ? col0 + 16 * g + r : 0) * ldb + 4 * q;
  }
  f32x4 acc[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int kfull_all = K & ~15;
  const int kq = ((kfull_all / 16 + KS - 1) / KS) * 16;
  const int kbeg = ks * kq < kfull_all ? ks * kq : kfull_all;
  const int kfull = kbeg + kq < kfull_all ? kbeg + kq : kfull_all;
  constexpr int UK = 4;                                  // k16 steps per trip: 4 + 16 row loads in flight before the first MFMA
  int k0 = kbeg;
  for (; k0 + 16 * UK <= kfull; k0 += 16 * UK) {
    float4 xa[UK], wb[NG][UK];
#pragma unroll
    for (int u = 0; u < UK; ++u) xa[u] = ld4(ap + k0 + 16 * u);
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
      for (int u = 0; u < UK; ++u) wb[g][u] = ld4(bp[g] + k0 + 16 * u);
#pragma unroll
    for (int u = 0; u < UK; ++u)
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[u].x, wb[g][u].x, acc[g], 0, 0, 0);
        acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[u].y, wb[g][u].y, acc[g], 0, 0, 0);
        acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[u].z, wb[g][u].z, acc[g], 0, 0, 0);
        acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[u].w, wb[g][u].w, acc[g], 0, 0, 0);
      }
  }
  for (; k0 < kfull; k0 += 16) {
    const float4 xa = ld4(ap + k0);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const float4 wb = ld4(bp[g] + k0);
      acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa.x, wb.x, acc[g], 0, 0, 0);
      acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa.y, wb.y, acc[g], 0, 0, 0);
      acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa.z, wb.z, acc[g], 0, 0, 0);
      acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa.w, wb.w, acc[g], 0, 0, 0);
    }
  }
  if (kfull_all < K && ks == KS - 1) {   // ragged tail (the last wave's): element-wise guards on both operands (row strides are
    const int kt = kfull_all, k = kt + 4 * q;           // multiples of 4 >= K: a chunk that starts below K lies inside the row)
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 xa = z4;
    if (k < K) {
      const float4 t_ = ld4(ap + kt);
      xa.x = t_.x; if (k + 1 < K) xa.y = t_.y; if (k + 2 < K) xa.z = t_.z; if (k + 3 < K) xa.w = t_.w;
    }
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      float4 wb = z4;
      if (k < K) {
        const float4 t_ = ld4(bp[g] + kt);
        wb.x = t_.x; if (k + 1 < K) wb.y = t_.y; if (k + 2 < K) wb.z = t_.z; if (k + 3 < K) wb.w = t_.w;
      }
      acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa.x, wb.x, acc[g], 0, 0, 0);
      acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa.y, wb.y, acc[g], 0, 0, 0);
      acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa.z, wb.z, acc[g], 0, 0, 0);
      acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa.w, wb.w, acc[g], 0, 0, 0);
    }
  }
  __shared__ f32x4 red[(KS - 1) * NG * 64];
  if (ks > 0) {
#pragma unroll
    for (int g = 0; g < NG; ++g) red[((ks - 1) * NG + g) * 64 + lane] = acc[g];
  }
  __syncthreads();
  if (ks > 0) return;
#pragma unroll
  for (int w = 1; w < KS; ++w)
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const f32x4 p_ = red[((w - 1) * NG + g) * 64 + lane];
      acc[g][0] += p_[0]; acc[g][1] += p_[1]; acc[g][2] += p_[2]; acc[g][3] += p_[3];
    }
  // C / D layout of 16x16x4: col = lane & 15 (+ 16 g), row = (lane >> 4) * 4 + reg
  float sl[4] = {0.f, 0.f, 0.f, 0.f}, sr[4] = {0.f, 0.f, 0.f, 0.f};
  const bool want_s = score_out != nullptr && col0 < score_cols;
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const int col = col0 + 16 * g + r;
    const float bcol = (bias && cvv[g]) ? bias[col] : 0.f;
    const float al = (want_s && cvv[g]) ? score_l[col] : 0.f, ar = (want_s && cvv[g]) ? score_r[col] : 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float v = acc[g][j] + bcol;
      if (want_s) { sl[j] = fmaf(v, al, sl[j]); sr[j] = fmaf(v, ar, sr[j]); }      // (score layers carry no bias / activation here)
      v = act_fwd(v, act);
      const int64_t orow = row0 + q * 4 + j;
      if (cvv[g] && orow < M) C[orow * ldc + col] = v;
    }
  }
  if (want_s) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float a_ = team_sum(sl[j], 16), b_ = team_sum(sr[j], 16);
      const int64_t orow = row0 + q * 4 + j;
      if (r == 0 && orow < M) {
        float* o = score_out + (orow * (score_cols / 64) + blockIdx.y) * 2;
        o[0] = a_; o[1] = b_;
      }
    }
  }
}

// one wave per 256 columns x one row range (four waves of a block side by side: a row is then read as one 4 KB
// run instead of 1 KB pieces at different times - DRAM-friendlier); partial sums per range, reduced by the caller
template <typename ST, int J>
__device__ __forceinline__ void scores_bwd_w_body(const float* __restrict__ gS, int64_t ldg,
                                                  const ST* __restrict__ X, int64_t ldx,
                                                  float* __restrict__ part, int Kp, int64_t N, int K,
                                                  int64_t rows_per_split, int jn, const unsigned bx, const unsigned by) {
  // a block's waves sit side by side on one row: 4 KB contiguous per row.  Every lane of a wave stays in the loop (the gS
  // rows travel through lanes, see below): lanes past the width read column 0 and store nothing; a float4 that hangs over
  // the width reads the row's padding (rows are 16-byte aligned with a stride that is a multiple of 4) and the columns
  // that do not exist are cleared before the store.
  const int k = (bx * blockDim.x + threadIdx.x) * 4;
  if ((k & ~255) >= K) return;         // a wave with no column at all (wave-uniform)
  const bool live = k < K;
  const int kk = live ? k : 0;
  const int64_t n0 = (int64_t)by * rows_per_split;
  const int64_t n1 = n0 + rows_per_split < N ? n0 + rows_per_split : N;
  float4 acc[J];
#pragma unroll
  for (int j = 0; j < J; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  int64_t n = n0;
  if (n + 4 <= n1) {                   // 4 rows per trip, the next trip's rows already in flight while this one is summed
    const ST* xr = X + n * ldx + kk;
    float4 x0 = ldv(xr), x1 = ldv(xr + ldx), x2 = ldv(xr + 2 * ldx), x3 = ldv(xr + 3 * ldx);
    // the four gS rows of a trip arrive as two vector loads issued with the x rows (lanes 0-31: one row, lanes 32-63: the
    // next; J <= 32) and are broadcast by v_readlane: as scalar loads at the point of use they were waited for in every trip
    const int gl = threadIdx.x & 63;
    const float* gp = gS + (int64_t)(gl >> 5) * ldg + ((gl & 31) < jn ? (gl & 31) : 0);
    float ga = gp[n * ldg], gb = gp[(n + 2) * ldg];
#define SPGNN_SBW_FMA(C0, C1, C2, C3, GA, GB)                                                                  \
    {                                                                                                          \
      _Pragma("unroll") for (int j = 0; j < J; ++j) {                                                          \
        if (j < jn) {                                                                                          \
          fma4(acc[j], __int_as_float(__builtin_amdgcn_readlane(__float_as_int(GA), j)), C0);                  \
          fma4(acc[j], __int_as_float(__builtin_amdgcn_readlane(__float_as_int(GA), 32 + j)), C1);             \
          fma4(acc[j], __int_as_float(__builtin_amdgcn_readlane(__float_as_int(GB), j)), C2);                  \
          fma4(acc[j], __int_as_float(__builtin_amdgcn_readlane(__float_as_int(GB), 32 + j)), C3);             \
        }                                                                                                      \
      }                                                                                                        \
    }
    for (; n + 8 <= n1; n += 4) {      // steady state: the prefetch is unconditional (a test around it made hipcc drain the queue)
      const float4 c0 = x0, c1 = x1, c2 = x2, c3 = x3;
      const float ca = ga, cb = gb;
      const ST* xn = X + (n + 4) * ldx + kk;
      x0 = ldv(xn); x1 = ldv(xn + ldx); x2 = ldv(xn + 2 * ldx); x3 = ldv(xn + 3 * ldx);
      ga = gp[(n + 4) * ldg]; gb = gp[(n + 6) * ldg];
      SPGNN_SBW_FMA(c0, c1, c2, c3, ca, cb)
    }
    SPGNN_SBW_FMA(x0, x1, x2, x3, ga, gb)      // last full trip
    n += 4;
#undef SPGNN_SBW_FMA
  }
  for (; n < n1; ++n) {
    const float4 x = ldv(X + n * ldx + kk);
    const float* g = gS + n * ldg;
#pragma unroll
    for (int j = 0; j < J; ++j)
      if (j < jn) fma4(acc[j], g[j], x);
  }
  if (!live) return;
#pragma unroll
  for (int j = 0; j < J; ++j)
    if (j < jn) {
      float4 q = acc[j];
      q.y = k + 1 < K ? q.y : 0.f; q.z = k + 2 < K ? q.z : 0.f; q.w = k + 3 < K ? q.w : 0.f;
      st4(part + ((int64_t)by * jn + j) * Kp + k, q);
    }
}

template <typename ST, int J>
__global__ __launch_bounds__(256) void scores_bwd_w_kernel(const float* __restrict__ gS, int64_t ldg,
                                                          const ST* __restrict__ X, int64_t ldx,
                                                          float* __restrict__ part, int Kp, int64_t N, int K,
                                                          int64_t rows_per_split, int jn) {
  scores_bwd_w_body<ST, J>(gS, ldg, X, ldx, part, Kp, N, K, rows_per_split, jn, blockIdx.x, blockIdx.y);
}

// Two independent passes in one launch (a level's structure and position layers: the attention-vector gradients of both,
// reference models.py:472-484): row ranges [0, nb0) of the grid's y run the first, the rest the second, each with the
// arithmetic it has alone (a wave with no column of its problem leaves at once).
struct ScoresBwdW { const float* gS; int64_t ldg; const float* X; int64_t ldx; float* part; int Kp; int64_t N; int K; int64_t rps; int jn; int gx; };
template <int J>
__global__ __launch_bounds__(256) void scores_bwd_w_pair_kernel(ScoresBwdW p0, ScoresBwdW p1, unsigned nb0) {
  const bool second = blockIdx.y >= nb0;
  const ScoresBwdW& p = second ? p1 : p0;
  if ((int)blockIdx.x >= p.gx) return;
  scores_bwd_w_body<float, J>(p.gS, p.ldg, p.X, p.ldx, p.part, p.Kp, p.N, p.K, p.rps, p.jn, blockIdx.x, second ? blockIdx.y - nb0 : blockIdx.y);
}

// Up to eight such passes in one launch (spgnn_scores_bwd_w_multi): every GATConv of a model leaves one of these latency-bound
// passes in its backward (75-150 rows per wave in a dependent chain, a quarter of the chip busy), and none of them feeds
// anything but its own attention vectors' gradient - so a training step collects them and runs them side by side once the
// backward pass is through (ops.AttnGradQueue).  Row ranges [y0[k], y0[k+1]) of the grid's y run job k, with the arithmetic
// it has in a launch of its own.
constexpr int kMaxScoreJobs = 8;
struct ScoresBwdWJobs { ScoresBwdW j[kMaxScoreJobs]; unsigned y0[kMaxScoreJobs + 1]; int n; };
template <typename ST, int J>
__global__ __launch_bounds__(256) void scores_bwd_w_multi_kernel(ScoresBwdWJobs a) {
  int k = 0;
  while (k + 1 < a.n && blockIdx.y >= a.y0[k + 1]) ++k;
  const ScoresBwdW& p = a.j[k];
  if ((int)blockIdx.x >= p.gx) return;
  scores_bwd_w_body<ST, J>(p.gS, p.ldg, reinterpret_cast<const ST*>(p.X), p.ldx, p.part, p.Kp, p.N, p.K, p.rps, p.jn, blockIdx.x,
                           blockIdx.y - a.y0[k]);
}

template <typename ST, int J>     // ST: storage type of the rows of gX (gS, W fp32)
__global__ __launch_bounds__(64) void scores_bwd_x_kernel(const float* __restrict__ gS, int64_t ldg,
                                                          const float* __restrict__ W, int Kp,
                                                          ST* __restrict__ gX, int64_t ldgx, int64_t N, int K,
                                                          int64_t rows_per_split, int jn, int accumulate) {
  const int k = (blockIdx.x * blockDim.x + threadIdx.x) * 4;      // a block's waves sit side by side on one row: 4 KB contiguous per row
  if ((k & ~255) >= K) return;          // a wave with no column at all (wave-uniform)
  // every lane of a live wave stays in the loop: the gS rows travel through lanes (below); lanes past the width read
  // column 0 of W and store nothing
  const bool live = k < K, full = k + 3 < K;
  const int kk = live ? k : 0;
  const int64_t n0 = (int64_t)blockIdx.y * rows_per_split;
  const int64_t n1 = n0 + rows_per_split < N ? n0 + rows_per_split : N;
  if (n0 >= n1) return;
  float4 w[J];
#pragma unroll
  for (int j = 0; j < J; ++j) w[j] = j < jn ? ld4(W + (int64_t)j * Kp + kk) : make_float4(0.f, 0.f, 0.f, 0.f);
  // Two gS rows arrive as ONE vector load (lanes 0-31: row n, lanes 32-63: row n + 1; J <= 32), issued one pair ahead,
  // and are broadcast by v_readlane instead of J scalar loads per row at the point of use.  Measured: K = 384, J = 22
  // 61 -> 58 us, K = 1024 118 -> 116 us - the kernel is bound by its J fma4 per row and lane (3.4 GFLOP at K = 1024:
  // ~30 TFLOP/s of plain fp32 VALU) and its stores, not by those loads; the fp32 MFMA form of spgnn_scores_fwd is the next step.
  const int gl = threadIdx.x & 63;
  const int gj = (gl & 31) < jn ? (gl & 31) : 0, gh = gl >> 5;
  const int64_t last = n1 - 1;
  auto grow = [&](int64_t n) { const int64_t r = n + gh; return gS[(r < last ? r : last) * ldg + gj]; };
  float ga = grow(n0);
  for (int64_t n = n0; n < n1; n += 2) {
    const float gnext = grow(n + 2 < n1 ? n + 2 : last);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int64_t row = n + half;
      if (row >= n1) break;             // wave-uniform
      float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int j = 0; j < J; ++j)
        if (j < jn) fma4(d, __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ga), 32 * half + j)), w[j]);
      if (live) {
        ST* xr = gX + row * ldgx + k;
        if constexpr (is_f32<ST>::value) {
          if (accumulate) {
            if (full) { float4 o = ld4(xr); o.x += d.x; o.y += d.y; o.z += d.z; o.w += d.w; st4(xr, o); }
            else { xr[0] += d.x; if (k + 1 < K) xr[1] += d.y; if (k + 2 < K) xr[2] += d.z; }
          } else {
            if (full) st4(xr, d);
            else { xr[0] = d.x; if (k + 1 < K) xr[1] = d.y; if (k + 2 < K) xr[2] = d.z; }
          }
        } else {                        // bf16 rows: whole 4-element chunks (the row padding up to a multiple of 4 is zero filled)
          if (accumulate) { const float4 o = ldv(xr); d.x += o.x; d.y += o.y; d.z += o.z; d.w += o.w; }
          d.y = k + 1 < K ? d.y : 0.f; d.z = k + 2 < K ? d.z : 0.f; d.w = k + 3 < K ? d.w : 0.f;
          stv(xr, d);
        }
      }
    }
    ga = gnext;
  }
}


// =================================================================================================
// Distance positional encoding on the device (reference job_runner.py:1759-1777):
//   pos_enc[t, a] = hop_distance(t, anchor_a) / diameter(tree of t)
// The reference runs networkx all-pairs shortest paths per tree on the host.  Here one workgroup takes one
// tree: its 4 waves run independent level-synchronous BFS passes (one anchor per wave at a time) over the
// tree's CSR rows with the distance array of each pass in LDS; the diameter comes from a double sweep
// (exact on trees).  The division is done in double and rounded once to float, like the reference's
// `dist / float(diameter)` stored into a float32 array, so results are bit-identical.
// =================================================================================================
constexpr int kPeMaxNodes = 2048;           // nodes per tree held in LDS (airway trees: 100-300)

__device__ __forceinline__ int wave_bfs(const int32_t* __restrict__ indptr, const int32_t* __restrict__ indices,
                                        int64_t base, int n, int src_local, volatile int* dist, int lane) {
  for (int i = lane; i < n; i += 64) dist[i] = -1;
  if (lane == 0) dist[src_local] = 0;
  int level = 0, far = src_local;
  for (;;) {                                             // every wave iteration is one BFS level
    bool grew = false;
    for (int i = lane; i < n; i += 64) {
      if (dist[i] == level) {
        for (int k = indptr[base + i]; k < indptr[base + i + 1]; ++k) {
          const int j = (int)(indices[k] - base);
          if (dist[j] < 0) { dist[j] = level + 1; grew = true; far = j; }
        }
      }
    }
    if (!__any(grew)) break;
    ++level;
  }
  // a node of the last level (for the double sweep): take the largest lane's candidate at that level
  int cand = -1;
  for (int i = lane; i < n; i += 64) if (dist[i] == level) cand = i;
  for (int off = 32; off > 0; off >>= 1) cand = max(cand, __shfl_xor(cand, off, 64));
  (void)far;
  return (level << 16) | (cand & 0xFFFF);                // [eccentricity | one farthest node]
}

__global__ __launch_bounds__(256) void tree_distance_encoding(const int32_t* __restrict__ indptr,
                                                              const int32_t* __restrict__ indices,
                                                              const int64_t* __restrict__ tree_ptr,
                                                              const int32_t* __restrict__ anchors, int A,
                                                              float* __restrict__ pe, int64_t pe_ld,
                                                              int32_t* __restrict__ diam_out) {
  __shared__ int dist_s[4][kPeMaxNodes];
  __shared__ int diam_s;
  const int t = blockIdx.x;
  const int64_t base = tree_ptr[t];
  const int n = (int)(tree_ptr[t + 1] - base);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (wave == 0) {
    const int r0 = wave_bfs(indptr, indices, base, n, 0, dist_s[0], lane);
    const int r1 = wave_bfs(indptr, indices, base, n, r0 & 0xFFFF, dist_s[0], lane);
    if (lane == 0) { diam_s = r1 >> 16; if (diam_out) diam_out[t] = r1 >> 16; }
  }
  __syncthreads();
  const double diam = (double)diam_s;
  for (int a = wave; a < A; a += 4) {
    const int src = (int)(anchors[(int64_t)t * A + a] - base);
    wave_bfs(indptr, indices, base, n, src, dist_s[wave], lane);
    for (int i = lane; i < n; i += 64)
      pe[(base + i) * pe_ld + a] = (float)((double)dist_s[wave][i] / diam);
  }
}

// =================================================================================================
// Anchor selection of the distance positional encoding on the device (reference
// job_runner.py:1727-1757 get_anchors_from_cnn_prediction and 1712-1725 add_distal_leafs), one batch per call.
//
// (1) greedy_anchors: per tree, for label = 1 .. num_labels: index = argmax_i prob[i, label] * mask[i] (first
//     occurrence of the maximum, as numpy.argmax), then mask[index] = 0.  One wave per tree; `prob` is the softmax
//     of the CNN logits (the reference forms it with torch on the device too, job_runner.py:1730).
// (2) distal_leafs: for each of the first num_distal anchors, its farthest descendant leaf in the "downstream" DAG
//     (edges u -> v with v > u: numpy.triu of the adjacency), the anchor itself when it has no child.  The reference
//     picks `sorted(leafs.items(), key=distance)[-1]` where `leafs` is a dict filled while iterating the Python SET
//     nx.descendants() returns: among leaves at the maximal distance the winner is the LAST one in that set's
//     iteration order, i.e. in hash-table slot order.  To return the same node, the kernel replays CPython's set of
//     small ints exactly (Objects/setobject.c: hash(i) = i, 8 initial slots, linear probes of 9 then
//     i = 5 i + 1 + perturb, growth to the next power of two above 4 * used when fill * 5 >= mask * 3, re-insertion in
//     slot order) with the descendants inserted in BFS discovery order (children ascending), as networkx builds it.
//     One thread per (tree, anchor); the table, the BFS queue and the distances live in a caller-provided workspace.
// =================================================================================================
__global__ __launch_bounds__(64) void greedy_anchors_kernel(const float* __restrict__ prob, int64_t ld, const int64_t* __restrict__ tree_ptr,
                                                            int32_t* __restrict__ anchors, int A, int num_labels,
                                                            uint8_t* __restrict__ taken) {
  const int t = blockIdx.x, lane = threadIdx.x;
  const int64_t base = tree_ptr[t];
  const int n = (int)(tree_ptr[t + 1] - base);
  for (int i = lane; i < n; i += 64) taken[base + i] = 0;
  __syncthreads();
  for (int label = 1; label <= num_labels; ++label) {
    double best = -1.0; int arg = 0x7FFFFFFF;
    for (int i = lane; i < n; i += 64) {
      const double v = taken[base + i] ? 0.0 : (double)prob[(base + i) * ld + label];
      if (v > best) { best = v; arg = i; }                     // ascending i within a lane: first occurrence kept
    }
    for (int off = 32; off > 0; off >>= 1) {
      const double ob = __shfl_xor(best, off, 64); const int oa = __shfl_xor(arg, off, 64);
      if (ob > best || (ob == best && oa < arg)) { best = ob; arg = oa; }
    }
    if (lane == 0) { anchors[(int64_t)t * A + (label - 1)] = (int32_t)(base + arg); taken[base + arg] = 1; }
    __syncthreads();
  }
}

__device__ __forceinline__ void pyset_insert_clean(uint16_t* table, unsigned mask, unsigned key) {
  unsigned perturb = key, i = key & mask;
  while (true) {
    if (table[i] == 0xFFFFu) { table[i] = (uint16_t)key; return; }
    if (i + 9 <= mask) {
      for (unsigned j = 1; j <= 9; ++j)
        if (table[i + j] == 0xFFFFu) { table[i + j] = (uint16_t)key; return; }
    }
    perturb >>= 5;
    i = (i * 5 + 1 + perturb) & mask;
  }
}

__global__ __launch_bounds__(64) void distal_leafs_kernel(const int32_t* __restrict__ indptr, const int32_t* __restrict__ indices,
                                                          const int64_t* __restrict__ tree_ptr, int32_t* __restrict__ anchors,
                                                          int A, int first_extra, int num_distal, int64_t num_trees,
                                                          uint16_t* __restrict__ ws, int64_t ws_per_thread, int n_cap) {
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= num_trees * num_distal) return;
  const int64_t t = tid / num_distal; const int k = (int)(tid % num_distal);
  const int64_t base = tree_ptr[t];
  const int n = (int)(tree_ptr[t + 1] - base);
  uint16_t* dist = ws + tid * ws_per_thread;          // [n_cap] hop distance from the anchor (0xFFFF: not reached)
  uint16_t* queue = dist + n_cap;                     // [n_cap] BFS queue = discovery order
  uint16_t* tab_a = queue + n_cap;                    // [8 n_cap] the emulated set's table ...
  uint16_t* tab_b = tab_a + 8 * n_cap;                // [8 n_cap] ... and the one it is rebuilt into on growth
  const int a = (int)(anchors[t * A + k] - base);
  for (int i = 0; i < n; ++i) dist[i] = 0xFFFFu;
  unsigned mask = 7, fill = 0;
  uint16_t* table = tab_a;
  for (unsigned i = 0; i <= mask; ++i) table[i] = 0xFFFFu;
  int head = 0, tail = 0;
  dist[a] = 0; queue[tail++] = (uint16_t)a;
  while (head < tail) {
    const int p = queue[head++];
    const int64_t gp = base + p;
    for (int j = indptr[gp]; j < indptr[gp + 1]; ++j) {        // out-neighbours ascending; children = those with a larger id
      const int c = (int)(indices[j] - base);
      if (c <= p || dist[c] != 0xFFFFu) continue;
      dist[c] = (uint16_t)(dist[p] + 1);
      queue[tail++] = (uint16_t)c;
      pyset_insert_clean(table, mask, (unsigned)c);            // a new key: set_add_entry takes the same probe sequence
      ++fill;
      if (fill * 5 >= mask * 3) {                              // set_table_resize(used * 4): next power of two above it
        unsigned newsize = 8;
        while (newsize <= fill * 4) newsize <<= 1;
        uint16_t* nt = table == tab_a ? tab_b : tab_a;
        for (unsigned i = 0; i < newsize; ++i) nt[i] = 0xFFFFu;
        for (unsigned i = 0; i <= mask; ++i)
          if (table[i] != 0xFFFFu) pyset_insert_clean(nt, newsize - 1, table[i]);
        table = nt; mask = newsize - 1;
      }
    }
  }
  int best = a, best_d = -1;
  for (unsigned i = 0; i <= mask; ++i) {                       // set iteration = slot order; stable sort by distance, last wins
    const unsigned c = table[i];
    if (c == 0xFFFFu) continue;
    const int64_t gc = base + c;
    bool leaf = true;
    for (int j = indptr[gc]; j < indptr[gc + 1]; ++j)
      if (indices[j] - base > (int64_t)c) { leaf = false; break; }
    if (leaf && (int)dist[c] >= best_d) { best_d = dist[c]; best = (int)c; }
  }
  anchors[t * A + first_extra + k] = (int32_t)(base + best);
}

// =================================================================================================
// The rows of a step that reach the loss (reference job_runner.py:1896-1900: ``pre[mask]`` - labelled nodes always, the others
// with probability SAMPLING_RATE).  The output layer's projection, the classifier and their backward products are row-wise:
// a row outside the mask influences neither the loss nor any gradient, so a training step may evaluate them on the kept
// rows only.  spgnn_loss_rows lists those rows in ascending node order (the same draw as the loss kernel: `draws`, or the
// counter hash of (draw_seed, step counter, node)), spgnn_gather_rows / spgnn_expand_rows move rows between the node order
// and the list.  Two launches: per-block counts, then every block sums the counts before it and writes its nodes - the
// list's order does not depend on scheduling.
// =================================================================================================
__device__ __forceinline__ bool loss_row_kept(const float* __restrict__ draws, uint64_t sd, const float* __restrict__ p, int64_t i) {
  const float rn = draws ? draws[i] : (float)(uint32_t)(mix64(sd, i) >> 40) * (1.0f / 16777216.0f);
  return rn < p[i];
}

__global__ __launch_bounds__(kBlock) void loss_rows_count_kernel(const float* __restrict__ draws, uint64_t draw_seed,
                                                                 const int64_t* __restrict__ seed_off, const float* __restrict__ p,
                                                                 int64_t N, int32_t* __restrict__ counts) {
  __shared__ int wsum[kBlock / 64];
  const uint64_t sd = draw_seed + (seed_off ? 0xD1B54A32D192ED03ull * (uint64_t)seed_off[0] : 0ull);
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const bool keep = i < N && loss_row_kept(draws, sd, p, i);
  const unsigned long long b = __ballot(keep);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = __popcll(b);
  __syncthreads();
  if (threadIdx.x == 0) {
    int t = 0;
#pragma unroll
    for (int q = 0; q < kBlock / 64; ++q) t += wsum[q];
    counts[blockIdx.x] = t;
  }
}

__global__ __launch_bounds__(kBlock) void loss_rows_write_kernel(const float* __restrict__ draws, uint64_t draw_seed,
                                                                 const int64_t* __restrict__ seed_off, const float* __restrict__ p,
                                                                 int64_t N, const int32_t* __restrict__ counts, int32_t cap,
                                                                 int32_t* __restrict__ idx, int32_t* __restrict__ inv,
                                                                 int32_t* __restrict__ cnt_flag) {
  __shared__ int red[kBlock / 64];
  __shared__ int wsum[kBlock / 64];
  __shared__ int base_s, total_s;
  // counts of the blocks before this one (and of all of them): each thread a strided share, then the block's sum
  int before = 0, all = 0;
  for (unsigned q = threadIdx.x; q < gridDim.x; q += kBlock) {
    const int c = counts[q];
    all += c;
    if (q < blockIdx.x) before += c;
  }
  for (int pass = 0; pass < 2; ++pass) {                        // integer sums: wave shuffles, then the block's waves through LDS
    int v = pass == 0 ? before : all;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
      int t = 0;
#pragma unroll
      for (int q = 0; q < kBlock / 64; ++q) t += red[q];
      if (pass == 0) base_s = t; else total_s = t;
    }
    __syncthreads();
  }
  const uint64_t sd = draw_seed + (seed_off ? 0xD1B54A32D192ED03ull * (uint64_t)seed_off[0] : 0ull);
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const bool keep = i < N && loss_row_kept(draws, sd, p, i);
  const unsigned long long b = __ballot(keep);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) wsum[wave] = __popcll(b);
  __syncthreads();
  int pos = base_s + __popcll(b & ((1ull << lane) - 1ull));
  for (int q = 0; q < wave; ++q) pos += wsum[q];
  if (i < N) {
    const bool in = keep && pos < cap;
    if (in) idx[pos] = (int32_t)i;
    inv[i] = in ? pos : -1;
  }
  if (blockIdx.x == 0) {
    const int total = total_s;
    const int cnt = total < cap ? total : cap;
    if (threadIdx.x == 0) {
      cnt_flag[0] = cnt;
      if (total > cap) cnt_flag[1] = 1;                       // sticky: the host reads it once per loader batch
    }
    for (int c = cnt + (int)threadIdx.x; c < cap; c += kBlock) idx[c] = 0;    // unused slots name a valid node
  }
}

// dst[c, :] = c < cnt ? src[idx[c], :] : 0, c in [0, cap)   (cols4 = columns / 4; rows 16-byte aligned)
__global__ __launch_bounds__(kBlock) void gather_rows_kernel(const float* __restrict__ src, int64_t lds_, const int32_t* __restrict__ idx,
                                                             const int32_t* __restrict__ cnt_flag, int64_t cap, int cols4,
                                                             float* __restrict__ dst, int64_t ldd) {
  const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (e >= cap * cols4) return;
  const int64_t c = e / cols4;
  const int k = (int)(e - c * cols4) * 4;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c < (int64_t)cnt_flag[0]) v = ld4(src + (int64_t)idx[c] * lds_ + k);
  st4(dst + c * ldd + k, v);
}

// dst[n, :] = inv[n] >= 0 ? src[inv[n], :] : 0, n in [0, N)
__global__ __launch_bounds__(kBlock) void expand_rows_kernel(const float* __restrict__ src, int64_t lds_, const int32_t* __restrict__ inv,
                                                             int64_t N, int cols4, float* __restrict__ dst, int64_t ldd) {
  const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (e >= N * cols4) return;
  const int64_t n = e / cols4;
  const int k = (int)(e - n * cols4) * 4;
  const int32_t c = inv[n];
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c >= 0) v = ld4(src + (int64_t)c * lds_ + k);
  st4(dst + n * ldd + k, v);
}

// =================================================================================================
// Masked, class-weighted cross entropy in one pass (reference job_runner.py:1896-1900:
// mask = rn < sampling_t; loss = F.cross_entropy(pre[mask], y[mask], weight=w)).  One thread per node:
//   m_i = rn_i < sampling_p[i];  nll_i = logsumexp(logits[i,:]) - logits[i, y_i]
//   partial[block] = (sum m_i w[y_i] nll_i, sum m_i w[y_i])          (numerator, denominator of the weighted mean)
//   g_logits[i,c]  = m_i w[y_i] (softmax(logits[i,:])_c - [c == y_i])  (gradient of the NUMERATOR)
// No boolean indexing (no host sync), no separate log-softmax / gather / multiply / reduce launches.
// rn_i comes from `draws`, or - `draws` null - from the kernels' counter hash: 24 bits of mix64(seed + offset, i), the
// offset read from device memory so that a captured step draws a fresh mask on every replay (the reference draws the
// GCN_STEPS x N matrix with numpy up front, job_runner.py:1889: neither stream can reproduce the other).
// With `sums` the block that arrives last (ticket counter, reset by it) adds the per-block pairs in block order: the
// two totals without a reduction launch, bitwise independent of the arrival order.
// =================================================================================================
__global__ __launch_bounds__(kBlock) void masked_ce_kernel(const float* __restrict__ logits, int64_t ld, const int64_t* __restrict__ labels,
                                                           const float* __restrict__ draws, uint64_t draw_seed,
                                                           const int64_t* __restrict__ seed_off, const float* __restrict__ sampling_p,
                                                           const float* __restrict__ class_w, float* __restrict__ partial,
                                                           float* __restrict__ sums, unsigned* __restrict__ ticket,
                                                           float* __restrict__ g_logits, int64_t g_ld, float* __restrict__ colpart,
                                                           float* __restrict__ colsum, int64_t N, int C,
                                                           const int32_t* __restrict__ row_list, const int32_t* __restrict__ rows_cnt) {
  // `rows_cnt` alone (spgnn_masked_ce_step_flagged): a dense pass whose weights turn NaN when rows_cnt[1] != 0 - the step's row
  // list overflowed and only its backward pass uses it.
  // `row_list` (spgnn_masked_ce_rows): logits / g_logits hold one row per LISTED node (spgnn_loss_rows: the nodes the mask keeps,
  // rows_cnt[0] of them, rows_cnt[1] != 0 when the list overflowed its capacity): row i belongs to node row_list[i], its mask
  // is i < rows_cnt[0] - no draw here, the list IS the draw - and an overflow turns every weight into NaN
  __shared__ float red[2][kBlock / 64];
  __shared__ bool last;
  // C <= 32 (the 22 airway labels): the block's 256 rows go through LDS, so that the global loads and the gradient stores are
  // contiguous across the block instead of one 88-byte row per lane (39 -> 12 us at 76 410 nodes); odd pitch: conflict-free
  __shared__ float tile[kBlock * 33];
  const bool staged = C <= 32;
  const int P = C | 1;
  const int64_t base = (int64_t)blockIdx.x * kBlock;
  const int rows = (int)(N - base < kBlock ? N - base : kBlock);
  if (staged) {
    for (int e = threadIdx.x; e < rows * C; e += kBlock) {
      const int r = e / C, c = e - r * C;
      tile[r * P + c] = logits[(base + r) * ld + c];
    }
    __syncthreads();
  }
  const int64_t i = base + threadIdx.x;
  float num = 0.f, den = 0.f;
  if (i < N) {
    const float* row = staged ? tile + threadIdx.x * P : logits + i * ld;
    const bool listed = row_list != nullptr && i < (int64_t)rows_cnt[0];
    const int64_t yl = row_list ? (listed ? labels[row_list[i]] : 0) : labels[i];
    const bool y_ok = yl >= 0 && yl < C && !(rows_cnt && rows_cnt[1] != 0);   // F.cross_entropy raises for such a label; here the node
    const int y = y_ok ? (int)yl : 0;                        // gets a NaN weight, so the loss is NaN instead of an out-of-bounds read
    float m;
    if (row_list) m = listed ? 1.f : 0.f;
    else {
      float rn;
      if (draws) rn = draws[i];
      else {
        const uint64_t sd = draw_seed + (seed_off ? 0xD1B54A32D192ED03ull * (uint64_t)seed_off[0] : 0ull);
        rn = (float)(uint32_t)(mix64(sd, i) >> 40) * (1.0f / 16777216.0f);
      }
      m = rn < sampling_p[i] ? 1.f : 0.f;
    }
    const float w = y_ok ? m * class_w[y] : NAN;
    float mx = -INFINITY;
    for (int c = 0; c < C; ++c) mx = fmaxf(mx, row[c]);
    float se = 0.f;
    for (int c = 0; c < C; ++c) se += expf(row[c] - mx);
    const float lse = mx + logf(se);
    num = w * (lse - row[y]);
    den = w;
    if (g_logits) {
      float* g = staged ? tile + threadIdx.x * P : g_logits + i * g_ld;      // staged: in place, the row is this lane's own
      const float inv = 1.f / se;
      for (int c = 0; c < C; ++c) g[c] = w * (expf(row[c] - mx) * inv - (c == y ? 1.f : 0.f));
    }
  }
  if (staged && g_logits) {
    __syncthreads();
    for (int e = threadIdx.x; e < rows * C; e += kBlock) {
      const int r = e / C, c = e - r * C;
      g_logits[(base + r) * g_ld + c] = tile[r * P + c];
    }
    if (colpart && threadIdx.x < C) {                        // column sums of the block's gradient rows (the classifier bias' gradient)
      float cs_ = 0.f;
      for (int r = 0; r < rows; ++r) cs_ += tile[r * P + threadIdx.x];
      __hip_atomic_store(colpart + (int64_t)blockIdx.x * 32 + threadIdx.x, cs_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  num = team_sum(num, 64); den = team_sum(den, 64);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = num; red[1][threadIdx.x >> 6] = den; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int q = 0; q < kBlock / 64; ++q) { a += red[0][q]; b += red[1][q]; }
    if (sums) {
      // The pair goes out as two device-scope atomic stores (write-through past this XCD's L2) and is waited for before the
      // ticket is taken; the block that arrives last reads all pairs with device-scope atomic loads.  A __threadfence()
      // here instead is an L2 write-back per workgroup (the block's gradient rows are dirty in it): +10 us on 299 blocks.
      __hip_atomic_store(partial + 2 * blockIdx.x, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(partial + 2 * blockIdx.x + 1, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
    } else {
      partial[2 * blockIdx.x] = a; partial[2 * blockIdx.x + 1] = b;
    }
  }
  if (!sums) return;
  __syncthreads();
  if (!last) return;
  if (colpart && colsum && staged && g_logits) {             // fixed order: bitwise independent of the arrival order
    // thread (column c, group g of 8): blocks g, g + 8, ... - four loads in flight (one thread per column walking all blocks
    // was 299 dependent device-scope loads: 90 us); then the eight groups in order
    const int c = threadIdx.x & 31, g8 = threadIdx.x >> 5;
    float cs_ = 0.f;
    unsigned q = g8;
    for (; q + 24 < gridDim.x; q += 32) {
      const float v0 = __hip_atomic_load(colpart + (int64_t)q * 32 + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const float v1 = __hip_atomic_load(colpart + (int64_t)(q + 8) * 32 + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const float v2 = __hip_atomic_load(colpart + (int64_t)(q + 16) * 32 + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const float v3 = __hip_atomic_load(colpart + (int64_t)(q + 24) * 32 + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      cs_ += v0; cs_ += v1; cs_ += v2; cs_ += v3;
    }
    for (; q < gridDim.x; q += 8) cs_ += __hip_atomic_load(colpart + (int64_t)q * 32 + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    tile[g8 * 32 + c] = cs_;
    __syncthreads();
    if (threadIdx.x < C) {
      float t_ = 0.f;
#pragma unroll
      for (int g = 0; g < kBlock / 32; ++g) t_ += tile[g * 32 + threadIdx.x];
      colsum[threadIdx.x] = t_;
    }
  }
  float a = 0.f, b = 0.f;                                    // thread t: blocks t, t + 256, ... in ascending order
  for (unsigned q = threadIdx.x; q < gridDim.x; q += kBlock) {
    a += __hip_atomic_load(partial + 2 * q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    b += __hip_atomic_load(partial + 2 * q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  a = team_sum(a, 64); b = team_sum(b, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a; red[1][threadIdx.x >> 6] = b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float x = 0.f, y = 0.f;
#pragma unroll
    for (int q = 0; q < kBlock / 64; ++q) { x += red[0][q]; y += red[1][q]; }
    sums[0] = x; sums[1] = y;
    *ticket = 0u;                                            // re-armed for the next launch on the stream
  }
}

// =================================================================================================
// SGD + momentum over a flat bucket
// =================================================================================================
__global__ void sgd_momentum_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf,
                                    const float* __restrict__ gscale, const float* __restrict__ gdenom,
                                    const float* __restrict__ loss_num, float* __restrict__ loss_out,
                                    const float* __restrict__ lr_dev, int64_t n, float lr, float mom, float wd, int first,
                                    unsigned* __restrict__ skipped) {
  const float sc = gdenom ? 1.f / gdenom[0] : (gscale ? gscale[0] : 1.f);
  if (loss_out && blockIdx.x == 0 && threadIdx.x == 0) loss_out[0] = loss_num[0] * sc;
  // guarded form (spgnn_sgd_momentum_step_guarded): a step whose loss is not finite is NOT applied - parameters and momentum
  // stay as they are, the counter goes up by one; the loss scalar above still reports the NaN.  Every rank sees the same
  // all-reduced loss numerator, so every rank skips the same steps.
  if (skipped && loss_num && !isfinite(loss_num[0] * sc)) {
    if (blockIdx.x == 0 && threadIdx.x == 0) skipped[0] += 1u;
    return;
  }
  if (lr_dev) lr = lr_dev[0];
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float w = p[i];
    const float gi = fmaf(wd, w, g[i] * sc);
    const float b = first ? gi : fmaf(mom, buf[i], gi);
    buf[i] = b;
    p[i] = w - lr * b;
  }
}

// The device state a training step arms before its first kernel, in one launch: the dropout / mask counter advances and
// every scale block of the step's pool returns to {-256, 0, 0, 0, 0 x 256} (spgnn_internal.h).
// Before a block is re-armed its range flag (header word 1, set by a GEMM that found rows of the operand outside the split's
// 2^18 envelope: spgnn_internal.h load_scale_monitored) is added to `violations` - a sticky device counter the host polls
// when it likes (integer adds: order-independent).
__global__ __launch_bounds__(kBlock) void step_begin_kernel(int64_t* __restrict__ counter, float* __restrict__ blocks, int64_t words,
                                                            unsigned* __restrict__ violations) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i == 0 && counter) counter[0] += 1;
  constexpr int W = spgnn_detail::kScaleHeader + spgnn_detail::kScaleSlots;
  if (i < words) {
    if (violations && i % W == 1 && blocks[i] != 0.f) atomicAdd(violations, 1u);
    blocks[i] = (i % W == 0) ? -(float)spgnn_detail::kScaleSlots : 0.f;
  }
}

// T < 64 comes with R = 1 (pick_team tries 64 lanes first) or, for the SpMM kernels' narrow rows, as 16 lanes x R = 2 / 4
constexpr bool kSpmmNarrow = true;   // SpMM kernels: rows of <= 256 floats on 16-lane teams (pick_team's `narrow`)
#define DISPATCH_R(T_, R_, KERNEL, ...)                                                          \
  if ((T_) != 64 && (R_) == 2) { hipLaunchKernelGGL((KERNEL<0, 2>), __VA_ARGS__); }               \
  else if ((T_) != 64 && (R_) == 4) { hipLaunchKernelGGL((KERNEL<0, 4>), __VA_ARGS__); }          \
  else if ((T_) != 64) { hipLaunchKernelGGL((KERNEL<0, 1>), __VA_ARGS__); }                      \
  else switch (R_) {                                                                             \
    case 1: hipLaunchKernelGGL((KERNEL<64, 1>), __VA_ARGS__); break;                             \
    case 2: hipLaunchKernelGGL((KERNEL<64, 2>), __VA_ARGS__); break;                             \
    case 4: hipLaunchKernelGGL((KERNEL<64, 4>), __VA_ARGS__); break;                             \
    default: hipLaunchKernelGGL((KERNEL<64, 8>), __VA_ARGS__); break;                            \
  }

// =================================================================================================
// Neighbour sampling on the device-resident CSC (dgl.sampling.sample_neighbors + dgl.to_block as the reference's
// sampled GraphSAGE loop uses them, job_runner.py:1484-1499).  Index work, one thread per seed / node / edge.
//
// sample: seed s (parent node v) keeps k = min(deg(v), fanout) of its in-edges, chosen uniformly without replacement
// by selection sampling (Knuth 3.4.2 S): walking the in-edges in CSC order, edge i of d is taken with probability
// (k - taken) / (d - i).  The draw for CSC slot j is the top 32 bits of mix64(seed, j) mapped to [0, d - i) by
// multiply-shift, so a sample is a pure function of (seed, graph, seeds) and the kept edges stay in CSC (= ascending
// parent edge id) order.  Sources that are not seeds themselves are flagged for the block's relabelling.
// =================================================================================================
__global__ __launch_bounds__(kBlock) void block_mark_seeds(const int64_t* __restrict__ seeds, int64_t S, int64_t N,
                                                           int32_t* __restrict__ local) {
  const int64_t s = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (s >= S) return;
  const int64_t v = seeds[s];
  if (v >= 0 && v < N) local[v] = (int32_t)s;       // an id outside the graph is never dereferenced
}

__global__ __launch_bounds__(kBlock) void sample_neighbors_kernel(const int32_t* __restrict__ indptr, const int32_t* __restrict__ indices,
                                                                  const int32_t* __restrict__ eid, const int64_t* __restrict__ seeds,
                                                                  int64_t S, int64_t N, int32_t fanout,
                                                                  const int32_t* __restrict__ out_indptr, uint64_t seed,
                                                                  const int32_t* __restrict__ local,
                                                                  int32_t* __restrict__ out_src, int32_t* __restrict__ out_eid,
                                                                  int32_t* __restrict__ flag) {
  const int64_t s = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (s >= S) return;
  const int64_t v = seeds[s];
  if (v < 0 || v >= N) return;
  const int32_t b = indptr[v], d = indptr[v + 1] - b;
  const int32_t k = (fanout < 0 || fanout > d) ? d : fanout;
  const int32_t o = out_indptr[s];
  int32_t m = 0;
  for (int32_t i = 0; i < d && m < k; ++i) {
    const uint32_t left = (uint32_t)(d - i), need = (uint32_t)(k - m);
    bool take = need >= left;
    if (!take) {
      const uint64_t r = mix64(seed, (int64_t)b + i) >> 32;
      take = (uint32_t)((r * (uint64_t)left) >> 32) < need;
    }
    if (take) {
      const int32_t u = indices[b + i];
      out_src[o + m] = u;
      if (out_eid) out_eid[o + m] = eid ? eid[b + i] : b + i;
      if (local[u] < 0) flag[u] = 1;                 // racing writers all store 1
      ++m;
    }
  }
}

// relabel: a flagged parent node v becomes block source S + rank[v] - 1 (rank = inclusive prefix sum of flag, i.e.
// the extra sources follow the seeds in ascending parent id); then every sampled edge's source is rewritten.
__global__ __launch_bounds__(kBlock) void block_number_sources(const int32_t* __restrict__ flag, const int32_t* __restrict__ rank,
                                                               int32_t* __restrict__ local, int64_t* __restrict__ extra_nodes,
                                                               int64_t N, int64_t S) {
  const int64_t v = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (v < N && flag[v]) {
    const int32_t r = rank[v] - 1;
    local[v] = (int32_t)S + r;
    extra_nodes[r] = v;
  }
}

__global__ __launch_bounds__(kBlock) void block_relabel_edges(const int32_t* __restrict__ local, const int32_t* __restrict__ out_src,
                                                              int32_t* __restrict__ src_local, int64_t E) {
  const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (e < E) src_local[e] = local[out_src[e]];
}

inline unsigned scalar_grid(int64_t total) { return (unsigned)((total + kBlock - 1) / kBlock); }

}  // namespace

// =================================================================================================
// C ABI
// =================================================================================================
extern "C" {

int spgnn_abi_version(void) { return SPGNN_ABI_VERSION; }
const char* spgnn_last_error(void) { return g_err; }

// Geometry of the vector GAT kernels for (H, D): team width T, chunks per lane R, chunks per head CH
// (0 = heads narrower than the team, reduction width W).  false -> scalar fallback.
// One node per team.  Measured (tools/npt_sweep.py, MI355X): looping a team over 2/4/8/16 consecutive nodes was
// 2/8/24/50 % slower - many short-lived workgroups hide the dependent index -> score -> row chain better.
static bool pick_gat(int H, int D, int& T, int& R, int& CH, int& W) {
  if (D % 4) return false;
  if (!pick_team((int64_t)H * D, T, R, H == 1 || H == 2)) return false;
  const int team_floats = 4 * T;
  W = T;
  if (D % team_floats == 0) {
    CH = D / team_floats;
    return (CH == 1 || CH == 2 || CH == 4 || CH == 8) && R % CH == 0;
  }
  if (team_floats % D == 0 && ((D / 4) & (D / 4 - 1)) == 0) { CH = 0; W = D / 4; return true; }
  return false;
}

#define SPGNN_FOR_R_CH(R_, CH_, X)                                                                  \
  switch ((R_) * 16 + (CH_)) {                                                                      \
    case 1 * 16 + 0: X(1, 0); break;  case 1 * 16 + 1: X(1, 1); break;                              \
    case 2 * 16 + 0: X(2, 0); break;  case 2 * 16 + 1: X(2, 1); break;  case 2 * 16 + 2: X(2, 2); break; \
    case 4 * 16 + 0: X(4, 0); break;  case 4 * 16 + 1: X(4, 1); break;  case 4 * 16 + 2: X(4, 2); break; \
    case 4 * 16 + 4: X(4, 4); break;                                                                \
    case 8 * 16 + 0: X(8, 0); break;  case 8 * 16 + 1: X(8, 1); break;  case 8 * 16 + 2: X(8, 2); break; \
    case 8 * 16 + 4: X(8, 4); break;  case 8 * 16 + 8: X(8, 8); break;                              \
    default: return fail(SPGNN_ERR_SHAPE, "internal: no kernel instance for this team geometry");   \
  }

int spgnn_gat_can_fuse_mean(int32_t H, int32_t D) {
  int T, R, CH, W;
  return (H > 0 && D > 0 && pick_gat(H, D, T, R, CH, W) && CH >= 1) ? 1 : 0;
}

}  // extern "C"

// rows stored as ST: 16-byte (fp32) or 8-byte (bf16) vector access
template <typename ST> static bool vec_ok_t(const ST* p, int64_t stride) {
  return p == nullptr || ((reinterpret_cast<uintptr_t>(p) & (4 * sizeof(ST) - 1)) == 0 && (stride & 3) == 0);
}

template <typename ST>
static int gat_fwd_impl(const int32_t* indptr, const int32_t* indices, const int32_t* nbr8, const ST* ft, int64_t ft_stride,
                        const float* el, const float* er, int64_t s_stride, const ST* res, int64_t res_stride,
                        const float* bias, ST* out, int64_t out_stride, float* out_mean, int64_t out_mean_stride,
                        float* attn, int64_t N, int64_t E, int32_t H, int32_t D, float negative_slope, int32_t activation,
                        float p_drop, uint64_t seed, const uint64_t* seed_offset, float out_drop_p, uint64_t out_drop_seed,
                        int32_t out_drop_total, int32_t out_drop_offset, float* out_absmax, spgnn_stream_t stream) {
  if (N < 0 || E < 0 || H <= 0 || D <= 0) return fail(SPGNN_ERR_SHAPE, "spgnn_gat_fwd: bad N/E/H/D");
  if (!(out_drop_p >= 0.f && out_drop_p < 1.f) || (out_drop_p > 0.f && (out_drop_offset < 0 || (out_drop_offset & 3) ||
      (int64_t)out_drop_total < (int64_t)out_drop_offset + (int64_t)H * D || !out)))
    return fail(SPGNN_ERR_ENUM, "spgnn_gat_fwd: out_drop_p in [0,1), out_drop_offset % 4 == 0, offset + H*D <= total, `out` set");
  if (N == 0) return SPGNN_OK;
  if (!indptr || !ft || !el || !er || (E > 0 && !attn) || (!out && !out_mean) || (E > 0 && !indices))
    return fail(SPGNN_ERR_NULLPTR, "spgnn_gat_fwd: null pointer");
  const int64_t HD = (int64_t)H * D;
  if (ft_stride < HD || (out && out_stride < HD) || s_stride < H || (res && res_stride < HD) ||
      (out_mean && out_mean_stride < D))
    return fail(SPGNN_ERR_STRIDE, "spgnn_gat_fwd: row stride smaller than row");
  if (activation < SPGNN_ACT_NONE || activation > SPGNN_ACT_RELU) return fail(SPGNN_ERR_ENUM, "spgnn_gat_fwd: activation");
  if (!(p_drop >= 0.f && p_drop < 1.f)) return fail(SPGNN_ERR_ENUM, "spgnn_gat_fwd: p_drop not in [0,1)");
  hipStream_t st = (hipStream_t)stream;
  GatFwdT<ST> a{indptr, indices, nbr8, ft, ft_stride, el, er, s_stride, res, res_stride, bias, out, out_stride, out_mean,
                out_mean_stride, attn, N, H, D, 0, negative_slope, activation, p_drop, 1.f / (1.f - p_drop), seed, seed_offset,
                out_drop_p, 1.f / (1.f - out_drop_p), out_drop_seed, out_drop_total, out_drop_offset, out_absmax};
  int T = 0, R = 0, CH = 0, W = 0;
  const bool vec = pick_gat(H, D, T, R, CH, W) && vec_ok_t(ft, ft_stride) && vec_ok_t(out, out_stride) &&
                   vec_ok_t(res, res_stride) && vec_ok(bias, 0) && vec_ok(out_mean, out_mean_stride);
  const bool fuse_mean = vec && out_mean && CH >= 1;
  if (!fuse_mean && !out)
    return fail(SPGNN_ERR_NULLPTR, "spgnn_gat_fwd: `out` is required when the head mean cannot be fused "
                                   "(see spgnn_gat_can_fuse_mean)");
  if (vec) {
    a.T = T;
    const dim3 grid(grid_for(N, kBlock / T)), block(kBlock);
    if (fuse_mean) {
#define X(R_, CH_) if (T == 64) hipLaunchKernelGGL((gat_fwd_vec<ST, 64, R_, CH_, true>), grid, block, 0, st, a); \
                   else launch_small_team<ST, R_, CH_, true>(grid, block, st, a)
      SPGNN_FOR_R_CH(R, CH, X)
#undef X
    } else {
      a.out_mean = nullptr;
#define X(R_, CH_) if (T == 64) hipLaunchKernelGGL((gat_fwd_vec<ST, 64, R_, CH_, false>), grid, block, 0, st, a); \
                   else launch_small_team<ST, R_, CH_, false>(grid, block, st, a)
      SPGNN_FOR_R_CH(R, CH, X)
#undef X
    }
  } else if (out_drop_p > 0.f || out_absmax) {
    return fail(SPGNN_ERR_SHAPE, "spgnn_gat_fwd: out_drop_p / out_absmax need a vector geometry (H*D = 4*T*R, aligned rows)");
  } else if constexpr (is_f32<ST>::value) {
    hipLaunchKernelGGL(gat_fwd_scalar, dim3(scalar_grid(N * HD)), dim3(kBlock), 0, st, a);
  } else {
    return fail(SPGNN_ERR_SHAPE, "spgnn_gat_fwd_bf16: needs a vector geometry (H*D = 4*T*R) and 8-byte aligned rows");
  }
  if constexpr (is_f32<ST>::value) {
    if (out_mean && !fuse_mean)
      hipLaunchKernelGGL(head_mean_scalar, dim3(scalar_grid(N * D)), dim3(kBlock), 0, st, out, out_stride, out_mean,
                         out_mean_stride, N, H, D);
  } else if (out_mean && !fuse_mean) {
    return fail(SPGNN_ERR_SHAPE, "spgnn_gat_fwd_bf16: the head mean must be fusable (see spgnn_gat_can_fuse_mean)");
  }
  return check_launch("spgnn_gat_fwd");
}

extern "C" {

int spgnn_gat_fwd(const int32_t* indptr, const int32_t* indices, const int32_t* nbr8, const float* ft, int64_t ft_stride,
                  const float* el, const float* er, int64_t s_stride, const float* res, int64_t res_stride,
                  const float* bias, float* out, int64_t out_stride, float* out_mean, int64_t out_mean_stride,
                  float* attn, int64_t N, int64_t E, int32_t H, int32_t D, float negative_slope, int32_t activation,
                  float p_drop, uint64_t seed, const uint64_t* seed_offset, float out_drop_p, uint64_t out_drop_seed,
                  int32_t out_drop_total, int32_t out_drop_offset, float* out_absmax, spgnn_stream_t stream) {
  return gat_fwd_impl<float>(indptr, indices, nbr8, ft, ft_stride, el, er, s_stride, res, res_stride, bias, out, out_stride, out_mean,
                             out_mean_stride, attn, N, E, H, D, negative_slope, activation, p_drop, seed, seed_offset, out_drop_p,
                             out_drop_seed, out_drop_total, out_drop_offset, out_absmax, stream);
}

int spgnn_gat_fwd_bf16(const int32_t* indptr, const int32_t* indices, const int32_t* nbr8, const uint16_t* ft, int64_t ft_stride,
                       const float* el, const float* er, int64_t s_stride, const uint16_t* res, int64_t res_stride,
                       const float* bias, uint16_t* out, int64_t out_stride, float* out_mean, int64_t out_mean_stride,
                       float* attn, int64_t N, int64_t E, int32_t H, int32_t D, float negative_slope, int32_t activation,
                       float p_drop, uint64_t seed, const uint64_t* seed_offset, float out_drop_p, uint64_t out_drop_seed,
                       int32_t out_drop_total, int32_t out_drop_offset, spgnn_stream_t stream) {
  return gat_fwd_impl<bf16s>(indptr, indices, nbr8, reinterpret_cast<const bf16s*>(ft), ft_stride, el, er, s_stride,
                             reinterpret_cast<const bf16s*>(res), res_stride, bias, reinterpret_cast<bf16s*>(out), out_stride,
                             out_mean, out_mean_stride, attn, N, E, H, D, negative_slope, activation, p_drop, seed, seed_offset,
                             out_drop_p, out_drop_seed, out_drop_total, out_drop_offset, nullptr, stream);
}

}  // extern "C"

template <typename ST>
static int gat_bwd_dst_impl(const int32_t* indptr, const int32_t* indices, const int32_t* nbr8, const ST* ft, int64_t ft_stride,
                            const float* el, const float* er, int64_t s_stride, const float* attn, const void* g_out,
                            int64_t g_out_stride, int32_t mean_heads, const ST* out, int64_t out_stride, ST* g_pre,
                            int64_t g_pre_stride, float* g_e, float* g_er, int64_t g_s_stride, float* absmax, int64_t N,
                            int64_t E, int32_t H, int32_t D, float negative_slope, int32_t activation, float p_drop,
                            uint64_t seed, const uint64_t* seed_offset, float out_drop_p, uint64_t out_drop_seed,
                            int32_t out_drop_total, int32_t out_drop_offset, spgnn_stream_t stream) {
  if (N < 0 || E < 0 || H <= 0 || D <= 0) return fail(SPGNN_ERR_SHAPE, "spgnn_gat_bwd_dst: bad N/E/H/D");
  if (!(out_drop_p >= 0.f && out_drop_p < 1.f) || (out_drop_p > 0.f && (out_drop_offset < 0 || (out_drop_offset & 3) || mean_heads ||
      (int64_t)out_drop_total < (int64_t)out_drop_offset + (int64_t)H * D)))
    return fail(SPGNN_ERR_ENUM, "spgnn_gat_bwd_dst: out_drop_p in [0,1), offset % 4 == 0, offset + H*D <= total, not with mean_heads");
  if (N == 0) return SPGNN_OK;
  if (activation < SPGNN_ACT_NONE || activation > SPGNN_ACT_RELU) return fail(SPGNN_ERR_ENUM, "spgnn_gat_bwd_dst: activation");
  if (!indptr || !ft || !el || !er || !g_out || !g_pre || !g_er || (E > 0 && (!indices || !attn || !g_e)) ||
      (activation != SPGNN_ACT_NONE && !out))
    return fail(SPGNN_ERR_NULLPTR, "spgnn_gat_bwd_dst: null pointer");
  const int64_t HD = (int64_t)H * D;
  if (ft_stride < HD || g_out_stride < (mean_heads ? D : HD) || g_pre_stride < HD || s_stride < H || g_s_stride < H ||
      (activation != SPGNN_ACT_NONE && out_stride < HD))
    return fail(SPGNN_ERR_STRIDE, "spgnn_gat_bwd_dst: row stride smaller than row");
  if (!(p_drop >= 0.f && p_drop < 1.f)) return fail(SPGNN_ERR_ENUM, "spgnn_gat_bwd_dst: p_drop not in [0,1)");
  hipStream_t st = (hipStream_t)stream;
  GatBwdDstT<ST> a{indptr, indices, nbr8, ft, ft_stride, el, er, s_stride, attn, g_out, g_out_stride, out, out_stride,
                   g_pre, g_pre_stride, g_e, g_er, g_s_stride, absmax, N, H, D, 0, 0, mean_heads ? 1 : 0, negative_slope,
                   activation, p_drop, 1.f / (1.f - p_drop), seed, seed_offset,
                   out_drop_p, 1.f / (1.f - out_drop_p), out_drop_seed, out_drop_total, out_drop_offset};
  int T = 0, R = 0, CH = 0, W = 0;
  const bool g_ok = mean_heads ? vec_ok(g_out, g_out_stride) : vec_ok_t(reinterpret_cast<const ST*>(g_out), g_out_stride);
  const bool vec = pick_gat(H, D, T, R, CH, W) && vec_ok_t(ft, ft_stride) && g_ok &&
                   vec_ok_t(g_pre, g_pre_stride) && (activation == SPGNN_ACT_NONE || vec_ok_t(out, out_stride));
  if (vec) {
    a.T = T; a.W = W;
    const dim3 grid(grid_for(N, kBlock / T)), block(kBlock);
#define X(R_, CH_) if (T == 64) hipLaunchKernelGGL((gat_bwd_dst_vec<ST, 64, R_, CH_>), grid, block, 0, st, a); \
                   else launch_small_team<ST, R_, CH_>(grid, block, st, a)
    SPGNN_FOR_R_CH(R, CH, X)
#undef X
  } else if (out_drop_p > 0.f) {
    return fail(SPGNN_ERR_SHAPE, "spgnn_gat_bwd_dst: out_drop_p needs a vector geometry (H*D = 4*T*R, aligned rows)");
  } else if constexpr (is_f32<ST>::value) {
    hipLaunchKernelGGL(gat_bwd_dst_scalar, dim3(scalar_grid(N * H)), dim3(kBlock), 0, st, a);
  } else {
    return fail(SPGNN_ERR_SHAPE, "spgnn_gat_bwd_dst_bf16: needs a vector geometry (H*D = 4*T*R) and 8-byte aligned rows");
  }
  return check_launch("spgnn_gat_bwd_dst");
}

extern "C" {

int spgnn_gat_bwd_dst(const int32_t* indptr, const int32_t* indices, const int32_t* nbr8, const float* ft, int64_t ft_stride,
                      const float* el, const float* er, int64_t s_stride, const float* attn, const float* g_out,
                      int64_t g_out_stride, int32_t mean_heads, const float* out, int64_t out_stride, float* g_pre,
                      int64_t g_pre_stride, float* g_e, float* g_er, int64_t g_s_stride, float* absmax, int64_t N,
                      int64_t E, int32_t H, int32_t D, float negative_slope, int32_t activation, float p_drop,
                      uint64_t seed, const uint64_t* seed_offset, float out_drop_p, uint64_t out_drop_seed,
                      int32_t out_drop_total, int32_t out_drop_offset, spgnn_stream_t stream) {
  return gat_bwd_dst_impl<float>(indptr, indices, nbr8, ft, ft_stride, el, er, s_stride, attn, g_out, g_out_stride, mean_heads, out,
                                 out_stride, g_pre, g_pre_stride, g_e, g_er, g_s_stride, absmax, N, E, H, D, negative_slope,
                                 activation, p_drop, seed, seed_offset, out_drop_p, out_drop_seed, out_drop_total, out_drop_offset,
                                 stream);
}

int spgnn_gat_bwd_dst_bf16(const int32_t* indptr, const int32_t* indices, const int32_t* nbr8, const uint16_t* ft, int64_t ft_stride,
                           const float* el, const float* er, int64_t s_stride, const float* attn, const void* g_out,
                           int64_t g_out_stride, int32_t mean_heads, const uint16_t* out, int64_t out_stride, uint16_t* g_pre,
                           int64_t g_pre_stride, float* g_e, float* g_er, int64_t g_s_stride, int64_t N,
                           int64_t E, int32_t H, int32_t D, float negative_slope, int32_t activation, float p_drop,
                           uint64_t seed, const uint64_t* seed_offset, float out_drop_p, uint64_t out_drop_seed,
                           int32_t out_drop_total, int32_t out_drop_offset, spgnn_stream_t stream) {
  return gat_bwd_dst_impl<bf16s>(indptr, indices, nbr8, reinterpret_cast<const bf16s*>(ft), ft_stride, el, er, s_stride, attn, g_out,
                                 g_out_stride, mean_heads, reinterpret_cast<const bf16s*>(out), out_stride,
                                 reinterpret_cast<bf16s*>(g_pre), g_pre_stride, g_e, g_er, g_s_stride, nullptr, N, E, H, D,
                                 negative_slope, activation, p_drop, seed, seed_offset, out_drop_p, out_drop_seed, out_drop_total,
                                 out_drop_offset, stream);
}

}  // extern "C"

template <typename ST>
static int gat_bwd_src_impl(const int32_t* out_indptr, const int32_t* out_indices, const int32_t* out_pos,
                            const int32_t* out_nbr8, const int32_t* out_pos8, const float* attn,
                            const float* g_e, const ST* g_pre, int64_t g_pre_stride, ST* g_ft, int64_t g_ft_stride,
                            float* g_el, int64_t g_s_stride, float* absmax, const float* score_l, const float* score_r,
                            const float* g_er, int64_t N, int64_t E, int32_t H, int32_t D,
                            float p_drop, uint64_t seed, const uint64_t* seed_offset, spgnn_stream_t stream) {
  if (score_l && (!score_r || !g_er)) return fail(SPGNN_ERR_NULLPTR, "spgnn_gat_bwd_src: score_l needs score_r and g_er");
  if (N < 0 || E < 0 || H <= 0 || D <= 0) return fail(SPGNN_ERR_SHAPE, "spgnn_gat_bwd_src: bad N/E/H/D");
  if (N == 0) return SPGNN_OK;
  if (!out_indptr || !g_pre || !g_ft || !g_el || (E > 0 && (!out_indices || !out_pos || !attn || !g_e)))
    return fail(SPGNN_ERR_NULLPTR, "spgnn_gat_bwd_src: null pointer");
  const int64_t HD = (int64_t)H * D;
  if (g_pre_stride < HD || g_ft_stride < HD || g_s_stride < H)
    return fail(SPGNN_ERR_STRIDE, "spgnn_gat_bwd_src: row stride smaller than row");
  if (!(p_drop >= 0.f && p_drop < 1.f)) return fail(SPGNN_ERR_ENUM, "spgnn_gat_bwd_src: p_drop not in [0,1)");
  hipStream_t st = (hipStream_t)stream;
  if ((out_nbr8 == nullptr) != (out_pos8 == nullptr)) return fail(SPGNN_ERR_NULLPTR, "spgnn_gat_bwd_src: out_nbr8 and out_pos8 come together");
  GatBwdSrcT<ST> a{out_indptr, out_indices, out_pos, out_nbr8, out_pos8, attn, g_e, g_pre, g_pre_stride, g_ft, g_ft_stride, g_el, g_s_stride,
                   absmax, N, H, D, 0, p_drop, 1.f / (1.f - p_drop), seed, seed_offset, score_l, score_r, g_er};
  int T = 0, R = 0, CH = 0, W = 0;
  if (pick_gat(H, D, T, R, CH, W) && vec_ok_t(g_pre, g_pre_stride) && vec_ok_t(g_ft, g_ft_stride) && vec_ok(score_l, 0) &&
      vec_ok(score_r, 0)) {
    a.T = T;
    const dim3 grid(grid_for(N, kBlock / T)), block(kBlock);
#define X(R_, CH_) if (T == 64) hipLaunchKernelGGL((gat_bwd_src_vec<ST, 64, R_, CH_>), grid, block, 0, st, a); \
                   else launch_small_team<ST, R_, CH_>(grid, block, st, a)
    SPGNN_FOR_R_CH(R, CH, X)
#undef X
  } else if constexpr (is_f32<ST>::value) {
    hipLaunchKernelGGL(gat_bwd_src_scalar, dim3(scalar_grid(N * HD)), dim3(kBlock), 0, st, a);
  } else {
    return fail(SPGNN_ERR_SHAPE, "spgnn_gat_bwd_src_bf16: needs a vector geometry (H*D = 4*T*R) and 8-byte aligned rows");
  }
  return check_launch("spgnn_gat_bwd_src");
}

extern "C" {

int spgnn_gat_bwd_src(const int32_t* out_indptr, const int32_t* out_indices, const int32_t* out_pos,
                      const int32_t* out_nbr8, const int32_t* out_pos8, const float* attn,
                      const float* g_e, const float* g_pre, int64_t g_pre_stride, float* g_ft, int64_t g_ft_stride,
                      float* g_el, int64_t g_s_stride, float* absmax, const float* score_l, const float* score_r,
                      const float* g_er, int64_t N, int64_t E, int32_t H, int32_t D,
                      float p_drop, uint64_t seed, const uint64_t* seed_offset, spgnn_stream_t stream) {
  return gat_bwd_src_impl<float>(out_indptr, out_indices, out_pos, out_nbr8, out_pos8, attn, g_e, g_pre, g_pre_stride, g_ft, g_ft_stride, g_el,
                                 g_s_stride, absmax, score_l, score_r, g_er, N, E, H, D, p_drop, seed, seed_offset, stream);
}

int spgnn_gat_bwd_src_bf16(const int32_t* out_indptr, const int32_t* out_indices, const int32_t* out_pos,
                           const int32_t* out_nbr8, const int32_t* out_pos8, const float* attn,
                           const float* g_e, const uint16_t* g_pre, int64_t g_pre_stride, uint16_t* g_ft, int64_t g_ft_stride,
                           float* g_el, int64_t g_s_stride, const float* score_l, const float* score_r,
                           const float* g_er, int64_t N, int64_t E, int32_t H, int32_t D,
                           float p_drop, uint64_t seed, const uint64_t* seed_offset, spgnn_stream_t stream) {
  return gat_bwd_src_impl<bf16s>(out_indptr, out_indices, out_pos, out_nbr8, out_pos8, attn, g_e, reinterpret_cast<const bf16s*>(g_pre), g_pre_stride,
                                 reinterpret_cast<bf16s*>(g_ft), g_ft_stride, g_el, g_s_stride, nullptr, score_l, score_r, g_er,
                                 N, E, H, D, p_drop, seed, seed_offset, stream);
}

int spgnn_gat_agg_supported(int32_t H, int32_t F) {
  return ((H == 1 || H == 2 || H == 4) && F > 0 && F % 4 == 0 && F <= 1024) ? 1 : 0;
}

#define SPGNN_FOR_H_R(H_, R_, X)                                                                    \
  switch ((H_) * 16 + (R_)) {                                                                       \
    case 1 * 16 + 1: X(1, 1); break;  case 1 * 16 + 2: X(1, 2); break;  case 1 * 16 + 4: X(1, 4); break; \
    case 2 * 16 + 1: X(2, 1); break;  case 2 * 16 + 2: X(2, 2); break;  case 2 * 16 + 4: X(2, 4); break; \
    case 4 * 16 + 1: X(4, 1); break;  case 4 * 16 + 2: X(4, 2); break;  case 4 * 16 + 4: X(4, 4); break; \
    default: return fail(SPGNN_ERR_SHAPE, "aggregate-first GAT: H must be 1, 2 or 4 and F <= 1024");  \
  }
static int agg_chunks(int F) { return F <= 256 ? 1 : F <= 512 ? 2 : 4; }
static bool agg_team16(int F) { return F <= 192; }
#define SPGNN_FOR_H_R16(H_, R_, X)                                                                  \
  switch ((H_) * 16 + (R_)) {                                                                       \
    case 1 * 16 + 1: X(1, 1); break;  case 1 * 16 + 2: X(1, 2); break;  case 1 * 16 + 3: X(1, 3); break; \
    case 2 * 16 + 1: X(2, 1); break;  case 2 * 16 + 2: X(2, 2); break;  case 2 * 16 + 3: X(2, 3); break; \
    case 4 * 16 + 1: X(4, 1); break;  case 4 * 16 + 2: X(4, 2); break;  case 4 * 16 + 3: X(4, 3); break; \
    default: return fail(SPGNN_ERR_SHAPE, "aggregate-first GAT: H must be 1, 2 or 4 and F <= 1024");  \
  }

}  // extern "C"

template <typename ST>
static int gat_agg_fwd_impl(const char* name, const int32_t* indptr, const int32_t* indices, const ST* x, int64_t x_stride, const float* el,
                      const float* er, int64_t s_stride, float* attn, ST* z, int64_t z_stride, int32_t head_stride,
                      int32_t x_copy_offset, float* absmax, int64_t N, int64_t E, int32_t H, int32_t F,
                      float negative_slope, float p_drop, uint64_t seed, const uint64_t* seed_offset,
                      spgnn_stream_t stream) {
  if (N < 0 || E < 0 || !spgnn_gat_agg_supported(H, F)) return fail(SPGNN_ERR_SHAPE, "spgnn_gat_agg_fwd: bad N/E/H/F");
  if (N == 0) return SPGNN_OK;
  if (!indptr || !x || !el || !er || !attn || !z || (E > 0 && !indices)) return fail(SPGNN_ERR_NULLPTR, "spgnn_gat_agg_fwd: null pointer");
  const int64_t need = x_copy_offset >= 0 ? (int64_t)x_copy_offset + F : F;
  if (x_copy_offset < -2) return fail(SPGNN_ERR_ENUM, "spgnn_gat_agg_fwd: x_copy_offset");
  if (x_stride < F || s_stride < H || head_stride < need || z_stride < (int64_t)H * head_stride + (x_copy_offset == -2 ? F : 0) ||
      (x_copy_offset >= 0 && x_copy_offset < F))
    return fail(SPGNN_ERR_STRIDE, "spgnn_gat_agg_fwd: row stride / block layout too small");
  if (!vec_ok_t(x, x_stride) || !vec_ok_t(z, z_stride) || (head_stride & 3) || (x_copy_offset > 0 && (x_copy_offset & 3)))
    return fail(SPGNN_ERR_STRIDE, "spgnn_gat_agg_fwd: rows must be 16-byte aligned");
  if (!(p_drop >= 0.f && p_drop < 1.f)) return fail(SPGNN_ERR_ENUM, "spgnn_gat_agg_fwd: p_drop not in [0,1)");
  GatAggFwdT<ST> a{indptr, indices, x, x_stride, el, er, s_stride, attn, z, z_stride, head_stride, x_copy_offset, absmax, N, F,
              negative_slope, p_drop, 1.f / (1.f - p_drop), seed, seed_offset};
  hipStream_t st = (hipStream_t)stream;
  const dim3 block(kBlock);
  if (agg_team16(F)) {
    const dim3 grid(grid_for(N, kBlock / 16));
#define X(H_, R_) hipLaunchKernelGGL((gat_agg_fwd<ST, H_, R_, 16>), grid, block, 0, st, a)
    SPGNN_FOR_H_R16(H, (F + 63) / 64, X)
#undef X
  } else {
    const dim3 grid(grid_for(N, kBlock / 64));
#define X(H_, R_) hipLaunchKernelGGL((gat_agg_fwd<ST, H_, R_, 64>), grid, block, 0, st, a)
    SPGNN_FOR_H_R(H, agg_chunks(F), X)
#undef X
  }
  return check_launch(name);
}

template <typename ST>
static int gat_agg_bwd_dst_impl(const char* name, const int32_t* indptr, const int32_t* indices, const ST* x, int64_t x_stride,
                          const float* el, const float* er, int64_t s_stride, const float* attn, const ST* g_z,
                          int64_t g_z_stride, int32_t head_stride, float* g_e, float* g_er, int64_t g_s_stride, int64_t N,
                          int64_t E, int32_t H, int32_t F, float negative_slope, float p_drop, uint64_t seed,
                          const uint64_t* seed_offset, spgnn_stream_t stream, const int32_t* inv = nullptr) {
  if (N < 0 || E < 0 || !spgnn_gat_agg_supported(H, F)) return fail(SPGNN_ERR_SHAPE, "spgnn_gat_agg_bwd_dst: bad N/E/H/F");
  if (N == 0) return SPGNN_OK;
  if (!indptr || !x || !el || !er || !attn || !g_z || !g_e || !g_er || (E > 0 && !indices))
    return fail(SPGNN_ERR_NULLPTR, "spgnn_gat_agg_bwd_dst: null pointer");
  if (x_stride < F || s_stride < H || g_s_stride < H || head_stride < F || g_z_stride < (int64_t)(H - 1) * head_stride + F)
    return fail(SPGNN_ERR_STRIDE, "spgnn_gat_agg_bwd_dst: row stride / block layout too small");
  if (!vec_ok_t(x, x_stride) || !vec_ok_t(g_z, g_z_stride) || (head_stride & 3))
    return fail(SPGNN_ERR_STRIDE, "spgnn_gat_agg_bwd_dst: rows must be 16-byte aligned");
  if (!(p_drop >= 0.f && p_drop < 1.f)) return fail(SPGNN_ERR_ENUM, "spgnn_gat_agg_bwd_dst: p_drop not in [0,1)");
  GatAggBwdDstT<ST> a{indptr, indices, x, x_stride, el, er, s_stride, attn, g_z, g_z_stride, head_stride, g_e, g_er, g_s_stride,
                 N, F, negative_slope, p_drop, 1.f / (1.f - p_drop), seed, seed_offset, inv};
  hipStream_t st = (hipStream_t)stream;
  const dim3 block(kBlock);
  if (agg_team16(F)) {
    const dim3 grid(grid_for(N, kBlock / 16));
#define X(H_, R_) hipLaunchKernelGGL((gat_agg_bwd_dst<ST, H_, R_, 16>), grid, block, 0, st, a)
    SPGNN_FOR_H_R16(H, (F + 63) / 64, X)
#undef X
  } else {
    const dim3 grid(grid_for(N, kBlock / 64));
#define X(H_, R_) hipLaunchKernelGGL((gat_agg_bwd_dst<ST, H_, R_, 64>), grid, block, 0, st, a)
    SPGNN_FOR_H_R(H, agg_chunks(F), X)
#undef X
  }
  return check_launch(name);
}

template <typename ST>
static int gat_agg_bwd_src_impl(const char* name, const int32_t* out_indptr, const int32_t* out_indices, const int32_t* out_pos, const float* attn,
                          const float* g_e, const ST* g_z, int64_t g_z_stride, int32_t head_stride,
                          int32_t x_copy_offset, const float* g_er, const float* w_lr, int64_t w_lr_stride, ST* g_x,
                          int64_t g_x_stride, float* g_el, int64_t g_s_stride, int64_t N, int64_t E, int32_t H, int32_t F,
                          float p_drop, uint64_t seed, const uint64_t* seed_offset, spgnn_stream_t stream, const int32_t* inv = nullptr) {
  if (N < 0 || E < 0 || !spgnn_gat_agg_supported(H, F)) return fail(SPGNN_ERR_SHAPE, "spgnn_gat_agg_bwd_src: bad N/E/H/F");
  if (N == 0) return SPGNN_OK;
  if (!out_indptr || !attn || !g_e || !g_z || !g_x || !g_el || (w_lr && !g_er) || (E > 0 && (!out_indices || !out_pos)))
    return fail(SPGNN_ERR_NULLPTR, "spgnn_gat_agg_bwd_src: null pointer");
  const int64_t need = x_copy_offset >= 0 ? (int64_t)x_copy_offset + F : F;
  if (g_x_stride < F || g_s_stride < H || head_stride < need || g_z_stride < (int64_t)(H - 1) * head_stride + need ||
      (w_lr && w_lr_stride < F))
    return fail(SPGNN_ERR_STRIDE, "spgnn_gat_agg_bwd_src: row stride / block layout too small");
  if (!vec_ok_t(g_x, g_x_stride) || !vec_ok_t(g_z, g_z_stride) || !vec_ok(w_lr, w_lr_stride) || (head_stride & 3) ||
      (x_copy_offset > 0 && (x_copy_offset & 3)))
    return fail(SPGNN_ERR_STRIDE, "spgnn_gat_agg_bwd_src: rows must be 16-byte aligned");
  if (!(p_drop >= 0.f && p_drop < 1.f)) return fail(SPGNN_ERR_ENUM, "spgnn_gat_agg_bwd_src: p_drop not in [0,1)");
  GatAggBwdSrcT<ST> a{out_indptr, out_indices, out_pos, attn, g_e, g_z, g_z_stride, head_stride, x_copy_offset, g_er, w_lr,
                 w_lr_stride, g_x, g_x_stride, g_el, g_s_stride, N, F, p_drop, 1.f / (1.f - p_drop), seed, seed_offset, inv};
  hipStream_t st = (hipStream_t)stream;
  const dim3 block(kBlock);
  if (agg_team16(F)) {
    const dim3 grid(grid_for(N, kBlock / 16));
#define X(H_, R_) hipLaunchKernelGGL((gat_agg_bwd_src<ST, H_, R_, 16>), grid, block, 0, st, a)
    SPGNN_FOR_H_R16(H, (F + 63) / 64, X)
#undef X
  } else {
    const dim3 grid(grid_for(N, kBlock / 64));
#define X(H_, R_) hipLaunchKernelGGL((gat_agg_bwd_src<ST, H_, R_, 64>), grid, block, 0, st, a)
    SPGNN_FOR_H_R(H, agg_chunks(F), X)
#undef X
  }
  return check_launch(name);
}

extern "C" {

int spgnn_gat_agg_fwd(const int32_t* indptr, const int32_t* indices, const float* x, int64_t x_stride, const float* el,
                      const float* er, int64_t s_stride, float* attn, float* z, int64_t z_stride, int32_t head_stride,
                      int32_t x_copy_offset, float* absmax, int64_t N, int64_t E, int32_t H, int32_t F,
                      float negative_slope, float p_drop, uint64_t seed, const uint64_t* seed_offset,
                      spgnn_stream_t stream) {
  return gat_agg_fwd_impl<float>("spgnn_gat_agg_fwd", indptr, indices, x, x_stride, el, er, s_stride, attn, z, z_stride, head_stride,
                                 x_copy_offset, absmax, N, E, H, F, negative_slope, p_drop, seed, seed_offset, stream);
}

int spgnn_gat_agg_fwd_bf16(const int32_t* indptr, const int32_t* indices, const uint16_t* x, int64_t x_stride, const float* el,
                           const float* er, int64_t s_stride, float* attn, uint16_t* z, int64_t z_stride, int32_t head_stride,
                           int32_t x_copy_offset, int64_t N, int64_t E, int32_t H, int32_t F, float negative_slope,
                           float p_drop, uint64_t seed, const uint64_t* seed_offset, spgnn_stream_t stream) {
  return gat_agg_fwd_impl<bf16s>("spgnn_gat_agg_fwd_bf16", indptr, indices, reinterpret_cast<const bf16s*>(x), x_stride, el, er,
                                 s_stride, attn, reinterpret_cast<bf16s*>(z), z_stride, head_stride, x_copy_offset, nullptr, N, E,
                                 H, F, negative_slope, p_drop, seed, seed_offset, stream);
}

int spgnn_gat_agg_bwd_dst(const int32_t* indptr, const int32_t* indices, const float* x, int64_t x_stride,
                          const float* el, const float* er, int64_t s_stride, const float* attn, const float* g_z,
                          int64_t g_z_stride, int32_t head_stride, float* g_e, float* g_er, int64_t g_s_stride, int64_t N,
                          int64_t E, int32_t H, int32_t F, float negative_slope, float p_drop, uint64_t seed,
                          const uint64_t* seed_offset, spgnn_stream_t stream) {
  return gat_agg_bwd_dst_impl<float>("spgnn_gat_agg_bwd_dst", indptr, indices, x, x_stride, el, er, s_stride, attn, g_z, g_z_stride,
                                     head_stride, g_e, g_er, g_s_stride, N, E, H, F, negative_slope, p_drop, seed, seed_offset,
                                     stream);
}

int spgnn_gat_agg_bwd_dst_rows(const int32_t* indptr, const int32_t* indices, const float* x, int64_t x_stride,
                               const float* el, const float* er, int64_t s_stride, const float* attn, const float* g_z_listed,
                               int64_t g_z_stride, int32_t head_stride, const int32_t* inv, float* g_e, float* g_er,
                               int64_t g_s_stride, int64_t N, int64_t E, int32_t H, int32_t F, float negative_slope, float p_drop,
                               uint64_t seed, const uint64_t* seed_offset, spgnn_stream_t stream) {
  if (!inv) return fail(SPGNN_ERR_NULLPTR, "spgnn_gat_agg_bwd_dst_rows: null pointer");
  return gat_agg_bwd_dst_impl<float>("spgnn_gat_agg_bwd_dst_rows", indptr, indices, x, x_stride, el, er, s_stride, attn, g_z_listed,
                                     g_z_stride, head_stride, g_e, g_er, g_s_stride, N, E, H, F, negative_slope, p_drop, seed,
                                     seed_offset, stream, inv);
}

int spgnn_gat_agg_bwd_src_rows(const int32_t* out_indptr, const int32_t* out_indices, const int32_t* out_pos, const float* attn,
                               const float* g_e, const float* g_z_listed, int64_t g_z_stride, int32_t head_stride,
                               int32_t x_copy_offset, const int32_t* inv, const float* g_er, const float* w_lr, int64_t w_lr_stride,
                               float* g_x, int64_t g_x_stride, float* g_el, int64_t g_s_stride, int64_t N, int64_t E, int32_t H,
                               int32_t F, float p_drop, uint64_t seed, const uint64_t* seed_offset, spgnn_stream_t stream) {
  if (!inv) return fail(SPGNN_ERR_NULLPTR, "spgnn_gat_agg_bwd_src_rows: null pointer");
  return gat_agg_bwd_src_impl<float>("spgnn_gat_agg_bwd_src_rows", out_indptr, out_indices, out_pos, attn, g_e, g_z_listed, g_z_stride,
                                     head_stride, x_copy_offset, g_er, w_lr, w_lr_stride, g_x, g_x_stride, g_el, g_s_stride, N, E, H,
                                     F, p_drop, seed, seed_offset, stream, inv);
}

int spgnn_gat_agg_bwd_dst_bf16(const int32_t* indptr, const int32_t* indices, const uint16_t* x, int64_t x_stride,
                               const float* el, const float* er, int64_t s_stride, const float* attn, const uint16_t* g_z,
                               int64_t g_z_stride, int32_t head_stride, float* g_e, float* g_er, int64_t g_s_stride, int64_t N,
                               int64_t E, int32_t H, int32_t F, float negative_slope, float p_drop, uint64_t seed,
                               const uint64_t* seed_offset, spgnn_stream_t stream) {
  return gat_agg_bwd_dst_impl<bf16s>("spgnn_gat_agg_bwd_dst_bf16", indptr, indices, reinterpret_cast<const bf16s*>(x), x_stride, el,
                                     er, s_stride, attn, reinterpret_cast<const bf16s*>(g_z), g_z_stride, head_stride, g_e, g_er,
                                     g_s_stride, N, E, H, F, negative_slope, p_drop, seed, seed_offset, stream);
}

int spgnn_gat_agg_bwd_src(const int32_t* out_indptr, const int32_t* out_indices, const int32_t* out_pos, const float* attn,
                          const float* g_e, const float* g_z, int64_t g_z_stride, int32_t head_stride,
                          int32_t x_copy_offset, const float* g_er, const float* w_lr, int64_t w_lr_stride, float* g_x,
                          int64_t g_x_stride, float* g_el, int64_t g_s_stride, int64_t N, int64_t E, int32_t H, int32_t F,
                          float p_drop, uint64_t seed, const uint64_t* seed_offset, spgnn_stream_t stream) {
  return gat_agg_bwd_src_impl<float>("spgnn_gat_agg_bwd_src", out_indptr, out_indices, out_pos, attn, g_e, g_z, g_z_stride,
                                     head_stride, x_copy_offset, g_er, w_lr, w_lr_stride, g_x, g_x_stride, g_el, g_s_stride, N, E,
                                     H, F, p_drop, seed, seed_offset, stream);
}

int spgnn_gat_agg_bwd_src_bf16(const int32_t* out_indptr, const int32_t* out_indices, const int32_t* out_pos, const float* attn,
                               const float* g_e, const uint16_t* g_z, int64_t g_z_stride, int32_t head_stride,
                               int32_t x_copy_offset, const float* g_er, const float* w_lr, int64_t w_lr_stride, uint16_t* g_x,
                               int64_t g_x_stride, float* g_el, int64_t g_s_stride, int64_t N, int64_t E, int32_t H, int32_t F,
                               float p_drop, uint64_t seed, const uint64_t* seed_offset, spgnn_stream_t stream) {
  return gat_agg_bwd_src_impl<bf16s>("spgnn_gat_agg_bwd_src_bf16", out_indptr, out_indices, out_pos, attn, g_e,
                                     reinterpret_cast<const bf16s*>(g_z), g_z_stride, head_stride, x_copy_offset, g_er, w_lr,
                                     w_lr_stride, reinterpret_cast<bf16s*>(g_x), g_x_stride, g_el, g_s_stride, N, E, H, F, p_drop,
                                     seed, seed_offset, stream);
}

int spgnn_head_mean(const float* out, int64_t out_stride, float* out_mean, int64_t out_mean_stride, int64_t N, int32_t H,
                    int32_t D, spgnn_stream_t stream) {
  if (N < 0 || H <= 0 || D <= 0) return fail(SPGNN_ERR_SHAPE, "spgnn_head_mean: bad N/H/D");
  if (N == 0) return SPGNN_OK;
  if (!out || !out_mean) return fail(SPGNN_ERR_NULLPTR, "spgnn_head_mean: null pointer");
  if (out_stride < (int64_t)H * D || out_mean_stride < D) return fail(SPGNN_ERR_STRIDE, "spgnn_head_mean: row stride smaller than row");
  hipStream_t st = (hipStream_t)stream;
  if (D % 4 == 0 && vec_ok(out, out_stride) && vec_ok(out_mean, out_mean_stride)) {
    int64_t blocks = (N * (D / 4) + kBlock - 1) / kBlock;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(head_mean_vec, dim3((unsigned)blocks), dim3(kBlock), 0, st, out, out_stride, out_mean, out_mean_stride, N, H, D);
  } else {
    hipLaunchKernelGGL(head_mean_scalar, dim3(scalar_grid(N * D)), dim3(kBlock), 0, st, out, out_stride, out_mean,
                       out_mean_stride, N, H, D);
  }
  return check_launch("spgnn_head_mean");
}

int spgnn_act_bwd(const float* g_out, int64_t g_out_stride, int32_t mean_heads, const float* out, int64_t out_stride,
                  float* g_pre, int64_t g_pre_stride, float* absmax, int64_t N, int32_t H, int32_t D, int32_t activation,
                  spgnn_stream_t stream) {
  if (N < 0 || H <= 0 || D <= 0 || D % 4) return fail(SPGNN_ERR_SHAPE, "spgnn_act_bwd: bad N/H/D (D must be a multiple of 4)");
  if (activation < SPGNN_ACT_NONE || activation > SPGNN_ACT_LRELU) return fail(SPGNN_ERR_ENUM, "spgnn_act_bwd: activation");
  if (N == 0) return SPGNN_OK;
  if (!g_out || !g_pre || (activation != SPGNN_ACT_NONE && !out)) return fail(SPGNN_ERR_NULLPTR, "spgnn_act_bwd: null pointer");
  const int64_t HD = (int64_t)H * D;
  if (g_out_stride < (mean_heads ? D : HD) || g_pre_stride < HD || (activation != SPGNN_ACT_NONE && out_stride < HD))
    return fail(SPGNN_ERR_STRIDE, "spgnn_act_bwd: row stride smaller than row");
  if (!vec_ok(g_out, g_out_stride) || !vec_ok(g_pre, g_pre_stride) || (activation != SPGNN_ACT_NONE && !vec_ok(out, out_stride)))
    return fail(SPGNN_ERR_STRIDE, "spgnn_act_bwd: rows must be 16-byte aligned");
  if (!mean_heads) {
    int64_t blocks = (N * (HD / 4) + kBlock - 1) / kBlock;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(act_bwd_flat_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, g_out, g_out_stride, out,
                       out_stride, g_pre, g_pre_stride, absmax, N, (int)HD, activation, 0.f, (uint64_t)0, (const uint64_t*)nullptr, 1.f);
    return check_launch("spgnn_act_bwd");
  }
  hipLaunchKernelGGL(act_bwd_kernel, dim3((unsigned)((N + kBlock / 64 - 1) / (kBlock / 64))), dim3(kBlock), 0,
                     (hipStream_t)stream, g_out, g_out_stride, mean_heads ? 1 : 0, out, out_stride, g_pre, g_pre_stride, absmax,
                     N, H, D, activation);
  return check_launch("spgnn_act_bwd");
}

static int act_bwd_dropout_impl(const float* g_out, int64_t g_out_stride, const float* out, int64_t out_stride, float* g_pre,
                                int64_t g_pre_stride, float* absmax, int64_t N, int32_t W, int32_t activation, float p_drop,
                                uint64_t seed, const uint64_t* seed_offset, bool out_dropped, spgnn_stream_t stream) {
  if (N < 0 || W <= 0 || W % 4) return fail(SPGNN_ERR_SHAPE, "spgnn_act_bwd_dropout: bad N/W (W must be a multiple of 4)");
  if (activation < SPGNN_ACT_NONE || activation > SPGNN_ACT_LRELU) return fail(SPGNN_ERR_ENUM, "spgnn_act_bwd_dropout: activation");
  if (!(p_drop >= 0.f && p_drop < 1.f)) return fail(SPGNN_ERR_SHAPE, "spgnn_act_bwd_dropout: p_drop outside [0, 1)");
  if (N == 0) return SPGNN_OK;
  if (!g_out || !g_pre || (activation != SPGNN_ACT_NONE && !out)) return fail(SPGNN_ERR_NULLPTR, "spgnn_act_bwd_dropout: null pointer");
  if (g_out_stride < W || g_pre_stride < W || (activation != SPGNN_ACT_NONE && out_stride < W))
    return fail(SPGNN_ERR_STRIDE, "spgnn_act_bwd_dropout: row stride smaller than row");
  if (!vec_ok(g_out, g_out_stride) || !vec_ok(g_pre, g_pre_stride) || (activation != SPGNN_ACT_NONE && !vec_ok(out, out_stride)))
    return fail(SPGNN_ERR_STRIDE, "spgnn_act_bwd_dropout: rows must be 16-byte aligned");
  int64_t blocks = (N * (W / 4) + kBlock - 1) / kBlock;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(act_bwd_flat_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, g_out, g_out_stride, out,
                     out_stride, g_pre, g_pre_stride, absmax, N, (int)W, activation, p_drop, seed, seed_offset,
                     out_dropped ? 1.f - p_drop : 1.f);
  return check_launch("spgnn_act_bwd_dropout");
}

int spgnn_act_bwd_dropout(const float* g_out, int64_t g_out_stride, const float* out, int64_t out_stride, float* g_pre,
                          int64_t g_pre_stride, float* absmax, int64_t N, int32_t W, int32_t activation, float p_drop,
                          uint64_t seed, const uint64_t* seed_offset, spgnn_stream_t stream) {
  return act_bwd_dropout_impl(g_out, g_out_stride, out, out_stride, g_pre, g_pre_stride, absmax, N, W, activation, p_drop, seed,
                              seed_offset, false, stream);
}

int spgnn_act_bwd_dropped(const float* g_out, int64_t g_out_stride, const float* out_dropped, int64_t out_stride, float* g_pre,
                          int64_t g_pre_stride, float* absmax, int64_t N, int32_t W, int32_t activation, float p_drop,
                          uint64_t seed, const uint64_t* seed_offset, spgnn_stream_t stream) {
  return act_bwd_dropout_impl(g_out, g_out_stride, out_dropped, out_stride, g_pre, g_pre_stride, absmax, N, W, activation, p_drop,
                              seed, seed_offset, true, stream);
}

int32_t spgnn_act_bwd_colsum_blocks(int64_t N, int32_t W) {
  if (N <= 0 || W <= 0 || W % 4 || kBlock % (W / 4)) return 0;          // unsupported width: W / 4 must divide 256
  const int64_t rows_per_block = kBlock / (W / 4);
  int64_t b = (N + 2 * rows_per_block - 1) / (2 * rows_per_block);      // two rows per thread and trip
  if (b > 2048) b = 2048;
  return (int32_t)(b < 1 ? 1 : b);
}

int spgnn_act_bwd_colsum(const float* g_out, int64_t g_out_stride, const float* out, int64_t out_stride, float* g_pre,
                         int64_t g_pre_stride, float* absmax, float* colsum_partials, int64_t N, int32_t W, int32_t activation,
                         float p_drop, uint64_t seed, const uint64_t* seed_offset, const float* dot_x, int64_t dot_x_stride,
                         spgnn_stream_t stream) {
  if (N <= 0 || W <= 0 || W % 4 || kBlock % (W / 4)) return fail(SPGNN_ERR_SHAPE, "spgnn_act_bwd_colsum: bad N/W (N > 0, W / 4 must divide 256)");
  if (activation < SPGNN_ACT_NONE || activation > SPGNN_ACT_LRELU) return fail(SPGNN_ERR_ENUM, "spgnn_act_bwd_colsum: activation");
  if (!(p_drop >= 0.f && p_drop < 1.f)) return fail(SPGNN_ERR_SHAPE, "spgnn_act_bwd_colsum: p_drop outside [0, 1)");
  if (!g_out || !g_pre || !colsum_partials || (activation != SPGNN_ACT_NONE && !out)) return fail(SPGNN_ERR_NULLPTR, "spgnn_act_bwd_colsum: null pointer");
  if (g_out_stride < W || g_pre_stride < W || (activation != SPGNN_ACT_NONE && out_stride < W))
    return fail(SPGNN_ERR_STRIDE, "spgnn_act_bwd_colsum: row stride smaller than row");
  if (!vec_ok(g_out, g_out_stride) || !vec_ok(g_pre, g_pre_stride) || (activation != SPGNN_ACT_NONE && !vec_ok(out, out_stride)) ||
      !aligned16(colsum_partials) || (dot_x && (dot_x_stride < W || !vec_ok(dot_x, dot_x_stride))))
    return fail(SPGNN_ERR_STRIDE, "spgnn_act_bwd_colsum: rows must be 16-byte aligned");
  const int32_t blocks = spgnn_act_bwd_colsum_blocks(N, W);
  hipLaunchKernelGGL(act_bwd_colsum_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, g_out, g_out_stride, out,
                     out_stride, g_pre, g_pre_stride, absmax, colsum_partials, N, (int)W, activation, p_drop, seed, seed_offset, dot_x,
                     dot_x_stride);
  return check_launch("spgnn_act_bwd_colsum");
}

int32_t spgnn_act_bwd_proj_blocks(int64_t N) {
  int64_t b = (N + 7) / 8;                           // >= 8 rows per block (the block's W slice, J x 4 KB, is read once per
  if (b > 1536) b = 1536;                            // block); 6 blocks per CU at large N, enough blocks at N ~ 1e4 too
  return (int32_t)(b < 1 ? 1 : b);
}

static int act_bwd_proj_launch(const float* g_s, int64_t g_s_stride, int32_t J, const float* w, int64_t w_stride, const float* out,
                               int64_t out_stride, float* g_pre, int64_t g_pre_stride, float* absmax_partials, int64_t N, int32_t H,
                               int32_t D, int32_t activation, const int32_t* row_list, const int32_t* rows_cnt, spgnn_stream_t stream) {
  if (N < 0 || H <= 0 || H > 4 || D <= 0 || D % 4 || D > 1024 || J <= 0 || J > 32)
    return fail(SPGNN_ERR_SHAPE, "spgnn_act_bwd_proj: bad N/H/D/J (H <= 4, D % 4 == 0, D <= 1024, J <= 32)");
  if (activation < SPGNN_ACT_NONE || activation > SPGNN_ACT_LRELU) return fail(SPGNN_ERR_ENUM, "spgnn_act_bwd_proj: activation");
  if (N == 0) return SPGNN_OK;
  if (!g_s || !w || !g_pre || !absmax_partials || (activation != SPGNN_ACT_NONE && !out))
    return fail(SPGNN_ERR_NULLPTR, "spgnn_act_bwd_proj: null pointer");
  const int64_t HD = (int64_t)H * D;
  if (g_s_stride < J || w_stride < D || g_pre_stride < HD || (activation != SPGNN_ACT_NONE && out_stride < HD))
    return fail(SPGNN_ERR_STRIDE, "spgnn_act_bwd_proj: row stride smaller than row");
  if (!vec_ok(w, w_stride) || !vec_ok(g_pre, g_pre_stride) || (activation != SPGNN_ACT_NONE && !vec_ok(out, out_stride)))
    return fail(SPGNN_ERR_STRIDE, "spgnn_act_bwd_proj: rows must be 16-byte aligned");
  const int32_t blocks = spgnn_act_bwd_proj_blocks(N);
  const int64_t rpb = (N + blocks - 1) / blocks;
  hipStream_t st = (hipStream_t)stream;
#define XH(JP, HT) hipLaunchKernelGGL((act_bwd_proj_kernel<JP, HT, kAbpRows, false>), dim3((unsigned)blocks), dim3(256), 0, st, g_s, g_s_stride, \
                                      (int)J, w, w_stride, out, out_stride, g_pre, g_pre_stride, absmax_partials, N, rpb, (int)H, (int)D, (int)activation, \
                                      (float*)nullptr, row_list, rows_cnt)
#define X(JP) { if (H == 2) XH(JP, 2); else if (H == 1) XH(JP, 1); else XH(JP, 0); }
  if (J <= 8) X(8) else if (J <= 16) X(16) else if (J <= 24) X(24) else X(32)
#undef X
#undef XH
  return check_launch("spgnn_act_bwd_proj");
}

int spgnn_act_bwd_proj(const float* g_s, int64_t g_s_stride, int32_t J, const float* w, int64_t w_stride, const float* out,
                       int64_t out_stride, float* g_pre, int64_t g_pre_stride, float* absmax_partials, int64_t N, int32_t H,
                       int32_t D, int32_t activation, spgnn_stream_t stream) {
  return act_bwd_proj_launch(g_s, g_s_stride, J, w, w_stride, out, out_stride, g_pre, g_pre_stride, absmax_partials, N, H, D, activation,
                             nullptr, nullptr, stream);
}

int spgnn_act_bwd_proj_rows(const float* g_s, int64_t g_s_stride, int32_t J, const float* w, int64_t w_stride, const float* out,
                            int64_t out_stride, const int32_t* rows, const int32_t* rows_cnt, float* g_pre, int64_t g_pre_stride,
                            float* absmax_partials, int64_t cap, int32_t H, int32_t D, int32_t activation, spgnn_stream_t stream) {
  if (!rows || !rows_cnt) return fail(SPGNN_ERR_NULLPTR, "spgnn_act_bwd_proj_rows: null pointer");
  return act_bwd_proj_launch(g_s, g_s_stride, J, w, w_stride, out, out_stride, g_pre, g_pre_stride, absmax_partials, cap, H, D, activation,
                             rows, rows_cnt, stream);
}

int spgnn_gemm_nt_skinny(const float* a, int64_t a_stride, const float* b, int64_t b_stride, float* c, int64_t c_stride, int64_t M,
                         int64_t N, int64_t K, const float* bias, int32_t activation, const float* score_l, const float* score_r,
                         float* score_out, int32_t score_cols, spgnn_stream_t stream) {
  if (M < 0 || N <= 0 || K <= 0 || N > (1 << 24) || K > (1 << 24)) return fail(SPGNN_ERR_SHAPE, "spgnn_gemm_nt_skinny: bad M/N/K");
  if (activation < SPGNN_ACT_NONE || activation > SPGNN_ACT_LRELU) return fail(SPGNN_ERR_ENUM, "spgnn_gemm_nt_skinny: activation");
  if (M == 0) return SPGNN_OK;
  if (!a || !b || !c) return fail(SPGNN_ERR_NULLPTR, "spgnn_gemm_nt_skinny: null pointer");
  if (a_stride < K || b_stride < K || c_stride < N) return fail(SPGNN_ERR_STRIDE, "spgnn_gemm_nt_skinny: row stride smaller than row");
  if (!vec_ok(a, a_stride) || !vec_ok(b, b_stride)) return fail(SPGNN_ERR_STRIDE, "spgnn_gemm_nt_skinny: operand rows must be 16-byte aligned");
  if (score_out && (!score_l || !score_r || score_cols <= 0 || score_cols % 64 || score_cols > N || bias || activation != SPGNN_ACT_NONE))
    return fail(SPGNN_ERR_SHAPE, "spgnn_gemm_nt_skinny: score partials need score_l / score_r, score_cols % 64 == 0 <= N, no bias / activation");
  const dim3 grid((unsigned)((M + 15) / 16), (unsigned)((N + 63) / 64));
  // deep products split K over eight waves (1063 -> 1024 at 150 rows: 17 -> 9 us), shallow ones over four
  if (K >= 512)
    hipLaunchKernelGGL((gemm_nt_skinny<8>), grid, dim3(512), 0, (hipStream_t)stream, a, a_stride, b, b_stride, c, c_stride, M,
                       (int)N, (int)K, bias, (int)activation, score_l, score_r, score_out, (int)score_cols);
  else
    hipLaunchKernelGGL((gemm_nt_skinny<4>), grid, dim3(256), 0, (hipStream_t)stream, a, a_stride, b, b_stride, c, c_stride, M,
                       (int)N, (int)K, bias, (int)activation, score_l, score_r, score_out, (int)score_cols);
  return check_launch("spgnn_gemm_nt_skinny");
}

int32_t spgnn_act_bwd_proj_wgrad_blocks(int64_t N) {
  int64_t b = (N + 31) / 32;                         // every block leaves a J x D partial: two blocks per CU at large N keep the
  if (b > 512) b = 512;                              // partials at 46 MB for the 22 x 1024 classifier (four rows per trip in flight)
  return (int32_t)(b < 1 ? 1 : b);
}

int spgnn_act_bwd_proj_wgrad(const float* g_s, int64_t g_s_stride, int32_t J, const float* w, int64_t w_stride, const float* out,
                             int64_t out_stride, float* g_pre, int64_t g_pre_stride, float* absmax_partials, float* w_grad_partials,
                             int64_t N, int32_t H, int32_t D, int32_t activation, spgnn_stream_t stream) {
  if (N < 0 || (H != 1 && H != 2) || D <= 0 || D % 4 || D > 1024 || J <= 0 || J > 24)
    return fail(SPGNN_ERR_SHAPE, "spgnn_act_bwd_proj_wgrad: bad N/H/D/J (H in {1, 2}, D % 4 == 0, D <= 1024, J <= 24)");
  if (activation <= SPGNN_ACT_NONE || activation > SPGNN_ACT_LRELU) return fail(SPGNN_ERR_ENUM, "spgnn_act_bwd_proj_wgrad: activation");
  if (N == 0) return SPGNN_OK;
  if (!g_s || !w || !g_pre || !absmax_partials || !out || !w_grad_partials)
    return fail(SPGNN_ERR_NULLPTR, "spgnn_act_bwd_proj_wgrad: null pointer");
  const int64_t HD = (int64_t)H * D;
  if (g_s_stride < J || w_stride < D || g_pre_stride < HD || out_stride < HD)
    return fail(SPGNN_ERR_STRIDE, "spgnn_act_bwd_proj_wgrad: row stride smaller than row");
  if (!vec_ok(w, w_stride) || !vec_ok(g_pre, g_pre_stride) || !vec_ok(out, out_stride) || !aligned16(w_grad_partials))
    return fail(SPGNN_ERR_STRIDE, "spgnn_act_bwd_proj_wgrad: rows must be 16-byte aligned");
  const int32_t blocks = spgnn_act_bwd_proj_wgrad_blocks(N);
  const int64_t rpb = (N + blocks - 1) / blocks;
  hipStream_t st = (hipStream_t)stream;
#define XH(JP, HT) hipLaunchKernelGGL((act_bwd_proj_kernel<JP, HT, kAbpRows, true>), dim3((unsigned)blocks), dim3(256), 0, st, g_s, g_s_stride, \
                                      (int)J, w, w_stride, out, out_stride, g_pre, g_pre_stride, absmax_partials, N, rpb, (int)H, (int)D, (int)activation, \
                                      w_grad_partials, (const int32_t*)nullptr, (const int32_t*)nullptr)
#define X(JP) { if (H == 2) XH(JP, 2); else XH(JP, 1); }
  if (J <= 8) X(8) else if (J <= 16) X(16) else X(24)
#undef X
#undef XH
  return check_launch("spgnn_act_bwd_proj_wgrad");
}

int spgnn_fold_scores_fwd(const float* W, int64_t w_stride, const float* attn_l, const float* attn_r, float* w_lr,
                          int32_t w_lr_stride, int32_t H, int32_t D, int32_t K, spgnn_stream_t stream) {
  if (H <= 0 || D <= 0 || K <= 0) return fail(SPGNN_ERR_SHAPE, "spgnn_fold_scores_fwd: bad H/D/K");
  if (!W || !attn_l || !attn_r || !w_lr) return fail(SPGNN_ERR_NULLPTR, "spgnn_fold_scores_fwd: null pointer");
  if (w_stride < K || w_lr_stride < K) return fail(SPGNN_ERR_STRIDE, "spgnn_fold_scores_fwd: row stride smaller than row");
  hipLaunchKernelGGL(fold_scores_fwd, dim3((unsigned)((w_lr_stride + 15) / 16), (unsigned)H), dim3(1024), 0, (hipStream_t)stream,
                     W, w_stride, attn_l, attn_r, w_lr, w_lr_stride, H, D, K);
  return check_launch("spgnn_fold_scores_fwd");
}

int spgnn_fold_scores_bwd(const float* W, int64_t w_stride, const float* attn_l, const float* attn_r, const float* g_w_lr,
                          int32_t w_lr_stride, float* g_W, int64_t g_w_stride, float* g_attn_l, float* g_attn_r, int32_t H,
                          int32_t D, int32_t K, spgnn_stream_t stream) {
  if (H <= 0 || D <= 0 || K <= 0) return fail(SPGNN_ERR_SHAPE, "spgnn_fold_scores_bwd: bad H/D/K");
  if (!W || !attn_l || !attn_r || !g_w_lr || !g_W || !g_attn_l || !g_attn_r)
    return fail(SPGNN_ERR_NULLPTR, "spgnn_fold_scores_bwd: null pointer");
  if (w_stride < K || w_lr_stride < K || g_w_stride < K) return fail(SPGNN_ERR_STRIDE, "spgnn_fold_scores_bwd: row stride smaller than row");
  const int rows = H * D;
  hipLaunchKernelGGL(fold_scores_bwd, dim3((unsigned)((rows + kBlock / 64 - 1) / (kBlock / 64))), dim3(kBlock), 0,
                     (hipStream_t)stream, W, w_stride, attn_l, attn_r, g_w_lr, w_lr_stride, g_W, g_w_stride, g_attn_l, g_attn_r,
                     H, D, K);
  return check_launch("spgnn_fold_scores_bwd");
}

int64_t spgnn_cat_dropout_blocks(int64_t N, int32_t width) {
  int64_t blocks = (N * ((width + 3) / 4) + kBlock - 1) / kBlock;
  return blocks > 32768 ? 32768 : (blocks < 1 ? 1 : blocks);
}

int spgnn_cat_dropout(const float* src, int64_t src_stride, float* dst, int64_t dst_stride, int64_t N, int32_t width,
                      int32_t col_offset, int32_t total_width, float p_drop, uint64_t seed, const uint64_t* seed_offset,
                      int32_t backward, float* absmax_partials, spgnn_stream_t stream) {
  if (N < 0 || width <= 0 || col_offset < 0 || total_width < col_offset + width) return fail(SPGNN_ERR_SHAPE, "spgnn_cat_dropout: bad N/width/offset");
  if (N == 0) return SPGNN_OK;
  if (!src || !dst) return fail(SPGNN_ERR_NULLPTR, "spgnn_cat_dropout: null pointer");
  if (src_stride < (backward ? total_width : width) || dst_stride < (backward ? width : total_width))
    return fail(SPGNN_ERR_STRIDE, "spgnn_cat_dropout: row stride smaller than row");
  if (!(p_drop >= 0.f && p_drop < 1.f)) return fail(SPGNN_ERR_ENUM, "spgnn_cat_dropout: p_drop not in [0,1)");
  int64_t blocks = (N * ((width + 3) / 4) + kBlock - 1) / kBlock;
  if (blocks > 32768) blocks = 32768;
  const bool vec = (width & 3) == 0 && (col_offset & 3) == 0 && vec_ok(src, src_stride) && vec_ok(dst, dst_stride);
  if (vec)
    hipLaunchKernelGGL((cat_dropout_kernel<float, true>), dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, src, src_stride, dst,
                       dst_stride, N, width, col_offset, total_width, p_drop, 1.f / (1.f - p_drop), seed, seed_offset, backward ? 1 : 0, absmax_partials);
  else
    hipLaunchKernelGGL((cat_dropout_kernel<float, false>), dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, src, src_stride, dst,
                       dst_stride, N, width, col_offset, total_width, p_drop, 1.f / (1.f - p_drop), seed, seed_offset, backward ? 1 : 0, absmax_partials);
  return check_launch("spgnn_cat_dropout");
}

int spgnn_cat_dropout_bf16(const uint16_t* src, int64_t src_stride, uint16_t* dst, int64_t dst_stride, int64_t N, int32_t width,
                           int32_t col_offset, int32_t total_width, float p_drop, uint64_t seed, const uint64_t* seed_offset,
                           int32_t backward, spgnn_stream_t stream) {
  if (N < 0 || width <= 0 || col_offset < 0 || total_width < col_offset + width) return fail(SPGNN_ERR_SHAPE, "spgnn_cat_dropout_bf16: bad N/width/offset");
  if (N == 0) return SPGNN_OK;
  if (!src || !dst) return fail(SPGNN_ERR_NULLPTR, "spgnn_cat_dropout_bf16: null pointer");
  if (src_stride < (backward ? total_width : width) || dst_stride < (backward ? width : total_width))
    return fail(SPGNN_ERR_STRIDE, "spgnn_cat_dropout_bf16: row stride smaller than row");
  if (!(p_drop >= 0.f && p_drop < 1.f)) return fail(SPGNN_ERR_ENUM, "spgnn_cat_dropout_bf16: p_drop not in [0,1)");
  const bf16s* s_ = reinterpret_cast<const bf16s*>(src);
  bf16s* d_ = reinterpret_cast<bf16s*>(dst);
  if ((width & 3) || (col_offset & 3) || !vec_ok_t(s_, src_stride) || !vec_ok_t(d_, dst_stride))
    return fail(SPGNN_ERR_STRIDE, "spgnn_cat_dropout_bf16: widths, offsets and strides must be multiples of 4, rows 8-byte aligned");
  int64_t blocks = (N * (width / 4) + kBlock - 1) / kBlock;
  if (blocks > 32768) blocks = 32768;
  hipLaunchKernelGGL((cat_dropout_kernel<bf16s, true>), dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, s_, src_stride, d_,
                     dst_stride, N, width, col_offset, total_width, p_drop, 1.f / (1.f - p_drop), seed, seed_offset, backward ? 1 : 0,
                     (float*)nullptr);
  return check_launch("spgnn_cat_dropout_bf16");
}

int spgnn_scores_from_parts(const float* parts, float* s, int64_t s_stride, int64_t N, int32_t H, int32_t D,
                            spgnn_stream_t stream) {
  if (N < 0 || H <= 0 || D <= 0 || (D & 63)) return fail(SPGNN_ERR_SHAPE, "spgnn_scores_from_parts: D must be a multiple of 64");
  if (N == 0) return SPGNN_OK;
  if (!parts || !s) return fail(SPGNN_ERR_NULLPTR, "spgnn_scores_from_parts: null pointer");
  if (s_stride < 2 * H) return fail(SPGNN_ERR_STRIDE, "spgnn_scores_from_parts: row stride smaller than row");
  hipLaunchKernelGGL(scores_from_parts_kernel, dim3((unsigned)((N * H + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                     (hipStream_t)stream, parts, s, s_stride, N, H, D / 64);
  return check_launch("spgnn_scores_from_parts");
}

int spgnn_spmm_sum(const int32_t* indptr, const int32_t* indices, const float* x, int64_t x_stride, const float* w_src,
                   const float* w_dst, const float* self_eps, const float* bias, int32_t activation, float* out,
                   int64_t out_stride, int64_t N, int64_t E, int32_t F, float* absmax_out, spgnn_stream_t stream) {
  return spgnn_spmm_sum_dropout(indptr, indices, x, x_stride, w_src, w_dst, self_eps, bias, activation, out, out_stride, N, E, F,
                                absmax_out, 0.f, 0, nullptr, stream);
}

int spgnn_spmm_sum_dropout(const int32_t* indptr, const int32_t* indices, const float* x, int64_t x_stride, const float* w_src,
                           const float* w_dst, const float* self_eps, const float* bias, int32_t activation, float* out,
                           int64_t out_stride, int64_t N, int64_t E, int32_t F, float* absmax_out, float p_drop, uint64_t seed,
                           const uint64_t* seed_offset, spgnn_stream_t stream) {
  if (N < 0 || E < 0 || F <= 0) return fail(SPGNN_ERR_SHAPE, "spgnn_spmm_sum: bad N/E/F");
  if (!(p_drop >= 0.f && p_drop < 1.f)) return fail(SPGNN_ERR_SHAPE, "spgnn_spmm_sum: p_drop outside [0, 1)");
  if (activation < SPGNN_ACT_NONE || activation > SPGNN_ACT_LRELU) return fail(SPGNN_ERR_ENUM, "spgnn_spmm_sum: activation");
  if (N == 0) return SPGNN_OK;
  if (!indptr || !x || !out || (E > 0 && !indices)) return fail(SPGNN_ERR_NULLPTR, "spgnn_spmm_sum: null pointer");
  if (x_stride < F || out_stride < F) return fail(SPGNN_ERR_STRIDE, "spgnn_spmm_sum: row stride smaller than row");
  hipStream_t st = (hipStream_t)stream;
  SpmmSum a{indptr, indices, x, x_stride, w_src, w_dst, self_eps, out, out_stride, N, F, 0, bias, (int)activation, absmax_out,
            p_drop, p_drop > 0.f ? 1.f / (1.f - p_drop) : 1.f, seed, seed_offset};
  int T, R;
  // rows of 128 / 256 floats on 16-lane teams (four nodes per wave, 2 / 4 float4 per lane) like the GAT kernels' narrow
  // layers: a wave walks four index -> row chains instead of one (st_gcn_3 1.285 -> 1.143 ms, st_gin_3 3.81 -> 3.70, one process)
  if (pick_team(F, T, R, kSpmmNarrow) && vec_ok(x, x_stride) && vec_ok(out, out_stride) && (!bias || aligned16(bias))) {
    a.T = T;
    DISPATCH_R(T, R, spmm_sum_vec, dim3(grid_for(N, kBlock / T)), dim3(kBlock), 0, st, a);
  } else if (p_drop > 0.f) {
    return fail(SPGNN_ERR_STRIDE, "spgnn_spmm_sum_dropout: dropout needs 16-byte rows of a supported width");
  } else {
    hipLaunchKernelGGL(spmm_sum_scalar, dim3(scalar_grid(N * F)), dim3(kBlock), 0, st, a);
  }
  return check_launch("spgnn_spmm_sum");
}

int spgnn_spmm_max_fwd(const int32_t* indptr, const int32_t* indices, const float* x, int64_t x_stride, float* out,
                       int64_t out_stride, int32_t* arg, int64_t arg_stride, int64_t N, int64_t E, int32_t F,
                       spgnn_stream_t stream) {
  if (N < 0 || E < 0 || F <= 0) return fail(SPGNN_ERR_SHAPE, "spgnn_spmm_max_fwd: bad N/E/F");
  if (N == 0) return SPGNN_OK;
  if (!indptr || !x || !out || !arg || (E > 0 && !indices)) return fail(SPGNN_ERR_NULLPTR, "spgnn_spmm_max_fwd: null pointer");
  if (x_stride < F || out_stride < F || arg_stride < F) return fail(SPGNN_ERR_STRIDE, "spgnn_spmm_max_fwd: row stride smaller than row");
  hipStream_t st = (hipStream_t)stream;
  SpmmMaxFwd a{indptr, indices, x, x_stride, out, out_stride, arg, arg_stride, N, F, 0, nullptr};
  int T, R;
  if (pick_team(F, T, R, kSpmmNarrow) && vec_ok(x, x_stride) && vec_ok(out, out_stride) && vec_ok(arg, arg_stride)) {
    a.T = T;
    DISPATCH_R(T, R, spmm_max_fwd_vec, dim3(grid_for(N, kBlock / T)), dim3(kBlock), 0, st, a);
  } else {
    hipLaunchKernelGGL(spmm_max_fwd_scalar, dim3(scalar_grid(N * F)), dim3(kBlock), 0, st, a);
  }
  return check_launch("spgnn_spmm_max_fwd");
}

int spgnn_spmm_max_bwd(const int32_t* out_indptr, const int32_t* out_indices, const int32_t* out_pos, const float* g_out,
                       int64_t g_out_stride, const int32_t* arg, int64_t arg_stride, float* g_x, int64_t g_x_stride,
                       int64_t N, int64_t E, int32_t F, spgnn_stream_t stream) {
  if (N < 0 || E < 0 || F <= 0) return fail(SPGNN_ERR_SHAPE, "spgnn_spmm_max_bwd: bad N/E/F");
  if (N == 0) return SPGNN_OK;
  if (!out_indptr || !g_out || !arg || !g_x || (E > 0 && (!out_indices || !out_pos)))
    return fail(SPGNN_ERR_NULLPTR, "spgnn_spmm_max_bwd: null pointer");
  if (g_out_stride < F || arg_stride < F || g_x_stride < F)
    return fail(SPGNN_ERR_STRIDE, "spgnn_spmm_max_bwd: row stride smaller than row");
  hipStream_t st = (hipStream_t)stream;
  SpmmMaxBwd a{out_indptr, out_indices, out_pos, g_out, g_out_stride, arg, arg_stride, g_x, g_x_stride, N, F, 0, nullptr, nullptr,
               nullptr, 0, nullptr};
  int T, R;
  if (pick_team(F, T, R, kSpmmNarrow) && vec_ok(g_out, g_out_stride) && vec_ok(arg, arg_stride) && vec_ok(g_x, g_x_stride)) {
    a.T = T;
    DISPATCH_R(T, R, spmm_max_bwd_vec, dim3(grid_for(N, kBlock / T)), dim3(kBlock), 0, st, a);
  } else {
    hipLaunchKernelGGL(spmm_max_bwd_scalar, dim3(scalar_grid(N * F)), dim3(kBlock), 0, st, a);
  }
  return check_launch("spgnn_spmm_max_bwd");
}

int32_t spgnn_spmm_max_u8_supported(int32_t F) { int T, R; return pick_team(F, T, R, kSpmmNarrow) ? 1 : 0; }

int spgnn_spmm_max_fwd_u8(const int32_t* indptr, const int32_t* indices, const float* x, int64_t x_stride, float* out,
                          int64_t out_stride, uint8_t* arg, int64_t arg_stride, int64_t N, int64_t E, int32_t F,
                          spgnn_stream_t stream) {
  int T, R;
  if (N < 0 || E < 0 || F <= 0 || !pick_team(F, T, R, kSpmmNarrow)) return fail(SPGNN_ERR_SHAPE, "spgnn_spmm_max_fwd_u8: bad N/E/F (see spgnn_spmm_max_u8_supported)");
  if (N == 0) return SPGNN_OK;
  if (!indptr || !x || !out || !arg || (E > 0 && !indices)) return fail(SPGNN_ERR_NULLPTR, "spgnn_spmm_max_fwd_u8: null pointer");
  if (x_stride < F || out_stride < F || arg_stride < F) return fail(SPGNN_ERR_STRIDE, "spgnn_spmm_max_fwd_u8: row stride smaller than row");
  if (!vec_ok(x, x_stride) || !vec_ok(out, out_stride) || (reinterpret_cast<uintptr_t>(arg) & 3) || (arg_stride & 3))
    return fail(SPGNN_ERR_STRIDE, "spgnn_spmm_max_fwd_u8: rows must be 16-byte (arg: 4-byte) aligned");
  SpmmMaxFwd a{indptr, indices, x, x_stride, out, out_stride, nullptr, arg_stride, N, F, T, arg};
  DISPATCH_R(T, R, spmm_max_fwd_vec_u8, dim3(grid_for(N, kBlock / T)), dim3(kBlock), 0, (hipStream_t)stream, a);
  return check_launch("spgnn_spmm_max_fwd_u8");
}

static int spmm_max_bwd_u8_impl(const int32_t* indptr, const int32_t* out_indptr, const int32_t* out_indices, const int32_t* out_pos,
                                const float* g_out, int64_t g_out_stride, const uint8_t* arg, int64_t arg_stride, float* g_x,
                                int64_t g_x_stride, const float* relu_of, int64_t relu_stride, float* absmax, int64_t N, int64_t E,
                                int32_t F, spgnn_stream_t stream) {
  int T, R;
  if (N < 0 || E < 0 || F <= 0 || !pick_team(F, T, R, kSpmmNarrow)) return fail(SPGNN_ERR_SHAPE, "spgnn_spmm_max_bwd_u8: bad N/E/F (see spgnn_spmm_max_u8_supported)");
  if (N == 0) return SPGNN_OK;
  if (!indptr || !out_indptr || !g_out || !arg || !g_x || (E > 0 && (!out_indices || !out_pos)))
    return fail(SPGNN_ERR_NULLPTR, "spgnn_spmm_max_bwd_u8: null pointer");
  if (g_out_stride < F || arg_stride < F || g_x_stride < F || (relu_of && relu_stride < F))
    return fail(SPGNN_ERR_STRIDE, "spgnn_spmm_max_bwd_u8: row stride smaller than row");
  if (!vec_ok(g_out, g_out_stride) || !vec_ok(g_x, g_x_stride) || !vec_ok(relu_of, relu_stride) || (reinterpret_cast<uintptr_t>(arg) & 3) ||
      (arg_stride & 3))
    return fail(SPGNN_ERR_STRIDE, "spgnn_spmm_max_bwd_u8: rows must be 16-byte (arg: 4-byte) aligned");
  SpmmMaxBwd a{out_indptr, out_indices, out_pos, g_out, g_out_stride, nullptr, arg_stride, g_x, g_x_stride, N, F, T, arg, indptr,
               relu_of, relu_stride, absmax};
  DISPATCH_R(T, R, spmm_max_bwd_vec_u8, dim3(grid_for(N, kBlock / T)), dim3(kBlock), 0, (hipStream_t)stream, a);
  return check_launch("spgnn_spmm_max_bwd_u8");
}

int spgnn_spmm_max_bwd_u8(const int32_t* indptr, const int32_t* out_indptr, const int32_t* out_indices, const int32_t* out_pos,
                          const float* g_out, int64_t g_out_stride, const uint8_t* arg, int64_t arg_stride, float* g_x,
                          int64_t g_x_stride, int64_t N, int64_t E, int32_t F, spgnn_stream_t stream) {
  return spmm_max_bwd_u8_impl(indptr, out_indptr, out_indices, out_pos, g_out, g_out_stride, arg, arg_stride, g_x, g_x_stride, nullptr, 0,
                              nullptr, N, E, F, stream);
}

int spgnn_spmm_max_bwd_u8_relu(const int32_t* indptr, const int32_t* out_indptr, const int32_t* out_indices, const int32_t* out_pos,
                               const float* g_out, int64_t g_out_stride, const uint8_t* arg, int64_t arg_stride, float* g_x,
                               int64_t g_x_stride, const float* relu_out, int64_t relu_out_stride, float* absmax, int64_t N,
                               int64_t E, int32_t F, spgnn_stream_t stream) {
  if (!relu_out) return fail(SPGNN_ERR_NULLPTR, "spgnn_spmm_max_bwd_u8_relu: null pointer");
  return spmm_max_bwd_u8_impl(indptr, out_indptr, out_indices, out_pos, g_out, g_out_stride, arg, arg_stride, g_x, g_x_stride, relu_out,
                              relu_out_stride, absmax, N, E, F, stream);
}

}  // extern "C"

// Rows per wave.  Measured at N = 76 410 (tools/scores_ab.py): the 22-column classifier product (two column groups, K = 1024)
// 105 us with 16 rows per wave, 91 with 32, 101 with 64 (too few waves); the 2H <= 16 column score products are fastest
// with 16 (K = 1063: 82 / 89 / 104 us) - their W fragment is one group, there is little to reuse.
constexpr int kScoresRG = 2;      // row groups of 16 per wave in the wide form
template <typename ST>
static void scores_fwd_launch(const ST* x, int64_t x_stride, const float* w, int32_t Kp, float* s, int64_t s_stride, int64_t N,
                              int32_t K, int32_t J, float* absmax, const float* bias, hipStream_t st) {
  const bool wide = kScoresRG > 1 && J > 16 && N >= 16 * kScoresRG * 1024;   // >= 1024 waves of the wide form
  const int rg = wide ? kScoresRG : 1;
  const int64_t waves = (N + 16 * rg - 1) / (16 * rg);                        // row sets
  // the four waves of a block split k (see the kernel): 1024 -> 22 at N = 76 410 100 -> 81-83 us (tools/scores_ab.py, one
  // process); with 64 rows per block instead of 32: 86
  // Small batches (the reference's 64-tree batch: N = 9 641) have too few 16-row sets to fill the chip with one wave each - 604
  // waves on 1024 SIMDs, 26 us for 39 MB - so the deep classifier product splits k over the block's four waves there too, with
  // one row group per wave (SPGNN_SCORES_NO_SMALL_KSPLIT: the A/B switch).
#ifndef SPGNN_SCORES_NO_SMALL_KSPLIT
  const bool ksplit_small = !wide && J > 16 && K >= 512;
#else
  const bool ksplit_small = false;
#endif
  const bool ksplit = (wide && K >= 512) || ksplit_small;
  const dim3 grid((unsigned)(ksplit ? waves : (waves + kBlock / 64 - 1) / (kBlock / 64))), block(kBlock);
#define X(NG_, RG_, KS_) hipLaunchKernelGGL((scores_fwd_mfma<ST, NG_, RG_, KS_>), grid, block, 0, st, x, x_stride, w, Kp, s, s_stride, N, K, J, absmax, bias)
  if (J <= 16) X(1, 1, 1);
  else if (ksplit_small) X(2, 1, kBlock / 64);
  else if (ksplit) X(2, kScoresRG, kBlock / 64);
  else if (wide) X(2, kScoresRG, 1);
  else X(2, 1, 1);
#undef X
}

extern "C" {

int spgnn_scores_fwd(const float* x, int64_t x_stride, const float* w, int32_t Kp, float* s, int64_t s_stride,
                     float* absmax, const float* bias, int64_t N, int32_t K, int32_t J, spgnn_stream_t stream) {
  if (N < 0 || K <= 0 || J <= 0 || J > 32 || Kp < K || (Kp & 15)) return fail(SPGNN_ERR_SHAPE, "spgnn_scores_fwd: bad N/K/Kp/J");
  if (N == 0) return SPGNN_OK;
  if (!x || !w || !s) return fail(SPGNN_ERR_NULLPTR, "spgnn_scores_fwd: null pointer");
  if (x_stride < K || s_stride < J || (x_stride & 3) || !aligned16(x) || !aligned16(w))
    return fail(SPGNN_ERR_STRIDE, "spgnn_scores_fwd: x rows and w must be 16-byte aligned (stride % 4 == 0)");
  scores_fwd_launch<float>(x, x_stride, w, Kp, s, s_stride, N, K, J, absmax, bias, (hipStream_t)stream);
  return check_launch("spgnn_scores_fwd");
}

int spgnn_scores_fwd_bf16(const uint16_t* x, int64_t x_stride, const float* w, int32_t Kp, float* s, int64_t s_stride,
                          const float* bias, int64_t N, int32_t K, int32_t J, spgnn_stream_t stream) {
  if (N < 0 || K <= 0 || J <= 0 || J > 32 || Kp < K || (Kp & 15)) return fail(SPGNN_ERR_SHAPE, "spgnn_scores_fwd_bf16: bad N/K/Kp/J");
  if (N == 0) return SPGNN_OK;
  if (!x || !w || !s) return fail(SPGNN_ERR_NULLPTR, "spgnn_scores_fwd_bf16: null pointer");
  if (x_stride < K || s_stride < J || (x_stride & 3) || (reinterpret_cast<uintptr_t>(x) & 7) || !aligned16(w))
    return fail(SPGNN_ERR_STRIDE, "spgnn_scores_fwd_bf16: x rows must be 8-byte aligned (stride % 4 == 0), w 16-byte aligned");
  scores_fwd_launch<bf16s>(reinterpret_cast<const bf16s*>(x), x_stride, w, Kp, s, s_stride, N, K, J, nullptr, bias, (hipStream_t)stream);
  return check_launch("spgnn_scores_fwd_bf16");
}

static int padded_j(int J) { return J <= 2 ? 2 : J <= 4 ? 4 : J <= 8 ? 8 : J <= 16 ? 16 : J <= 24 ? 24 : 32; }

int spgnn_scores_bwd_w(const float* gs, int64_t gs_stride, const float* x, int64_t x_stride, float* part,
                       int32_t splits, int32_t Kp, int64_t N, int32_t K, int32_t J, spgnn_stream_t stream) {
  if (N < 0 || K <= 0 || splits <= 0 || Kp < K || (Kp & 15) || J <= 0 || J > 32)
    return fail(SPGNN_ERR_SHAPE, "spgnn_scores_bwd_w: bad N/K/Kp/splits/J");
  if (!gs || !x || !part) return fail(SPGNN_ERR_NULLPTR, "spgnn_scores_bwd_w: null pointer");
  if (x_stride < K || gs_stride < J || (x_stride & 3) || !aligned16(x) || !aligned16(part))
    return fail(SPGNN_ERR_STRIDE, "spgnn_scores_bwd_w: x rows must be 16-byte aligned (stride % 4 == 0)");
  const int64_t rps = (N + splits - 1) / splits;
  const int bw = K >= 1024 ? 4 : (K + 255) / 256;        // waves per block = 256-column pieces of a row (narrow rows: no idle waves)
  const dim3 grid((unsigned)((K + 256 * bw - 1) / (256 * bw)), (unsigned)splits), block(64 * bw);
  hipStream_t st = (hipStream_t)stream;
#define X(JP) hipLaunchKernelGGL((scores_bwd_w_kernel<float, JP>), grid, block, 0, st, gs, gs_stride, x, x_stride, part, Kp, N, K, rps, J)
  switch (padded_j(J)) { case 2: X(2); break; case 4: X(4); break; case 8: X(8); break; case 16: X(16); break;
                         case 24: X(24); break; default: X(32); break; }
#undef X
  return check_launch("spgnn_scores_bwd_w");
}

int spgnn_scores_bwd_w_pair(const float* gs0, int64_t gs0_stride, const float* x0, int64_t x0_stride, float* part0, int32_t splits0,
                            int32_t Kp0, int32_t K0, int32_t J0, const float* gs1, int64_t gs1_stride, const float* x1,
                            int64_t x1_stride, float* part1, int32_t splits1, int32_t Kp1, int32_t K1, int32_t J1, int64_t N,
                            spgnn_stream_t stream) {
  if (N < 0 || K0 <= 0 || K1 <= 0 || splits0 <= 0 || splits1 <= 0 || Kp0 < K0 || Kp1 < K1 || ((Kp0 | Kp1) & 15) || J0 <= 0 || J1 <= 0 ||
      J0 > 8 || J1 > 8)
    return fail(SPGNN_ERR_SHAPE, "spgnn_scores_bwd_w_pair: bad N/K/Kp/splits/J (J <= 8)");
  if (!gs0 || !x0 || !part0 || !gs1 || !x1 || !part1) return fail(SPGNN_ERR_NULLPTR, "spgnn_scores_bwd_w_pair: null pointer");
  if (x0_stride < K0 || gs0_stride < J0 || (x0_stride & 3) || !aligned16(x0) || !aligned16(part0) || x1_stride < K1 || gs1_stride < J1 ||
      (x1_stride & 3) || !aligned16(x1) || !aligned16(part1))
    return fail(SPGNN_ERR_STRIDE, "spgnn_scores_bwd_w_pair: x rows must be 16-byte aligned (stride % 4 == 0)");
  const int bw0 = K0 >= 1024 ? 4 : (K0 + 255) / 256, bw1 = K1 >= 1024 ? 4 : (K1 + 255) / 256;
  const int bw = bw0 > bw1 ? bw0 : bw1;                    // one block shape for both: the wider problem's
  ScoresBwdW p0{gs0, gs0_stride, x0, x0_stride, part0, Kp0, N, K0, (N + splits0 - 1) / splits0, J0, (K0 + 256 * bw - 1) / (256 * bw)};
  ScoresBwdW p1{gs1, gs1_stride, x1, x1_stride, part1, Kp1, N, K1, (N + splits1 - 1) / splits1, J1, (K1 + 256 * bw - 1) / (256 * bw)};
  const dim3 grid((unsigned)(p0.gx > p1.gx ? p0.gx : p1.gx), (unsigned)(splits0 + splits1)), block(64 * bw);
  hipStream_t st = (hipStream_t)stream;
  const int jp = padded_j(J0 > J1 ? J0 : J1);
  if (jp <= 2) hipLaunchKernelGGL((scores_bwd_w_pair_kernel<2>), grid, block, 0, st, p0, p1, (unsigned)splits0);
  else if (jp <= 4) hipLaunchKernelGGL((scores_bwd_w_pair_kernel<4>), grid, block, 0, st, p0, p1, (unsigned)splits0);
  else hipLaunchKernelGGL((scores_bwd_w_pair_kernel<8>), grid, block, 0, st, p0, p1, (unsigned)splits0);
  return check_launch("spgnn_scores_bwd_w_pair");
}

int spgnn_scores_bwd_w_multi(const spgnn_scores_bwd_w_job* jobs, int32_t n_jobs, int64_t N, int32_t x_is_bf16, spgnn_stream_t stream) {
  if (n_jobs < 0 || n_jobs > kMaxScoreJobs || N < 0) return fail(SPGNN_ERR_SHAPE, "spgnn_scores_bwd_w_multi: n_jobs outside [0, 8] or N < 0");
  if (n_jobs == 0 || N == 0) return SPGNN_OK;
  if (!jobs) return fail(SPGNN_ERR_NULLPTR, "spgnn_scores_bwd_w_multi: null pointer");
  ScoresBwdWJobs a{};
  int bw = 1, jmax = 0;
  for (int i = 0; i < n_jobs; ++i) {
    const spgnn_scores_bwd_w_job& q = jobs[i];
    if (q.K <= 0 || q.splits <= 0 || q.Kp < q.K || (q.Kp & 15) || q.J <= 0 || q.J > 8)
      return fail(SPGNN_ERR_SHAPE, "spgnn_scores_bwd_w_multi: bad K/Kp/splits/J (J <= 8)");
    if (!q.g_s || !q.x || !q.partials) return fail(SPGNN_ERR_NULLPTR, "spgnn_scores_bwd_w_multi: null pointer");
    const bool rows_ok = x_is_bf16 ? vec_ok_t(reinterpret_cast<const bf16s*>(q.x), q.x_stride) : (!(q.x_stride & 3) && aligned16(q.x));
    if (q.x_stride < q.K || q.g_s_stride < q.J || !rows_ok || !aligned16(q.partials))
      return fail(SPGNN_ERR_STRIDE, "spgnn_scores_bwd_w_multi: x rows must be vector aligned (stride % 4 == 0), partials 16-byte aligned");
    const int b = q.K >= 1024 ? 4 : (q.K + 255) / 256;
    bw = b > bw ? b : bw;
    jmax = q.J > jmax ? q.J : jmax;
  }
  unsigned y = 0; int gx = 0;
  for (int i = 0; i < n_jobs; ++i) {
    const spgnn_scores_bwd_w_job& q = jobs[i];
    a.j[i] = ScoresBwdW{q.g_s, q.g_s_stride, reinterpret_cast<const float*>(q.x), q.x_stride, q.partials, q.Kp, N, q.K,
                        (N + q.splits - 1) / q.splits, q.J, (q.K + 256 * bw - 1) / (256 * bw)};
    a.y0[i] = y; y += (unsigned)q.splits;
    gx = a.j[i].gx > gx ? a.j[i].gx : gx;
  }
  a.y0[n_jobs] = y; a.n = n_jobs;
  const dim3 grid((unsigned)gx, y), block(64 * bw);
  hipStream_t st = (hipStream_t)stream;
  const int jp = padded_j(jmax);
#define X(ST_, JP) hipLaunchKernelGGL((scores_bwd_w_multi_kernel<ST_, JP>), grid, block, 0, st, a)
  if (x_is_bf16) { if (jp <= 2) X(bf16s, 2); else if (jp <= 4) X(bf16s, 4); else X(bf16s, 8); }
  else { if (jp <= 2) X(float, 2); else if (jp <= 4) X(float, 4); else X(float, 8); }
#undef X
  return check_launch("spgnn_scores_bwd_w_multi");
}

int spgnn_scores_bwd_w_bf16(const float* gs, int64_t gs_stride, const uint16_t* x, int64_t x_stride, float* part,
                            int32_t splits, int32_t Kp, int64_t N, int32_t K, int32_t J, spgnn_stream_t stream) {
  if (N < 0 || K <= 0 || splits <= 0 || Kp < K || (Kp & 15) || J <= 0 || J > 32)
    return fail(SPGNN_ERR_SHAPE, "spgnn_scores_bwd_w_bf16: bad N/K/Kp/splits/J (J <= 8)");
  if (!gs || !x || !part) return fail(SPGNN_ERR_NULLPTR, "spgnn_scores_bwd_w_bf16: null pointer");
  const bf16s* x_ = reinterpret_cast<const bf16s*>(x);
  if (x_stride < K || gs_stride < J || !vec_ok_t(x_, x_stride) || !aligned16(part))
    return fail(SPGNN_ERR_STRIDE, "spgnn_scores_bwd_w_bf16: x rows must be 8-byte aligned (stride % 4 == 0)");
  const int64_t rps = (N + splits - 1) / splits;
  const int bw = K >= 1024 ? 4 : (K + 255) / 256;        // waves per block = 256-column pieces of a row (narrow rows: no idle waves)
  const dim3 grid((unsigned)((K + 256 * bw - 1) / (256 * bw)), (unsigned)splits), block(64 * bw);
  hipStream_t st = (hipStream_t)stream;
#define X(JP) hipLaunchKernelGGL((scores_bwd_w_kernel<bf16s, JP>), grid, block, 0, st, gs, gs_stride, x_, x_stride, part, Kp, N, K, rps, J)
  switch (padded_j(J)) { case 2: X(2); break; case 4: X(4); break; case 8: X(8); break; case 16: X(16); break;
                         case 24: X(24); break; default: X(32); break; }
#undef X
  return check_launch("spgnn_scores_bwd_w_bf16");
}

constexpr int64_t kBwdXWaves = 8192;
int spgnn_scores_bwd_x(const float* gs, int64_t gs_stride, const float* w, int32_t Kp, float* gx, int64_t gx_stride,
                       int32_t accumulate, int64_t N, int32_t K, int32_t J, spgnn_stream_t stream) {
  if (N < 0 || K <= 0 || Kp < K || (Kp & 15) || J <= 0 || J > 32) return fail(SPGNN_ERR_SHAPE, "spgnn_scores_bwd_x: bad N/K/Kp/J");
  if (N == 0) return SPGNN_OK;
  if (!gs || !w || !gx) return fail(SPGNN_ERR_NULLPTR, "spgnn_scores_bwd_x: null pointer");
  if (gx_stride < K || gs_stride < J || (gx_stride & 3) || !aligned16(gx) || !aligned16(w))
    return fail(SPGNN_ERR_STRIDE, "spgnn_scores_bwd_x: gx rows must be 16-byte aligned (stride % 4 == 0)");
  // every row waits for its own scalar score loads, so the latency is hidden by waves, not by unrolling (measured):
  // ~8 waves per SIMD
  int64_t splits = kBwdXWaves / ((K + 255) / 256);
  if (splits < 1) splits = 1;
  if (splits > N) splits = N;
  const int64_t rps = (N + splits - 1) / splits;
  const dim3 grid((unsigned)((K + 255) / 256), (unsigned)((N + rps - 1) / rps)), block(64);
  hipStream_t st = (hipStream_t)stream;
#define X(JP) hipLaunchKernelGGL((scores_bwd_x_kernel<float, JP>), grid, block, 0, st, gs, gs_stride, w, Kp, gx, gx_stride, N, K, rps, J, accumulate)
  switch (padded_j(J)) { case 2: X(2); break; case 4: X(4); break; case 8: X(8); break; case 16: X(16); break;
                         case 24: X(24); break; default: X(32); break; }
#undef X
  return check_launch("spgnn_scores_bwd_x");
}

int spgnn_scores_bwd_x_bf16(const float* gs, int64_t gs_stride, const float* w, int32_t Kp, uint16_t* gx, int64_t gx_stride,
                            int32_t accumulate, int64_t N, int32_t K, int32_t J, spgnn_stream_t stream) {
  if (N < 0 || K <= 0 || Kp < K || (Kp & 15) || J <= 0 || J > 32) return fail(SPGNN_ERR_SHAPE, "spgnn_scores_bwd_x_bf16: bad N/K/Kp/J");
  if (N == 0) return SPGNN_OK;
  if (!gs || !w || !gx) return fail(SPGNN_ERR_NULLPTR, "spgnn_scores_bwd_x_bf16: null pointer");
  bf16s* gx_ = reinterpret_cast<bf16s*>(gx);
  if (gx_stride < ((K + 3) & ~3) || gs_stride < J || !vec_ok_t(gx_, gx_stride) || !aligned16(w))
    return fail(SPGNN_ERR_STRIDE, "spgnn_scores_bwd_x_bf16: gx rows must be 8-byte aligned with a stride that is a multiple of 4 >= K");
  int64_t splits = kBwdXWaves / ((K + 255) / 256);
  if (splits < 1) splits = 1;
  if (splits > N) splits = N;
  const int64_t rps = (N + splits - 1) / splits;
  const dim3 grid((unsigned)((K + 255) / 256), (unsigned)((N + rps - 1) / rps)), block(64);
  hipStream_t st = (hipStream_t)stream;
#define X(JP) hipLaunchKernelGGL((scores_bwd_x_kernel<bf16s, JP>), grid, block, 0, st, gs, gs_stride, w, Kp, gx_, gx_stride, N, K, rps, J, accumulate)
  switch (padded_j(J)) { case 2: X(2); break; case 4: X(4); break; case 8: X(8); break; case 16: X(16); break;
                         case 24: X(24); break; default: X(32); break; }
#undef X
  return check_launch("spgnn_scores_bwd_x_bf16");
}

int spgnn_tree_distance_encoding(const int32_t* out_indptr, const int32_t* out_indices, const int64_t* tree_ptr,
                                 const int32_t* anchors, int32_t num_anchors, float* pos_enc, int64_t pos_enc_stride,
                                 int32_t* diameters, int64_t num_trees, int64_t max_tree_nodes, spgnn_stream_t stream) {
  if (num_trees < 0 || num_anchors <= 0 || max_tree_nodes < 0) return fail(SPGNN_ERR_SHAPE, "spgnn_tree_distance_encoding: bad sizes");
  if (num_trees == 0) return SPGNN_OK;
  if (max_tree_nodes > kPeMaxNodes || max_tree_nodes > 65535)
    return fail(SPGNN_ERR_SHAPE, "spgnn_tree_distance_encoding: a tree exceeds 2048 nodes (LDS-resident distances)");
  if (!out_indptr || !out_indices || !tree_ptr || !anchors || !pos_enc)
    return fail(SPGNN_ERR_NULLPTR, "spgnn_tree_distance_encoding: null pointer");
  if (pos_enc_stride < num_anchors) return fail(SPGNN_ERR_STRIDE, "spgnn_tree_distance_encoding: row stride smaller than row");
  hipLaunchKernelGGL(tree_distance_encoding, dim3((unsigned)num_trees), dim3(256), 0, (hipStream_t)stream, out_indptr,
                     out_indices, tree_ptr, anchors, num_anchors, pos_enc, pos_enc_stride, diameters);
  return check_launch("spgnn_tree_distance_encoding");
}

int64_t spgnn_tree_anchors_workspace(int64_t num_trees, int32_t num_distal, int64_t max_tree_nodes) {
  return num_trees * (int64_t)num_distal * 18 * max_tree_nodes * (int64_t)sizeof(uint16_t) + num_trees * max_tree_nodes;
}

int spgnn_tree_anchors(const float* prob, int64_t prob_stride, const int32_t* out_indptr, const int32_t* out_indices,
                       const int64_t* tree_ptr, int64_t num_trees, int64_t num_nodes, int64_t max_tree_nodes, int32_t num_labels,
                       int32_t num_distal, int32_t* anchors, void* workspace, spgnn_stream_t stream) {
  if (num_trees < 0 || num_nodes < 0 || num_labels <= 0 || num_distal < 0 || num_distal > num_labels || max_tree_nodes < 0 ||
      max_tree_nodes >= 65535)
    return fail(SPGNN_ERR_SHAPE, "spgnn_tree_anchors: bad sizes (trees of at most 65534 nodes, num_distal <= num_labels)");
  if (num_trees == 0) return SPGNN_OK;
  if (!prob || !out_indptr || !out_indices || !tree_ptr || !anchors || !workspace)
    return fail(SPGNN_ERR_NULLPTR, "spgnn_tree_anchors: null pointer");
  if (prob_stride <= num_labels) return fail(SPGNN_ERR_STRIDE, "spgnn_tree_anchors: prob needs num_labels + 1 columns");
  hipStream_t st = (hipStream_t)stream;
  const int A = num_labels + num_distal;
  uint16_t* ws = reinterpret_cast<uint16_t*>(workspace);
  const int64_t per_thread = 18 * max_tree_nodes;
  uint8_t* taken = reinterpret_cast<uint8_t*>(ws + num_trees * (int64_t)num_distal * per_thread);
  hipLaunchKernelGGL(greedy_anchors_kernel, dim3((unsigned)num_trees), dim3(64), 0, st, prob, prob_stride, tree_ptr, anchors, A,
                     num_labels, taken);
  if (num_distal > 0) {
    const int64_t threads = num_trees * num_distal;
    hipLaunchKernelGGL(distal_leafs_kernel, dim3((unsigned)((threads + 63) / 64)), dim3(64), 0, st, out_indptr, out_indices, tree_ptr,
                       anchors, A, num_labels, num_distal, num_trees, ws, per_thread, (int)max_tree_nodes);
  }
  return check_launch("spgnn_tree_anchors");
}

int spgnn_masked_ce(const float* logits, int64_t logits_stride, const int64_t* labels, const float* draws,
                    const float* sampling_p, const float* class_weight, float* partials, float* g_logits, int64_t g_stride,
                    int64_t N, int32_t C, spgnn_stream_t stream) {
  if (!draws) return fail(SPGNN_ERR_NULLPTR, "spgnn_masked_ce: null pointer");
  return spgnn_masked_ce_step(logits, logits_stride, labels, draws, 0, nullptr, sampling_p, class_weight, partials, nullptr, nullptr,
                              g_logits, g_stride, nullptr, nullptr, N, C, stream);
}

static int masked_ce_launch(const float* logits, int64_t logits_stride, const int64_t* labels, const float* draws, uint64_t draw_seed,
                            const int64_t* seed_offset, const float* sampling_p, const float* class_weight, float* partials,
                            float* sums, uint32_t* ticket, float* g_logits, int64_t g_stride, float* colsum_partials, float* g_colsum,
                            int64_t N, int32_t C, const int32_t* rows, const int32_t* rows_cnt, spgnn_stream_t stream) {
  if (N < 0 || C <= 0) return fail(SPGNN_ERR_SHAPE, "spgnn_masked_ce: bad N/C");
  if (N == 0) {
    if (sums) { const hipError_t e = hipMemsetAsync(sums, 0, 2 * sizeof(float), (hipStream_t)stream); if (e != hipSuccess) return fail(-(1000 + (int)e), "spgnn_masked_ce_step: hipMemsetAsync"); }
    return SPGNN_OK;
  }
  if (!logits || !labels || !class_weight || !partials || (sums && !ticket)) return fail(SPGNN_ERR_NULLPTR, "spgnn_masked_ce: null pointer");
  if (logits_stride < C || (g_logits && g_stride < C)) return fail(SPGNN_ERR_STRIDE, "spgnn_masked_ce: row stride smaller than row");
  hipLaunchKernelGGL(masked_ce_kernel, dim3((unsigned)((N + kBlock - 1) / kBlock)), dim3(kBlock), 0, (hipStream_t)stream, logits,
                     logits_stride, labels, draws, draw_seed, seed_offset, sampling_p, class_weight, partials, sums, ticket, g_logits,
                     g_stride, (sums && g_colsum && C <= 32) ? colsum_partials : nullptr, g_colsum, N, C, rows, rows_cnt);
  return check_launch("spgnn_masked_ce");
}

int spgnn_masked_ce_step(const float* logits, int64_t logits_stride, const int64_t* labels, const float* draws, uint64_t draw_seed,
                         const int64_t* seed_offset, const float* sampling_p, const float* class_weight, float* partials,
                         float* sums, uint32_t* ticket, float* g_logits, int64_t g_stride, float* colsum_partials, float* g_colsum,
                         int64_t N, int32_t C, spgnn_stream_t stream) {
  if (!sampling_p) return fail(SPGNN_ERR_NULLPTR, "spgnn_masked_ce: null pointer");
  return masked_ce_launch(logits, logits_stride, labels, draws, draw_seed, seed_offset, sampling_p, class_weight, partials, sums, ticket,
                          g_logits, g_stride, colsum_partials, g_colsum, N, C, nullptr, nullptr, stream);
}

int spgnn_masked_ce_step_flagged(const float* logits, int64_t logits_stride, const int64_t* labels, const float* draws, uint64_t draw_seed,
                                 const int64_t* seed_offset, const float* sampling_p, const int32_t* rows_cnt, const float* class_weight,
                                 float* partials, float* sums, uint32_t* ticket, float* g_logits, int64_t g_stride,
                                 float* colsum_partials, float* g_colsum, int64_t N, int32_t C, spgnn_stream_t stream) {
  if (!sampling_p || !rows_cnt) return fail(SPGNN_ERR_NULLPTR, "spgnn_masked_ce_step_flagged: null pointer");
  return masked_ce_launch(logits, logits_stride, labels, draws, draw_seed, seed_offset, sampling_p, class_weight, partials, sums, ticket,
                          g_logits, g_stride, colsum_partials, g_colsum, N, C, nullptr, rows_cnt, stream);
}

int spgnn_masked_ce_rows(const float* logits, int64_t logits_stride, const int64_t* labels, const int32_t* rows, const int32_t* rows_cnt,
                         const float* class_weight, float* partials, float* sums, uint32_t* ticket, float* g_logits, int64_t g_stride,
                         float* colsum_partials, float* g_colsum, int64_t cap, int32_t C, spgnn_stream_t stream) {
  if (!rows || !rows_cnt) return fail(SPGNN_ERR_NULLPTR, "spgnn_masked_ce_rows: null pointer");
  return masked_ce_launch(logits, logits_stride, labels, nullptr, 0, nullptr, nullptr, class_weight, partials, sums, ticket, g_logits,
                          g_stride, colsum_partials, g_colsum, cap, C, rows, rows_cnt, stream);
}

int spgnn_loss_rows(const float* draws, uint64_t draw_seed, const int64_t* seed_offset, const float* sampling_p, int64_t N,
                    int32_t* block_counts, int32_t cap, int32_t* idx, int32_t* inv, int32_t* cnt_flag, spgnn_stream_t stream) {
  if (N < 0 || cap <= 0 || N > (1ll << 30)) return fail(SPGNN_ERR_SHAPE, "spgnn_loss_rows: bad N / cap");
  if (!sampling_p || !block_counts || !idx || !inv || !cnt_flag) return fail(SPGNN_ERR_NULLPTR, "spgnn_loss_rows: null pointer");
  const unsigned nb = N > 0 ? scalar_grid(N) : 1u;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(loss_rows_count_kernel, dim3(nb), dim3(kBlock), 0, st, draws, draw_seed, seed_offset, sampling_p, N, block_counts);
  hipLaunchKernelGGL(loss_rows_write_kernel, dim3(nb), dim3(kBlock), 0, st, draws, draw_seed, seed_offset, sampling_p, N, block_counts,
                     cap, idx, inv, cnt_flag);
  return check_launch("spgnn_loss_rows");
}

int spgnn_gather_rows(const float* src, int64_t src_stride, const int32_t* idx, const int32_t* cnt_flag, int64_t cap, int32_t cols,
                      float* dst, int64_t dst_stride, spgnn_stream_t stream) {
  if (cap < 0 || cols <= 0 || (cols & 3)) return fail(SPGNN_ERR_SHAPE, "spgnn_gather_rows: cols must be a positive multiple of 4");
  if (cap == 0) return SPGNN_OK;
  if (!src || !idx || !cnt_flag || !dst) return fail(SPGNN_ERR_NULLPTR, "spgnn_gather_rows: null pointer");
  if (src_stride < cols || dst_stride < cols || (src_stride & 3) || (dst_stride & 3) || !aligned16(src) || !aligned16(dst))
    return fail(SPGNN_ERR_STRIDE, "spgnn_gather_rows: rows must be 16-byte aligned (stride % 4 == 0)");
  hipLaunchKernelGGL(gather_rows_kernel, dim3(scalar_grid(cap * (cols / 4))), dim3(kBlock), 0, (hipStream_t)stream, src, src_stride, idx,
                     cnt_flag, cap, cols / 4, dst, dst_stride);
  return check_launch("spgnn_gather_rows");
}

int spgnn_expand_rows(const float* src, int64_t src_stride, const int32_t* inv, int64_t N, int32_t cols, float* dst, int64_t dst_stride,
                      spgnn_stream_t stream) {
  if (N < 0 || cols <= 0 || (cols & 3)) return fail(SPGNN_ERR_SHAPE, "spgnn_expand_rows: cols must be a positive multiple of 4");
  if (N == 0) return SPGNN_OK;
  if (!src || !inv || !dst) return fail(SPGNN_ERR_NULLPTR, "spgnn_expand_rows: null pointer");
  if (src_stride < cols || dst_stride < cols || (src_stride & 3) || (dst_stride & 3) || !aligned16(src) || !aligned16(dst))
    return fail(SPGNN_ERR_STRIDE, "spgnn_expand_rows: rows must be 16-byte aligned (stride % 4 == 0)");
  hipLaunchKernelGGL(expand_rows_kernel, dim3(scalar_grid(N * (cols / 4))), dim3(kBlock), 0, (hipStream_t)stream, src, src_stride, inv, N,
                     cols / 4, dst, dst_stride);
  return check_launch("spgnn_expand_rows");
}

int spgnn_sample_neighbors(const int32_t* indptr, const int32_t* indices, const int32_t* eid, int64_t num_nodes,
                           const int64_t* seeds, int64_t num_seeds, int32_t fanout, const int32_t* out_indptr, uint64_t seed,
                           int32_t* local, int32_t* out_src, int32_t* out_eid, int32_t* flag, spgnn_stream_t stream) {
  if (num_nodes < 0 || num_seeds < 0) return fail(SPGNN_ERR_SHAPE, "spgnn_sample_neighbors: negative size");
  if (num_seeds == 0) return SPGNN_OK;
  if (num_nodes == 0) return fail(SPGNN_ERR_SHAPE, "spgnn_sample_neighbors: seeds given for an empty graph");
  if (!indptr || !indices || !seeds || !out_indptr || !local || !out_src || !flag)
    return fail(SPGNN_ERR_NULLPTR, "spgnn_sample_neighbors: null pointer");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(block_mark_seeds, dim3(scalar_grid(num_seeds)), dim3(kBlock), 0, st, seeds, num_seeds, num_nodes, local);
  hipLaunchKernelGGL(sample_neighbors_kernel, dim3(scalar_grid(num_seeds)), dim3(kBlock), 0, st, indptr, indices, eid, seeds,
                     num_seeds, num_nodes, fanout, out_indptr, seed, local, out_src, out_eid, flag);
  return check_launch("spgnn_sample_neighbors");
}

int spgnn_block_relabel(const int32_t* flag, const int32_t* rank, int32_t* local, int64_t num_nodes, int64_t num_seeds,
                        const int32_t* out_src, int64_t num_edges, int64_t* extra_nodes, int32_t* src_local,
                        spgnn_stream_t stream) {
  if (num_nodes < 0 || num_seeds < 0 || num_edges < 0) return fail(SPGNN_ERR_SHAPE, "spgnn_block_relabel: negative size");
  if (num_nodes == 0) return SPGNN_OK;
  if (!flag || !rank || !local || !extra_nodes || (num_edges && (!out_src || !src_local)))
    return fail(SPGNN_ERR_NULLPTR, "spgnn_block_relabel: null pointer");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(block_number_sources, dim3(scalar_grid(num_nodes)), dim3(kBlock), 0, st, flag, rank, local, extra_nodes,
                     num_nodes, num_seeds);
  if (num_edges)
    hipLaunchKernelGGL(block_relabel_edges, dim3(scalar_grid(num_edges)), dim3(kBlock), 0, st, local, out_src, src_local, num_edges);
  return check_launch("spgnn_block_relabel");
}

static int sgd_launch(float* param, const float* grad, float* momentum_buf, const float* grad_scale, const float* grad_denom,
                      const float* loss_num, float* loss_out, const float* lr_dev, int64_t n, float lr, float momentum,
                      float weight_decay, int32_t first_step, spgnn_stream_t stream, uint32_t* skipped = nullptr) {
  if (n < 0) return fail(SPGNN_ERR_SHAPE, "spgnn_sgd_momentum_step: n < 0");
  if (n == 0 && !loss_out) return SPGNN_OK;
  if (!param || !grad || !momentum_buf || (loss_out && !loss_num)) return fail(SPGNN_ERR_NULLPTR, "spgnn_sgd_momentum_step: null pointer");
  int64_t blocks = (n + kBlock - 1) / kBlock;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(sgd_momentum_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, param, grad,
                     momentum_buf, grad_scale, grad_denom, loss_num, loss_out, lr_dev, n, lr, momentum, weight_decay, first_step, skipped);
  return check_launch("spgnn_sgd_momentum_step");
}

int spgnn_sgd_momentum_step(float* param, const float* grad, float* momentum_buf, const float* grad_scale,
                            const float* lr_dev, int64_t n, float lr, float momentum, float weight_decay,
                            int32_t first_step, spgnn_stream_t stream) {
  return sgd_launch(param, grad, momentum_buf, grad_scale, nullptr, nullptr, nullptr, lr_dev, n, lr, momentum, weight_decay,
                    first_step, stream);
}

int spgnn_sgd_momentum_step_mean(float* param, const float* grad, float* momentum_buf, const float* weight_sum,
                                 const float* loss_num, float* loss_out, const float* lr_dev, int64_t n, float lr,
                                 float momentum, float weight_decay, int32_t first_step, spgnn_stream_t stream) {
  if (!weight_sum) return fail(SPGNN_ERR_NULLPTR, "spgnn_sgd_momentum_step_mean: null pointer");
  return sgd_launch(param, grad, momentum_buf, nullptr, weight_sum, loss_num, loss_out, lr_dev, n, lr, momentum, weight_decay,
                    first_step, stream);
}

int spgnn_sgd_momentum_step_guarded(float* param, const float* grad, float* momentum_buf, const float* weight_sum,
                                    const float* loss_num, float* loss_out, const float* lr_dev, uint32_t* skipped_steps, int64_t n,
                                    float lr, float momentum, float weight_decay, int32_t first_step, spgnn_stream_t stream) {
  if (!weight_sum || !loss_num || !skipped_steps) return fail(SPGNN_ERR_NULLPTR, "spgnn_sgd_momentum_step_guarded: null pointer");
  return sgd_launch(param, grad, momentum_buf, nullptr, weight_sum, loss_num, loss_out, lr_dev, n, lr, momentum, weight_decay,
                    first_step, stream, skipped_steps);
}

int spgnn_step_begin(int64_t* counter, float* scale_blocks, int32_t n_scale_blocks, uint32_t* range_violations, spgnn_stream_t stream) {
  if (n_scale_blocks < 0) return fail(SPGNN_ERR_SHAPE, "spgnn_step_begin: n_scale_blocks < 0");
  if (n_scale_blocks > 0 && !scale_blocks) return fail(SPGNN_ERR_NULLPTR, "spgnn_step_begin: null pointer");
  if (!counter && n_scale_blocks == 0) return SPGNN_OK;
  const int64_t words = (int64_t)n_scale_blocks * (spgnn_detail::kScaleHeader + spgnn_detail::kScaleSlots);
  const int64_t blocks = words > 0 ? (words + kBlock - 1) / kBlock : 1;
  hipLaunchKernelGGL(step_begin_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, counter, scale_blocks, words, range_violations);
  return check_launch("spgnn_step_begin");
}

}  // extern "C"
