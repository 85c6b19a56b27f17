// spgnn_lspe.hip — one traversal per SPGNN level: the structure GATConv (two heads) and the position GATConv (one head,
// the learnable positional-encoding update, LSPE) of reference models.py:472-484 walk the batched CSC together.
//
//   reference, per level l:   h_s = GATConv_s(g, dropout(cat[h_s, h_p])).flatten(1)        (models.py:477-479)
//                             h_p = GATConv_p(g, dropout(h_p)).flatten(1)                  (models.py:441-456, 476-479)
//
// In every config of the reference the position layer has ONE head as wide as a structure head (pos_hiddens ==
// num_hiddens, exp_settings/st_pgat_spgnn_3.py:86-115), so the pair is one GAT traversal over THREE heads of D columns whose
// rows come from two projected tensors: heads 0, 1 from Y_s = [ft_s | res_s], head 2 from Y_p = [ft_p | res_p].  A team of
// T = D / 4 lanes owns a node; lane l holds float4 chunk l of each head.  Everything that differs between the two layers
// (activation ELU / tanh, attention dropout rate and seed, score vectors, bias) is per GROUP (0 = structure, 1 = position);
// the head -> group map is a compile-time property of the chunk index, so the row phase has no selects at all.
//
//   forward   one (edge slot, head) entry per lane: leaky-relu, segment max / sum over the 8 lanes of a head, exp, division,
//             attention store, dropout hash - once per entry - then the 3 x 8 weights are broadcast and the neighbour rows of
//             all three heads are gathered in one pass.  The 3 D-wide result is written straight into the NEXT level's
//             structure input (N, 3D) under that layer's feature-dropout mask (the reference's cat + GATConv.feat_drop), and
//             the position head once more, under the next POSITION layer's mask, as that layer's input (N, D).
//   bwd_dst   g_pre of all heads (the gradient of the position rows is the sum of the two consumers' masked gradients), the
//             per-edge dots summed by a reduce-scatter over a 32-entry table (24 used), softmax / LeakyReLU backward.
//   bwd_src   g_ft of all heads over the out-edges, g_el, the score term g_el attn_l + g_er attn_r.
//
// Nodes are assumed to have 1 <= degree <= 8 in both directions (the host checks it: airway trees have degree <= 5 plus the
// self loop) and the padded (N, 8) neighbour rows are required; anything else takes the two-launch path of spgnn_kernels.hip.
// The forward is bit-identical to that path; the backward sums its per-edge dots over a different team geometry, so it
// agrees to fp32 rounding.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

#include "spgnn_hip.h"
#include "spgnn_internal.h"
#include "spgnn_rows.h"

namespace {

using spgnn_detail::check_launch;
using spgnn_detail::fail;

constexpr int kHeads0 = 2;                 // structure heads (group 0); group 1 = the one position head
constexpr int kNS = 3;                     // heads of a level
constexpr int kEnt = 32;                   // entry table: (edge slot, head) = 8 x 4, the fourth head is padding

// value `e` of a per-team table held one entry per lane (register e / T of lane e % T)
template <int T, int NREG>
__device__ __forceinline__ float table_get(const float (&tab)[NREG], int e, int tbase) {
  if constexpr (T == 64) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(tab[0]), e));
  } else {
    float v0 = __shfl(tab[0], tbase + (e & (T - 1)), 64);
    if constexpr (NREG > 1) { const float v1 = __shfl(tab[NREG - 1], tbase + (e & (T - 1)), 64); v0 = (e / T) ? v1 : v0; }
    return v0;
  }
}

template <int N> __device__ __forceinline__ void act_fwd_n(float4* o, int act) {
#define SPGNN_ROWS(EXPR) _Pragma("unroll") for (int r = 0; r < N; ++r) { \
    { float x = o[r].x; o[r].x = (EXPR); } { float x = o[r].y; o[r].y = (EXPR); } \
    { float x = o[r].z; o[r].z = (EXPR); } { float x = o[r].w; o[r].w = (EXPR); } }
  if (act == SPGNN_ACT_ELU) { SPGNN_ROWS(elu_fwd(x)) }
  else if (act == SPGNN_ACT_TANH) { SPGNN_ROWS(tanhf(x)) }
  else if (act == SPGNN_ACT_RELU) { SPGNN_ROWS(x > 0.f ? x : 0.f) }
#undef SPGNN_ROWS
}
template <int N> __device__ __forceinline__ void act_bwd_n(float4* g, const float4* o, int act) {
#define SPGNN_ROWS(EXPR) _Pragma("unroll") for (int r = 0; r < N; ++r) { \
    { float y = o[r].x; g[r].x *= (EXPR); } { float y = o[r].y; g[r].y *= (EXPR); } \
    { float y = o[r].z; g[r].z *= (EXPR); } { float y = o[r].w; g[r].w *= (EXPR); } }
  if (act == SPGNN_ACT_ELU) { SPGNN_ROWS(y > 0.f ? 1.f : y + 1.f) }
  else if (act == SPGNN_ACT_TANH) { SPGNN_ROWS(1.f - y * y) }
  else if (act == SPGNN_ACT_RELU) { SPGNN_ROWS(y > 0.f ? 1.f : 0.f) }
#undef SPGNN_ROWS
}
__device__ __forceinline__ float4 mul4(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ float4 scale4(float4 a, float s) { return make_float4(a.x * s, a.y * s, a.z * s, a.w * s); }

// -------------------------------------------------------------------------------------------------
// forward
// -------------------------------------------------------------------------------------------------
struct GrpF {
  const float* ft; int64_t ft_ld; const float* res; int64_t res_ld; const float* bias;
  float* el; float* er; int64_t s_ld; float* attn;
  const float* parts;                    // PARTS kernels: the projection GEMM's score partials (N, H*D/64, 2); el / er are then OUTPUTS
  int H; int act; float slope; float p; float inv_keep; uint64_t seed;
};
struct LspeFwd {
  const int32_t* indptr; const int32_t* nbr8;
  GrpF g[2];
  float* out; int64_t out_ld; float fp; float finv; uint64_t fseed; int ftotal;
  float* out2; int64_t out2_ld; float fp2; float finv2; uint64_t fseed2;
  float* absmax; float* absmax2;
  int64_t N; int D; const uint64_t* seed_off;
};

// T lanes per node, D = 4 T: chunk r of a lane = head r (heads 0, 1: group 0; head 2: group 1).
// PARTS: el / er are summed here from the projection GEMMs' per-64-column score partials (D / 64 = T / 16 of them per head, in
// ascending order: the arithmetic of spgnn_scores_from_parts, which this replaces) and the node's own pair is written out for
// the backward kernels.
template <int T, bool PARTS>
__global__ __launch_bounds__(kBlock) void lspe_fwd_kernel(LspeFwd a) {
  constexpr bool WAVE = T == 64;
  constexpr int NREG = T >= 32 ? 1 : 2;
  uint64_t seed_g[2] = {a.g[0].seed, a.g[1].seed};
  if (a.seed_off) { const uint64_t o = a.seed_off[0]; seed_g[0] += o; seed_g[1] += o; a.fseed += o; a.fseed2 += o; }
  const int lane = threadIdx.x % T;
  const int64_t v = xcd_block() * (kBlock / T) + uni<WAVE>((int)(threadIdx.x / T));
  if (v >= a.N) return;
  const int beg = uni<WAVE>(a.indptr[v]), end = uni<WAVE>(a.indptr[v + 1]), deg = end - beg;
  const int D = a.D, c0 = lane * 4;                  // column of this lane's chunk inside a head

  // the accumulators start from the residual rows (+ bias): loads that depend on nothing (see gat_fwd_vec)
  float4 acc[kNS];
#pragma unroll
  for (int r = 0; r < kNS; ++r) {
    // (no branch around a load: hipcc drains the memory queue at every such join; a missing bias reads the residual row
    // instead and is scaled by zero - fma(1, b, r) is r + b exactly)
    const GrpF& g = a.g[r < kHeads0 ? 0 : 1];
    const int cg = (r < kHeads0 ? r * D : 0) + c0;
    acc[r] = ld4(g.res + v * g.res_ld + cg);
    const float4 q = ld4(g.bias ? g.bias + cg : g.res + v * g.res_ld + cg);
    const float bsc = g.bias ? 1.f : 0.f;
    fma4(acc[r], bsc, q);
  }
  int u[kMaxFast];
#pragma unroll
  for (int k = 0; k < kMaxFast; ++k) u[k] = uni<WAVE>(a.nbr8[v * 8 + k]);

  // one (edge slot, head) entry per lane: e = head * 8 + slot
  const int wl = threadIdx.x & 63, tbase = wl & ~(T - 1);
  float al[NREG];
#pragma unroll
  for (int i = 0; i < NREG; ++i) {
    const int e = lane + i * T;
    const int k = e & 7, s_ = e >> 3;
    const bool own = s_ < kNS, valid = own && k < deg;
    const int kk = k < deg ? k : deg - 1;
    const bool g1 = own && s_ >= kHeads0;
    const int hl = (own && !g1) ? s_ : 0;
    float* elp = g1 ? a.g[1].el : a.g[0].el;
    float* erp = g1 ? a.g[1].er : a.g[0].er;
    const int64_t sld = g1 ? a.g[1].s_ld : a.g[0].s_ld;
    const int Hg = g1 ? a.g[1].H : a.g[0].H;
    const float slope = g1 ? a.g[1].slope : a.g[0].slope, p = g1 ? a.g[1].p : a.g[0].p, ik = g1 ? a.g[1].inv_keep : a.g[0].inv_keep;
    float* attn = g1 ? a.g[1].attn : a.g[0].attn;
    const uint64_t sd = g1 ? seed_g[1] : seed_g[0];
    const int ue = a.nbr8[v * 8 + k];
    float x;
    if constexpr (PARTS) {
      constexpr int NB = T / 16;                       // 64-column blocks per head
      const float2* pp = reinterpret_cast<const float2*>(g1 ? a.g[1].parts : a.g[0].parts);
      const float2* pu = pp + ((int64_t)ue * Hg + hl) * NB;
      const float2* pv = pp + (v * Hg + hl) * NB;
      float2 qu[NB], qv[NB];
#pragma unroll
      for (int b = 0; b < NB; ++b) { qu[b] = pu[b]; qv[b] = pv[b]; }
      float elu = 0.f, elv = 0.f, erv = 0.f;
#pragma unroll
      for (int b = 0; b < NB; ++b) { elu += qu[b].x; elv += qv[b].x; erv += qv[b].y; }
      x = elu + erv;
      if (own && k == 0) { elp[v * sld + hl] = elv; erp[v * sld + hl] = erv; }
    } else {
      x = elp[(int64_t)ue * sld + hl] + erp[v * sld + hl];
    }
    x = valid ? lrelu(x, slope) : -INFINITY;
    const float mx = group8_max(x);
    const float ex = valid ? expf(x - mx) : 0.f;
    const float sm = group8_sum(ex);
    float a_ = ex / sm;
    if (valid) attn[(int64_t)(beg + k) * Hg + hl] = a_;
    if (p > 0.f) a_ *= keep_scale(sd, (int64_t)(beg + kk) * Hg + hl, p, ik);
    al[i] = valid ? a_ : 0.f;
  }
  float w[kMaxFast][kNS];
#pragma unroll
  for (int s = 0; s < kNS; ++s)
#pragma unroll
    for (int k = 0; k < kMaxFast; ++k) w[k][s] = table_get<T, NREG>(al, s * 8 + k, tbase);

  constexpr int kGather = 2;                          // edges per batch: 6 float4 in flight per lane (1, 4, 8 measured: no better)
#pragma unroll
  for (int k0 = 0; k0 < kMaxFast; k0 += kGather) {
    if (WAVE ? !(k0 < deg) : !__any(k0 < deg)) break;
    float4 x[kGather][kNS];
#pragma unroll
    for (int q = 0; q < kGather; ++q)
#pragma unroll
      for (int r = 0; r < kNS; ++r) {
        const GrpF& g = a.g[r < kHeads0 ? 0 : 1];
        x[q][r] = ld4(g.ft + (int64_t)u[k0 + q] * g.ft_ld + (r < kHeads0 ? r * D : 0) + c0);
      }
#pragma unroll
    for (int q = 0; q < kGather; ++q)
#pragma unroll
      for (int r = 0; r < kNS; ++r) fma4(acc[r], w[k0 + q][r], x[q][r]);
  }
  act_fwd_n<kHeads0>(acc, a.g[0].act);
  act_fwd_n<kNS - kHeads0>(acc + kHeads0, a.g[1].act);

  float amx = 0.f, amx2 = 0.f;
#pragma unroll
  for (int r = 0; r < kNS; ++r) {                      // all heads -> the next structure input, under ITS feature dropout
    const int c = r * D + c0;
    float4 d = acc[r];
    if (a.fp > 0.f) d = mul4(d, feat_keep4(a.fseed, v * a.ftotal + c, a.fp, a.finv));
    st4(a.out + v * a.out_ld + c, d);
    amx = absmax4(amx, d);
  }
  if (a.out2) {                                        // the position head once more: the next position layer's input
#pragma unroll
    for (int r = kHeads0; r < kNS; ++r) {
      const int c = (r - kHeads0) * D + c0;
      float4 d = acc[r];
      if (a.fp2 > 0.f) d = mul4(d, feat_keep4(a.fseed2, v * (int64_t)((kNS - kHeads0) * D) + c, a.fp2, a.finv2));
      st4(a.out2 + v * a.out2_ld + c, d);
      amx2 = absmax4(amx2, d);
    }
  }
  if (a.absmax) { amx = team_max(amx, T); if (lane == 0) spgnn_detail::slots_max(a.absmax, amx, (unsigned)v); }
  if (a.absmax2) { amx2 = team_max(amx2, T); if (lane == 0) spgnn_detail::slots_max(a.absmax2, amx2, (unsigned)v); }
}

// -------------------------------------------------------------------------------------------------
// backward, dst-major half
// -------------------------------------------------------------------------------------------------
struct GrpD {
  const float* ft; int64_t ft_ld; const float* el; const float* er; int64_t s_ld; const float* attn;
  float* g_pre; int64_t g_pre_ld; float* g_e; float* g_er; int64_t gs_ld; float* absmax;
  int H; int act; float slope; float p; float inv_keep; uint64_t seed;
};
struct LspeBwdDst {
  const int32_t* indptr; const int32_t* nbr8;
  GrpD g[2];
  const float* g_out; int64_t g_out_ld;      // gradient of the (N, 3D) buffer the forward wrote
  const float* g_out2; int64_t g_out2_ld;    // gradient of the position head's second copy (N, D), or null
  const float* out; int64_t out_ld; float fp; float finv; uint64_t fseed; int ftotal;
  const float* out2; int64_t out2_ld; float fp2; float finv2; uint64_t fseed2;
  int64_t N; int D; const uint64_t* seed_off;
};

template <int T>
__global__ __launch_bounds__(kBlock) void lspe_bwd_dst_kernel(LspeBwdDst a) {
  constexpr bool WAVE = T == 64;
  constexpr int NREG = T >= 32 ? 1 : 2;
  uint64_t seed_g[2] = {a.g[0].seed, a.g[1].seed};
  if (a.seed_off) { const uint64_t o = a.seed_off[0]; seed_g[0] += o; seed_g[1] += o; a.fseed += o; a.fseed2 += o; }
  const int lane = threadIdx.x % T;
  const int64_t v = xcd_block() * (kBlock / T) + uni<WAVE>((int)(threadIdx.x / T));
  if (v >= a.N) return;
  const int beg = uni<WAVE>(a.indptr[v]), end = uni<WAVE>(a.indptr[v + 1]), deg = end - beg;
  const int D = a.D, c0 = lane * 4;

  // g = dL/d(pre-activation rows).  The buffer's gradient under the buffer's mask; the position head also takes the gradient
  // of its second copy under that copy's mask (the two consumers of h_p: models.py:477-481 and 476).
  float4 g[kNS], o[kNS], kf1[kNS];
#pragma unroll
  for (int r = 0; r < kNS; ++r) {
    const int c = r * D + c0;
    g[r] = ld4(a.g_out + v * a.g_out_ld + c);
    o[r] = ld4(a.out + v * a.out_ld + c);
    kf1[r] = make_float4(1.f, 1.f, 1.f, 1.f);
  }
  if (a.fp > 0.f) {
    const float un = 1.f - a.fp;
#pragma unroll
    for (int r = 0; r < kNS; ++r) {
      kf1[r] = feat_keep4(a.fseed, v * a.ftotal + r * D + c0, a.fp, a.finv);
      g[r] = mul4(g[r], kf1[r]);
      o[r] = scale4(o[r], un);                         // kept elements: out = stored / inv_keep (dropped ones: g is 0 there)
    }
  }
  {
    // (unconditional loads: a missing g_out2 reads out2 once more and is scaled by zero; see the forward)
    const float* g2p = a.g_out2 ? a.g_out2 : a.out2;
    const int64_t g2ld = a.g_out2 ? a.g_out2_ld : a.out2_ld;
    const float g2s = a.g_out2 ? 1.f : 0.f;
#pragma unroll
    for (int r = kHeads0; r < kNS; ++r) {
      const int c = (r - kHeads0) * D + c0;
      float4 o2 = ld4(a.out2 + v * a.out2_ld + c);
      const float4 q2 = ld4(g2p + v * g2ld + c);
      float4 kf2 = make_float4(1.f, 1.f, 1.f, 1.f);
      if (a.fp2 > 0.f) {
        kf2 = feat_keep4(a.fseed2, v * (int64_t)((kNS - kHeads0) * D) + c, a.fp2, a.finv2);
        o2 = scale4(o2, 1.f - a.fp2);
      }
      fma4(g[r], g2s, mul4(q2, kf2));
      // the activation's output: exact from the second copy when that one is stored plain, else from whichever copy kept it
      if (a.fp2 > 0.f) {
        o[r].x = kf1[r].x != 0.f ? o[r].x : o2.x; o[r].y = kf1[r].y != 0.f ? o[r].y : o2.y;
        o[r].z = kf1[r].z != 0.f ? o[r].z : o2.z; o[r].w = kf1[r].w != 0.f ? o[r].w : o2.w;
      } else {
        o[r] = o2;
      }
    }
  }
  act_bwd_n<kHeads0>(g, o, a.g[0].act);
  act_bwd_n<kNS - kHeads0>(g + kHeads0, o + kHeads0, a.g[1].act);
#pragma unroll
  for (int r = 0; r < kNS; ++r) {
    const GrpD& gr = a.g[r < kHeads0 ? 0 : 1];
    st4(gr.g_pre + v * gr.g_pre_ld + (r < kHeads0 ? r * D : 0) + c0, g[r]);
  }
  if (a.g[0].absmax) {
    float mx = 0.f;
#pragma unroll
    for (int r = 0; r < kHeads0; ++r) mx = absmax4(mx, g[r]);
    mx = team_max(mx, T);
    if (lane == 0) spgnn_detail::slots_max(a.g[0].absmax, mx, (unsigned)v);
  }
  if (a.g[1].absmax) {
    float mx = 0.f;
#pragma unroll
    for (int r = kHeads0; r < kNS; ++r) mx = absmax4(mx, g[r]);
    mx = team_max(mx, T);
    if (lane == 0) spgnn_detail::slots_max(a.g[1].absmax, mx, (unsigned)v);
  }

  int u[kMaxFast];
#pragma unroll
  for (int k = 0; k < kMaxFast; ++k) u[k] = uni<WAVE>(a.nbr8[v * 8 + k]);
  // per-edge dots <ft[u, head], g_pre[v, head]>: per-lane partials for the 32-entry table (entry = head * 8 + slot)
  float pd[kEnt];
#pragma unroll
  for (int e = 0; e < kEnt; ++e) pd[e] = 0.f;
  constexpr int kGather = 2;
#pragma unroll
  for (int k0 = 0; k0 < kMaxFast; k0 += kGather) {
    if (WAVE ? !(k0 < deg) : !__any(k0 < deg)) break;
    float4 x[kGather][kNS];
#pragma unroll
    for (int q = 0; q < kGather; ++q)
#pragma unroll
      for (int r = 0; r < kNS; ++r) {
        const GrpD& gr = a.g[r < kHeads0 ? 0 : 1];
        x[q][r] = ld4(gr.ft + (int64_t)u[k0 + q] * gr.ft_ld + (r < kHeads0 ? r * D : 0) + c0);
      }
#pragma unroll
    for (int q = 0; q < kGather; ++q)
#pragma unroll
      for (int r = 0; r < kNS; ++r) pd[r * 8 + k0 + q] += dot4(x[q][r], g[r]);
  }
  // reduce-scatter over the team: each round halves the values a lane carries and leaves entry e complete in lane e
  // (T = 16: two tables of 16 entries, registers 0 and 1; T = 64: the two half-waves hold copies, folded at the end)
  float ga[NREG];
  if constexpr (T >= 32) {
#pragma unroll
    for (int half = kEnt / 2; half >= 1; half >>= 1) {
      const bool up = (lane & half) != 0;
#pragma unroll
      for (int i = 0; i < half; ++i) {
        float keep, send;
        rs_pair(up, pd[i], pd[i + half], keep, send);
        pd[i] = keep + __shfl_xor(send, half, 64);
      }
    }
    ga[0] = single_pass(pd[0]);
    if constexpr (T == 64) ga[0] = single_pass(ga[0] + __shfl_xor(ga[0], 32, 64));
  } else {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int half = 8; half >= 1; half >>= 1) {
        const bool up = (lane & half) != 0;
#pragma unroll
        for (int i = 0; i < half; ++i) {
          float keep, send;
          rs_pair(up, pd[t * 16 + i], pd[t * 16 + i + half], keep, send);
          pd[t * 16 + i] = keep + __shfl_xor(send, half, 64);
        }
      }
      ga[t] = single_pass(pd[t * 16]);
    }
  }
#pragma unroll
  for (int i = 0; i < NREG; ++i) {
    const int e = T >= 32 ? (lane & (kEnt - 1)) : lane + 16 * i;
    const int k = e & 7, s_ = e >> 3;
    const bool own = s_ < kNS, valid = own && k < deg;
    const int kk = k < deg ? k : deg - 1;
    const bool g1 = own && s_ >= kHeads0;
    const int hl = (own && !g1) ? s_ : 0;
    const GrpD& gr = a.g[0];                          // (selected per lane below)
    const float* elp = g1 ? a.g[1].el : gr.el;
    const float* erp = g1 ? a.g[1].er : gr.er;
    const int64_t sld = g1 ? a.g[1].s_ld : gr.s_ld;
    const int Hg = g1 ? a.g[1].H : gr.H;
    const float slope = g1 ? a.g[1].slope : gr.slope, p = g1 ? a.g[1].p : gr.p, ik = g1 ? a.g[1].inv_keep : gr.inv_keep;
    const float* attn = g1 ? a.g[1].attn : gr.attn;
    float* g_e = g1 ? a.g[1].g_e : gr.g_e;
    float* g_er = g1 ? a.g[1].g_er : gr.g_er;
    const int64_t gsld = g1 ? a.g[1].gs_ld : gr.gs_ld;
    const uint64_t sd = g1 ? seed_g[1] : seed_g[0];
    const int64_t slot = (int64_t)(beg + kk) * Hg + hl;
    float al = attn[slot];
    const int ue = a.nbr8[v * 8 + k];
    const float epre = elp[(int64_t)ue * sld + hl] + erp[v * sld + hl];
    al = valid ? al : 0.f;
    float gav = ga[i];
    if (p > 0.f) gav *= keep_scale(sd, slot, p, ik);
    const float S = group8_sum(valid ? al * gav : 0.f);
    float ge = al * gav - al * S;
    ge = epre > 0.f ? ge : ge * slope;
    ge = valid ? ge : 0.f;
    const bool writer = T == 64 ? lane < kEnt : true;  // T = 64: lanes 32-63 hold identical copies of the table
    if (writer && valid) g_e[slot] = ge;
    const float ger = group8_sum(ge);
    if (writer && own && k == 0) g_er[v * gsld + hl] = ger;
  }
}

// -------------------------------------------------------------------------------------------------
// backward, src-major half
// -------------------------------------------------------------------------------------------------
struct GrpS {
  const float* attn; const float* g_e; const float* g_pre; int64_t g_pre_ld; float* g_ft; int64_t g_ft_ld;
  float* g_el; const float* g_er; int64_t gs_ld; const float* sc_l; const float* sc_r; float* absmax;
  int H; float p; float inv_keep; uint64_t seed;
};
struct LspeBwdSrc {
  const int32_t* out_indptr; const int32_t* out_nbr8; const int32_t* out_pos8;
  GrpS g[2];
  int64_t N; int D; const uint64_t* seed_off;
};

// T lanes per node, RR float4 chunks per lane and head: D = 4 T RR.  RR = 1 is the geometry of the other two kernels (T = D / 4);
// RR > 1 puts MORE NODES IN A WAVE (D = 256 on 32- or 16-lane teams, D = 128 on 16-lane teams): this kernel is the one bound by
// its per-node chain - out-list row -> 24 scattered attention / g_e words by CSC slot -> neighbour rows - not by bytes (3.4-3.8
// TB/s against 4.9-5.1 for the dst-major kernels), and a wave hides more of that chain with two or four of them in flight
// (the step the SpMM kernels took in round 3).  Same arithmetic in the same order per element: results do not depend on RR.
template <int T, int RR>
__global__ __launch_bounds__(kBlock) void lspe_bwd_src_kernel(LspeBwdSrc a) {
  constexpr bool WAVE = T == 64;
  constexpr int NREG = T >= 32 ? 1 : 2;
  uint64_t seed_g[2] = {a.g[0].seed, a.g[1].seed};
  if (a.seed_off) { const uint64_t o = a.seed_off[0]; seed_g[0] += o; seed_g[1] += o; }
  const int lane = threadIdx.x % T;
  const int64_t u = xcd_block() * (kBlock / T) + uni<WAVE>((int)(threadIdx.x / T));
  if (u >= a.N) return;
  const int beg = uni<WAVE>(a.out_indptr[u]), end = uni<WAVE>(a.out_indptr[u + 1]), deg = end - beg;
  const int D = a.D;
  int vv[kMaxFast];
#pragma unroll
  for (int k = 0; k < kMaxFast; ++k) vv[k] = uni<WAVE>(a.out_nbr8[u * 8 + k]);

  const int wl = threadIdx.x & 63, tbase = wl & ~(T - 1);
  float wv[NREG], gsum[NREG];
#pragma unroll
  for (int i = 0; i < NREG; ++i) {
    const int e = lane + i * T;
    const int k = e & 7, s_ = e >> 3;
    const bool own = s_ < kNS, valid = own && k < deg;
    const bool g1 = own && s_ >= kHeads0;
    const int hl = (own && !g1) ? s_ : 0;
    const int Hg = g1 ? a.g[1].H : a.g[0].H;
    const float* attn = g1 ? a.g[1].attn : a.g[0].attn;
    const float* g_e = g1 ? a.g[1].g_e : a.g[0].g_e;
    const float p = g1 ? a.g[1].p : a.g[0].p, ik = g1 ? a.g[1].inv_keep : a.g[0].inv_keep;
    const uint64_t sd = g1 ? seed_g[1] : seed_g[0];
    const int pe = a.out_pos8[u * 8 + k];
    const int64_t slot = (int64_t)pe * Hg + hl;
    float x = attn[slot];
    const float gq = g_e[slot];
    if (p > 0.f) x *= keep_scale(sd, slot, p, ik);
    wv[i] = valid ? x : 0.f;
    gsum[i] = group8_sum(valid ? gq : 0.f);
  }
  float w[kMaxFast][kNS], gel[kNS];
#pragma unroll
  for (int s = 0; s < kNS; ++s) {
    gel[s] = table_get<T, NREG>(gsum, s * 8, tbase);    // entry (slot 0, head s): its lane holds the head's g_el sum
#pragma unroll
    for (int k = 0; k < kMaxFast; ++k) w[k][s] = table_get<T, NREG>(wv, s * 8 + k, tbase);
  }
  float4 acc[kNS][RR];
#pragma unroll
  for (int r = 0; r < kNS; ++r)
#pragma unroll
    for (int c = 0; c < RR; ++c) acc[r][c] = make_float4(0.f, 0.f, 0.f, 0.f);
  constexpr int kGather = RR >= 4 ? 1 : 2;             // neighbour rows requested together: 6-12 float4 in flight per lane
#pragma unroll
  for (int k0 = 0; k0 < kMaxFast; k0 += kGather) {
    if (WAVE ? !(k0 < deg) : !__any(k0 < deg)) break;
    float4 x[kGather][kNS][RR];
#pragma unroll
    for (int q = 0; q < kGather; ++q)
#pragma unroll
      for (int r = 0; r < kNS; ++r) {
        const GrpS& gr = a.g[r < kHeads0 ? 0 : 1];
        const float* row = gr.g_pre + (int64_t)vv[k0 + q] * gr.g_pre_ld + (r < kHeads0 ? r * D : 0);
#pragma unroll
        for (int c = 0; c < RR; ++c) x[q][r][c] = ld4(row + (c * T + lane) * 4);
      }
#pragma unroll
    for (int q = 0; q < kGather; ++q)
#pragma unroll
      for (int r = 0; r < kNS; ++r)
#pragma unroll
        for (int c = 0; c < RR; ++c) fma4(acc[r][c], w[k0 + q][r], x[q][r][c]);
  }
  // score term (el / er are taken FROM ft, DGL's own form): g_ft[u, h, :] += g_el[u, h] attn_l[h, :] + g_er[u, h] attn_r[h, :]
#pragma unroll
  for (int r = 0; r < kNS; ++r) {
    const GrpS& gr = a.g[r < kHeads0 ? 0 : 1];
    const int hl = r < kHeads0 ? r : r - kHeads0;
    const float ger = gr.g_er[u * gr.gs_ld + hl];
#pragma unroll
    for (int c = 0; c < RR; ++c) {
      const int cg = hl * D + (c * T + lane) * 4;
      fma4(acc[r][c], gel[r], ld4(gr.sc_l + cg));
      fma4(acc[r][c], ger, ld4(gr.sc_r + cg));
      st4(gr.g_ft + u * gr.g_ft_ld + cg, acc[r][c]);
    }
    if (lane == 0) gr.g_el[u * gr.gs_ld + hl] = gel[r];
  }
  if (a.g[0].absmax) {
    float mx = 0.f;
#pragma unroll
    for (int r = 0; r < kHeads0; ++r)
#pragma unroll
      for (int c = 0; c < RR; ++c) mx = absmax4(mx, acc[r][c]);
    mx = team_max(mx, T);
    if (lane == 0) spgnn_detail::slots_max(a.g[0].absmax, mx, (unsigned)u);
  }
  if (a.g[1].absmax) {
    float mx = 0.f;
#pragma unroll
    for (int r = kHeads0; r < kNS; ++r)
#pragma unroll
      for (int c = 0; c < RR; ++c) mx = absmax4(mx, acc[r][c]);
    mx = team_max(mx, T);
    if (lane == 0) spgnn_detail::slots_max(a.g[1].absmax, mx, (unsigned)u);
  }
}

// src-major geometry: 16-lane teams for every level width (four nodes per wave), RR = D / 64 chunks per lane and head.  Measured as
// three libraries in one process (tools/step_ab.py; T = D / 4 as the other two kernels | D = 256 on 32 lanes, 128 on 16 | both on 16):
// 5.164 | 5.173 | 5.144 ms per step at 512 trees, 1.078 | 1.077 | 1.071 at 64.
int launch_lspe_bwd_src(const LspeBwdSrc& a, int D, hipStream_t st) {
  const dim3 grid(grid_for(a.N, kBlock / 16)), block(kBlock);
  if (D == 256) hipLaunchKernelGGL((lspe_bwd_src_kernel<16, 4>), grid, block, 0, st, a);
  else if (D == 128) hipLaunchKernelGGL((lspe_bwd_src_kernel<16, 2>), grid, block, 0, st, a);
  else hipLaunchKernelGGL((lspe_bwd_src_kernel<16, 1>), grid, block, 0, st, a);
  return check_launch("spgnn_lspe_bwd_src");
}

bool lspe_geometry(int32_t D, int& T) {
  if (D == 256) { T = 64; return true; }
  if (D == 128) { T = 32; return true; }
  if (D == 64) { T = 16; return true; }
  return false;
}

#define LSPE_CHECK(cond, code) do { if (!(cond)) return spgnn_detail::fail_at(code, __func__, __LINE__); } while (0)

int check_rows(const void* p, int64_t ld, int64_t width) {
  if (!p) return SPGNN_ERR_NULLPTR;
  if (ld < width || (ld & 3) || (reinterpret_cast<uintptr_t>(p) & 15)) return SPGNN_ERR_STRIDE;
  return SPGNN_OK;
}

template <class ARGS, class K16, class K32, class K64>
int launch_lspe(const ARGS& a, int T, hipStream_t st, K16 k16, K32 k32, K64 k64, const char* what) {
  const dim3 grid(grid_for(a.N, kBlock / T)), block(kBlock);
  if (T == 64) hipLaunchKernelGGL(k64, grid, block, 0, st, a);
  else if (T == 32) hipLaunchKernelGGL(k32, grid, block, 0, st, a);
  else hipLaunchKernelGGL(k16, grid, block, 0, st, a);
  return check_launch(what);
}

}  // namespace

extern "C" {

int spgnn_lspe_supported(int32_t D) {
  int T;
  return lspe_geometry(D, T) ? 1 : 0;
}

int spgnn_lspe_fwd(const int32_t* indptr, const int32_t* nbr8, const spgnn_lspe_fwd_group* groups, float* out, int64_t out_stride,
                   float out_drop_p, uint64_t out_drop_seed, float* out2, int64_t out2_stride, float out2_drop_p,
                   uint64_t out2_drop_seed, float* out_absmax, float* out2_absmax, int64_t N, int64_t E, int32_t D,
                   const uint64_t* seed_offset, spgnn_stream_t stream) {
  int T;
  LSPE_CHECK(N >= 0 && E >= 0 && lspe_geometry(D, T), SPGNN_ERR_SHAPE);
  if (N == 0) return SPGNN_OK;
  LSPE_CHECK(indptr && nbr8 && groups && out, SPGNN_ERR_NULLPTR);
  LSPE_CHECK(groups[0].H == kHeads0 && groups[1].H == kNS - kHeads0, SPGNN_ERR_SHAPE);
  LSPE_CHECK(out_drop_p >= 0.f && out_drop_p < 1.f && out2_drop_p >= 0.f && out2_drop_p < 1.f, SPGNN_ERR_ENUM);
  int rc;
  if ((rc = check_rows(out, out_stride, (int64_t)kNS * D)) != SPGNN_OK) return spgnn_detail::fail_at(rc, __func__, __LINE__);
  if (out2 && (rc = check_rows(out2, out2_stride, D)) != SPGNN_OK) return spgnn_detail::fail_at(rc, __func__, __LINE__);
  LspeFwd a{};
  a.indptr = indptr; a.nbr8 = nbr8;
  const bool parts = groups[0].score_parts != nullptr;
  for (int i = 0; i < 2; ++i) {
    const spgnn_lspe_fwd_group& s = groups[i];
    const int64_t w = (int64_t)s.H * D;
    if ((rc = check_rows(s.ft, s.ft_stride, w)) != SPGNN_OK) return spgnn_detail::fail_at(rc, __func__, __LINE__);
    if ((rc = check_rows(s.res, s.res_stride, w)) != SPGNN_OK) return spgnn_detail::fail_at(rc, __func__, __LINE__);
    LSPE_CHECK(s.el && s.er && s.attn && s.s_stride >= s.H, SPGNN_ERR_NULLPTR);
    LSPE_CHECK(!s.bias || !(reinterpret_cast<uintptr_t>(s.bias) & 15), SPGNN_ERR_STRIDE);
    LSPE_CHECK(s.act >= SPGNN_ACT_NONE && s.act <= SPGNN_ACT_RELU && s.p_drop >= 0.f && s.p_drop < 1.f, SPGNN_ERR_ENUM);
    LSPE_CHECK((s.score_parts != nullptr) == parts && !(reinterpret_cast<uintptr_t>(s.score_parts) & 7), SPGNN_ERR_NULLPTR);
    a.g[i] = GrpF{s.ft, s.ft_stride, s.res, s.res_stride, s.bias, s.el, s.er, s.s_stride, s.attn, s.score_parts, s.H, s.act, s.slope,
                  s.p_drop, 1.f / (1.f - s.p_drop), s.seed};
  }
  a.out = out; a.out_ld = out_stride; a.fp = out_drop_p; a.finv = 1.f / (1.f - out_drop_p); a.fseed = out_drop_seed;
  a.ftotal = kNS * D;
  a.out2 = out2; a.out2_ld = out2_stride; a.fp2 = out2_drop_p; a.finv2 = 1.f / (1.f - out2_drop_p); a.fseed2 = out2_drop_seed;
  a.absmax = out_absmax; a.absmax2 = out2 ? out2_absmax : nullptr;
  a.N = N; a.D = D; a.seed_off = seed_offset;
  if (parts)
    return launch_lspe(a, T, (hipStream_t)stream, lspe_fwd_kernel<16, true>, lspe_fwd_kernel<32, true>, lspe_fwd_kernel<64, true>, "spgnn_lspe_fwd");
  return launch_lspe(a, T, (hipStream_t)stream, lspe_fwd_kernel<16, false>, lspe_fwd_kernel<32, false>, lspe_fwd_kernel<64, false>, "spgnn_lspe_fwd");
}

int spgnn_lspe_bwd_dst(const int32_t* indptr, const int32_t* nbr8, const spgnn_lspe_bwd_dst_group* groups, const float* g_out,
                       int64_t g_out_stride, const float* g_out2, int64_t g_out2_stride, const float* out, int64_t out_stride,
                       float out_drop_p, uint64_t out_drop_seed, const float* out2, int64_t out2_stride, float out2_drop_p,
                       uint64_t out2_drop_seed, int64_t N, int64_t E, int32_t D, const uint64_t* seed_offset,
                       spgnn_stream_t stream) {
  int T;
  LSPE_CHECK(N >= 0 && E >= 0 && lspe_geometry(D, T), SPGNN_ERR_SHAPE);
  if (N == 0) return SPGNN_OK;
  LSPE_CHECK(indptr && nbr8 && groups && g_out && out, SPGNN_ERR_NULLPTR);
  LSPE_CHECK(groups[0].H == kHeads0 && groups[1].H == kNS - kHeads0, SPGNN_ERR_SHAPE);
  LSPE_CHECK(out_drop_p >= 0.f && out_drop_p < 1.f && out2_drop_p >= 0.f && out2_drop_p < 1.f, SPGNN_ERR_ENUM);
  int rc;
  if ((rc = check_rows(g_out, g_out_stride, (int64_t)kNS * D)) != SPGNN_OK) return spgnn_detail::fail_at(rc, __func__, __LINE__);
  if ((rc = check_rows(out, out_stride, (int64_t)kNS * D)) != SPGNN_OK) return spgnn_detail::fail_at(rc, __func__, __LINE__);
  if (g_out2 && (rc = check_rows(g_out2, g_out2_stride, D)) != SPGNN_OK) return spgnn_detail::fail_at(rc, __func__, __LINE__);
  if ((rc = check_rows(out2, out2_stride, D)) != SPGNN_OK) return spgnn_detail::fail_at(rc, __func__, __LINE__);
  LspeBwdDst a{};
  a.indptr = indptr; a.nbr8 = nbr8;
  for (int i = 0; i < 2; ++i) {
    const spgnn_lspe_bwd_dst_group& s = groups[i];
    const int64_t w = (int64_t)s.H * D;
    if ((rc = check_rows(s.ft, s.ft_stride, w)) != SPGNN_OK) return spgnn_detail::fail_at(rc, __func__, __LINE__);
    if ((rc = check_rows(s.g_pre, s.g_pre_stride, w)) != SPGNN_OK) return spgnn_detail::fail_at(rc, __func__, __LINE__);
    LSPE_CHECK(s.el && s.er && s.attn && s.g_e && s.g_er && s.s_stride >= s.H && s.gs_stride >= s.H, SPGNN_ERR_NULLPTR);
    LSPE_CHECK(s.act >= SPGNN_ACT_NONE && s.act <= SPGNN_ACT_RELU && s.p_drop >= 0.f && s.p_drop < 1.f, SPGNN_ERR_ENUM);
    a.g[i] = GrpD{s.ft, s.ft_stride, s.el, s.er, s.s_stride, s.attn, s.g_pre, s.g_pre_stride, s.g_e, s.g_er, s.gs_stride, s.absmax,
                  s.H, s.act, s.slope, s.p_drop, 1.f / (1.f - s.p_drop), s.seed};
  }
  a.g_out = g_out; a.g_out_ld = g_out_stride; a.g_out2 = g_out2; a.g_out2_ld = g_out2_stride;
  a.out = out; a.out_ld = out_stride; a.fp = out_drop_p; a.finv = 1.f / (1.f - out_drop_p); a.fseed = out_drop_seed; a.ftotal = kNS * D;
  a.out2 = out2; a.out2_ld = out2_stride; a.fp2 = out2_drop_p; a.finv2 = 1.f / (1.f - out2_drop_p); a.fseed2 = out2_drop_seed;
  a.N = N; a.D = D; a.seed_off = seed_offset;
  return launch_lspe(a, T, (hipStream_t)stream, lspe_bwd_dst_kernel<16>, lspe_bwd_dst_kernel<32>, lspe_bwd_dst_kernel<64>,
                     "spgnn_lspe_bwd_dst");
}

int spgnn_lspe_bwd_src(const int32_t* out_indptr, const int32_t* out_nbr8, const int32_t* out_pos8,
                       const spgnn_lspe_bwd_src_group* groups, int64_t N, int64_t E, int32_t D, const uint64_t* seed_offset,
                       spgnn_stream_t stream) {
  int T;
  LSPE_CHECK(N >= 0 && E >= 0 && lspe_geometry(D, T), SPGNN_ERR_SHAPE);
  if (N == 0) return SPGNN_OK;
  LSPE_CHECK(out_indptr && out_nbr8 && out_pos8 && groups, SPGNN_ERR_NULLPTR);
  LSPE_CHECK(groups[0].H == kHeads0 && groups[1].H == kNS - kHeads0, SPGNN_ERR_SHAPE);
  int rc;
  LspeBwdSrc a{};
  a.out_indptr = out_indptr; a.out_nbr8 = out_nbr8; a.out_pos8 = out_pos8;
  for (int i = 0; i < 2; ++i) {
    const spgnn_lspe_bwd_src_group& s = groups[i];
    const int64_t w = (int64_t)s.H * D;
    if ((rc = check_rows(s.g_pre, s.g_pre_stride, w)) != SPGNN_OK) return spgnn_detail::fail_at(rc, __func__, __LINE__);
    if ((rc = check_rows(s.g_ft, s.g_ft_stride, w)) != SPGNN_OK) return spgnn_detail::fail_at(rc, __func__, __LINE__);
    LSPE_CHECK(s.attn && s.g_e && s.g_el && s.gs_stride >= s.H && s.p_drop >= 0.f && s.p_drop < 1.f, SPGNN_ERR_NULLPTR);
    LSPE_CHECK(s.score_l && s.score_r && s.g_er, SPGNN_ERR_NULLPTR);
    LSPE_CHECK(!((reinterpret_cast<uintptr_t>(s.score_l) | reinterpret_cast<uintptr_t>(s.score_r)) & 15), SPGNN_ERR_STRIDE);
    a.g[i] = GrpS{s.attn, s.g_e, s.g_pre, s.g_pre_stride, s.g_ft, s.g_ft_stride, s.g_el, s.g_er, s.gs_stride, s.score_l, s.score_r,
                  s.absmax, s.H, s.p_drop, 1.f / (1.f - s.p_drop), s.seed};
  }
  a.N = N; a.D = D; a.seed_off = seed_offset;
  return launch_lspe_bwd_src(a, D, (hipStream_t)stream);
}

}  // extern "C"
