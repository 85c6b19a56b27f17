// spgnn_tile.hip — tree-resident LDS tiles for the GAT traversals (gfx950 / MI355X).
//
// The batched graph is block-diagonal: every neighbour of a node lives in the node's own tree, and a tree is 100-300
// consecutive rows (reference job_runner.py:1882: dgl.batch of per-scan trees).  The row kernels of spgnn_kernels.hip
// gather neighbour rows through the CU's vector-memory path and rely on L2; for narrow rows (<= 512 bytes per node) they
// are bound by the per-node chain of 4-byte gathers (neighbour ids -> scores -> attention words -> rows), not by bytes:
// 2.4 TB/s on the 2 x 64 bf16 layers of st_gat_6 against 4.9 TB/s for a plain copy.
//
// Here a WORKGROUP owns a TILE - a run of consecutive nodes [n0, n1), normally one or a few whole trees - and a COLUMN
// SLICE of the rows (64 R columns inside one head):
//   phase A  the tile's slice of the gathered tensor (ft, or g_pre for the source-major half), its scores el / er, its
//            padded neighbour rows and its CSC offsets are streamed into LDS with coalesced 16-byte loads (no dependent
//            address anywhere: every load of the phase is in flight at once);
//   phase B  teams of 16 lanes walk the tile's nodes: softmax one (edge slot) entry per lane exactly as the row kernels
//            do, every gather - scores, attention words, neighbour rows - served from LDS; the only global accesses left
//            are the node's OWN rows (residual, gradients, results), which are streaming accesses.
// A tile need not be closed under neighbours: an id outside [n0, n1) takes a global load on a separate (rare) path, so
// any node range is correct - uniform ranges for graphs without tree boundaries, split ranges for trees larger than the
// LDS budget.  Nodes must have 1..8 edges in the direction walked and the padded (N, 8) neighbour rows are required
// (spgnn_gat_fwd's nbr8); the host checks both and keeps everything else on the row kernels.
//
// Arithmetic: identical per element to gat_fwd_vec / gat_bwd_src_vec (same operation order: bit-identical results); the
// destination-major half sums its per-edge dots over another lane geometry (fp32 rounding).  No atomics, no LDS
// reductions across teams: results are run-to-run bitwise reproducible.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

#include "spgnn_hip.h"
#include "spgnn_internal.h"
#include "spgnn_rows.h"

namespace {

using spgnn_detail::check_launch;
using spgnn_detail::fail;

constexpr int kTeam = 16;                   // lanes per node: one 64-column chunk group per R

#define SPGNN_CHECK_ARG(cond, code) do { if (!(cond)) return spgnn_detail::fail_at((code), __func__, __LINE__); } while (0)

// LDS rows: 16-byte access for both storage types (fp32: 4 elements, bf16: 4 elements = 8 bytes as in the row kernels)
__device__ __forceinline__ float4 lds_ldv(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 lds_ldv(const bf16s* p) {
  const uint2 u = *reinterpret_cast<const uint2*>(p);
  return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xFFFF0000u), __uint_as_float(u.y << 16),
                     __uint_as_float(u.y & 0xFFFF0000u));
}

// ---- phase A helpers -------------------------------------------------------------------------------------------------
// Compile-time tile geometry: at most kCap nodes per tile, kThreads threads per workgroup (64 teams: at most kIter nodes
// per team).  With both fixed every thread knows at compile time how many 16-byte pieces it can be asked to stage, so
// phase A is written as "issue every load, then store": ONE memory latency for the whole tile instead of one per loop
// trip (a loop of load -> ds_write pairs waits for each load before the next is issued).
constexpr int kCap = 192;
constexpr int kThreads = 1024;
constexpr int kTeams = kThreads / kTeam;
constexpr int kIter = (kCap + kTeams - 1) / kTeams;            // 3

// rows [n0, n0 + nt) x columns [col0, col0 + CW) of a (N, ld) tensor: this thread's pieces -> registers / -> LDS
template <typename ST, int CW> struct RowStage {
  static constexpr int EPV = 16 / (int)sizeof(ST);              // elements per 16-byte piece
  static constexpr int PPR = CW / EPV;                          // pieces per row
  static constexpr int NP = (kCap * PPR + kThreads - 1) / kThreads;
  uint4 q[NP];
  __device__ __forceinline__ void load(const ST* __restrict__ src, int64_t ld, int64_t n0, int nt, int col0) {
    const int total = nt * PPR;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const int idx = min((int)threadIdx.x + j * kThreads, total - 1);      // clamped: unconditional loads
      const int row = idx / PPR, pc = idx % PPR;
      q[j] = *reinterpret_cast<const uint4*>(src + (n0 + row) * ld + col0 + pc * EPV);
    }
  }
  __device__ __forceinline__ void store(ST* __restrict__ dst, int nt) const {
    const int total = nt * PPR;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const int idx = (int)threadIdx.x + j * kThreads;
      if (idx < total) *reinterpret_cast<uint4*>(dst + (idx / PPR) * CW + (idx % PPR) * EPV) = q[j];
    }
  }
};
// per-node scalars and the padded (nt, 8) rows: one element / one 16-byte piece per thread (kThreads >= 2 kCap + 1)
static_assert(kThreads >= 2 * kCap + 1, "one piece per thread in phase A");

__device__ __forceinline__ int align4(int x) { return (x + 3) & ~3; }

// a lane's chunk of a row as it is stored (fp32: 16 bytes, bf16: 8 bytes): what a prefetched own row is kept as
template <typename ST> struct Raw;
template <> struct Raw<float> { typedef uint4 T; };
template <> struct Raw<bf16s> { typedef uint2 T; };
__device__ __forceinline__ uint4 ld_raw(const float* p) { return *reinterpret_cast<const uint4*>(p); }
__device__ __forceinline__ uint2 ld_raw(const bf16s* p) { return *reinterpret_cast<const uint2*>(p); }
__device__ __forceinline__ float4 cvt_raw(uint4 u) {
  return make_float4(__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w));
}
__device__ __forceinline__ float4 cvt_raw(uint2 u) {
  return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xFFFF0000u), __uint_as_float(u.y << 16),
                     __uint_as_float(u.y & 0xFFFF0000u));
}

// =====================================================================================================================
// forward
// =====================================================================================================================
template <typename ST> struct TileFwd {
  const int32_t* tile_ptr; const int32_t* indptr; const int32_t* nbr8;
  const ST* ft; int64_t ft_ld; const float* el; const float* er; int64_t s_ld;
  const ST* res; int64_t res_ld; const float* bias; ST* out; int64_t out_ld; float* attn;
  int H; int D; int cap;
  float slope; int act; float p; float inv_keep; uint64_t seed; const uint64_t* seed_off;
  float fp; float finv; uint64_t fseed; int ftotal; int foff; float* absmax;
};

// grid.x = tiles, grid.y = H * D / (64 R) slices; R float4 chunks per lane, all inside one head
template <typename ST, int R>
__global__ __launch_bounds__(kThreads) void gat_fwd_tile(TileFwd<ST> a) {
  constexpr int CW = 64 * R;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if (a.seed_off) { a.seed += a.seed_off[0]; a.fseed += a.seed_off[0]; }
  const int64_t n0 = a.tile_ptr[blockIdx.x];
  const int nt = min((int)(a.tile_ptr[blockIdx.x + 1] - n0), kCap);
  if (nt <= 0) return;
  const int sph = a.D / CW;                             // slices per head
  const int h = blockIdx.y / sph, cg = blockIdx.y % sph;
  const int col0 = h * a.D + cg * CW;
  ST* rows_s = reinterpret_cast<ST*>(smem);
  float* el_s = reinterpret_cast<float*>(smem + (size_t)kCap * CW * sizeof(ST));
  float* er_s = el_s + kCap;
  int* ip_s = reinterpret_cast<int*>(er_s + kCap);
  int* nbr_s = ip_s + align4(kCap + 1);
  const int tid = threadIdx.x, lane = tid & (kTeam - 1), team = tid / kTeam;
  const int ccol = col0 + lane * 4;                     // this lane's column in chunk 0 (chunk r: + 64 r)

  // ---- every load of the tile at once: the slice of ft, scores, offsets, padded neighbour rows, the teams' own rows ----
  RowStage<ST, CW> rs;
  rs.load(a.ft, a.ft_ld, n0, nt, col0);
  const int ic = min(tid, nt - 1);
  const float el_v = a.el[(n0 + ic) * a.s_ld + h], er_v = a.er[(n0 + ic) * a.s_ld + h];
  const int ip_v = a.indptr[n0 + min(tid, nt)];
  const uint4 nb_v = reinterpret_cast<const uint4*>(a.nbr8 + n0 * 8)[min(tid, 2 * nt - 1)];
  typename Raw<ST>::T resv[kIter][R];
  if (a.res) {
#pragma unroll
    for (int it = 0; it < kIter; ++it) {
      const int64_t v = n0 + min(team + it * kTeams, nt - 1);
#pragma unroll
      for (int r = 0; r < R; ++r) resv[it][r] = ld_raw(a.res + v * a.res_ld + ccol + 64 * r);
    }
  }
  float4 bq[R];
#pragma unroll
  for (int r = 0; r < R; ++r) bq[r] = a.bias ? ld4(a.bias + ccol + 64 * r) : make_float4(0.f, 0.f, 0.f, 0.f);
  rs.store(rows_s, nt);
  if (tid < nt) { el_s[tid] = el_v; er_s[tid] = er_v; }
  if (tid <= nt) ip_s[tid] = ip_v;
  if (tid < 2 * nt) reinterpret_cast<uint4*>(nbr_s)[tid] = nb_v;
  __syncthreads();

  const int k = lane & 7;
  const int tbase = (tid & 63) & ~(kTeam - 1);
  const bool write_attn = cg == 0 && lane < 8;
#pragma unroll
  for (int it = 0; it < kIter; ++it) {
    const int i = team + it * kTeams;
    if (i >= nt) break;
    const int64_t v = n0 + i;
    const int beg = ip_s[i], deg = ip_s[i + 1] - beg;
    float4 acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = a.res ? cvt_raw(resv[it][r]) : make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.bias) {
#pragma unroll
      for (int r = 0; r < R; ++r) { acc[r].x += bq[r].x; acc[r].y += bq[r].y; acc[r].z += bq[r].z; acc[r].w += bq[r].w; }
    }
    // one (edge slot) entry per lane, lanes 8..15 repeat 0..7 (see gat_fwd_vec: same arithmetic per entry)
    const bool valid = k < deg;
    const int kk = valid ? k : max(deg - 1, 0);
    const int ue = nbr_s[i * 8 + k];
    const int ul = ue - (int)n0;
    const bool in = (unsigned)ul < (unsigned)nt;
    float elu;
    if (in) elu = el_s[ul]; else elu = a.el[(int64_t)ue * a.s_ld + h];
    float x = elu + er_s[i];
    x = valid ? lrelu(x, a.slope) : -INFINITY;
    const float mx = group8_max(x);
    const float ex = valid ? expf(x - mx) : 0.f;
    const float sm = group8_sum(ex);
    float a_ = ex / sm;
    if (write_attn && valid) a.attn[(int64_t)(beg + k) * a.H + h] = a_;
    if (a.p > 0.f) a_ *= keep_scale(a.seed, (int64_t)(beg + kk) * a.H + h, a.p, a.inv_keep);
    const float al = valid ? a_ : 0.f;

    if (__all(in)) {
#pragma unroll
      for (int k0 = 0; k0 < kMaxFast; k0 += 2) {
        if (!__any(k0 < deg)) break;
        int uk[2]; float wk[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) { uk[q] = nbr_s[i * 8 + k0 + q] - (int)n0; wk[q] = __shfl(al, tbase + k0 + q, 64); }
        float4 xr[2][R];
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int r = 0; r < R; ++r) xr[q][r] = lds_ldv(rows_s + uk[q] * CW + (r * kTeam + lane) * 4);
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int r = 0; r < R; ++r) fma4(acc[r], wk[q], xr[q][r]);
      }
    } else {                                            // a neighbour outside the tile: global rows for those (rare)
#pragma unroll 1
      for (int k0 = 0; k0 < kMaxFast; ++k0) {
        if (!__any(k0 < deg)) break;
        const int ug = nbr_s[i * 8 + k0];
        const int ulk = ug - (int)n0;
        const float wk = __shfl(al, tbase + k0, 64);
        const bool ink = (unsigned)ulk < (unsigned)nt;
#pragma unroll
        for (int r = 0; r < R; ++r) {
          float4 xr;
          if (ink) xr = lds_ldv(rows_s + ulk * CW + (r * kTeam + lane) * 4);
          else xr = ldv(a.ft + (int64_t)ug * a.ft_ld + ccol + 64 * r);
          fma4(acc[r], wk, xr);
        }
      }
    }
    act_fwd_rows<R>(acc, a.act);
    float amx = 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int c = ccol + 64 * r;
      float4 d = acc[r];
      if (a.fp > 0.f) {
        const float4 kf = feat_keep4(a.fseed, v * a.ftotal + a.foff + c, a.fp, a.finv);
        d = make_float4(d.x * kf.x, d.y * kf.y, d.z * kf.z, d.w * kf.w);
      }
      stv(a.out + v * a.out_ld + c, d);
      amx = absmax4(amx, d);
    }
    if (a.absmax) {
      amx = team_max(amx, kTeam);
      if (lane == 0) spgnn_detail::slots_max(a.absmax, amx, (unsigned)v);
    }
  }
}

// =====================================================================================================================
// backward, destination-major half: one head per slice (the per-edge dots run over the whole head), R = D / 64
// =====================================================================================================================
template <typename ST> struct TileBwdDst {
  const int32_t* tile_ptr; const int32_t* indptr; const int32_t* nbr8;
  const ST* ft; int64_t ft_ld; const float* el; const float* er; int64_t s_ld; const float* attn;
  const ST* g_out; int64_t g_out_ld; const ST* out; int64_t out_ld;
  ST* g_pre; int64_t g_pre_ld; float* g_e; float* g_er; int64_t gs_ld; float* absmax;
  int H; int D; int cap;
  float slope; int act; float p; float inv_keep; uint64_t seed; const uint64_t* seed_off;
  float fp; float finv; uint64_t fseed; int ftotal; int foff;
};

constexpr int kSlotPieces = (8 * kCap + kThreads - 1) / kThreads;     // attention words of a tile's slots per thread (2)

template <typename ST, int R>
__global__ __launch_bounds__(kThreads) void gat_bwd_dst_tile(TileBwdDst<ST> a) {
  constexpr int CW = 64 * R;                            // == D
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if (a.seed_off) { a.seed += a.seed_off[0]; a.fseed += a.seed_off[0]; }
  const int64_t n0 = a.tile_ptr[blockIdx.x];
  const int nt = min((int)(a.tile_ptr[blockIdx.x + 1] - n0), kCap);
  if (nt <= 0) return;
  const int h = blockIdx.y;
  const int col0 = h * a.D;
  ST* rows_s = reinterpret_cast<ST*>(smem);
  float* el_s = reinterpret_cast<float*>(smem + (size_t)kCap * CW * sizeof(ST));
  float* er_s = el_s + kCap;
  int* ip_s = reinterpret_cast<int*>(er_s + kCap);
  int* nbr_s = ip_s + align4(kCap + 1);
  float* at_s = reinterpret_cast<float*>(nbr_s + kCap * 8);      // attention words of the tile's slots (this head)
  const int tid = threadIdx.x, lane = tid & (kTeam - 1), team = tid / kTeam;
  const int ccol = col0 + lane * 4;

  RowStage<ST, CW> rs;
  rs.load(a.ft, a.ft_ld, n0, nt, col0);
  const int ic = min(tid, nt - 1);
  const float el_v = a.el[(n0 + ic) * a.s_ld + h], er_v = a.er[(n0 + ic) * a.s_ld + h];
  const int ip_v = a.indptr[n0 + min(tid, nt)];
  const uint4 nb_v = reinterpret_cast<const uint4*>(a.nbr8 + n0 * 8)[min(tid, 2 * nt - 1)];
  const int s0 = a.indptr[n0];
  const int ns = min(a.indptr[n0 + nt] - s0, 8 * kCap);
  float at_v[kSlotPieces];
#pragma unroll
  for (int j = 0; j < kSlotPieces; ++j) at_v[j] = a.attn[(int64_t)(s0 + min(tid + j * kThreads, ns - 1)) * a.H + h];
  typename Raw<ST>::T gv[kIter][R], ov[kIter][R];
  const ST* op = a.act != SPGNN_ACT_NONE ? a.out : a.g_out;      // no activation: `out` is not read (a second copy of g_out, unused)
  const int64_t old_ = a.act != SPGNN_ACT_NONE ? a.out_ld : a.g_out_ld;
#pragma unroll
  for (int it = 0; it < kIter; ++it) {
    const int64_t v = n0 + min(team + it * kTeams, nt - 1);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      gv[it][r] = ld_raw(a.g_out + v * a.g_out_ld + ccol + 64 * r);
      ov[it][r] = ld_raw(op + v * old_ + ccol + 64 * r);
    }
  }
  rs.store(rows_s, nt);
  if (tid < nt) { el_s[tid] = el_v; er_s[tid] = er_v; }
  if (tid <= nt) ip_s[tid] = ip_v;
  if (tid < 2 * nt) reinterpret_cast<uint4*>(nbr_s)[tid] = nb_v;
#pragma unroll
  for (int j = 0; j < kSlotPieces; ++j)
    if (tid + j * kThreads < ns) at_s[tid + j * kThreads] = at_v[j];
  __syncthreads();

  const int k = lane & 7;
#pragma unroll
  for (int it = 0; it < kIter; ++it) {
    const int i = team + it * kTeams;
    if (i >= nt) break;
    const int64_t v = n0 + i;
    const int beg = ip_s[i], deg = ip_s[i + 1] - beg;
    float4 g[R];
#pragma unroll
    for (int r = 0; r < R; ++r) g[r] = cvt_raw(gv[it][r]);
    if (a.fp > 0.f) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const float4 kf = feat_keep4(a.fseed, v * a.ftotal + a.foff + ccol + 64 * r, a.fp, a.finv);
        g[r].x *= kf.x; g[r].y *= kf.y; g[r].z *= kf.z; g[r].w *= kf.w;
      }
    }
    if (a.act != SPGNN_ACT_NONE) {
      float4 o[R];
#pragma unroll
      for (int r = 0; r < R; ++r) o[r] = cvt_raw(ov[it][r]);
      if (a.fp > 0.f) {
        const float un = 1.f - a.fp;
#pragma unroll
        for (int r = 0; r < R; ++r) { o[r].x *= un; o[r].y *= un; o[r].z *= un; o[r].w *= un; }
      }
      act_bwd_rows<R>(g, o, a.act);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) stv(a.g_pre + v * a.g_pre_ld + ccol + 64 * r, g[r]);
    if (!is_f32<ST>::value) {            // the dots must see what the source-major half and the GEMMs read: the ROUNDED g_pre
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const f32x4_t f = {g[r].x, g[r].y, g[r].z, g[r].w};
        const bf16x4_t hq = __builtin_convertvector(f, bf16x4_t);
        const f32x4_t b = __builtin_convertvector(hq, f32x4_t);
        g[r] = make_float4(b[0], b[1], b[2], b[3]);
      }
    }
    if (a.absmax) {
      float mxa = 0.f;
#pragma unroll
      for (int r = 0; r < R; ++r) mxa = absmax4(mxa, g[r]);
      mxa = team_max(mxa, kTeam);
      if (lane == 0) spgnn_detail::slots_max(a.absmax, mxa, (unsigned)v);
    }
    // per-edge dots <ft[u, h, :], g_pre[v, h, :]>: lane partials for the 8 slots
    float pd[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) pd[e] = 0.f;
    const int ue = nbr_s[i * 8 + k];
    const int ul = ue - (int)n0;
    const bool in = (unsigned)ul < (unsigned)nt;
    if (__all(in)) {
      constexpr int G = R >= 4 ? 1 : 2;                 // edges per batch of LDS reads (registers: 4 R per edge)
#pragma unroll
      for (int k0 = 0; k0 < kMaxFast; k0 += G) {
        if (!__any(k0 < deg)) break;
        float4 xr[G][R];
#pragma unroll
        for (int q = 0; q < G; ++q) {
          const int uk = nbr_s[i * 8 + k0 + q] - (int)n0;
#pragma unroll
          for (int r = 0; r < R; ++r) xr[q][r] = lds_ldv(rows_s + uk * CW + (r * kTeam + lane) * 4);
        }
#pragma unroll
        for (int q = 0; q < G; ++q)
#pragma unroll
          for (int r = 0; r < R; ++r) pd[k0 + q] += dot4(xr[q][r], g[r]);
      }
    } else {                                            // a neighbour outside the tile (rare): rolled loop, the slot's partial
#pragma unroll 1                                        // lands in its register through selects (no dynamic register index)
      for (int k0 = 0; k0 < kMaxFast; ++k0) {
        if (!__any(k0 < deg)) break;
        const int ug = nbr_s[i * 8 + k0];
        const int ulk = ug - (int)n0;
        const bool ink = (unsigned)ulk < (unsigned)nt;
        float d = 0.f;
#pragma unroll
        for (int r = 0; r < R; ++r) {
          float4 xr;
          if (ink) xr = lds_ldv(rows_s + ulk * CW + (r * kTeam + lane) * 4);
          else xr = ldv(a.ft + (int64_t)ug * a.ft_ld + ccol + 64 * r);
          d += dot4(xr, g[r]);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) pd[e] = (e == k0) ? d : pd[e];
      }
    }
    // reduce-scatter over lane bits 4, 2, 1 (entry e ends up in the lanes with (lane & 7) == e), then the two half teams
#pragma unroll
    for (int half = 4; half >= 1; half >>= 1) {
      const bool up = (lane & half) != 0;
#pragma unroll
      for (int e = 0; e < half; ++e) {
        float keep, send;
        rs_pair(up, pd[e], pd[e + half], keep, send);
        pd[e] = keep + __shfl_xor(send, half, 64);
      }
    }
    float ga = single_pass(pd[0]);
    ga = single_pass(ga + __shfl_xor(ga, 8, 64));
    const bool valid = k < deg;
    const int kk = valid ? k : max(deg - 1, 0);
    const int64_t slot = (int64_t)(beg + kk) * a.H + h;
    const int sl = beg + kk - s0;
    float al;
    if ((unsigned)sl < (unsigned)ns) al = at_s[sl]; else al = a.attn[slot];
    float elu;
    if (in) elu = el_s[ul]; else elu = a.el[(int64_t)ue * a.s_ld + h];
    const float epre = elu + er_s[i];
    al = valid ? al : 0.f;
    if (a.p > 0.f) ga *= keep_scale(a.seed, slot, a.p, a.inv_keep);
    const float S = group8_sum(valid ? al * ga : 0.f);
    float ge = al * ga - al * S;
    ge = epre > 0.f ? ge : ge * a.slope;
    ge = valid ? ge : 0.f;
    if (lane < 8 && valid) a.g_e[slot] = ge;
    const float ger = group8_sum(ge);
    if (lane == 0) a.g_er[v * a.gs_ld + h] = ger;
  }
}

// =====================================================================================================================
// backward, source-major half
// =====================================================================================================================
template <typename ST> struct TileBwdSrc {
  const int32_t* tile_ptr; const int32_t* indptr; const int32_t* out_indptr; const int32_t* out_nbr8; const int32_t* out_pos8;
  const float* attn; const float* g_e; const ST* g_pre; int64_t g_pre_ld; ST* g_ft; int64_t g_ft_ld;
  float* g_el; int64_t gs_ld; float* absmax; const float* sc_l; const float* sc_r; const float* g_er;
  int H; int D; int cap;
  float p; float inv_keep; uint64_t seed; const uint64_t* seed_off;
};

template <typename ST, int R>
__global__ __launch_bounds__(kThreads) void gat_bwd_src_tile(TileBwdSrc<ST> a) {
  constexpr int CW = 64 * R;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if (a.seed_off) a.seed += a.seed_off[0];
  const int64_t n0 = a.tile_ptr[blockIdx.x];
  const int nt = min((int)(a.tile_ptr[blockIdx.x + 1] - n0), kCap);
  if (nt <= 0) return;
  const int sph = a.D / CW;
  const int h = blockIdx.y / sph, cg = blockIdx.y % sph;
  const int col0 = h * a.D + cg * CW;
  ST* rows_s = reinterpret_cast<ST*>(smem);
  int* oip_s = reinterpret_cast<int*>(smem + (size_t)kCap * CW * sizeof(ST));
  int* onb_s = oip_s + align4(kCap + 1);
  int* ops_s = onb_s + kCap * 8;
  float* at_s = reinterpret_cast<float*>(ops_s + kCap * 8);
  float* ge_s = at_s + kCap * 8;
  const int tid = threadIdx.x, lane = tid & (kTeam - 1), team = tid / kTeam;
  const int ccol = col0 + lane * 4;

  RowStage<ST, CW> rs;
  rs.load(a.g_pre, a.g_pre_ld, n0, nt, col0);
  const int oip_v = a.out_indptr[n0 + min(tid, nt)];
  const uint4 onb_v = reinterpret_cast<const uint4*>(a.out_nbr8 + n0 * 8)[min(tid, 2 * nt - 1)];
  const uint4 ops_v = reinterpret_cast<const uint4*>(a.out_pos8 + n0 * 8)[min(tid, 2 * nt - 1)];
  const int s0 = a.indptr[n0];
  const int ns = min(a.indptr[n0 + nt] - s0, 8 * kCap);
  float at_v[kSlotPieces], ge_v[kSlotPieces];
#pragma unroll
  for (int j = 0; j < kSlotPieces; ++j) {
    const int64_t w = (int64_t)(s0 + min(tid + j * kThreads, ns - 1)) * a.H + h;
    at_v[j] = a.attn[w];
    ge_v[j] = a.g_e[w];
  }
  float gerv[kIter];
#pragma unroll
  for (int it = 0; it < kIter; ++it) gerv[it] = a.sc_l ? a.g_er[(n0 + min(team + it * kTeams, nt - 1)) * a.gs_ld + h] : 0.f;
  float4 scl[R], scr[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    scl[r] = a.sc_l ? ld4(a.sc_l + ccol + 64 * r) : make_float4(0.f, 0.f, 0.f, 0.f);
    scr[r] = a.sc_l ? ld4(a.sc_r + ccol + 64 * r) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  rs.store(rows_s, nt);
  if (tid <= nt) oip_s[tid] = oip_v;
  if (tid < 2 * nt) { reinterpret_cast<uint4*>(onb_s)[tid] = onb_v; reinterpret_cast<uint4*>(ops_s)[tid] = ops_v; }
#pragma unroll
  for (int j = 0; j < kSlotPieces; ++j)
    if (tid + j * kThreads < ns) { at_s[tid + j * kThreads] = at_v[j]; ge_s[tid + j * kThreads] = ge_v[j]; }
  __syncthreads();

  const int k = lane & 7;
  const int tbase = (tid & 63) & ~(kTeam - 1);
#pragma unroll
  for (int it = 0; it < kIter; ++it) {
    const int i = team + it * kTeams;
    if (i >= nt) break;
    const int64_t u = n0 + i;
    const int deg = oip_s[i + 1] - oip_s[i];
    const bool valid = k < deg;
    const int pe = ops_s[i * 8 + k];
    const int pl = pe - s0;
    const int64_t slot = (int64_t)pe * a.H + h;
    float x, gq;
    if ((unsigned)pl < (unsigned)ns) { x = at_s[pl]; gq = ge_s[pl]; } else { x = a.attn[slot]; gq = a.g_e[slot]; }
    if (a.p > 0.f) x *= keep_scale(a.seed, slot, a.p, a.inv_keep);
    const float wv = valid ? x : 0.f;
    const float gel = group8_sum(valid ? gq : 0.f);
    const int vg = onb_s[i * 8 + k];
    const bool in = (unsigned)(vg - (int)n0) < (unsigned)nt;

    float4 acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (__all(in)) {
#pragma unroll
      for (int k0 = 0; k0 < kMaxFast; k0 += 2) {
        if (!__any(k0 < deg)) break;
        int vk[2]; float wk[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) { vk[q] = onb_s[i * 8 + k0 + q] - (int)n0; wk[q] = __shfl(wv, tbase + k0 + q, 64); }
        float4 xr[2][R];
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int r = 0; r < R; ++r) xr[q][r] = lds_ldv(rows_s + vk[q] * CW + (r * kTeam + lane) * 4);
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int r = 0; r < R; ++r) fma4(acc[r], wk[q], xr[q][r]);
      }
    } else {
#pragma unroll 1
      for (int k0 = 0; k0 < kMaxFast; ++k0) {
        if (!__any(k0 < deg)) break;
        const int vgk = onb_s[i * 8 + k0];
        const int vlk = vgk - (int)n0;
        const float wk = __shfl(wv, tbase + k0, 64);
        const bool ink = (unsigned)vlk < (unsigned)nt;
#pragma unroll
        for (int r = 0; r < R; ++r) {
          float4 xr;
          if (ink) xr = lds_ldv(rows_s + vlk * CW + (r * kTeam + lane) * 4);
          else xr = ldv(a.g_pre + (int64_t)vgk * a.g_pre_ld + ccol + 64 * r);
          fma4(acc[r], wk, xr);
        }
      }
    }
    if (a.sc_l) {
#pragma unroll
      for (int r = 0; r < R; ++r) { fma4(acc[r], gel, scl[r]); fma4(acc[r], gerv[it], scr[r]); }
    }
    float amx = 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      stv(a.g_ft + u * a.g_ft_ld + ccol + 64 * r, acc[r]);
      amx = absmax4(amx, acc[r]);
    }
    if (cg == 0 && lane == 0) a.g_el[u * a.gs_ld + h] = gel;
    if (a.absmax) {
      amx = team_max(amx, kTeam);
      if (lane == 0) spgnn_detail::slots_max(a.absmax, amx, (unsigned)u);
    }
  }
}

// ---- host side ---------------------------------------------------------------------------------------------------------
constexpr int kLdsBudget = 160 * 1024;

// chunks per lane of the slice: as wide as fits `budget` bytes of LDS for the rows of one tile (and a head)
template <typename ST> int pick_r(int D, int cap, int budget) {
  const int cands[3] = {4, 2, 1};
  for (int r : cands)
    if (D % (64 * r) == 0 && (int64_t)cap * 64 * r * (int)sizeof(ST) <= budget) return r;
  return 0;
}
int fwd_extra(int cap) { return 4 * (2 * cap + ((cap + 4) & ~3) + 8 * cap); }
int dst_extra(int cap) { return fwd_extra(cap) + 4 * 8 * cap; }
int src_extra(int cap) { return 4 * (((cap + 4) & ~3) + 4 * 8 * cap); }
int pick_threads(int) { return kThreads; }

template <typename K, typename A>
int launch_tile(K kernel, const A& a, int64_t n_tiles, int slices, int lds, hipStream_t st, const char* name) {
  const int rc = spgnn_detail::ensure_dynamic_lds(reinterpret_cast<const void*>(kernel), lds);
  if (rc != SPGNN_OK) return rc;
  hipLaunchKernelGGL(kernel, dim3((unsigned)n_tiles, (unsigned)slices), dim3((unsigned)pick_threads(lds)), (size_t)lds, st, a);
  return check_launch(name);
}

template <typename ST>
int gat_fwd_tile_impl(const char* name, const int32_t* tile_ptr, int64_t n_tiles, int32_t cap, const int32_t* indptr, const int32_t* nbr8,
                      const ST* ft, int64_t ft_stride, const float* el, const float* er, int64_t s_stride, const ST* res,
                      int64_t res_stride, const float* bias, ST* out, int64_t out_stride, float* attn, float* absmax, int64_t N,
                      int32_t H, int32_t D, float slope, int32_t act, float p_drop, uint64_t seed, const uint64_t* seed_offset,
                      float out_drop_p, uint64_t out_drop_seed, int32_t out_drop_total, int32_t out_drop_offset, hipStream_t st) {
  SPGNN_CHECK_ARG(N >= 0 && n_tiles >= 0 && H >= 1 && D >= 64 && D % 64 == 0 && cap >= 1 && cap <= kCap, SPGNN_ERR_SHAPE);
  if (N == 0 || n_tiles == 0) return SPGNN_OK;
  SPGNN_CHECK_ARG(tile_ptr && indptr && nbr8 && ft && el && er && out && attn, SPGNN_ERR_NULLPTR);
  SPGNN_CHECK_ARG(p_drop >= 0.f && p_drop < 1.f && out_drop_p >= 0.f && out_drop_p < 1.f, SPGNN_ERR_ENUM);
  SPGNN_CHECK_ARG(act >= SPGNN_ACT_NONE && act <= SPGNN_ACT_RELU, SPGNN_ERR_ENUM);
  constexpr int al = 16 / (int)sizeof(ST) >= 8 ? 8 : 4;       // row strides in elements: 16-byte rows
  SPGNN_CHECK_ARG(ft_stride % al == 0 && out_stride % 4 == 0 && (!res || res_stride % 4 == 0), SPGNN_ERR_STRIDE);
  SPGNN_CHECK_ARG(aligned16(ft) && (reinterpret_cast<uintptr_t>(out) & 7) == 0 && (!res || (reinterpret_cast<uintptr_t>(res) & 7) == 0),
                  SPGNN_ERR_STRIDE);
  const int R = pick_r<ST>(D, kCap, 64 * 1024);
  SPGNN_CHECK_ARG(R > 0, SPGNN_ERR_SHAPE);
  TileFwd<ST> a{tile_ptr, indptr, nbr8, ft, ft_stride, el, er, s_stride, res, res_stride, bias, out, out_stride, attn, H, D, cap,
                slope, act, p_drop, 1.f / (1.f - p_drop), seed, seed_offset, out_drop_p, 1.f / (1.f - out_drop_p), out_drop_seed,
                out_drop_total, out_drop_offset, absmax};
  const int lds = kCap * 64 * R * (int)sizeof(ST) + fwd_extra(kCap);
  SPGNN_CHECK_ARG(lds <= kLdsBudget, SPGNN_ERR_SHAPE);
  const int slices = H * D / (64 * R);
  if (R == 4) return launch_tile(gat_fwd_tile<ST, 4>, a, n_tiles, slices, lds, st, name);
  if (R == 2) return launch_tile(gat_fwd_tile<ST, 2>, a, n_tiles, slices, lds, st, name);
  return launch_tile(gat_fwd_tile<ST, 1>, a, n_tiles, slices, lds, st, name);
}

template <typename ST>
int gat_bwd_dst_tile_impl(const char* name, const int32_t* tile_ptr, int64_t n_tiles, int32_t cap, const int32_t* indptr,
                          const int32_t* nbr8, const ST* ft, int64_t ft_stride, const float* el, const float* er, int64_t s_stride,
                          const float* attn, const ST* g_out, int64_t g_out_stride, const ST* out, int64_t out_stride, ST* g_pre,
                          int64_t g_pre_stride, float* g_e, float* g_er, int64_t g_s_stride, float* absmax, int64_t N, int32_t H,
                          int32_t D, float slope, int32_t act, float p_drop, uint64_t seed, const uint64_t* seed_offset,
                          float out_drop_p, uint64_t out_drop_seed, int32_t out_drop_total, int32_t out_drop_offset, hipStream_t st) {
  SPGNN_CHECK_ARG(N >= 0 && n_tiles >= 0 && H >= 1 && (D == 64 || D == 128 || D == 256) && cap >= 1 && cap <= kCap, SPGNN_ERR_SHAPE);
  if (N == 0 || n_tiles == 0) return SPGNN_OK;
  SPGNN_CHECK_ARG(tile_ptr && indptr && nbr8 && ft && el && er && attn && g_out && g_pre && g_e && g_er, SPGNN_ERR_NULLPTR);
  SPGNN_CHECK_ARG(act == SPGNN_ACT_NONE || out, SPGNN_ERR_NULLPTR);
  SPGNN_CHECK_ARG(p_drop >= 0.f && p_drop < 1.f && out_drop_p >= 0.f && out_drop_p < 1.f, SPGNN_ERR_ENUM);
  SPGNN_CHECK_ARG(act >= SPGNN_ACT_NONE && act <= SPGNN_ACT_RELU, SPGNN_ERR_ENUM);
  constexpr int al = 16 / (int)sizeof(ST) >= 8 ? 8 : 4;
  SPGNN_CHECK_ARG(ft_stride % al == 0 && g_out_stride % 4 == 0 && g_pre_stride % 4 == 0 && (!out || out_stride % 4 == 0), SPGNN_ERR_STRIDE);
  SPGNN_CHECK_ARG(aligned16(ft) && (reinterpret_cast<uintptr_t>(g_out) & 7) == 0 && (reinterpret_cast<uintptr_t>(g_pre) & 7) == 0 &&
                  (!out || (reinterpret_cast<uintptr_t>(out) & 7) == 0), SPGNN_ERR_STRIDE);
  const int R = D / 64;
  TileBwdDst<ST> a{tile_ptr, indptr, nbr8, ft, ft_stride, el, er, s_stride, attn, g_out, g_out_stride, out, out_stride, g_pre,
                   g_pre_stride, g_e, g_er, g_s_stride, absmax, H, D, cap, slope, act, p_drop, 1.f / (1.f - p_drop), seed,
                   seed_offset, out_drop_p, 1.f / (1.f - out_drop_p), out_drop_seed, out_drop_total, out_drop_offset};
  const int lds = kCap * D * (int)sizeof(ST) + dst_extra(kCap);
  SPGNN_CHECK_ARG(lds <= kLdsBudget, SPGNN_ERR_SHAPE);
  if (R == 4) return launch_tile(gat_bwd_dst_tile<ST, 4>, a, n_tiles, H, lds, st, name);
  if (R == 2) return launch_tile(gat_bwd_dst_tile<ST, 2>, a, n_tiles, H, lds, st, name);
  return launch_tile(gat_bwd_dst_tile<ST, 1>, a, n_tiles, H, lds, st, name);
}

template <typename ST>
int gat_bwd_src_tile_impl(const char* name, const int32_t* tile_ptr, int64_t n_tiles, int32_t cap, const int32_t* indptr,
                          const int32_t* out_indptr, const int32_t* out_nbr8, const int32_t* out_pos8, const float* attn,
                          const float* g_e, const ST* g_pre, int64_t g_pre_stride, ST* g_ft, int64_t g_ft_stride, float* g_el,
                          int64_t g_s_stride, float* absmax, const float* score_l, const float* score_r, const float* g_er,
                          int64_t N, int32_t H, int32_t D, float p_drop, uint64_t seed, const uint64_t* seed_offset, hipStream_t st) {
  SPGNN_CHECK_ARG(N >= 0 && n_tiles >= 0 && H >= 1 && D >= 64 && D % 64 == 0 && cap >= 1 && cap <= kCap, SPGNN_ERR_SHAPE);
  if (N == 0 || n_tiles == 0) return SPGNN_OK;
  SPGNN_CHECK_ARG(tile_ptr && indptr && out_indptr && out_nbr8 && out_pos8 && attn && g_e && g_pre && g_ft && g_el, SPGNN_ERR_NULLPTR);
  SPGNN_CHECK_ARG((score_l == nullptr) == (score_r == nullptr) && (!score_l || g_er), SPGNN_ERR_NULLPTR);
  SPGNN_CHECK_ARG(p_drop >= 0.f && p_drop < 1.f, SPGNN_ERR_ENUM);
  constexpr int al = 16 / (int)sizeof(ST) >= 8 ? 8 : 4;
  SPGNN_CHECK_ARG(g_pre_stride % al == 0 && g_ft_stride % 4 == 0, SPGNN_ERR_STRIDE);
  SPGNN_CHECK_ARG(aligned16(g_pre) && (reinterpret_cast<uintptr_t>(g_ft) & 7) == 0 && (!score_l || (aligned16(score_l) && aligned16(score_r))),
                  SPGNN_ERR_STRIDE);
  const int R = pick_r<ST>(D, kCap, 64 * 1024);
  SPGNN_CHECK_ARG(R > 0, SPGNN_ERR_SHAPE);
  TileBwdSrc<ST> a{tile_ptr, indptr, out_indptr, out_nbr8, out_pos8, attn, g_e, g_pre, g_pre_stride, g_ft, g_ft_stride, g_el,
                   g_s_stride, absmax, score_l, score_r, g_er, H, D, cap, p_drop, 1.f / (1.f - p_drop), seed, seed_offset};
  const int lds = kCap * 64 * R * (int)sizeof(ST) + src_extra(kCap);
  SPGNN_CHECK_ARG(lds <= kLdsBudget, SPGNN_ERR_SHAPE);
  const int slices = H * D / (64 * R);
  if (R == 4) return launch_tile(gat_bwd_src_tile<ST, 4>, a, n_tiles, slices, lds, st, name);
  if (R == 2) return launch_tile(gat_bwd_src_tile<ST, 2>, a, n_tiles, slices, lds, st, name);
  return launch_tile(gat_bwd_src_tile<ST, 1>, a, n_tiles, slices, lds, st, name);
}

}  // namespace

extern "C" {

int spgnn_gat_tile_supported(int32_t H, int32_t D, int32_t elem_bytes, int32_t max_tile_nodes) {
  if (H < 1 || max_tile_nodes < 1 || max_tile_nodes > kCap || (elem_bytes != 2 && elem_bytes != 4)) return 0;
  if (D != 64 && D != 128 && D != 256) return 0;                       // the destination-major half holds a whole head
  return (int64_t)kCap * D * elem_bytes + dst_extra(kCap) <= kLdsBudget ? 1 : 0;
}

#define SPGNN_TILE_FWD_ARGS(ST)                                                                                          \
    const int32_t* tile_ptr, int64_t n_tiles, int32_t max_tile_nodes, const int32_t* indptr, const int32_t* nbr8, const ST* ft, \
    int64_t ft_stride, const float* el, const float* er, int64_t s_stride, const ST* res, int64_t res_stride, const float* bias, \
    ST* out, int64_t out_stride, float* attn, float* absmax, int64_t N, int32_t H, int32_t D, float negative_slope,       \
    int32_t activation, float p_drop, uint64_t seed, const uint64_t* seed_offset, float out_drop_p, uint64_t out_drop_seed, \
    int32_t out_drop_total, int32_t out_drop_offset, spgnn_stream_t stream
#define SPGNN_TILE_FWD_PASS                                                                                               \
    tile_ptr, n_tiles, max_tile_nodes, indptr, nbr8, ft, ft_stride, el, er, s_stride, res, res_stride, bias, out, out_stride, attn, \
    absmax, N, H, D, negative_slope, activation, p_drop, seed, seed_offset, out_drop_p, out_drop_seed, out_drop_total,    \
    out_drop_offset, (hipStream_t)stream

int spgnn_gat_fwd_tile(SPGNN_TILE_FWD_ARGS(float)) { return gat_fwd_tile_impl<float>("spgnn_gat_fwd_tile", SPGNN_TILE_FWD_PASS); }
int spgnn_gat_fwd_tile_bf16(const int32_t* tile_ptr, int64_t n_tiles, int32_t max_tile_nodes, const int32_t* indptr, const int32_t* nbr8,
                            const uint16_t* ft_, int64_t ft_stride, const float* el, const float* er, int64_t s_stride,
                            const uint16_t* res_, int64_t res_stride, const float* bias, uint16_t* out_, int64_t out_stride, float* attn,
                            float* absmax, int64_t N, int32_t H, int32_t D, float negative_slope, int32_t activation, float p_drop,
                            uint64_t seed, const uint64_t* seed_offset, float out_drop_p, uint64_t out_drop_seed, int32_t out_drop_total,
                            int32_t out_drop_offset, spgnn_stream_t stream) {
  const bf16s* ft = reinterpret_cast<const bf16s*>(ft_); const bf16s* res = reinterpret_cast<const bf16s*>(res_);
  bf16s* out = reinterpret_cast<bf16s*>(out_);
  return gat_fwd_tile_impl<bf16s>("spgnn_gat_fwd_tile_bf16", SPGNN_TILE_FWD_PASS);
}

#define SPGNN_TILE_DST_PASS                                                                                               \
    tile_ptr, n_tiles, max_tile_nodes, indptr, nbr8, ft, ft_stride, el, er, s_stride, attn, g_out, g_out_stride, out, out_stride, \
    g_pre, g_pre_stride, g_e, g_er, g_s_stride, absmax, N, H, D, negative_slope, activation, p_drop, seed, seed_offset,   \
    out_drop_p, out_drop_seed, out_drop_total, out_drop_offset, (hipStream_t)stream

int spgnn_gat_bwd_dst_tile(const int32_t* tile_ptr, int64_t n_tiles, int32_t max_tile_nodes, const int32_t* indptr, const int32_t* nbr8,
                           const float* ft, int64_t ft_stride, const float* el, const float* er, int64_t s_stride, const float* attn,
                           const float* g_out, int64_t g_out_stride, const float* out, int64_t out_stride, float* g_pre,
                           int64_t g_pre_stride, float* g_e, float* g_er, int64_t g_s_stride, float* absmax, int64_t N, int32_t H,
                           int32_t D, float negative_slope, int32_t activation, float p_drop, uint64_t seed, const uint64_t* seed_offset,
                           float out_drop_p, uint64_t out_drop_seed, int32_t out_drop_total, int32_t out_drop_offset,
                           spgnn_stream_t stream) {
  return gat_bwd_dst_tile_impl<float>("spgnn_gat_bwd_dst_tile", SPGNN_TILE_DST_PASS);
}
int spgnn_gat_bwd_dst_tile_bf16(const int32_t* tile_ptr, int64_t n_tiles, int32_t max_tile_nodes, const int32_t* indptr,
                                const int32_t* nbr8, const uint16_t* ft_, int64_t ft_stride, const float* el, const float* er,
                                int64_t s_stride, const float* attn, const uint16_t* g_out_, int64_t g_out_stride, const uint16_t* out_,
                                int64_t out_stride, uint16_t* g_pre_, int64_t g_pre_stride, float* g_e, float* g_er, int64_t g_s_stride,
                                float* absmax, int64_t N, int32_t H, int32_t D, float negative_slope, int32_t activation, float p_drop,
                                uint64_t seed, const uint64_t* seed_offset, float out_drop_p, uint64_t out_drop_seed,
                                int32_t out_drop_total, int32_t out_drop_offset, spgnn_stream_t stream) {
  const bf16s* ft = reinterpret_cast<const bf16s*>(ft_); const bf16s* g_out = reinterpret_cast<const bf16s*>(g_out_);
  const bf16s* out = reinterpret_cast<const bf16s*>(out_); bf16s* g_pre = reinterpret_cast<bf16s*>(g_pre_);
  return gat_bwd_dst_tile_impl<bf16s>("spgnn_gat_bwd_dst_tile_bf16", SPGNN_TILE_DST_PASS);
}

#define SPGNN_TILE_SRC_PASS                                                                                               \
    tile_ptr, n_tiles, max_tile_nodes, indptr, out_indptr, out_nbr8, out_pos8, attn, g_e, g_pre, g_pre_stride, g_ft, g_ft_stride, \
    g_el, g_s_stride, absmax, score_l, score_r, g_er, N, H, D, p_drop, seed, seed_offset, (hipStream_t)stream

int spgnn_gat_bwd_src_tile(const int32_t* tile_ptr, int64_t n_tiles, int32_t max_tile_nodes, const int32_t* indptr,
                           const int32_t* out_indptr, const int32_t* out_nbr8, const int32_t* out_pos8, const float* attn,
                           const float* g_e, const float* g_pre, int64_t g_pre_stride, float* g_ft, int64_t g_ft_stride, float* g_el,
                           int64_t g_s_stride, float* absmax, const float* score_l, const float* score_r, const float* g_er, int64_t N,
                           int32_t H, int32_t D, float p_drop, uint64_t seed, const uint64_t* seed_offset, spgnn_stream_t stream) {
  return gat_bwd_src_tile_impl<float>("spgnn_gat_bwd_src_tile", SPGNN_TILE_SRC_PASS);
}
int spgnn_gat_bwd_src_tile_bf16(const int32_t* tile_ptr, int64_t n_tiles, int32_t max_tile_nodes, const int32_t* indptr,
                                const int32_t* out_indptr, const int32_t* out_nbr8, const int32_t* out_pos8, const float* attn,
                                const float* g_e, const uint16_t* g_pre_, int64_t g_pre_stride, uint16_t* g_ft_, int64_t g_ft_stride,
                                float* g_el, int64_t g_s_stride, float* absmax, const float* score_l, const float* score_r,
                                const float* g_er, int64_t N, int32_t H, int32_t D, float p_drop, uint64_t seed,
                                const uint64_t* seed_offset, spgnn_stream_t stream) {
  const bf16s* g_pre = reinterpret_cast<const bf16s*>(g_pre_); bf16s* g_ft = reinterpret_cast<bf16s*>(g_ft_);
  return gat_bwd_src_tile_impl<bf16s>("spgnn_gat_bwd_src_tile_bf16", SPGNN_TILE_SRC_PASS);
}

}  // extern "C"
