// spgnn_graph.hip — batch assembly on the device: from the per-tree adjacency matrices of a loader batch to the batched
// edge list, CSC and CSR the message-passing kernels walk (SURVEY.md section 8a-G / 8b `spgnn_build_csc`).
//
// Reference rule (job_runner.py:1319-1344 and 1779-1801, batching by dgl.batch at 1390 / 1882), per tree with adjacency
// `adj` (n x n uint8, dataset.py:418-419): `DGLGraph(nx.DiGraph(adj))` enumerates the non-zero entries row by row, i.e. the
// directed pairs (u, v) sorted by (u, v); `dgl.remove_self_loop` drops the diagonal keeping that order; then
// `g.add_edges(g.nodes(), g.nodes())` appends the n self loops (i, i); `dgl.batch` offsets node ids by the running node count
// and concatenates the trees' edge lists.  Edge ids are therefore, for tree t with first node f_t and nnz_t off-diagonal
// non-zeros:   id(u, v) = [sum over earlier trees of (nnz + n)] + (off-diagonal non-zeros before (u, v) in row-major order),
//              id(i, i) = [the same base] + nnz_t + i.
// DGL's COO -> CSC conversion is stable in the edge id, so the in-list of v is its off-diagonal column in ascending u followed
// by the self loop, and the out-list of u its off-diagonal row in ascending v followed by the self loop.
//
// Two kernels, one workgroup per tree: `count` (off-diagonal non-zeros per row and per column) and, after two prefix sums
// over the node arrays, `fill` (edge list, both index structures and the CSR -> CSC slot map out_pos).  The adjacency bytes
// of a tree (<= 32 KB for 180 branches) are read from L1 / L2; nothing is staged.  All integer work: results are bit-exact
// against oracle/graph_rule_nx.py (tests/test_data.py).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "spgnn_hip.h"
#include "spgnn_internal.h"

namespace {

using spgnn_detail::check_launch;
using spgnn_detail::fail;

constexpr int kThreads = 256;

__global__ __launch_bounds__(kThreads) void build_csc_count_kernel(const uint8_t* __restrict__ adj, const int64_t* __restrict__ adj_ptr,
                                                                  const int64_t* __restrict__ tree_ptr, int32_t* __restrict__ row_count,
                                                                  int32_t* __restrict__ col_count) {
  const int64_t t = blockIdx.x;
  const int64_t first = tree_ptr[t];
  const int n = (int)(tree_ptr[t + 1] - first);
  const uint8_t* a = adj + adj_ptr[t];
  for (int i = threadIdx.x; i < n; i += kThreads) {
    int r = 0, c = 0;
    for (int j = 0; j < n; ++j) {
      r += (j != i && a[(int64_t)i * n + j] != 0) ? 1 : 0;        // row i: out-edges of i
      c += (j != i && a[(int64_t)j * n + i] != 0) ? 1 : 0;        // column i: in-edges of i (coalesced across threads)
    }
    row_count[first + i] = r;
    col_count[first + i] = c;
  }
}

__global__ __launch_bounds__(kThreads) void build_csc_fill_kernel(const uint8_t* __restrict__ adj, const int64_t* __restrict__ adj_ptr,
                                                                 const int64_t* __restrict__ tree_ptr, int64_t num_trees,
                                                                 const int64_t* __restrict__ row_start, const int64_t* __restrict__ col_start,
                                                                 int32_t* __restrict__ src, int32_t* __restrict__ dst,
                                                                 int32_t* __restrict__ indptr, int32_t* __restrict__ indices,
                                                                 int32_t* __restrict__ eid, int32_t* __restrict__ out_indptr,
                                                                 int32_t* __restrict__ out_indices, int32_t* __restrict__ out_pos,
                                                                 int64_t N, int64_t E) {
  const int64_t t = blockIdx.x;
  const int64_t first = tree_ptr[t];
  const int n = (int)(tree_ptr[t + 1] - first);
  const uint8_t* a = adj + adj_ptr[t];
  // ids of this tree's edges start at row_start[first] (off-diagonal edges of earlier trees) + first (their self loops)
  const int64_t self_base = row_start[first + n] + first;         // id of (0, 0): after the tree's off-diagonal edges
  // phase A, one thread per row u: edge list entries and the out-list (ascending v, self loop last)
  for (int u = threadIdx.x; u < n; u += kThreads) {
    const int64_t ug = first + u;
    const int64_t id0 = row_start[ug] + first;                    // id of the first off-diagonal edge of row u
    const int64_t o0 = row_start[ug] + ug;                        // out_indptr[ug]
    const int64_t i0 = col_start[ug] + ug;                        // indptr[ug]
    out_indptr[ug] = (int32_t)o0;
    indptr[ug] = (int32_t)i0;
    int r = 0;
    for (int v = 0; v < n; ++v) {
      if (v != u && a[(int64_t)u * n + v] != 0) {
        src[id0 + r] = (int32_t)ug;
        dst[id0 + r] = (int32_t)(first + v);
        out_indices[o0 + r] = (int32_t)(first + v);
        ++r;
      }
    }
    const int64_t sid = self_base + u;
    src[sid] = (int32_t)ug; dst[sid] = (int32_t)ug;
    out_indices[o0 + r] = (int32_t)ug;
    const int cin = (int)(col_start[ug + 1] - col_start[ug]);
    out_pos[o0 + r] = (int32_t)(i0 + cin);                        // the self loop is the last in-edge of u
    indices[i0 + cin] = (int32_t)ug;
    eid[i0 + cin] = (int32_t)sid;
  }
  if (t == num_trees - 1 && threadIdx.x == 0) { indptr[N] = (int32_t)E; out_indptr[N] = (int32_t)E; }
  __threadfence_block();
  __syncthreads();                                                // phase B reads the out-lists written above
  // phase B, one thread per column v: the in-list (ascending u), its edge ids, and the CSR -> CSC slot map
  for (int v = threadIdx.x; v < n; v += kThreads) {
    const int64_t vg = first + v;
    const int64_t i0 = col_start[vg] + vg;
    int c = 0;
    for (int u = 0; u < n; ++u) {
      if (u != v && a[(int64_t)u * n + v] != 0) {
        const int64_t ug = first + u;
        const int64_t o0 = row_start[ug] + ug;
        const int cnt = (int)(row_start[ug + 1] - row_start[ug]);
        int r = 0;                                                // rank of v in row u: its place in u's sorted out-list
        while (r < cnt && out_indices[o0 + r] != (int32_t)vg) ++r;
        indices[i0 + c] = (int32_t)ug;
        eid[i0 + c] = (int32_t)(row_start[ug] + first + r);
        out_pos[o0 + r] = (int32_t)(i0 + c);
        ++c;
      }
    }
  }
}

// The padded neighbour rows of DeviceCSC.ell (spgnn_amd/graph.py): out[v, k] = arr[min(ptr[v] + min(k, max(deg(v) - 1, 0)), E - 1)],
// k < 8, for up to two arrays over the same slot order (indices; or out_indices and out_pos).  One thread per (node, k).
__global__ __launch_bounds__(256) void ell_rows_kernel(const int32_t* __restrict__ ptr, const int32_t* __restrict__ a0,
                                                       const int32_t* __restrict__ a1, int64_t N, int64_t E,
                                                       int32_t* __restrict__ o0, int32_t* __restrict__ o1) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= N * 8) return;
  const int64_t v = i >> 3;
  const int k = (int)(i & 7);
  const int beg = ptr[v], deg = ptr[v + 1] - beg;
  int64_t pos = (int64_t)beg + (k < deg - 1 ? k : (deg > 0 ? deg - 1 : 0));
  pos = pos < E - 1 ? pos : E - 1;
  o0[i] = a0[pos];
  if (a1) o1[i] = a1[pos];
}

// Several int32 arrays filled in one launch: dst[i] = src[i] for i < n, then dst[n + i] = pad[i] + pad_add for i < n_pad (a batch
// arena's index arrays: the loaded batch's array, then the pad component's shifted by the batch's node / edge count).
__global__ __launch_bounds__(256) void copy_pad_i32_kernel(spgnn_copy_pad_jobs jobs) {
  const spgnn_copy_pad_job& j = jobs.job[blockIdx.y];
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (int64_t)j.n + j.n_pad; i += (int64_t)gridDim.x * 256)
    j.dst[i] = i < j.n ? j.src[i] : j.pad[i - j.n] + j.pad_add;
}

// An arena load in ONE launch (ABI 62): the index arrays of spgnn_copy_pad_i32 (blockIdx.y < n_i32) and 2-D row copies of
// 4-byte words (the rest): dst[r, col : col + width] = src[r, 0 : width] for r < rows_copy, zeros for rows_copy <= r < rows_total
// - node data into the arena's buffers (pad rows a larger earlier batch left behind back to zero) and the derived tensors
// (cat[fvs, pos_enc], 16-byte-row copies) straight from the incoming batch.
__global__ __launch_bounds__(256) void arena_load_kernel(spgnn_copy_pad_jobs ij, spgnn_row_copy_jobs rj) {
  if ((int)blockIdx.y < ij.n_jobs) {
    const spgnn_copy_pad_job& j = ij.job[blockIdx.y];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (int64_t)j.n + j.n_pad; i += (int64_t)gridDim.x * 256)
      j.dst[i] = i < j.n ? j.src[i] : j.pad[i - j.n] + j.pad_add;
    return;
  }
  const spgnn_row_copy_job& j = rj.job[blockIdx.y - ij.n_jobs];
  const int64_t w = j.width, total = j.rows_total * w;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / w, c = i - r * w;
    j.dst[r * j.dst_stride + j.dst_col + c] = r < j.rows_copy ? j.src[r * j.src_stride + c] : 0u;
  }
}

// Both directions' padded neighbour rows in one launch: blockIdx.y = 0: nbr8 from (indptr, indices); 1: out_nbr8 / out_pos8.
__global__ __launch_bounds__(256) void ell_rows_both_kernel(const int32_t* __restrict__ indptr, const int32_t* __restrict__ indices,
                                                            const int32_t* __restrict__ out_indptr, const int32_t* __restrict__ out_indices,
                                                            const int32_t* __restrict__ out_pos, int64_t N, int64_t E,
                                                            int32_t* __restrict__ nbr8, int32_t* __restrict__ out_nbr8,
                                                            int32_t* __restrict__ out_pos8) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= N * 8) return;
  const bool outd = blockIdx.y == 1;
  const int32_t* ptr = outd ? out_indptr : indptr;
  const int64_t v = i >> 3;
  const int k = (int)(i & 7);
  const int beg = ptr[v], deg = ptr[v + 1] - beg;
  int64_t pos = (int64_t)beg + (k < deg - 1 ? k : (deg > 0 ? deg - 1 : 0));
  pos = pos < E - 1 ? pos : E - 1;
  if (outd) { out_nbr8[i] = out_indices[pos]; out_pos8[i] = out_pos[pos]; }
  else nbr8[i] = indices[pos];
}

}  // namespace

extern "C" {

int spgnn_arena_load(const spgnn_copy_pad_jobs* i32_jobs, const spgnn_row_copy_jobs* row_jobs, spgnn_stream_t stream) {
  spgnn_copy_pad_jobs ij; ij.n_jobs = 0;
  spgnn_row_copy_jobs rj; rj.n_jobs = 0;
  if (i32_jobs) ij = *i32_jobs;
  if (row_jobs) rj = *row_jobs;
  if (ij.n_jobs < 0 || ij.n_jobs > SPGNN_COPY_PAD_MAX_JOBS || rj.n_jobs < 0 || rj.n_jobs > SPGNN_ROW_COPY_MAX_JOBS)
    return fail(SPGNN_ERR_SHAPE, "spgnn_arena_load: bad job count");
  int64_t longest = 0;
  for (int q = 0; q < ij.n_jobs; ++q) {
    const spgnn_copy_pad_job& j = ij.job[q];
    if (j.n < 0 || j.n_pad < 0) return fail(SPGNN_ERR_SHAPE, "spgnn_arena_load: negative length");
    if (!j.dst || (j.n > 0 && !j.src) || (j.n_pad > 0 && !j.pad)) return fail(SPGNN_ERR_NULLPTR, "spgnn_arena_load: null pointer");
    longest = longest > (int64_t)j.n + j.n_pad ? longest : (int64_t)j.n + j.n_pad;
  }
  for (int q = 0; q < rj.n_jobs; ++q) {
    const spgnn_row_copy_job& j = rj.job[q];
    if (j.width <= 0 || j.rows_copy < 0 || j.rows_total < j.rows_copy || j.dst_col < 0 || j.dst_stride < j.dst_col + j.width || j.src_stride < j.width)
      return fail(SPGNN_ERR_SHAPE, "spgnn_arena_load: bad row-copy job");
    if (!j.dst || (j.rows_copy > 0 && !j.src)) return fail(SPGNN_ERR_NULLPTR, "spgnn_arena_load: null pointer");
    const int64_t n = j.rows_total * j.width;
    longest = longest > n ? longest : n;
  }
  if (longest == 0 || ij.n_jobs + rj.n_jobs == 0) return SPGNN_OK;
  int64_t bx = (longest + 1023) / 1024;                    // about four words per thread
  if (bx > 2048) bx = 2048;
  hipLaunchKernelGGL(arena_load_kernel, dim3((unsigned)bx, (unsigned)(ij.n_jobs + rj.n_jobs)), dim3(256), 0, (hipStream_t)stream, ij, rj);
  return check_launch("spgnn_arena_load");
}

int spgnn_ell_rows_both(const int32_t* indptr, const int32_t* indices, const int32_t* out_indptr, const int32_t* out_indices,
                        const int32_t* out_pos, int64_t N, int64_t E, int32_t* nbr8, int32_t* out_nbr8, int32_t* out_pos8,
                        spgnn_stream_t stream) {
  if (N < 0 || E < 0 || N > (1ll << 27)) return fail(SPGNN_ERR_SHAPE, "spgnn_ell_rows_both: bad N / E");
  if (N == 0) return SPGNN_OK;
  if (E == 0) return fail(SPGNN_ERR_SHAPE, "spgnn_ell_rows_both: a graph without edges has no neighbour rows");
  if (!indptr || !indices || !out_indptr || !out_indices || !out_pos || !nbr8 || !out_nbr8 || !out_pos8)
    return fail(SPGNN_ERR_NULLPTR, "spgnn_ell_rows_both: null pointer");
  hipLaunchKernelGGL(ell_rows_both_kernel, dim3((unsigned)((N * 8 + 255) / 256), 2u), dim3(256), 0, (hipStream_t)stream, indptr, indices,
                     out_indptr, out_indices, out_pos, N, E, nbr8, out_nbr8, out_pos8);
  return check_launch("spgnn_ell_rows_both");
}

int spgnn_copy_pad_i32(const spgnn_copy_pad_jobs* jobs, spgnn_stream_t stream) {
  if (!jobs) return fail(SPGNN_ERR_NULLPTR, "spgnn_copy_pad_i32: null pointer");
  if (jobs->n_jobs < 0 || jobs->n_jobs > SPGNN_COPY_PAD_MAX_JOBS) return fail(SPGNN_ERR_SHAPE, "spgnn_copy_pad_i32: bad job count");
  if (jobs->n_jobs == 0) return SPGNN_OK;
  int64_t longest = 0;
  for (int q = 0; q < jobs->n_jobs; ++q) {
    const spgnn_copy_pad_job& j = jobs->job[q];
    if (j.n < 0 || j.n_pad < 0) return fail(SPGNN_ERR_SHAPE, "spgnn_copy_pad_i32: negative length");
    if (!j.dst || (j.n > 0 && !j.src) || (j.n_pad > 0 && !j.pad)) return fail(SPGNN_ERR_NULLPTR, "spgnn_copy_pad_i32: null pointer");
    longest = longest > (int64_t)j.n + j.n_pad ? longest : (int64_t)j.n + j.n_pad;
  }
  if (longest == 0) return SPGNN_OK;
  int64_t bx = (longest + 255) / 256;
  if (bx > 4096) bx = 4096;
  hipLaunchKernelGGL(copy_pad_i32_kernel, dim3((unsigned)bx, (unsigned)jobs->n_jobs), dim3(256), 0, (hipStream_t)stream, *jobs);
  return check_launch("spgnn_copy_pad_i32");
}

int spgnn_ell_rows(const int32_t* ptr, const int32_t* a0, const int32_t* a1, int64_t N, int64_t E, int32_t* out0, int32_t* out1,
                   spgnn_stream_t stream) {
  if (N < 0 || E < 0 || N > (1ll << 27)) return fail(SPGNN_ERR_SHAPE, "spgnn_ell_rows: bad N / E");
  if (N == 0) return SPGNN_OK;
  if (E == 0) return fail(SPGNN_ERR_SHAPE, "spgnn_ell_rows: a graph without edges has no neighbour rows (the caller writes zeros)");
  if (!ptr || !a0 || !out0 || (a1 && !out1)) return fail(SPGNN_ERR_NULLPTR, "spgnn_ell_rows: null pointer");
  hipLaunchKernelGGL(ell_rows_kernel, dim3((unsigned)((N * 8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, ptr, a0, a1, N, E, out0,
                     out1);
  return check_launch("spgnn_ell_rows");
}

int spgnn_build_csc_count(const uint8_t* adj, const int64_t* adj_ptr, const int64_t* tree_ptr, int64_t num_trees,
                          int32_t* row_count, int32_t* col_count, spgnn_stream_t stream) {
  if (num_trees < 0 || num_trees > (1ll << 30)) return fail(SPGNN_ERR_SHAPE, "spgnn_build_csc_count: bad num_trees");
  if (num_trees == 0) return SPGNN_OK;
  if (!adj || !adj_ptr || !tree_ptr || !row_count || !col_count) return fail(SPGNN_ERR_NULLPTR, "spgnn_build_csc_count: null pointer");
  hipLaunchKernelGGL(build_csc_count_kernel, dim3((unsigned)num_trees), dim3(kThreads), 0, (hipStream_t)stream, adj, adj_ptr, tree_ptr,
                     row_count, col_count);
  return check_launch("spgnn_build_csc_count");
}

int spgnn_build_csc(const uint8_t* adj, const int64_t* adj_ptr, const int64_t* tree_ptr, int64_t num_trees, const int64_t* row_start,
                    const int64_t* col_start, int32_t* src, int32_t* dst, int32_t* indptr, int32_t* indices, int32_t* eid,
                    int32_t* out_indptr, int32_t* out_indices, int32_t* out_pos, int64_t N, int64_t E, spgnn_stream_t stream) {
  if (num_trees < 0 || num_trees > (1ll << 30) || N < 0 || E < N || E >= (1ll << 31)) return fail(SPGNN_ERR_SHAPE, "spgnn_build_csc: bad num_trees / N / E");
  if (!indptr || !out_indptr) return fail(SPGNN_ERR_NULLPTR, "spgnn_build_csc: null pointer");
  if (num_trees == 0) return SPGNN_OK;
  if (!adj || !adj_ptr || !tree_ptr || !row_start || !col_start || !src || !dst || !indices || !eid || !out_indices || !out_pos)
    return fail(SPGNN_ERR_NULLPTR, "spgnn_build_csc: null pointer");
  hipLaunchKernelGGL(build_csc_fill_kernel, dim3((unsigned)num_trees), dim3(kThreads), 0, (hipStream_t)stream, adj, adj_ptr, tree_ptr,
                     num_trees, row_start, col_start, src, dst, indptr, indices, eid, out_indptr, out_indices, out_pos, N, E);
  return check_launch("spgnn_build_csc");
}

}  // extern "C"
