/*
 * spgnn_hip.h — C ABI of the MI355X (gfx950) message-passing library, libspgnn_hip.so.
 *
 * The reference (DIAGNijmegen/spgnn) has no FFI of its own: its graph-convolution hot path
 * calls DGL's Python layers (reference models.py:8), which bottom out in DGL's internal
 * _CAPI_DGLKernelSpMM / _CAPI_DGLKernelSDDMM / edge_softmax kernels.  Each entry point below
 * names the DGL primitive sequence (and the reference call site that triggers it) it replaces.
 *
 * Conventions
 *  - Every pointer is a DEVICE pointer owned by the caller (PyTorch); the library allocates
 *    nothing, keeps no state besides a thread-local last-error string, never synchronises,
 *    and enqueues all work on `stream` (a hipStream_t passed as void*).  Safe under HIP graph
 *    capture.  Re-entrant.
 *  - Graph arrays are int32.  CSC (dst-major): in-edges of v are slots indptr[v]..indptr[v+1]-1,
 *    `indices[slot]` = source node.  CSR (src-major): out-edges of u are
 *    out_indptr[u]..out_indptr[u+1]-1, `out_indices[k]` = destination node, `out_pos[k]` = the CSC
 *    slot of the same edge.  Per-edge arrays (attn, g_e) are always indexed by CSC slot.
 *  - Feature matrices are row-major fp32 with an explicit row stride in ELEMENTS, so callers
 *    can point into wider buffers (fused [fc|res] GEMM output, concat buffers).
 *  - Entry points ending in _bf16 are the bf16-STORAGE path (BASELINE config "st_gat_6 ... bf16"): node-feature
 *    rows, projected rows and their gradients are bfloat16 in HBM (uint16_t* here), every sum is accumulated in
 *    fp32, and parameters, scores (el/er), attention weights and weight gradients stay fp32.  Same semantics and
 *    argument meaning as the fp32 entry point of the same name unless stated.
 *  - Return value: 0 on success; SPGNN_ERR_* (< 0) on bad arguments; -(1000 + hipError_t) when
 *    a launch fails.  spgnn_last_error() describes the last failure on the calling thread.
 */
#ifndef SPGNN_HIP_H_
#define SPGNN_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPGNN_ABI_VERSION 63

#define SPGNN_OK            0
#define SPGNN_ERR_NULLPTR  -1
#define SPGNN_ERR_SHAPE    -2
#define SPGNN_ERR_STRIDE   -3
#define SPGNN_ERR_ENUM     -4

/* activations fused into the GAT epilogue (reference models.py:303,309,440,445: F.elu, F.tanh, None) */
#define SPGNN_ACT_NONE 0
#define SPGNN_ACT_ELU  1
#define SPGNN_ACT_TANH 2
#define SPGNN_ACT_RELU 3
#define SPGNN_ACT_LRELU 4   /* LeakyReLU with torch.nn.LeakyReLU's default slope 0.01 (the reference's GIN MLPs, models.py:236-246): the
                               * GEMM epilogue (spgnn_gemm_nt), spgnn_spmm_sum and spgnn_act_bwd take it; the GAT kernels do not */

typedef void* spgnn_stream_t;   /* hipStream_t */

int         spgnn_abi_version(void);
const char* spgnn_last_error(void);

/*
 * GATConv message passing, forward.  Replaces, per GATConv.forward call (reference
 * models.py:324-326, 477-482, 535-538): gsddmm(u_add_v) + LeakyReLU + edge_softmax
 * (5 DGL kernels) + attn_drop + gspmm(u_mul_e, sum) + residual add + bias add + activation.
 *
 *   e_uv  = leaky_relu(el[u,h] + er[v,h], slope)
 *   a_uv  = exp(e_uv - max_{in(v)} e) / sum_{in(v)} exp(e - max)        -> attn[slot,h]
 *   out[v,h,:] = act( sum_{in(v)} drop(a_uv) * ft[u,h,:] + res[v,h,:] + bias[h,:] )
 *
 * ft/res/out: (N, H*D) with row strides.  el/er: (N, H) with row stride s_stride.
 * res, bias may be NULL.  attn: (E, H) contiguous, always written (saved for backward).
 * out_mean (nullable): (N, D) head mean of the activated output, mean_h out[v,h,:] — the
 * `.mean(1)` the reference applies to the last layer (models.py:327, 482) fused into the epilogue.
 * With out_mean set, `out` may be NULL iff spgnn_gat_can_fuse_mean(H, D) (the backward only needs the
 * per-head output when an activation is fused).
 * p_drop in [0,1): attention dropout; the keep mask is a counter-based hash of (seed, slot, h),
 * regenerated (not stored) by the backward entry points.  seed_offset (nullable): a device word added to
 * `seed` at run time, so a captured HIP graph draws a fresh mask on every replay.
 * nbr8 (nullable): the in-neighbour lists once more as padded rows, (N, 8) int32 with
 *   nbr8[v, k] = indices[indptr[v] + min(k, deg(v) - 1)]   (any valid node id for deg(v) == 0 or k beyond a degree > 8)
 * - "ELL" rows built once per loader batch next to the CSC.  A node's neighbour ids then depend on v alone and are
 * fetched together with indptr[v]: two dependent memory round trips per node (ids + degree; scores + rows) instead of
 * three.  Pure speed: nodes of degree > 8 and calls without nbr8 walk the CSC.  spgnn_gat_bwd_dst takes the same rows,
 * spgnn_gat_bwd_src their out-edge counterparts out_nbr8 / out_pos8 (rows of out_indices / out_pos; both or neither).
 * out_drop_p in [0,1) (0 = off): the CONSUMER's feature dropout applied to the stored per-head output.  The reference feeds
 * a hidden layer's output through the next GATConv's feat_drop, after torch.cat with the position stream
 * (models.py:477-481: h_s = cat[h_s, h_p]; GATConv.feat_drop); with `out` pointing at column out_drop_offset of that
 * next layer's input buffer (row stride out_stride, out_drop_total columns) the rows are written there once, already
 * dropped - element (v, c) scaled by spgnn_cat_dropout's keep factor for (out_drop_seed, out_drop_total,
 * out_drop_offset + c) - instead of being stored plain and copied + masked by a second kernel.  The plain rows are
 * not needed again: spgnn_gat_bwd_dst takes the same four values, applies the mask to g_out and recovers the kept
 * elements of `out` for the activation derivative (where an element was dropped its gradient is zero).  Not with out_mean.
 */
int spgnn_gat_fwd(const int32_t* indptr, const int32_t* indices,
                  const int32_t* nbr8 /* nullable, see below */,
                  const float* ft, int64_t ft_stride,
                  const float* el, const float* er, int64_t s_stride,
                  const float* res, int64_t res_stride,
                  const float* bias,
                  float* out, int64_t out_stride,
                  float* out_mean, int64_t out_mean_stride,
                  float* attn,
                  int64_t N, int64_t E, int32_t H, int32_t D,
                  float negative_slope, int32_t activation,
                  float p_drop, uint64_t seed, const uint64_t* seed_offset /* nullable device word added to seed */,
                  float out_drop_p, uint64_t out_drop_seed, int32_t out_drop_total, int32_t out_drop_offset,
                  float* out_absmax /* nullable: SCALE BLOCK (see spgnn_gemm_nt) that takes max |out| as stored: the consumer GEMM's operand scale */,
                  spgnn_stream_t stream);

/* 1 if the vector kernel fuses the head mean for this (H, D) (a head is at least one team wide). */
int spgnn_gat_can_fuse_mean(int32_t H, int32_t D);

/*
 * GATConv backward, destination-major half (replaces DGL autograd of the sequence above:
 * gsddmm(dot) for g_a, edge_softmax backward, LeakyReLU backward, gspmm(copy_e,sum) for g_er).
 *
 *   g_pre[v,:]  = g_out[v,:] * act'(out[v,:])                -> g_pre   (N, H*D)
 *   g_a_uv      = <ft[u,h,:], g_pre[v,h,:]> * keep_uv/(1-p)
 *   g_e_uv      = (a_uv*g_a_uv - a_uv * sum_{in(v)} a*g_a) * lrelu'(el[u,h]+er[v,h])  -> g_e[slot,h]
 *   g_er[v,h]   = sum_{in(v)} g_e_uv
 *
 * `out` is the forward per-head output (post-activation); ignored when activation == NONE (may be NULL).
 * mean_heads != 0: g_out is the gradient of the head mean, shape (N, D); each head receives g_out/H.
 * g_pre is also the gradient of the residual branch and of the bias (column sums).
 */
int spgnn_gat_bwd_dst(const int32_t* indptr, const int32_t* indices,
                      const int32_t* nbr8 /* nullable */,
                      const float* ft, int64_t ft_stride,
                      const float* el, const float* er, int64_t s_stride,
                      const float* attn,
                      const float* g_out, int64_t g_out_stride, int32_t mean_heads,
                      const float* out, int64_t out_stride,
                      float* g_pre, int64_t g_pre_stride,
                      float* g_e,
                      float* g_er, int64_t g_s_stride,
                      float* absmax /* nullable: scale block taking max |g_pre| */,
                      int64_t N, int64_t E, int32_t H, int32_t D,
                      float negative_slope, int32_t activation,
                      float p_drop, uint64_t seed, const uint64_t* seed_offset,
                      float out_drop_p, uint64_t out_drop_seed, int32_t out_drop_total, int32_t out_drop_offset,
                      spgnn_stream_t stream);

/*
 * GATConv backward, source-major half (DGL: gspmm on the reverse graph for g_ft, and
 * gspmm(copy_e,sum) on the reverse graph for g_el).  Atomics-free.
 *
 *   g_ft[u,h,:] = sum_{v in out(u)} drop(a_uv) * g_pre[v,h,:]
 *   g_el[u,h]   = sum_{v in out(u)} g_e_uv
 * With score_l / score_r set (el = (ft * attn_l).sum(-1) taken from ft itself, DGL's own formulation, see
 * spgnn_gemm_nt's score partials) the scores' gradient returns to ft here:
 *   g_ft[u,h,:] += g_el[u,h] * attn_l[h,:] + g_er[u,h] * attn_r[h,:]
 */
int spgnn_gat_bwd_src(const int32_t* out_indptr, const int32_t* out_indices, const int32_t* out_pos,
                      const int32_t* out_nbr8 /* nullable */, const int32_t* out_pos8 /* nullable */,
                      const float* attn, const float* g_e,
                      const float* g_pre, int64_t g_pre_stride,
                      float* g_ft, int64_t g_ft_stride,
                      float* g_el, int64_t g_s_stride,
                      float* absmax /* nullable: scale block taking max |g_ft| (the same block as spgnn_gat_bwd_dst's: [g_ft | g_pre] is one operand) */,
                      const float* score_l, const float* score_r /* nullable: attn_l, attn_r flat (H*D) */,
                      const float* g_er /* with score_l: (N, H) at stride g_s_stride, from spgnn_gat_bwd_dst */,
                      int64_t N, int64_t E, int32_t H, int32_t D,
                      float p_drop, uint64_t seed, const uint64_t* seed_offset,
                      spgnn_stream_t stream);

/*
 * Tree-resident LDS tiles for the three GATConv traversals above (ABI 54; csrc/spgnn_tile.hip) - BASELINE.json north_star's
 * "LDS staging of neighbour tiles".  Replaces the same DGL primitives as spgnn_gat_fwd / _bwd_dst / _bwd_src (reference
 * models.py:301-314, 425-456); arguments have the meaning they have there.  The batched graph is block-diagonal
 * (dgl.batch, job_runner.py:1882): a workgroup owns a TILE of consecutive nodes - tile t = [tile_ptr[t], tile_ptr[t + 1]),
 * at most max_tile_nodes of them, normally whole trees - and a 64 R-column slice of the rows; it streams the tile's slice of
 * ft (g_pre for the source-major half), its scores, attention words, padded neighbour rows and CSC offsets into LDS once
 * (coalesced 16-byte loads) and serves every gather from there; only a node's own rows are read from / written to global
 * memory.  Tiles need not be closed under neighbours (an id outside the tile takes a global load), so any partition of
 * [0, N) into runs of <= max_tile_nodes is correct; empty tiles are allowed (a fixed-length tile_ptr for batch arenas).
 * Requirements (the caller keeps everything else on the row kernels): 1 <= degree <= 8 in the direction walked, the
 * padded (N, 8) rows nbr8 / out_nbr8 / out_pos8, D in {64, 128, 256} with spgnn_gat_tile_supported(H, D, bytes per row
 * element, max_tile_nodes), no head-mean output.  Forward and source-major half: bit-identical to the row kernels; the
 * destination-major half sums its per-edge dots over another lane geometry (fp32 rounding).
 */
int spgnn_gat_tile_supported(int32_t H, int32_t D, int32_t elem_bytes, int32_t max_tile_nodes);
int spgnn_gat_fwd_tile(const int32_t* tile_ptr, int64_t n_tiles, int32_t max_tile_nodes,
                       const int32_t* indptr, const int32_t* nbr8,
                       const float* ft, int64_t ft_stride,
                       const float* el, const float* er, int64_t s_stride,
                       const float* res /* nullable */, int64_t res_stride,
                       const float* bias /* nullable */,
                       float* out, int64_t out_stride,
                       float* attn,
                       float* absmax /* nullable */,
                       int64_t N, int32_t H, int32_t D,
                       float negative_slope, int32_t activation,
                       float p_drop, uint64_t seed, const uint64_t* seed_offset,
                       float out_drop_p, uint64_t out_drop_seed, int32_t out_drop_total, int32_t out_drop_offset,
                       spgnn_stream_t stream);
int spgnn_gat_bwd_dst_tile(const int32_t* tile_ptr, int64_t n_tiles, int32_t max_tile_nodes,
                           const int32_t* indptr, const int32_t* nbr8,
                           const float* ft, int64_t ft_stride,
                           const float* el, const float* er, int64_t s_stride,
                           const float* attn,
                           const float* g_out, int64_t g_out_stride,
                           const float* out /* nullable iff activation == NONE */, int64_t out_stride,
                           float* g_pre, int64_t g_pre_stride,
                           float* g_e,
                           float* g_er, int64_t g_s_stride,
                           float* absmax /* nullable */,
                           int64_t N, int32_t H, int32_t D,
                           float negative_slope, int32_t activation,
                           float p_drop, uint64_t seed, const uint64_t* seed_offset,
                           float out_drop_p, uint64_t out_drop_seed, int32_t out_drop_total, int32_t out_drop_offset,
                           spgnn_stream_t stream);
int spgnn_gat_bwd_src_tile(const int32_t* tile_ptr, int64_t n_tiles, int32_t max_tile_nodes,
                           const int32_t* indptr,
                           const int32_t* out_indptr, const int32_t* out_nbr8, const int32_t* out_pos8,
                           const float* attn, const float* g_e,
                           const float* g_pre, int64_t g_pre_stride,
                           float* g_ft, int64_t g_ft_stride,
                           float* g_el, int64_t g_s_stride,
                           float* absmax /* nullable */,
                           const float* score_l, const float* score_r /* nullable: attn_l, attn_r flat (H*D) */,
                           const float* g_er /* with score_l */,
                           int64_t N, int32_t H, int32_t D,
                           float p_drop, uint64_t seed, const uint64_t* seed_offset,
                           spgnn_stream_t stream);
/* the same on bf16 rows (bf16 storage, fp32 accumulate; see spgnn_gat_fwd_bf16) */
int spgnn_gat_fwd_tile_bf16(const int32_t* tile_ptr, int64_t n_tiles, int32_t max_tile_nodes,
                            const int32_t* indptr, const int32_t* nbr8,
                            const uint16_t* ft, int64_t ft_stride,
                            const float* el, const float* er, int64_t s_stride,
                            const uint16_t* res, int64_t res_stride,
                            const float* bias,
                            uint16_t* out, int64_t out_stride,
                            float* attn,
                            float* absmax,
                            int64_t N, int32_t H, int32_t D,
                            float negative_slope, int32_t activation,
                            float p_drop, uint64_t seed, const uint64_t* seed_offset,
                            float out_drop_p, uint64_t out_drop_seed, int32_t out_drop_total, int32_t out_drop_offset,
                            spgnn_stream_t stream);
int spgnn_gat_bwd_dst_tile_bf16(const int32_t* tile_ptr, int64_t n_tiles, int32_t max_tile_nodes,
                                const int32_t* indptr, const int32_t* nbr8,
                                const uint16_t* ft, int64_t ft_stride,
                                const float* el, const float* er, int64_t s_stride,
                                const float* attn,
                                const uint16_t* g_out, int64_t g_out_stride,
                                const uint16_t* out, int64_t out_stride,
                                uint16_t* g_pre, int64_t g_pre_stride,
                                float* g_e,
                                float* g_er, int64_t g_s_stride,
                                float* absmax,
                                int64_t N, int32_t H, int32_t D,
                                float negative_slope, int32_t activation,
                                float p_drop, uint64_t seed, const uint64_t* seed_offset,
                                float out_drop_p, uint64_t out_drop_seed, int32_t out_drop_total, int32_t out_drop_offset,
                                spgnn_stream_t stream);
int spgnn_gat_bwd_src_tile_bf16(const int32_t* tile_ptr, int64_t n_tiles, int32_t max_tile_nodes,
                                const int32_t* indptr,
                                const int32_t* out_indptr, const int32_t* out_nbr8, const int32_t* out_pos8,
                                const float* attn, const float* g_e,
                                const uint16_t* g_pre, int64_t g_pre_stride,
                                uint16_t* g_ft, int64_t g_ft_stride,
                                float* g_el, int64_t g_s_stride,
                                float* absmax,
                                const float* score_l, const float* score_r,
                                const float* g_er,
                                int64_t N, int32_t H, int32_t D,
                                float p_drop, uint64_t seed, const uint64_t* seed_offset,
                                spgnn_stream_t stream);

/*
 * Aggregate-first form of the same GATConv (reference call site models.py:456-482, the 192 -> 2 x 1024 output layer
 * of st_pgat_spgnn_3; DGL semantics as spgnn_gat_fwd).  When the layer input (F columns) is narrower than one head's
 * output, sum_u a_uv (W_h x_u) = W_h (sum_u a_uv x_u): the attention-weighted sums run over the INPUT rows, once per
 * head, and the projection follows as a GEMM on [z_h | x] (fc and res_fc in one product, bias + activation in its
 * epilogue: spgnn_gemm_nt).  The H*D-wide projected rows are never gathered.
 *
 *   a_uv        = edge softmax of LeakyReLU(el[u,h] + er[v,h])                      -> attn[slot,h]
 *   z[v, h*head_stride + f]                 = sum_{u in in(v)} drop(a_uv) * x[u,f]     f in [0,F)
 *   z[v, h*head_stride + x_copy_offset + f] = x[v,f]        (x_copy_offset >= F; -1: no residual operand copy;
 *                                                            -2: ONE copy behind the last head's block, z[v, H*head_stride + f] -
 *                                                            the [z_0 | ... | z_{H-1} | x] operand of the linear-mean form)
 *   absmax (nullable): scale block taking max |z| over the columns written (split-GEMM scale of the operand)
 *
 * H in {1,2,4}, F % 4 == 0, F <= 1024 (spgnn_gat_agg_supported); rows 16-byte aligned; head_stride % 4 == 0.
 */
int spgnn_gat_agg_supported(int32_t H, int32_t F);
int spgnn_gat_agg_fwd(const int32_t* indptr, const int32_t* indices,
                      const float* x, int64_t x_stride,
                      const float* el, const float* er, int64_t s_stride,
                      float* attn,
                      float* z, int64_t z_stride, int32_t head_stride, int32_t x_copy_offset,
                      float* absmax,
                      int64_t N, int64_t E, int32_t H, int32_t F,
                      float negative_slope, float p_drop, uint64_t seed, const uint64_t* seed_offset,
                      spgnn_stream_t stream);

/*
 * Its backward halves (replace DGL autograd as spgnn_gat_bwd_dst / spgnn_gat_bwd_src do).  g_z has the layout of z.
 *   dst-major:  g_a_uv = <x[u,:], g_z[v, h-block]> * keep_uv/(1-p);  g_e, g_er as in spgnn_gat_bwd_dst
 *   src-major:  g_x[u,:] = sum_h g_z[u, h-block copy of x]  +  sum_{v in out(u)} sum_h drop(a_uv) * g_z[v, h-block]
 *                          + g_el[u,:] @ w_lr[:H] + g_er[u,:] @ w_lr[H:]      (score projection backward, w_lr nullable)
 *               g_el[u,h] = sum_{v in out(u)} g_e_uv
 */
int spgnn_gat_agg_bwd_dst(const int32_t* indptr, const int32_t* indices,
                          const float* x, int64_t x_stride,
                          const float* el, const float* er, int64_t s_stride,
                          const float* attn,
                          const float* g_z, int64_t g_z_stride, int32_t head_stride,
                          float* g_e, float* g_er, int64_t g_s_stride,
                          int64_t N, int64_t E, int32_t H, int32_t F,
                          float negative_slope, float p_drop, uint64_t seed, const uint64_t* seed_offset,
                          spgnn_stream_t stream);
int spgnn_gat_agg_bwd_src(const int32_t* out_indptr, const int32_t* out_indices, const int32_t* out_pos,
                          const float* attn, const float* g_e,
                          const float* g_z, int64_t g_z_stride, int32_t head_stride, int32_t x_copy_offset,
                          const float* g_er, const float* w_lr, int64_t w_lr_stride,
                          float* g_x, int64_t g_x_stride,
                          float* g_el, int64_t g_s_stride,
                          int64_t N, int64_t E, int32_t H, int32_t F,
                          float p_drop, uint64_t seed, const uint64_t* seed_offset,
                          spgnn_stream_t stream);

/* out_mean[v,d] = mean_h out[v, h*D + d]: `rst.mean(1)` of the reference's output layer (models.py:327, 482) when
 * the per-head rows come out of a GEMM epilogue instead of spgnn_gat_fwd. */
int spgnn_head_mean(const float* out, int64_t out_stride, float* out_mean, int64_t out_mean_stride,
                    int64_t N, int32_t H, int32_t D, spgnn_stream_t stream);

/* g_pre[v,c] = g_out[v, mean_heads ? c % D : c] * (mean_heads ? 1/H : 1) * act'(out[v,c]) (the first phase of
 * spgnn_gat_bwd_dst as an entry point of its own); absmax (nullable): scale block taking max |g_pre|.  D % 4 == 0. */
int spgnn_act_bwd(const float* g_out, int64_t g_out_stride, int32_t mean_heads,
                  const float* out, int64_t out_stride,
                  float* g_pre, int64_t g_pre_stride, float* absmax,
                  int64_t N, int32_t H, int32_t D, int32_t activation, spgnn_stream_t stream);
/* The flat form (no head mean) behind a feature dropout: g_out is the gradient of dropout(act(pre), p_drop) under
 * spgnn_cat_dropout's counter-hash mask (seed [+ *seed_offset], element = row * W + column; total width W, offset 0), which is
 * regenerated:  g_pre = g_out * keep / (1 - p) * act'(out)  - the dropout's backward and the activation's in one pass (the
 * GIN MLP's Linear, Dropout, LeakyReLU, reference models.py:236-246).  `out` = act(pre), BEFORE the dropout. */
int spgnn_act_bwd_dropout(const float* g_out, int64_t g_out_stride, const float* out, int64_t out_stride, float* g_pre,
                          int64_t g_pre_stride, float* absmax /* nullable scale block */, int64_t N, int32_t W,
                          int32_t activation, float p_drop, uint64_t seed, const uint64_t* seed_offset, spgnn_stream_t stream);
/* The same when only the DROPPED activation output was kept (a product or aggregation that applied the dropout in its own
 * epilogue: spgnn_gemm_nt_problem.drop_p, spgnn_spmm_sum_dropout): a kept element is out * (1 - p_drop) again - to one
 * rounding - and a dropped element's derivative does not matter (its mask is 0), so ELU / tanh derivatives come from
 * `out_dropped` as well; for ReLU / LeakyReLU (sign only) both entries give identical results. */
int spgnn_act_bwd_dropped(const float* g_out, int64_t g_out_stride, const float* out_dropped, int64_t out_stride, float* g_pre,
                          int64_t g_pre_stride, float* absmax, int64_t N, int32_t W, int32_t activation, float p_drop,
                          uint64_t seed, const uint64_t* seed_offset, spgnn_stream_t stream);
/* spgnn_act_bwd_dropout that ALSO leaves the column sums of g_pre - the bias gradient when the bias sits in the aggregation's
 * epilogue (GraphConv, reference models.py:172-182; GINConv's first Linear applied before the aggregation) - as per-block
 * partials colsum_partials[spgnn_act_bwd_colsum_blocks(N, W)][W]; spgnn_sum_partials over the blocks gives the (W) sums in
 * a fixed order.  W / 4 must divide 256 (the _blocks helper returns 0 otherwise: use spgnn_act_bwd + a reduction).
 * p_drop = 0: no dropout (then `out` is the undropped activation output, any SPGNN_ACT_*).
 * dot_x (nullable, (N, W) rows): the pass also forms sum_v <g_pre[v, :], dot_x[v, :]> - GINConv's eps gradient
 * (reference models.py:358-383) with dot_x = the aggregation's input rows.  The partial rows then have W + 4 floats
 * (float W = the block's share of the dot product, the rest 0): colsum_partials[blocks][W + 4]. */
int32_t spgnn_act_bwd_colsum_blocks(int64_t N, int32_t W);
int spgnn_act_bwd_colsum(const float* g_out, int64_t g_out_stride, const float* out, int64_t out_stride, float* g_pre,
                         int64_t g_pre_stride, float* absmax, float* colsum_partials, int64_t N, int32_t W, int32_t activation,
                         float p_drop, uint64_t seed, const uint64_t* seed_offset, const float* dot_x, int64_t dot_x_stride,
                         spgnn_stream_t stream);

/* spgnn_act_bwd for a mean-over-heads output layer that feeds a skinny Linear (the reference's classifier
 * `gnn_out = nn.Linear(node_embed_dim, out_ch)`, models.py:1125, on the head mean of models.py:482), with that
 * Linear's input gradient formed on the fly instead of written by spgnn_scores_bwd_x and re-read:
 *   g_pre[v, h*D + c] = (1/H) * (sum_j g_s[v, j] * w[j, c]) * act'(out[v, h*D + c]),  j < J <= 32, H <= 4, D <= 1024.
 * absmax_partials: scale block taking max |g_pre| (spgnn_act_bwd_proj_blocks(N) = the number of workgroups, informational).
 * `out` may be NULL when activation == SPGNN_ACT_NONE. */
int spgnn_act_bwd_proj(const float* g_s, int64_t g_s_stride, int32_t J, const float* w, int64_t w_stride, const float* out,
                       int64_t out_stride, float* g_pre, int64_t g_pre_stride, float* absmax_partials, int64_t N, int32_t H,
                       int32_t D, int32_t activation, spgnn_stream_t stream);
int32_t spgnn_act_bwd_proj_blocks(int64_t N);
/* spgnn_act_bwd_proj that ALSO forms the skinny Linear's weight gradient (ABI 55) from the rows it reads anyway:
 *   g_w[j, c] = sum_v g_s[v, j] * (1/H) sum_h out[v, h*D + c]         (the classifier's g_logits^T @ head mean; DGL: autograd of
 *                                                                      nn.Linear on `.mean(1)`, reference models.py:482, 1125)
 * as per-workgroup partials w_grad_partials[spgnn_act_bwd_proj_wgrad_blocks(N)][J][D]; the caller adds the blocks in order
 * (spgnn_sum_partials / spgnn_sum_partials_multi).  The head mean then need not be read a second time
 * (spgnn_scores_bwd_w).  H in {1, 2}, J <= 24, an activation (so that `out` is read), D % 4 == 0, D <= 1024. */
/* Projection of a SMALL batch (ABI 56): C = act(A B^T + bias), A (M, K), B (N, K), fp32, for M up to a few hundred rows - the
 * reference's per-scan inference (ONE tree per model.forward, job_runner.py:2046-2052) and few-tree batches.  Same contract as
 * spgnn_gemm_nt for the options it has (bias, activation, GATConv's score partials score_out (M, score_cols / 64, 2) of the
 * first score_cols columns with score_l / score_r; those exclude bias / activation), on the fp32 matrix pipe
 * (v_mfma_f32_16x16x4_f32: exact fp32 products, no operand scales, no pre-split forms), 16 x 64 output tiles with the
 * reduction split over the four waves of a workgroup: M / 16 x N / 64 short workgroups instead of a handful of long ones.
 * Replaces rocBLAS (nn.Linear -> cuBLAS in the reference) below ops.SKINNY_ROWS rows. */
int spgnn_gemm_nt_skinny(const float* a, int64_t a_stride, const float* b, int64_t b_stride, float* c, int64_t c_stride, int64_t M,
                         int64_t N, int64_t K, const float* bias /* nullable */, int32_t activation,
                         const float* score_l, const float* score_r, float* score_out /* nullable */, int32_t score_cols,
                         spgnn_stream_t stream);
int32_t spgnn_act_bwd_proj_wgrad_blocks(int64_t N);
int spgnn_act_bwd_proj_wgrad(const float* g_s, int64_t g_s_stride, int32_t J, const float* w, int64_t w_stride, const float* out,
                             int64_t out_stride, float* g_pre, int64_t g_pre_stride, float* absmax_partials, float* w_grad_partials,
                             int64_t N, int32_t H, int32_t D, int32_t activation, spgnn_stream_t stream);

/*
 * Layer-input assembly of the hidden SPGNN layers, dropout(cat[h_s, h_p]) (reference models.py:477-481 + GATConv's
 * feat_drop), one source per call:  dst[:, col_offset : col_offset + width] = src * keep/(1-p)   (backward == 0)
 * and its gradient                    dst = src[:, col_offset : col_offset + width] * keep/(1-p)   (backward != 0, src =
 * gradient of the concatenation).  keep: one 64-bit counter hash (the attention dropout's mixer) per group of four
 * columns starting at a multiple of 4 inside the source, counter = row * total_width + col_offset + c; 16 bits per
 * element, kept iff bits >= p * 65536.
 */
int spgnn_cat_dropout(const float* src, int64_t src_stride, float* dst, int64_t dst_stride, int64_t N, int32_t width,
                      int32_t col_offset, int32_t total_width, float p_drop, uint64_t seed, const uint64_t* seed_offset,
                      int32_t backward,
                      float* absmax_partials /* nullable: scale block taking max |dst| (GEMM operand scale) */,
                      spgnn_stream_t stream);
int64_t spgnn_cat_dropout_blocks(int64_t N, int32_t width);

/* el / er from spgnn_gemm_nt's score partials: s[v,h] = sum_b parts[v, h*D/64 + b, 0], s[v,H+h] = sum_b parts[v, h*D/64 + b, 1]
 * (parts: (N, H*D/64, 2); D % 64 == 0).  Replaces DGL GATConv's (ft * attn).sum(-1) pair. */
int spgnn_scores_from_parts(const float* parts, float* s, int64_t s_stride, int64_t N, int32_t H, int32_t D,
                            spgnn_stream_t stream);

/*
 * Folding of GATConv's score vectors through fc (what makes el/er a projection of the layer INPUT; reference
 * models.py:301-314 via DGL GATConv: el = (fc(x).view(N,H,D) * attn_l).sum(-1)):
 *   w_lr[h,:]   = sum_d attn_l[h,d] * W[h*D+d,:]      w_lr[H+h,:] = sum_d attn_r[h,d] * W[h*D+d,:]
 * w_lr: (2H, w_lr_stride) with columns [K, w_lr_stride) zeroed (the padded form spgnn_scores_* take).
 * Backward: g_attn_l[h,d] = <g_w_lr[h,:], W[h*D+d,:]> (g_attn_r likewise) and the fc-weight term
 *   g_W[h*D+d,:] = attn_l[h,d] * g_w_lr[h,:] + attn_r[h,d] * g_w_lr[H+h,:]   (written, not accumulated).
 */
int spgnn_fold_scores_fwd(const float* W, int64_t w_stride, const float* attn_l, const float* attn_r,
                          float* w_lr, int32_t w_lr_stride, int32_t H, int32_t D, int32_t K, spgnn_stream_t stream);
int spgnn_fold_scores_bwd(const float* W, int64_t w_stride, const float* attn_l, const float* attn_r,
                          const float* g_w_lr, int32_t w_lr_stride,
                          float* g_W, int64_t g_w_stride, float* g_attn_l, float* g_attn_r,
                          int32_t H, int32_t D, int32_t K, spgnn_stream_t stream);

/*
 * Attention-score projections of GATConv: el = (fc(x) * attn_l).sum(-1), er likewise (reference call sites as
 * spgnn_gat_fwd; DGL computes them with two elementwise multiplies + reductions over ft).  With the score
 * vectors folded through fc, W[j,:] = sum_d attn[h,d] * fc.weight[h*D+d,:] (J = 2H rows: el heads, then er
 * heads), they are one skinny projection of the layer input and its two gradients:
 *
 *   spgnn_scores_fwd    S[n,j]   = sum_k x[n,k] * w[j,k]
 *   spgnn_scores_bwd_w  part[s,j,k] = sum_{n in row range s} gs[n,j] * x[n,k]   (caller sums over s)
 *   spgnn_scores_bwd_x  gx[n,k] (+)= sum_j gs[n,j] * w[j,k]        (accumulate != 0: add into gx)
 *
 * w and part rows are zero-padded to Kp = 16*ceil(K/16) floats (part: splits x J x Kp).  x / gx rows must be
 * 16-byte aligned (stride % 4 == 0); spgnn_scores_bwd_w reads (and discards) the up to three floats of row padding that
 * complete a row's last 16-byte group.  J <= 32.  The same three kernels serve the model's classifier head
 * `gnn_out = Linear(1024, 22)` (reference models.py:1125, 1169): a 22-column projection of 76k rows.
 */
int spgnn_scores_fwd(const float* x, int64_t x_stride, const float* w, int32_t Kp,
                     float* s, int64_t s_stride, float* absmax /* nullable: scale block taking max |x| */,
                     const float* bias /* nullable, J floats: s += bias (a skinny Linear's bias, e.g. the classifier's) */,
                     int64_t N, int32_t K, int32_t J, spgnn_stream_t stream);
int spgnn_scores_bwd_w(const float* gs, int64_t gs_stride, const float* x, int64_t x_stride,
                       float* part, int32_t splits, int32_t Kp, int64_t N, int32_t K, int32_t J,
                       spgnn_stream_t stream);
/* Two spgnn_scores_bwd_w passes over the same N nodes in ONE launch (J0, J1 <= 8): a level's structure and position layers'
 * attention-vector gradients (reference models.py:472-484).  Partials exactly as from the two single calls. */
int spgnn_scores_bwd_w_pair(const float* gs0, int64_t gs0_stride, const float* x0, int64_t x0_stride, float* part0, int32_t splits0,
                            int32_t Kp0, int32_t K0, int32_t J0, const float* gs1, int64_t gs1_stride, const float* x1,
                            int64_t x1_stride, float* part1, int32_t splits1, int32_t Kp1, int32_t K1, int32_t J1, int64_t N,
                            spgnn_stream_t stream);

/* Up to eight spgnn_scores_bwd_w passes over the same N rows in ONE launch (fp32 rows, or bf16 rows with x_is_bf16): the
 * attention-vector gradients of all GATConv layers of a model, g_attn = g_s^T ft (DGL: the autograd of (ft * attn).sum(-1),
 * reference models.py:301-314 call sites), collected by the training step and run once the backward pass is through - each
 * pass alone is latency-bound and fills a fraction of the chip.  Job k = the arguments of spgnn_scores_bwd_w (J <= 8); results
 * bit-identical to the single launches.  Partials are summed by the caller (spgnn_sum_partials_multi). */
typedef struct spgnn_scores_bwd_w_job {
  const float* g_s; int64_t g_s_stride; const void* x; int64_t x_stride; float* partials; int32_t splits; int32_t Kp; int32_t K; int32_t J;
} spgnn_scores_bwd_w_job;
int spgnn_scores_bwd_w_multi(const spgnn_scores_bwd_w_job* jobs, int32_t n_jobs, int64_t N, int32_t x_is_bf16, spgnn_stream_t stream);
int spgnn_scores_bwd_x(const float* gs, int64_t gs_stride, const float* w, int32_t Kp,
                       float* gx, int64_t gx_stride, int32_t accumulate, int64_t N, int32_t K, int32_t J,
                       spgnn_stream_t stream);

/*
 * Weighted-sum SpMM (DGL gspmm(copy_u, sum) with the degree normalisations of GraphConv
 * norm='both' / GINConv 'mean' folded in; reference models.py:172-182, 358-383):
 *
 *   out[v,:] = self_coef * x[v,:] + w_dst[v] * sum_{u in in(v)} w_src[u] * x[u,:]
 *
 * w_src, w_dst: per-node scales (N) or NULL (= 1).  self_eps: device pointer to GINConv's `eps`
 * (self_coef = 1 + *self_eps) or NULL (self_coef = 0).  The backward w.r.t. x is the same call
 * on the transposed structure (out_indptr/out_indices) with w_src and w_dst swapped.
 * Optional epilogue out = act(out + bias[col]) (bias nullable, F floats; activation = SPGNN_ACT_*): GraphConv's bias and
 * activation when the aggregation FOLLOWS the projection (in_feats > out_feats, reference models.py:172-182).
 */
int spgnn_spmm_sum(const int32_t* indptr, const int32_t* indices,
                   const float* x, int64_t x_stride,
                   const float* w_src, const float* w_dst, const float* self_eps,
                   const float* bias, int32_t activation,
                   float* out, int64_t out_stride,
                   int64_t N, int64_t E, int32_t F,
                   float* absmax_out /* nullable scale block (see spgnn_gemm_nt): max |out| folded into its slots */,
                   spgnn_stream_t stream);
/* The same with feature dropout of the stored rows, out = dropout(act(...), p_drop), under spgnn_cat_dropout's mask for an
 * F-wide row (counter v * F + column; seed + *seed_offset when the pointer is set) - GINConv's MLP
 * (reference models.py:236-246: Linear -> Dropout -> LeakyReLU) when the first Linear is applied BEFORE the aggregation
 * (in_feats > out_feats: both are linear, so  ((1 + eps) x + A x) W^T = (1 + eps) (x W^T) + A (x W^T)  and the gather runs on
 * the narrow rows).  absmax_out sees the dropped values; spgnn_act_bwd_dropout undoes dropout + activation from them. */
int spgnn_spmm_sum_dropout(const int32_t* indptr, const int32_t* indices, const float* x, int64_t x_stride, const float* w_src,
                           const float* w_dst, const float* self_eps, const float* bias, int32_t activation, float* out,
                           int64_t out_stride, int64_t N, int64_t E, int32_t F, float* absmax_out, float p_drop, uint64_t seed,
                           const uint64_t* seed_offset, spgnn_stream_t stream);

/*
 * Max SpMM (DGL gspmm(copy_u, max); SAGEConv 'pool', reference models.py:668-679):
 *   out[v,f] = max_{u in in(v)} x[u,f]   (0 when v has no in-edge);  arg[v,f] = CSC slot of the
 *   winning edge (first one on ties, as DGL), or -1.
 */
int spgnn_spmm_max_fwd(const int32_t* indptr, const int32_t* indices,
                       const float* x, int64_t x_stride,
                       float* out, int64_t out_stride,
                       int32_t* arg, int64_t arg_stride,
                       int64_t N, int64_t E, int32_t F,
                       spgnn_stream_t stream);

/*
 *   g_x[u,f] = sum_{k in out(u)} [arg[out_indices[k], f] == out_pos[k]] * g_out[out_indices[k], f]
 */
int spgnn_spmm_max_bwd(const int32_t* out_indptr, const int32_t* out_indices, const int32_t* out_pos,
                       const float* g_out, int64_t g_out_stride,
                       const int32_t* arg, int64_t arg_stride,
                       float* g_x, int64_t g_x_stride,
                       int64_t N, int64_t E, int32_t F,
                       spgnn_stream_t stream);

/* The same pair with a COMPACT argmax: one byte per element = the winner's position inside v's in-edge list (255: none; every
 * in-degree must be <= 254), rows 4-byte aligned.  The backward pass gathers the arg row of every out-neighbour, so this
 * quarter-size form is what the training path uses whenever spgnn_spmm_max_u8_supported(F) (F / 4 a whole number of 16-,
 * 32- or 64-lane teams with 1, 2, 4 or 8 float4 per lane); `indptr` = the CSC offsets the forward call walked. */
int32_t spgnn_spmm_max_u8_supported(int32_t F);
int spgnn_spmm_max_fwd_u8(const int32_t* indptr, const int32_t* indices, const float* x, int64_t x_stride, float* out,
                          int64_t out_stride, uint8_t* arg, int64_t arg_stride, int64_t N, int64_t E, int32_t F,
                          spgnn_stream_t stream);
int spgnn_spmm_max_bwd_u8(const int32_t* indptr, const int32_t* out_indptr, const int32_t* out_indices, const int32_t* out_pos,
                          const float* g_out, int64_t g_out_stride, const uint8_t* arg, int64_t arg_stride, float* g_x,
                          int64_t g_x_stride, int64_t N, int64_t E, int32_t F, spgnn_stream_t stream);
/* spgnn_spmm_max_bwd_u8 for an x that was a ReLU output (SAGEConv 'pool': max over relu(fc_pool(h)), reference
 * models.py:668-679): g_x comes out already multiplied by relu'(x) = [relu_out > 0] - the gradient of fc_pool's
 * pre-activation - and max |g_x| is folded into the scale block `absmax` (nullable): no activation-backward pass over g_x. */
int spgnn_spmm_max_bwd_u8_relu(const int32_t* indptr, const int32_t* out_indptr, const int32_t* out_indices, const int32_t* out_pos,
                               const float* g_out, int64_t g_out_stride, const uint8_t* arg, int64_t arg_stride, float* g_x,
                               int64_t g_x_stride, const float* relu_out, int64_t relu_out_stride, float* absmax, int64_t N,
                               int64_t E, int32_t F, spgnn_stream_t stream);

/*
 * fp32-accurate projection GEMM on the fp16 matrix cores (replaces the cuBLAS/rocBLAS SGEMMs behind
 * nn.Linear `fc` / `res_fc` inside DGL's GATConv, reference models.py:301-314, and their input gradients):
 *
 *   C[M,N] = A[M,K] * B[N,K]^T            A, B, C fp32 row-major with row strides lda, ldb, ldc
 *
 * Every operand value is split on the fly into two fp16 terms after multiplication by a per-tensor
 * power-of-two scale (scale_a, scale_b: SCALE BLOCKS in device memory, or NULL = 1) and the three
 * leading products are accumulated in fp32 (v_mfma_f32_32x32x16_f16); the result carries fp32-GEMM accuracy.
 * A scale block is either {s}: one positive float, the scale itself (spgnn_pow2_scale, spgnn_weight_prep), or
 * {-256, 0, 0, 0, m_1 ... m_256}: 256 "slots" of maxima which the kernels that PRODUCE the operand fold their rows'
 * largest magnitudes into (their `absmax` arguments; one result-free atomicMax per node team - max is order-independent,
 * so the scale is deterministic); the GEMM derives s = 2^(14 - e), max_i m_i <= 2^e, itself.  The slots must be zero
 * before the first producer runs.  This replaces one reduction launch per operand and step.  Header word 1 of a slot block is an
 * OUTPUT of the consuming product: it is set to 1 when a non-zero slot lies more than 2^18 below the largest one (the range monitor,
 * see spgnn_step_begin) - the pointers are `const` because the scale itself is read-only, the flag word is not.
 * A and B rows must be 16-byte aligned (lda, ldb multiples of 4); K may be ragged.
 * Optional exact fp32 rank-J update fused into the epilogue: C += U[M,J] * V[J,N] (upd_j <= 32; V rows
 * 16-byte aligned and zero padded to a multiple of 4 columns; upd_j = 0 disables it).  The layer uses it for
 * the score term of the input gradient, g_X = g_Y * W + g_S * W_lr, instead of a second pass over g_X.
 * Optional epilogue C = act(C + bias[col]) (bias nullable, N floats; activation = SPGNN_ACT_*): GATConv's bias and
 * activation when the projection FOLLOWS the aggregation (spgnn_gat_agg_fwd).
 * Optional score partials (score_out nullable): GATConv's el = (ft * attn_l).sum(-1), er likewise, taken from the
 * product while it is still in registers.  For each row and each 64-column block b of the first score_cols output
 * columns (= H*D, a multiple of 64; D % 64 == 0 so that no block straddles two heads)
 *   score_out[(row * (score_cols/64) + b) * 2 + 0] = <C[row, 64b:64b+64], score_l[64b:64b+64]>     (score_l = attn_l flat)
 *   score_out[(row * (score_cols/64) + b) * 2 + 1] = <C[row, 64b:64b+64], score_r[64b:64b+64]>
 * computed on the raw product (before rank-J / bias / activation); spgnn_scores_from_parts sums a head's blocks.
 * b_presplit = 1: B is the PRE-SPLIT form of the operand written by spgnn_presplit with scale_b (same shape and strides;
 * every 16-byte group of four fp32 values replaced by the packed fp16 pairs [hi01, hi23, lo01, lo23] the kernel would
 * otherwise form itself for every row tile).  Bit-identical results; weights are split once per step this way.
 * The argument is a MASK (every NT entry point): SPGNN_PRESPLIT_B (1) as above; SPGNN_PRESPLIT_A | SPGNN_PRESPLIT_B (3):
 * A is pre-split as well (spgnn_presplit with scale_a).  A model's first layer multiplies node DATA - cat[fvs, pos_enc],
 * reference models.py:425-428, 474-477 - which is constant over the GCN_STEPS = 300 inner steps on a loader batch
 * (job_runner.py:1892): it is split once per batch, and its products neither convert nor stage it as fp32 any more.
 * A-only (2) is not instantiated (SPGNN_ERR_ENUM).  Bit-identical to the in-kernel split under the same scale.
 */
#define SPGNN_PRESPLIT_B 1
#define SPGNN_PRESPLIT_A 2
/* SPGNN_GEMM_WIDE (may be or-ed into the mask of every NT entry point; SPGNN_TN_WIDE for the TN ones): WIDE-RANGE arithmetic.
 * hi + lo carries 22 bits only for values within 2^18 of the tensor's maximum - below that lo is an fp16 subnormal - so an
 * operand whose rows differ by more than that (gradients late in training, un-normalised features) loses relative accuracy
 * on its small rows.  In the wide form lo is kept as 2^11 lo (a normal fp16 wherever hi is one), the two cross products go to
 * a second accumulator set that the epilogue adds times 2^-11: the same three MFMAs per product, 22 bits within ~2^28 of the
 * maximum.  The wide kernels run in 128 x 128 (NT) / 128-row (TN) tiles, one workgroup per CU (two accumulator sets); an operand pre-split for one form
 * (spgnn_presplit `wide`, spgnn_weight_prep mode bit 1) must be consumed in that form. */
#define SPGNN_GEMM_WIDE 4
int spgnn_gemm_nt(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc,
                  int64_t M, int64_t N, int64_t K, const float* scale_a, const float* scale_b,
                  const float* upd_u, int64_t upd_u_stride, const float* upd_v, int64_t upd_v_stride, int32_t upd_j,
                  const float* bias, int32_t activation,
                  const float* score_l, const float* score_r, float* score_out, int32_t score_cols,
                  int32_t b_presplit, spgnn_stream_t stream);

/* Pre-split form of up to two fp32 matrices sharing ONE power-of-two scale (a weight operand and its transpose): dst has
 * the shape and row stride of src (16-byte rows, stride >= the width rounded up to 4; the pad columns are written as zero).
 * The scale is either given (scale_in, device scalar) or derived from block maxima partials[0..n_partials) exactly as
 * spgnn_scale_from_partials would (then scale_in = null); scale_out (nullable) receives it.  src1 / dst1 nullable. */
int spgnn_presplit(const float* partials, int64_t n_partials, const float* scale_in, float* scale_out,
                   const float* src0, int64_t ld0, int64_t R0, int64_t K0, float* dst0,
                   const float* src1, int64_t ld1, int64_t R1, int64_t K1, float* dst1, int32_t wide /* 1: the SPGNN_GEMM_WIDE form */,
                   spgnn_stream_t stream);

/* spgnn_gemm_nt for the SECOND head of a two-head layer whose heads are averaged (the reference's output GATConv,
 * `.mean(1)` at models.py:327 / 482): besides C = act(A B^T + bias) it writes
 *   mean_out[row, col] = 0.5 * (C[row, col] + other_head[row, col])
 * from the tile in registers (other_head = the first head's C, computed by an earlier launch on the same stream), which
 * replaces spgnn_head_mean's pass over both heads.  Strides % 4 == 0, 16-byte aligned bases. */
int spgnn_gemm_nt_headmean(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc,
                           int64_t M, int64_t N, int64_t K, const float* scale_a, const float* scale_b,
                           const float* bias, int32_t activation,
                           const float* other_head, int64_t other_head_stride, float* mean_out, int64_t mean_out_stride,
                           int32_t b_presplit, spgnn_stream_t stream);

/* C = act(A B^T + bias + addend): the second of two products that share an output adds the first one's result in its epilogue
 * (SAGEConv: fc_self(h) + fc_neigh(neigh), reference models.py:668-679) - no separate addition and activation passes.
 * addend: (M, N) fp32, 16-byte aligned rows (may be C itself). */
int spgnn_gemm_nt_add(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int64_t M, int64_t N,
                      int64_t K, const float* scale_a, const float* scale_b, const float* bias, int32_t activation,
                      const float* addend, int64_t addend_stride, int32_t b_presplit, spgnn_stream_t stream);

/* spgnn_gemm_nt with the block tile pinned: tile = 0 chosen from the shape (= spgnn_gemm_nt), 2 = 128 x 128,
 * 4 = 256 x 128, 5 = 256 x 256 (operand extents below 2^31 bytes).  Every tile shape performs the same arithmetic in the
 * same order per output element, so the results are bit-identical; a caller that knows its shapes can skip the heuristic. */
int spgnn_gemm_nt_tile(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc,
                       int64_t M, int64_t N, int64_t K, const float* scale_a, const float* scale_b,
                       const float* upd_u, int64_t upd_u_stride, const float* upd_v, int64_t upd_v_stride, int32_t upd_j,
                       const float* bias, int32_t activation,
                       const float* score_l, const float* score_r, float* score_out, int32_t score_cols,
                       int32_t tile, int32_t b_presplit, spgnn_stream_t stream);

/* out[i] = sum over s < splits of partials[s * split_stride + i], i < n (n % 4 == 0, 16-byte aligned): the deterministic
 * reduction of the split-K partial tiles of spgnn_gemm_tn and spgnn_scores_bwd_w (fixed summation order). */
int spgnn_sum_partials(const float* partials, int64_t split_stride, int32_t splits, int64_t n, float* out, spgnn_stream_t stream);

/* The block diagonal of the summed partials of spgnn_scores_bwd_w(g_s (N, 2H), ft (N, H*D)) as (2, H, D):
 * out[(w*H + h)*D + d] = sum_s partials[s][w*H + h][h*D + d] = the gradients of DGL GATConv's attn_l (w = 0) and
 * attn_r (w = 1), `(ft * attn).sum(-1)`; ld = row stride of a partial (>= H*D). */
int spgnn_sum_partials_blockdiag(const float* partials, int64_t split_stride, int32_t splits, int32_t H, int32_t D, int32_t ld,
                                 float* out, spgnn_stream_t stream);

/* The same reduction for partial TILES with padded rows (spgnn_gemm_tn: M rows of ld_in >= N floats per split), written
 * where the gradients live: columns [0, N) to out (row stride out_stride), or - with out2 - columns [0, split_col) to
 * out and [split_col, N) to out2 (the aggregate-first layer's [W_fc,h | W_res,h] gradient goes to the two parameters'
 * row blocks directly); extra (nullable, M floats) = the same sum over column extra_col of the partial rows (the bias
 * column sums that spgnn_gemm_tn lets ride in a spare column). */
int spgnn_sum_partials_compact(const float* partials, int64_t split_stride, int32_t splits, int32_t M, int32_t N, int64_t ld_in,
                               float* out, int64_t out_stride, float* out2, int64_t out2_stride, int32_t split_col, float* extra,
                               int32_t extra_col, spgnn_stream_t stream);

/* Weight preparation of a projection layer in one pass.  The reference multiplies by fc.weight and res_fc.weight
 * separately (DGL GATConv: self.fc(h), self.res_fc(h); models.py:301-314 call sites); here both share one GEMM, whose B
 * operand is dst = [a ; b] (rows_a + rows_b rows of K columns, row stride dst_stride >= K with the pad columns zeroed,
 * 16-byte rows).  dst_t (nullable) receives the transpose (K rows, row stride dst_t_stride >= rows_a + rows_b, pad
 * columns zeroed): the B operand of the input-gradient product.  absmax_partials gets
 * spgnn_weight_cat_partials(rows_a + rows_b, K, dst_stride, dst_t ? dst_t_stride : 0) block maxima of |w| for
 * spgnn_scale_from_partials.  rows_b may be 0 (b ignored). */
int spgnn_weight_cat(const float* a, int64_t a_stride, int32_t rows_a, const float* b, int64_t b_stride, int32_t rows_b, int32_t K,
                     float* dst, int64_t dst_stride, float* dst_t, int64_t dst_t_stride, float* absmax_partials,
                     spgnn_stream_t stream);
int64_t spgnn_weight_cat_partials(int32_t rows, int32_t K, int64_t dst_stride, int64_t dst_t_stride);

/* scale[0] = 2^(14 - e) with factor * max_i partials[i] <= 2^e: turns the partial maxima emitted by
 * spgnn_scores_fwd / spgnn_gat_bwd_dst / spgnn_gat_bwd_src (which stream the tensors anyway) into a GEMM scale. */
int spgnn_scale_from_partials(const float* partials, int64_t n, float factor, float* scale,
                              uint32_t* workspace /* nullable; 2 words, zeroed ONCE by the caller, self-resetting */,
                              spgnn_stream_t stream);

/*
 * Weight-gradient form: C[M,N] = A[R,M]^T * B[R,N] with the reduction over the R rows (nodes) of both
 * operands (A = g_Y, B = X; replaces the SGEMM-TN behind nn.Linear's weight gradient).  The row range is
 * cut into `splits` chunks, chunk s writing its partial product to C + s*split_stride (each M x ldc);
 * the caller sums the partials (splits == 1: C is the result).  Same split-fp16 arithmetic as spgnn_gemm_nt.
 * colsum_a (nullable): per-split column sums of A in plain fp32, taken from the operand stream the kernel reads
 * anyway — with A = [g_ft | g_pre] its right half summed over splits is the bias gradient.  Element (split, m) goes
 * to colsum_a[split * colsum_split_stride + m * colsum_stride], so the sums can live in a spare column of the
 * partial-product buffer and fall out of the same reduction over splits.
 */
int spgnn_gemm_tn(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc,
                  int64_t split_stride, int32_t splits, int64_t R, int64_t M, int64_t N,
                  const float* scale_a, const float* scale_b,
                  float* colsum_a, int64_t colsum_stride, int64_t colsum_split_stride, spgnn_stream_t stream);

/* Two INDEPENDENT products in one launch each: workgroups [0, b0) run the first product, the rest the second, each with
 * the arithmetic and tile order it has in a launch of its own (results bit-identical to two calls).  A GATConv level's
 * structure and position layers (reference models.py:472-484: `self.gat_layers[l](g, h)`, `self.pgnn_layers[l](g, p)`)
 * project independently; the small product's memory-bound tiles then run on the CUs the large one's last round of tiles
 * leaves idle instead of in a launch of their own.  Fields = the arguments of spgnn_gemm_nt / spgnn_gemm_tn.  The NT
 * pair runs both products with the block tile spgnn_gemm_nt would choose for `first` (pass the larger one first). */
typedef struct spgnn_gemm_nt_problem {
  const float* A; int64_t lda; const float* B; int64_t ldb; float* C; int64_t ldc; int64_t M; int64_t N; int64_t K;
  const float* scale_a; const float* scale_b; const float* upd_u; int64_t upd_u_stride; const float* upd_v; int64_t upd_v_stride;
  const float* bias; const float* score_l; const float* score_r; float* score_out;
  const float* addend; int64_t addend_stride;      /* nullable: C = act(A B^T + bias + addend), as spgnn_gemm_nt_add */
  float* absmax_out;                               /* nullable scale block: max |C| is folded into its slots, so the result
                                                      can be the next product's operand without an absmax pass over it */
  int32_t upd_j; int32_t activation; int32_t score_cols; int32_t reserved;
  /* optional feature dropout of the stored result, C = dropout(act(...), drop_p), under spgnn_cat_dropout's mask for an N-wide
   * row (counter row * N + column; seed + *drop_seed_offset when the pointer is set): the reference's GIN MLP
   * (models.py:236-246, Linear -> Dropout -> LeakyReLU; the two commute for ReLU / LeakyReLU) without a pass over the
   * product.  drop_p = 0: none.  Needs N % 4 == 0.  absmax_out sees the dropped values.  spgnn_act_bwd_dropout undoes
   * both from the stored result alone (ReLU / LeakyReLU: the derivative depends on the sign only). */
  const uint64_t* drop_seed_offset; uint64_t drop_seed; float drop_p; float reserved2;
} spgnn_gemm_nt_problem;
typedef struct spgnn_gemm_tn_problem {
  const float* A; int64_t lda; const float* B; int64_t ldb; float* C; int64_t ldc; int64_t split_stride; int64_t R; int64_t M; int64_t N;
  const float* scale_a; const float* scale_b; float* colsum_a; int64_t colsum_stride; int64_t colsum_split_stride;
  int32_t splits;
  int32_t flags;           /* SPGNN_TN_B_PRESPLIT: B (R x N; X, the layer input) is the pre-split form written by spgnn_presplit with
                              scale_b - the weight gradient of a model's first layer reads constant node data (see spgnn_gemm_nt);
                              SPGNN_TN_TILE_128 / _256: pin the block tile's rows (default: spgnn_gemm_tn_tile_rows) */
} spgnn_gemm_tn_problem;
#define SPGNN_TN_B_PRESPLIT 1
#define SPGNN_TN_TILE_128 0x10
#define SPGNN_TN_TILE_256 0x20
#define SPGNN_TN_WIDE 0x100          /* the wide-range arithmetic of SPGNN_GEMM_WIDE (B pre-split in that form, if pre-split) */
/* Rows of the (rows x 128) block tile spgnn_gemm_tn / _problem_run / _pair take for this shape: 256 (8 waves, one workgroup per
 * CU) when M is a multiple of 256, M * N >= 384 * 1024 and R >= 4096, else 128.  Both tile shapes give bit-identical results; a caller
 * choosing `splits` counts tiles with it: ceil(M / rows) * ceil(N / 128). */
int32_t spgnn_gemm_tn_tile_rows(int64_t R, int64_t M, int64_t N, int32_t flags);
int spgnn_gemm_nt_problem_run(const spgnn_gemm_nt_problem* problem, int32_t b_presplit, spgnn_stream_t stream);   /* one product, every option */
int spgnn_gemm_nt_pair(const spgnn_gemm_nt_problem* first, const spgnn_gemm_nt_problem* second, int32_t b_presplit,
                       spgnn_stream_t stream);
int spgnn_gemm_tn_pair(const spgnn_gemm_tn_problem* first, const spgnn_gemm_tn_problem* second, spgnn_stream_t stream);   /* same SPGNN_TN_B_PRESPLIT in both; the first one's tile */
int spgnn_gemm_tn_problem_run(const spgnn_gemm_tn_problem* problem, spgnn_stream_t stream);                               /* one product (= spgnn_gemm_tn + flags) */

/* scale[0] = 2^(14 - e), max|x| <= 2^e (1 for an all-zero tensor).  workspace: up to 2048 floats of device
 * memory for per-block partial maxima (no atomics).  x rows must be 16-byte aligned. */
int spgnn_pow2_scale(const float* x, int64_t x_stride, int64_t rows, int64_t cols, float* scale,
                     float* workspace, int32_t workspace_floats, spgnn_stream_t stream);

/*
 * Distance positional encoding of a batch of trees (reference job_runner.py:1759-1777: networkx all-pairs
 * shortest paths + diameter per tree, on the host):
 *   pos_enc[t, a] = (float)((double)hops(t, anchors[tree(t), a]) / (double)diameter(tree(t)))
 * out_indptr/out_indices: CSR of the batched graph (self loops allowed, ignored by the BFS); tree_ptr[B+1]:
 * node offsets of the trees (int64); anchors: (B, num_anchors) GLOBAL node ids; diameters (nullable): (B).
 * Trees must be connected and hold at most 2048 nodes (pass the batch maximum as max_tree_nodes).
 */
int spgnn_tree_distance_encoding(const int32_t* out_indptr, const int32_t* out_indices, const int64_t* tree_ptr,
                                 const int32_t* anchors, int32_t num_anchors, float* pos_enc, int64_t pos_enc_stride,
                                 int32_t* diameters, int64_t num_trees, int64_t max_tree_nodes, spgnn_stream_t stream);

/*
 * Anchor selection of the distance positional encoding, one batch per call (reference job_runner.py:1727-1757
 * get_anchors_from_cnn_prediction, 1712-1725 add_distal_leafs).  prob: (num_nodes, >= num_labels + 1)
 * softmax of the CNN logits.  Per tree, anchors[t, l-1] = argmax_i prob[i, l] over the nodes not yet taken (first
 * maximum), l = 1 .. num_labels; then anchors[t, num_labels + k] = the farthest descendant leaf of anchor k in the
 * downstream DAG (edges u -> v, v > u), k < num_distal, ties resolved as the reference's Python does (last leaf in
 * the iteration order of the set nx.descendants returns: CPython's small-int set is replayed).  anchors:
 * (num_trees, num_labels + num_distal) int32 GLOBAL node ids - the input of spgnn_tree_distance_encoding.
 * Every tree needs at least num_labels nodes (caller checks).  workspace: spgnn_tree_anchors_workspace() bytes.
 */
int64_t spgnn_tree_anchors_workspace(int64_t num_trees, int32_t num_distal, int64_t max_tree_nodes);
int spgnn_tree_anchors(const float* prob, int64_t prob_stride, const int32_t* out_indptr, const int32_t* out_indices,
                       const int64_t* tree_ptr, int64_t num_trees, int64_t num_nodes, int64_t max_tree_nodes,
                       int32_t num_labels, int32_t num_distal, int32_t* anchors, void* workspace, spgnn_stream_t stream);

/*
 * Masked, class-weighted cross entropy of the training step in one pass (reference job_runner.py:1896-1900:
 * `mask = rn < sampling_t`, `F.cross_entropy(pre[mask], y[mask], weight=w)`):
 *   partials[2b], partials[2b+1] = per-256-node-block sums of  m_i w[y_i] nll_i  and  m_i w[y_i]   (b < ceil(N/256))
 *   g_logits[i,c] (nullable)     = m_i w[y_i] (softmax(logits[i,:])_c - [c == y_i])   (gradient of the numerator)
 * with m_i = draws[i] < sampling_p[i].  labels: int64.  The loss is sum(partials[0::2]) / sum(partials[1::2]).
 */
int spgnn_masked_ce(const float* logits, int64_t logits_stride, const int64_t* labels, const float* draws,
                    const float* sampling_p, const float* class_weight, float* partials,
                    float* g_logits, int64_t g_stride, int64_t N, int32_t C, spgnn_stream_t stream);
/*
 * The same pass as the training step issues it.  `draws` nullable: rn_i = 24 bits of the kernels' counter hash of
 * (draw_seed + 0xD1B54A32D192ED03 * seed_offset[0], i) / 2^24 - `seed_offset` (nullable, device int64) is the step counter a
 * captured step reads, so every replay draws a fresh mask (the reference draws its GCN_STEPS x N matrix with numpy up
 * front, job_runner.py:1889-1890).  `sums` (nullable, 2 floats) with `ticket` (one uint32, zero before the first launch;
 * the kernel re-arms it): the workgroup that arrives last adds the per-block pairs in block order and writes
 * [sum m w nll, sum m w] - what `partials.sum(0)` gives, without a reduction launch, bitwise independent of arrival order.
 */
int spgnn_masked_ce_step(const float* logits, int64_t logits_stride, const int64_t* labels, const float* draws, uint64_t draw_seed,
                         const int64_t* seed_offset, const float* sampling_p, const float* class_weight, float* partials,
                         float* sums, uint32_t* ticket, float* g_logits, int64_t g_stride,
                         float* colsum_partials /* nullable: 32 floats per 256-node block */,
                         float* g_colsum /* nullable, C <= 32, needs sums + g_logits: sum_i g_logits[i, :] - the gradient of a classifier
                                            bias - added by the last workgroup in block order */,
                         int64_t N, int32_t C, spgnn_stream_t stream);

/*
 * The rows of a step that reach the loss.  The reference evaluates ``F.cross_entropy(pre[mask], y[mask], weight=w)``
 * (job_runner.py:1896-1900): a node outside the mask contributes neither to the loss nor to any gradient, and the output
 * layer's projection, the classifier (models.py:1127-1174) and their backward products are row-wise - a training step may
 * run them on the kept rows only (spgnn_amd.train.TrainStep(loss_rows_only=True); identical loss and gradients).
 * spgnn_loss_rows lists the kept nodes in ascending order with the loss kernel's own draw (`draws`, or - null - the counter
 * hash of (draw_seed, seed_offset[0], node)): idx[0 .. cnt) = the nodes, idx[cnt .. cap) = 0, inv[n] = position of node n
 * or -1, cnt_flag[0] = cnt = min(kept, cap), cnt_flag[1] set (never cleared) when kept > cap - spgnn_masked_ce_rows then
 * returns NaN sums.  block_counts: workspace of ceil(N / 256) int32.  Two launches, no host synchronisation.
 */
int spgnn_loss_rows(const float* draws /* nullable */, uint64_t draw_seed, const int64_t* seed_offset /* nullable */,
                    const float* sampling_p, int64_t N, int32_t* block_counts, int32_t cap, int32_t* idx, int32_t* inv,
                    int32_t* cnt_flag, spgnn_stream_t stream);
/*
 * spgnn_act_bwd_proj for a row list: g_pre (cap rows) row q = the pass's result for node rows[q] (rows of g_s and out are
 * addressed through the list) when q < rows_cnt[0], zero otherwise.  In a dense step the rows of g_pre outside the mask are
 * exactly zero (their g_s rows are): the layer's two backward products may run on the listed rows alone.
 */
int spgnn_act_bwd_proj_rows(const float* g_s, int64_t g_s_stride, int32_t J, const float* w, int64_t w_stride, const float* out,
                            int64_t out_stride, const int32_t* rows, const int32_t* rows_cnt, float* g_pre, int64_t g_pre_stride,
                            float* absmax_partials, int64_t cap, int32_t H, int32_t D, int32_t activation, spgnn_stream_t stream);
/*
 * spgnn_gat_agg_bwd_dst / _src reading the gradient of the z blocks through a row list: g_z_listed holds one row per listed
 * node, node v's row is g_z_listed[inv[v]] and is all zeros when inv[v] < 0 (such a node's g_e and g_er are written as zeros
 * without reading its neighbours' rows; its contribution to every g_x is nothing).  Every CSC / CSR slot is still visited.
 */
int spgnn_gat_agg_bwd_dst_rows(const int32_t* indptr, const int32_t* indices, const float* x, int64_t x_stride,
                               const float* el, const float* er, int64_t s_stride, const float* attn, const float* g_z_listed,
                               int64_t g_z_stride, int32_t head_stride, const int32_t* inv, float* g_e, float* g_er,
                               int64_t g_s_stride, int64_t N, int64_t E, int32_t H, int32_t F, float negative_slope, float p_drop,
                               uint64_t seed, const uint64_t* seed_offset, spgnn_stream_t stream);
int spgnn_gat_agg_bwd_src_rows(const int32_t* out_indptr, const int32_t* out_indices, const int32_t* out_pos, const float* attn,
                               const float* g_e, const float* g_z_listed, int64_t g_z_stride, int32_t head_stride,
                               int32_t x_copy_offset, const int32_t* inv, const float* g_er, const float* w_lr, int64_t w_lr_stride,
                               float* g_x, int64_t g_x_stride, float* g_el, int64_t g_s_stride, int64_t N, int64_t E, int32_t H,
                               int32_t F, float p_drop, uint64_t seed, const uint64_t* seed_offset, spgnn_stream_t stream);
/* dst[c, :] = c < cnt_flag[0] ? src[idx[c], :] : 0 for c in [0, cap); cols % 4 == 0, 16-byte aligned rows. */
int spgnn_gather_rows(const float* src, int64_t src_stride, const int32_t* idx, const int32_t* cnt_flag, int64_t cap, int32_t cols,
                      float* dst, int64_t dst_stride, spgnn_stream_t stream);
/* dst[n, :] = inv[n] >= 0 ? src[inv[n], :] : 0 for n in [0, N): the listed rows back in node order, zeros elsewhere. */
int spgnn_expand_rows(const float* src, int64_t src_stride, const int32_t* inv, int64_t N, int32_t cols, float* dst,
                      int64_t dst_stride, spgnn_stream_t stream);
/* spgnn_masked_ce_step whose class weights turn NaN when rows_cnt[1] != 0 (the step's row list overflowed; a step that uses the
 * list in its backward pass only has a dense loss, which would otherwise not show it). */
int spgnn_masked_ce_step_flagged(const float* logits, int64_t logits_stride, const int64_t* labels, const float* draws, uint64_t draw_seed,
                                 const int64_t* seed_offset, const float* sampling_p, const int32_t* rows_cnt, const float* class_weight,
                                 float* partials, float* sums, uint32_t* ticket, float* g_logits, int64_t g_stride,
                                 float* colsum_partials, float* g_colsum, int64_t N, int32_t C, spgnn_stream_t stream);
/*
 * spgnn_masked_ce_step on LISTED rows: logits / g_logits have `cap` rows, row i belongs to node rows[i] (its label:
 * labels[rows[i]]) and counts when i < rows_cnt[0]; no draw (the list is the draw).  Same sums, gradient and column sums.
 */
int spgnn_masked_ce_rows(const float* logits, int64_t logits_stride, const int64_t* labels, const int32_t* rows,
                         const int32_t* rows_cnt, const float* class_weight, float* partials, float* sums, uint32_t* ticket,
                         float* g_logits, int64_t g_stride, float* colsum_partials, float* g_colsum, int64_t cap, int32_t C,
                         spgnn_stream_t stream);

/*
 * ABI 61.  The tail of a training step behind the output layer in ONE pass over its rows: the skinny classifier
 * (reference models.py:1125, 1167-1170: n_out = gnn_out(n_embed)), the masked class-weighted cross entropy of
 * spgnn_masked_ce_step (job_runner.py:1896-1900; same mask rule, same draws, same NaN weight for a label outside [0, J))
 * and the classifier's weight / bias gradient.  Replaces spgnn_scores_fwd + spgnn_masked_ce_step + spgnn_scores_bwd_w on
 * (x, w): x (N, K) is read from HBM once instead of twice.
 *   logits[n, :]   = x[n, :] w^T + bias                      w (J, Kp) zero padded to Kp = 16 ceil(K / 16), J <= 32
 *   g_logits[n, :] = m_n class_weight[y_n] (softmax(logits[n, :]) - e_{y_n})        (gradient of the loss NUMERATOR)
 *   sums           = [sum_n m_n cw[y_n] nll_n, sum_n m_n cw[y_n]]   (last workgroup, block order; `ticket` as spgnn_masked_ce_step)
 *   w_partials     (B, J, Kp): partials of g_logits^T x, B = spgnn_classifier_ce_partial_slices(N, K, J);
 *                  add them in block order (spgnn_sum_partials / _multi) for the classifier's weight gradient
 *   g_colsum       (J) nullable with colsum_partials (B, 32): sum_n g_logits[n, :], the classifier bias' gradient
 * `flag` nullable: int32[2], flag[1] != 0 turns every weight into NaN (a step whose row list overflowed).
 * K % 128 == 0, K <= 1024; x rows 16-byte aligned.  partials: (B, 2) workspace.  No host synchronisation.
 */
int spgnn_classifier_ce_rows_per_block(int64_t N);
/* slices of `w_partials` (the B above) for a problem: workgroups x row groups (rows of at most 512 columns with 17 .. 24
 * classes are shared by 2 or 4 groups of the workgroup's column threads, each with a partial slice of its own) */
int64_t spgnn_classifier_ce_partial_slices(int64_t N, int32_t K, int32_t J);
int spgnn_classifier_ce(const float* x, int64_t x_stride, const float* w, int32_t Kp, const float* bias /* nullable */,
                        const int64_t* labels, const float* draws /* nullable */, uint64_t draw_seed,
                        const int64_t* seed_offset /* nullable */, const float* sampling_p, const float* class_weight,
                        const int32_t* flag /* nullable */, float* logits, int64_t logits_stride, float* g_logits, int64_t g_stride,
                        float* w_partials, float* partials, float* sums, uint32_t* ticket, float* colsum_partials /* nullable */,
                        float* g_colsum /* nullable */, int64_t N, int32_t K, int32_t J, spgnn_stream_t stream);
/* ABI 63: the same pass over bf16 rows (BASELINE config 4: the folded classifier of the linear-mean output layer reads the bf16
 * rows [z_0 .. z_{H-1} | x]); the rows are widened to fp32 exactly, every product and sum is fp32 as above.  x rows 8-byte aligned. */
int spgnn_classifier_ce_bf16(const uint16_t* x, int64_t x_stride, const float* w, int32_t Kp, const float* bias /* nullable */,
                             const int64_t* labels, const float* draws /* nullable */, uint64_t draw_seed,
                             const int64_t* seed_offset /* nullable */, const float* sampling_p, const float* class_weight,
                             const int32_t* flag /* nullable */, float* logits, int64_t logits_stride, float* g_logits, int64_t g_stride,
                             float* w_partials, float* partials, float* sums, uint32_t* ticket, float* colsum_partials /* nullable */,
                             float* g_colsum /* nullable */, int64_t N, int32_t K, int32_t J, spgnn_stream_t stream);

/*
 * Neighbour sampling on the device-resident CSC: dgl.sampling.sample_neighbors + dgl.to_block of the reference's
 * sampled GraphSAGE loop (job_runner.py:1484-1499, dgl.dataloading.MultiLayerNeighborSampler(node_ks) behind
 * NodeDataLoader).  Two calls per block with one host read of (num_edges, num_extra) between them.
 *
 * spgnn_sample_neighbors: seed s = parent node seeds[s] keeps k_s = min(indegree, fanout) in-edges (all when
 * fanout < 0), uniformly without replacement, in CSC order.  The caller passes out_indptr[S+1] = exclusive prefix sum
 * of k_s, `local` = int32[num_nodes] filled with -1 and `flag` = int32[num_nodes] zeros.  On return
 *   out_src[out_indptr[s] .. out_indptr[s+1]) = parent ids of the kept sources (out_eid likewise the parent edge ids,
 *     or CSC slots when eid is null; out_eid nullable),
 *   local[seeds[s]] = s, flag[u] = 1 for every kept source u that is not a seed.
 * Seeds must be distinct (a duplicate leaves local[seeds[s]] != s for some s; the host side checks that).
 * The draws are a pure function of (seed, CSC slot): mix64 as in spgnn_cat_dropout, top 32 bits, multiply-shift.
 *
 * spgnn_block_relabel: rank = inclusive prefix sum of flag.  Numbers the flagged nodes num_seeds + rank - 1 (the
 * block's extra sources, ascending parent id), writes them to extra_nodes[num_extra] (int64 parent ids) and rewrites
 * the edge sources: src_local[e] = local[out_src[e]].
 */
int spgnn_sample_neighbors(const int32_t* indptr, const int32_t* indices, const int32_t* eid, int64_t num_nodes,
                           const int64_t* seeds, int64_t num_seeds, int32_t fanout, const int32_t* out_indptr, uint64_t seed,
                           int32_t* local, int32_t* out_src, int32_t* out_eid, int32_t* flag, spgnn_stream_t stream);
int spgnn_block_relabel(const int32_t* flag, const int32_t* rank, int32_t* local, int64_t num_nodes, int64_t num_seeds,
                        const int32_t* out_src, int64_t num_edges, int64_t* extra_nodes, int32_t* src_local,
                        spgnn_stream_t stream);

/*
 * SGD with momentum over one flat fp32 parameter bucket (torch.optim.SGD semantics, dampening 0,
 * no nesterov; reference exp_settings/st_pgat_spgnn_3.py OPTIMIZER, job_runner.py:1919):
 *   g = grad[i] * (*grad_scale if grad_scale else 1) + weight_decay * p[i]
 *   buf[i] = first_step ? g : momentum * buf[i] + g ;  p[i] -= lr * buf[i]
 * grad_scale is a DEVICE scalar (1 / global sum of class weights after the all-reduce).
 */
int spgnn_sgd_momentum_step(float* param, const float* grad, float* momentum_buf,
                            const float* grad_scale, const float* lr_dev /* nullable: overrides lr (graph replay) */,
                            int64_t n, float lr, float momentum, float weight_decay, int32_t first_step,
                            spgnn_stream_t stream);
/*
 * The same update when the step's loss is a class-weighted MEAN: `weight_sum` (device scalar, the global sum of class
 * weights after the all-reduce) replaces grad_scale = 1 / weight_sum, and - `loss_out` non-null - the first thread also
 * writes loss_out[0] = loss_num[0] / weight_sum[0] (the reciprocal and the loss scalar were two launches).
 */
int spgnn_sgd_momentum_step_mean(float* param, const float* grad, float* momentum_buf, const float* weight_sum,
                                 const float* loss_num, float* loss_out, const float* lr_dev, int64_t n, float lr,
                                 float momentum, float weight_decay, int32_t first_step, spgnn_stream_t stream);
/*
 * spgnn_sgd_momentum_step_mean with a guard: a step whose loss (loss_num[0] / weight_sum[0]) is not finite is not applied -
 * parameters and momentum buffer untouched, skipped_steps[0] += 1, loss_out still carries the NaN.  Used by the loss-rows
 * training steps, whose row list can (with probability ~1e-15 per step on the batch it was sized on, more on a later batch of
 * the arena with more labelled nodes) overflow its fixed capacity: the loss is then NaN by construction, the step is lost
 * instead of the parameters, and the host enlarges the list at the next loader batch (TrainStep.check_loss_rows).
 */
int spgnn_sgd_momentum_step_guarded(float* param, const float* grad, float* momentum_buf, const float* weight_sum,
                                    const float* loss_num, float* loss_out, const float* lr_dev, uint32_t* skipped_steps, int64_t n,
                                    float lr, float momentum, float weight_decay, int32_t first_step, spgnn_stream_t stream);
/*
 * What a training step arms before its first kernel, in one launch: `counter` (nullable, device int64: the dropout / mask
 * stream position the kernels read through their `seed_offset` arguments) += 1, and the `n_scale_blocks` scale blocks at
 * `scale_blocks` (260 floats each, see spgnn_gemm_nt) return to {-256, 0, 0, 0, 0 x 256}.
 * RANGE MONITOR: a split GEMM whose slot-block operand has a non-zero slot more than 2^18 below its largest slot (whole rows
 * or blocks of the tensor outside the range in which hi + lo carries 22 bits) sets header word 1 of that block; this call adds
 * the number of flagged blocks to `range_violations` (nullable, device uint32, sticky) before re-arming them.  Detection
 * only - the products' arithmetic is unchanged; a caller that sees the counter move re-runs in plain fp32.
 */
int spgnn_step_begin(int64_t* counter, float* scale_blocks, int32_t n_scale_blocks, uint32_t* range_violations, spgnn_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * bf16-storage path (BASELINE.json config 4: st_gat_6, 512 trees, bf16; reference precision hook
 * job_runner.py:263-280, model exp_settings/st_gat_6.py:81-106).  Rows are bfloat16 with 8-byte aligned rows
 * (stride % 4 == 0); only vector geometries (H*D = 4*T*R) are supported - there is no scalar fallback.
 * ------------------------------------------------------------------------------------------------ */

/* spgnn_gat_fwd with ft / res / out as bf16 rows; out_mean (the head mean, `.mean(1)` of the output layer) stays fp32. */
int spgnn_gat_fwd_bf16(const int32_t* indptr, const int32_t* indices, const int32_t* nbr8,
                       const uint16_t* ft, int64_t ft_stride,
                       const float* el, const float* er, int64_t s_stride,
                       const uint16_t* res, int64_t res_stride,
                       const float* bias,
                       uint16_t* out, int64_t out_stride,
                       float* out_mean, int64_t out_mean_stride,
                       float* attn,
                       int64_t N, int64_t E, int32_t H, int32_t D,
                       float negative_slope, int32_t activation,
                       float p_drop, uint64_t seed, const uint64_t* seed_offset,
                       float out_drop_p, uint64_t out_drop_seed, int32_t out_drop_total, int32_t out_drop_offset,
                       spgnn_stream_t stream);

/* spgnn_gat_bwd_dst with ft / out / g_pre as bf16 rows.  g_out: bf16 (N, H*D), or - mean_heads != 0 - the fp32
 * gradient of the head mean (N, D).  The per-edge dots use the ROUNDED g_pre (what the other kernels read back). */
int spgnn_gat_bwd_dst_bf16(const int32_t* indptr, const int32_t* indices, const int32_t* nbr8,
                           const uint16_t* ft, int64_t ft_stride,
                           const float* el, const float* er, int64_t s_stride,
                           const float* attn,
                           const void* g_out, int64_t g_out_stride, int32_t mean_heads,
                           const uint16_t* out, int64_t out_stride,
                           uint16_t* g_pre, int64_t g_pre_stride,
                           float* g_e,
                           float* g_er, int64_t g_s_stride,
                           int64_t N, int64_t E, int32_t H, int32_t D,
                           float negative_slope, int32_t activation,
                           float p_drop, uint64_t seed, const uint64_t* seed_offset,
                           float out_drop_p, uint64_t out_drop_seed, int32_t out_drop_total, int32_t out_drop_offset,
                           spgnn_stream_t stream);

/* spgnn_gat_bwd_src with g_pre / g_ft as bf16 rows (score vectors and score gradients fp32). */
int spgnn_gat_bwd_src_bf16(const int32_t* out_indptr, const int32_t* out_indices, const int32_t* out_pos,
                           const int32_t* out_nbr8, const int32_t* out_pos8,
                           const float* attn, const float* g_e,
                           const uint16_t* g_pre, int64_t g_pre_stride,
                           uint16_t* g_ft, int64_t g_ft_stride,
                           float* g_el, int64_t g_s_stride,
                           const float* score_l, const float* score_r, const float* g_er,
                           int64_t N, int64_t E, int32_t H, int32_t D,
                           float p_drop, uint64_t seed, const uint64_t* seed_offset,
                           spgnn_stream_t stream);

/* The aggregate-first kernels (spgnn_gat_agg_fwd / _bwd_dst / _bwd_src above) and spgnn_scores_fwd on bf16 rows: x, the z
 * blocks, their gradient g_z and g_x are bf16 (8-byte aligned rows, stride % 4 == 0); el / er / attn / g_e / g_el / g_er
 * and w / w_lr stay fp32.  Same arguments otherwise (no absmax by-product: the bf16 GEMMs need no operand scale).  Used
 * for the GAT OUTPUT layer without activation whose head mean is one product on [z_0 | .. | z_{H-1} | x]
 * (reference models.py:320-327). */
int spgnn_gat_agg_fwd_bf16(const int32_t* indptr, const int32_t* indices, const uint16_t* x, int64_t x_stride,
                           const float* el, const float* er, int64_t s_stride, float* attn,
                           uint16_t* z, int64_t z_stride, int32_t head_stride, int32_t x_copy_offset,
                           int64_t N, int64_t E, int32_t H, int32_t F,
                           float negative_slope, float p_drop, uint64_t seed, const uint64_t* seed_offset,
                           spgnn_stream_t stream);
int spgnn_gat_agg_bwd_dst_bf16(const int32_t* indptr, const int32_t* indices, const uint16_t* x, int64_t x_stride,
                               const float* el, const float* er, int64_t s_stride, const float* attn,
                               const uint16_t* g_z, int64_t g_z_stride, int32_t head_stride,
                               float* g_e, float* g_er, int64_t g_s_stride,
                               int64_t N, int64_t E, int32_t H, int32_t F,
                               float negative_slope, float p_drop, uint64_t seed, const uint64_t* seed_offset,
                               spgnn_stream_t stream);
int spgnn_gat_agg_bwd_src_bf16(const int32_t* out_indptr, const int32_t* out_indices, const int32_t* out_pos,
                               const float* attn, const float* g_e,
                               const uint16_t* g_z, int64_t g_z_stride, int32_t head_stride, int32_t x_copy_offset,
                               const float* g_er, const float* w_lr, int64_t w_lr_stride,
                               uint16_t* g_x, int64_t g_x_stride, float* g_el, int64_t g_s_stride,
                               int64_t N, int64_t E, int32_t H, int32_t F,
                               float p_drop, uint64_t seed, const uint64_t* seed_offset,
                               spgnn_stream_t stream);
int spgnn_scores_fwd_bf16(const uint16_t* x, int64_t x_stride, const float* w, int32_t Kp,
                          float* s, int64_t s_stride, const float* bias /* nullable */, int64_t N, int32_t K, int32_t J,
                          spgnn_stream_t stream);

/* spgnn_scores_bwd_x writing (or accumulating into) bf16 rows g_x; the row padding up to a multiple of 4 is written as zeros. */
int spgnn_scores_bwd_x_bf16(const float* g_s, int64_t g_s_stride, const float* w, int32_t Kp, uint16_t* g_x, int64_t g_x_stride,
                            int32_t accumulate, int64_t N, int32_t K, int32_t J, spgnn_stream_t stream);

/* spgnn_scores_bwd_w with x as bf16 rows (the attention vectors' gradients g_s^T ft; the folded classifier's g_logits^T Zx). */
int spgnn_scores_bwd_w_bf16(const float* g_s, int64_t g_s_stride, const uint16_t* x, int64_t x_stride, float* partials,
                            int32_t splits, int32_t Kp, int64_t N, int32_t K, int32_t J, spgnn_stream_t stream);

/* spgnn_cat_dropout on bf16 rows (widths, offsets and strides multiples of 4; no absmax by-product). */
int spgnn_cat_dropout_bf16(const uint16_t* src, int64_t src_stride, uint16_t* dst, int64_t dst_stride, int64_t N,
                           int32_t width, int32_t col_offset, int32_t total_width, float p_drop, uint64_t seed,
                           const uint64_t* seed_offset, int32_t backward, spgnn_stream_t stream);

/*
 * C[M,N] = act(A[M,K] * B[N,K]^T + bias) on v_mfma_f32_32x32x16_bf16, fp32 accumulate: GATConv's `fc` / `res_fc`
 * projections as ONE product Y = X [W_fc ; W_res]^T (reference models.py:301-314 -> DGL GATConv.fc / res_fc), and the
 * input gradient g_X = g_Y W with B = W^T.  A, B: bf16 rows, 16-byte aligned, stride % 8 == 0 and >= K rounded up to 8
 * with columns [K, K8) zero.  C: bf16 (c_is_f32 == 0) or fp32; N % 4 == 0.  bias (N, fp32) / activation optional.
 * score_out (nullable): per row and per 64-column block b of the first score_cols columns,
 *   score_out[(row * score_cols/64 + b) * 2 + {0,1}] = <C[row, 64b:64b+64], score_l / score_r[64b:64b+64]>
 * taken from the values AS STORED (bf16-rounded): DGL's el / er = (ft * attn_l / attn_r).sum(-1) without a pass over ft
 * (spgnn_scores_from_parts adds a head's blocks).  Not combined with bias / activation.
 */
int spgnn_gemm_nt_bf16(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, void* C, int64_t ldc,
                       int32_t c_is_f32, int64_t M, int64_t N, int64_t K, const float* bias, int32_t activation,
                       const float* score_l, const float* score_r, float* score_out, int32_t score_cols,
                       spgnn_stream_t stream);
/* The same product with the block tile named (tests): 0 = chosen from the shape, 2 = 128 x 128, 4 = 256 x 128,
 * 5 = 256 x 256 (waves of 128 x 64).  Every tile performs the same arithmetic per output element: bit-identical results. */
int spgnn_gemm_nt_bf16_tile(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, void* C, int64_t ldc,
                            int32_t c_is_f32, int64_t M, int64_t N, int64_t K, const float* bias, int32_t activation,
                            const float* score_l, const float* score_r, float* score_out, int32_t score_cols,
                            int32_t tile, spgnn_stream_t stream);
/* The product with score partials and the LAYOUT of score_out named: 0 = (M, score_cols / 64, 2) partial pairs as above;
 * 1 = (M, 2 * score_cols / 64) rows [el_0 .. el_{H-1} | er_0 .. er_{H-1}] - for layers whose heads are exactly one 64-column
 * block wide (D = 64: the five 2 x 64 layers of st_gat_6, reference exp_settings/st_gat_6.py:81-106) the dots ARE el / er and
 * go straight where spgnn_gat_fwd_bf16 reads them: no spgnn_scores_from_parts launch. */
int spgnn_gemm_nt_bf16_scores(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, void* C, int64_t ldc,
                              int32_t c_is_f32, int64_t M, int64_t N, int64_t K, const float* score_l, const float* score_r,
                              float* score_out, int32_t score_cols, int32_t score_layout, spgnn_stream_t stream);

/*
 * Weight gradients: partial[s] (M, N; row stride ldc; fp32) = A[rows of split s, :M]^T * B[rows of split s, :N] for
 * `splits` row ranges of the R rows (A = g_Y, B = X, both bf16 rows, 16-byte aligned, stride % 8 == 0, pad columns up
 * to a multiple of 8 zero); the caller sums the partials (spgnn_sum_partials_compact).  colsum (nullable): per split
 * the column sums of A (the bias gradient), element (s, m) at colsum[s * colsum_split_stride + m * colsum_stride].
 */
int spgnn_gemm_tn_bf16(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, float* partials, int64_t ldc,
                       int64_t split_stride, int32_t splits, int64_t R, int64_t M, int64_t N, float* colsum,
                       int64_t colsum_stride, int64_t colsum_split_stride, spgnn_stream_t stream);

/* [w_a ; w_b] (fp32 parameter row blocks, w_b nullable) -> bf16 GEMM operand w (rows_a + rows_b, w_stride) and, w_t
 * non-null, its transpose (K, w_t_stride); pad columns are written as zeros. */
int spgnn_weight_cat_bf16(const float* w_a, int64_t a_stride, int32_t rows_a, const float* w_b, int64_t b_stride,
                          int32_t rows_b, int32_t K, uint16_t* w, int64_t w_stride, uint16_t* w_t, int64_t w_t_stride,
                          spgnn_stream_t stream);

/* The same for every projection layer of a forward pass in ONE launch.  `jobs`: a DEVICE table, one entry per layer (fields =
 * the arguments above; tiles_x from spgnn_weight_cat_bf16_blocks, first_block = the running sum of that function's results,
 * total_blocks = the final sum).  Results are bit-identical to n_jobs single calls. */
typedef struct spgnn_weight_cat_bf16_job {
  const float* a; int64_t a_stride; const float* b; int64_t b_stride; uint16_t* w; int64_t w_stride; uint16_t* w_t; int64_t w_t_stride;
  int32_t rows_a; int32_t rows_b; int32_t K; int32_t first_block; int32_t tiles_x; int32_t reserved;
} spgnn_weight_cat_bf16_job;
int32_t spgnn_weight_cat_bf16_blocks(int32_t rows, int32_t K, int64_t w_stride, int64_t w_t_stride /* 0: no transpose */, int32_t* tiles_x);
int spgnn_weight_cat_bf16_multi(const spgnn_weight_cat_bf16_job* jobs, int32_t n_jobs, int32_t total_blocks, spgnn_stream_t stream);

/* x (N, K) fp32 -> y (N, y_stride) bf16 (round to nearest even), columns [K, y_stride) zero: node data (fvs, pos_enc)
 * converted once per loader batch. */
int spgnn_cast_rows_bf16(const float* x, int64_t x_stride, int64_t N, int32_t K, uint16_t* y, int64_t y_stride,
                         spgnn_stream_t stream);

/* =================================================================================================
 * One traversal per SPGNN level (LSPE): the structure GATConv (two heads) and the position GATConv (one head) of
 * reference models.py:472-484 - `h_s = gat_layers[l](g, cat[h_s, h_p])`, `h_p = pgnn_layers[l](g, h_p)` - walk the batched CSC
 * together.  Replaces, per level, two spgnn_gat_fwd / two spgnn_gat_bwd_dst / two spgnn_gat_bwd_src launches, the
 * concatenation + feature-dropout passes over the position rows (spgnn_cat_dropout forward and backward) and the addition
 * of h_p's two gradient contributions.  groups[0] = the structure layer (H = 2), groups[1] = the position layer (H = 1);
 * both have out_feats = D, D in {64, 128, 256} (spgnn_lspe_supported).  Every node must have 1 <= degree <= 8 in both
 * directions and the padded neighbour rows (nbr8 / out_nbr8 / out_pos8, see spgnn_gat_fwd) are required.
 *
 * Per group: ft / res (N, H*D) projected and residual rows (both required: the reference's position layer always has a residual
 * and so has every structure layer of its configs), bias (H*D, nullable), el / er (N, H) with row
 * stride s_stride, attn (E, H) (written by the forward, read by the backward), activation, LeakyReLU slope, attention dropout
 * (p_drop, seed) with the same counter hash as spgnn_gat_fwd (index slot * H + h), so a level gives exactly the masks the
 * two layers would have drawn on their own.
 * ================================================================================================= */
typedef struct spgnn_lspe_fwd_group {
  const float* ft; int64_t ft_stride; const float* res; int64_t res_stride; const float* bias;
  float* el; float* er; int64_t s_stride; float* attn;
  /* nullable (both groups or neither): spgnn_gemm_nt's score partials of this layer's projection, (N, H*D/64, 2).  The kernel
   * then forms el / er itself (the sums of spgnn_scores_from_parts, same order) and WRITES them to el / er for the backward. */
  const float* score_parts;
  int32_t H; int32_t act; float slope; float p_drop; uint64_t seed;
} spgnn_lspe_fwd_group;

int spgnn_lspe_supported(int32_t D);

/*
 * out (N, 3D): [structure head 0 | structure head 1 | position head], activated, stored under the NEXT structure layer's
 * feature dropout (out_drop_p, out_drop_seed; mask of spgnn_cat_dropout for a 3D-wide concatenation: the reference's
 * `dropout(cat[h_s, h_p])`, models.py:477-481 + GATConv.feat_drop) - that layer's input buffer, written once.
 * out2 (N, D): the position head once more under the next POSITION layer's feature dropout (out2_drop_p = 0: plain, e.g. the
 * model's h_p output) - that layer's input.  out_absmax / out2_absmax (nullable): scale blocks taking the maxima of the stored
 * rows (operand scales of the next projections).
 */
int spgnn_lspe_fwd(const int32_t* indptr, const int32_t* nbr8, const spgnn_lspe_fwd_group* groups /* [2] */, float* out,
                   int64_t out_stride, float out_drop_p, uint64_t out_drop_seed, float* out2, int64_t out2_stride,
                   float out2_drop_p, uint64_t out2_drop_seed, float* out_absmax, float* out2_absmax, int64_t N, int64_t E,
                   int32_t D, const uint64_t* seed_offset, spgnn_stream_t stream);

typedef struct spgnn_lspe_bwd_dst_group {
  const float* ft; int64_t ft_stride; const float* el; const float* er; int64_t s_stride; const float* attn;
  float* g_pre; int64_t g_pre_stride;      /* out: gradient of the pre-activation rows (N, H*D) */
  float* g_e;                               /* out: (E, H) */
  float* g_er; int64_t gs_stride;           /* out: (N, H) */
  float* absmax;                            /* nullable: scale block of [g_ft | g_pre] (max |g_pre| folded in) */
  int32_t H; int32_t act; float slope; float p_drop; uint64_t seed;
} spgnn_lspe_bwd_dst_group;

/*
 * dst-major half of the level's backward pass.  g_out (N, 3D): gradient of the buffer spgnn_lspe_fwd wrote; g_out2 (N, D;
 * nullable): gradient of the position head's second copy - the position rows have two consumers and their gradient is the
 * sum of both, each under its own mask.  out / out2: the stored rows (for the activation derivative; kept elements are
 * recovered as stored * (1 - p), and where a position element was dropped in `out` it is taken from `out2`).
 */
int spgnn_lspe_bwd_dst(const int32_t* indptr, const int32_t* nbr8, const spgnn_lspe_bwd_dst_group* groups /* [2] */,
                       const float* g_out, int64_t g_out_stride, const float* g_out2, int64_t g_out2_stride, const float* out,
                       int64_t out_stride, float out_drop_p, uint64_t out_drop_seed, const float* out2, int64_t out2_stride,
                       float out2_drop_p, uint64_t out2_drop_seed, int64_t N, int64_t E, int32_t D,
                       const uint64_t* seed_offset, spgnn_stream_t stream);

typedef struct spgnn_lspe_bwd_src_group {
  const float* attn; const float* g_e;      /* (E, H) in CSC slot order */
  const float* g_pre; int64_t g_pre_stride;
  float* g_ft; int64_t g_ft_stride;         /* out: gradient of the projected rows (N, H*D) */
  float* g_el; const float* g_er; int64_t gs_stride;   /* g_el out, g_er in: (N, H) */
  const float* score_l; const float* score_r;           /* attn_l / attn_r (H*D), both or neither: the score term */
  float* absmax;                            /* nullable: the same scale block (max |g_ft| folded in) */
  int32_t H; float p_drop; uint64_t seed;
} spgnn_lspe_bwd_src_group;

/* src-major half: g_ft[u, h, :] = sum over out-edges of drop(a) * g_pre[v, h, :] + g_el[u, h] attn_l[h, :] + g_er[u, h] attn_r[h, :]. */
int spgnn_lspe_bwd_src(const int32_t* out_indptr, const int32_t* out_nbr8, const int32_t* out_pos8,
                       const spgnn_lspe_bwd_src_group* groups /* [2] */, int64_t N, int64_t E, int32_t D,
                       const uint64_t* seed_offset, spgnn_stream_t stream);

/* =================================================================================================
 * Batch assembly on the device (reference job_runner.py:1319-1344, 1779-1801 and dgl.batch at 1390 / 1882): from the packed
 * adjacency matrices of a loader batch - tree t is the n_t x n_t uint8 matrix at adj + adj_ptr[t] (row-major), its nodes are
 * tree_ptr[t] .. tree_ptr[t+1]-1 of the batch - to the batched edge list in the reference's edge-id order (per tree: the
 * off-diagonal non-zeros (u, v) sorted by (u, v), then the n self loops) and the int32 index structures every kernel of this
 * library takes: CSC (indptr, indices, eid; in-lists in ascending edge id) and CSR (out_indptr, out_indices, out_pos =
 * the CSC slot of the same edge).  The diagonal of adj is ignored (the rule drops it and appends the self loops).
 *
 *   1. spgnn_build_csc_count: row_count[i] / col_count[i] = off-diagonal non-zeros in node i's row / column;
 *   2. the caller forms the exclusive prefix sums row_start / col_start (N + 1 entries, int64) and E = row_start[N] + N;
 *   3. spgnn_build_csc fills src / dst (E), indptr / out_indptr (N + 1), indices / eid / out_indices / out_pos (E).
 * ================================================================================================= */
/*
 * A batch arena's index arrays in one launch (spgnn_amd/arena.py: a loader batch or an inference scan copied into fixed
 * buffers and padded to its size class): per job dst[i] = src[i], i < n; dst[n + i] = pad[i] + pad_add, i < n_pad.
 */
#define SPGNN_COPY_PAD_MAX_JOBS 8
typedef struct spgnn_copy_pad_job {
  int32_t* dst; const int32_t* src; const int32_t* pad;
  int32_t n, n_pad, pad_add, reserved;
} spgnn_copy_pad_job;
typedef struct spgnn_copy_pad_jobs { spgnn_copy_pad_job job[SPGNN_COPY_PAD_MAX_JOBS]; int32_t n_jobs; } spgnn_copy_pad_jobs;
int spgnn_copy_pad_i32(const spgnn_copy_pad_jobs* jobs, spgnn_stream_t stream);
/*
 * ABI 62.  A whole arena load in one launch: the index arrays of spgnn_copy_pad_i32 (`i32_jobs`, nullable) plus 2-D copies of
 * rows of 4-byte WORDS (`row_jobs`, nullable; all lengths and strides in words, so fp32 and int64 node data alike):
 *   dst[r, dst_col : dst_col + width] = src[r, 0 : width]  for r < rows_copy,   = 0  for rows_copy <= r < rows_total
 * - the loaded batch's node data into the arena's buffers (pad rows a larger earlier batch filled go back to zero) and the
 * tensors derived from it (cat[fvs, pos_enc] with 16-byte rows, aligned copies; reference models.py:474-477 reads them per
 * forward) straight from the incoming batch.  Per-scan inference (job_runner.py:2046-2052) paid ~17 small launches here.
 */
#define SPGNN_ROW_COPY_MAX_JOBS 16
typedef struct spgnn_row_copy_job {
  uint32_t* dst; const uint32_t* src;
  int64_t dst_stride, src_stride, rows_copy, rows_total;
  int32_t dst_col, width;
} spgnn_row_copy_job;
typedef struct spgnn_row_copy_jobs { spgnn_row_copy_job job[SPGNN_ROW_COPY_MAX_JOBS]; int32_t n_jobs; } spgnn_row_copy_jobs;
int spgnn_arena_load(const spgnn_copy_pad_jobs* i32_jobs /* nullable */, const spgnn_row_copy_jobs* row_jobs /* nullable */,
                     spgnn_stream_t stream);
/* spgnn_ell_rows for both directions of a graph in one launch: nbr8 from (indptr, indices), out_nbr8 / out_pos8 from
 * (out_indptr, out_indices, out_pos). */
int spgnn_ell_rows_both(const int32_t* indptr, const int32_t* indices, const int32_t* out_indptr, const int32_t* out_indices,
                        const int32_t* out_pos, int64_t N, int64_t E, int32_t* nbr8, int32_t* out_nbr8, int32_t* out_pos8,
                        spgnn_stream_t stream);
/*
 * The padded neighbour rows the row kernels read next to indptr (`nbr8` / `out_nbr8` / `out_pos8`, N x 8 int32):
 * out[v, k] = arr[min(ptr[v] + min(k, max(deg(v) - 1, 0)), E - 1)] for one or two arrays in the slot order of `ptr`.
 * One launch per call (the torch formulation was a dozen small launches per loader batch / per inference scan).
 */
int spgnn_ell_rows(const int32_t* ptr, const int32_t* a0, const int32_t* a1 /* nullable */, int64_t N, int64_t E, int32_t* out0,
                   int32_t* out1 /* nullable with a1 */, spgnn_stream_t stream);
int spgnn_build_csc_count(const uint8_t* adj, const int64_t* adj_ptr, const int64_t* tree_ptr, int64_t num_trees,
                          int32_t* row_count, int32_t* col_count, spgnn_stream_t stream);
int spgnn_build_csc(const uint8_t* adj, const int64_t* adj_ptr, const int64_t* tree_ptr, int64_t num_trees,
                    const int64_t* row_start, const int64_t* col_start, int32_t* src, int32_t* dst, int32_t* indptr,
                    int32_t* indices, int32_t* eid, int32_t* out_indptr, int32_t* out_indices, int32_t* out_pos, int64_t N,
                    int64_t E, spgnn_stream_t stream);

/* =================================================================================================
 * Weight operands of every projection layer of a model, per training step, in one call (two kernel launches) instead of spgnn_weight_cat + spgnn_presplit per layer.  `table` (DEVICE memory, n_layers entries): per layer
 * the two row blocks a (rows_a x K) / b (rows_b x K, nullable) - fc.weight and res_fc.weight of a GATConv (reference
 * models.py:425-456) - and the outputs: dst = [a; b] with 16-byte rows (row stride dst_stride = K rounded up to 4, pad
 * columns zero), dst_t = its transpose (K rows, row stride dst_t_stride = rows rounded up to 4; nullable together with ps_t),
 * ps / ps_t = the same two matrices in the pre-split form spgnn_gemm_nt takes with b_presplit, scale = the operand's
 * power-of-two scale (one float).  first_block = running sum of spgnn_weight_prep_blocks over the earlier entries;
 * total_blocks = the sum over all; workspace: total_blocks floats of scratch (no initialisation needed).  Results are bit-identical
 * to the per-layer calls.  mode bit 1 (value 2): ps / ps_t in the wide-range form (SPGNN_GEMM_WIDE).  mode bit 0 clear: [a; b] as above;
 * set: the COLUMN concatenation [a | b] (rows_a rows, K = all columns,
 * rows_b = the columns `a` contributes; b nullable with rows_b = K) - the aggregate-first output layer's per-head operand
 * [W_fc,h | W_res,h] (spgnn_gat_agg_fwd); head h of the transpose is the column block [h D, (h+1) D) of dst_t / ps_t.
 * ================================================================================================= */
typedef struct spgnn_weight_prep_layer {
  const float* a; int64_t a_stride; const float* b; int64_t b_stride;
  float* dst; float* ps; int64_t dst_stride;
  float* dst_t; float* ps_t; int64_t dst_t_stride;
  float* scale;
  int64_t first_block;
  int32_t rows_a; int32_t rows_b; int32_t K; int32_t mode;
} spgnn_weight_prep_layer;

int64_t spgnn_weight_prep_blocks(int32_t rows, int64_t dst_stride, int64_t dst_t_stride);
int spgnn_weight_prep(const spgnn_weight_prep_layer* table, int32_t n_layers, int64_t total_blocks, float* workspace,
                      spgnn_stream_t stream);

/* =================================================================================================
 * The weight-space half of an output GATConv WITHOUT activation whose heads are averaged and whose classifier is folded
 * through the product (reference models.py:320-327 `self.gat_layers[-1](g, h).mean(1)` with the *Net's `gnn_out`,
 * models.py:921-933): with Zx = [z_0 | ... | z_{H-1} | x] (spgnn_gat_agg_fwd), mean_h out_h = Zx W_comb^T + b_mean and
 * logits = Zx P^T + c0.  One launch each way instead of ~28 tiny ones.
 *   fwd:  W_comb[d, h F + f] = W_fc[h D + d, f] / H,  W_comb[d, H F + f] = sum_h W_res[h D + d, f] / H  (w_res nullable: 0),
 *         b_mean[d] = sum_h bias[h D + d] / H (bias nullable: 0; b_mean nullable),  P = w_cls W_comb (J x Kc, J <= 32),
 *         c0 = w_cls b_mean + b_cls (b_cls nullable).  Kc = (H + 1) F; every image is written zero padded to Kp = Kc rounded up
 *         to 16 columns (strides >= Kp).  w_comb_bf16 (nullable, uint16 bf16 bits): the combined weight is rounded to bf16
 *         first, that image is written too, and w_comb / P hold the ROUNDED values (the function as the bf16 product evaluates
 *         it).  absmax_out (nullable): a scale block taking max |W_comb|.
 *   bwd:  from M1 = g_logits^T Zx (J x Kc) and cs = colsum(g_logits) (J):  g_W_fc, g_W_res (nullable) <- (w_cls^T M1) / H,
 *         g_bias (nullable) <- (w_cls^T cs) / H (every head's copy),  g_w_cls (nullable) = M1 W_comb^T + cs b_mean^T.
 * ================================================================================================= */
int spgnn_linear_mean_fold_fwd(const float* w_fc, int64_t w_fc_stride, const float* w_res, int64_t w_res_stride, const float* bias,
                               const float* w_cls, int64_t w_cls_stride, const float* b_cls, int32_t H, int32_t D, int32_t F, int32_t J,
                               float* w_comb, int64_t w_comb_stride, uint16_t* w_comb_bf16, int64_t w_comb_bf16_stride,
                               float* b_mean, float* P, int64_t P_stride, float* c0, float* absmax_out,
                               float* workspace /* spgnn_linear_mean_fold_workspace floats */,
                               uint32_t* tickets /* Kp / 32 + 1 words, zero before the first call; re-armed by the kernel */,
                               int32_t x_block /* 1: Kc = (H + 1) F as above; 0: Kc = H F, no x block (w_res must be null) - with H = 1
                                                  a plain Linear + classifier (GraphConv's linear output layer); with H = 1 and
                                                  x_block = 1, [W_fc | W_res] is SAGEConv's [fc_neigh | fc_self] on [neigh | h] */,
                               spgnn_stream_t stream);
int64_t spgnn_linear_mean_fold_workspace(int32_t H, int32_t F);
int spgnn_linear_mean_fold_bwd(const float* M1, int64_t M1_stride, const float* cs, const float* w_cls, int64_t w_cls_stride,
                               const float* w_comb, int64_t w_comb_stride, const float* b_mean, int32_t H, int32_t D, int32_t F,
                               int32_t J, float* g_w_fc, int64_t g_w_fc_stride, float* g_w_res, int64_t g_w_res_stride, float* g_bias,
                               float* g_w_cls, int64_t g_w_cls_stride, int32_t x_block, spgnn_stream_t stream);

/* Up to 24 (8 before ABI 53) of the deterministic split-K reductions above in ONE launch (a level's two weight gradients and two
 * attention-vector gradients come out of four spgnn_gemm_tn / spgnn_scores_bwd_w calls whose partial sums were four more launches;
 * a training step queues the reductions of its whole backward pass - nothing in it reads a weight gradient - and issues one).  `jobs`
 * is a HOST array; kind 0 = spgnn_sum_partials (n, out), 1 = spgnn_sum_partials_blockdiag (H, D, ld, out), 2 =
 * spgnn_sum_partials_compact (M, N, ld_in, out / out_stride, out2 / out2_stride / split_col, extra / extra_col); the fields
 * have the meaning of the same-named arguments there and results are bit-identical to the single calls. */
typedef struct spgnn_sum_job {
  int32_t kind; int32_t splits;
  const float* partials; int64_t split_stride;
  float* out; int64_t out_stride;
  int64_t n;
  int32_t H; int32_t D; int32_t ld; int32_t M; int32_t N; int32_t split_col;
  int64_t ld_in;
  float* out2; int64_t out2_stride;
  float* extra; int32_t extra_col; int32_t reserved;
} spgnn_sum_job;
int spgnn_sum_partials_multi(const spgnn_sum_job* jobs, int32_t n_jobs, spgnn_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif  /* SPGNN_HIP_H_ */
