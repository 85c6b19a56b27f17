"""Autograd operators over the C ABI (include/spgnn_hip.h).

Everything here runs on the GPU through libspgnn_hip.so; there is no CPU implementation
(CPU restatements live under oracle/ and are test infrastructure only).

``gat_layer``      X -> fused [fc | res_fc] projection GEMM + score GEMM (rocBLAS via torch.mm)
                   -> spgnn_gat_fwd; backward = spgnn_gat_bwd_dst + spgnn_gat_bwd_src writing
                   straight into the GEMM-gradient buffers, then two GEMMs.
``spmm_sum``       degree-normalised neighbour sum (GraphConv / GINConv).
``spmm_max``       neighbour max with arg slots (SAGEConv 'pool').
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import _capi
from .graph import DeviceCSC

ACT_NONE, ACT_ELU, ACT_TANH, ACT_RELU, ACT_LRELU = 0, 1, 2, 3, 4      # ACT_LRELU: slope 0.01 (nn.LeakyReLU default); GEMM / SpMM epilogues only

# Projection GEMMs: "f16x3" = hand-written split-fp16 MFMA kernels (spgnn_gemm.hip; fp32-GEMM accuracy), "fp32" =
# rocBLAS/hipBLASLt SGEMM through torch.mm (kept as the arithmetic cross-check of tests/test_hip_gemm.py).
GEMM_MODE = "f16x3"

# Optional device word (int64 tensor, one element) added to every attention-dropout seed at run time.  A step
# captured into a HIP graph bakes its host-drawn seeds in; incrementing this word inside the captured step
# (train.TrainStep does) gives every replay fresh masks.
DROPOUT_SEED_OFFSET: Optional["torch.Tensor"] = None


def _seed_off_ptr(device) -> int:
    t = DROPOUT_SEED_OFFSET
    return t.data_ptr() if t is not None and t.device == device else 0


def _s64(c: int) -> int:
    return c - (1 << 64) if c >= (1 << 63) else c


def attn_dropout_multiplier(E: int, H: int, p: float, seed: int, device) -> torch.Tensor:
    """(E, H) multiplier of the attention dropout the GAT kernels apply: 1/(1-p) where CSC slot e, head h is kept, 0 where
    it is dropped - the kernels' counter hash keep_scale(seed, e*H + h) restated in int64 tensor arithmetic (wrap-around
    multiplies, logical shifts), for ``get_attention=True`` in training mode (DGL returns attn_drop(edge_softmax(e)))."""
    def lsr(z, k):
        return (z >> k) & ((1 << (64 - k)) - 1)
    idx = torch.arange(E * H, dtype=torch.int64, device=device) + 1
    z = idx * _s64(0x9E3779B97F4A7C15) + _s64(seed & ((1 << 64) - 1))
    if DROPOUT_SEED_OFFSET is not None and DROPOUT_SEED_OFFSET.device == idx.device:
        z = z + DROPOUT_SEED_OFFSET.view(torch.int64)[0]
    z = (z ^ lsr(z, 30)) * _s64(0xBF58476D1CE4E5B9)
    z = (z ^ lsr(z, 27)) * _s64(0x94D049BB133111EB)
    z = z ^ lsr(z, 31)
    u = lsr(z, 40).to(torch.float32) * (1.0 / 16777216.0)
    keep = torch.tensor(1.0, dtype=torch.float32) / (torch.tensor(1.0, dtype=torch.float32) - torch.tensor(p, dtype=torch.float32))
    return torch.where(u >= p, keep.to(device), torch.zeros((), device=device)).view(E, H)


class KernelTimer:
    """Optional per-launch HIP-event timing of the message-passing kernels (bench.py's roofline leg).
    Events are recorded on the stream the kernel is launched on (torch's current stream)."""
    enabled = False
    records: dict = {}
    only = None          # when set: record these (kernel, shape) keys only (a few events per step: no perturbation)
    sequence: list = []  # the recorded keys in launch order (tools/pmc_step.py matches profiler rows to keys with it)

    @classmethod
    def start(cls, only=None):
        if only is not None and only and not isinstance(next(iter(only)), tuple):
            only = (tuple(only),)                      # a single key
        cls.records, cls.enabled, cls.only = {}, True, (None if only is None else frozenset(only))
        cls.sequence = []

    @classmethod
    def stop(cls):
        """-> {(kernel, shape): [ms, ...]} after synchronising."""
        cls.enabled = False
        torch.cuda.synchronize()
        out = {k: [a.elapsed_time(b) for a, b in v] for k, v in cls.records.items()}
        cls.records = {}
        return out


class _timed:
    def __init__(self, name, shape):
        self.key = (name,) + tuple(shape)

    def __enter__(self):
        self.on = KernelTimer.enabled and (KernelTimer.only is None or self.key in KernelTimer.only)
        if self.on:
            self.a = torch.cuda.Event(enable_timing=True)
            self.a.record()
        return self

    def __exit__(self, *exc):
        if self.on:
            b = torch.cuda.Event(enable_timing=True)
            b.record()
            KernelTimer.records.setdefault(self.key, []).append((self.a, b))
            KernelTimer.sequence.append(self.key)
        return False


def _stream(t: torch.Tensor) -> int:
    return torch.cuda.current_stream(t.device).cuda_stream


def _require_cuda(*tensors: Optional[torch.Tensor]) -> None:
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("spgnn_amd ops run only on a ROCm device (no CPU fallback); got a CPU tensor. "
                               "Move the graph and features to 'cuda'.")


def _rowmajor(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.float32:
        t = t.float()
    if t.dim() != 2 or t.stride(1) != 1 or t.stride(0) < t.shape[1]:
        t = t.contiguous()
    return t


def _ptr(t: Optional[torch.Tensor]) -> int:
    return 0 if t is None else t.data_ptr()


USE_ELL = True     # hand the GAT kernels the padded (N, 8) neighbour rows (graph.DeviceCSC.ell); False: CSC walk only (A/B, tests)


def _ell(csc):
    """(nbr8, out_nbr8, out_pos8) device pointers, zeros when the rows are not used."""
    if not USE_ELL or csc.num_nodes == 0:
        return 0, 0, 0
    a, b, c = csc.ell()
    return a.data_ptr(), b.data_ptr(), c.data_ptr()


# --------------------------------------------------------------------------------------------
# tree-resident LDS tiles (csrc/spgnn_tile.hip)
# --------------------------------------------------------------------------------------------
TILE_KERNELS = True      # tree-resident LDS tiles where they are measured faster (tile_plan); False: the row kernels everywhere
TILE_FORCE = False       # tests / tools: every traversal the tile kernels support runs on them, whatever the table says
TILE_NODES = 192         # most nodes of a tile (the kernels' compile-time capacity): whole trees of up to 192 nodes stay closed
TILE_MIN_NODES = 32768   # below this a launch is one short round of workgroups either way and the row kernels' 64-node blocks win
# Where the tiles win (tools/tile_ab.py, MI355X, 512 trees, one process; profiles/r05_tile_ab_*.json).  Only the SOURCE-MAJOR
# half - the traversal whose every gather is a 4-byte word by CSC slot besides the rows - and mostly on bf16 rows: 2 x 256
# 43.6 -> 38.5 us, 2 x 128 25.3 -> 23.0, 2 x 64 17.2 -> 16.3; fp32 1 x 64 13.2 -> 11.1.  The forward and the destination-major
# half LOSE 10-35 % at every shape (their time is the node's own rows streaming in and out plus per-node VALU work - softmax,
# hashes, activation - which staging the neighbours in LDS does not touch), and at 64 trees everything loses (64 tiles).
TILE_TABLE = {("src", 2): lambda H, D: True, ("src", 4): lambda H, D: H * D == 64}


def tile_plan(csc: DeviceCSC, H: int, D: int, elem_bytes: int, kind: str = "src"):
    """(tile_ptr, n_tiles, cap) when traversal ``kind`` ("fwd" / "dst" / "src") of an (H, D) GATConv layer on ``csc`` runs as
    LDS tiles (csrc/spgnn_tile.hip), else None: every node needs 1..8 edges in both directions and the padded neighbour rows
    (the tile kernels have no general-degree path), a head must fit the kernels' LDS (spgnn_gat_tile_supported), and the
    shape must be one where the tiles are measured faster than the row kernels (TILE_TABLE) unless TILE_FORCE."""
    if not (TILE_KERNELS or TILE_FORCE) or not USE_ELL or csc.num_nodes == 0 or getattr(csc, "num_dst", None) is not None:
        return None
    if not TILE_FORCE:
        use = TILE_TABLE.get((kind, elem_bytes))
        if use is None or csc.num_nodes < TILE_MIN_NODES or not use(H, D):
            return None
    if not (1 <= csc.min_in_degree and csc.max_in_degree <= 8 and 1 <= int(getattr(csc, "min_out_degree", 0) or 0) and csc.max_out_degree <= 8):
        return None
    if not _capi.load().spgnn_gat_tile_supported(H, D, elem_bytes, TILE_NODES):
        return None
    t, n = csc.tiles(TILE_NODES)
    return t, n, TILE_NODES


# --------------------------------------------------------------------------------------------
# raw launches (thin, shape-checked wrappers; used by the autograd functions and by tests)
# --------------------------------------------------------------------------------------------
def can_fuse_mean(H: int, D: int) -> bool:
    return bool(_capi.load().spgnn_gat_can_fuse_mean(H, D))


def gat_fwd_raw(csc: DeviceCSC, ft, el, er, res, bias, H: int, D: int, slope: float, act: int, p_drop: float = 0.0,
                seed: int = 0, out: Optional[torch.Tensor] = None, mean: bool = False, need_out: bool = True,
                out_drop=None, out_absmax: Optional[torch.Tensor] = None):
    """-> (out (N,H*D) or None, out_mean (N,D) or None, attn (E,H)).  ``mean``: also produce the head mean
    (fused into the epilogue when the geometry allows); ``need_out=False`` lets the per-head output be
    skipped when only the mean is consumed and nothing in the backward needs it.
    ``out_drop`` = (p, seed, total, offset): store the consumer's feature dropout of the output (``out`` = the column
    block [offset, offset + H*D) of the consumer's (N, total) input buffer); ``out_absmax``: a scale block
    (new_scale_block) that takes the maxima of the stored rows."""
    _require_cuda(ft, el, er, res, bias)
    N, E = csc.num_nodes, csc.num_edges
    assert ft.shape[0] == N and ft.shape[1] == H * D and ft.stride(1) == 1
    assert el.shape == (N, H) and er.shape == (N, H) and el.stride(0) == er.stride(0) and el.stride(1) == 1 == er.stride(1)
    if res is not None:
        assert res.shape == ft.shape and res.stride(1) == 1
    if bias is not None:
        assert bias.numel() == H * D and bias.is_contiguous()
    skip_out = mean and not need_out and out is None and can_fuse_mean(H, D)
    if out is None and not skip_out:
        out = torch.empty((N, H * D), dtype=torch.float32, device=ft.device)
    out_mean = torch.empty((N, D), dtype=torch.float32, device=ft.device) if mean else None
    attn = torch.empty((E, H), dtype=torch.float32, device=ft.device)
    lib = _capi.load()
    plan = None if mean else tile_plan(csc, H, D, 4, "fwd")
    if plan is not None:
        with torch.cuda.device(ft.device), _timed("gat_fwd", (N, E, H, D, int(res is not None), 0, 1)):
            _capi.check(lib.spgnn_gat_fwd_tile(plan[0].data_ptr(), plan[1], plan[2], csc.indptr.data_ptr(), _ell(csc)[0], ft.data_ptr(),
                                               ft.stride(0), el.data_ptr(), er.data_ptr(), el.stride(0), _ptr(res),
                                               res.stride(0) if res is not None else 0, _ptr(bias), out.data_ptr(), out.stride(0),
                                               attn.data_ptr(), _ptr(out_absmax), N, H, D, slope, act, p_drop, seed,
                                               _seed_off_ptr(ft.device), *(out_drop or (0.0, 0, 0, 0)), _stream(ft)), "spgnn_gat_fwd_tile")
        return out, out_mean, attn
    with torch.cuda.device(ft.device), _timed("gat_fwd", (N, E, H, D, int(res is not None), int(mean), int(out is not None))):
        _capi.check(lib.spgnn_gat_fwd(csc.indptr.data_ptr(), csc.indices.data_ptr(), _ell(csc)[0], ft.data_ptr(), ft.stride(0),
                                      el.data_ptr(), er.data_ptr(), el.stride(0), _ptr(res),
                                      res.stride(0) if res is not None else 0, _ptr(bias), _ptr(out),
                                      out.stride(0) if out is not None else 0, _ptr(out_mean),
                                      out_mean.stride(0) if mean else 0, attn.data_ptr(), N, E, H, D, slope, act,
                                      p_drop, seed, _seed_off_ptr(ft.device), *(out_drop or (0.0, 0, 0, 0)), _ptr(out_absmax),
                                      _stream(ft)), "spgnn_gat_fwd")
    return out, out_mean, attn


def gat_bwd_raw(csc: DeviceCSC, ft, el, er, attn, g_out, out, H: int, D: int, slope: float, act: int,
                p_drop: float, seed: int, g_pre: torch.Tensor, g_ft: torch.Tensor, g_el: torch.Tensor,
                g_er: torch.Tensor, mean: bool = False, absmax: Optional[torch.Tensor] = None,
                score_l: Optional[torch.Tensor] = None, score_r: Optional[torch.Tensor] = None, out_drop=None,
                absmax_dst: bool = True) -> torch.Tensor:
    """Runs both backward halves. g_pre/g_ft (N,H*D), g_el/g_er (N,H) are written in place
    (may be strided views). ``mean``: g_out is the (N,D) gradient of the head mean.  ``absmax``: scale block taking the
    maxima of |g_pre| and |g_ft| (the split-GEMM scale of [g_ft | g_pre]); ``absmax_src=False``: of |g_pre| only.
    Returns g_e (E,H) in CSC slot order."""
    _require_cuda(ft, g_out)
    N, E = csc.num_nodes, csc.num_edges
    assert g_out.shape == (N, D if mean else H * D) and g_out.stride(1) == 1
    g_e = torch.empty((E, H), dtype=torch.float32, device=ft.device)
    assert g_el.stride(0) == g_er.stride(0)
    lib = _capi.load()
    plan_d = None if mean else tile_plan(csc, H, D, 4, "dst")
    plan_s = tile_plan(csc, H, D, 4, "src")
    with torch.cuda.device(ft.device):
        st = _stream(ft)
        ell = _ell(csc)
        with _timed("gat_bwd_dst", (N, E, H, D, act, int(mean))):
            if plan_d is not None:
                _capi.check(lib.spgnn_gat_bwd_dst_tile(plan_d[0].data_ptr(), plan_d[1], plan_d[2], csc.indptr.data_ptr(), ell[0], ft.data_ptr(),
                                                       ft.stride(0), el.data_ptr(), er.data_ptr(), el.stride(0), attn.data_ptr(),
                                                       g_out.data_ptr(), g_out.stride(0), _ptr(out), out.stride(0) if out is not None else 0,
                                                       g_pre.data_ptr(), g_pre.stride(0), g_e.data_ptr(), g_er.data_ptr(), g_er.stride(0),
                                                       _ptr(absmax) if absmax_dst else 0, N, H, D, slope, act, p_drop, seed,
                                                       _seed_off_ptr(ft.device), *(out_drop or (0.0, 0, 0, 0)), st), "spgnn_gat_bwd_dst_tile")
            else:
                _capi.check(lib.spgnn_gat_bwd_dst(csc.indptr.data_ptr(), csc.indices.data_ptr(), ell[0], ft.data_ptr(), ft.stride(0),
                                                  el.data_ptr(), er.data_ptr(), el.stride(0), attn.data_ptr(),
                                                  g_out.data_ptr(), g_out.stride(0), int(mean), _ptr(out),
                                                  out.stride(0) if out is not None else 0, g_pre.data_ptr(), g_pre.stride(0),
                                                  g_e.data_ptr(), g_er.data_ptr(), g_er.stride(0), _ptr(absmax) if absmax_dst else 0, N, E, H, D, slope,
                                                  act, p_drop, seed, _seed_off_ptr(ft.device), *(out_drop or (0.0, 0, 0, 0)), st),
                            "spgnn_gat_bwd_dst")
        with _timed("gat_bwd_src", (N, E, H, D)):
            if plan_s is not None:
                _capi.check(lib.spgnn_gat_bwd_src_tile(plan_s[0].data_ptr(), plan_s[1], plan_s[2], csc.indptr.data_ptr(), csc.out_indptr.data_ptr(),
                                                       ell[1], ell[2], attn.data_ptr(), g_e.data_ptr(), g_pre.data_ptr(), g_pre.stride(0),
                                                       g_ft.data_ptr(), g_ft.stride(0), g_el.data_ptr(), g_el.stride(0), _ptr(absmax),
                                                       _ptr(score_l), _ptr(score_r), g_er.data_ptr() if score_l is not None else 0, N, H, D,
                                                       p_drop, seed, _seed_off_ptr(ft.device), st), "spgnn_gat_bwd_src_tile")
            else:
                _capi.check(lib.spgnn_gat_bwd_src(csc.out_indptr.data_ptr(), csc.out_indices.data_ptr(),
                                                  csc.out_pos.data_ptr(), ell[1], ell[2], attn.data_ptr(), g_e.data_ptr(), g_pre.data_ptr(),
                                                  g_pre.stride(0), g_ft.data_ptr(), g_ft.stride(0), g_el.data_ptr(),
                                                  g_el.stride(0), _ptr(absmax), _ptr(score_l),
                                                  _ptr(score_r), g_er.data_ptr() if score_l is not None else 0, N, E, H, D,
                                                  p_drop, seed, _seed_off_ptr(ft.device), st), "spgnn_gat_bwd_src")
    return g_e


# --------------------------------------------------------------------------------------------
# GAT layer
# --------------------------------------------------------------------------------------------
def _dw_gemm(g: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """g^T @ x for tall operands (N x C)^T (N x K): the output is small and the reduction dimension is the
    node count, so a plain GEMM fills only C*K/tile^2 workgroups.  Split the node dimension into chunks
    (batched GEMM) and sum the partial products: 1.2-1.5x faster than one rocBLAS call on MI355X."""
    N, C = g.shape
    K = x.shape[1]
    tiles = ((C + 127) // 128) * ((K + 127) // 128)
    splits = 1
    while splits < 32 and tiles * splits < 512 and N // (splits * 2) >= 1024:
        splits *= 2
    if splits == 1:
        return torch.mm(g.t(), x)
    n_main = N // splits * splits
    part = torch.bmm(g[:n_main].view(splits, n_main // splits, C).transpose(1, 2),
                     x[:n_main].view(splits, n_main // splits, K)).sum(0)
    if n_main < N:
        part.addmm_(g[n_main:].t(), x[n_main:])
    return part


def _rows_aligned(t: torch.Tensor) -> bool:
    return t.stride(1) == 1 and t.stride(0) % 4 == 0 and t.data_ptr() % 16 == 0


def _pad16(k: int) -> int:
    return (k + 15) // 16 * 16


_SCALE_WS: dict = {}     # per (device, stream): two zeroed words the multi-block reduction resets after each use


_TN_MIN_ELEMS = 16384   # weight gradients smaller than this go to rocBLAS (measured: 65536 -> 16384 moves the position stream's 512 x 39 and 128 x 128 gradients to the split-K kernel with the bias column sums riding along: 6.73 -> 6.67 ms/step)


TN_PAIR_SPLITS = True    # a level's two weight-gradient products choose their split count together (tn_pair_splits)
DEFER_STEP_SUMS = True   # inside a training step, every split-K reduction of the backward pass waits for ONE launch after it
DEBUG_POISON_DEFERRED = False   # tests: deferred outputs start as NaN, so a read before the flush cannot pass unnoticed
STEP_SUMS: Optional["StepSums"] = None          # installed by train.TrainStep while it issues a step's forward and backward


OVERLAP_TN = True        # a training step issues its weight-gradient products on a side stream (SideLaunch): they overlap the
TN_SIDE: Optional["SideLaunch"] = None      # traversals that follow; installed by train.TrainStep around forward + backward
OVERLAP_TN_MIN_ROWS = 32768    # ... for batches of at least this many nodes.  One process, alternating (tools/step_toggle_ab.py, MI355X,
                               # round 5): st_pgat_spgnn_3 4.990 / 4.980 vs 5.058 / 5.037 ms at 512 trees (-1.2 %), st_gat_3 2.880 vs 2.903;
                               # at 64 trees 1.027 vs 1.003 (the fork and join cost more than the short products hide): off there


def side_for(rows: int, device) -> Optional["SideLaunch"]:
    """The running step's side stream for a weight-gradient product over ``rows`` node rows, or None."""
    q = TN_SIDE
    return q if (q is not None and OVERLAP_TN and rows >= OVERLAP_TN_MIN_ROWS and q.device == torch.device(device)) else None


class SideLaunch:
    """Weight-gradient products (spgnn_gemm_tn / _pair) of a training step on a SIDE stream.  Nothing inside a backward pass
    reads a weight gradient, while the input-gradient product and the next level's traversals are on the critical path: an
    MFMA-bound product and the HBM-bound traversals use different parts of the chip, and issued on two streams they overlap
    (tools/overlap_probe.py, MI355X: a 558 us weight-gradient product next to 864 us of GAT traversals: 1 283 us against 1 389
    one after the other).  Under capture the side stream becomes a parallel branch of the step's HIP graph.  Tensors the side
    launches read are kept alive here until ``join()``: the caching allocator may otherwise hand their memory to a later
    allocation of the main stream while the product still reads it."""

    def __init__(self, device):
        self.device = torch.device(device)
        self.side = torch.cuda.Stream(device=self.device)
        self.keep, self.pending = [], False

    def run(self, fn, *keep):
        main = torch.cuda.current_stream(self.device)
        self.side.wait_stream(main)                  # everything issued so far (the operands' producers) comes first
        with torch.cuda.stream(self.side):
            fn()
        self.keep.extend(keep)
        self.pending = True

    def join(self):
        if self.pending:
            torch.cuda.current_stream(self.device).wait_stream(self.side)
            self.pending = False
        self.keep.clear()


class SumJobs:
    """Deferred split-K reductions: gemm_tn / scores_bwd_w append their partial-sum step here instead of launching it, and
    ``flush()`` runs up to ``MAX`` of them in ONE launch (spgnn_sum_partials_multi; bit-identical to the single launches).
    The outputs the producers returned are filled by the flush.  While a training step has installed its own queue
    (``STEP_SUMS``) the jobs go there instead and ``flush()`` does nothing: nothing inside a backward pass reads a weight
    gradient, so the whole step's reductions run as one launch after it (five launches of 11-18 us before)."""

    MAX = 24                                     # spgnn_sum_partials_multi's limit (kMaxSumJobs)

    def __init__(self, device, local: bool = False):
        """``local``: these outputs are read again inside the backward pass - they feed another autograd node that computes
        with them (the folded score weights' gradient, which fold_scores' backward consumes), or they are one of TWO gradients of
        a parameter, which autograd adds on arrival: never handed to the step."""
        self.device, self.jobs, self.keep = device, [], []
        q = STEP_SUMS
        self.step = q if (q is not None and not local and DEFER_STEP_SUMS and q.device == torch.device(device)) else None

    def add(self, job, *tensors):
        if self.step is not None:
            self.step.add(job, *tensors)
            return
        self.jobs.append(job)
        self.keep.extend(tensors)
        if len(self.jobs) == self.MAX:
            self.flush()

    def flush(self):
        if not self.jobs:
            return
        if TN_SIDE is not None:
            TN_SIDE.join()                           # these sums read partials a side-stream product may still be writing
        arr = (_capi.SumJob * len(self.jobs))(*self.jobs)
        with torch.cuda.device(self.device):
            _capi.check(_capi.load().spgnn_sum_partials_multi(arr, len(self.jobs), torch.cuda.current_stream(self.device).cuda_stream),
                        "spgnn_sum_partials_multi")
        self.jobs, self.keep = [], []


class StepSums:
    """The reductions of one training step's backward pass (see SumJobs).  The partials and outputs are held through
    ``detach()`` aliases: they keep the storage alive without adding a reference to the tensors themselves - autograd takes a
    gradient over as it is only while it holds the last reference to it, and clones it otherwise, which here would copy an
    output the flush has not filled yet."""

    def __init__(self, device):
        self.device, self.jobs, self.keep, self.outs = torch.device(device), [], [], []

    def add(self, job, part, *outs):
        """``part``: the split partials; ``outs``: what the reduction will fill (weight gradient [, its second column range,
        its bias column sums])."""
        self.jobs.append(job)
        self.keep.append(part.detach())
        for t in outs:
            if t is None:
                continue
            d = t.detach()
            if DEBUG_POISON_DEFERRED:        # tests: an early read of an unfilled output must not pass unnoticed
                d.fill_(float("nan"))
            self.keep.append(d)
            self.outs.append(d)

    def check_taken_over(self, params) -> None:
        """Call after ``backward()`` and BEFORE ``flush()``: every deferred output must by now BE (the storage of) some
        parameter's ``.grad`` - autograd took the tensor over as it was.  Where it cloned or added on arrival instead (a
        parameter used by two layers or twice by one, a tensor hook, ``retain_grad``, a non-leaf parameter) it copied memory
        this queue has not filled yet, and the step's gradients would be silently wrong: raise."""
        if not self.outs:
            return
        owned = {p.grad.untyped_storage().data_ptr() for p in params if p.grad is not None}
        lost = [o for o in self.outs if o.untyped_storage().data_ptr() not in owned]
        if lost:
            raise RuntimeError(
                f"{len(lost)} of {len(self.outs)} deferred split-K gradient sums were copied by autograd before they were filled "
                "(a parameter shared by two layers or used twice, a gradient hook, retain_grad or a non-leaf parameter): their "
                "values are not valid.  Set spgnn_amd.ops.DEFER_STEP_SUMS = False (and DEFER_ATTN_GRADS = False) for this model.")

    def flush(self):
        jobs, keep = self.jobs, self.keep
        self.jobs, self.keep, self.outs = [], [], []
        for i0 in range(0, len(jobs), SumJobs.MAX):
            chunk = jobs[i0:i0 + SumJobs.MAX]
            arr = (_capi.SumJob * len(chunk))(*chunk)
            with torch.cuda.device(self.device):
                _capi.check(_capi.load().spgnn_sum_partials_multi(arr, len(chunk), torch.cuda.current_stream(self.device).cuda_stream),
                            "spgnn_sum_partials_multi")
        del keep


def sum_partials(part: torch.Tensor) -> torch.Tensor:
    """part (S, ...) contiguous -> part.sum(0) in a fixed order, one launch that fills the chip for any S
    (torch's reduction took 16-47 us on the thousands of small score-gradient partials)."""
    S = part.shape[0]
    n = part[0].numel()
    if S == 1:
        return part[0]
    if n % 4 or not part.is_contiguous() or part.data_ptr() % 16:
        return part.sum(0)
    out = torch.empty(part.shape[1:], dtype=torch.float32, device=part.device)
    with torch.cuda.device(part.device):
        _capi.check(_capi.load().spgnn_sum_partials(part.data_ptr(), n, S, n, out.data_ptr(), _stream(part)), "spgnn_sum_partials")
    return out


SCALE_SLOTS, SCALE_HEADER = 256, 4          # spgnn_internal.h: {-256, 0, 0, 0, m_1 ... m_256}
_SCALE_TEMPLATE: dict = {}


def _scale_template(device, rows: int) -> torch.Tensor:
    key = (str(device), rows)
    t = _SCALE_TEMPLATE.get(key)
    if t is None:
        t = torch.zeros((rows, SCALE_HEADER + SCALE_SLOTS), dtype=torch.float32, device=device)
        t[:, 0] = -float(SCALE_SLOTS)
        _SCALE_TEMPLATE[key] = t
    return t


class ScalePool:
    """The scale blocks of one training step (include/spgnn_hip.h, spgnn_gemm_nt): ``begin()`` re-initialises every block
    with ONE copy launch and hands them out in request order, so a step that is captured into a HIP graph always finds the
    same addresses; requests beyond the capacity, or outside ``begin() ... end()``, get a block of their own (one clone)."""

    def __init__(self, device, capacity: int = 128):
        self.device = torch.device(device)
        self.capacity = capacity
        self.buf = _scale_template(self.device, capacity).clone()
        self.cursor, self.active = 0, False
        # range monitor (include/spgnn_hip.h, spgnn_step_begin): operands of this pool's steps that had whole rows / blocks more
        # than 2^18 below their maximum - outside the envelope in which the split GEMMs are fp32-accurate - counted on the device
        self.violations = torch.zeros(1, dtype=torch.int32, device=self.device)

    def begin(self, counter: Optional[torch.Tensor] = None):
        """Re-arm every block; ``counter`` (device int64 scalar, optional) advances by one in the same launch
        (spgnn_step_begin): the step's dropout / mask stream position."""
        with torch.cuda.device(self.device):
            _capi.check(_capi.load().spgnn_step_begin(_ptr(counter), self.buf.data_ptr(), self.capacity, self.violations.data_ptr(),
                                                      _stream(self.buf)), "spgnn_step_begin")
        self.cursor, self.active = 0, True

    def end(self):
        self.active = False

    def take(self) -> Optional[torch.Tensor]:
        if not self.active or self.cursor >= self.capacity:
            return None
        b = self.buf[self.cursor]
        self.cursor += 1
        return b


_SCALE_POOLS: dict = {}


def _dev_key(device) -> str:
    """'cuda' and 'cuda:<current>' are one device: one pool."""
    d = torch.device(device)
    if d.type == "cuda" and d.index is None:
        d = torch.device("cuda", torch.cuda.current_device())
    return str(d)


def scale_pool(device) -> ScalePool:
    key = _dev_key(device)
    p = _SCALE_POOLS.get(key)
    if p is None:
        p = _SCALE_POOLS[key] = ScalePool(device)
    return p


def new_scale_block(device) -> torch.Tensor:
    """A zeroed scale block for the producers of one GEMM operand to fold their maxima into; the tensor itself is then
    passed wherever a scale is taken (gemm_nt / gemm_tn ``scale_*``, ``_spgnn_scale``)."""
    p = _SCALE_POOLS.get(_dev_key(device))
    b = p.take() if p is not None else None
    return b if b is not None else _scale_template(device, 1)[0].clone()


def range_violations(device) -> int:
    """Operands seen so far by the training steps on ``device`` that left the split GEMMs' accuracy envelope (synchronises:
    poll it per epoch, not per step).  The flags of the step in flight are counted when the NEXT step begins."""
    p = _SCALE_POOLS.get(_dev_key(device))
    return int(p.violations.item()) if p is not None else 0


def range_flag(scale: torch.Tensor) -> bool:
    """Whether a product that consumed scale block ``scale`` flagged it (header word 1; tests, eager use)."""
    return scale.numel() > 1 and float(scale[1]) != 0.0


def scale_value(scale: torch.Tensor) -> float:
    """Host value of a scale operand (tests, diagnostics): the scalar itself or 2^(14 - e) of a block's largest slot."""
    import math
    if scale.numel() == 1:
        return float(scale)
    m = float(scale[SCALE_HEADER:].max())
    return 1.0 if not (m > 0.0 and math.isfinite(m)) else math.ldexp(1.0, 14 - math.frexp(m)[1])


def scale_from_partials(partials: torch.Tensor, factor: float = 1.0) -> torch.Tensor:
    """Device scalar 2^(14 - e), factor * max(partials) <= 2^e (see pow2_scale)."""
    scale = torch.empty(1, dtype=torch.float32, device=partials.device)
    key = (partials.device, _stream(partials))
    ws = _SCALE_WS.get(key)
    if ws is None:
        ws = _SCALE_WS[key] = torch.zeros(4, dtype=torch.int32, device=partials.device)
    with torch.cuda.device(partials.device):
        _capi.check(_capi.load().spgnn_scale_from_partials(partials.data_ptr(), partials.numel(), factor, scale.data_ptr(),
                                                           ws.data_ptr(), _stream(partials)), "spgnn_scale_from_partials")
    return scale


class _FoldScoresFn(torch.autograd.Function):
    """(fc.weight (H*D, K), attn_l (H, D), attn_r (H, D)) -> w_lr (2H, K): a view of a (2H, pad16(K)) zero-padded
    buffer, the form the score kernels take.  Two kernels (forward, backward) per layer and step."""

    @staticmethod
    def forward(ctx, w, attn_l, attn_r):
        ctx.attn_shape = attn_l.shape           # (H, D) or the parameter's own (1, H, D): no select / zeros / copy autograd nodes
        H, D = attn_l.shape[-2:]
        K = w.shape[1]
        Kp = _pad16(K)
        al, ar = attn_l.reshape(H, D).contiguous(), attn_r.reshape(H, D).contiguous()
        buf = torch.empty((2 * H, Kp), dtype=torch.float32, device=w.device)
        with torch.cuda.device(w.device):
            _capi.check(_capi.load().spgnn_fold_scores_fwd(w.data_ptr(), w.stride(0), al.data_ptr(), ar.data_ptr(),
                                                           buf.data_ptr(), Kp, H, D, K, _stream(w)), "spgnn_fold_scores_fwd")
        ctx.save_for_backward(w, al, ar)
        return buf[:, :K]

    @staticmethod
    def backward(ctx, g):
        w, al, ar = ctx.saved_tensors
        H, D = al.shape
        K = w.shape[1]
        if not (g.stride(1) == 1 and g.stride(0) >= K):
            g = g.contiguous()
        g_w = torch.empty_like(w)
        g_al, g_ar = torch.empty_like(al), torch.empty_like(ar)
        with torch.cuda.device(w.device):
            _capi.check(_capi.load().spgnn_fold_scores_bwd(w.data_ptr(), w.stride(0), al.data_ptr(), ar.data_ptr(), g.data_ptr(),
                                                           g.stride(0), g_w.data_ptr(), g_w.stride(0), g_al.data_ptr(),
                                                           g_ar.data_ptr(), H, D, K, _stream(w)), "spgnn_fold_scores_bwd")
        return g_w, g_al.view(ctx.attn_shape), g_ar.view(ctx.attn_shape)


# Inference with FROZEN weights (spgnn_amd.infer.ForwardRunner): everything a forward pass derives from the parameters alone -
# the projection operands of spgnn_weight_prep, the folded score vectors - is computed once when the forward is captured and
# is NOT part of the captured graph: a replay starts at the first kernel that touches node data.  A dict while a ForwardRunner
# issues the model's forward (its entries live as long as the runner's capture), else None.  The caller owns the contract:
# parameters must not change between capture and replay (ForwardRunner.reset() after loading other weights).
FROZEN_WEIGHTS: Optional[dict] = None


def fold_scores(w_fc: torch.Tensor, attn_l: torch.Tensor, attn_r: torch.Tensor) -> torch.Tensor:
    _require_cuda(w_fc, attn_l, attn_r)
    assert w_fc.stride(1) == 1
    fz = FROZEN_WEIGHTS
    if fz is not None and INFERENCE:
        key = ("fold", id(w_fc), id(attn_l), id(attn_r))
        hit = fz.get(key)
        if hit is None and not torch.cuda.is_current_stream_capturing():
            hit = fz[key] = _FoldScoresFn.apply(w_fc, attn_l, attn_r)
        if hit is not None:
            return hit
    return _FoldScoresFn.apply(w_fc, attn_l, attn_r)


def _padded_rows(w: torch.Tensor, Kp: int) -> torch.Tensor:
    """(J, Kp) zero-padded contiguous copy of w (J, K) - or w's own buffer when it already is one (fold_scores)."""
    if w.shape[1] == Kp and w.is_contiguous():
        return w
    if w.stride(1) == 1 and w.stride(0) == Kp and w.storage_offset() == 0 and getattr(w, "_base", None) is not None \
            and w._base.shape == (w.shape[0], Kp):
        return w._base
    return torch.nn.functional.pad(w, (0, Kp - w.shape[1])).contiguous()


def scores_fwd(x: torch.Tensor, w_lr: torch.Tensor, want_scale: bool = False, bias: Optional[torch.Tensor] = None):
    """S = x @ w_lr^T (N, J), J = 2H: the MFMA streaming kernel when the rows of x are 16-byte aligned and
    J <= 16, rocBLAS otherwise.  ``want_scale``: also return the split-GEMM scale of x — the kernel reads every
    element of x anyway, so its absmax costs nothing extra."""
    N, K = x.shape
    J = w_lr.shape[0]
    if not (_rows_aligned(x) and J <= 32) or N == 0:
        s = torch.mm(x, w_lr.t())
        if bias is not None:
            s += bias
        return (s, pow2_scale(x) if N > 0 else None) if want_scale else s
    Kp = _pad16(K)
    w_p = _padded_rows(w_lr, Kp)
    s = torch.empty((N, J), dtype=torch.float32, device=x.device)
    part = new_scale_block(x.device) if want_scale else None
    bias_c = None if bias is None else bias.detach().contiguous()          # (J,) added by the kernel: no launch of its own
    with torch.cuda.device(x.device), _timed("scores_fwd", (N, K, J)):
        _capi.check(_capi.load().spgnn_scores_fwd(x.data_ptr(), x.stride(0), w_p.data_ptr(), Kp, s.data_ptr(), s.stride(0),
                                                  _ptr(part), _ptr(bias_c), N, K, J, _stream(x)), "spgnn_scores_fwd")
    return (s, part) if want_scale else s


_SCORES_SPLIT_WAVES = 2048   # row ranges x column groups of scores_bwd_w: 2048 measured best (7.38 vs 7.46 ms/step at 4096: half the partials; 1024: 7.43, 512: 7.65)


_SCORES_SPLIT_WAVES_SMALL = 1024   # the same for J <= 8 (attention-vector gradients)


def scores_bwd_w(g_s: torch.Tensor, x: torch.Tensor, blockdiag_heads: int = 0, defer: Optional["SumJobs"] = None) -> torch.Tensor:
    """g_w_lr = g_s^T @ x (J, K).  ``blockdiag_heads`` = H (J = 2H, K = H*D): only the (2, H, D) block diagonal
    [w, h, :] = (g_s[:, w*H + h]^T @ x)[h*D:(h+1)*D] - the attention vectors' gradients."""
    N, K = x.shape
    J = g_s.shape[1]
    if not (_rows_aligned(x) and J <= 32) or N == 0:
        m = torch.mm(g_s.t(), x)
        if blockdiag_heads:
            H = blockdiag_heads; D = K // H
            return torch.stack([torch.stack([m[w * H + h, h * D:(h + 1) * D] for h in range(H)]) for w in range(2)])
        return m
    Kp = _pad16(K)
    waves = _SCORES_SPLIT_WAVES if J > 8 else _SCORES_SPLIT_WAVES_SMALL
    splits = max(1, min(waves // ((K + 255) // 256), N // 16))
    part = torch.empty((splits, J, Kp), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device), _timed("scores_bwd_w", (N, K, J)):
        _capi.check(_capi.load().spgnn_scores_bwd_w(g_s.data_ptr(), g_s.stride(0), x.data_ptr(), x.stride(0),
                                                    part.data_ptr(), splits, Kp, N, K, J, _stream(x)),
                    "spgnn_scores_bwd_w")
    if blockdiag_heads:
        H = blockdiag_heads; D = K // H
        out = torch.empty((2, H, D), dtype=torch.float32, device=x.device)
        if defer is not None:
            j = _capi.SumJob(kind=1, splits=splits, partials=part.data_ptr(), split_stride=J * Kp, out=out.data_ptr(), H=H, D=D, ld=Kp)
            defer.add(j, part, out)
            return out
        with torch.cuda.device(x.device):
            _capi.check(_capi.load().spgnn_sum_partials_blockdiag(part.data_ptr(), J * Kp, splits, H, D, Kp, out.data_ptr(),
                                                                  _stream(x)), "spgnn_sum_partials_blockdiag")
        return out
    if defer is not None and splits > 1 and (J * Kp) % 4 == 0:
        out = torch.empty((J, Kp), dtype=torch.float32, device=x.device)
        defer.add(_capi.SumJob(kind=0, splits=splits, partials=part.data_ptr(), split_stride=J * Kp, out=out.data_ptr(), n=J * Kp), part, out)
        return out[:, :K]
    return sum_partials(part)[:, :K]


DEFER_ATTN_GRADS = True   # a training step runs every layer's attention-vector gradient pass in ONE launch after the backward pass
ATTN_GRAD_QUEUE = None    # the AttnGradQueue of the step whose backward pass is being issued (train.TrainStep._front), else None


class AttnGradQueue:
    """The attention-vector gradients of every GATConv of a model, g_attn_l / g_attn_r = block diagonal of g_s^T ft (DGL:
    autograd of ``(ft * attn).sum(-1)``; reference call sites models.py:301-314, 425-456), as ONE streaming launch per
    training step instead of one per layer or level (spgnn_scores_bwd_w_multi; each pass alone is latency-bound and fills a
    quarter of the chip).  Nothing else depends on these gradients, so a layer's backward only RECORDS its pass here and
    returns no gradient for attn_l / attn_r; ``flush()`` - called by the step after ``backward()`` - launches all of them,
    sums the split partials in one more launch and hands each parameter its gradient (``.grad``, accumulated if one is
    there).  Only a training step installs a queue: plain ``loss.backward()`` keeps the per-layer launches and autograd's
    own bookkeeping.  Values are bit-identical either way."""

    def __init__(self, device):
        self.device, self.items = torch.device(device), []

    def add(self, g_s: torch.Tensor, ft: torch.Tensor, H: int, p_l: torch.Tensor, p_r: torch.Tensor) -> bool:
        """Record one layer's pass: g_s (N, 2H) fp32, ft (N, H*D) rows (fp32 or bf16), the two parameters.  -> False when the
        shapes do not fit the kernel (the caller then runs its own launch)."""
        N, K = ft.shape
        J = g_s.shape[1]
        if N == 0 or J > 8 or ft.stride(1) != 1 or ft.stride(0) % 4 or (self.items and self.items[0][1].shape[0] != N) \
                or (self.items and self.items[0][1].dtype != ft.dtype):
            return False
        if ft.data_ptr() % (16 if ft.dtype == torch.float32 else 8):
            return False
        self.items.append((g_s, ft, H, p_l, p_r))
        return True

    def flush(self) -> None:
        items, self.items = self.items, []
        if not items:
            return
        import ctypes
        lib = _capi.load()
        # The step-wide queue fills these outputs only AFTER this method returns, which is fine as long as every gradient is
        # just ASSIGNED below.  A parameter that already holds a gradient (autograd delivered another layer's first) or that two
        # recorded passes share is ADDED to here and now: those sums must be complete when this method reads them.
        seen, shared = set(), False
        for (_g, _f, _h, p_l, p_r) in items:
            for p in (p_l, p_r):
                shared = shared or p.grad is not None or id(p) in seen
                seen.add(id(p))
        sums = SumJobs(self.device, local=shared)
        outs = []
        for i0 in range(0, len(items), 8):
            chunk = items[i0:i0 + 8]
            jobs = (_capi.ScoresBwdWJob * len(chunk))()
            keep = []
            N = chunk[0][1].shape[0]
            bf16 = chunk[0][1].dtype == torch.bfloat16
            for q, (g_s, ft, H, p_l, p_r) in zip(jobs, chunk):
                K, J = ft.shape[1], g_s.shape[1]
                Kp = _pad16(K)
                splits = max(1, min(_SCORES_SPLIT_WAVES_SMALL // ((K + 255) // 256), N // 16))      # as scores_bwd_w / attn_vector_grads
                part = torch.empty((splits, J, Kp), dtype=torch.float32, device=ft.device)
                out = torch.empty((2, H, K // H), dtype=torch.float32, device=ft.device)
                q.g_s, q.g_s_stride, q.x, q.x_stride, q.partials = g_s.data_ptr(), g_s.stride(0), ft.data_ptr(), ft.stride(0), part.data_ptr()
                q.splits, q.Kp, q.K, q.J = splits, Kp, K, J
                keep.append((part, out))
                outs.append((out, p_l, p_r))
                sums.add(_capi.SumJob(kind=1, splits=splits, partials=part.data_ptr(), split_stride=J * Kp, out=out.data_ptr(), H=H,
                                      D=K // H, ld=Kp), part, out)
            key = (N, sum(it[1].shape[1] for it in chunk), sum(it[0].shape[1] for it in chunk), len(chunk), 2 if bf16 else 4)
            with torch.cuda.device(self.device), _timed("scores_bwd_w_multi", key):
                _capi.check(lib.spgnn_scores_bwd_w_multi(jobs, len(chunk), N, int(bf16), _stream(chunk[0][1])), "spgnn_scores_bwd_w_multi")
        sums.flush()
        for out, p_l, p_r in outs:
            for p, g in ((p_l, out[0]), (p_r, out[1])):
                g = g.view(p.shape)
                p.grad = g if p.grad is None else p.grad + g


def queue_attn_grads(g_s: torch.Tensor, ft: torch.Tensor, H: int, p_l, p_r) -> bool:
    """Hand a layer's attention-vector gradient pass to the running step's queue (see AttnGradQueue).  -> whether it took it."""
    q = ATTN_GRAD_QUEUE
    return bool(q is not None and DEFER_ATTN_GRADS and p_l is not None and p_r is not None and p_l.requires_grad and p_r.requires_grad
                and p_l.is_leaf and p_r.is_leaf               # the queue ASSIGNS .grad: a non-leaf attention vector would lose its gradient
                and q.add(g_s, ft, H, p_l, p_r))


PAIR_SCORE_GRADS = True   # a level's two attention-vector gradient passes (structure J = 2H, position J = 2) as one launch


def scores_bwd_w_blockdiag_pair(g_s0: torch.Tensor, x0: torch.Tensor, H0: int, g_s1: torch.Tensor, x1: torch.Tensor, H1: int,
                                defer: "SumJobs"):
    """``scores_bwd_w(g_s0, x0, blockdiag_heads=H0, defer=...)`` and the same for (g_s1, x1, H1) with BOTH streaming passes
    in one launch (spgnn_scores_bwd_w_pair) - bit-identical to the two calls, which is what runs when the shapes do not fit."""
    N = x0.shape[0]
    ok = (PAIR_SCORE_GRADS and defer is not None and N == x1.shape[0] and N > 0 and g_s0.shape[1] <= 8 and g_s1.shape[1] <= 8
          and _rows_aligned(x0) and _rows_aligned(x1) and x0.device == x1.device)
    if not ok:
        return (scores_bwd_w(g_s0, x0, blockdiag_heads=H0, defer=defer), scores_bwd_w(g_s1, x1, blockdiag_heads=H1, defer=defer))
    parts, outs, args = [], [], []
    for g_s, x, H in ((g_s0, x0, H0), (g_s1, x1, H1)):
        K, J = x.shape[1], g_s.shape[1]
        Kp = _pad16(K)
        splits = max(1, min(_SCORES_SPLIT_WAVES_SMALL // ((K + 255) // 256), N // 16))       # as scores_bwd_w
        part = torch.empty((splits, J, Kp), dtype=torch.float32, device=x.device)
        D = K // H
        out = torch.empty((2, H, D), dtype=torch.float32, device=x.device)
        parts.append(part); outs.append(out)
        args += [g_s.data_ptr(), g_s.stride(0), x.data_ptr(), x.stride(0), part.data_ptr(), splits, Kp, K, J]
        defer_job = _capi.SumJob(kind=1, splits=splits, partials=part.data_ptr(), split_stride=J * Kp, out=out.data_ptr(), H=H, D=D, ld=Kp)
        outs.append(defer_job)
    with torch.cuda.device(x0.device), _timed("scores_bwd_w_pair", (N, x0.shape[1], g_s0.shape[1], x1.shape[1], g_s1.shape[1])):
        _capi.check(_capi.load().spgnn_scores_bwd_w_pair(*args, N, _stream(x0)), "spgnn_scores_bwd_w_pair")
    defer.add(outs[1], parts[0], outs[0])
    defer.add(outs[3], parts[1], outs[2])
    return outs[0], outs[2]


def scores_bwd_x_(g_x: torch.Tensor, g_s: torch.Tensor, w_lr: torch.Tensor, accumulate: bool = True) -> None:
    """g_x (+)= g_s @ w_lr, in place."""
    N, K = g_x.shape
    J = g_s.shape[1]
    if not (_rows_aligned(g_x) and J <= 32) or N == 0:
        if accumulate:
            g_x.addmm_(g_s, w_lr)
        else:
            torch.mm(g_s, w_lr, out=g_x)
        return
    Kp = _pad16(K)
    w_p = _padded_rows(w_lr, Kp)
    with torch.cuda.device(g_x.device), _timed("scores_bwd_x", (N, K, J)):
        _capi.check(_capi.load().spgnn_scores_bwd_x(g_s.data_ptr(), g_s.stride(0), w_p.data_ptr(), Kp, g_x.data_ptr(),
                                                    g_x.stride(0), int(accumulate), N, K, J, _stream(g_x)),
                    "spgnn_scores_bwd_x")


class _SkinnyLinearFn(torch.autograd.Function):
    """y = x @ W^T + b for a tall x (N ~ 1e5 rows) and <= 32 output features: the three streaming kernels of the
    score projections instead of rocBLAS (which leaves most of the chip idle on a 22-column output)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        x = _rowmajor(x)
        y = scores_fwd(x, weight, bias=bias)
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, g_y):
        x, weight = ctx.saved_tensors
        cs_ = column_sums(g_y) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        g_y = g_y.contiguous()
        g_x = g_w = g_b = None
        if ctx.needs_input_grad[0]:
            g_x = torch.empty_like(x) if _rows_aligned(x) and x.is_contiguous() else torch.empty(x.shape, device=x.device)
            scores_bwd_x_(g_x, g_y, weight, accumulate=False)
        if ctx.needs_input_grad[1]:
            g_w = scores_bwd_w(g_y, x)
        if cs_ is not None:
            g_b = cs_
        return g_x, g_w, g_b


def skinny_linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor]) -> torch.Tensor:
    return _SkinnyLinearFn.apply(x, weight, bias)


class _CatPad(torch.autograd.Function):
    """cat(tensors, dim=1) into a buffer whose row stride is rounded up to 4 floats (16-byte rows for the
    vector kernels and aligned GEMM operands, e.g. 1063 -> 1064 for the first SPGNN layer); returns the
    (N, F) view.  Backward hands out column views of the incoming gradient (no copies)."""

    @staticmethod
    def forward(ctx, *tensors):
        widths = [t.shape[1] for t in tensors]
        F_ = sum(widths)
        Fp = (F_ + 3) // 4 * 4
        buf = torch.empty((tensors[0].shape[0], Fp), dtype=torch.float32, device=tensors[0].device)
        off = 0
        for t in tensors:
            buf[:, off:off + t.shape[1]].copy_(t)
            off += t.shape[1]
        if Fp > F_:
            buf[:, F_:].zero_()
        ctx.widths = widths
        return buf[:, :F_]

    @staticmethod
    def backward(ctx, g):
        outs, off = [], 0
        for w, need in zip(ctx.widths, ctx.needs_input_grad):
            outs.append(g[:, off:off + w] if need else None)
            off += w
        return tuple(outs)


class _MaskedCE(torch.autograd.Function):
    """(logits, labels, draws, sampling_p, class_weight) -> [numerator, denominator] of the masked class-weighted
    cross entropy (train.weighted_nll_sums) in one kernel that also leaves the numerator's gradient behind."""

    _opts = None         # per-call options set by masked_ce_sums (not autograd inputs: ``out`` would otherwise make the outputs
                         # count as views of an input)
    _tickets: dict = {}

    @staticmethod
    def forward(ctx, logits, labels, draws, sampling_p, class_weight):
        opts, _MaskedCE._opts = (_MaskedCE._opts or {}), None
        out, seed, unit = opts.get("out"), opts.get("draw_seed", 0), bool(opts.get("unit_grad"))
        rows = opts.get("rows")                 # LossRows: ``logits`` holds one row per LISTED node
        flag = opts.get("flag")                 # (2,) int32 [count, overflow]: a dense pass that turns NaN when the step's list overflowed
        N, C = logits.shape
        if logits.stride(1) != 1:
            logits = logits.contiguous()
        dev = logits.device
        nb = (N + 255) // 256
        part = torch.empty((max(nb, 1), 2), dtype=torch.float32, device=dev)
        g = torch.empty((N, C), dtype=torch.float32, device=dev) if ctx.needs_input_grad[0] else None
        s = out if out is not None else torch.empty((2,), dtype=torch.float32, device=dev)
        ticket = _MaskedCE._tickets.get(str(dev))
        if ticket is None:
            ticket = _MaskedCE._tickets[str(dev)] = torch.zeros((1,), dtype=torch.int32, device=dev)
        # unit gradient: the column sums of g ARE the gradient of a classifier bias - the last workgroup adds them too
        colsum = torch.empty((C,), dtype=torch.float32, device=dev) if (unit and g is not None and C <= 32 and N > 0) else None
        colpart = torch.empty((max(nb, 1), 32), dtype=torch.float32, device=dev) if colsum is not None else None
        with torch.cuda.device(dev), _timed("masked_ce", (N, C)):
            # the last workgroup adds the per-block pairs (block order): no reduction launch
            if rows is not None:
                assert N == rows.cap, "the logits of a loss-rows step have one row per listed node"
                _capi.check(_capi.load().spgnn_masked_ce_rows(logits.data_ptr(), logits.stride(0), labels.data_ptr(), rows.idx.data_ptr(),
                                                              rows.cnt.data_ptr(), class_weight.data_ptr(), part.data_ptr(), s.data_ptr(),
                                                              ticket.data_ptr(), _ptr(g), C, _ptr(colpart), _ptr(colsum), N, C,
                                                              _stream(logits)), "spgnn_masked_ce_rows")
            elif flag is not None:
                _capi.check(_capi.load().spgnn_masked_ce_step_flagged(logits.data_ptr(), logits.stride(0), labels.data_ptr(), _ptr(draws),
                                                                      int(seed) & 0xFFFFFFFFFFFFFFFF, _seed_off_ptr(dev) if draws is None else 0,
                                                                      sampling_p.data_ptr(), flag.data_ptr(), class_weight.data_ptr(),
                                                                      part.data_ptr(), s.data_ptr(), ticket.data_ptr(), _ptr(g), C,
                                                                      _ptr(colpart), _ptr(colsum), N, C, _stream(logits)),
                            "spgnn_masked_ce_step_flagged")
            else:
              _capi.check(_capi.load().spgnn_masked_ce_step(logits.data_ptr(), logits.stride(0), labels.data_ptr(), _ptr(draws), int(seed) & 0xFFFFFFFFFFFFFFFF,
                                                          _seed_off_ptr(dev) if draws is None else 0, sampling_p.data_ptr(),
                                                          class_weight.data_ptr(), part.data_ptr(), s.data_ptr(), ticket.data_ptr(),
                                                          _ptr(g), C, _ptr(colpart), _ptr(colsum), N, C, _stream(logits)),
                        "spgnn_masked_ce_step")
        ctx.save_for_backward(g, colsum)
        ctx.unit = unit
        ctx.set_materialize_grads(False)
        num, den = s[0], s[1]                   # two outputs: indexing ONE output outside would add a select node whose
        ctx.mark_non_differentiable(den)        # backward is a zero fill + a copy
        return num, den

    @staticmethod
    def backward(ctx, g_num, _g_den):
        g, colsum = ctx.saved_tensors
        if g is None or g_num is None:
            return None, None, None, None, None
        if ctx.unit:                   # the caller back-propagates the numerator itself (grad 1): the stored gradient as it is
            if colsum is not None:
                g._spgnn_colsum = colsum          # read by the node that produced the logits, if it is the direct consumer (column_sums)
            return g, None, None, None, None
        return g * g_num, None, None, None, None


# --------------------------------------------------------------------------------------------
# The step's loss joined to the node that ends in the classifier (train.TrainStep installs LOSS_HEAD around model(g)):
# logits, masked cross entropy, logit gradient and the classifier's weight / bias gradient from ONE pass over the classifier's
# input rows (spgnn_classifier_ce) instead of spgnn_scores_fwd + spgnn_masked_ce_step + spgnn_scores_bwd_w.
# --------------------------------------------------------------------------------------------
LOSS_HEAD: Optional["LossHead"] = None
FUSED_LOSS_HEAD = True           # False: the three separate launches (A/B, tests)


class LossHead:
    """What the loss of the step being issued needs (labels, sampling probabilities, the draws or the seed of the kernel's own
    draw, class weights, where the two sums go).  The node that takes it sets ``used`` and leaves ``g_logits`` - the gradient of
    the loss NUMERATOR with respect to the logits it returned: the step starts backward from it
    (``torch.autograd.backward(logits, g_logits)``) and the node's backward finds the classifier's gradients already formed."""

    def __init__(self, labels, sampling_p, draws, draw_seed: int, class_weight, sums_out, flag=None):
        self.labels, self.sampling_p, self.draws, self.draw_seed = labels, sampling_p, draws, int(draw_seed)
        self.class_weight, self.sums_out, self.flag = class_weight, sums_out, flag
        self.used, self.g_logits = False, None


_CE_TICKETS: dict = {}


def classifier_ce_supported(x: torch.Tensor, w_cls: torch.Tensor) -> bool:
    """fp32 rows (16-byte aligned) or bf16 rows (8-byte aligned, row stride a multiple of 4 elements); K % 128 == 0, K <= 1024."""
    if not (FUSED_LOSS_HEAD and x.is_cuda and x.dim() == 2 and x.shape[0] > 0 and x.stride(1) == 1 and x.shape[1] % 128 == 0
            and x.shape[1] <= 1024 and w_cls.shape[0] <= 32 and w_cls.shape[1] == x.shape[1] and w_cls.dtype == torch.float32):
        return False
    if x.dtype == torch.float32:
        return bool(_rows_aligned(x))
    return bool(x.dtype == torch.bfloat16 and x.stride(0) % 4 == 0 and x.data_ptr() % 8 == 0)


def classifier_ce(x: torch.Tensor, w_cls: torch.Tensor, b_cls: Optional[torch.Tensor], head: LossHead):
    """-> (logits (N, J), g_logits (N, J), w_partials (B, J, Kp), g_bias (J,) or None); the two loss sums land in
    ``head.sums_out``.  The classifier's weight gradient is ``w_partials.sum(0)[:, :K]`` (a deferred SumJob of the caller)."""
    _require_cuda(x, w_cls, head.labels, head.sampling_p, head.class_weight)
    N, K = x.shape
    J = w_cls.shape[0]
    dev = x.device
    Kp = _pad16(K)
    w_p = _padded_rows(w_cls.detach(), Kp)
    lib = _capi.load()
    rps = int(lib.spgnn_classifier_ce_rows_per_block(N))
    nblk = (N + rps - 1) // rps                                  # workgroups: the loss / column-sum partials
    B = int(lib.spgnn_classifier_ce_partial_slices(N, K, J))     # weight-gradient partial slices: workgroups x row groups
    logits = torch.empty((N, J), dtype=torch.float32, device=dev)
    g_logits = torch.empty((N, J), dtype=torch.float32, device=dev)
    wpart = torch.empty((B, J, Kp), dtype=torch.float32, device=dev)
    part = torch.empty((nblk, 2), dtype=torch.float32, device=dev)
    colpart = torch.empty((nblk, 32), dtype=torch.float32, device=dev)
    colsum = torch.empty((J,), dtype=torch.float32, device=dev)
    ticket = _CE_TICKETS.get(str(dev))
    if ticket is None:
        ticket = _CE_TICKETS[str(dev)] = torch.zeros((1,), dtype=torch.int32, device=dev)
    bias_c = None if b_cls is None else b_cls.detach().contiguous()
    draws = None if head.draws is None else head.draws.contiguous()
    bf16 = x.dtype == torch.bfloat16
    fn = lib.spgnn_classifier_ce_bf16 if bf16 else lib.spgnn_classifier_ce
    with torch.cuda.device(dev), _timed("classifier_ce_bf16" if bf16 else "classifier_ce", (N, K, J)):
        _capi.check(fn(x.data_ptr(), x.stride(0), w_p.data_ptr(), Kp, _ptr(bias_c), head.labels.data_ptr(), _ptr(draws),
                                            head.draw_seed & 0xFFFFFFFFFFFFFFFF, _seed_off_ptr(dev) if draws is None else 0,
                                            head.sampling_p.data_ptr(), head.class_weight.data_ptr(), _ptr(head.flag), logits.data_ptr(),
                                            logits.stride(0), g_logits.data_ptr(), g_logits.stride(0), wpart.data_ptr(), part.data_ptr(),
                                            head.sums_out.data_ptr(), ticket.data_ptr(), colpart.data_ptr(), colsum.data_ptr(), N, K, J,
                                            _stream(x)), "spgnn_classifier_ce")
    return logits, g_logits, wpart, colsum


def column_sums(g: torch.Tensor) -> torch.Tensor:
    """g.sum(0) - or, when ``g`` is the logit gradient the loss kernel just handed over, the column sums that kernel's last
    workgroup already formed (the attribute exists only on that very tensor: any op in between makes a new one)."""
    cs = getattr(g, "_spgnn_colsum", None)
    return cs if cs is not None and cs.shape[0] == g.shape[1] else g.sum(0)


# --------------------------------------------------------------------------------------------
# The rows of a step that reach the loss (reference job_runner.py:1896-1900: F.cross_entropy(pre[mask], y[mask], weight=w)).
# A node outside the mask contributes neither to the loss nor to any gradient, and what follows the last aggregation - the
# output layer's projection, the head mean, the classifier, their backward products - is row-wise: a training step may run
# that part on the kept rows only (about 0.27 N at SAMPLING_RATE 0.15 with ~21 labelled nodes per tree) with the same loss
# and gradients.  train.TrainStep(loss_rows_only=True) lists the rows before the forward pass and installs the list here; the
# node that fuses output layer and classifier (_GATAggFirstFn) picks it up and hands back ONE ROW PER LISTED NODE.
# --------------------------------------------------------------------------------------------
LOSS_ROWS: Optional["LossRows"] = None
LIST_AWARE_TRAVERSALS = True     # the output layer's backward traversals read the listed g_z rows through LossRows.inv (off: expand first)


class LossRows:
    """``idx`` (cap,) int32: the kept nodes in ascending order (slots past the count name node 0); ``inv`` (N,) int32: a node's
    slot or -1; ``cnt`` (2,) int32 on the device: [count, overflow flag].  ``cap`` is fixed on the host (a captured step's launch
    grids are), the count is not: slots past it hold zero rows, and a draw that keeps more than ``cap`` nodes sets the flag
    (never cleared by the kernels) and turns the loss into NaN."""

    def __init__(self, N: int, cap: int, device, cnt: Optional[torch.Tensor] = None, forward: bool = True):
        # ``forward`` False: the forward pass runs every row (the model's outputs are the reference's, row for row) and only the
        # BACKWARD products use the list - the rows they skip are exactly zero in the dense step
        self.N, self.cap, self.used, self.forward = int(N), int(cap), False, bool(forward)
        self.idx = torch.empty((self.cap,), dtype=torch.int32, device=device)
        self.inv = torch.empty((max(self.N, 1),), dtype=torch.int32, device=device)
        self.cnt = cnt if cnt is not None else torch.zeros((2,), dtype=torch.int32, device=device)
        self._counts = torch.empty(((self.N + 255) // 256 or 1,), dtype=torch.int32, device=device)


def loss_rows(sampling_p: torch.Tensor, draws: Optional[torch.Tensor], draw_seed: int, cap: int,
              cnt: Optional[torch.Tensor] = None, forward: bool = True) -> LossRows:
    """The nodes with ``rn < sampling_p`` - rn from ``draws`` or, None, from the counter hash of (``draw_seed``, the step counter
    installed as DROPOUT_SEED_OFFSET, node): exactly the mask spgnn_masked_ce_step would draw from the same arguments."""
    _require_cuda(sampling_p, draws)
    N = sampling_p.shape[0]
    r = LossRows(N, cap, sampling_p.device, cnt, forward)
    with torch.cuda.device(sampling_p.device), _timed("loss_rows", (N, cap)):
        _capi.check(_capi.load().spgnn_loss_rows(_ptr(draws), int(draw_seed) & 0xFFFFFFFFFFFFFFFF,
                                                 _seed_off_ptr(sampling_p.device) if draws is None else 0, sampling_p.data_ptr(), N,
                                                 r._counts.data_ptr(), r.cap, r.idx.data_ptr(), r.inv.data_ptr(), r.cnt.data_ptr(),
                                                 _stream(sampling_p)), "spgnn_loss_rows")
    return r


def gather_rows(x: torch.Tensor, rows: LossRows) -> torch.Tensor:
    """x (N, C) -> (cap, C): the listed rows in list order, zero rows behind them."""
    assert x.shape[0] == rows.N and x.shape[1] % 4 == 0 and x.stride(1) == 1 and _rows_aligned(x)
    out = torch.empty((rows.cap, x.shape[1]), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device), _timed("gather_rows", (rows.cap, x.shape[1])):
        _capi.check(_capi.load().spgnn_gather_rows(x.data_ptr(), x.stride(0), rows.idx.data_ptr(), rows.cnt.data_ptr(), rows.cap, x.shape[1],
                                                   out.data_ptr(), out.stride(0), _stream(x)), "spgnn_gather_rows")
    return out


def expand_rows(xc: torch.Tensor, rows: LossRows) -> torch.Tensor:
    """xc (cap, C) -> (N, C): the listed rows back at their nodes, zero rows everywhere else."""
    assert xc.shape[0] == rows.cap and xc.shape[1] % 4 == 0 and xc.stride(1) == 1 and _rows_aligned(xc)
    out = torch.empty((rows.N, xc.shape[1]), dtype=torch.float32, device=xc.device)
    with torch.cuda.device(xc.device), _timed("expand_rows", (rows.N, xc.shape[1])):
        _capi.check(_capi.load().spgnn_expand_rows(xc.data_ptr(), xc.stride(0), rows.inv.data_ptr(), rows.N, xc.shape[1], out.data_ptr(),
                                                   out.stride(0), _stream(xc)), "spgnn_expand_rows")
    return out


class _GatherRowsFn(torch.autograd.Function):
    """x (N, C) -> the listed rows (cap, C), differentiable: the gradient goes back to the listed nodes' rows, zero rows to all
    others.  Put behind a model's LAST aggregation in a loss-rows step, it makes every row-wise operation after it - Linear
    stacks, activations, the classifier - run on the kept rows without knowing.  fp32 rows, or bf16 rows moved as pairs."""

    @staticmethod
    def forward(ctx, x, rows: LossRows):
        ctx.rows, ctx.dtype = rows, x.dtype
        xv = x if x.dtype == torch.float32 else x.view(torch.float32)
        out = gather_rows(xv, rows)
        return out if x.dtype == torch.float32 else out.view(x.dtype)

    @staticmethod
    def backward(ctx, g):
        g = _rowmajor(g)
        if g.dtype != torch.float32:
            return expand_rows(g.contiguous().view(torch.float32), ctx.rows).view(ctx.dtype), None
        if not _rows_aligned(g):
            g = g.contiguous()
        return expand_rows(g, ctx.rows), None


def take_loss_rows(x: torch.Tensor, has_classifier: bool) -> torch.Tensor:
    """``x`` = what a model's last aggregation produced.  In a loss-rows step whose forward pass uses the list
    (TrainStep(loss_rows_only=True)) and whose head ends in the fused classifier: the listed rows of ``x`` (and the list is marked
    as taken: the logits will have one row per listed node).  Otherwise ``x`` itself."""
    rows = LOSS_ROWS
    if rows is None or not rows.forward or not has_classifier or rows.N != x.shape[0] or x.stride(1) != 1:
        return x
    ok = (x.dtype == torch.float32 and x.shape[1] % 4 == 0 and _rows_aligned(x)) or \
         (x.dtype == torch.bfloat16 and x.shape[1] % 8 == 0 and x.stride(0) % 8 == 0 and x.data_ptr() % 16 == 0)
    if not ok:
        return x
    rows.used = True
    return _GatherRowsFn.apply(x, rows)


def masked_ce_sums(logits: torch.Tensor, labels: torch.Tensor, draws: Optional[torch.Tensor], sampling_p: torch.Tensor,
                   class_weight: torch.Tensor, out: Optional[torch.Tensor] = None, draw_seed: int = 0, unit_grad: bool = False,
                   rows: Optional["LossRows"] = None, flag: Optional[torch.Tensor] = None):
    """-> (sum_i m_i w[y_i] nll_i, sum_i m_i w[y_i]), m = draws < sampling_p (reference job_runner.py:1896-1900).
    ``out`` (2,) fp32, optional: where the two sums are to be written (train.FlatBucket's tail slots).  ``draws`` None: the
    kernel draws rn_i itself from its counter hash of (``draw_seed``, the step counter installed as DROPOUT_SEED_OFFSET, i).
    ``unit_grad``: the caller promises to back-propagate the numerator with gradient exactly 1 (``num.backward()``), so the
    stored gradient is handed on without the multiplication.  ``rows`` (:class:`LossRows`, made from the same draws):
    ``logits`` has one row per listed node instead of one per node - the same two sums up to summation order.  ``flag`` (a
    LossRows.cnt tensor, without ``rows``): the dense pass, NaN when the list of this step overflowed."""
    _require_cuda(logits, labels, draws, sampling_p, class_weight)
    assert labels.dtype == torch.int64 and logits.dtype == torch.float32
    _MaskedCE._opts = {"out": out, "draw_seed": int(draw_seed), "unit_grad": unit_grad, "rows": rows, "flag": flag}
    return _MaskedCE.apply(logits, labels.contiguous(), None if draws is None else draws.contiguous(), sampling_p.contiguous(),
                           class_weight.contiguous())


def _prepared_linear_operands(weight: torch.Tensor, C: int, K: int):
    """(w, w_ps, scale, (w_t, w_t_ps) or None) for ``weight`` (C, K) when this forward pass prepared it (prepared_weights) -
    either the parameter itself, or the transposed view ``p.t()`` of a prepared (K, C) parameter (GraphConv stores its weight
    as (in, out)): then the prepared TRANSPOSE is the product's operand and the prepared matrix its transpose.  Else None."""
    if not PRESPLIT_B:
        return None
    hit = _PREP_ACTIVE.get((id(weight), 0))
    if hit is not None and hit[0][5] == (C, 0, K, C):
        dst, ps, dst_t, ps_t, sw, _ = hit[0]
        return dst[:, :K], ps[:, :K], sw, ((dst_t[:, :C], ps_t[:, :C]) if hit[1] else None)
    base = weight._base
    if base is not None and base.dim() == 2 and tuple(weight.shape) == (base.shape[1], base.shape[0]) \
            and weight.stride() == (base.stride(1), base.stride(0)) and weight.data_ptr() == base.data_ptr():
        hit = _PREP_ACTIVE.get((id(base), 0))
        if hit is not None and hit[1] and hit[0][5] == (K, 0, C, K):
            dst, ps, dst_t, ps_t, sw, _ = hit[0]                       # dst = base (K, C); dst_t = base^T = weight (C, K)
            return dst_t[:, :K], ps_t[:, :K], sw, (dst[:, :C], ps[:, :C])
    return None


class _LinearFn(torch.autograd.Function):
    """act(x @ W^T + b) for tall x on the fp32-accurate matrix-core GEMMs (the nn.Linear / weight products inside
    GraphConv, GINConv and SAGEConv; reference models.py:172-182, 236-246, 668-679): forward and input gradient on
    spgnn_gemm_nt (bias + activation in its epilogue), weight gradient on spgnn_gemm_tn with the bias gradient as a
    by-product of the operand stream."""

    @staticmethod
    def forward(ctx, x, weight, bias, act, addend=None, drop=None, w_cls=None, b_cls=None):
        """``drop`` = (p, seed): the result is dropout(act(...), p) under spgnn_cat_dropout's hash mask; the backward pass
        then undoes dropout and activation in ONE pass.  The mask is applied by the product's own epilogue and only the dropped
        result is kept: ReLU / LeakyReLU derivatives need its sign, ELU / tanh ones the value, which for a kept element is
        the stored one times (1 - p) again (spgnn_act_bwd_dropped).
        ``w_cls`` (J, C) / ``b_cls``: a skinny classifier on the result joins the node, outputs (y, logits) - when only the
        logits carry a gradient the (N, C) gradient of y is formed, multiplied by act' and consumed in one pass
        (spgnn_act_bwd_proj) instead of written by spgnn_scores_bwd_x and re-read by spgnn_act_bwd."""
        ctx.set_materialize_grads(False)
        x = _rowmajor(x)
        if not _rows_aligned(x):
            x = cat_padded((x,))                                           # 16-byte rows for the GEMM operand
        N, K = x.shape
        C = weight.shape[0]
        prep = _prepared_linear_operands(weight, C, K)
        ctx.wt = None
        ctx.w_transposed = weight.dim() == 2 and weight.stride(1) != 1 and weight.stride(0) == 1 and K % 4 == 0 and C % 4 == 0
        if prep is not None:
            # padded rows, scale, pre-split form and the transposes: built by this pass's spgnn_weight_prep
            w, wb, sw, ctx.wt = prep
            bps = True
        else:
            w = weight if weight.stride(1) == 1 else weight.contiguous()      # e.g. GraphConv's (in, out) weight seen as W^T
            if not _rows_aligned(w):
                w = torch.nn.functional.pad(w, (0, -w.shape[1] % 4)).contiguous()[:, :w.shape[1]]
            sw, wb, bps = pow2_scale(w), w, False
        sx = operand_scale(x)
        y = torch.empty((N, C), dtype=torch.float32, device=x.device)
        blk = new_scale_block(x.device) if EMIT_SCALES else None            # max |y| from the product's own epilogue
        xa, aps = const_operand(x, sx) if bps else (x, False)             # node data (the first layer's input): split once per batch
        ctx.x_ps = xa if aps else None
        q = NtProblem(xa, wb, sx, sw, out=y, b_presplit=bps, a_presplit=aps)
        q.c.bias, q.c.activation, q.c.absmax_out = _ptr(bias), int(act), _ptr(blk)
        if addend is not None:                                             # act(x W^T + b + addend) in one epilogue
            addend = _rowmajor(addend)
            assert addend.shape == (N, C) and _rows_aligned(addend)
            q.c.addend, q.c.addend_stride = addend.data_ptr(), addend.stride(0)
        dropping = drop is not None and drop[0] > 0.0
        assert not dropping or C % 4 == 0, "linear(drop=...): the output width must be a multiple of 4"
        in_epilogue = dropping and blk is not None and EPILOGUE_DROPOUT
        if in_epilogue:
            q.c.drop_p, q.c.drop_seed, q.c.drop_seed_offset = float(drop[0]), int(drop[1]), _seed_off_ptr(x.device)
        import ctypes
        with torch.cuda.device(x.device), _timed("gemm_nt", (N, C, K)):
            _capi.check(_capi.load().spgnn_gemm_nt_problem_run(ctypes.byref(q.c), q.b_presplit, _stream(x)), "spgnn_gemm_nt_problem_run")
        ctx.act, ctx.has_bias, ctx.has_addend, ctx.scale_block, ctx.drop = act, bias is not None, addend is not None, blk, None
        ctx.has_cls = w_cls is not None
        if dropping:
            assert w_cls is None
            ctx.drop = (float(drop[0]), int(drop[1]))
            ctx.drop_in_epilogue = bool(in_epilogue)
            if in_epilogue:                                               # y IS the dropped result
                ctx.save_for_backward(x, w, sx, sw, y if act != ACT_NONE else None)
                return y
            yd = torch.empty_like(y)
            blk = new_scale_block(x.device)
            with torch.cuda.device(x.device):
                _capi.check(_capi.load().spgnn_cat_dropout(y.data_ptr(), y.stride(0), yd.data_ptr(), yd.stride(0), N, C, 0, C, float(drop[0]),
                                                           int(drop[1]), _seed_off_ptr(x.device), 0, blk.data_ptr(), _stream(x)),
                            "spgnn_cat_dropout")
            ctx.scale_block = blk
            ctx.save_for_backward(x, w, sx, sw, y if act != ACT_NONE else None)
            return yd
        if w_cls is not None:
            logits = scores_fwd(y, w_cls.detach(), bias=b_cls.detach() if b_cls is not None else None)
            ctx.has_bcls = b_cls is not None
            ctx.save_for_backward(x, w, sx, sw, y, w_cls)
            return y, logits
        ctx.save_for_backward(x, w, sx, sw, y if act != ACT_NONE else None)
        return y

    @staticmethod
    def backward(ctx, g, g_logits=None):
        n_in = 8
        if g is None and g_logits is None:
            return (None,) * n_in
        w_cls = None
        if ctx.has_cls:
            x, w, sx, sw, y, w_cls = ctx.saved_tensors
        else:
            x, w, sx, sw, y = ctx.saved_tensors
        N, K = x.shape
        C = w.shape[0]
        g_wcls = g_bcls = sg = None
        if getattr(ctx, "pre_activated", False):      # the caller hands over the PRE-activation gradient with its scale (ops.pool_max)
            g, sg = _rowmajor(g), ctx.pre_scale
        if g_logits is not None:
            cs = column_sums(g_logits)
            g_logits = _rowmajor(g_logits)
            wc = w_cls.detach()
            fused = g is None and C % 4 == 0 and act_bwd_proj_supported(1, C, wc.shape[0], wc)
            ride = fused and ctx.needs_input_grad[6] and act_bwd_proj_wgrad_supported(1, C, wc.shape[0], ctx.act, y)
            if ctx.needs_input_grad[6] and not ride:
                g_wcls = scores_bwd_w(g_logits, y)
            g_bcls = cs if ctx.has_bcls and ctx.needs_input_grad[7] else None
            if ride:                                      # ... with the classifier's weight gradient g_logits^T y from the same rows
                wj = SumJobs(x.device)                    # (the step's queue when one is installed: only the parameter reads g_wcls)
                g, sg, g_wcls = act_bwd_proj(g_logits, wc, y, 1, C, ctx.act, wgrad_jobs=wj)
                wj.flush()
            elif fused:
                # g_logits Wc * act'(y) in one pass: the (N, C) gradient of y is never written
                g, sg = act_bwd_proj(g_logits, wc, y if ctx.act != ACT_NONE else None, 1, C, ctx.act)
            else:
                if g is None:
                    g = torch.empty((N, (C + 3) // 4 * 4), dtype=torch.float32, device=x.device)[:, :C]
                    scores_bwd_x_(g, g_logits, wc, accumulate=False)
                else:
                    g0 = _rowmajor(g)
                    g = g0.clone() if _rows_aligned(g0) else cat_padded((g0,))
                    scores_bwd_x_(g, g_logits, wc, accumulate=True)
        if sg is not None:
            pass                                          # activation already undone (spgnn_act_bwd_proj)
        elif ctx.drop is not None:                        # dropout's and the activation's backward in one pass, mask regenerated
            g = _rowmajor(g)
            if not _rows_aligned(g):
                g = g.contiguous()
            g_pre = torch.empty((N, C), dtype=torch.float32, device=g.device)
            sg = new_scale_block(g.device)
            with torch.cuda.device(g.device), _timed("act_bwd", (N, 1, C, ctx.act, 0)):
                # from the dropped result when that is all the forward kept (the product's epilogue applied the mask)
                fn = _capi.load().spgnn_act_bwd_dropped if ctx.drop_in_epilogue else _capi.load().spgnn_act_bwd_dropout
                _capi.check(fn(g.data_ptr(), g.stride(0), _ptr(y), y.stride(0) if y is not None else 0,
                               g_pre.data_ptr(), g_pre.stride(0), sg.data_ptr(), N, C, ctx.act,
                               ctx.drop[0], ctx.drop[1], _seed_off_ptr(g.device), _stream(g)), "spgnn_act_bwd_dropout")
            g = g_pre
        elif ctx.act != ACT_NONE and C % 4 == 0:
            g, sg = act_bwd(_rowmajor(g), y, 1, C, ctx.act, False)
        else:
            g = _rowmajor(g)
            if ctx.act != ACT_NONE:
                g = g * {ACT_ELU: torch.where(y > 0, torch.ones_like(y), y + 1), ACT_TANH: 1 - y * y,
                         ACT_RELU: (y > 0).to(y.dtype), ACT_LRELU: torch.where(y > 0, torch.ones_like(y), torch.full_like(y, 0.01))}[ctx.act]
            if not _rows_aligned(g):
                g = cat_padded((g,))
            sg = operand_scale(g)
        g_x = g_w = g_b = None
        if ctx.needs_input_grad[0]:
            g_x = torch.empty((N, (K + 3) // 4 * 4), dtype=torch.float32, device=x.device)[:, :K]
            if ctx.wt is not None:
                gemm_nt(g, ctx.wt[1], sg, sw, out=g_x, b_presplit=True)
            else:
                w_t = w.t().contiguous() if C % 4 == 0 else torch.nn.functional.pad(w.t(), (0, -C % 4)).contiguous()[:, :C]
                gemm_nt(g, w_t, sg, sw, out=g_x)
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            xps = getattr(ctx, "x_ps", None)
            xb, bps = (xps, True) if xps is not None else (x, False)
            if ctx.has_bias:
                g_w, g_b = gemm_tn(g, xb, sg, sx, want_colsum=True, b_presplit=bps)
            elif ctx.w_transposed:
                # the weight is the transposed view of an (in, out) parameter (GraphConv): x^T g lands in the parameter's own
                # layout, so autograd's accumulation takes the tensor as it is instead of copying a transposed view
                g_w = gemm_tn(x, g, sx, sg).t()
            else:
                g_w = gemm_tn(g, xb, sg, sx, b_presplit=bps)
        g_add = None
        if ctx.has_addend and ctx.needs_input_grad[4]:
            g_add = g                                    # the addend's producer (the pair's first product) takes it with its scale
            if sg is not None:
                g._spgnn_scale = (g._version, sg)
        return g_x, g_w, g_b, None, g_add, None, g_wcls, g_bcls


def linear_drop_supported(x: torch.Tensor, weight: torch.Tensor) -> bool:
    """Whether linear(x, weight, ..., drop=...) can run: the matrix-core path and an output width that is a multiple of 4."""
    return bool(x.is_cuda and GEMM_MODE == "f16x3" and x.dim() == 2 and x.shape[0] >= MIN_GEMM_ROWS and weight.shape[0] >= 32
                and weight.shape[1] >= 32 and x.dtype == torch.float32 and weight.dtype == torch.float32 and weight.shape[0] % 4 == 0)


def linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, act: int = 0,
           addend: Optional[torch.Tensor] = None, drop=None) -> torch.Tensor:
    """act(F.linear(x, weight, bias) + addend).  Tall operands (>= 512 rows, both widths >= 32) on a ROCm device take the
    matrix-core GEMM path; anything else goes to torch (tiny products are launch-bound either way).  ``drop`` = (p, seed):
    dropout of the result under the hash mask (matrix-core path with a width that is a multiple of 4 only; see
    :func:`linear_drop_supported`)."""
    N = x.shape[0] if x.dim() == 2 else 0
    # a small inference batch goes to the skinny product (exact fp32, no operand scale) even where the row count would also
    # admit the matrix-core path (512 <= N <= SKINNY_ROWS): a replayed per-scan forward must not depend on a scale that was
    # computed from the FIRST scan's data
    skinny = bool(skinny_rows(N) and addend is None and drop is None and act in (ACT_NONE, ACT_ELU, ACT_TANH, ACT_RELU, ACT_LRELU))
    if (not skinny and x.is_cuda and GEMM_MODE == "f16x3" and x.dim() == 2 and N >= MIN_GEMM_ROWS and weight.shape[0] >= 32 and weight.shape[1] >= 32
            and x.dtype == torch.float32 and weight.dtype == torch.float32
            and (addend is None or (weight.shape[0] % 4 == 0 and addend.shape == (N, weight.shape[0])))):
        y = _LinearFn.apply(x, weight, bias, act, addend, drop)
        blk = getattr(y.grad_fn, "scale_block", None) if y.grad_fn is not None else None
        if blk is not None:
            y._spgnn_scale = (y._version, blk)          # the result's GEMM operand scale, from the product's epilogue
        return y
    assert drop is None, "linear(drop=...) needs the matrix-core path (linear_drop_supported)"
    if (x.is_cuda and x.dim() == 2 and skinny_rows(N) and addend is None and x.dtype == torch.float32 and weight.dtype == torch.float32
            and act in (ACT_NONE, ACT_ELU, ACT_TANH, ACT_RELU, ACT_LRELU)):
        # a small inference batch (one scan): the library's skinny product instead of rocBLAS, bias and activation in its epilogue
        xa = x if _rows_aligned(x) else cat_padded((_rowmajor(x),))
        wa = weight if (_rows_aligned(weight) and weight.stride(1) == 1) else cat_padded((weight.detach(),))
        if xa.shape[1] == wa.shape[1]:
            return gemm_nt_skinny(xa, wa, bias=bias.contiguous() if bias is not None else None, act=act)
    y = torch.nn.functional.linear(x, weight, bias)
    if addend is not None:
        y = y + addend
    if act == ACT_LRELU:
        y = torch.nn.functional.leaky_relu(y, 0.01)
    if act == ACT_ELU:
        y = torch.nn.functional.elu(y)
    elif act == ACT_TANH:
        y = torch.tanh(y)
    elif act == ACT_RELU:
        y = torch.relu(y)
    return y


def linear_act_classifier_supported(x: torch.Tensor, weight: torch.Tensor, w_cls: torch.Tensor) -> bool:
    N = x.shape[0] if x.dim() == 2 else 0
    return (LINEAR_ACT_CLASSIFIER and x.is_cuda and GEMM_MODE == "f16x3" and x.dim() == 2 and N >= MIN_GEMM_ROWS and weight.shape[0] >= 32
            and weight.shape[1] >= 32 and x.dtype == torch.float32 and weight.dtype == torch.float32 and weight.shape[0] % 4 == 0
            and w_cls.shape[0] <= 32 and w_cls.shape[1] == weight.shape[0] and w_cls.dtype == torch.float32)


def linear_act_classifier(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], act: int, w_cls: torch.Tensor,
                          b_cls: Optional[torch.Tensor]):
    """y = act(F.linear(x, weight, bias)) and logits = F.linear(y, w_cls, b_cls) as ONE autograd node -> (y, logits): the last
    product of the reference's GIN MLP (models.py:236-246) followed by the *Net's ``gnn_out`` (models.py:988).  See
    _LinearFn.forward; needs :func:`linear_act_classifier_supported`."""
    assert linear_act_classifier_supported(x, weight, w_cls)
    y, logits = _LinearFn.apply(x, weight, bias, act, None, None, w_cls, b_cls)
    blk = getattr(y.grad_fn, "scale_block", None) if y.grad_fn is not None else None
    if blk is not None:
        y._spgnn_scale = (y._version, blk)
    return y, logits


class _LinearClassifierFn(torch.autograd.Function):
    """y = x W^T + b (N, C) and a skinny classifier on it, logits = y Wc^T + bc (J <= 32), as ONE autograd node (the *Net's
    ``gnn_out`` on the output layer's head mean: reference models.py:921-933 with 320-327).  The classifier is folded
    through the product: with P = Wc W (J, K),  logits = x P^T + (Wc b + bc)  - a skinny pass over x instead of one over
    the (N, C) result - and  g_Wc = (g_logits^T x) W^T + colsum(g_logits) b^T  needs no pass over y either.  When only
    the logits carry a gradient (the training step: the embedding is returned but not part of the loss) the (N, C)
    gradient of y is never formed:  g_x = g_logits P,  g_W = Wc^T (g_logits^T x),  g_b = Wc^T colsum(g_logits).
    Otherwise g_y + g_logits Wc takes the ordinary route."""

    @staticmethod
    def forward(ctx, x, weight, bias, w_cls, b_cls):
        ctx.set_materialize_grads(False)
        x = _rowmajor(x)
        if not _rows_aligned(x):
            x = cat_padded((x,))
        prep = _prepared_linear_operands(weight, weight.shape[0], weight.shape[1])
        if prep is not None:
            w, w_ps, sw, _wt = prep
            sx = operand_scale(x)
            y = gemm_nt(x, w_ps, sx, sw, bias=bias, b_presplit=True)
        else:
            w = weight if weight.stride(1) == 1 else weight.contiguous()
            if not _rows_aligned(w):
                w = torch.nn.functional.pad(w, (0, -w.shape[1] % 4)).contiguous()[:, :w.shape[1]]
            sx, sw = operand_scale(x), pow2_scale(w)
            y = gemm_nt(x, w, sx, sw, bias=bias)
        wc = w_cls.detach()
        P = torch.mm(wc, w.detach())
        c0 = b_cls
        if bias is not None:
            c0 = torch.mv(wc, bias.detach()) if c0 is None else torch.addmv(c0.detach(), wc, bias.detach())
        logits = scores_fwd(x, P, bias=c0)
        ctx.has_bias, ctx.has_bcls = bias is not None, b_cls is not None
        ctx.save_for_backward(x, w, sx, sw, P, w_cls, bias)
        return y, logits

    @staticmethod
    def backward(ctx, g_y, g_logits):
        if g_y is None and g_logits is None:
            return None, None, None, None, None
        x, w, sx, sw, P, w_cls, bias = ctx.saved_tensors
        N, K = x.shape
        C = w.shape[0]
        wc = w_cls.detach()
        g_x = g_w = g_b = g_wcls = g_bcls = cs = M1 = None
        if g_logits is not None:
            cs = column_sums(g_logits)
            g_logits = _rowmajor(g_logits)
            M1 = scores_bwd_w(g_logits, x)                               # g_logits^T x  (J, K)
            if ctx.needs_input_grad[3]:
                g_wcls = torch.mm(M1, w.t())
                if ctx.has_bias:
                    g_wcls.addr_(cs, bias.detach())
            g_bcls = cs if ctx.has_bcls and ctx.needs_input_grad[4] else None
        if g_y is None:                                   # the folded route: no (N, C) gradient
            if ctx.needs_input_grad[0]:
                g_x = torch.empty((N, (K + 3) // 4 * 4), dtype=torch.float32, device=x.device)[:, :K]
                scores_bwd_x_(g_x, g_logits, P, accumulate=False)
            if ctx.needs_input_grad[1]:
                g_w = torch.mm(wc.t(), M1)
            if ctx.has_bias and ctx.needs_input_grad[2]:
                g_b = torch.mv(wc.t(), cs)
            return g_x, g_w, g_b, g_wcls, g_bcls
        g = _rowmajor(g_y)
        if g_logits is not None:
            g = g.clone() if g.data_ptr() == g_y.data_ptr() else g
            if not _rows_aligned(g):
                g = cat_padded((g,))
            scores_bwd_x_(g, g_logits, wc, accumulate=True)
        elif not _rows_aligned(g):
            g = cat_padded((g,))
        sg = pow2_scale(g)
        if ctx.needs_input_grad[0]:
            g_x = torch.empty((N, (K + 3) // 4 * 4), dtype=torch.float32, device=x.device)[:, :K]
            w_t = w.t().contiguous() if C % 4 == 0 else torch.nn.functional.pad(w.t(), (0, -C % 4)).contiguous()[:, :C]
            gemm_nt(g, w_t, sg, sw, out=g_x)
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            if ctx.has_bias:
                g_w, g_b = gemm_tn(g, x, sg, sx, want_colsum=True)
            else:
                g_w = gemm_tn(g, x, sg, sx)
        return g_x, g_w, g_b, g_wcls, g_bcls


def linear_classifier_supported(x: torch.Tensor, weight: torch.Tensor, w_cls: torch.Tensor) -> bool:
    return (x.is_cuda and GEMM_MODE == "f16x3" and x.dim() == 2 and x.shape[0] >= MIN_GEMM_ROWS and weight.shape[0] >= 32
            and weight.shape[1] >= 32 and x.dtype == torch.float32 and weight.dtype == torch.float32
            and w_cls.shape[0] <= 32 and w_cls.shape[1] == weight.shape[0] and weight.shape[0] % 4 == 0)


class _CatDropout(torch.autograd.Function):
    """dropout(cat(tensors, dim=1), p) in one pass per source into a buffer with 16-byte rows; the keep mask is a
    counter hash of (seed, element) that the backward regenerates - no mask tensor, no separate cat copy."""

    @staticmethod
    def forward(ctx, p, seed, *tensors):
        ctx.set_materialize_grads(False)       # no zero tensor for the unused gradient of the scale output
        widths = [t.shape[1] for t in tensors]
        F_ = sum(widths)
        Fp = (F_ + 3) // 4 * 4
        N = tensors[0].shape[0]
        buf = torch.empty((N, Fp), dtype=torch.float32, device=tensors[0].device)
        if Fp > F_:
            buf[:, F_:].zero_()
        lib = _capi.load()
        scale = new_scale_block(buf.device)                # every source folds the maxima of what it writes into it
        off = 0
        with torch.cuda.device(buf.device):
            for t in tensors:
                t = t if t.stride(1) == 1 else t.contiguous()
                _capi.check(lib.spgnn_cat_dropout(t.data_ptr(), t.stride(0), buf.data_ptr(), buf.stride(0), N, t.shape[1], off, F_,
                                                  p, seed, _seed_off_ptr(buf.device), 0, scale.data_ptr(), _stream(buf)),
                            "spgnn_cat_dropout")
                off += t.shape[1]
        ctx.widths, ctx.p, ctx.seed = widths, p, seed
        ctx.mark_non_differentiable(scale)
        return buf[:, :F_], scale

    @staticmethod
    def backward(ctx, g, _g_scale):
        if g is None:
            return (None, None) + (None,) * len(ctx.widths)
        if g.stride(1) != 1:
            g = g.contiguous()
        N, F_ = g.shape
        lib = _capi.load()
        outs, off = [], 0
        with torch.cuda.device(g.device):
            for w, need in zip(ctx.widths, ctx.needs_input_grad[2:]):
                if need and ctx.p == 0.0 and off % 4 == 0 and g.stride(0) % 4 == 0 and g.data_ptr() % 16 == 0:
                    outs.append(g[:, off:off + w])      # a plain concatenation: its gradient's column blocks, no copy
                elif need:
                    wp = (w + 3) // 4 * 4
                    go = torch.empty((N, wp), dtype=torch.float32, device=g.device)[:, :w]
                    _capi.check(lib.spgnn_cat_dropout(g.data_ptr(), g.stride(0), go.data_ptr(), go.stride(0), N, w, off, F_,
                                                      ctx.p, ctx.seed, _seed_off_ptr(g.device), 1, 0, _stream(g)), "spgnn_cat_dropout")
                    outs.append(go)
                else:
                    outs.append(None)
                off += w
        return (None, None) + tuple(outs)


def cat_dropout(tensors, p: float = 0.0, seed: int = 0) -> torch.Tensor:
    """dropout(cat(tensors, 1), p) with 16-byte-aligned rows; p = 0: a plain concatenation.  The result carries
    its split-GEMM operand scale (``_spgnn_scale``, from maxima the kernel collects while writing)."""
    _require_cuda(*tensors)
    y, scale = _CatDropout.apply(float(p), int(seed), *tensors)
    y._spgnn_scale = (y._version, scale)
    return y


def operand_scale(x: torch.Tensor) -> torch.Tensor:
    """Power-of-two GEMM scale of x: the one its producer attached (cat_dropout), else one absmax pass - remembered on
    the tensor when it is constant data (the batch's cached layer-0 input), so that pass runs once per batch."""
    tag = getattr(x, "_spgnn_scale", None)
    if tag is not None and tag[0] == x._version:
        return tag[1]
    sc = pow2_scale(x)
    if not x.requires_grad:
        x._spgnn_scale = (x._version, sc)
    return sc


GEMM_WIDE = False       # split GEMMs in their WIDE-RANGE form (include/spgnn_hip.h, SPGNN_GEMM_WIDE): lo kept as 2^11 lo, the cross products in a
                        # second accumulator set - 22 bits within ~2^28 of an operand's maximum instead of 2^18, in 128 x 128 tiles (slower).  Set
                        # between steps only (a step's forward and backward must agree); train.TrainStep(range_policy="auto") sets it when the
                        # range monitor reports operands outside the narrow envelope.
MIN_GEMM_ROWS = 512    # dense layers (nn.Linear inside GraphConv / GINConv / SAGEConv, the linear-mean output layer) with fewer rows go
                       # to torch.mm: a product of a few hundred rows is launch-bound either way.  Tests set it to 1 so that the
                       # 2-3-tree parity cases of rows D / E / F run the library's own matrix-core kernels too.
A_PRESPLIT = True      # constant node data (a model's first-layer input) goes to its products pre-split, once per loader batch


def mark_batch_constant(t: torch.Tensor) -> torch.Tensor:
    """``t`` is node DATA of the current loader batch (fvs, cat[fvs, pos_enc], the aligned pos_enc): constant over the
    reference's GCN_STEPS = 300 inner steps (job_runner.py:1892), so what the GEMMs derive from it - the power-of-two scale and
    the pre-split image - is made once per batch and kept on the tensor (:func:`const_operand`)."""
    t._spgnn_const = True
    return t


def const_operand(x: torch.Tensor, scale: torch.Tensor):
    """(operand, pre-split?) for ``x`` as the fp32 side of a product whose other operand is pre-split: the image
    :func:`presplit` makes of a batch constant under ``scale`` (built on first use, remembered with the tensor's version),
    else ``x`` itself."""
    if not (A_PRESPLIT and PRESPLIT_B and getattr(x, "_spgnn_const", False)) or x.requires_grad or not _rows_aligned(x):
        return x, False
    wide = bool(GEMM_WIDE)
    images = getattr(x, "_spgnn_aps", None)                  # {wide: (version, image, scale)}: one image per arithmetic form
    tag = images.get(wide) if images else None
    if tag is not None and tag[0] == x._version and tag[2] is scale:
        if CAPTURE_REFS is not None and torch.cuda.is_current_stream_capturing():
            CAPTURE_REFS.append((tag[1], scale))             # the recorded launches address the image: the step keeps it alive
        return tag[1], True
    if torch.cuda.is_current_stream_capturing():
        return x, False                      # never allocate a persistent image inside a capture (the warm-up steps made it)
    if INFERENCE:                            # (not torch.is_grad_enabled(): that is False inside EVERY autograd.Function.forward,
        return x, False                      # which is where the products are issued from - the training step never got its image)
                                             # a single inference pass (reference test.py): one product does not repay a split pass
    ps = presplit(x, scale=scale)[0]
    if images is None:
        images = x._spgnn_aps = {}
    images[wide] = (x._version, ps, scale)
    return ps, True


def refresh_batch_constant(t: torch.Tensor) -> None:
    """After ``t`` was rewritten IN PLACE with the next loader batch (arena.BatchArena): its scale and its pre-split image are
    recomputed into the tensors the captured step already addresses."""
    tag = getattr(t, "_spgnn_scale", None)
    if tag is not None:
        tag[1].copy_(pow2_scale(t))
        t._spgnn_scale = (t._version, tag[1])
    for wide, aps in (getattr(t, "_spgnn_aps", None) or {}).items():
        presplit(t, scale=aps[2], out=aps[1], wide=wide)
        t._spgnn_aps[wide] = (t._version, aps[1], aps[2])


BIAS_COLSUM = True     # spmm_sum's backward takes the bias gradient from the activation-backward pass (spgnn_act_bwd_colsum)
LINEAR_ACT_CLASSIFIER = True   # a skinny classifier behind Linear + activation joins that product's node (GIN's last MLP)
EPILOGUE_DROPOUT = True   # linear(drop=): the hash mask applied by the product's epilogue (ReLU / LeakyReLU / none)
EMIT_SCALES = True     # spmm_sum and ops.linear leave max |result| in a scale block (no absmax pass when the result feeds a product)
PRESPLIT_B = True      # weight operands of the NT products pre-split once per step (spgnn_presplit); False: fp32 rows, split per tile


def _tagged(w: torch.Tensor, name: str):
    """An attachment weight_cat left on ``w`` under ``name`` as ``(w._version, tensor)``, or None when it is absent or
    ``w`` was written since (the pre-split form and the scale must describe the same values)."""
    tag = getattr(w, name, None)
    return tag[1] if tag is not None and tag[0] == w._version else None


SKINNY_GEMM = True       # small inference batches: projections on spgnn_gemm_nt_skinny (16 x 64 tiles, k split over the waves, fp32 MFMA)
SKINNY_ROWS = 640        # ... up to this many rows (about four airway trees); the split-fp16 kernels tile for tens of thousands


INFERENCE = False        # set for the duration of a model call made under torch.no_grad() (inference_scope)


class inference_scope:
    """``with inference_scope():`` around a model / head call: ``ops.INFERENCE`` is True inside iff the CALLER has autograd off
    (``torch.no_grad()`` / ``torch.inference_mode()``).  torch.is_grad_enabled() cannot be asked further down: it is False
    inside every autograd.Function.forward, training or not."""

    def __enter__(self):
        global INFERENCE
        self.prev = INFERENCE
        INFERENCE = INFERENCE or not torch.is_grad_enabled()
        return self

    def __exit__(self, *exc):
        global INFERENCE
        INFERENCE = self.prev
        return False


def skinny_rows(rows: int) -> bool:
    """Whether a product with ``rows`` output rows belongs on the skinny kernel: inference only (a model called under
    torch.no_grad(): the per-scan forward of reference job_runner.py:2046-2052 - a training step keeps one arithmetic for its
    forward and backward products), few rows."""
    return bool(SKINNY_GEMM and INFERENCE and GEMM_MODE == "f16x3" and 0 < rows <= SKINNY_ROWS)


def gemm_nt_skinny(a: torch.Tensor, b: torch.Tensor, out: Optional[torch.Tensor] = None, bias: Optional[torch.Tensor] = None,
                   act: int = 0, score_l=None, score_r=None, score_out=None) -> torch.Tensor:
    """a (M, K) @ b (N, K)^T -> (M, N), fp32 operands with 16-byte rows, on spgnn_gemm_nt_skinny."""
    _require_cuda(a, b)
    M, K = a.shape
    N = b.shape[0]
    assert b.shape[1] == K and _rows_aligned(a) and _rows_aligned(b)
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    assert out.shape == (M, N) and out.stride(1) == 1
    with torch.cuda.device(a.device), _timed("gemm_nt_skinny", (M, N, K)):
        _capi.check(_capi.load().spgnn_gemm_nt_skinny(a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), out.data_ptr(), out.stride(0),
                                                      M, N, K, _ptr(bias), act, _ptr(score_l), _ptr(score_r), _ptr(score_out),
                                                      score_l.numel() if score_out is not None else 0, _stream(a)), "spgnn_gemm_nt_skinny")
    return out


def _b_operand(w: torch.Tensor, rows: Optional[int] = None):
    """(tensor, b_presplit) to pass as the ``b`` operand of gemm_nt for a weight operand: its pre-split form when
    weight_cat attached one (and ``w`` is unchanged since), else the fp32 rows themselves.  ``rows``: the product's row
    count - a small inference batch (:func:`skinny_rows`) takes the fp32 rows: its kernel computes in fp32."""
    if rows is not None and skinny_rows(rows):
        return w, False
    ps = _tagged(w, "_spgnn_ps") if PRESPLIT_B else None
    return (ps, True) if ps is not None else (w, False)


def _bt_operand(w: torch.Tensor, use_ps: bool):
    """(W^T with 16-byte rows or None, its pre-split form or None) as attached by weight_cat; ``use_ps`` is the decision the
    FORWARD product took (one form per autograd node, whatever PRESPLIT_B is by the time backward runs)."""
    return _tagged(w, "_spgnn_t"), (_tagged(w, "_spgnn_t_ps") if use_ps else None)


class _WeightPrep:
    """Persistent operand buffers and the device table of spgnn_weight_prep for one list of projection layers."""

    def __init__(self, specs, device):
        import ctypes
        lib = _capi.load()
        self.n = len(specs)
        self.entries, self.keep = [], []
        tab = (_capi.WeightPrepLayer * self.n)()
        first = 0
        for i, sp in enumerate(specs):
            w_a, w_b, want_t = sp[:3]
            cols = len(sp) > 3 and sp[3] == "cols"           # [w_a | w_b]: the aggregate-first layer's per-head operand
            R1, K = w_a.shape
            R2 = 0 if w_b is None else w_b.shape[0]
            if cols:
                Ka, K = K, K + (0 if w_b is None else w_b.shape[1])
                R = R1
            else:
                R = R1 + R2
            Kp, Rp = (K + 3) // 4 * 4, (R + 3) // 4 * 4
            dst = torch.empty((R, Kp), dtype=torch.float32, device=device)
            ps = torch.empty((R, Kp), dtype=torch.float32, device=device)
            dst_t = torch.empty((K, Rp), dtype=torch.float32, device=device) if want_t else None
            ps_t = torch.empty((K, Rp), dtype=torch.float32, device=device) if want_t else None
            scale = torch.empty((1,), dtype=torch.float32, device=device)
            t = tab[i]
            t.a, t.a_stride = w_a.data_ptr(), w_a.stride(0)
            t.b, t.b_stride = (w_b.data_ptr(), w_b.stride(0)) if w_b is not None else (0, 0)
            t.dst, t.ps, t.dst_stride = dst.data_ptr(), ps.data_ptr(), Kp
            t.dst_t, t.ps_t, t.dst_t_stride = (dst_t.data_ptr(), ps_t.data_ptr(), Rp) if want_t else (0, 0, 0)
            t.scale, t.first_block, t.rows_a, t.rows_b, t.K = scale.data_ptr(), first, R1, (Ka if cols else R2), K
            t.mode = (1 if cols else 0) | (2 if GEMM_WIDE else 0)          # bit 1: pre-split images in the wide-range form
            first += int(lib.spgnn_weight_prep_blocks(R, Kp, Rp if want_t else 0))
            self.entries.append((dst, ps, dst_t, ps_t, scale, (R1, R2, K, R)))
        self.blocks = first
        raw = bytes(memoryview(tab))
        self.table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(device)
        self.workspace = torch.empty((max(self.blocks, 1),), dtype=torch.float32, device=device)

    def run(self):
        with torch.cuda.device(self.table.device):
            _capi.check(_capi.load().spgnn_weight_prep(self.table.data_ptr(), self.n, self.blocks, self.workspace.data_ptr(),
                                                       _stream(self.table)), "spgnn_weight_prep")


_PREP_CACHE: dict = {}       # layout key -> _WeightPrep (buffers and table are reused by every forward pass)
_PREP_CACHE_MAX = 8          # unpinned entries kept (least recently used goes first)
CAPTURE_REFS = None          # a list while train.TrainStep.capture records a step: everything whose device addresses the
                             # HIP graph bakes in is appended, and the captured step keeps the list alive as long as its graphs


def _prep_lookup(cache: dict, key, make):
    """The cached operand set for one layout, created on a miss.  Eviction is LRU over entries that no HIP graph has
    recorded: a prep that ran while the stream was capturing is pinned (its buffers' addresses are inside the graph; a
    replay after a drop would read and write freed memory) and is also handed to the capturing step (CAPTURE_REFS), so
    the buffers live exactly as long as something can replay them."""
    prep = cache.pop(key, None)
    if prep is None:
        prep = make()
        prep.pinned = False
        loose = [k for k, v in cache.items() if not v.pinned]
        for k in loose[:max(0, len(loose) + 1 - _PREP_CACHE_MAX)]:
            del cache[k]
    cache[key] = prep                                   # re-inserted last: dict order is the recency order
    if torch.cuda.is_current_stream_capturing():
        prep.pinned = True
        if CAPTURE_REFS is not None:
            CAPTURE_REFS.append(prep)
    return prep
_PREP_ACTIVE: dict = {}      # (id(w_a), id(w_b)) -> (entry, want_t): installed for the duration of one model forward
BATCH_WEIGHT_PREP = True     # all projection layers' operands in one spgnn_weight_prep call per forward; False: per layer


class prepared_weights:
    """``with prepared_weights(specs):`` - ``specs`` = [(w_a, w_b or None, want_t[, "cols"]), ...] of every GATConv a
    forward pass will run (project-first layers: the row concatenation; ``"cols"``: the aggregate-first layer's
    [W_fc | W_res], whose head-h transpose is a column block of the transposed image).  One spgnn_weight_prep call builds all their GEMM operands ([W_fc; W_res] with 16-byte rows, its
    transpose, both pre-split, the scale); inside the block :func:`weight_cat` hands them out without launching anything.
    Values are those of the parameters at entry (the block must not update them)."""

    def __init__(self, specs):
        self.specs = [sp for sp in specs if sp[0].is_cuda and sp[0].dtype == torch.float32 and sp[0].stride(1) == 1
                      and (sp[1] is None or sp[1].stride(1) == 1)] if (BATCH_WEIGHT_PREP and GEMM_MODE == "f16x3") else []
        self.prev = None

    def __enter__(self):
        global _PREP_ACTIVE
        self.prev = _PREP_ACTIVE
        if not self.specs:
            return self
        dev = self.specs[0][0].device
        key = (str(dev), bool(GEMM_WIDE)) + tuple((sp[0].data_ptr(), sp[0].stride(0), tuple(sp[0].shape), 0 if sp[1] is None else sp[1].data_ptr(),
                                   0 if sp[1] is None else sp[1].stride(0), None if sp[1] is None else tuple(sp[1].shape), bool(sp[2]),
                                   sp[3] if len(sp) > 3 else "") for sp in self.specs)
        prep = _prep_lookup(_PREP_CACHE, key, lambda: _WeightPrep(self.specs, dev))
        fz = FROZEN_WEIGHTS
        if fz is not None and INFERENCE and fz.get(("prep", key)) is prep:
            pass                                         # frozen weights: the operand set built for this runner is current
        else:
            prep.run()
            if fz is not None and INFERENCE and not torch.cuda.is_current_stream_capturing():
                fz[("prep", key)] = prep                 # (keeps the buffers alive with the runner's capture)
        _PREP_ACTIVE = {(id(sp[0]), id(sp[1]) if sp[1] is not None else 0) + ((sp[3],) if len(sp) > 3 else ()): (e, bool(sp[2]))
                        for sp, e in zip(self.specs, prep.entries)}
        return self

    def __exit__(self, *exc):
        global _PREP_ACTIVE
        _PREP_ACTIVE = self.prev
        return False


class _WeightCat(torch.autograd.Function):
    """[w_a ; w_b] (rows of fc.weight, then of res_fc.weight) as ONE GEMM operand with 16-byte rows, its transpose (the
    operand of the input-gradient product) and its split-GEMM scale: one kernel + the scale kernel per layer and step
    (spgnn_weight_cat) instead of cat, pad, slice, absmax and a transposing copy.  Backward: the two row ranges of the
    incoming gradient as views - no kernels."""

    @staticmethod
    def forward(ctx, w_a, w_b, want_t: bool):
        ctx.set_materialize_grads(False)       # no zero tensors for the gradients of the scale / transpose outputs
        R1, K = w_a.shape
        R2 = 0 if w_b is None else w_b.shape[0]
        R, Kp, Rp = R1 + R2, (K + 3) // 4 * 4, (R1 + R2 + 3) // 4 * 4
        hit = _PREP_ACTIVE.get((id(w_a), id(w_b) if w_b is not None else 0))
        if hit is not None and (hit[1] or not want_t) and hit[0][5] == (R1, R2, K, R):
            dst, ps, dst_t, ps_t, scale, _ = hit[0]          # built by spgnn_weight_prep for this forward pass: nothing to launch
            ctx.rows = (R1, R2)
            outs = (dst[:, :K], scale, ps[:, :K]) + ((dst_t[:, :R], ps_t[:, :R]) if want_t else ())
            ctx.mark_non_differentiable(*outs[1:])
            return outs
        wa = w_a if w_a.stride(1) == 1 else w_a.contiguous()
        wb = None if w_b is None else (w_b if w_b.stride(1) == 1 else w_b.contiguous())
        buf = torch.empty((R, Kp), dtype=torch.float32, device=w_a.device)
        buf_t = torch.empty((K, Rp), dtype=torch.float32, device=w_a.device) if want_t else None
        lib = _capi.load()
        nb = lib.spgnn_weight_cat_partials(R, K, Kp, Rp if want_t else 0)
        part = torch.empty((nb,), dtype=torch.float32, device=w_a.device)
        with torch.cuda.device(w_a.device):
            _capi.check(lib.spgnn_weight_cat(wa.data_ptr(), wa.stride(0), R1, _ptr(wb), 0 if wb is None else wb.stride(0), R2, K,
                                             buf.data_ptr(), Kp, _ptr(buf_t), Rp if want_t else 0, part.data_ptr(),
                                             _stream(w_a)), "spgnn_weight_cat")
        out = buf[:, :K]
        out_t = buf_t[:, :R] if want_t else None
        # the operand's scale and the pre-split forms of W and W^T in ONE launch (it replaces the scale reduction)
        ps, ps_t, scale = presplit(out, partials=part, w2=out_t)
        ctx.rows = (R1, R2)
        outs = (out, scale, ps) + ((out_t, ps_t) if want_t else ())
        ctx.mark_non_differentiable(*outs[1:])
        return outs

    @staticmethod
    def backward(ctx, g, *_unused):
        R1, R2 = ctx.rows
        if g is None:
            return None, None, None
        return g[:R1], (g[R1:] if R2 else None), None


def weight_cat(w_a: torch.Tensor, w_b: Optional[torch.Tensor] = None, want_t: bool = True) -> torch.Tensor:
    """-> w_cat (R, K) view of a 16-byte-row buffer; carries ``_spgnn_scale`` (its GEMM scale), ``_spgnn_ps`` (its pre-split
    form, the ``b`` operand of the forward product) and, with ``want_t``, ``_spgnn_t`` = w_cat^T (K, R) with 16-byte rows
    and ``_spgnn_t_ps`` (pre-split: the ``b`` operand of the input-gradient product)."""
    _require_cuda(w_a, w_b)
    outs = _WeightCat.apply(w_a, w_b, want_t)
    w = outs[0]
    w._spgnn_scale = (w._version, outs[1])
    w._spgnn_ps = (w._version, outs[2])
    w._spgnn_t = (w._version, outs[3]) if want_t else None
    w._spgnn_t_ps = (w._version, outs[4]) if want_t else None
    return w


def cat_padded(tensors) -> torch.Tensor:
    return _CatPad.apply(*tensors)


class _GATLayerFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w_cat, w_lr, bias, csc: DeviceCSC, H: int, D: int, has_res: bool, slope: float, act: int,
                p_drop: float, seed: int, mean: bool):
        ctx.set_materialize_grads(False)       # no zero tensor for the unused gradient of `attn`
        HD = H * D
        x = _rowmajor(x)
        split = GEMM_MODE == "f16x3" and _rows_aligned(x) and _rows_aligned(w_cat) and x.shape[0] > 0
        if split:                                      # (N, HD [+HD]) = [ft | res] on the fp16 matrix cores
            s, sx = scores_fwd(x, w_lr, want_scale=True)   # (N, 2H) = [el | er]; the scale of x comes for free
            sw = operand_scale(w_cat)                      # attached by weight_cat, else one absmax pass
            wb, ps = _b_operand(w_cat, x.shape[0])
            y = gemm_nt(x, wb, sx, sw, b_presplit=ps)
        else:
            sx = sw = None
            ps = False
            y = torch.mm(x, w_cat.t())
            s = scores_fwd(x, w_lr)
        ft = y[:, :HD]
        res = y[:, HD:] if has_res else None
        out, out_mean, attn = gat_fwd_raw(csc, ft, s[:, :H], s[:, H:], res, bias, H, D, slope, act, p_drop, seed,
                                          mean=mean, need_out=(act != ACT_NONE))
        ctx.csc, ctx.cfg = csc, (H, D, has_res, slope, act, p_drop, seed, mean)
        ctx.has_bias = bias is not None
        ctx.w_t, ctx.w_t_ps = _bt_operand(w_cat, split and ps)     # W^T with 16-byte rows, written by weight_cat alongside W
        ctx.save_for_backward(x, w_cat, w_lr, y, s, attn, out if act != ACT_NONE else None, sx, sw)
        ctx.mark_non_differentiable(attn)
        return (out_mean if mean else out), attn

    @staticmethod
    def backward(ctx, g_out, _g_attn):
        if g_out is None:                      # only the (non-differentiable) attention output was used
            return (None,) * 13
        x, w_cat, w_lr, y, s, attn, out, sx, sw = ctx.saved_tensors
        H, D, has_res, slope, act, p_drop, seed, mean = ctx.cfg
        csc = ctx.csc
        HD = H * D
        N, K = x.shape
        g_out = _rowmajor(g_out)
        g_y = torch.empty_like(y)
        g_s = torch.empty_like(s)
        g_pre = g_y[:, HD:] if has_res else torch.empty((N, HD), dtype=torch.float32, device=x.device)
        split = sx is not None
        sg = new_scale_block(x.device) if split else None
        # [g_ft | g_pre] fills g_y when the layer has a residual; without one only g_ft does
        gat_bwd_raw(csc, y[:, :HD], s[:, :H], s[:, H:], attn, g_out, out, H, D, slope, act, p_drop, seed,
                    g_pre, g_y[:, :HD], g_s[:, :H], g_s[:, H:], mean=mean, absmax=sg, absmax_dst=has_res)
        need_bias = ctx.has_bias and ctx.needs_input_grad[3]
        g_bias = None
        g_wcat = None
        if ctx.needs_input_grad[1]:
            # tiny outputs (position stream, 39-wide inputs) leave the 128x128-tile kernel mostly idle: rocBLAS there
            big = g_y.shape[1] * K >= _TN_MIN_ELEMS
            if split and big:
                if need_bias and has_res:        # column sums of g_pre ride along with the operand stream
                    g_wcat, cs = gemm_tn(g_y, x, sg, sx, want_colsum=True)
                    g_bias = cs[HD:]
                else:
                    g_wcat = gemm_tn(g_y, x, sg, sx)
            else:
                g_wcat = _dw_gemm(g_y, x)
        if need_bias and g_bias is None:
            g_bias = g_pre.sum(0)
        g_wlr = scores_bwd_w(g_s, x) if ctx.needs_input_grad[2] else None
        g_x = None
        if ctx.needs_input_grad[0]:
            Kp = (K + 3) // 4 * 4                      # 16-byte rows for the vector kernels downstream
            g_x = torch.empty((N, Kp), dtype=torch.float32, device=x.device)[:, :K]
            if split:
                ps = ctx.w_t_ps is not None
                w_t = ctx.w_t_ps if ps else (ctx.w_t if ctx.w_t is not None else w_cat.t().contiguous())   # (K, C): an NT product with W^T
                J = g_s.shape[1]
                if J <= 32:                            # + g_S @ W_lr as an exact fp32 rank-2H update in the epilogue
                    w_lr_p = _padded_rows(w_lr, _pad16(K))
                    gemm_nt(g_y, w_t, sg, sw, out=g_x, upd_u=g_s, upd_v=w_lr_p, b_presplit=ps)   # W^T shares W's scale
                else:
                    gemm_nt(g_y, w_t, sg, sw, out=g_x, b_presplit=ps)
                    scores_bwd_x_(g_x, g_s, w_lr)
            else:
                torch.mm(g_y, w_cat, out=g_x)
                scores_bwd_x_(g_x, g_s, w_lr)
        return g_x, g_wcat, g_wlr, g_bias, None, None, None, None, None, None, None, None, None


def scores_from_parts(parts: torch.Tensor, H: int, D: int) -> torch.Tensor:
    N = parts.shape[0]
    s = torch.empty((N, 2 * H), dtype=torch.float32, device=parts.device)
    with torch.cuda.device(parts.device), _timed("scores_from_parts", (N, H, D)):
        _capi.check(_capi.load().spgnn_scores_from_parts(parts.data_ptr(), s.data_ptr(), s.stride(0), N, H, D, _stream(parts)),
                    "spgnn_scores_from_parts")
    return s


class _GATLayerScoresFromFtFn(torch.autograd.Function):
    """Project-first GATConv with el / er taken from ft = fc(x) as DGL does, (ft * attn_l).sum(-1), instead of from x
    through folded score weights: the products fall out of the projection GEMM's epilogue while the tile is in
    registers (64-column partials, summed per head by a tiny kernel), and in the backward pass the scores' gradient
    returns to g_ft inside spgnn_gat_bwd_src.  Every projected layer of the configs has K > H*D, so this reads less
    than the folded form (no pass over x for the scores, none for their weight gradient, no folding kernels)."""

    @staticmethod
    def forward(ctx, x, w_cat, attn_l, attn_r, bias, csc: DeviceCSC, H: int, D: int, has_res: bool, slope: float, act: int,
                p_drop: float, seed: int, mean: bool, sx, fuse):
        """``fuse`` = None, or (total, p, seed, extra): the output is written, already under the CONSUMER's feature
        dropout (p, seed), into columns [0, H*D) of a fresh (N, total) buffer - the next layer's input, whose remaining
        columns ``fill_cols_dropout`` adds in place - and that buffer is returned instead of the (N, H*D) rows, together
        with a third output: per-node maxima of the stored rows followed by ``extra`` free slots for the filler's maxima
        (the buffer's split-GEMM scale needs no extra pass).  The third output is the buffer's scale block."""
        ctx.set_materialize_grads(False)       # no zero tensor for the unused gradient of `attn`
        HD = H * D
        x = _rowmajor(x)
        N = x.shape[0]
        if sx is None:
            sx = pow2_scale(x)
        sw = operand_scale(w_cat)                  # attached by weight_cat, else one absmax pass
        ctx.attn_shape = attn_l.shape              # (H, D) or the parameter's own (1, H, D): no select / stack autograd nodes
        ctx.attn_params = (attn_l, attn_r)
        al, ar = attn_l.reshape(-1).contiguous(), attn_r.reshape(-1).contiguous()
        parts = torch.empty((N, HD // 64, 2), dtype=torch.float32, device=x.device)
        wb, ps = _b_operand(w_cat, N)
        ctx.w_t, ctx.w_t_ps = _bt_operand(w_cat, ps)
        xa, aps = const_operand(x, sx) if ps else (x, False)       # node data (a model's first layer): split once per loader batch
        ctx.x_ps = xa if aps else None
        y = gemm_nt(xa, wb, sx, sw, score_l=al, score_r=ar, score_out=parts, b_presplit=ps, a_presplit=aps)
        s = scores_from_parts(parts, H, D)
        ft = y[:, :HD]
        res = y[:, HD:] if has_res else None
        ctx.fused = None
        if fuse is not None:
            total, fp, fseed, extra = fuse
            assert not mean and total >= HD
            assert total % 4 == 0                  # (the buffer itself is returned, never a view of it: it is completed in place)
            buf = torch.empty((N, total), dtype=torch.float32, device=x.device)
            amax = new_scale_block(x.device)
            od = (float(fp), int(fseed), int(total), 0)
            out, _, attn = gat_fwd_raw(csc, ft, s[:, :H], s[:, H:], res, bias, H, D, slope, act, p_drop, seed, out=buf[:, :HD],
                                       out_drop=od, out_absmax=amax)
            # the stored rows are stashed, not saved: the buffer is completed IN PLACE afterwards (fill_cols_dropout writes
            # the other columns) and autograd's version check would reject a saved tensor over it; this function only ever
            # reads its own columns
            ctx.fused = (out.detach() if act != ACT_NONE else None, od)
            ctx.csc, ctx.cfg = csc, (H, D, has_res, slope, act, p_drop, seed, False)
            ctx.has_bias = bias is not None
            ctx.save_for_backward(x, w_cat, al, ar, y, s, attn, None, sx, sw)
            ctx.mark_non_differentiable(attn, amax)
            return buf, attn, amax
        out, out_mean, attn = gat_fwd_raw(csc, ft, s[:, :H], s[:, H:], res, bias, H, D, slope, act, p_drop, seed,
                                          mean=mean, need_out=(act != ACT_NONE))
        ctx.csc, ctx.cfg = csc, (H, D, has_res, slope, act, p_drop, seed, mean)
        ctx.has_bias = bias is not None
        ctx.save_for_backward(x, w_cat, al, ar, y, s, attn, out if act != ACT_NONE else None, sx, sw)
        ctx.mark_non_differentiable(attn)
        return (out_mean if mean else out), attn

    @staticmethod
    def backward(ctx, g_out, _g_attn, _g_amax=None):
        if g_out is None:                      # only the (non-differentiable) attention output was used
            return (None,) * 16
        x, w_cat, al, ar, y, s, attn, out, sx, sw = ctx.saved_tensors
        out_drop = None
        if ctx.fused is not None:              # g_out is the gradient of the whole (N, total) buffer: this layer's columns come first
            out, out_drop = ctx.fused
            g_out = g_out[:, :ctx.cfg[0] * ctx.cfg[1]]
        H, D, has_res, slope, act, p_drop, seed, mean = ctx.cfg
        csc = ctx.csc
        HD = H * D
        N, K = x.shape
        g_out = _rowmajor(g_out)
        g_y = torch.empty_like(y)
        g_s = torch.empty_like(s)
        g_pre = g_y[:, HD:] if has_res else torch.empty((N, HD), dtype=torch.float32, device=x.device)
        sg = new_scale_block(x.device)
        jobs = SumJobs(x.device)
        gat_bwd_raw(csc, y[:, :HD], s[:, :H], s[:, H:], attn, g_out, out, H, D, slope, act, p_drop, seed,
                    g_pre, g_y[:, :HD], g_s[:, :H], g_s[:, H:], mean=mean, absmax=sg, score_l=al, score_r=ar, out_drop=out_drop,
                    absmax_dst=has_res)
        need_bias = ctx.has_bias and ctx.needs_input_grad[4]
        g_bias = g_wcat = None

        def input_gradient():
            Kp = (K + 3) // 4 * 4
            gx = torch.empty((N, Kp), dtype=torch.float32, device=x.device)[:, :K]
            if ctx.w_t_ps is not None:
                gemm_nt(g_y, ctx.w_t_ps, sg, sw, out=gx, b_presplit=True)
            else:
                gemm_nt(g_y, ctx.w_t if ctx.w_t is not None else w_cat.t().contiguous(), sg, sw, out=gx)
            return gx
        g_x = None
        nt_first = ctx.needs_input_grad[0] and side_for(N, x.device) is not None
        if nt_first:                                   # the critical-path product first: the weight gradient then runs on the step's
            g_x = input_gradient()                     # side stream next to the following layer's traversals (SideLaunch)
        if ctx.needs_input_grad[1]:
            big = g_y.shape[1] * K >= _TN_MIN_ELEMS
            if big:
                xb, bps = (ctx.x_ps, True) if ctx.x_ps is not None else (x, False)
                if need_bias and has_res:
                    g_wcat, cs = gemm_tn(g_y, xb, sg, sx, want_colsum=True, defer=jobs, b_presplit=bps)
                    g_bias = cs[HD:]
                else:
                    g_wcat = gemm_tn(g_y, xb, sg, sx, defer=jobs, b_presplit=bps)
            else:
                g_wcat = _dw_gemm(g_y, x)
        if need_bias and g_bias is None:
            g_bias = g_pre.sum(0)
        g_al = g_ar = None
        if ctx.needs_input_grad[2] and ctx.needs_input_grad[3] and queue_attn_grads(g_s, y[:, :HD], H, *ctx.attn_params):
            pass                                       # the step's queue runs the pass with every other layer's (AttnGradQueue)
        elif ctx.needs_input_grad[2] or ctx.needs_input_grad[3]:
            m = scores_bwd_w(g_s, y[:, :HD], blockdiag_heads=H, defer=jobs)  # (2, H, D): [0] = g_attn_l, [1] = g_attn_r (contiguous views)
            g_al, g_ar = m[0].view(ctx.attn_shape), m[1].view(ctx.attn_shape)
        jobs.flush()                               # the two split-K reductions in one launch
        if ctx.needs_input_grad[0] and not nt_first:
            g_x = input_gradient()
        return g_x, g_wcat, g_al, g_ar, g_bias, None, None, None, None, None, None, None, None, None, None, None


def scores_from_ft_supported(x: torch.Tensor, w_cat: torch.Tensor, D: int) -> bool:
    return (GEMM_MODE == "f16x3" and D % 64 == 0 and x.dim() == 2 and x.shape[0] > 0 and x.dtype == torch.float32
            and _rows_aligned(w_cat))


def gat_layer_scores_from_ft(csc: DeviceCSC, x, w_cat, attn_l, attn_r, bias, H: int, D: int, has_res: bool, slope: float,
                             act: int, p_drop: float = 0.0, seed: int = 0, mean: bool = False, fuse=None):
    """Same contract as gat_layer, with the score vectors attn_l / attn_r (H, D) instead of folded score weights.
    ``fuse`` (see _GATLayerScoresFromFtFn.forward): -> (next layer's input buffer (N, total), attn, maxima)."""
    _require_cuda(x, w_cat, attn_l, attn_r, bias)
    xr = _rowmajor(x)
    sx = operand_scale(x) if xr is x and _rows_aligned(x) else None
    if not _rows_aligned(xr):
        xr = cat_padded((xr,))
    return _GATLayerScoresFromFtFn.apply(xr, w_cat, attn_l, attn_r, bias, csc, H, D, has_res, slope, act, p_drop, seed, mean, sx,
                                         fuse)


class _FillColsDropout(torch.autograd.Function):
    """buf[:, off:off+w] = dropout(src, p) IN PLACE, with the mask spgnn_cat_dropout gives column block ``off`` of a
    ``total``-wide concatenation under ``seed`` (the same seed as the block a fused GATConv already wrote into ``buf``): the
    second half of ``dropout(cat[h_s, h_p])`` (reference models.py:477-481) when the first half came straight from the
    producing layer.  Backward: the incoming gradient goes on to the producer unchanged (it reads its own columns only)
    and this block's gradient is the masked column block."""

    @staticmethod
    def forward(ctx, buf, src, off, total, p, seed, part):
        N, w = src.shape
        s_ = src if src.stride(1) == 1 else src.contiguous()
        base = buf._base if buf._base is not None else buf
        with torch.cuda.device(buf.device):
            _capi.check(_capi.load().spgnn_cat_dropout(s_.data_ptr(), s_.stride(0), buf.data_ptr(), buf.stride(0), N, w, off, total,
                                                       p, seed, _seed_off_ptr(buf.device), 0, _ptr(part), _stream(buf)),
                        "spgnn_cat_dropout")
        ctx.cfg = (off, w, total, p, seed)
        ctx.mark_dirty(buf)
        return buf

    @staticmethod
    def backward(ctx, g):
        off, w, total, p, seed = ctx.cfg
        if g is None:
            return (None,) * 7
        if g.stride(1) != 1:
            g = g.contiguous()
        g_src = None
        if ctx.needs_input_grad[1]:
            if p == 0.0 and off % 4 == 0 and g.stride(0) % 4 == 0 and g.data_ptr() % 16 == 0:
                g_src = g[:, off:off + w]
            else:
                N = g.shape[0]
                g_src = torch.empty((N, (w + 3) // 4 * 4), dtype=torch.float32, device=g.device)[:, :w]
                with torch.cuda.device(g.device):
                    _capi.check(_capi.load().spgnn_cat_dropout(g.data_ptr(), g.stride(0), g_src.data_ptr(), g_src.stride(0), N, w, off,
                                                               total, p, seed, _seed_off_ptr(g.device), 1, 0, _stream(g)),
                                "spgnn_cat_dropout")
        return g, g_src, None, None, None, None, None


def fill_cols_dropout(buf: torch.Tensor, src: torch.Tensor, off: int, total: int, p: float, seed: int,
                      amax: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Complete a fused GATConv's output buffer (see _GATLayerScoresFromFtFn.forward): columns [off, off + w) = dropout(src).
    ``amax``: the scale block the layer returned; it takes this block's maxima too and is attached to the result as the
    buffer's split-GEMM scale."""
    _require_cuda(buf, src)
    N, w = src.shape
    lib = _capi.load()
    y = _FillColsDropout.apply(buf, src, int(off), int(total), float(p), int(seed), amax)
    if amax is not None and N > 0:
        y._spgnn_scale = (y._version, amax)            # the layer's scale block, now holding this block's maxima too
    return y


def fused_extra_partials(N: int, width: int) -> int:
    """(kept for callers of the round-2 interface: the fused layer's maxima now live in one scale block, nothing to reserve)"""
    return 0


def gat_layer(csc: DeviceCSC, x, w_cat, w_lr, bias, H: int, D: int, has_res: bool, slope: float, act: int,
              p_drop: float = 0.0, seed: int = 0, mean: bool = False):
    """out (N, H*D) [or the head mean (N, D) when ``mean``], attn (E, H; CSC slot order, not differentiable)."""
    _require_cuda(x, w_cat, w_lr, bias)
    return _GATLayerFn.apply(x, w_cat, w_lr, bias, csc, H, D, has_res, slope, act, p_drop, seed, mean)


# --------------------------------------------------------------------------------------------
# one SPGNN level (structure GATConv + position GATConv, LSPE) in one traversal
# --------------------------------------------------------------------------------------------
def lspe_level_supported(csc: DeviceCSC, x_s: torch.Tensor, x_p: torch.Tensor, D: int) -> bool:
    """The fused level (spgnn_lspe_*) needs fp32 rows on a ROCm device, out_feats D in {64, 128, 256} for both layers, the
    padded neighbour rows and 1 <= degree <= 8 in both directions (airway trees: <= 5 + the self loop)."""
    return (x_s.is_cuda and x_s.dtype == torch.float32 and x_p.dtype == torch.float32 and x_s.dim() == 2 and x_s.shape[0] > 0
            and GEMM_MODE == "f16x3" and USE_ELL and getattr(csc, "num_dst", None) is None
            and 1 <= csc.min_in_degree and csc.max_in_degree <= 8 and 1 <= getattr(csc, "min_out_degree", 0)
            and csc.max_out_degree <= 8 and bool(_capi.load().spgnn_lspe_supported(int(D))))


LSPE_SCORES_IN_KERNEL = True     # spgnn_lspe_fwd sums el / er from the GEMMs' score partials itself; False: spgnn_scores_from_parts


class _LspeLevelFn(torch.autograd.Function):
    """One level of GATPSPGNN (reference models.py:472-484): the structure GATConv (2 heads) on x_s = dropout(cat[h_s, h_p]) and
    the position GATConv (1 head, tanh, residual) on x_p = dropout(h_p), both project-first with el / er from the projection
    GEMMs' epilogues, aggregated by ONE traversal of the graph (spgnn_lspe_fwd) that writes the next level's two inputs:
    buf (N, 3D) = [h_s' | h_p'] under the next structure layer's feature dropout and xp (N, D) = h_p' under the next position
    layer's.  Backward: spgnn_lspe_bwd_dst + spgnn_lspe_bwd_src straight into the GEMM-gradient buffers of both layers, then
    their weight / input gradient products."""

    @staticmethod
    def forward(ctx, x_s, x_p, w_s, w_p, al_s, ar_s, al_p, ar_p, bias_s, bias_p, csc: DeviceCSC, D: int, cfg: dict):
        ctx.set_materialize_grads(False)
        N = x_s.shape[0]
        E = csc.num_edges
        dev = x_s.device
        lib = _capi.load()
        ys, ss, parts, scales, wts, x_ps = [], [], [], [], [], []
        ctx.attn_shapes = (al_s.shape, al_p.shape)
        vecs = [(al_s.reshape(-1).contiguous(), ar_s.reshape(-1).contiguous()), (al_p.reshape(-1).contiguous(), ar_p.reshape(-1).contiguous())]
        prods = []
        for x, w, (al, ar), H in ((x_s, w_s, vecs[0], 2), (x_p, w_p, vecs[1], 1)):
            sx, sw = operand_scale(x), operand_scale(w)
            wb, ps = _b_operand(w, N)
            xa, aps = const_operand(x, sx) if ps else (x, False)         # level 0: node data, split once per loader batch
            pt = torch.empty((N, H * D // 64, 2), dtype=torch.float32, device=dev)
            prods.append(NtProblem(xa, wb, sx, sw, score_l=al, score_r=ar, score_out=pt, b_presplit=ps, a_presplit=aps))
            parts.append(pt); scales.append((sx, sw)); wts.append(_bt_operand(w, ps)); x_ps.append(xa if aps else None)
        ys = list(gemm_nt_pair(prods[0], prods[1]))        # the two projections: one launch
        for pt, H in zip(parts, (2, 1)):
            ss.append(scores_from_parts(pt, H, D) if not LSPE_SCORES_IN_KERNEL else torch.empty((N, 2 * H), dtype=torch.float32, device=dev))
        buf = torch.empty((N, 3 * D), dtype=torch.float32, device=dev)
        xp = torch.empty((N, D), dtype=torch.float32, device=dev)
        sc_buf, sc_xp = new_scale_block(dev), new_scale_block(dev)
        attn = [torch.empty((E, 2), dtype=torch.float32, device=dev), torch.empty((E, 1), dtype=torch.float32, device=dev)]
        nbr8 = csc.ell()[0]
        G = (_capi.LspeFwdGroup * 2)()
        for i, (H, has_res, bias) in enumerate(((2, cfg["res_s"], bias_s), (1, cfg["res_p"], bias_p))):
            y, s_ = ys[i], ss[i]
            g = G[i]
            g.ft, g.ft_stride = y.data_ptr(), y.stride(0)
            g.res, g.res_stride = (y[:, H * D:].data_ptr(), y.stride(0)) if has_res else (0, 0)
            g.bias = _ptr(bias)
            g.el, g.er, g.s_stride = s_.data_ptr(), s_[:, H:].data_ptr(), s_.stride(0)
            g.attn = attn[i].data_ptr()
            g.score_parts = parts[i].data_ptr() if LSPE_SCORES_IN_KERNEL else 0      # el / er summed (and written) by the kernel
            g.H, g.act, g.slope, g.p_drop, g.seed = H, cfg["act"][i], cfg["slope"][i], cfg["p_attn"][i], cfg["seed_attn"][i]
        with torch.cuda.device(dev), _timed("lspe_fwd", (N, E, D)):
            _capi.check(lib.spgnn_lspe_fwd(csc.indptr.data_ptr(), nbr8.data_ptr(), G, buf.data_ptr(), buf.stride(0), cfg["fp"], cfg["fseed"],
                                           xp.data_ptr(), xp.stride(0), cfg["fp2"], cfg["fseed2"], sc_buf.data_ptr(), sc_xp.data_ptr(),
                                           N, E, D, _seed_off_ptr(dev), _stream(x_s)), "spgnn_lspe_fwd")
        ctx.csc, ctx.D, ctx.cfg, ctx.wts, ctx.has_bias = csc, D, cfg, wts, (bias_s is not None, bias_p is not None)
        ctx.x_ps = x_ps                          # the pre-split images of constant inputs (not autograd tensors: batch data)
        ctx.attn_params = ((al_s, ar_s), (al_p, ar_p))     # the parameters themselves: a step's AttnGradQueue hands them their gradients
        ctx.save_for_backward(x_s, x_p, w_s, w_p, vecs[0][0], vecs[0][1], vecs[1][0], vecs[1][1], ys[0], ys[1], ss[0], ss[1], attn[0], attn[1],
                              scales[0][0], scales[0][1], scales[1][0], scales[1][1], buf, xp)
        ctx.mark_non_differentiable(attn[0], attn[1], sc_buf, sc_xp)
        return buf, xp, attn[0], attn[1], sc_buf, sc_xp

    @staticmethod
    def backward(ctx, g_buf, g_xp, *_unused):
        if g_buf is None and g_xp is None:
            return (None,) * 13
        (x_s, x_p, w_s, w_p, al_s, ar_s, al_p, ar_p, y_s, y_p, s_s, s_p, attn_s, attn_p, sx_s, sw_s, sx_p, sw_p, buf, xp) = ctx.saved_tensors
        csc, D, cfg = ctx.csc, ctx.D, ctx.cfg
        N, E = x_s.shape[0], csc.num_edges
        dev = x_s.device
        lib = _capi.load()
        if g_buf is None:
            g_buf = torch.zeros_like(buf)
        g_buf = _rowmajor(g_buf)
        if not _rows_aligned(g_buf):
            g_buf = g_buf.contiguous()
        if g_xp is not None:
            g_xp = _rowmajor(g_xp)
            if not _rows_aligned(g_xp):
                g_xp = g_xp.contiguous()
        xs, ws, ys, ss, attns = (x_s, x_p), (w_s, w_p), (y_s, y_p), (s_s, s_p), (attn_s, attn_p)
        vecs, scl = ((al_s, ar_s), (al_p, ar_p)), ((sx_s, sw_s), (sx_p, sw_p))
        Hs, res = (2, 1), (cfg["res_s"], cfg["res_p"])
        g_y = [torch.empty_like(y) for y in ys]
        g_s = [torch.empty_like(s_) for s_ in ss]
        g_e = [torch.empty((E, H), dtype=torch.float32, device=dev) for H in Hs]
        amax = [new_scale_block(dev) for _ in Hs]           # one block per layer: [g_ft | g_pre] is one operand
        g_pre = [g_y[i][:, Hs[i] * D:] if res[i] else torch.empty((N, Hs[i] * D), dtype=torch.float32, device=dev) for i in range(2)]
        nbr8, out_nbr8, out_pos8 = csc.ell()
        GD = (_capi.LspeBwdDstGroup * 2)()
        GS = (_capi.LspeBwdSrcGroup * 2)()
        for i, H in enumerate(Hs):
            d, q = GD[i], GS[i]
            d.ft, d.ft_stride = ys[i].data_ptr(), ys[i].stride(0)
            d.el, d.er, d.s_stride, d.attn = ss[i].data_ptr(), ss[i][:, H:].data_ptr(), ss[i].stride(0), attns[i].data_ptr()
            d.g_pre, d.g_pre_stride = g_pre[i].data_ptr(), g_pre[i].stride(0)
            d.g_e, d.g_er, d.gs_stride, d.absmax = g_e[i].data_ptr(), g_s[i][:, H:].data_ptr(), g_s[i].stride(0), amax[i].data_ptr()
            d.H, d.act, d.slope, d.p_drop, d.seed = H, cfg["act"][i], cfg["slope"][i], cfg["p_attn"][i], cfg["seed_attn"][i]
            q.attn, q.g_e, q.g_pre, q.g_pre_stride = attns[i].data_ptr(), g_e[i].data_ptr(), g_pre[i].data_ptr(), g_pre[i].stride(0)
            q.g_ft, q.g_ft_stride = g_y[i].data_ptr(), g_y[i].stride(0)
            q.g_el, q.g_er, q.gs_stride = g_s[i].data_ptr(), g_s[i][:, H:].data_ptr(), g_s[i].stride(0)
            q.score_l, q.score_r, q.absmax = vecs[i][0].data_ptr(), vecs[i][1].data_ptr(), amax[i].data_ptr()
            q.H, q.p_drop, q.seed = H, cfg["p_attn"][i], cfg["seed_attn"][i]
        with torch.cuda.device(dev):
            st = _stream(x_s)
            with _timed("lspe_bwd_dst", (N, E, D, int(g_xp is not None))):
                _capi.check(lib.spgnn_lspe_bwd_dst(csc.indptr.data_ptr(), nbr8.data_ptr(), GD, g_buf.data_ptr(), g_buf.stride(0), _ptr(g_xp),
                                                   g_xp.stride(0) if g_xp is not None else 0, buf.data_ptr(), buf.stride(0), cfg["fp"],
                                                   cfg["fseed"], xp.data_ptr(), xp.stride(0), cfg["fp2"], cfg["fseed2"], N, E, D,
                                                   _seed_off_ptr(dev), st), "spgnn_lspe_bwd_dst")
            with _timed("lspe_bwd_src", (N, E, D)):
                _capi.check(lib.spgnn_lspe_bwd_src(csc.out_indptr.data_ptr(), out_nbr8.data_ptr(), out_pos8.data_ptr(), GS, N, E, D,
                                                   _seed_off_ptr(dev), st), "spgnn_lspe_bwd_src")
        grads_x, grads_w, grads_al, grads_ar, grads_b = [None, None], [None, None], [None, None], [None, None], [None, None]
        jobs = SumJobs(dev)                        # the level's four split-K reductions run as one launch at the end
        tn, nt = [None, None], [None, None]        # the weight-gradient and input-gradient products of both layers: one launch each
        pair_splits = (None, None)
        if PAIR_GEMMS and TN_PAIR_SPLITS and all(ctx.needs_input_grad[2 + i] and g_y[i].shape[1] * xs[i].shape[1] >= _TN_MIN_ELEMS for i in range(2)) \
                and (ctx.x_ps[0] is not None) == (ctx.x_ps[1] is not None):
            pair_splits = tn_pair_splits(N, (g_y[0].shape[1], xs[0].shape[1]), (g_y[1].shape[1], xs[1].shape[1]), ctx.x_ps[0] is not None)
        for i, H in enumerate(Hs):
            HD = H * D
            x, K = xs[i], xs[i].shape[1]
            sx, sw = scl[i]
            sg = amax[i]
            need_bias = ctx.has_bias[i] and ctx.needs_input_grad[8 + i]
            if ctx.needs_input_grad[2 + i]:
                if g_y[i].shape[1] * K >= _TN_MIN_ELEMS:
                    xb = ctx.x_ps[i]
                    tn[i] = TnProblem(g_y[i], xb if xb is not None else x, sg, sx, want_colsum=bool(need_bias and res[i]), defer=jobs,
                                      b_presplit=xb is not None, splits=pair_splits[i])
                else:
                    grads_w[i] = _dw_gemm(g_y[i], x)
            if ctx.needs_input_grad[i]:
                gx = torch.empty((N, (K + 3) // 4 * 4), dtype=torch.float32, device=dev)[:, :K]
                w_t, w_t_ps = ctx.wts[i]
                if w_t_ps is not None:
                    nt[i] = NtProblem(g_y[i], w_t_ps, sg, sw, out=gx, b_presplit=True)
                else:
                    nt[i] = NtProblem(g_y[i], w_t if w_t is not None else ws[i].t().contiguous(), sg, sw, out=gx)
                grads_x[i] = gx
        def input_gradients():
            if nt[0] is not None and nt[1] is not None:
                gemm_nt_pair(nt[0], nt[1])
            else:
                for q in nt:
                    if q is not None:
                        q.run()
        # With a side stream for the weight gradients (SideLaunch) the input-gradient products go FIRST: they are on the critical
        # path, and the weight-gradient pair then starts behind them - next to the NEXT level's HBM-bound traversals, which is
        # where an MFMA-bound product overlaps (two products side by side only time-slice the CUs).
        nt_first = side_for(N, dev) is not None
        if nt_first:
            input_gradients()
        if tn[0] is not None and tn[1] is not None:
            r = gemm_tn_pair(tn[0], tn[1])
        else:
            r = [t.launch().finish() if t is not None else None for t in tn]
        for i, H in enumerate(Hs):
            if tn[i] is not None:
                if tn[i].want_colsum:
                    grads_w[i], cs = r[i]
                    grads_b[i] = cs[H * D:]
                else:
                    grads_w[i] = r[i]
            need_bias = ctx.has_bias[i] and ctx.needs_input_grad[8 + i]
            if need_bias and grads_b[i] is None:
                grads_b[i] = g_pre[i].sum(0)
        want_a = [ctx.needs_input_grad[4 + 2 * i] or ctx.needs_input_grad[5 + 2 * i] for i in range(2)]
        for i in range(2):                               # a training step runs these passes for ALL levels in one launch afterwards
            if want_a[i] and ctx.needs_input_grad[4 + 2 * i] and ctx.needs_input_grad[5 + 2 * i] and \
                    queue_attn_grads(g_s[i], ys[i][:, :Hs[i] * D], Hs[i], *ctx.attn_params[i]):
                want_a[i] = False
        if want_a[0] and want_a[1]:                      # both layers' attention-vector gradients: one streaming launch
            ms = scores_bwd_w_blockdiag_pair(g_s[0], ys[0][:, :Hs[0] * D], Hs[0], g_s[1], ys[1][:, :Hs[1] * D], Hs[1], jobs)
        else:
            ms = [scores_bwd_w(g_s[i], ys[i][:, :Hs[i] * D], blockdiag_heads=Hs[i], defer=jobs) if want_a[i] else None for i in range(2)]
        for i in range(2):
            if ms[i] is not None:
                grads_al[i], grads_ar[i] = ms[i][0].view(ctx.attn_shapes[i]), ms[i][1].view(ctx.attn_shapes[i])
        if not nt_first:
            input_gradients()
        jobs.flush()
        return (grads_x[0], grads_x[1], grads_w[0], grads_w[1], grads_al[0], grads_ar[0], grads_al[1], grads_ar[1], grads_b[0], grads_b[1],
                None, None, None)


def lspe_level(csc: DeviceCSC, x_s: torch.Tensor, x_p: torch.Tensor, w_s, w_p, attn_s, attn_p, bias_s, bias_p, D: int, cfg: dict):
    """-> (buf (N, 3D), xp (N, D)): the next level's structure and position inputs (see _LspeLevelFn), each carrying its
    split-GEMM operand scale.  ``w_s`` / ``w_p``: weight_cat operands [W_fc; W_res]; ``attn_s`` / ``attn_p``: (attn_l, attn_r)."""
    _require_cuda(x_s, x_p, w_s, w_p)
    N = x_s.shape[0]
    buf, xp, _a0, _a1, sc_buf, sc_xp = _LspeLevelFn.apply(x_s, x_p, w_s, w_p, attn_s[0], attn_s[1], attn_p[0], attn_p[1], bias_s, bias_p, csc,
                                                          int(D), cfg)
    buf._spgnn_scale = (buf._version, sc_buf)
    xp._spgnn_scale = (xp._version, sc_xp)
    return buf, xp


# --------------------------------------------------------------------------------------------
# aggregate-first GAT layer (input narrower than one head's output)
# --------------------------------------------------------------------------------------------
def agg_first_supported(H: int, F_in: int) -> bool:
    return bool(_capi.load().spgnn_gat_agg_supported(H, F_in))


def gat_agg_fwd_raw(csc: DeviceCSC, x, el, er, H: int, slope: float, p_drop: float, seed: int, with_x_copy: bool,
                    out: Optional[torch.Tensor] = None, x_tail: bool = False):
    """-> (z (N, H*zs), attn (E,H), scale block of z); head h's block: [z_h | x] (zs = 2F) or [z_h] (zs = F).  ``out``: a
    buffer at least (N, H*zs) wide to write the blocks into (its row stride is used).  ``x_tail`` (with zs = F): one copy of
    x behind the last block, [z_0 | ... | z_{H-1} | x] (``out`` at least (N, (H+1) F)); the scale block covers it."""
    N, E = csc.num_nodes, csc.num_edges
    F_ = x.shape[1]
    zs = 2 * F_ if with_x_copy else F_
    assert not (x_tail and with_x_copy)
    z = out if out is not None else torch.empty((N, H * zs + (F_ if x_tail else 0)), dtype=torch.float32, device=x.device)
    attn = torch.empty((E, H), dtype=torch.float32, device=x.device)
    amax = new_scale_block(x.device)
    with torch.cuda.device(x.device), _timed("gat_agg_fwd", (N, E, H, F_, 2 if x_tail else int(with_x_copy))):
        _capi.check(_capi.load().spgnn_gat_agg_fwd(csc.indptr.data_ptr(), csc.indices.data_ptr(), x.data_ptr(), x.stride(0),
                                                   el.data_ptr(), er.data_ptr(), el.stride(0), attn.data_ptr(), z.data_ptr(),
                                                   z.stride(0), zs, -2 if x_tail else (F_ if with_x_copy else -1), amax.data_ptr(), N, E, H, F_,
                                                   slope, p_drop, seed, _seed_off_ptr(x.device), _stream(x)),
                    "spgnn_gat_agg_fwd")
    return z, attn, amax


def head_mean(out: torch.Tensor, H: int, D: int) -> torch.Tensor:
    N = out.shape[0]
    om = torch.empty((N, D), dtype=torch.float32, device=out.device)
    with torch.cuda.device(out.device), _timed("head_mean", (N, H, D)):
        _capi.check(_capi.load().spgnn_head_mean(out.data_ptr(), out.stride(0), om.data_ptr(), om.stride(0), N, H, D,
                                                 _stream(out)), "spgnn_head_mean")
    return om


def act_bwd(g_out: torch.Tensor, out: Optional[torch.Tensor], H: int, D: int, act: int, mean: bool):
    """-> (g_pre (N, H*D), its scale block)."""
    N = g_out.shape[0]
    g_pre = torch.empty((N, H * D), dtype=torch.float32, device=g_out.device)
    amax = new_scale_block(g_out.device)
    with torch.cuda.device(g_out.device), _timed("act_bwd", (N, H, D, act, int(mean))):
        _capi.check(_capi.load().spgnn_act_bwd(g_out.data_ptr(), g_out.stride(0), int(mean), _ptr(out),
                                               out.stride(0) if out is not None else 0, g_pre.data_ptr(), g_pre.stride(0),
                                               amax.data_ptr(), N, H, D, act, _stream(g_out)), "spgnn_act_bwd")
    return g_pre, amax


# The classifier's weight gradient riding in spgnn_act_bwd_proj (spgnn_act_bwd_proj_wgrad: no second pass over the head mean).
# Built, bit-comparable (tests/test_hip_models.py), measured NEUTRAL in one process (tools/step_toggle_ab.py, MI355X, round 5;
# profiles/r05_act_bwd_proj_wgrad_ab.txt): st_pgat_spgnn_3 4.965 / 5.010 vs 4.980 ms at 512 trees, 1.000 vs 0.995 at 64, st_gin_3
# 3.609 vs 3.608 - the 88 extra accumulator registers (250 VGPRs: two waves per SIMD) and the doubled FMA count cost the pass what
# the removed launch (0.09 ms) saved.  Off by default: the two-pass form has the smaller kernel.
ACT_BWD_PROJ_WGRAD = False


def act_bwd_proj(g_s: torch.Tensor, w: torch.Tensor, out: Optional[torch.Tensor], H: int, D: int, act: int,
                 wgrad_jobs: Optional["SumJobs"] = None, rows: Optional["LossRows"] = None):
    """act_bwd of a mean-over-heads layer whose mean feeds a skinny Linear (weight ``w`` (J, D)), with that Linear's
    input gradient g_s @ w formed on the fly -> (g_pre (N, H*D), its scale block).  With ``wgrad_jobs`` (a SumJobs queue) and
    a supported shape the pass also forms that Linear's WEIGHT gradient g_s^T @ mean_h(out) from the rows it reads anyway
    (spgnn_act_bwd_proj_wgrad; the per-block partials are summed by the queue): -> (g_pre, block, g_w (J, D))."""
    N, J = g_s.shape
    lib = _capi.load()
    if rows is not None:
        # ``g_s`` / ``out`` in node order, the result one row per LISTED node (zero rows behind the count)
        assert wgrad_jobs is None and N == rows.N
        g_pre = torch.empty((rows.cap, H * D), dtype=torch.float32, device=g_s.device)
        part = new_scale_block(g_s.device)
        with torch.cuda.device(g_s.device), _timed("act_bwd_proj", (rows.cap, H, D, act, J)):
            _capi.check(lib.spgnn_act_bwd_proj_rows(g_s.data_ptr(), g_s.stride(0), J, w.data_ptr(), w.stride(0), _ptr(out),
                                                    out.stride(0) if out is not None else 0, rows.idx.data_ptr(), rows.cnt.data_ptr(),
                                                    g_pre.data_ptr(), g_pre.stride(0), part.data_ptr(), rows.cap, H, D, act,
                                                    _stream(g_s)), "spgnn_act_bwd_proj_rows")
        return g_pre, part
    g_pre = torch.empty((N, H * D), dtype=torch.float32, device=g_s.device)
    part = new_scale_block(g_s.device)
    if wgrad_jobs is not None:
        blocks = int(lib.spgnn_act_bwd_proj_wgrad_blocks(N))
        wpart = torch.empty((blocks, J, D), dtype=torch.float32, device=g_s.device)
        g_w = torch.empty((J, D), dtype=torch.float32, device=g_s.device)
        with torch.cuda.device(g_s.device), _timed("act_bwd_proj", (N, H, D, act, J)):
            _capi.check(lib.spgnn_act_bwd_proj_wgrad(g_s.data_ptr(), g_s.stride(0), J, w.data_ptr(), w.stride(0), out.data_ptr(),
                                                     out.stride(0), g_pre.data_ptr(), g_pre.stride(0), part.data_ptr(), wpart.data_ptr(),
                                                     N, H, D, act, _stream(g_s)), "spgnn_act_bwd_proj_wgrad")
        wgrad_jobs.add(_capi.SumJob(kind=0, splits=blocks, partials=wpart.data_ptr(), split_stride=J * D, out=g_w.data_ptr(), n=J * D),
                       wpart, g_w)
        return g_pre, part, g_w
    with torch.cuda.device(g_s.device), _timed("act_bwd_proj", (N, H, D, act, J)):
        _capi.check(lib.spgnn_act_bwd_proj(g_s.data_ptr(), g_s.stride(0), J, w.data_ptr(), w.stride(0), _ptr(out),
                                           out.stride(0) if out is not None else 0, g_pre.data_ptr(), g_pre.stride(0),
                                           part.data_ptr(), N, H, D, act, _stream(g_s)), "spgnn_act_bwd_proj")
    return g_pre, part


def act_bwd_proj_supported(H: int, D: int, J: int, w: torch.Tensor) -> bool:
    return H <= 4 and D % 4 == 0 and D <= 1024 and J <= 32 and w.stride(1) == 1 and _rows_aligned(w)


def act_bwd_proj_wgrad_supported(H: int, D: int, J: int, act: int, out: Optional[torch.Tensor]) -> bool:
    return bool(ACT_BWD_PROJ_WGRAD and H in (1, 2) and J <= 24 and act != ACT_NONE and out is not None and (J * D) % 4 == 0)


class _GATAggFirstFn(torch.autograd.Function):
    """GATConv with the projection AFTER the aggregation: out_h = act([z_h | x] @ [W_fc,h | W_res,h]^T + b_h),
    z_h[v] = sum_u a_h(u,v) x[u].  Same function as _GATLayerFn; chosen when the input is narrower than one head's
    output, so that the (N, 2*H*D) projected rows are neither written nor gathered."""

    @staticmethod
    def forward(ctx, x, w_fc, w_res, w_lr, bias, w_cls, b_cls, csc: DeviceCSC, H: int, D: int, slope: float, act: int,
                p_drop: float, seed: int, mean: bool):
        """``w_cls`` (J, D) / ``b_cls`` (J,), optional, with ``mean``: the classifier on the head mean (reference
        ``gnn_out``) joins this node, third output = logits; its input gradient then never exists as a tensor
        (backward: spgnn_act_bwd_proj)."""
        ctx.set_materialize_grads(False)       # no zero tensors for the unused gradients of `attn` / mean / logits
        x = _rowmajor(x)
        N, F_ = x.shape
        has_res = w_res is not None
        s = scores_fwd(x, w_lr)
        z, attn, sz = gat_agg_fwd_raw(csc, x, s[:, :H], s[:, H:], H, slope, p_drop, seed, has_res)
        zs = z.shape[1] // H
        # a loss-rows step (LOSS_ROWS): everything behind the aggregation runs on the listed rows only - M rows from here on,
        # outputs and saved tensors included; z's scale block stays the one of all rows (a maximum over a superset)
        rows = LOSS_ROWS if (LOSS_ROWS is not None and w_cls is not None and mean and LOSS_ROWS.N == N and z.shape[1] % 4 == 0
                             and _rows_aligned(z)) else None
        ctx.rows = rows
        if rows is not None:
            rows.used = True
            if rows.forward:
                z = gather_rows(z, rows)
                N = rows.cap
        hit = _PREP_ACTIVE.get((id(w_fc), id(w_res) if has_res else 0, "cols")) if PRESPLIT_B else None
        ctx.wt = None
        if hit is not None and hit[0][5][2] == zs and hit[0][5][3] == H * D and zs % 4 == 0 and D % 4 == 0:
            # [W_fc,h | W_res,h], its scale, its pre-split form and the transposes: built by this pass's spgnn_weight_prep
            dst, ps, dst_t, ps_t, sw, _ = hit[0]
            wc, wc_ps = dst.view(H, D, zs), ps.view(H, D, zs)
            ctx.wt = (dst_t, ps_t) if hit[1] else None
        else:
            w3 = w_fc.view(H, D, F_)
            wc = torch.cat([w3, w_res.view(H, D, F_)], dim=2).contiguous() if has_res else w3.contiguous()   # (H, D, zs)
            sw = pow2_scale(wc.view(H * D, zs))
            wc_ps = presplit(wc.view(H * D, zs), scale=sw)[0].view(H, D, zs) if (PRESPLIT_B and zs % 4 == 0) else None   # the products' b operand, split once
        out = torch.empty((N, H * D), dtype=torch.float32, device=x.device)
        fuse_mean = mean and headmean_fusable(out, H, D)
        rst = None
        wb, ps = (wc_ps, True) if (wc_ps is not None and not skinny_rows(N)) else (wc, False)
        for h in range(H):
            bh_ = bias[h * D:(h + 1) * D] if bias is not None else None
            if fuse_mean and h == 1:                   # the second head's tiles also write 0.5 * (head 0 + head 1)
                rst = torch.empty((N, D), dtype=torch.float32, device=x.device)
                gemm_nt_headmean(z[:, zs:2 * zs], wb[1], sz, sw, out[:, D:2 * D], out[:, :D], rst, bias=bh_, act=act, b_presplit=ps)
            else:
                gemm_nt(z[:, h * zs:(h + 1) * zs], wb[h], sz, sw, out=out[:, h * D:(h + 1) * D], bias=bh_, act=act, b_presplit=ps)
        ctx.csc, ctx.cfg = csc, (H, D, has_res, slope, act, p_drop, seed, mean)
        ctx.has_bias, ctx.presplit = bias is not None, ps
        if rst is None:
            rst = head_mean(out, H, D) if mean else out
        has_cls = w_cls is not None and mean
        logits = None
        ctx.head = None
        if has_cls:
            head = LOSS_HEAD
            if (head is not None and not head.used and (rows is None or not rows.forward) and rst.shape[0] == head.labels.shape[0]
                    and classifier_ce_supported(rst, w_cls)):
                # the step's loss joins this node: logits, loss sums, logit gradient and the classifier's own gradients from one
                # pass over the head mean (spgnn_classifier_ce); backward starts from head.g_logits
                logits, head.g_logits, wpart, colsum = classifier_ce(rst, w_cls, b_cls, head)
                head.used = True
                ctx.head = (wpart, colsum)
            else:
                logits = scores_fwd(rst, w_cls, bias=b_cls)
        ctx.has_cls, ctx.has_cls_bias = has_cls, has_cls and b_cls is not None
        ctx.save_for_backward(x, wc, w_lr, s, attn, z, out if act != ACT_NONE else None, sz, sw,
                              rst if has_cls else None, w_cls if has_cls else None)
        if has_cls:
            ctx.mark_non_differentiable(attn)
            return rst, attn, logits
        empty = x.new_empty(0)
        ctx.mark_non_differentiable(attn, empty)       # one call: a second one would replace the first
        return rst, attn, empty

    @staticmethod
    def backward(ctx, g_out, _g_attn, g_logits):
        if g_out is None and (g_logits is None or not ctx.has_cls):   # only the (non-differentiable) attention output was used
            return (None,) * 15
        x, wc, w_lr, s, attn, z, out, sz, sw, rst, w_cls = ctx.saved_tensors
        H, D, has_res, slope, act, p_drop, seed, mean = ctx.cfg
        csc = ctx.csc
        N, F_ = x.shape
        E = csc.num_edges
        zs = z.shape[1] // H
        rows = ctx.rows                        # a loss-rows step: z, out, rst, g_logits and g_pre have one row per listed node
        compact = rows is not None and rows.forward        # (or, list used by the backward pass only: g_pre and z from here on)
        g_wcls = g_bcls = None
        # every split-K reduction of this node in one launch at the end - its own, not the step's: fc.weight receives a second
        # gradient through the folded score projection (fold_scores' backward), which autograd ADDS to this one on arrival
        jobs = SumJobs(x.device, local=True)
        if ctx.has_cls and g_logits is not None:
            head, ctx.head = ctx.head, None
            cs_ = None
            if head is None:
                cs_ = column_sums(g_logits) if (ctx.has_cls_bias and ctx.needs_input_grad[6]) else None
            elif ctx.has_cls_bias and ctx.needs_input_grad[6]:
                cs_ = head[1]                                       # the loss kernel's own column sums
            g_logits = g_logits.contiguous()
            fused = g_out is None and act_bwd_proj_supported(H, D, g_logits.shape[1], w_cls)
            ride = head is None and fused and ctx.needs_input_grad[5] and act_bwd_proj_wgrad_supported(H, D, g_logits.shape[1], act, out)
            if ctx.needs_input_grad[5] and head is not None:
                # g_logits^T rst: the per-workgroup partials are there since the forward pass, only their sum is left
                wpart = head[0]
                Bp, Jc, Kpc = wpart.shape
                if Bp == 1:
                    g_wcls = wpart[0][:, :rst.shape[1]]
                else:
                    g_full = torch.empty((Jc, Kpc), dtype=torch.float32, device=x.device)
                    jobs.add(_capi.SumJob(kind=0, splits=Bp, partials=wpart.data_ptr(), split_stride=Jc * Kpc, out=g_full.data_ptr(), n=Jc * Kpc),
                             wpart, g_full)
                    g_wcls = g_full[:, :rst.shape[1]]
                    del g_full
                del wpart, head
            elif ctx.needs_input_grad[5] and not ride:
                g_wcls = scores_bwd_w(g_logits, rst, defer=jobs)
            if cs_ is not None:
                g_bcls = cs_
            if ride:
                # ... and the classifier's weight gradient g_logits^T mean_h(out) from the same rows: no second pass over the head mean
                g_pre, amax, g_wcls = act_bwd_proj(g_logits, w_cls, out, H, D, act, wgrad_jobs=jobs)
            elif fused and rows is not None and not rows.forward:
                # dense forward, list in the backward pass: the rows of g_pre outside the mask are exactly zero (their g_logits
                # rows are) - form the listed ones only, and run the layer's two products on them
                g_pre, amax = act_bwd_proj(g_logits, w_cls, out, H, D, act, rows=rows)
                z = gather_rows(z, rows)
                compact = True
            elif fused:
                # the usual training case (only the logits reach the loss): g_mean = g_logits W is formed inside act_bwd
                g_pre, amax = act_bwd_proj(g_logits, w_cls, out, H, D, act)
            else:
                g_mean = torch.empty_like(rst) if g_out is None else _rowmajor(g_out).clone()
                scores_bwd_x_(g_mean, g_logits, w_cls, accumulate=g_out is not None)
                g_pre, amax = act_bwd(g_mean, out, H, D, act, mean)
        else:
            g_pre, amax = act_bwd(_rowmajor(g_out), out, H, D, act, mean)
        sg = amax
        need_w = ctx.needs_input_grad[1] or (has_res and ctx.needs_input_grad[2])
        need_bias = ctx.has_bias and ctx.needs_input_grad[4]
        g_z = torch.empty_like(z)
        g_wfc = torch.empty((H * D, F_), dtype=torch.float32, device=x.device) if need_w else None
        g_wres = torch.empty((H * D, F_), dtype=torch.float32, device=x.device) if (need_w and has_res) else None
        g_bias = torch.empty((H * D,), dtype=torch.float32, device=x.device) if need_bias else None
        ps = ctx.presplit and D % 4 == 0       # the forward's decision: one operand form per autograd node
        if ctx.wt is not None and ps:          # head h's W^T = columns [h D, (h + 1) D) of the prepared transpose
            wct = [ctx.wt[1][:zs, h * D:(h + 1) * D] for h in range(H)]
        else:
            wct = wc.transpose(1, 2).contiguous()          # (H, zs, D): every head's W^T in one copy
            if ps:
                wct = presplit(wct.view(H * zs, D), scale=sw)[0].view(H, zs, D)
        nts, tns = [], []
        for h in range(H):
            gp_h = g_pre[:, h * D:(h + 1) * D]
            nts.append(NtProblem(gp_h, wct[h], sg, sw, out=g_z[:, h * zs:(h + 1) * zs], b_presplit=ps))
            if need_w:                                 # [g_W_fc,h | g_W_res,h] (D, 2F) straight into the two parameters' row blocks
                tns.append(TnProblem(gp_h, z[:, h * zs:(h + 1) * zs], sg, sz, want_colsum=need_bias, out=g_wfc[h * D:(h + 1) * D],
                                     out2=g_wres[h * D:(h + 1) * D] if has_res else None,
                                     colsum_out=g_bias[h * D:(h + 1) * D] if need_bias else None, defer=jobs))
        # the heads' products are independent and equal in shape: two per launch (the second head's tiles fill the first's
        # last round: 2 x 3.5 rounds of 128 x 128 tiles become 7)
        for q in range(0, H - 1, 2):
            gemm_nt_pair(nts[q], nts[q + 1])
            if tns:
                gemm_tn_pair(tns[q], tns[q + 1])
        if H % 2:
            nts[-1].run()
            if tns:
                tns[-1].launch().finish()
        if need_bias and not need_w:
            g_bias = g_pre.sum(0)
        # (a loss-rows step: g_z stays one row per LISTED node - the two traversals read it through rows.inv, a node outside the
        # list has a zero row; LIST_AWARE_TRAVERSALS off: the expanded copy in node order, as the dense step has it)
        listed_gz = compact and LIST_AWARE_TRAVERSALS
        if compact and not listed_gz:
            g_z = expand_rows(g_z, rows)
        g_s = torch.empty_like(s)
        g_e = torch.empty((E, H), dtype=torch.float32, device=x.device)
        lib = _capi.load()
        with torch.cuda.device(x.device):
            st = _stream(x)
            with _timed("gat_agg_bwd_dst", (N, E, H, F_)):
                if listed_gz:
                    _capi.check(lib.spgnn_gat_agg_bwd_dst_rows(csc.indptr.data_ptr(), csc.indices.data_ptr(), x.data_ptr(), x.stride(0),
                                                               s.data_ptr(), s[:, H:].data_ptr(), s.stride(0), attn.data_ptr(),
                                                               g_z.data_ptr(), g_z.stride(0), zs, rows.inv.data_ptr(), g_e.data_ptr(),
                                                               g_s[:, H:].data_ptr(), g_s.stride(0), N, E, H, F_, slope, p_drop, seed,
                                                               _seed_off_ptr(x.device), st), "spgnn_gat_agg_bwd_dst_rows")
                else:
                    _capi.check(lib.spgnn_gat_agg_bwd_dst(csc.indptr.data_ptr(), csc.indices.data_ptr(), x.data_ptr(), x.stride(0),
                                                          s.data_ptr(), s[:, H:].data_ptr(), s.stride(0), attn.data_ptr(),
                                                          g_z.data_ptr(), g_z.stride(0), zs, g_e.data_ptr(), g_s[:, H:].data_ptr(),
                                                          g_s.stride(0), N, E, H, F_, slope, p_drop, seed,
                                                          _seed_off_ptr(x.device), st), "spgnn_gat_agg_bwd_dst")
            need_x = ctx.needs_input_grad[0]
            Fp = (F_ + 3) // 4 * 4
            g_x = torch.empty((N, Fp), dtype=torch.float32, device=x.device)[:, :F_]
            w_lr_c = w_lr.contiguous()
            with _timed("gat_agg_bwd_src", (N, E, H, F_)):
                if listed_gz:
                    _capi.check(lib.spgnn_gat_agg_bwd_src_rows(csc.out_indptr.data_ptr(), csc.out_indices.data_ptr(),
                                                               csc.out_pos.data_ptr(), attn.data_ptr(), g_e.data_ptr(), g_z.data_ptr(),
                                                               g_z.stride(0), zs, F_ if has_res else -1, rows.inv.data_ptr(),
                                                               g_s[:, H:].data_ptr(), w_lr_c.data_ptr(), w_lr_c.stride(0), g_x.data_ptr(),
                                                               g_x.stride(0), g_s.data_ptr(), g_s.stride(0), N, E, H, F_, p_drop, seed,
                                                               _seed_off_ptr(x.device), st), "spgnn_gat_agg_bwd_src_rows")
                else:
                    _capi.check(lib.spgnn_gat_agg_bwd_src(csc.out_indptr.data_ptr(), csc.out_indices.data_ptr(),
                                                          csc.out_pos.data_ptr(), attn.data_ptr(), g_e.data_ptr(), g_z.data_ptr(),
                                                          g_z.stride(0), zs, F_ if has_res else -1, g_s[:, H:].data_ptr(),
                                                          w_lr_c.data_ptr(), w_lr_c.stride(0), g_x.data_ptr(), g_x.stride(0),
                                                          g_s.data_ptr(), g_s.stride(0), N, E, H, F_, p_drop, seed,
                                                          _seed_off_ptr(x.device), st), "spgnn_gat_agg_bwd_src")
        g_wlr = scores_bwd_w(g_s, x, defer=jobs) if ctx.needs_input_grad[3] else None
        jobs.flush()
        return ((g_x if need_x else None), g_wfc, g_wres, g_wlr, g_bias, g_wcls, g_bcls, None, None, None, None, None, None,
                None, None)


class _GATAggregateFn(torch.autograd.Function):
    """x (N, F) -> Zx = [z_0 | ... | z_{H-1} | x] (N, (H+1) F), z_h[v] = sum_u a_h(u, v) x[u]: the attention-weighted sums
    of the INPUT rows for every head, next to the rows themselves.  What an output GATConv WITHOUT activation needs when
    its heads are averaged (reference models.py:320-327: ``self.gat_layers[-1](g, h).mean(1)``): the layer is then linear
    in Zx,  mean_h(W_h z_h + Wres_h x + b_h) = Zx [W_0 | ... | W_{H-1} | sum_h Wres_h]^T / H + mean_h b_h,
    ONE product with (H+1) F reduction columns instead of H products of 2 F, and no (N, H*D) tensor exists at all -
    neither the per-head outputs nor their gradients."""

    @staticmethod
    def forward(ctx, x, w_lr, csc: DeviceCSC, H: int, slope: float, p_drop: float, seed: int):
        ctx.set_materialize_grads(False)
        x = _rowmajor(x)
        N, F_ = x.shape
        s = scores_fwd(x, w_lr)
        zx = torch.empty((N, (H + 1) * F_), dtype=torch.float32, device=x.device)
        _, attn, amax = gat_agg_fwd_raw(csc, x, s[:, :H], s[:, H:], H, slope, p_drop, seed, False, out=zx, x_tail=True)
        ctx.csc, ctx.cfg, ctx.scale_block = csc, (H, slope, p_drop, seed), amax      # x's copy is the kernel's own: its maxima too
        ctx.save_for_backward(x, w_lr, s, attn)
        ctx.mark_non_differentiable(attn)
        return zx, attn

    @staticmethod
    def backward(ctx, g_zx, _g_attn):
        if g_zx is None:
            return (None,) * 7
        x, w_lr, s, attn = ctx.saved_tensors
        H, slope, p_drop, seed = ctx.cfg
        csc = ctx.csc
        N, F_ = x.shape
        E = csc.num_edges
        g_zx = _rowmajor(g_zx)
        if not _rows_aligned(g_zx):
            g_zx = g_zx.contiguous()
        g_s = torch.empty_like(s)
        g_e = torch.empty((E, H), dtype=torch.float32, device=x.device)
        lib = _capi.load()
        g_x = torch.empty((N, (F_ + 3) // 4 * 4), dtype=torch.float32, device=x.device)[:, :F_]
        w_lr_c = w_lr.contiguous()
        with torch.cuda.device(x.device):
            st = _stream(x)
            with _timed("gat_agg_bwd_dst", (N, E, H, F_)):
                _capi.check(lib.spgnn_gat_agg_bwd_dst(csc.indptr.data_ptr(), csc.indices.data_ptr(), x.data_ptr(), x.stride(0),
                                                      s.data_ptr(), s[:, H:].data_ptr(), s.stride(0), attn.data_ptr(),
                                                      g_zx.data_ptr(), g_zx.stride(0), F_, g_e.data_ptr(), g_s[:, H:].data_ptr(),
                                                      g_s.stride(0), N, E, H, F_, slope, p_drop, seed,
                                                      _seed_off_ptr(x.device), st), "spgnn_gat_agg_bwd_dst")
            with _timed("gat_agg_bwd_src", (N, E, H, F_)):
                _capi.check(lib.spgnn_gat_agg_bwd_src(csc.out_indptr.data_ptr(), csc.out_indices.data_ptr(),
                                                      csc.out_pos.data_ptr(), attn.data_ptr(), g_e.data_ptr(), g_zx.data_ptr(),
                                                      g_zx.stride(0), F_, -1, g_s[:, H:].data_ptr(),
                                                      w_lr_c.data_ptr(), w_lr_c.stride(0), g_x.data_ptr(), g_x.stride(0),
                                                      g_s.data_ptr(), g_s.stride(0), N, E, H, F_, p_drop, seed,
                                                      _seed_off_ptr(x.device), st), "spgnn_gat_agg_bwd_src")
        g_wlr = scores_bwd_w(g_s, x) if ctx.needs_input_grad[1] else None
        if ctx.needs_input_grad[0]:
            g_x = g_x + g_zx[:, H * F_:]                 # the copy of x inside Zx
        return (g_x if ctx.needs_input_grad[0] else None), g_wlr, None, None, None, None, None


_FOLD_WS: dict = {}
FUSE_LINEAR_MEAN_FOLD = True     # the weight-space half of gat_layer_linear_mean + classifier as two launches (spgnn_linear_mean_fold_*)


def linear_mean_fold_buffers(w_fc, w_res, bias, w_cls, b_cls, H: int, D: int, bf16: bool = False, x_block: bool = True):
    """One spgnn_linear_mean_fold_fwd call -> (w_comb (D, Kp) fp32, its scale block, w_comb as bf16 rows or None, b_mean or
    None, P (J, Kp), c0 (J,)); Kc = (H + 1) F real columns, every image zero padded to Kp = Kc rounded up to 16."""
    F_, J = w_fc.shape[1], w_cls.shape[0]
    Kp = _pad16((H + int(x_block)) * F_)
    dev = w_fc.device
    w_comb = torch.empty((D, Kp), dtype=torch.float32, device=dev)
    w_bf = torch.empty((D, Kp), dtype=torch.bfloat16, device=dev) if bf16 else None
    b_mean = torch.empty((D,), dtype=torch.float32, device=dev) if bias is not None else None
    P = torch.empty((J, Kp), dtype=torch.float32, device=dev)
    c0 = torch.empty((J,), dtype=torch.float32, device=dev)
    blk = None if bf16 else new_scale_block(dev)
    ws = _FOLD_WS.get((str(dev), H, F_))
    if ws is None:                                          # persistent scratch: partial sums and the (self re-arming) tickets
        n = int(_capi.load().spgnn_linear_mean_fold_workspace(H, F_))
        ws = _FOLD_WS[(str(dev), H, F_)] = (torch.empty((n,), dtype=torch.float32, device=dev),
                                            torch.zeros((Kp // 32 + 2,), dtype=torch.int32, device=dev))
    wf = w_fc if w_fc.stride(1) == 1 else w_fc.contiguous()
    wr = None if w_res is None else (w_res if w_res.stride(1) == 1 else w_res.contiguous())
    wc = w_cls if w_cls.stride(1) == 1 else w_cls.contiguous()
    with torch.cuda.device(dev):
        _capi.check(_capi.load().spgnn_linear_mean_fold_fwd(wf.data_ptr(), wf.stride(0), _ptr(wr), wr.stride(0) if wr is not None else 0,
                                                            _ptr(bias), wc.data_ptr(), wc.stride(0), _ptr(b_cls), H, D, F_, J,
                                                            w_comb.data_ptr(), Kp, _ptr(w_bf), Kp, _ptr(b_mean), P.data_ptr(), Kp,
                                                            c0.data_ptr(), _ptr(blk), ws[0].data_ptr(), ws[1].data_ptr(), int(x_block),
                                                            _stream(w_comb)), "spgnn_linear_mean_fold_fwd")
    return w_comb, blk, w_bf, b_mean, P, c0


def linear_mean_fold_grads(M1, cs, w_cls, w_comb, b_mean, H: int, D: int, F_: int, has_res: bool, has_bias: bool, x_block: bool = True):
    """One spgnn_linear_mean_fold_bwd call -> (g_w_fc, g_w_res or None, g_bias or None, g_w_cls)."""
    dev = M1.device
    J = w_cls.shape[0]
    g_fc = torch.empty((H * D, F_), dtype=torch.float32, device=dev)
    g_res = torch.empty((H * D, F_), dtype=torch.float32, device=dev) if has_res else None
    g_bias = torch.empty((H * D,), dtype=torch.float32, device=dev) if has_bias else None
    g_wc = torch.empty((J, D), dtype=torch.float32, device=dev)
    wc = w_cls if w_cls.stride(1) == 1 else w_cls.contiguous()
    m1 = M1 if M1.stride(1) == 1 else M1.contiguous()
    cs = cs.contiguous()
    with torch.cuda.device(dev):
        _capi.check(_capi.load().spgnn_linear_mean_fold_bwd(m1.data_ptr(), m1.stride(0), cs.data_ptr(), wc.data_ptr(), wc.stride(0),
                                                            w_comb.data_ptr(), w_comb.stride(0), _ptr(b_mean), H, D, F_, J,
                                                            g_fc.data_ptr(), F_, _ptr(g_res), F_, _ptr(g_bias), g_wc.data_ptr(), D,
                                                            int(x_block), _stream(m1)), "spgnn_linear_mean_fold_bwd")
    return g_fc, g_res, g_bias, g_wc


def _linear_mean_general_grads(g, zx, sg, sx, w_comb, Kc, H, D, F_, has_res, has_bias):
    """The ordinary route (the embedding carries a gradient too): g (N, D) -> (g_zx, g_w_fc, g_w_res, g_bias) through W_comb."""
    N = zx.shape[0]
    g_zx = torch.empty((N, (Kc + 3) // 4 * 4), dtype=torch.float32, device=zx.device)[:, :Kc]
    gemm_nt(g, w_comb[:, :Kc].t().contiguous(), sg, pow2_scale(w_comb), out=g_zx)
    if has_bias:
        g_wc_, g_bm = gemm_tn(g, zx, sg, sx, want_colsum=True)
    else:
        g_wc_, g_bm = gemm_tn(g, zx, sg, sx), None
    g_wc_ = g_wc_ * (1.0 / H)
    g_fc = g_wc_[:, :H * F_].reshape(D, H, F_).permute(1, 0, 2).reshape(H * D, F_)
    g_res = g_wc_[:, H * F_:].unsqueeze(0).expand(H, D, F_).reshape(H * D, F_) if has_res else None
    g_bias = (g_bm * (1.0 / H)).repeat(H) if has_bias else None
    return g_zx, g_fc, g_res, g_bias


class _LinearMeanClassifierFn(torch.autograd.Function):
    """(Zx, W_fc, W_res, bias, Wc, bc) -> (mean_h out_h = Zx W_comb^T + b_mean, logits = Zx P^T + c0): _LinearClassifierFn with
    the assembly of W_comb / b_mean / P / c0 from the layer's parameters and the way back to their gradients inside the
    node - one launch each (spgnn_linear_mean_fold_fwd / _bwd) instead of ~28 tiny torch / rocBLAS launches per step."""

    @staticmethod
    def forward(ctx, zx, w_fc, w_res, bias, w_cls, b_cls, H, D, x_block=True):
        """``x_block`` False (with H = 1, w_res None): a plain Linear + folded classifier, W_comb = W_fc.  ``w_fc`` may be the
        transposed view of a prepared (F, D) parameter (GraphConv): its prepared transpose image is read then."""
        ctx.set_materialize_grads(False)
        zx = _rowmajor(zx)
        if not _rows_aligned(zx):
            zx = cat_padded((zx,))
        Kc = zx.shape[1]
        if w_fc.stride(1) != 1:
            prep = _prepared_linear_operands(w_fc, w_fc.shape[0], w_fc.shape[1])
            w_fc = prep[0] if prep is not None else w_fc.contiguous()
        assert Kc == (H + int(x_block)) * w_fc.shape[1]
        ctx.x_block = bool(x_block)
        w_comb, blk, _, b_mean, P, c0 = linear_mean_fold_buffers(w_fc, w_res, bias, w_cls, b_cls, H, D, x_block=x_block)
        sx = operand_scale(zx)
        y = gemm_nt(zx, w_comb[:, :Kc], sx, blk, bias=b_mean)
        head, ctx.head = LOSS_HEAD, None
        if head is not None and not head.used and zx.shape[0] == head.labels.shape[0] and classifier_ce_supported(zx, P[:, :Kc]):
            # the step's loss joins this node (spgnn_classifier_ce on the folded classifier): logits, loss sums, logit gradient
            # and the partials of g_logits^T Zx from one pass over Zx
            logits, head.g_logits, wpart, colsum = classifier_ce(zx, P[:, :Kc], c0, head)
            head.used = True
            ctx.head = (wpart, colsum)
        else:
            logits = scores_fwd(zx, P[:, :Kc], bias=c0)
        ctx.cfg = (H, D, w_fc.shape[1], Kc, w_res is not None, bias is not None, b_cls is not None)
        ctx.save_for_backward(zx, w_comb, sx, blk, P, w_cls, b_mean)
        return y, logits

    @staticmethod
    def backward(ctx, g_y, g_logits):
        if g_y is None and g_logits is None:
            return (None,) * 9
        zx, w_comb, sx, blk, P, w_cls, b_mean = ctx.saved_tensors
        H, D, F_, Kc, has_res, has_bias, has_bcls = ctx.cfg
        N = zx.shape[0]
        g_zx = g_fc = g_res = g_bias = g_wcls = g_bcls = None
        cs = M1 = None
        if g_logits is not None:
            head, ctx.head = ctx.head, None
            g_logits = _rowmajor(g_logits)
            if head is not None:                          # the loss kernel left the column sums and the partials of g_logits^T Zx
                cs = head[1]
                M1 = sum_partials(head[0])[:, :Kc]
            else:
                cs = column_sums(g_logits)
                M1 = scores_bwd_w(g_logits, zx)                          # g_logits^T Zx (J, Kc)
            g_bcls = cs if has_bcls else None
        if g_y is None:                                   # the folded route (the training step): no (N, D) gradient exists
            if ctx.needs_input_grad[0]:
                g_zx = torch.empty((N, (Kc + 3) // 4 * 4), dtype=torch.float32, device=zx.device)[:, :Kc]
                scores_bwd_x_(g_zx, g_logits, P[:, :Kc], accumulate=False)
            g_fc, g_res, g_bias, g_wcls = linear_mean_fold_grads(M1, cs, w_cls, w_comb, b_mean, H, D, F_, has_res, has_bias,
                                                                 x_block=ctx.x_block)
            return g_zx, g_fc, g_res, g_bias, g_wcls, g_bcls, None, None, None
        g = _rowmajor(g_y)
        if g_logits is not None:
            g = g.clone() if g.data_ptr() == g_y.data_ptr() else g
            if not _rows_aligned(g):
                g = cat_padded((g,))
            scores_bwd_x_(g, g_logits, w_cls.detach(), accumulate=True)
            g_wcls = torch.mm(M1, w_comb[:, :Kc].t())
            if has_bias:
                g_wcls.addr_(cs, b_mean)
        elif not _rows_aligned(g):
            g = cat_padded((g,))
        g_zx, g_fc, g_res, g_bias = _linear_mean_general_grads(g, zx, pow2_scale(g), sx, w_comb, Kc, H, D, F_, has_res, has_bias)
        return g_zx, g_fc, g_res, g_bias, g_wcls, g_bcls, None, None, None


def gat_layer_linear_mean(csc: DeviceCSC, x, w_fc, w_res, w_lr, bias, H: int, D: int, slope: float, p_drop: float = 0.0,
                          seed: int = 0, w_cls=None, b_cls=None):
    """mean over heads of a GATConv WITHOUT activation (see _GATAggregateFn) -> (mean (N, D), attn (E, H)).  The weight
    assembly below is a few tiny torch ops, so autograd hands the gradients back to ``w_fc`` / ``w_res`` / ``bias``.
    With a classifier (``w_cls`` (J, D), ``b_cls``): -> (mean, attn, logits), the classifier joined to the product's node
    (_LinearClassifierFn) when its shapes allow, a plain ``linear`` after it otherwise."""
    _require_cuda(x, w_fc, w_res, w_lr, bias, w_cls, b_cls)
    F_ = x.shape[1]
    zx, attn = _GATAggregateFn.apply(x, w_lr, csc, H, slope, p_drop, seed)
    blk = getattr(zx.grad_fn, "scale_block", None) if zx.grad_fn is not None else None
    zx = take_loss_rows(zx, w_cls is not None)         # a loss-rows step: the product, the mean and the classifier on the kept rows
    if blk is not None:
        zx._spgnn_scale = (zx._version, blk)           # (the maximum over all rows bounds the listed ones)
    if (FUSE_LINEAR_MEAN_FOLD and w_cls is not None and w_cls.shape[0] <= 32 and w_cls.shape[1] == D and D % 4 == 0
            and zx.shape[0] >= MIN_GEMM_ROWS and GEMM_MODE == "f16x3" and zx.dtype == torch.float32):
        out, logits = _LinearMeanClassifierFn.apply(zx, w_fc, w_res, bias, w_cls, b_cls, H, D)
        return out, attn, logits
    parts = [w_fc.view(H, D, F_).permute(1, 0, 2).reshape(D, H * F_)]
    parts.append(w_res.view(H, D, F_).sum(0) if w_res is not None else w_fc.new_zeros((D, F_)))
    w_comb = torch.cat(parts, dim=1) * (1.0 / H)
    b_mean = bias.view(H, D).mean(0) if bias is not None else None
    if w_cls is None:
        return linear(zx, w_comb, b_mean), attn
    if linear_classifier_supported(zx, w_comb, w_cls):
        out, logits = _LinearClassifierFn.apply(zx, w_comb, b_mean, w_cls, b_cls)
    else:
        out = linear(zx, w_comb, b_mean)
        logits = torch.nn.functional.linear(out, w_cls, b_cls)
    return out, attn, logits


def gat_layer_agg_first(csc: DeviceCSC, x, w_fc, w_res, w_lr, bias, H: int, D: int, slope: float, act: int,
                        p_drop: float = 0.0, seed: int = 0, mean: bool = False, w_cls=None, b_cls=None):
    """Same contract as gat_layer; ``w_fc`` / ``w_res`` (H*D, F) separately (w_res may be None).  With ``mean`` and a
    classifier (``w_cls`` (J, D), ``b_cls``): returns (mean, attn, logits), the classifier fused into the node."""
    _require_cuda(x, w_fc, w_res, w_lr, bias, w_cls, b_cls)
    rst, attn, logits = _GATAggFirstFn.apply(x, w_fc, w_res, w_lr, bias, w_cls, b_cls, csc, H, D, slope, act, p_drop, seed, mean)
    if w_cls is not None and mean:
        return rst, attn, logits
    return rst, attn


# --------------------------------------------------------------------------------------------
# SpMM sum / max
# --------------------------------------------------------------------------------------------
def spmm_sum_raw(indptr, indices, x, w_src, w_dst, eps, N: int, E: int, bias=None, act: int = ACT_NONE, absmax=None,
                 drop=None) -> torch.Tensor:
    """``absmax``: a scale block the kernel folds max |out| into (the result as the next product's operand).  ``drop`` =
    (p, seed): the stored rows are dropout(act(...), p) under the hash mask."""
    _require_cuda(x, w_src, w_dst, eps, bias)
    F_ = x.shape[1]
    if bias is not None and bias.data_ptr() % 16:
        bias = bias.clone()                                   # the vector kernels read it with 16-byte loads
    out = torch.empty((N, F_), dtype=torch.float32, device=x.device)
    lib = _capi.load()
    with torch.cuda.device(x.device), _timed("spmm_sum", (N, E, F_)):
        if drop is not None and drop[0] > 0.0:
            _capi.check(lib.spgnn_spmm_sum_dropout(indptr.data_ptr(), indices.data_ptr(), x.data_ptr(), x.stride(0), _ptr(w_src),
                                                   _ptr(w_dst), _ptr(eps), _ptr(bias), act, out.data_ptr(), out.stride(0), N, E, F_,
                                                   _ptr(absmax), float(drop[0]), int(drop[1]), _seed_off_ptr(x.device), _stream(x)),
                        "spgnn_spmm_sum_dropout")
        else:
            _capi.check(lib.spgnn_spmm_sum(indptr.data_ptr(), indices.data_ptr(), x.data_ptr(), x.stride(0), _ptr(w_src),
                                           _ptr(w_dst), _ptr(eps), _ptr(bias), act, out.data_ptr(), out.stride(0), N, E, F_, _ptr(absmax),
                                           _stream(x)), "spgnn_spmm_sum")
    return out


class _SpmmSumFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, eps, csc: DeviceCSC, w_src, w_dst, bias=None, act: int = ACT_NONE, drop=None):
        x = _rowmajor(x)
        fuse = (bias is not None or act != ACT_NONE) and x.shape[1] % 4 == 0
        blk = new_scale_block(x.device) if (EMIT_SCALES and x.shape[1] % 4 == 0) else None
        dropping = drop is not None and drop[0] > 0.0
        if dropping and not (x.shape[1] % 4 == 0 and act in (ACT_NONE, ACT_RELU, ACT_LRELU)):
            raise RuntimeError("spmm_sum(drop=...): width % 4 == 0 and ReLU / LeakyReLU / no activation only")
        out = spmm_sum_raw(csc.indptr, csc.indices, x, w_src, w_dst, eps, csc.num_nodes, csc.num_edges, bias if fuse else None,
                           act if fuse else ACT_NONE, absmax=blk, drop=drop if dropping else None)
        ctx.scale_block = blk
        if not fuse and (bias is not None or act != ACT_NONE):
            raise RuntimeError("spmm_sum: the bias / activation epilogue needs a width that is a multiple of 4")
        ctx.csc, ctx.w, ctx.act, ctx.has_bias = csc, (w_src, w_dst), act, bias is not None
        ctx.drop = (float(drop[0]), int(drop[1])) if dropping else None
        ctx.save_for_backward(x if eps is not None else None, eps, out if act != ACT_NONE else None)
        return out

    @staticmethod
    def backward(ctx, g_out):
        x, eps, out = ctx.saved_tensors
        csc = ctx.csc
        w_src, w_dst = ctx.w
        g_out = _rowmajor(g_out)
        g_bias = None
        want_bias = ctx.has_bias and ctx.needs_input_grad[5]
        want_eps = eps is not None and ctx.needs_input_grad[1]
        g_eps = None
        drop_p, drop_seed = ctx.drop if ctx.drop is not None else (0.0, 0)
        if ctx.drop is not None or ctx.act != ACT_NONE:
            # dropout's and the activation's backward in one pass (mask regenerated; with dropout the derivative is taken from the
            # dropped rows: sign only), and - when the bias gradient is wanted - its column sums from the same pass
            g = g_out if _rows_aligned(g_out) else g_out.contiguous()
            N, C = g.shape
            lib = _capi.load()
            nb = lib.spgnn_act_bwd_colsum_blocks(N, C) if (want_bias and BIAS_COLSUM) else 0
            if nb > 0:
                # eps' gradient sum_v <g_pre[v], x[v]> rides along (float C of every partial row)
                with_dot = want_eps and x.stride(1) == 1 and _rows_aligned(x)
                g_pre = torch.empty((N, C), dtype=torch.float32, device=g.device)
                part = torch.empty((nb, C + 4 if with_dot else C), dtype=torch.float32, device=g.device)
                with torch.cuda.device(g.device), _timed("act_bwd", (N, 1, C, ctx.act, 0)):
                    _capi.check(lib.spgnn_act_bwd_colsum(g.data_ptr(), g.stride(0), _ptr(out), out.stride(0) if out is not None else 0,
                                                         g_pre.data_ptr(), g_pre.stride(0), 0, part.data_ptr(), N, C, ctx.act,
                                                         drop_p, drop_seed, _seed_off_ptr(g.device), x.data_ptr() if with_dot else 0,
                                                         x.stride(0) if with_dot else 0, _stream(g)),
                                "spgnn_act_bwd_colsum")
                sums = sum_partials(part)
                g_out, g_bias = g_pre, sums[:C]
                if with_dot:
                    g_eps = sums[C:C + 1].reshape(eps.shape)
            elif ctx.drop is not None:
                g_pre = torch.empty((N, C), dtype=torch.float32, device=g.device)
                with torch.cuda.device(g.device), _timed("act_bwd", (N, 1, C, ctx.act, 0)):
                    _capi.check(lib.spgnn_act_bwd_dropout(g.data_ptr(), g.stride(0), _ptr(out), out.stride(0) if out is not None else 0,
                                                          g_pre.data_ptr(), g_pre.stride(0), 0, N, C, ctx.act,
                                                          drop_p, drop_seed, _seed_off_ptr(g.device), _stream(g)),
                                "spgnn_act_bwd_dropout")
                g_out = g_pre
            else:
                g_out, _ = act_bwd(g, out, 1, out.shape[1], ctx.act, False)
        if want_bias and g_bias is None:
            g_bias = g_out.sum(0)
        g_x = None
        blk = None
        if ctx.needs_input_grad[0]:   # transpose: swap the roles of the two scalings
            blk = new_scale_block(g_out.device) if (EMIT_SCALES and g_out.shape[1] % 4 == 0) else None
            g_x = spmm_sum_raw(csc.out_indptr, csc.out_indices, g_out, w_dst, w_src, eps, csc.num_nodes, csc.num_edges, absmax=blk)
            if blk is not None:
                g_x._spgnn_scale = (g_x._version, blk)   # for the node behind (a product's backward): no absmax pass over g_x
        if want_eps and g_eps is None:
            if g_out.is_contiguous() and x.is_contiguous():      # one pass over both tensors, no (N, F) temporary
                g_eps = torch.dot(g_out.reshape(-1), x.reshape(-1)).reshape(eps.shape)
            else:
                g_eps = (g_out * x).sum().reshape(eps.shape)
        return g_x, g_eps, None, None, None, g_bias, None, None


def spmm_sum(csc: DeviceCSC, x, w_src=None, w_dst=None, eps=None, bias=None, act: int = ACT_NONE, drop=None) -> torch.Tensor:
    """out[v] = act((1+eps)*x[v] (if eps given) + w_dst[v] * sum_{u in in(v)} w_src[u] * x[u] + bias); ``drop`` = (p, seed):
    followed by dropout under the hash mask, in the same kernel."""
    _require_cuda(x)
    out = _SpmmSumFn.apply(x, eps, csc, w_src, w_dst, bias, act, drop)
    blk = getattr(out.grad_fn, "scale_block", None) if out.grad_fn is not None else None
    if blk is not None:
        out._spgnn_scale = (out._version, blk)          # its GEMM operand scale, emitted by the kernel itself
    return out


COMPACT_MAX_ARG = True   # spmm_max keeps its argmax as one byte per element (position inside the in-edge list) when it can


class _SpmmMaxFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, csc: DeviceCSC):
        x = _rowmajor(x)
        N, E, F_ = csc.num_nodes, csc.num_edges, x.shape[1]
        out = torch.empty((N, F_), dtype=torch.float32, device=x.device)
        lib = _capi.load()
        ctx.u8 = bool(COMPACT_MAX_ARG and csc.max_in_degree <= 254 and _rows_aligned(x) and lib.spgnn_spmm_max_u8_supported(F_))
        arg = torch.empty((N, F_), dtype=torch.uint8 if ctx.u8 else torch.int32, device=x.device)
        with torch.cuda.device(x.device), _timed("spmm_max_fwd", (N, E, F_)):
            if ctx.u8:
                _capi.check(lib.spgnn_spmm_max_fwd_u8(csc.indptr.data_ptr(), csc.indices.data_ptr(), x.data_ptr(), x.stride(0),
                                                      out.data_ptr(), out.stride(0), arg.data_ptr(), arg.stride(0), N, E, F_,
                                                      _stream(x)), "spgnn_spmm_max_fwd_u8")
            else:
                _capi.check(lib.spgnn_spmm_max_fwd(csc.indptr.data_ptr(), csc.indices.data_ptr(), x.data_ptr(), x.stride(0),
                                                   out.data_ptr(), out.stride(0), arg.data_ptr(), arg.stride(0), N, E, F_,
                                                   _stream(x)), "spgnn_spmm_max_fwd")
        ctx.csc = csc
        ctx.save_for_backward(arg)
        return out

    @staticmethod
    def backward(ctx, g_out):
        (arg,) = ctx.saved_tensors
        csc = ctx.csc
        g_out = _rowmajor(g_out)
        N, E, F_ = csc.num_nodes, csc.num_edges, g_out.shape[1]
        lib = _capi.load()
        if ctx.u8 and not _rows_aligned(g_out):
            g_out = g_out.contiguous()
        g_x = torch.empty((N, F_), dtype=torch.float32, device=g_out.device)
        with torch.cuda.device(g_out.device), _timed("spmm_max_bwd", (N, E, F_)):
            if ctx.u8:
                _capi.check(lib.spgnn_spmm_max_bwd_u8(csc.indptr.data_ptr(), csc.out_indptr.data_ptr(), csc.out_indices.data_ptr(),
                                                      csc.out_pos.data_ptr(), g_out.data_ptr(), g_out.stride(0),
                                                      arg.data_ptr(), arg.stride(0), g_x.data_ptr(), g_x.stride(0), N, E, F_,
                                                      _stream(g_out)), "spgnn_spmm_max_bwd_u8")
            else:
                _capi.check(lib.spgnn_spmm_max_bwd(csc.out_indptr.data_ptr(), csc.out_indices.data_ptr(),
                                                   csc.out_pos.data_ptr(), g_out.data_ptr(), g_out.stride(0),
                                                   arg.data_ptr(), arg.stride(0), g_x.data_ptr(), g_x.stride(0), N, E, F_,
                                                   _stream(g_out)), "spgnn_spmm_max_bwd")
        return g_x, None


class _PlainCtx:
    """Stand-in for an autograd ctx when one Function runs another's forward / backward as a part of its own."""

    def __init__(self):
        self.saved_tensors, self.needs_input_grad = (), ()

    def set_materialize_grads(self, _value):
        pass

    def save_for_backward(self, *tensors):
        self.saved_tensors = tensors


class _PoolMaxFn(torch.autograd.Function):
    """SAGEConv 'pool' up to the aggregation as ONE node: neigh[v] = max_{u in in(v)} relu(h[u] W^T + b) (reference
    models.py:668-679 via dgl SAGEConv).  Forward = ops.linear (ReLU epilogue) + the max aggregation with the one-byte argmax;
    backward: the routing kernel masks its result by relu' itself (spgnn_spmm_max_bwd_u8_relu) - the gradient of fc_pool's
    pre-activation in one pass instead of routing + an activation-backward pass over (N, in_feats) - and the product's own
    backward (_LinearFn) continues from there."""

    @staticmethod
    def forward(ctx, h, weight, bias, csc: DeviceCSC):
        inner = _PlainCtx()
        pool = _LinearFn.forward(inner, h, weight, bias, ACT_RELU)
        N, E, F_ = csc.num_nodes, csc.num_edges, pool.shape[1]
        out = torch.empty((N, F_), dtype=torch.float32, device=pool.device)
        arg = torch.empty((N, F_), dtype=torch.uint8, device=pool.device)
        with torch.cuda.device(pool.device), _timed("spmm_max_fwd", (N, E, F_)):
            _capi.check(_capi.load().spgnn_spmm_max_fwd_u8(csc.indptr.data_ptr(), csc.indices.data_ptr(), pool.data_ptr(), pool.stride(0),
                                                           out.data_ptr(), out.stride(0), arg.data_ptr(), arg.stride(0), N, E, F_,
                                                           _stream(pool)), "spgnn_spmm_max_fwd_u8")
        ctx.inner, ctx.csc = inner, csc
        ctx.save_for_backward(pool, arg)
        ctx.scale_block = inner.scale_block       # every element of a neighbourhood maximum is an element of pool (or 0)
        return out

    @staticmethod
    def backward(ctx, g_out):
        csc, inner = ctx.csc, ctx.inner
        pool, arg = ctx.saved_tensors
        g_out = _rowmajor(g_out)
        if not _rows_aligned(g_out):
            g_out = g_out.contiguous()
        N, E, F_ = csc.num_nodes, csc.num_edges, g_out.shape[1]
        g_pre = torch.empty((N, F_), dtype=torch.float32, device=g_out.device)
        sg = new_scale_block(g_out.device)
        with torch.cuda.device(g_out.device), _timed("spmm_max_bwd", (N, E, F_)):
            _capi.check(_capi.load().spgnn_spmm_max_bwd_u8_relu(csc.indptr.data_ptr(), csc.out_indptr.data_ptr(), csc.out_indices.data_ptr(),
                                                                csc.out_pos.data_ptr(), g_out.data_ptr(), g_out.stride(0), arg.data_ptr(),
                                                                arg.stride(0), g_pre.data_ptr(), g_pre.stride(0), pool.data_ptr(),
                                                                pool.stride(0), sg.data_ptr(), N, E, F_, _stream(g_out)),
                        "spgnn_spmm_max_bwd_u8_relu")
        inner.needs_input_grad = tuple(ctx.needs_input_grad[:3]) + (False,) * 5
        inner.pre_activated, inner.pre_scale = True, sg
        g_x, g_w, g_b = _LinearFn.backward(inner, g_pre)[:3]
        return g_x, g_w, g_b, None


POOL_MAX_FUSED = True    # SAGEConv 'pool': fc_pool + ReLU + max aggregation as one autograd node (relu' applied by the routing kernel)


def pool_max_supported(csc: DeviceCSC, h: torch.Tensor, weight: torch.Tensor) -> bool:
    N = h.shape[0] if h.dim() == 2 else 0
    return bool(POOL_MAX_FUSED and COMPACT_MAX_ARG and h.is_cuda and GEMM_MODE == "f16x3" and N >= MIN_GEMM_ROWS and N == csc.num_nodes
                and weight.shape[0] >= 32 and weight.shape[1] >= 32 and h.dtype == torch.float32 and weight.dtype == torch.float32
                and csc.max_in_degree <= 254 and getattr(csc, "num_dst", None) is None
                and _capi.load().spgnn_spmm_max_u8_supported(int(weight.shape[0])))


def pool_max(csc: DeviceCSC, h: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor]) -> torch.Tensor:
    """max over in-neighbours of relu(F.linear(h, weight, bias)); needs :func:`pool_max_supported`."""
    out = _PoolMaxFn.apply(h, weight, bias, csc)
    blk = getattr(out.grad_fn, "scale_block", None) if out.grad_fn is not None else None
    if blk is not None:
        out._spgnn_scale = (out._version, blk)
    return out


def spmm_max(csc: DeviceCSC, x) -> torch.Tensor:
    _require_cuda(x)
    return _SpmmMaxFn.apply(x, csc)


# --------------------------------------------------------------------------------------------
# optimizer step over a flat bucket
# --------------------------------------------------------------------------------------------
def sgd_momentum_step_(param: torch.Tensor, grad: torch.Tensor, buf: torch.Tensor, lr: float, momentum: float,
                       weight_decay: float = 0.0, first_step: bool = False,
                       grad_scale: Optional[torch.Tensor] = None, lr_dev: Optional[torch.Tensor] = None,
                       weight_sum: Optional[torch.Tensor] = None, loss_num: Optional[torch.Tensor] = None,
                       loss_out: Optional[torch.Tensor] = None, skipped: Optional[torch.Tensor] = None) -> None:
    """``lr_dev`` (device scalar) overrides ``lr`` at run time (learning-rate schedules under graph replay).  ``weight_sum``
    (device scalar) instead of ``grad_scale``: the gradient is divided by it in the kernel, and with ``loss_num`` /
    ``loss_out`` the same launch writes loss_out[0] = loss_num[0] / weight_sum[0] (spgnn_sgd_momentum_step_mean)."""
    _require_cuda(param, grad, buf, grad_scale, lr_dev, weight_sum, loss_num, loss_out)
    assert param.is_contiguous() and grad.is_contiguous() and buf.is_contiguous()
    assert param.dtype == grad.dtype == buf.dtype == torch.float32 and param.numel() == grad.numel() == buf.numel()
    lib = _capi.load()
    if weight_sum is not None:
        assert grad_scale is None and (loss_out is None or loss_num is not None)
        if skipped is not None:         # (1,) int32 counter: a step with a non-finite loss is not applied (spgnn_sgd_momentum_step_guarded)
            with torch.cuda.device(param.device):
                _capi.check(lib.spgnn_sgd_momentum_step_guarded(param.data_ptr(), grad.data_ptr(), buf.data_ptr(), weight_sum.data_ptr(),
                                                                loss_num.data_ptr(), _ptr(loss_out), _ptr(lr_dev), skipped.data_ptr(),
                                                                param.numel(), lr, momentum, weight_decay, int(first_step),
                                                                _stream(param)), "spgnn_sgd_momentum_step_guarded")
            return
        with torch.cuda.device(param.device):
            _capi.check(lib.spgnn_sgd_momentum_step_mean(param.data_ptr(), grad.data_ptr(), buf.data_ptr(), weight_sum.data_ptr(),
                                                         _ptr(loss_num), _ptr(loss_out), _ptr(lr_dev), param.numel(), lr, momentum,
                                                         weight_decay, int(first_step), _stream(param)), "spgnn_sgd_momentum_step_mean")
        return
    with torch.cuda.device(param.device):
        _capi.check(lib.spgnn_sgd_momentum_step(param.data_ptr(), grad.data_ptr(), buf.data_ptr(), _ptr(grad_scale),
                                                _ptr(lr_dev), param.numel(), lr, momentum, weight_decay,
                                                int(first_step), _stream(param)), "spgnn_sgd_momentum_step")


# --------------------------------------------------------------------------------------------
# fp32-accurate GEMM on the fp16 matrix cores (spgnn_gemm.hip)
# --------------------------------------------------------------------------------------------
def pow2_scale(x: torch.Tensor) -> torch.Tensor:
    """Device scalar 2^(14 - e), max|x| <= 2^e: centres the tensor in the fp16 range for the split GEMM."""
    _require_cuda(x)
    assert x.dim() == 2 and x.dtype == torch.float32
    if not _rows_aligned(x):
        x = x.contiguous() if x.shape[1] % 4 == 0 else torch.nn.functional.pad(x, (0, 4 - x.shape[1] % 4))
    buf = torch.empty(1 + 2048, dtype=torch.float32, device=x.device)        # [scale | per-block partial maxima]
    with torch.cuda.device(x.device), _timed("absmax", (x.shape[0], x.shape[1])):
        _capi.check(_capi.load().spgnn_pow2_scale(x.data_ptr(), x.stride(0), x.shape[0], x.shape[1], buf.data_ptr(),
                                                  buf[4:].data_ptr(), 2044, _stream(x)), "spgnn_pow2_scale")
    return buf[:1]


def presplit(w: torch.Tensor, scale: Optional[torch.Tensor] = None, partials: Optional[torch.Tensor] = None,
             w2: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None, wide: Optional[bool] = None):
    """Pre-split form of a GEMM ``b`` operand (spgnn_presplit): ``w`` (R, K) fp32 with 16-byte rows -> a tensor of the same
    shape and strides holding, per group of four columns, the packed fp16 hi / lo pairs of scale * w.  The scale is given
    (``scale``) or derived from ``partials`` (block maxima) - then it is returned too.  ``w2``: a second matrix under the same
    scale (the transpose), split by the same launch.  -> (w_ps, w2_ps or None, scale)."""
    _require_cuda(w, w2)
    assert (scale is None) != (partials is None) and _rows_aligned(w) and (w2 is None or _rows_aligned(w2))
    def like(t):
        buf = torch.empty((t.shape[0], t.stride(0)), dtype=torch.float32, device=t.device)
        return buf[:, :t.shape[1]]
    if out is not None:                      # refill an existing pre-split image (a batch arena keeps its address)
        assert out.shape == w.shape and out.stride() == w.stride() and out.dtype == torch.float32
    w_ps, w2_ps = (out if out is not None else like(w)), (like(w2) if w2 is not None else None)
    sc_out = torch.empty(1, dtype=torch.float32, device=w.device) if scale is None else None
    with torch.cuda.device(w.device):
        _capi.check(_capi.load().spgnn_presplit(_ptr(partials), partials.numel() if partials is not None else 0, _ptr(scale), _ptr(sc_out),
                                                w.data_ptr(), w.stride(0), w.shape[0], w.shape[1], w_ps.data_ptr(),
                                                _ptr(w2), w2.stride(0) if w2 is not None else 0, w2.shape[0] if w2 is not None else 0,
                                                w2.shape[1] if w2 is not None else 0, _ptr(w2_ps), int(GEMM_WIDE if wide is None else wide),
                                                _stream(w)), "spgnn_presplit")
    return w_ps, w2_ps, (scale if scale is not None else sc_out)


def gemm_nt(a: torch.Tensor, b: torch.Tensor, scale_a: Optional[torch.Tensor] = None,
            scale_b: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
            upd_u: Optional[torch.Tensor] = None, upd_v: Optional[torch.Tensor] = None,
            bias: Optional[torch.Tensor] = None, act: int = 0, score_l: Optional[torch.Tensor] = None,
            score_r: Optional[torch.Tensor] = None, score_out: Optional[torch.Tensor] = None, tile: int = 0,
            b_presplit: bool = False, a_presplit: bool = False, wide: Optional[bool] = None) -> torch.Tensor:
    """a (M,K) @ b (N,K)^T [+ upd_u (M,J) @ upd_v (J,N), exact fp32, fused into the epilogue] -> (M,N); fp32
    in/out, fp16x3 split on the matrix cores.  ``bias`` (N,) / ``act``: epilogue act(C + bias).  ``score_out``
    (M, C/64, 2) with ``score_l`` / ``score_r`` (C,): per 64-column block dot products of the first C output columns.
    ``b_presplit``: ``b`` is the pre-split form of the operand (:func:`presplit`, made with ``scale_b``); ``a_presplit``
    (only together with it): ``a`` likewise, made with ``scale_a`` (constant node data, split once per loader batch)."""
    if (not b_presplit and not a_presplit and upd_u is None and not tile and skinny_rows(a.shape[0]) and _rows_aligned(a) and _rows_aligned(b)
            and (score_out is None or (bias is None and act == 0 and score_l.numel() % 64 == 0))):
        return gemm_nt_skinny(a, b, out=out, bias=bias, act=act, score_l=score_l, score_r=score_r, score_out=score_out)
    b_presplit = int(bool(b_presplit)) | (2 if a_presplit else 0) | (4 if (GEMM_WIDE if wide is None else wide) else 0)
    _require_cuda(a, b)
    M, K = a.shape
    N = b.shape[0]
    assert b.shape[1] == K and _rows_aligned(a) and _rows_aligned(b)
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    assert out.shape == (M, N) and out.stride(1) == 1
    assert bias is None or (bias.numel() == N and bias.is_contiguous())
    J = 0
    if upd_u is not None:
        J = upd_u.shape[1]
        assert upd_u.shape[0] == M and upd_u.stride(1) == 1 and upd_v.shape[0] == J and upd_v.shape[1] >= N
        assert _rows_aligned(upd_v) and upd_v.stride(0) >= (N + 3) // 4 * 4 and J <= 32
    args = (a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), out.data_ptr(), out.stride(0), M, N, K, _ptr(scale_a),
            _ptr(scale_b), _ptr(upd_u), upd_u.stride(0) if J else 0, _ptr(upd_v), upd_v.stride(0) if J else 0, J, _ptr(bias), act,
            _ptr(score_l), _ptr(score_r), _ptr(score_out), score_l.numel() if score_out is not None else 0)
    with torch.cuda.device(a.device), _timed("gemm_nt", (M, N, K)):
        if tile:                                       # block tile pinned by the caller (2 / 4 / 5): bit-identical results
            _capi.check(_capi.load().spgnn_gemm_nt_tile(*args, tile, int(b_presplit), _stream(a)), "spgnn_gemm_nt_tile")
        else:
            _capi.check(_capi.load().spgnn_gemm_nt(*args, int(b_presplit), _stream(a)), "spgnn_gemm_nt")
    return out


PAIR_GEMMS = True       # a level's structure + position products as ONE launch each (spgnn_gemm_nt_pair / _tn_pair); False: two


class NtProblem:
    """One gemm_nt call, described but not launched (see :func:`gemm_nt_pair`); ``out`` is allocated here."""

    def __init__(self, a, b, scale_a=None, scale_b=None, out=None, score_l=None, score_r=None, score_out=None, b_presplit=False,
                 a_presplit=False):
        _require_cuda(a, b)
        M, K = a.shape
        N = b.shape[0]
        assert b.shape[1] == K and _rows_aligned(a) and _rows_aligned(b)
        if out is None:
            out = torch.empty((M, N), dtype=torch.float32, device=a.device)
        assert out.shape == (M, N) and out.stride(1) == 1
        self.out, self.shape = out, (M, N, K)
        self.b_presplit = int(bool(b_presplit)) | (2 if a_presplit else 0) | (4 if GEMM_WIDE else 0)   # the C ABI's mask
        self.kw = dict(scale_a=scale_a, scale_b=scale_b, out=out, score_l=score_l, score_r=score_r, score_out=score_out,
                       b_presplit=b_presplit, a_presplit=a_presplit)
        self.a, self.b = a, b
        q = self.c = _capi.GemmNtProblem()
        q.A, q.lda, q.B, q.ldb, q.C, q.ldc, q.M, q.N, q.K = a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), out.data_ptr(), out.stride(0), M, N, K
        q.scale_a, q.scale_b = _ptr(scale_a), _ptr(scale_b)
        q.score_l, q.score_r, q.score_out = _ptr(score_l), _ptr(score_r), _ptr(score_out)
        q.score_cols = score_l.numel() if score_out is not None else 0

    def run(self) -> torch.Tensor:
        return gemm_nt(self.a, self.b, **self.kw)


def gemm_nt_pair(first: NtProblem, second: NtProblem):
    """Both products in one launch (spgnn_gemm_nt_pair; pass the larger one first: its block tile is used for both) -
    bit-identical to ``first.run(); second.run()``, which is what happens when PAIR_GEMMS is off or the operand forms differ."""
    if not PAIR_GEMMS or first.b_presplit != second.b_presplit or first.a.device != second.a.device or skinny_rows(first.shape[0]):
        return first.run(), second.run()                   # (a small inference batch: each product on the skinny kernel)
    import ctypes
    with torch.cuda.device(first.a.device), _timed("gemm_nt_pair", first.shape + second.shape):
        _capi.check(_capi.load().spgnn_gemm_nt_pair(ctypes.byref(first.c), ctypes.byref(second.c), int(first.b_presplit),
                                                    _stream(first.a)), "spgnn_gemm_nt_pair")
    return first.out, second.out


def gemm_nt_headmean(a: torch.Tensor, b: torch.Tensor, scale_a, scale_b, out: torch.Tensor, other: torch.Tensor,
                     mean_out: torch.Tensor, bias: Optional[torch.Tensor] = None, act: int = 0, b_presplit: bool = False) -> None:
    """out = act(a @ b^T + bias) and mean_out = 0.5 * (out + other): the second head's projection of a two-head layer also
    writes the mean over heads from its tiles (spgnn_gemm_nt_headmean)."""
    _require_cuda(a, b, out, other, mean_out)
    M, K = a.shape
    N = b.shape[0]
    assert _rows_aligned(a) and _rows_aligned(b) and out.shape == (M, N) and other.shape == (M, N) and mean_out.shape == (M, N)
    with torch.cuda.device(out.device), _timed("gemm_nt", (M, N, K)):
        _capi.check(_capi.load().spgnn_gemm_nt_headmean(a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), out.data_ptr(),
                                                        out.stride(0), M, N, K, _ptr(scale_a), _ptr(scale_b), _ptr(bias), act,
                                                        other.data_ptr(), other.stride(0), mean_out.data_ptr(),
                                                        mean_out.stride(0), int(b_presplit) | (4 if GEMM_WIDE else 0), _stream(out)), "spgnn_gemm_nt_headmean")


def gemm_nt_add(a: torch.Tensor, b: torch.Tensor, scale_a, scale_b, addend: torch.Tensor, bias: Optional[torch.Tensor] = None,
                act: int = 0, b_presplit: bool = False) -> torch.Tensor:
    """act(a @ b^T + bias + addend): the second product of a pair that shares its output (spgnn_gemm_nt_add)."""
    _require_cuda(a, b, addend)
    M, K = a.shape
    N = b.shape[0]
    assert _rows_aligned(a) and _rows_aligned(b) and addend.shape == (M, N) and _rows_aligned(addend)
    out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    with torch.cuda.device(a.device), _timed("gemm_nt", (M, N, K)):
        _capi.check(_capi.load().spgnn_gemm_nt_add(a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), out.data_ptr(), out.stride(0),
                                                   M, N, K, _ptr(scale_a), _ptr(scale_b), _ptr(bias), act, addend.data_ptr(),
                                                   addend.stride(0), int(b_presplit) | (4 if GEMM_WIDE else 0), _stream(a)), "spgnn_gemm_nt_add")
    return out


def headmean_fusable(out: torch.Tensor, H: int, D: int) -> bool:
    if skinny_rows(out.shape[0]):
        return False                                       # small inference batch: skinny products + spgnn_head_mean
    return H == 2 and D % 4 == 0 and out.stride(0) % 4 == 0 and out.data_ptr() % 16 == 0 and GEMM_MODE == "f16x3"


TN_TILE = 0             # block-tile rows of the weight-gradient kernel: 0 = chosen by the library from the shape, 128 / 256 = pinned (A/B, tests)


def _tn_splits(tiles: int, R: int, rows: int = 128) -> int:
    # Split count.  The kernel deals the (split, tile) work items to the XCDs in contiguous ranges, so any count keeps a
    # split's row range in one or two L2s.  Small products: enough splits for ~512 workgroups (two per CU), at least 256
    # rows each.  Large ones (tools/tn_splits.py, R = 76 410, incl. the partial-sum reduction): 1024 x 1063 (72 tiles)
    # 7 splits 704 us, 14: 634, 21: 609, 28: 605, 32: 598, 35: 608 - whole rounds of workgroups do not matter (the
    # kernel is bound chip-wide, not per CU), fewer splits lose to the shorter pipeline per byte of L2 refill;
    # 1024 x 384 / 512 x 768 (24 tiles) 21: 209 / 209, 32: 215 / 213, 42: 216 / 215, 64: 229 / 226;
    # 256 x 384 (6) 64: 72, 85: 65, 128: 70; 256 x 256 (4) 64: 52, 128: 48, 256: 60; 128 x 128 (1) 64: 41, 256: 26, 512: 29.
    if rows == 256:             # 256 x 128 tiles, one workgroup per CU: half the tiles of the 128-row form, the same row ranges
        # tools/tn_tiles.py, R = 76 410: 1024 x 1063 (36 tiles) 16 splits 664 us, 21: 579, 32: 614, 43: 663; 1024 x 384 / 512 x 768
        # (12 tiles) 16: 232 / 228, 21: 211 / 210, 32: 242 / 243; R = 9 641: 36 tiles 7: 93, 16: 115; 12 tiles 16: 44, 21: 45
        splits = max(1, min(256, 256 // tiles, R // 256))
        if tiles >= 8 and R >= 32 * 512:
            splits = 21
        return splits
    splits = max(1, min(256, 512 // tiles, R // 256))
    if tiles >= 16 and R >= 32 * 512:
        splits = 32 if tiles >= 48 else 21
    return splits


def tn_pair_splits(R: int, shape0, shape1, b_presplit: bool = False):
    """Split counts for two weight-gradient products that will share ONE launch (gemm_tn_pair), or (None, None) to keep each
    product's own count.  Both run in the first product's tile rows; with each product's own count the pair can spill a few
    short workgroups into a second round (R = 9 641: 12 x 21 + 2 x 37 = 326 workgroups on 256 slots).  One common count that
    fits the pair into a single round gives every workgroup the same row range; taken when its range is shorter than the own
    counts' first-round range plus the spilled one."""
    if R >= 32 * 512:          # the large-R counts of _tn_splits are measured optima of the pair as it runs (a common count for the
        return None, None      # 256 x 384 + 128 x 128 pair at R = 76 410 cost the step 1 %)
    (M0, N0), (M1, N1) = shape0, shape1
    flags = int(bool(b_presplit)) | (0x10 if TN_TILE == 128 else 0x20 if TN_TILE == 256 else 0) | (0x100 if GEMM_WIDE else 0)
    lib = _capi.load()
    rows = int(lib.spgnn_gemm_tn_tile_rows(R, M0, N0, flags))
    rows1 = int(lib.spgnn_gemm_tn_tile_rows(R, M1, N1, flags))
    tiles = lambda M, N, r: ((M + r - 1) // r) * ((N + 127) // 128)
    t0, t1 = tiles(M0, N0, rows), tiles(M1, N1, rows)
    s0, s1 = _tn_splits(t0, R, rows), _tn_splits(tiles(M1, N1, rows1), R, rows1)
    slots = 256 if rows == 256 else 512
    if t0 * s0 + t1 * s1 <= slots:
        return None, None
    sj = max(1, min(slots // (t0 + t1), R // 256))
    if R / sj < R / s0 + R / s1:
        return sj, sj
    return None, None


class TnProblem:
    """One gemm_tn call: the partial-tile buffer and outputs are allocated here; ``launch()`` (or :func:`gemm_tn_pair`) runs
    the product, ``finish()`` issues or defers the deterministic partial sums and returns gemm_tn's result."""

    def __init__(self, a: torch.Tensor, b: torch.Tensor, scale_a: Optional[torch.Tensor] = None,
                 scale_b: Optional[torch.Tensor] = None, want_colsum: bool = False, out: Optional[torch.Tensor] = None,
                 out2: Optional[torch.Tensor] = None, colsum_out: Optional[torch.Tensor] = None, defer: Optional["SumJobs"] = None,
                 b_presplit: bool = False, tile: Optional[int] = None, splits: Optional[int] = None):
        _require_cuda(a, b)
        R, M = a.shape
        N = b.shape[1]
        assert b.shape[0] == R and _rows_aligned(a) and _rows_aligned(b)
        tile = TN_TILE if tile is None else tile
        flags = int(bool(b_presplit)) | (0x10 if tile == 128 else 0x20 if tile == 256 else 0) | (0x100 if GEMM_WIDE else 0)   # SPGNN_TN_B_PRESPLIT | _TILE_* | _WIDE
        rows = int(_capi.load().spgnn_gemm_tn_tile_rows(R, M, N, flags))
        tiles = ((M + rows - 1) // rows) * ((N + 127) // 128)
        splits = _tn_splits(tiles, R, rows) if splits is None else int(splits)
        ldn = (N + 3) // 4 * 4
        ldc = ldn + 4 if want_colsum else ldn          # the column sums ride in a spare column: one reduction over splits
        part = torch.empty((splits, M, ldc), dtype=torch.float32, device=a.device)
        cs_ptr = part[0, 0, ldn:].data_ptr() if want_colsum else 0
        # compact outputs: (M, N) contiguous and the column sums as their own vector - autograd takes contiguous gradients
        # over as they are, the row-strided views of the padded tile were cloned once per parameter (18 copies per step)
        if out is None:
            out = torch.empty((M, N), dtype=torch.float32, device=a.device)
        split_col = out.shape[1] if out2 is not None else 0
        assert out.stride(1) == 1 and out.shape[0] == M and (out2 is None or (out2.stride(1) == 1 and out2.shape == (M, N - split_col)))
        cs = (colsum_out if colsum_out is not None else torch.empty((M,), dtype=torch.float32, device=a.device)) if want_colsum else None
        assert cs is None or cs.is_contiguous()
        self.a, self.b, self.part, self.out, self.out2, self.cs, self.defer = a, b, part, out, out2, cs, defer
        self.shape, self.splits, self.ldc, self.ldn, self.split_col, self.want_colsum = (R, M, N), splits, ldc, ldn, split_col, want_colsum
        self.scales = (scale_a, scale_b)
        q = self.c = _capi.GemmTnProblem()
        q.A, q.lda, q.B, q.ldb, q.C, q.ldc, q.split_stride = a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), part.data_ptr(), ldc, M * ldc
        q.R, q.M, q.N, q.scale_a, q.scale_b = R, M, N, _ptr(scale_a), _ptr(scale_b)
        q.colsum_a, q.colsum_stride, q.colsum_split_stride, q.splits = cs_ptr, ldc, M * ldc, splits
        q.flags, self.b_presplit, self.rows = flags, int(bool(b_presplit)), rows      # b_presplit: b = the pre-split image of X (presplit() under scale_b)

    def _side(self):
        """The step's side stream when this product's sums are deferred (nothing reads its partials before a join)."""
        return side_for(self.shape[0], self.a.device) if self.defer is not None else None

    def launch(self):
        import ctypes

        def go():
            with torch.cuda.device(self.a.device), _timed("gemm_tn", self.shape):
                _capi.check(_capi.load().spgnn_gemm_tn_problem_run(ctypes.byref(self.c), _stream(self.a)), "spgnn_gemm_tn_problem_run")
        q = self._side()
        if q is not None:
            q.run(go, self.a, self.b, self.part)
        else:
            go()
        return self

    def finish(self):
        (R, M, N), part, out, out2, cs, ldc = self.shape, self.part, self.out, self.out2, self.cs, self.ldc
        if self.defer is not None:
            self.defer.add(_capi.SumJob(kind=2, splits=self.splits, partials=part.data_ptr(), split_stride=M * ldc, out=out.data_ptr(),
                                        out_stride=out.stride(0), M=M, N=N, ld_in=ldc, out2=_ptr(out2),
                                        out2_stride=out2.stride(0) if out2 is not None else 0, split_col=self.split_col, extra=_ptr(cs),
                                        extra_col=self.ldn if self.want_colsum else 0), part, out, out2, cs)
        else:
            with torch.cuda.device(self.a.device):
                _capi.check(_capi.load().spgnn_sum_partials_compact(part.data_ptr(), M * ldc, self.splits, M, N, ldc, out.data_ptr(),
                                                                    out.stride(0), _ptr(out2), out2.stride(0) if out2 is not None else 0,
                                                                    self.split_col, _ptr(cs), self.ldn if self.want_colsum else 0,
                                                                    _stream(self.a)), "spgnn_sum_partials_compact")
        return (out, cs) if self.want_colsum else out


def gemm_tn(a: torch.Tensor, b: torch.Tensor, scale_a: Optional[torch.Tensor] = None,
            scale_b: Optional[torch.Tensor] = None, want_colsum: bool = False, out: Optional[torch.Tensor] = None,
            out2: Optional[torch.Tensor] = None, colsum_out: Optional[torch.Tensor] = None, defer: Optional["SumJobs"] = None,
            b_presplit: bool = False, tile: Optional[int] = None, splits: Optional[int] = None):
    """a (R,M)^T @ b (R,N) -> (M,N): reduction over the rows of both operands (weight gradients), split-K
    over row chunks with a deterministic partial-sum reduction.  ``want_colsum``: also return a.sum(0) (M,),
    accumulated from the operand stream the kernel reads anyway.  ``out`` [, ``out2``] (row-major views, unit column
    stride): write the result there - with ``out2`` columns [0, out.shape[1]) to ``out`` and the rest to ``out2``;
    ``colsum_out`` (M,) contiguous likewise for the column sums.  ``b_presplit``: ``b`` is the pre-split image of the
    operand (:func:`presplit` under ``scale_b``)."""
    return TnProblem(a, b, scale_a, scale_b, want_colsum, out, out2, colsum_out, defer, b_presplit, tile, splits).launch().finish()


def gemm_tn_pair(first: TnProblem, second: TnProblem):
    """Both weight-gradient products in one launch (spgnn_gemm_tn_pair), then each one's ``finish()`` - bit-identical to two
    gemm_tn calls, which is what runs when PAIR_GEMMS is off."""
    if not PAIR_GEMMS or first.a.device != second.a.device or first.b_presplit != second.b_presplit:
        return first.launch().finish(), second.launch().finish()
    import ctypes

    def go():
        with torch.cuda.device(first.a.device), _timed("gemm_tn_pair", first.shape + second.shape):
            _capi.check(_capi.load().spgnn_gemm_tn_pair(ctypes.byref(first.c), ctypes.byref(second.c), _stream(first.a)), "spgnn_gemm_tn_pair")
    q = first._side() if second._side() is not None else None
    if q is not None:
        q.run(go, first.a, first.b, first.part, second.a, second.b, second.part)
    else:
        go()
    return first.finish(), second.finish()
