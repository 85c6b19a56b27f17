"""bf16-STORAGE operators over the C ABI (include/spgnn_hip.h, entry points ending in ``_bf16``).

BASELINE.json config 4 ("st_gat_6 deep GAT, batch=512 trees, bf16"): node-feature rows, projected rows and their
gradients are ``torch.bfloat16`` tensors in HBM; every kernel accumulates in fp32; parameters (and therefore weight
gradients, the optimizer and the all-reduce), attention scores el/er and attention weights stay fp32.  The reference
itself is fp32 only (its AMP hook, job_runner.py:263-280, is unused by every config), so this path is a build-side
extension and the fp32 path in ops.py stays the parity path.

Layout rules (what the kernels require and every producer here guarantees):
  * rows are 16-byte aligned: row stride a multiple of 8 elements, base pointer 16-byte aligned;
  * columns between the logical width and the width rounded up to 8 hold zeros (the GEMMs read whole 8-element chunks).
There is no CPU implementation; CPU tensors raise like everywhere else in this package.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import _capi
from .graph import DeviceCSC
from .ops import (ACT_NONE, _ell, _pad16, _padded_rows, _ptr, _require_cuda, _rowmajor, _seed_off_ptr, _stream, _timed,
                  scores_from_parts, sum_partials)
from . import ops as _ops

SIDE_TN_BF16 = False      # weight-gradient products of the bf16 layers on the step's side stream (see gemm_tn): measured slower

BF16 = torch.bfloat16


def _pad8(k: int) -> int:
    return (k + 7) // 8 * 8


def rows_ok(t: torch.Tensor) -> bool:
    """bf16 rows a GEMM can read: unit column stride, row stride % 8 == 0 and >= width rounded up to 8, 16-byte aligned."""
    return (t.dtype == BF16 and t.dim() == 2 and t.stride(1) == 1 and t.stride(0) % 8 == 0 and t.stride(0) >= _pad8(t.shape[1])
            and t.data_ptr() % 16 == 0)


def empty_rows(n: int, width: int, device) -> torch.Tensor:
    """(n, width) bf16 view of a buffer whose rows are padded to a multiple of 8 elements; the pad columns are zeroed."""
    wp = _pad8(width)
    buf = torch.empty((n, wp), dtype=BF16, device=device)
    if wp > width:
        buf[:, width:].zero_()
    return buf[:, :width]


def as_rows(x: torch.Tensor) -> torch.Tensor:
    """x as GEMM-readable bf16 rows (a padded copy when its layout does not qualify: never on the hot path)."""
    if rows_ok(x):
        return x
    out = empty_rows(x.shape[0], x.shape[1], x.device)
    out.copy_(x)
    return out


def cast_rows(x: torch.Tensor) -> torch.Tensor:
    """fp32 node data (N, K) -> bf16 rows (round to nearest even), 16-byte rows, zero padded: once per loader batch."""
    _require_cuda(x)
    assert x.dim() == 2 and x.dtype == torch.float32
    if x.stride(1) != 1:
        x = x.contiguous()
    N, K = x.shape
    out = torch.empty((N, _pad8(K)), dtype=BF16, device=x.device)
    with torch.cuda.device(x.device):
        _capi.check(_capi.load().spgnn_cast_rows_bf16(x.data_ptr(), x.stride(0), N, K, out.data_ptr(), out.stride(0), _stream(x)),
                    "spgnn_cast_rows_bf16")
    return out[:, :K]


class _WeightPrepBf16:
    """Persistent bf16 operand buffers and the device job table of spgnn_weight_cat_bf16_multi for one list of layers."""

    def __init__(self, specs, device):
        import ctypes
        lib = _capi.load()
        self.n = len(specs)
        tab = (_capi.WeightCatBf16Job * self.n)()
        self.entries, first = [], 0
        for i, (w_a, w_b, want_t) in enumerate(specs):
            R1, K = w_a.shape
            R2 = 0 if w_b is None else w_b.shape[0]
            R = R1 + R2
            w = torch.empty((R, _pad8(K)), dtype=BF16, device=device)
            w_t = torch.empty((K, _pad8(R)), dtype=BF16, device=device) if want_t else None
            t = tab[i]
            t.a, t.a_stride = w_a.data_ptr(), w_a.stride(0)
            t.b, t.b_stride = (w_b.data_ptr(), w_b.stride(0)) if w_b is not None else (0, 0)
            t.w, t.w_stride = w.data_ptr(), w.stride(0)
            t.w_t, t.w_t_stride = (w_t.data_ptr(), w_t.stride(0)) if want_t else (0, 0)
            t.rows_a, t.rows_b, t.K, t.first_block = R1, R2, K, first
            tx = ctypes.c_int32(0)
            first += int(lib.spgnn_weight_cat_bf16_blocks(R, K, w.stride(0), w_t.stride(0) if want_t else 0, ctypes.addressof(tx)))
            t.tiles_x = tx.value
            self.entries.append((w[:, :K], w_t[:, :R] if want_t else None))
        self.blocks = first
        self.table = torch.frombuffer(bytearray(bytes(memoryview(tab))), dtype=torch.uint8).to(device)

    def run(self):
        with torch.cuda.device(self.table.device):
            _capi.check(_capi.load().spgnn_weight_cat_bf16_multi(self.table.data_ptr(), self.n, self.blocks, _stream(self.table)),
                        "spgnn_weight_cat_bf16_multi")


_PREP_CACHE: dict = {}       # layout key -> _WeightPrepBf16 (buffers and table are reused by every forward pass)
_PREP_ACTIVE: dict = {}      # (id(w_a), id(w_b)) -> ((w, w_t), want_t): installed for the duration of one model forward
SCORES_DIRECT = True         # D = 64 layers: el / er straight from the projection's epilogue (no spgnn_scores_from_parts launch)
BATCH_WEIGHT_PREP = True     # every project-first layer's bf16 operands in one launch per forward; False: one launch per layer


class prepared_weights:
    """``with prepared_weights(specs):`` - ``specs`` = [(w_a, w_b or None, want_t), ...] of the project-first GATConv layers a
    forward pass will run on bf16 rows: ONE spgnn_weight_cat_bf16_multi launch builds all their operands (bit-identical to one
    spgnn_weight_cat_bf16 per layer); inside the block :func:`weight_operands` hands them out without launching anything.
    Values are those of the parameters at entry (the block must not update them)."""

    def __init__(self, specs):
        self.specs = [tuple(sp[:3]) for sp in specs if len(sp) == 3 and sp[0].is_cuda and sp[0].dtype == torch.float32
                      and sp[0].stride(1) == 1 and (sp[1] is None or sp[1].stride(1) == 1)] if BATCH_WEIGHT_PREP else []
        self.prev = None

    def __enter__(self):
        global _PREP_ACTIVE
        self.prev = _PREP_ACTIVE
        if not self.specs:
            return self
        dev = self.specs[0][0].device
        key = (str(dev),) + tuple((a.data_ptr(), a.stride(0), tuple(a.shape), 0 if b is None else b.data_ptr(),
                                   0 if b is None else b.stride(0), None if b is None else tuple(b.shape), bool(t))
                                  for a, b, t in self.specs)
        prep = _ops._prep_lookup(_PREP_CACHE, key, lambda: _WeightPrepBf16(self.specs, dev))     # LRU; pinned once a HIP graph holds it
        prep.run()
        _PREP_ACTIVE = {(id(a), id(b) if b is not None else 0): (e, bool(t)) for (a, b, t), e in zip(self.specs, prep.entries)}
        return self

    def __exit__(self, *exc):
        global _PREP_ACTIVE
        _PREP_ACTIVE = self.prev
        return False


def weight_operands(w_a: torch.Tensor, w_b: Optional[torch.Tensor], want_t: bool) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """[w_a ; w_b] (fp32 parameters) -> (W (R, K) bf16 rows, W^T (K, R) bf16 rows or None): from the forward pass's
    :class:`prepared_weights` launch when there is one, else one kernel per layer and step."""
    hit = _PREP_ACTIVE.get((id(w_a), id(w_b) if w_b is not None else 0))
    if hit is not None and (hit[1] or not want_t):
        return hit[0][0], (hit[0][1] if want_t else None)
    R1, K = w_a.shape
    R2 = 0 if w_b is None else w_b.shape[0]
    R = R1 + R2
    wa = w_a if w_a.stride(1) == 1 else w_a.contiguous()
    wb = None if w_b is None else (w_b if w_b.stride(1) == 1 else w_b.contiguous())
    w = torch.empty((R, _pad8(K)), dtype=BF16, device=w_a.device)
    w_t = torch.empty((K, _pad8(R)), dtype=BF16, device=w_a.device) if want_t else None
    with torch.cuda.device(w_a.device):
        _capi.check(_capi.load().spgnn_weight_cat_bf16(wa.data_ptr(), wa.stride(0), R1, _ptr(wb), 0 if wb is None else wb.stride(0),
                                                       R2, K, w.data_ptr(), w.stride(0), _ptr(w_t),
                                                       w_t.stride(0) if want_t else 0, _stream(w_a)), "spgnn_weight_cat_bf16")
    return w[:, :K], (w_t[:, :R] if want_t else None)


def gemm_nt(a: torch.Tensor, b: torch.Tensor, out: Optional[torch.Tensor] = None, out_f32: bool = False,
            bias: Optional[torch.Tensor] = None, act: int = ACT_NONE, score_l: Optional[torch.Tensor] = None,
            score_r: Optional[torch.Tensor] = None, score_out: Optional[torch.Tensor] = None, tile: int = 0,
            score_direct: bool = False) -> torch.Tensor:
    """a (M,K) @ b (N,K)^T -> (M,N) bf16 rows (or fp32 with ``out_f32``); bf16 MFMA, fp32 accumulate.  ``bias`` (N,) fp32 /
    ``act``: epilogue act(C + bias).  ``score_out`` (M, C/64, 2) fp32 with ``score_l`` / ``score_r`` (C,) fp32: per
    64-column block dot products of the first C output columns, taken from the values as stored."""
    _require_cuda(a, b)
    M, K = a.shape
    N = b.shape[0]
    assert b.shape[1] == K and rows_ok(a) and rows_ok(b), "bf16 GEMM operands must be 16-byte-row bf16 tensors"
    assert N % 4 == 0, "bf16 GEMM output width must be a multiple of 4"
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=a.device) if out_f32 else empty_rows(M, N, a.device)
    assert out.shape == (M, N) and out.stride(1) == 1 and out.dtype == (torch.float32 if out_f32 else BF16)
    with torch.cuda.device(a.device), _timed("gemm_nt_bf16", (M, N, K)):
        args = (a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), out.data_ptr(), out.stride(0), int(out_f32), M, N, K, _ptr(bias),
                act, _ptr(score_l), _ptr(score_r), _ptr(score_out), score_l.numel() if score_out is not None else 0)
        if score_direct:                           # score_out is the (M, 2H) [el | er] tensor itself (heads of one 64-column block)
            assert score_out is not None and bias is None and act == ACT_NONE and not tile
            _capi.check(_capi.load().spgnn_gemm_nt_bf16_scores(*args[:10], *args[12:], 1, _stream(a)), "spgnn_gemm_nt_bf16_scores")
        elif tile:                                 # block tile pinned by the caller (2 / 4 / 5): bit-identical results
            _capi.check(_capi.load().spgnn_gemm_nt_bf16_tile(*args, tile, _stream(a)), "spgnn_gemm_nt_bf16_tile")
        else:
            _capi.check(_capi.load().spgnn_gemm_nt_bf16(*args, _stream(a)), "spgnn_gemm_nt_bf16")
    return out


def _tn_splits(R: int, M: int, N: int) -> int:
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    splits = max(1, min(128, 512 // tiles, R // 256))       # one- to four-tile products: 128 row ranges (256 x 128: 27 -> 22 us; tools/tn_splits_bf16.py)
    if splits >= 8 or tiles >= 16:          # one split per XCD (see ops.gemm_tn)
        splits = 32 if tiles >= 16 and R >= 32 * 512 else (splits // 8 * 8 if splits >= 8 else splits)
        if tiles >= 64 and splits == 32:       # 1024 x 1024: 16 splits 232 us, 32: 244 (tools/tn_splits_bf16.py): the partial tiles weigh more
            splits = 16
    return splits


def gemm_tn(a: torch.Tensor, b: torch.Tensor, want_colsum: bool = False, splits: Optional[int] = None, defer=None):
    """a (R,M)^T @ b (R,N) -> (M,N) fp32 (weight gradients): bf16 MFMA, fp32 accumulate, split over row ranges with a
    deterministic partial-sum reduction.  ``want_colsum``: also a.sum(0) (M,) fp32 from the operand stream."""
    _require_cuda(a, b)
    R, M = a.shape
    N = b.shape[1]
    assert b.shape[0] == R and rows_ok(a) and rows_ok(b)
    lib = _capi.load()
    splits = splits or _tn_splits(R, M, N)
    ldn = (N + 3) // 4 * 4
    ldc = ldn + 4 if want_colsum else ldn
    part = torch.empty((splits, M, ldc), dtype=torch.float32, device=a.device)
    cs_ptr = part[0, 0, ldn:].data_ptr() if want_colsum else 0

    def go():
        with torch.cuda.device(a.device), _timed("gemm_tn_bf16", (R, M, N)):
            _capi.check(lib.spgnn_gemm_tn_bf16(a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), part.data_ptr(), ldc, M * ldc,
                                               splits, R, M, N, cs_ptr, ldc, M * ldc, _stream(a)), "spgnn_gemm_tn_bf16")
    # a training step's side stream (ops.SideLaunch) - measured SLOWER on bf16 rows (st_gat_6, 512 trees: 2.117 vs 2.074 ms per
    # step: the single-product weight gradients are too short for the fork / join to pay), so off unless SIDE_TN_BF16 is set
    side = _ops.side_for(R, a.device) if (defer is not None and SIDE_TN_BF16) else None
    if side is not None:
        side.run(go, a, b, part)
    else:
        go()
    out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    cs = torch.empty((M,), dtype=torch.float32, device=a.device) if want_colsum else None
    if defer is not None:                     # ops.SumJobs: the reduction joins the caller's other ones in one launch
        defer.add(_capi.SumJob(kind=2, splits=splits, partials=part.data_ptr(), split_stride=M * ldc, out=out.data_ptr(),
                               out_stride=out.stride(0), M=M, N=N, ld_in=ldc, extra=_ptr(cs), extra_col=ldn if want_colsum else 0),
                  part, out, cs)
    else:
        with torch.cuda.device(a.device):
            _capi.check(lib.spgnn_sum_partials_compact(part.data_ptr(), M * ldc, splits, M, N, ldc, out.data_ptr(), out.stride(0),
                                                       0, 0, 0, _ptr(cs), ldn if want_colsum else 0, _stream(a)),
                        "spgnn_sum_partials_compact")
    return (out, cs) if want_colsum else out


def attn_vector_grads(g_s: torch.Tensor, ft: torch.Tensor, H: int, defer=None) -> torch.Tensor:
    """(2, H, D): [w, h, :] = g_s[:, w*H + h]^T @ ft[:, h*D:(h+1)*D] - the gradients of attn_l / attn_r (fp32), ft bf16."""
    N, K = ft.shape
    J = g_s.shape[1]
    D = K // H
    Kp = (K + 15) // 16 * 16
    splits = max(1, min(1024 // ((K + 255) // 256), N // 16))
    part = torch.empty((splits, J, Kp), dtype=torch.float32, device=ft.device)
    lib = _capi.load()
    out = torch.empty((2, H, D), dtype=torch.float32, device=ft.device)
    with torch.cuda.device(ft.device), _timed("scores_bwd_w_bf16", (N, K, J)):
        _capi.check(lib.spgnn_scores_bwd_w_bf16(g_s.data_ptr(), g_s.stride(0), ft.data_ptr(), ft.stride(0), part.data_ptr(), splits,
                                                Kp, N, K, J, _stream(ft)), "spgnn_scores_bwd_w_bf16")
        if defer is not None:
            defer.add(_capi.SumJob(kind=1, splits=splits, partials=part.data_ptr(), split_stride=J * Kp, out=out.data_ptr(), H=H, D=D, ld=Kp),
                      part, out)
        else:
            _capi.check(lib.spgnn_sum_partials_blockdiag(part.data_ptr(), J * Kp, splits, H, D, Kp, out.data_ptr(), _stream(ft)),
                        "spgnn_sum_partials_blockdiag")
    return out


def gat_fwd_raw(csc: DeviceCSC, ft, el, er, res, bias, H: int, D: int, slope: float, act: int, p_drop: float, seed: int,
                mean: bool, need_out: bool, out_drop=None):
    """-> (out (N, H*D) bf16 or None, out_mean (N, D) fp32 or None, attn (E, H) fp32).  ``out_drop`` = (p, seed, total,
    offset): the stored rows carry the consumer's feature dropout (ops.gat_fwd_raw)."""
    N, E = csc.num_nodes, csc.num_edges
    lib = _capi.load()
    fuse = mean and bool(lib.spgnn_gat_can_fuse_mean(H, D))
    skip_out = fuse and not need_out
    out = None if skip_out else torch.empty((N, H * D), dtype=BF16, device=ft.device)
    out_mean = torch.empty((N, D), dtype=torch.float32, device=ft.device) if fuse else None
    attn = torch.empty((E, H), dtype=torch.float32, device=ft.device)
    plan = None if (mean or out is None) else _ops.tile_plan(csc, H, D, 2, "fwd")
    if plan is not None:                        # tree-resident LDS tiles (csrc/spgnn_tile.hip): narrow rows
        with torch.cuda.device(ft.device), _timed("gat_fwd_bf16", (N, E, H, D, int(res is not None), 0, 1)):
            _capi.check(lib.spgnn_gat_fwd_tile_bf16(plan[0].data_ptr(), plan[1], plan[2], csc.indptr.data_ptr(), _ell(csc)[0], ft.data_ptr(),
                                                    ft.stride(0), el.data_ptr(), er.data_ptr(), el.stride(0), _ptr(res),
                                                    res.stride(0) if res is not None else 0, _ptr(bias), out.data_ptr(), out.stride(0),
                                                    attn.data_ptr(), 0, N, H, D, slope, act, p_drop, seed, _seed_off_ptr(ft.device),
                                                    *(out_drop or (0.0, 0, 0, 0)), _stream(ft)), "spgnn_gat_fwd_tile_bf16")
        return out, out_mean, attn
    with torch.cuda.device(ft.device), _timed("gat_fwd_bf16", (N, E, H, D, int(res is not None), int(fuse), int(out is not None))):
        _capi.check(lib.spgnn_gat_fwd_bf16(csc.indptr.data_ptr(), csc.indices.data_ptr(), _ell(csc)[0], ft.data_ptr(), ft.stride(0),
                                           el.data_ptr(), er.data_ptr(), el.stride(0), _ptr(res),
                                           res.stride(0) if res is not None else 0, _ptr(bias), _ptr(out),
                                           out.stride(0) if out is not None else 0, _ptr(out_mean),
                                           out_mean.stride(0) if fuse else 0, attn.data_ptr(), N, E, H, D, slope, act, p_drop,
                                           seed, _seed_off_ptr(ft.device), *(out_drop or (0.0, 0, 0, 0)), _stream(ft)),
                    "spgnn_gat_fwd_bf16")
    return out, out_mean, attn


class _GATLayerBf16Fn(torch.autograd.Function):
    """Project-first GATConv on bf16 rows: Y = X [W_fc ; W_res]^T on the bf16 matrix cores with el / er from the stored ft
    in the GEMM epilogue (DGL's (ft * attn).sum(-1)), spgnn_gat_fwd_bf16, and in the backward pass spgnn_gat_bwd_dst/src_bf16
    writing [g_ft | g_pre] straight into the GEMM-gradient buffer, one weight-gradient product (fp32 result, bias gradient
    from its operand stream) and one input-gradient product.  The fp32 parameters enter here directly: their gradients must
    stay fp32 (autograd would cast the gradient of a bf16 weight tensor to bf16)."""

    @staticmethod
    def forward(ctx, x, w_fc, w_res, attn_l, attn_r, bias, csc: DeviceCSC, H: int, D: int, slope: float, act: int,
                p_drop: float, seed: int, mean: bool, out_drop):
        """``out_drop`` = None or (p, seed): the rows are stored under the NEXT layer's feature dropout (nn.GATConv
        fuse_out with total == H*D): no separate dropout pass over them."""
        ctx.set_materialize_grads(False)
        HD = H * D
        N = x.shape[0]
        has_res = w_res is not None
        need_gx = ctx.needs_input_grad[0]
        w, w_t = weight_operands(w_fc, w_res, want_t=need_gx)
        ctx.attn_shape = attn_l.shape
        ctx.attn_params = (attn_l, attn_r)
        al, ar = attn_l.reshape(-1).contiguous(), attn_r.reshape(-1).contiguous()
        if D == 64 and SCORES_DIRECT:            # a head is one 64-column block: the epilogue's dots ARE el / er
            s = torch.empty((N, 2 * H), dtype=torch.float32, device=x.device)
            y = gemm_nt(x, w, score_l=al, score_r=ar, score_out=s, score_direct=True)
        else:
            parts = torch.empty((N, HD // 64, 2), dtype=torch.float32, device=x.device)
            y = gemm_nt(x, w, score_l=al, score_r=ar, score_out=parts)
            s = scores_from_parts(parts, H, D)
        ft = y[:, :HD]
        res = y[:, HD:] if has_res else None
        od = None if out_drop is None else (float(out_drop[0]), int(out_drop[1]), HD, 0)
        out, out_mean, attn = gat_fwd_raw(csc, ft, s[:, :H], s[:, H:], res, bias, H, D, slope, act, p_drop, seed, mean,
                                          need_out=(act != ACT_NONE) or od is not None, out_drop=od)
        fused_mean = out_mean is not None
        ctx.out_drop = od
        ctx.csc, ctx.cfg = csc, (H, D, has_res, slope, act, p_drop, seed, fused_mean)
        ctx.has_bias = bias is not None
        ctx.save_for_backward(x, w_t, al, ar, y, s, attn, out if act != ACT_NONE else None)
        ctx.mark_non_differentiable(attn)
        return (out_mean if fused_mean else out), attn

    @staticmethod
    def backward(ctx, g_out, _g_attn):
        if g_out is None:
            return (None,) * 15
        x, w_t, al, ar, y, s, attn, out = ctx.saved_tensors
        H, D, has_res, slope, act, p_drop, seed, mean = ctx.cfg
        csc = ctx.csc
        HD = H * D
        N, K = x.shape
        E = csc.num_edges
        if mean:
            g_out = g_out.float() if g_out.dtype != torch.float32 else g_out
        elif g_out.dtype != BF16:
            g_out = g_out.to(BF16)
        if g_out.stride(1) != 1 or g_out.stride(0) % 4 or g_out.data_ptr() % 16:
            g_out = g_out.contiguous()
        g_y = torch.empty_like(y)
        g_s = torch.empty_like(s)
        g_pre = g_y[:, HD:] if has_res else torch.empty((N, HD), dtype=BF16, device=x.device)
        g_e = torch.empty((E, H), dtype=torch.float32, device=x.device)
        lib = _capi.load()
        plan_d = None if mean else _ops.tile_plan(csc, H, D, 2, "dst")
        plan_s = _ops.tile_plan(csc, H, D, 2, "src")        # tree-resident LDS tiles where they are faster (ops.TILE_TABLE)
        with torch.cuda.device(x.device):
            st = _stream(x)
            ell = _ell(csc)
            with _timed("gat_bwd_dst_bf16", (N, E, H, D, act, int(mean))):
                if plan_d is not None:
                    _capi.check(lib.spgnn_gat_bwd_dst_tile_bf16(plan_d[0].data_ptr(), plan_d[1], plan_d[2], csc.indptr.data_ptr(), ell[0],
                                                                y.data_ptr(), y.stride(0), s.data_ptr(), s[:, H:].data_ptr(), s.stride(0),
                                                                attn.data_ptr(), g_out.data_ptr(), g_out.stride(0), _ptr(out),
                                                                out.stride(0) if out is not None else 0, g_pre.data_ptr(), g_pre.stride(0),
                                                                g_e.data_ptr(), g_s[:, H:].data_ptr(), g_s.stride(0), 0, N, H, D, slope, act,
                                                                p_drop, seed, _seed_off_ptr(x.device), *(ctx.out_drop or (0.0, 0, 0, 0)), st),
                                "spgnn_gat_bwd_dst_tile_bf16")
                else:
                    _capi.check(lib.spgnn_gat_bwd_dst_bf16(csc.indptr.data_ptr(), csc.indices.data_ptr(), ell[0], y.data_ptr(), y.stride(0),
                                                           s.data_ptr(), s[:, H:].data_ptr(), s.stride(0), attn.data_ptr(),
                                                           g_out.data_ptr(), g_out.stride(0), int(mean), _ptr(out),
                                                           out.stride(0) if out is not None else 0, g_pre.data_ptr(), g_pre.stride(0),
                                                           g_e.data_ptr(), g_s[:, H:].data_ptr(), g_s.stride(0), N, E, H, D, slope, act,
                                                           p_drop, seed, _seed_off_ptr(x.device), *(ctx.out_drop or (0.0, 0, 0, 0)), st),
                                "spgnn_gat_bwd_dst_bf16")
            with _timed("gat_bwd_src_bf16", (N, E, H, D)):
                if plan_s is not None:
                    _capi.check(lib.spgnn_gat_bwd_src_tile_bf16(plan_s[0].data_ptr(), plan_s[1], plan_s[2], csc.indptr.data_ptr(),
                                                                csc.out_indptr.data_ptr(), ell[1], ell[2], attn.data_ptr(), g_e.data_ptr(),
                                                                g_pre.data_ptr(), g_pre.stride(0), g_y.data_ptr(), g_y.stride(0),
                                                                g_s.data_ptr(), g_s.stride(0), 0, al.data_ptr(), ar.data_ptr(),
                                                                g_s[:, H:].data_ptr(), N, H, D, p_drop, seed, _seed_off_ptr(x.device), st),
                                "spgnn_gat_bwd_src_tile_bf16")
                else:
                    _capi.check(lib.spgnn_gat_bwd_src_bf16(csc.out_indptr.data_ptr(), csc.out_indices.data_ptr(),
                                                           csc.out_pos.data_ptr(), ell[1], ell[2], attn.data_ptr(), g_e.data_ptr(), g_pre.data_ptr(),
                                                           g_pre.stride(0), g_y.data_ptr(), g_y.stride(0), g_s.data_ptr(),
                                                           g_s.stride(0), al.data_ptr(), ar.data_ptr(), g_s[:, H:].data_ptr(), N, E, H, D,
                                                           p_drop, seed, _seed_off_ptr(x.device), st), "spgnn_gat_bwd_src_bf16")
        need_bias = ctx.has_bias and ctx.needs_input_grad[5]
        g_wfc = g_wres = g_bias = None
        from .ops import SumJobs
        g_x = None
        nt_first = SIDE_TN_BF16 and ctx.needs_input_grad[0] and _ops.side_for(N, x.device) is not None
        if nt_first:                                    # the critical-path product first; the weight gradient then runs on the side
            g_x = gemm_nt(g_y, w_t)                     # stream next to the following layer's traversals (ops.SideLaunch)
        jobs = SumJobs(x.device)                        # the layer's two split-K reductions in one launch
        if ctx.needs_input_grad[1] or (has_res and ctx.needs_input_grad[2]):
            if need_bias and has_res:                   # column sums of g_pre ride along with the operand stream
                g_w, cs = gemm_tn(g_y, x, want_colsum=True, defer=jobs)
                g_bias = cs[HD:]
            else:
                g_w = gemm_tn(g_y, x, defer=jobs)
            g_wfc = g_w[:HD]
            g_wres = g_w[HD:] if has_res else None
        if need_bias and g_bias is None:
            g_bias = g_pre.float().sum(0)
        g_al = g_ar = None
        if ctx.needs_input_grad[3] and ctx.needs_input_grad[4] and _ops.queue_attn_grads(g_s, y[:, :HD], H, *ctx.attn_params):
            pass                                        # one launch for every layer's pass after the backward (ops.AttnGradQueue)
        elif ctx.needs_input_grad[3] or ctx.needs_input_grad[4]:
            m = attn_vector_grads(g_s, y[:, :HD], H, defer=jobs)
            g_al, g_ar = m[0].view(ctx.attn_shape), m[1].view(ctx.attn_shape)
        jobs.flush()
        if ctx.needs_input_grad[0] and not nt_first:
            g_x = gemm_nt(g_y, w_t)
        return g_x, g_wfc, g_wres, g_al, g_ar, g_bias, None, None, None, None, None, None, None, None, None


def gat_layer_supported(x: torch.Tensor, H: int, D: int) -> bool:
    """The bf16 project-first layer needs 64-column score blocks (D % 64 == 0) and an input gradient width the GEMM can
    write (K % 4 == 0 when the input needs a gradient)."""
    return (x.is_cuda and x.dtype == BF16 and x.dim() == 2 and x.shape[0] > 0 and D % 64 == 0
            and (not x.requires_grad or x.shape[1] % 4 == 0))


def gat_layer(csc: DeviceCSC, x, w_fc, w_res, attn_l, attn_r, bias, H: int, D: int, slope: float, act: int,
              p_drop: float = 0.0, seed: int = 0, mean: bool = False, out_drop=None):
    """-> (out (N, H*D) bf16, or the fp32 head mean (N, D) when ``mean`` and the geometry fuses it; attn (E, H) fp32).
    ``out_drop`` = (p, seed): the rows are stored under the next layer's feature dropout."""
    _require_cuda(x, w_fc, w_res, attn_l, attn_r, bias)
    return _GATLayerBf16Fn.apply(as_rows(x), w_fc, w_res, attn_l, attn_r, bias, csc, H, D, slope, act, p_drop, seed, mean,
                                 out_drop)


# --------------------------------------------------------------------------------------------
# GAT output layer without activation, heads averaged: one product on [z_0 | .. | z_{H-1} | x]  (ops._GATAggregateFn)
# --------------------------------------------------------------------------------------------
def scores_fwd(x: torch.Tensor, w_lr: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """S = x @ w_lr^T (N, J) fp32, x bf16 rows, w_lr (J, K) fp32 (the attention vectors folded through W_fc)."""
    N, K = x.shape
    J = w_lr.shape[0]
    Kp = _pad16(K)
    w_p = _padded_rows(w_lr, Kp)
    s = torch.empty((N, J), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device), _timed("scores_fwd_bf16", (N, K, J)):
        _capi.check(_capi.load().spgnn_scores_fwd_bf16(x.data_ptr(), x.stride(0), w_p.data_ptr(), Kp, s.data_ptr(), s.stride(0),
                                                       _ptr(None if bias is None else bias.detach().contiguous()), N, K, J, _stream(x)),
                    "spgnn_scores_fwd_bf16")
    return s


def scores_bwd_w(g_s: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """g_w_lr = g_s^T @ x (J, K) fp32, x bf16 rows; J <= 8."""
    N, K = x.shape
    J = g_s.shape[1]
    Kp = _pad16(K)
    splits = max(1, min((2048 if J > 8 else 1024) // ((K + 255) // 256), N // 16))      # as ops.scores_bwd_w
    part = torch.empty((splits, J, Kp), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device), _timed("scores_bwd_w_bf16", (N, K, J)):
        _capi.check(_capi.load().spgnn_scores_bwd_w_bf16(g_s.data_ptr(), g_s.stride(0), x.data_ptr(), x.stride(0), part.data_ptr(),
                                                         splits, Kp, N, K, J, _stream(x)), "spgnn_scores_bwd_w_bf16")
    return sum_partials(part)[:, :K]


def linear_mean_supported(x: torch.Tensor, H: int, F_in: int) -> bool:
    """bf16 rows the aggregate kernels and the GEMM can read: F % 8 == 0 keeps every block of Zx 16-byte aligned."""
    return (x.is_cuda and x.dtype == BF16 and x.dim() == 2 and x.shape[0] > 0 and F_in % 8 == 0 and 2 * H <= 8
            and bool(_capi.load().spgnn_gat_agg_supported(H, F_in)))


class _GATAggregateBf16Fn(torch.autograd.Function):
    """ops._GATAggregateFn on bf16 rows: x (N, F) -> Zx = [z_0 | .. | z_{H-1} | x] (N, (H+1) F) bf16, scores and attention
    fp32.  ``w_lr`` (2H, F) fp32."""

    @staticmethod
    def forward(ctx, x, w_lr, csc: DeviceCSC, H: int, slope: float, p_drop: float, seed: int):
        ctx.set_materialize_grads(False)
        N, F_ = x.shape
        E = csc.num_edges
        s = scores_fwd(x, w_lr)
        zx = empty_rows(N, (H + 1) * F_, x.device)
        attn = torch.empty((E, H), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device), _timed("gat_agg_fwd_bf16", (N, E, H, F_, 2)):
            _capi.check(_capi.load().spgnn_gat_agg_fwd_bf16(csc.indptr.data_ptr(), csc.indices.data_ptr(), x.data_ptr(), x.stride(0),
                                                            s.data_ptr(), s[:, H:].data_ptr(), s.stride(0), attn.data_ptr(),
                                                            zx.data_ptr(), zx.stride(0), F_, -2, N, E, H, F_, slope, p_drop, seed,
                                                            _seed_off_ptr(x.device), _stream(x)), "spgnn_gat_agg_fwd_bf16")   # -2: x's copy behind the last block
        ctx.csc, ctx.cfg = csc, (H, slope, p_drop, seed)
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
        if g_zx.dtype != BF16 or g_zx.stride(1) != 1 or g_zx.stride(0) % 4 or g_zx.data_ptr() % 8:
            g_zx = g_zx.to(BF16).contiguous()
        g_s = torch.empty_like(s)
        g_e = torch.empty((E, H), dtype=torch.float32, device=x.device)
        g_x = empty_rows(N, F_, x.device)
        w_lr_c = w_lr.contiguous()
        lib = _capi.load()
        with torch.cuda.device(x.device):
            st = _stream(x)
            with _timed("gat_agg_bwd_dst_bf16", (N, E, H, F_)):
                _capi.check(lib.spgnn_gat_agg_bwd_dst_bf16(csc.indptr.data_ptr(), csc.indices.data_ptr(), x.data_ptr(), x.stride(0),
                                                           s.data_ptr(), s[:, H:].data_ptr(), s.stride(0), attn.data_ptr(),
                                                           g_zx.data_ptr(), g_zx.stride(0), F_, g_e.data_ptr(),
                                                           g_s[:, H:].data_ptr(), g_s.stride(0), N, E, H, F_, slope, p_drop, seed,
                                                           _seed_off_ptr(x.device), st), "spgnn_gat_agg_bwd_dst_bf16")
            with _timed("gat_agg_bwd_src_bf16", (N, E, H, F_)):
                _capi.check(lib.spgnn_gat_agg_bwd_src_bf16(csc.out_indptr.data_ptr(), csc.out_indices.data_ptr(),
                                                           csc.out_pos.data_ptr(), attn.data_ptr(), g_e.data_ptr(),
                                                           g_zx.data_ptr(), g_zx.stride(0), F_, -1, g_s[:, H:].data_ptr(),
                                                           w_lr_c.data_ptr(), w_lr_c.stride(0), g_x.data_ptr(), g_x.stride(0),
                                                           g_s.data_ptr(), g_s.stride(0), N, E, H, F_, p_drop, seed,
                                                           _seed_off_ptr(x.device), st), "spgnn_gat_agg_bwd_src_bf16")
        g_wlr = scores_bwd_w(g_s, x) if ctx.needs_input_grad[1] else None
        if ctx.needs_input_grad[0]:
            g_x += g_zx[:, H * F_:]                      # the copy of x inside Zx (fp32 add, one rounding)
        return (g_x if ctx.needs_input_grad[0] else None), g_wlr, None, None, None, None, None


class _LinearBf16Fn(torch.autograd.Function):
    """x (N, K) bf16 rows, fp32 ``weight`` (C, K) / ``bias`` (C,) -> x W^T + b as an fp32 (N, C) tensor: bf16 MFMA, fp32
    accumulate.  Backward: the incoming fp32 gradient is rounded to bf16 rows once (it is a GEMM operand twice), input
    gradient as bf16 rows, weight / bias gradients fp32."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        w, w_t = weight_operands(weight, None, want_t=ctx.needs_input_grad[0])
        y = gemm_nt(x, w, out_f32=True, bias=bias)
        ctx.has_bias = bias is not None
        ctx.save_for_backward(x, w_t)
        return y

    @staticmethod
    def backward(ctx, g):
        x, w_t = ctx.saved_tensors
        gb = cast_rows(g if g.dtype == torch.float32 else g.float())
        g_x = g_w = g_b = None
        if ctx.needs_input_grad[0]:
            g_x = gemm_nt(gb, w_t)
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            if ctx.has_bias:
                g_w, g_b = gemm_tn(gb, x, want_colsum=True)
            else:
                g_w = gemm_tn(gb, x)
        return g_x, g_w, g_b


def scores_bwd_x(g_s: torch.Tensor, w: torch.Tensor, K: int) -> torch.Tensor:
    """g_x = g_s @ w (N, K) as bf16 rows; g_s (N, J <= 32) fp32, w (J, K) fp32."""
    N, J = g_s.shape
    Kp = _pad16(K)
    w_p = _padded_rows(w, Kp)
    g_x = torch.empty((N, _pad8(K)), dtype=BF16, device=g_s.device)
    if _pad8(K) > (K + 3) // 4 * 4:
        g_x[:, (K + 3) // 4 * 4:].zero_()
    with torch.cuda.device(g_s.device), _timed("scores_bwd_x_bf16", (N, K, J)):
        _capi.check(_capi.load().spgnn_scores_bwd_x_bf16(g_s.data_ptr(), g_s.stride(0), w_p.data_ptr(), Kp, g_x.data_ptr(), g_x.stride(0),
                                                         0, N, K, J, _stream(g_s)), "spgnn_scores_bwd_x_bf16")
    return g_x[:, :K]


class _LinearClassifierBf16Fn(torch.autograd.Function):
    """ops._LinearClassifierFn on bf16 rows: y = x W^T + b as an fp32 (N, C) tensor (bf16 MFMA) and the classifier folded
    through the product, logits = x (Wc W)^T + (Wc b + bc) (fp32 skinny kernel on the bf16 rows).  When only the logits
    carry a gradient, g_x = g_logits (Wc W) is written as bf16 rows straight from the fp32 logit gradient and
    g_W = Wc^T (g_logits^T x): no (N, C) gradient, no rounding of it.  W here is the bf16 operand the forward product
    used, so the gradients are those of the function as evaluated."""

    @staticmethod
    def forward(ctx, x, weight, bias, w_cls, b_cls):
        ctx.set_materialize_grads(False)
        w, w_t = weight_operands(weight, None, want_t=ctx.needs_input_grad[0])
        y = gemm_nt(x, w, out_f32=True, bias=bias)
        wc = w_cls.detach()
        wf = w.float()
        P = torch.mm(wc, wf)
        logits = scores_fwd(x, P)
        c0 = b_cls
        if bias is not None:
            c0 = torch.mv(wc, bias.detach()) if c0 is None else torch.addmv(c0.detach(), wc, bias.detach())
        if c0 is not None:
            logits += c0
        ctx.has_bias, ctx.has_bcls = bias is not None, b_cls is not None
        ctx.save_for_backward(x, wf, w_t, P, w_cls, bias)
        return y, logits

    @staticmethod
    def backward(ctx, g_y, g_logits):
        if g_y is None and g_logits is None:
            return None, None, None, None, None
        x, wf, w_t, P, w_cls, bias = ctx.saved_tensors
        K = x.shape[1]
        wc = w_cls.detach()
        g_x = g_w = g_b = g_wcls = g_bcls = cs = M1 = None
        if g_logits is not None:
            cs = _ops.column_sums(g_logits)
            g_logits = _rowmajor(g_logits)
            M1 = scores_bwd_w(g_logits, x)
            if ctx.needs_input_grad[3]:
                g_wcls = torch.mm(M1, wf.t())
                if ctx.has_bias:
                    g_wcls.addr_(cs, bias.detach())
            g_bcls = cs if ctx.has_bcls and ctx.needs_input_grad[4] else None
        if g_y is None:
            if ctx.needs_input_grad[0]:
                g_x = scores_bwd_x(g_logits, P, K)
            if ctx.needs_input_grad[1]:
                g_w = torch.mm(wc.t(), M1)
            if ctx.has_bias and ctx.needs_input_grad[2]:
                g_b = torch.mv(wc.t(), cs)
            return g_x, g_w, g_b, g_wcls, g_bcls
        g = g_y if g_y.dtype == torch.float32 else g_y.float()
        if g_logits is not None:
            g = torch.addmm(g, g_logits, wc)
        gb = cast_rows(g)
        if ctx.needs_input_grad[0]:
            g_x = gemm_nt(gb, w_t)
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            if ctx.has_bias:
                g_w, g_b = gemm_tn(gb, x, want_colsum=True)
            else:
                g_w = gemm_tn(gb, x)
        return g_x, g_w, g_b, g_wcls, g_bcls


class _LinearMeanClassifierBf16Fn(torch.autograd.Function):
    """ops._LinearMeanClassifierFn on bf16 rows: W_comb is rounded to bf16 inside spgnn_linear_mean_fold_fwd (the image the bf16
    product reads), P / c0 / every gradient use the rounded values - the function as evaluated, as _LinearClassifierBf16Fn."""

    @staticmethod
    def forward(ctx, zx, w_fc, w_res, bias, w_cls, b_cls, H, D):
        from . import ops
        ctx.set_materialize_grads(False)
        Kc = zx.shape[1]
        w_comb, _, w_bf, b_mean, P, c0 = ops.linear_mean_fold_buffers(w_fc, w_res, bias, w_cls, b_cls, H, D, bf16=True)
        y = gemm_nt(zx, w_bf[:, :Kc], out_f32=True, bias=b_mean)
        head, ctx.head = ops.LOSS_HEAD, None
        if head is not None and not head.used and zx.shape[0] == head.labels.shape[0] and ops.classifier_ce_supported(zx, P[:, :Kc]):
            # the step's loss joins this node (spgnn_classifier_ce_bf16 on the folded classifier): one pass over the bf16 rows
            logits, head.g_logits, wpart, colsum = ops.classifier_ce(zx, P[:, :Kc], c0, head)
            head.used = True
            ctx.head = (wpart, colsum)
        else:
            logits = scores_fwd(zx, P[:, :Kc], bias=c0)
        ctx.cfg = (H, D, w_fc.shape[1], Kc, w_res is not None, bias is not None, b_cls is not None)
        ctx.save_for_backward(zx, w_comb, w_bf, P, w_cls, b_mean)
        return y, logits

    @staticmethod
    def backward(ctx, g_y, g_logits):
        from . import ops
        if g_y is None and g_logits is None:
            return (None,) * 8
        zx, w_comb, w_bf, P, w_cls, b_mean = ctx.saved_tensors
        H, D, F_, Kc, has_res, has_bias, has_bcls = ctx.cfg
        g_zx = g_fc = g_res = g_bias = g_wcls = g_bcls = None
        cs = M1 = None
        if g_logits is not None:
            head, ctx.head = ctx.head, None
            g_logits = _rowmajor(g_logits)
            if head is not None:                          # the loss kernel left the column sums and the partials of g_logits^T Zx
                cs = head[1]
                M1 = ops.sum_partials(head[0])[:, :Kc]
            else:
                cs = _ops.column_sums(g_logits)
                M1 = scores_bwd_w(g_logits, zx)
            g_bcls = cs if has_bcls else None
        if g_y is None:                                   # folded route: g_Zx as bf16 rows straight from the fp32 logit gradient
            if ctx.needs_input_grad[0]:
                g_zx = scores_bwd_x(g_logits, P[:, :Kc], Kc)
            g_fc, g_res, g_bias, g_wcls = ops.linear_mean_fold_grads(M1, cs, w_cls, w_comb, b_mean, H, D, F_, has_res, has_bias)
            return g_zx, g_fc, g_res, g_bias, g_wcls, g_bcls, None, None
        g = g_y if g_y.dtype == torch.float32 else g_y.float()
        if g_logits is not None:
            g = torch.addmm(g, g_logits, w_cls.detach())
            g_wcls = torch.mm(M1, w_comb[:, :Kc].t())
            if has_bias:
                g_wcls.addr_(cs, b_mean)
        gb = cast_rows(g)
        if ctx.needs_input_grad[0]:
            w_t = torch.zeros((Kc, _pad8(D)), dtype=BF16, device=zx.device)
            w_t[:, :D] = w_bf[:, :Kc].t()
            g_zx = gemm_nt(gb, w_t[:, :D])
        if has_bias:
            g_wc_, g_bm = gemm_tn(gb, zx, want_colsum=True)
        else:
            g_wc_, g_bm = gemm_tn(gb, zx), None
        g_wc_ = g_wc_ * (1.0 / H)
        g_fc = g_wc_[:, :H * F_].reshape(D, H, F_).permute(1, 0, 2).reshape(H * D, F_)
        g_res = g_wc_[:, H * F_:].unsqueeze(0).expand(H, D, F_).reshape(H * D, F_) if has_res else None
        g_bias = (g_bm * (1.0 / H)).repeat(H) if has_bias else None
        return g_zx, g_fc, g_res, g_bias, g_wcls, g_bcls, None, None


def gat_layer_linear_mean(csc: DeviceCSC, x, w_fc, w_res, w_lr, bias, H: int, D: int, slope: float, p_drop: float = 0.0,
                          seed: int = 0, w_cls=None, b_cls=None):
    """ops.gat_layer_linear_mean on bf16 rows -> (mean (N, D) fp32, attn (E, H) fp32[, logits (N, J) fp32]).  The combined
    weight is assembled from the fp32 parameters by a few tiny torch ops (autograd hands its fp32 gradient back to
    w_fc / w_res / bias).  ``w_cls`` (J <= 32, D), ``b_cls``: the classifier joined to the product's node."""
    _require_cuda(x, w_fc, w_res, w_lr, bias, w_cls, b_cls)
    F_ = x.shape[1]
    zx, attn = _GATAggregateBf16Fn.apply(as_rows(x), w_lr, csc, H, slope, p_drop, seed)
    from . import ops as _ops
    zx = _ops.take_loss_rows(zx, w_cls is not None)    # a loss-rows step: the product, the mean and the classifier on the kept rows
    if (_ops.FUSE_LINEAR_MEAN_FOLD and w_cls is not None and w_cls.shape[0] <= 32 and w_cls.shape[1] == D and D % 8 == 0
            and ((H + 1) * F_) % 8 == 0):
        out, logits = _LinearMeanClassifierBf16Fn.apply(zx, w_fc, w_res, bias, w_cls, b_cls, H, D)
        return out, attn, logits
    parts = [w_fc.view(H, D, F_).permute(1, 0, 2).reshape(D, H * F_)]
    parts.append(w_res.view(H, D, F_).sum(0) if w_res is not None else w_fc.new_zeros((D, F_)))
    w_comb = torch.cat(parts, dim=1) * (1.0 / H)
    b_mean = bias.view(H, D).mean(0) if bias is not None else None
    if w_cls is None:
        return _LinearBf16Fn.apply(zx, w_comb, b_mean), attn
    if w_cls.shape[0] <= 32 and w_cls.shape[1] == D and D % 4 == 0:
        out, logits = _LinearClassifierBf16Fn.apply(zx, w_comb, b_mean, w_cls, b_cls)
    else:
        out = _LinearBf16Fn.apply(zx, w_comb, b_mean)
        logits = torch.nn.functional.linear(out, w_cls, b_cls)
    return out, attn, logits


class _CatDropoutBf16(torch.autograd.Function):
    """dropout(cat(tensors, dim=1), p) on bf16 rows in one pass per source; the keep mask is the counter hash of
    spgnn_cat_dropout, regenerated by the backward pass."""

    @staticmethod
    def forward(ctx, p, seed, *tensors):
        widths = [t.shape[1] for t in tensors]
        F_ = sum(widths)
        N = tensors[0].shape[0]
        buf = empty_rows(N, F_, tensors[0].device)
        lib = _capi.load()
        off = 0
        with torch.cuda.device(buf.device):
            for t in tensors:
                if t.stride(1) != 1 or t.stride(0) % 4 or t.data_ptr() % 8:
                    t = t.contiguous()
                _capi.check(lib.spgnn_cat_dropout_bf16(t.data_ptr(), t.stride(0), buf.data_ptr(), buf.stride(0), N, t.shape[1], off,
                                                       F_, p, seed, _seed_off_ptr(buf.device), 0, _stream(buf)),
                            "spgnn_cat_dropout_bf16")
                off += t.shape[1]
        ctx.widths, ctx.p, ctx.seed = widths, p, seed
        return buf

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return (None, None) + (None,) * len(ctx.widths)
        if g.stride(1) != 1 or g.stride(0) % 4 or g.data_ptr() % 8:
            g = g.contiguous()
        N, F_ = g.shape
        lib = _capi.load()
        outs, off = [], 0
        with torch.cuda.device(g.device):
            for w, need in zip(ctx.widths, ctx.needs_input_grad[2:]):
                if need and ctx.p == 0.0 and off % 8 == 0 and g.stride(0) % 8 == 0 and g.data_ptr() % 16 == 0:
                    outs.append(g[:, off:off + w])      # a plain concatenation: column blocks of its gradient, no copy
                elif need:
                    go = empty_rows(N, w, g.device)
                    _capi.check(lib.spgnn_cat_dropout_bf16(g.data_ptr(), g.stride(0), go.data_ptr(), go.stride(0), N, w, off, F_,
                                                           ctx.p, ctx.seed, _seed_off_ptr(g.device), 1, _stream(g)),
                                "spgnn_cat_dropout_bf16")
                    outs.append(go)
                else:
                    outs.append(None)
                off += w
        return (None, None) + tuple(outs)


def cat_dropout(tensors, p: float = 0.0, seed: int = 0) -> torch.Tensor:
    """dropout(cat(tensors, 1), p) as GEMM-readable bf16 rows; every width must be a multiple of 4."""
    _require_cuda(*tensors)
    assert all(t.dtype == BF16 and t.shape[1] % 4 == 0 for t in tensors)
    return _CatDropoutBf16.apply(float(p), int(seed), *tensors)
