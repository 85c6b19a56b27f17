"""ORACLE (test infrastructure, not product code) — CPU restatement of the DGL layer
semantics the reference's graph-convolution hot path relies on.

PARITY UNPINNED: the arithmetic of this path lives in DGL (dmlc/dgl), an un-vendored,
un-pinned dependency of the reference (``git clone`` of master at image build,
reference docker_base/Dockerfile:130-137; README floor "0.6.x", README.md:35; API usage
brackets it to 0.6-0.8). DGL is absent from /root/reference and not installable here, and the
reference ships no tests or golden vectors for this path. What follows
restates DGL's published layer definitions as used at the reference call sites
(models.py:8,172-182,301-314,358-383,425-456,506-521,668-679); it is pinned only by
(a) the independent dense formulation in ``oracle/dense.py`` and (b) hand-derived
known-answer cases (tests/test_oracle.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Every function is plain PyTorch on whatever dtype/device its inputs have (fp64 for
cross-checks, fp32 for the parity target), differentiable by autograd, and written on the
edge list (``index_select`` / ``index_add_`` / ``scatter_reduce``), mirroring how DGL lowers
the layers to gsddmm / edge_softmax / gspmm.
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------------------
# primitives (DGL: dgl.ops.edge_softmax, gspmm copy_u/u_mul_e with sum / mean / max)
# --------------------------------------------------------------------------------------
def edge_softmax(dst: Tensor, e: Tensor, num_nodes: int) -> Tensor:
    """softmax of edge scores over the edges that share a destination.
    DGL edge_softmax: e - max_dst -> exp -> / sum_dst."""
    shape = (num_nodes,) + tuple(e.shape[1:])
    idx = dst.view(-1, *([1] * (e.dim() - 1))).expand_as(e)
    emax = torch.full(shape, -float("inf"), dtype=e.dtype, device=e.device)
    emax = emax.scatter_reduce(0, idx, e, reduce="amax", include_self=True)
    ex = torch.exp(e - emax.index_select(0, dst))
    esum = torch.zeros(shape, dtype=e.dtype, device=e.device).index_add_(0, dst, ex)
    return ex / esum.index_select(0, dst)


def spmm_sum(src: Tensor, dst: Tensor, x: Tensor, num_nodes: int, w: Optional[Tensor] = None) -> Tensor:
    m = x.index_select(0, src)
    if w is not None:
        m = m * w
    return torch.zeros((num_nodes,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device).index_add_(0, dst, m)


def spmm_max(src: Tensor, dst: Tensor, x: Tensor, num_nodes: int) -> Tensor:
    m = x.index_select(0, src)
    idx = dst.view(-1, *([1] * (x.dim() - 1))).expand_as(m)
    out = torch.full((num_nodes,) + tuple(x.shape[1:]), -float("inf"), dtype=x.dtype, device=x.device)
    out = out.scatter_reduce(0, idx, m, reduce="amax", include_self=True)
    return torch.where(torch.isinf(out), torch.zeros_like(out), out)   # DGL: zero for isolated nodes


def in_degrees(dst: Tensor, num_nodes: int, dtype) -> Tensor:
    return torch.zeros(num_nodes, dtype=dtype, device=dst.device).index_add_(
        0, dst, torch.ones(dst.shape[0], dtype=dtype, device=dst.device))


# --------------------------------------------------------------------------------------
# reduced-precision STORAGE emulation (the build's bf16-storage extension, BASELINE config 4; the reference is fp32
# only).  The product keeps node-feature rows, projected rows and their gradients as bfloat16 in HBM and accumulates in
# fp32; here that is modelled by rounding a tensor to bf16 where the product stores it:
#   store(x)      value rounded in the forward pass AND its gradient rounded in the backward pass (a stored activation
#                 and its stored gradient);
#   store_fwd(x)  value rounded, gradient passed through (fp32 parameters fed to a bf16 GEMM: their gradient stays fp32);
#   round_grad(x) identity whose gradient is rounded (the pre-activation gradient g_pre, which the product stores before
#                 any other kernel reads it).
# Arithmetic between those points runs in the tensors' own dtype (fp32 or fp64).
# --------------------------------------------------------------------------------------
def _rb(x: Tensor) -> Tensor:
    return x.to(torch.bfloat16).to(x.dtype)


class _Store(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return _rb(x)

    @staticmethod
    def backward(ctx, g):
        return _rb(g)


class _StoreFwd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return _rb(x)

    @staticmethod
    def backward(ctx, g):
        return g


class _RoundGrad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return _rb(g)


def _act_kind(fn) -> str:
    if fn is None:
        return "none"
    if fn is F.elu:
        return "elu"
    if fn in (torch.tanh, F.tanh):
        return "tanh"
    if fn in (F.relu, torch.relu):
        return "relu"
    raise ValueError("storage emulation knows elu / tanh / relu / None activations")


class _ActStore(torch.autograd.Function):
    """y = act(x) with the product's storage behaviour: the activated rows are stored rounded (``store_out``; the output
    layer's rows feed an fp32 head mean unrounded, but the copy kept for the backward pass is rounded all the same), the
    derivative is formed from that STORED output (ELU: y > 0 ? 1 : y + 1; tanh: 1 - y^2 - all the backward kernel has),
    the incoming gradient of a stored tensor is itself stored, and the pre-activation gradient is stored before anything
    else reads it."""

    @staticmethod
    def forward(ctx, x, kind, store_out):
        y = {"none": lambda t: t, "elu": F.elu, "tanh": torch.tanh, "relu": F.relu}[kind](x)
        yr = _rb(y)
        ctx.save_for_backward(yr)
        ctx.kind, ctx.store_out = kind, store_out
        return yr if store_out else y

    @staticmethod
    def backward(ctx, g):
        (yr,) = ctx.saved_tensors
        if ctx.store_out:
            g = _rb(g)
        if ctx.kind == "elu":
            g = g * torch.where(yr > 0, torch.ones_like(yr), yr + 1)
        elif ctx.kind == "tanh":
            g = g * (1 - yr * yr)
        elif ctx.kind == "relu":
            g = g * (yr > 0).to(g.dtype)
        return _rb(g), None, None


class Bf16Storage:
    """Hooks for ``gat_conv(..., storage=Bf16Storage)``."""
    store = staticmethod(_Store.apply)
    store_fwd = staticmethod(_StoreFwd.apply)
    round_grad = staticmethod(_RoundGrad.apply)

    @staticmethod
    def activate(x, fn, store_out: bool):
        return _ActStore.apply(x, _act_kind(fn), store_out)


# --------------------------------------------------------------------------------------
# layers
# --------------------------------------------------------------------------------------
def gat_conv(src: Tensor, dst: Tensor, num_nodes: int, feat: Tensor, fc_weight: Tensor, attn_l: Tensor,
             attn_r: Tensor, res_fc_weight: Optional[Tensor] = None, bias: Optional[Tensor] = None,
             negative_slope: float = 0.2, activation: Optional[Callable] = None,
             attn_keep: Optional[Tensor] = None, residual_identity: bool = False, storage=None,
             store_out: bool = True) -> Tuple[Tensor, Tensor]:
    """dgl.nn.pytorch.GATConv.forward (Appendix A.1). Returns (rst (N,H,D), a (E,H)).

    ``attn_keep`` (E,H), if given, is the already scaled dropout multiplier applied to the
    attention (mask / (1-p)), standing in for ``attn_drop``.
    ``storage`` (e.g. ``Bf16Storage``): emulate reduced-precision storage of ft / res / out and of their gradients;
    ``store_out=False`` leaves the layer output unrounded (the output layer, whose head mean is kept in fp32).
    """
    _, H, D = attn_l.shape
    if storage is not None:
        fc_weight = storage.store_fwd(fc_weight)
        res_fc_weight = storage.store_fwd(res_fc_weight) if res_fc_weight is not None else None
    ft = F.linear(feat, fc_weight)
    if storage is not None:
        ft = storage.store(ft)
    ft = ft.view(num_nodes, H, D)
    el = (ft * attn_l).sum(-1)                                       # (N,H)
    er = (ft * attn_r).sum(-1)
    e = F.leaky_relu(el.index_select(0, src) + er.index_select(0, dst), negative_slope)   # u_add_v
    a = edge_softmax(dst, e, num_nodes)                               # (E,H)
    a_used = a if attn_keep is None else a * attn_keep
    rst = spmm_sum(src, dst, ft, num_nodes, a_used.unsqueeze(-1))     # u_mul_e / sum
    if res_fc_weight is not None:
        res = F.linear(feat, res_fc_weight)
        if storage is not None:
            res = storage.store(res)
        rst = rst + res.view(num_nodes, H, D)
    elif residual_identity:
        rst = rst + feat.view(num_nodes, -1, D)
    if bias is not None:
        rst = rst + bias.view(1, H, D)
    if storage is not None:
        rst = storage.activate(rst, activation, store_out)
    elif activation is not None:
        rst = activation(rst)
    return rst, a


def linear_mean_form(H: int, D: int, f_in: int, has_res_weight: bool) -> bool:
    """Where the build's bf16 path evaluates an output GATConv (no activation, heads averaged) in the linear-mean form
    below instead of projecting first: input narrower than a head, [z | x] no wider than the per-head rows, vector-aligned
    widths, a learned (or no) residual.  Only the STORAGE POINTS differ between the two forms; in exact arithmetic they
    are the same function (tests/test_oracle.py checks that)."""
    return H in (1, 2, 4) and f_in < D and (H + 1) * f_in <= H * D and f_in % 8 == 0 and f_in <= 1024


def gat_conv_linear_mean(src: Tensor, dst: Tensor, num_nodes: int, feat: Tensor, fc_weight: Tensor, attn_l: Tensor,
                         attn_r: Tensor, res_fc_weight: Optional[Tensor] = None, bias: Optional[Tensor] = None,
                         negative_slope: float = 0.2, attn_keep: Optional[Tensor] = None, storage=None,
                         round_out_grad: bool = True) -> Tuple[Tensor, Tensor]:
    """``gat_conv(..., activation=None)[0].mean(1)`` restated as ONE product (reference models.py:320-327,
    ``self.gat_layers[-1](g, h).mean(1)`` with DGL's GATConv, Appendix A.1):
        el = x (W_h^T attn_l_h),  z_h[v] = sum_u a_h(u, v) x[u],
        mean_h(W_h z_h + Wres_h x + b_h) = [z_0 | .. | z_{H-1} | x] [W_0 | .. | W_{H-1} | sum_h Wres_h]^T / H + mean_h b_h.
    Returns (mean (N, D), a (E, H)).  ``storage``: z and the combined weight are what the bf16 path stores rounded (the
    scores come from the fp32 parameters, the product is kept in fp32); the gradients it stores are those of the
    product, of [z | x] and of x.  ``round_out_grad=False``: the product's gradient is never stored - the build joins the
    *Net's classifier to this node and, when only the logits carry a gradient, goes from the fp32 logit gradient to the
    gradient of [z | x] directly (ops_bf16._LinearClassifierBf16Fn)."""
    _, H, D = attn_l.shape
    n, f_in = feat.shape
    W = fc_weight.view(H, D, f_in)
    w_l = (W * attn_l.view(H, D, 1)).sum(1)                          # (H, F)
    w_r = (W * attn_r.view(H, D, 1)).sum(1)
    xa = feat if storage is None else storage.round_grad(feat)       # the aggregate kernels store their g_x
    el, er = xa @ w_l.t(), xa @ w_r.t()
    e = F.leaky_relu(el.index_select(0, src) + er.index_select(0, dst), negative_slope)
    a = edge_softmax(dst, e, num_nodes)
    a_used = a if attn_keep is None else a * attn_keep
    z = spmm_sum(src, dst, xa.unsqueeze(1).expand(n, H, f_in), num_nodes, a_used.unsqueeze(-1))   # (N, H, F)
    if storage is not None:
        z = storage.store_fwd(z)
    zx = torch.cat([z.flatten(1), feat], dim=1)
    if storage is not None:
        zx = storage.round_grad(zx)
    w_res = res_fc_weight.view(H, D, f_in).sum(0) if res_fc_weight is not None else fc_weight.new_zeros((D, f_in))
    w_comb = torch.cat([W.permute(1, 0, 2).reshape(D, H * f_in), w_res], dim=1) / H
    if storage is not None:
        w_comb = storage.store_fwd(w_comb)
    out = zx @ w_comb.t()
    if bias is not None:
        out = out + bias.view(H, D).mean(0)
    if storage is not None and round_out_grad:
        out = storage.round_grad(out)                                # the incoming gradient is a bf16 GEMM operand
    return out, a


def graph_conv(src: Tensor, dst: Tensor, num_nodes: int, feat: Tensor, weight: Tensor, bias: Optional[Tensor],
               activation: Optional[Callable] = None) -> Tensor:
    """dgl.nn.pytorch.GraphConv.forward with norm='both' (Appendix A.2)."""
    out_deg = in_degrees(src, num_nodes, feat.dtype).clamp(min=1)
    in_deg = in_degrees(dst, num_nodes, feat.dtype).clamp(min=1)
    h = feat * out_deg.pow(-0.5).unsqueeze(-1)
    f_in, f_out = weight.shape
    if f_in > f_out:
        rst = spmm_sum(src, dst, h @ weight, num_nodes)
    else:
        rst = spmm_sum(src, dst, h, num_nodes) @ weight
    rst = rst * in_deg.pow(-0.5).unsqueeze(-1)
    if bias is not None:
        rst = rst + bias
    if activation is not None:
        rst = activation(rst)
    return rst


def gin_conv(src: Tensor, dst: Tensor, num_nodes: int, feat: Tensor, eps: Tensor, apply_func: Callable,
             aggregator_type: str = "mean") -> Tensor:
    """dgl.nn.pytorch.GINConv.forward (Appendix A.3)."""
    if aggregator_type == "sum":
        neigh = spmm_sum(src, dst, feat, num_nodes)
    elif aggregator_type == "mean":
        deg = in_degrees(dst, num_nodes, feat.dtype).clamp(min=1)
        neigh = spmm_sum(src, dst, feat, num_nodes) / deg.unsqueeze(-1)
    elif aggregator_type == "max":
        neigh = spmm_max(src, dst, feat, num_nodes)
    else:
        raise KeyError(aggregator_type)
    rst = (1 + eps) * feat + neigh
    return apply_func(rst) if apply_func is not None else rst


def sage_conv_pool(src: Tensor, dst: Tensor, num_nodes: int, feat: Tensor, fc_pool_w: Tensor, fc_pool_b: Tensor,
                   fc_self_w: Tensor, fc_self_b: Optional[Tensor], fc_neigh_w: Tensor, fc_neigh_b: Optional[Tensor],
                   bias: Optional[Tensor] = None, activation: Optional[Callable] = None) -> Tensor:
    """dgl.nn.pytorch.SAGEConv.forward, aggregator 'pool' (Appendix A.4)."""
    m = F.relu(F.linear(feat, fc_pool_w, fc_pool_b))
    neigh = spmm_max(src, dst, m, num_nodes)
    rst = F.linear(feat, fc_self_w, fc_self_b) + F.linear(neigh, fc_neigh_w, fc_neigh_b)
    if bias is not None:
        rst = rst + bias
    if activation is not None:
        rst = activation(rst)
    return rst


# --------------------------------------------------------------------------------------
# model stacks (reference models.py:160-540, 650-696) driven by a state_dict
# --------------------------------------------------------------------------------------
def _gat_layer(sd: Dict[str, Tensor], prefix: str, src, dst, n, h, slope, act, storage=None, store_out=True,
               residual_identity=False, feat_keep=None, attn_keep=None):
    """``residual_identity``: the layer is residual with in_feats == out_feats (DGL then uses an Identity ``res_fc``, which
    has no state_dict entry, so it cannot be inferred from ``sd``).
    ``feat_keep`` (N, F_in) / ``attn_keep`` (E, H): already scaled dropout multipliers (mask / (1 - p)) standing in for the
    layer's ``feat_drop`` / ``attn_drop`` in training mode: DGL's GATConv applies ``feat_drop`` to the layer INPUT - ``fc`` and
    ``res_fc`` both read the dropped rows - and ``attn_drop`` to the softmax weights (Appendix A.1)."""
    w_res = sd.get(prefix + "res_fc.weight")
    if feat_keep is not None:
        h = h * feat_keep
    return gat_conv(src, dst, n, h, sd[prefix + "fc.weight"], sd[prefix + "attn_l"], sd[prefix + "attn_r"],
                    w_res, sd.get(prefix + "bias"), slope, act, storage=storage, store_out=store_out,
                    residual_identity=residual_identity and w_res is None, attn_keep=attn_keep)[0]


def _count_layers(sd: Dict[str, Tensor], prefix: str) -> int:
    idx = {int(k[len(prefix):].split(".")[0]) for k in sd if k.startswith(prefix)}
    return max(idx) + 1 if idx else 0


def _is_identity_res(sd, prefix: str, residual: bool) -> bool:
    """A residual GATConv whose in_feats equals its per-head out_feats has DGL's Identity ``res_fc`` (no weight)."""
    if not residual or (prefix + "res_fc.weight") in sd:
        return False
    H_D, f_in = sd[prefix + "fc.weight"].shape
    return f_in == sd[prefix + "attn_l"].shape[-1]


def gat_stack(sd, src, dst, n, fvs, prefix="gat_layers.", negative_slope=0.2, activation=F.elu, norm=False, storage=None,
              residual=True, classifier_joined=False):
    """reference models.py:321-329 (GAT.forward).  ``residual``: the stack was built with ``residual=True`` (every
    shipped GAT config): layers without a ``res_fc.weight`` whose widths match then carry DGL's identity residual."""
    L = _count_layers(sd, prefix)
    h = fvs if storage is None else storage.store_fwd(fvs)
    for l in range(L - 1):
        p = f"{prefix}{l}."
        h = _gat_layer(sd, p, src, dst, n, h, negative_slope, activation, storage,
                       residual_identity=_is_identity_res(sd, p, residual)).flatten(1)
    p = f"{prefix}{L - 1}."
    _, H, D = sd[p + "attn_l"].shape
    ident = _is_identity_res(sd, p, residual)
    if storage is not None and not ident and linear_mean_form(H, D, h.shape[1], (p + "res_fc.weight") in sd):
        out = gat_conv_linear_mean(src, dst, n, h, sd[p + "fc.weight"], sd[p + "attn_l"], sd[p + "attn_r"],
                                   sd.get(p + "res_fc.weight"), sd.get(p + "bias"), negative_slope, storage=storage,
                                   round_out_grad=not classifier_joined)[0]
    else:
        out = _gat_layer(sd, p, src, dst, n, h, negative_slope, None, storage, store_out=False, residual_identity=ident).mean(1)
    return F.normalize(out, p=2, dim=1) if norm else out


def spgnn_pel_stack(sd, src, dst, n, fvs, pos_enc, negative_slope=0.2, activation=F.elu, p_activation=torch.tanh, drop=None):
    """reference models.py:472-484 (GATPSPGNN.forward).
    ``drop``: training mode with GIVEN masks - {("feat" | "attn", "s" | "p", layer): scaled multiplier} for the structure
    ("s": gat_layers) and position ("p": pgnn_layers) GATConv's ``feat_drop`` (N, F_in) and ``attn_drop`` (E, H); a missing
    key = that dropout is off (the reference builds layer 0 of both streams, the last position layer and the output layer
    with rate 0: models.py:431-434, 449-456).  None: eval mode."""
    L = _count_layers(sd, "pgnn_layers.")
    d = drop or {}
    h_s, h_p = fvs, pos_enc
    for l in range(L):
        h_s = torch.cat([h_s, h_p], dim=1)
        h_s = _gat_layer(sd, f"gat_layers.{l}.", src, dst, n, h_s, negative_slope, activation,
                         feat_keep=d.get(("feat", "s", l)), attn_keep=d.get(("attn", "s", l))).flatten(1)
        h_p = _gat_layer(sd, f"pgnn_layers.{l}.", src, dst, n, h_p, negative_slope, p_activation,
                         feat_keep=d.get(("feat", "p", l)), attn_keep=d.get(("attn", "p", l))).flatten(1)
    h_s = torch.cat([h_s, h_p], dim=1)
    h_s = _gat_layer(sd, f"gat_layers.{L}.", src, dst, n, h_s, negative_slope, activation,
                     feat_keep=d.get(("feat", "s", L)), attn_keep=d.get(("attn", "s", L))).mean(1)
    return h_s, h_p


def spgnn_penl_stack(sd, src, dst, n, fvs, pos_enc, negative_slope=0.2, activation=F.elu):
    """reference models.py:529-540 (GATPSPGNNNL.forward)."""
    L = _count_layers(sd, "gat_layers.")
    h_s, h_p = fvs, pos_enc
    for l in range(L - 1):
        h_s = torch.cat([h_s, h_p], dim=1)
        h_s = _gat_layer(sd, f"gat_layers.{l}.", src, dst, n, h_s, negative_slope, activation).flatten(1)
    h_s = torch.cat([h_s, h_p], dim=1)
    h_s = _gat_layer(sd, f"gat_layers.{L - 1}.", src, dst, n, h_s, negative_slope, activation).mean(1)
    return h_s, h_p


def gcn_stack(sd, src, dst, n, fvs, prefix="gcn_layers.", activation=F.elu):
    """reference models.py:188-194 (GCN.forward)."""
    L = _count_layers(sd, prefix)
    h = fvs
    for l in range(L):
        h = graph_conv(src, dst, n, h, sd[f"{prefix}{l}.weight"], sd.get(f"{prefix}{l}.bias"),
                       activation if l < L - 1 else None)
    return h


def gin_stack(sd, src, dst, n, fvs, prefix="gin_layers.", norm=False):
    """reference models.py:386-392 (GIN.forward); MLP = Linear, Dropout(0.1) [identity in eval],
    LeakyReLU, Linear, LeakyReLU (models.py:358-383)."""
    L = _count_layers(sd, prefix)
    h = fvs
    for l in range(L):
        p = f"{prefix}{l}."

        def mlp(x, p=p):
            x = F.leaky_relu(F.linear(x, sd[p + "apply_func.0.weight"], sd[p + "apply_func.0.bias"]))
            return F.leaky_relu(F.linear(x, sd[p + "apply_func.3.weight"], sd[p + "apply_func.3.bias"]))
        h = gin_conv(src, dst, n, h, sd[p + "eps"], mlp, "mean")
    return F.normalize(h, p=2, dim=1) if norm else h


def sage_stack(sd, src, dst, n, fvs, prefix="g_layers.", activation=F.elu):
    """reference models.py:691-696 (SAGE.forward); hidden layers use ``activation``, output none."""
    L = _count_layers(sd, prefix)
    h = fvs
    for l in range(L):
        p = f"{prefix}{l}."
        h = sage_conv_pool(src, dst, n, h, sd[p + "fc_pool.weight"], sd[p + "fc_pool.bias"],
                           sd[p + "fc_self.weight"], sd.get(p + "fc_self.bias"),
                           sd[p + "fc_neigh.weight"], sd.get(p + "fc_neigh.bias"), sd.get(p + "bias"),
                           activation if l < L - 1 else None)
    return h


# --------------------------------------------------------------------------------------
# the same layers on a message-flow graph ("block") of neighbour-sampled training
# (reference forward_batch: models.py:331-340, 394-400, 685-689; job_runner.py:1499-1503)
#
# DGL: on a block ``expand_as_pair(feat, g)`` gives feat_src = feat (num_src rows) and
# feat_dst = feat[:g.number_of_dst_nodes()] — the dst nodes are the first src nodes — messages
# flow src -> dst and every reduction yields num_dst rows.  Written bipartite here, directly:
# edge endpoints ``src`` in [0,num_src), ``dst`` in [0,num_dst).
# --------------------------------------------------------------------------------------
def gat_conv_block(src, dst, num_src: int, num_dst: int, feat, fc_weight, attn_l, attn_r, res_fc_weight=None, bias=None,
                   negative_slope: float = 0.2, activation=None, residual_identity: bool = False):
    """GATConv.forward on a block: fc on src and dst rows, el from src, er from dst, residual from feat_dst."""
    _, H, D = attn_l.shape
    ft_src = F.linear(feat, fc_weight).view(num_src, H, D)
    ft_dst = ft_src[:num_dst]
    el = (ft_src * attn_l).sum(-1)
    er = (ft_dst * attn_r).sum(-1)
    e = F.leaky_relu(el.index_select(0, src) + er.index_select(0, dst), negative_slope)
    a = edge_softmax(dst, e, num_dst)
    rst = spmm_sum(src, dst, ft_src, num_dst, a.unsqueeze(-1))
    h_dst = feat[:num_dst]
    if res_fc_weight is not None:
        rst = rst + F.linear(h_dst, res_fc_weight).view(num_dst, H, D)
    elif residual_identity:
        rst = rst + h_dst.view(num_dst, -1, D)
    if bias is not None:
        rst = rst + bias.view(1, H, D)
    if activation is not None:
        rst = activation(rst)
    return rst, a


def gin_conv_block(src, dst, num_src: int, num_dst: int, feat, eps, apply_func, aggregator_type: str = "mean"):
    if aggregator_type == "sum":
        neigh = spmm_sum(src, dst, feat, num_dst)
    elif aggregator_type == "mean":
        neigh = spmm_sum(src, dst, feat, num_dst) / in_degrees(dst, num_dst, feat.dtype).clamp(min=1).unsqueeze(-1)
    elif aggregator_type == "max":
        neigh = spmm_max(src, dst, feat, num_dst)
    else:
        raise KeyError(aggregator_type)
    rst = (1 + eps) * feat[:num_dst] + neigh
    return apply_func(rst) if apply_func is not None else rst


def graph_conv_block(src, dst, num_src: int, num_dst: int, feat, weight, bias, activation=None):
    out_deg = in_degrees(src, num_src, feat.dtype).clamp(min=1)
    in_deg = in_degrees(dst, num_dst, feat.dtype).clamp(min=1)
    h = feat * out_deg.pow(-0.5).unsqueeze(-1)
    f_in, f_out = weight.shape
    rst = spmm_sum(src, dst, h @ weight, num_dst) if f_in > f_out else spmm_sum(src, dst, h, num_dst) @ weight
    rst = rst * in_deg.pow(-0.5).unsqueeze(-1)
    if bias is not None:
        rst = rst + bias
    return activation(rst) if activation is not None else rst


def sage_conv_pool_block(src, dst, num_src: int, num_dst: int, feat, fc_pool_w, fc_pool_b, fc_self_w, fc_self_b,
                         fc_neigh_w, fc_neigh_b, bias=None, activation=None):
    """SAGEConv 'pool' on a block: max over the sampled in-neighbours of relu(fc_pool(feat_src)); h_self = feat_dst."""
    neigh = spmm_max(src, dst, F.relu(F.linear(feat, fc_pool_w, fc_pool_b)), num_dst)
    rst = F.linear(feat[:num_dst], fc_self_w, fc_self_b) + F.linear(neigh, fc_neigh_w, fc_neigh_b)
    if bias is not None:
        rst = rst + bias
    return activation(rst) if activation is not None else rst


def stack_blocks(kind: str, sd: Dict[str, Tensor], blocks, x, negative_slope=0.2, activation=F.elu, norm=False):
    """``forward_batch(blocks, x)`` of SAGE / GAT / GIN (reference models.py:685-689, 331-340, 394-400): layer ``l``
    runs on ``blocks[l]`` = (src, dst, num_src, num_dst), each block's src rows being the previous block's dst rows."""
    prefix = {"sage": "g_layers.", "gat": "gat_layers.", "gin": "gin_layers."}[kind]
    L = _count_layers(sd, prefix)
    if len(blocks) != L:
        raise ValueError(f"{L} layers but {len(blocks)} blocks")
    h = x
    for l, (src, dst, ns, nd) in enumerate(blocks):
        p = f"{prefix}{l}."
        if kind == "sage":
            h = sage_conv_pool_block(src, dst, ns, nd, h, sd[p + "fc_pool.weight"], sd[p + "fc_pool.bias"],
                                     sd[p + "fc_self.weight"], sd.get(p + "fc_self.bias"), sd[p + "fc_neigh.weight"],
                                     sd.get(p + "fc_neigh.bias"), sd.get(p + "bias"), activation if l < L - 1 else None)
        elif kind == "gat":
            r = gat_conv_block(src, dst, ns, nd, h, sd[p + "fc.weight"], sd[p + "attn_l"], sd[p + "attn_r"],
                               sd.get(p + "res_fc.weight"), sd.get(p + "bias"), negative_slope,
                               activation if l < L - 1 else None)[0]
            h = r.flatten(1) if l < L - 1 else r.mean(1)
        else:
            def mlp(t, p=p):
                t = F.leaky_relu(F.linear(t, sd[p + "apply_func.0.weight"], sd[p + "apply_func.0.bias"]))
                return F.leaky_relu(F.linear(t, sd[p + "apply_func.3.weight"], sd[p + "apply_func.3.bias"]))
            h = gin_conv_block(src, dst, ns, nd, h, sd[p + "eps"], mlp, "mean")
    if norm and kind in ("gat", "gin"):
        h = F.normalize(h, p=2, dim=1)
    return h


def net_forward_batch(kind: str, sd: Dict[str, Tensor], blocks, x, **kw):
    """``*Net.forward_batch(blocks, x)`` (reference models.py:814-817, 930-933): (gnn_out(embed), embed)."""
    pfx = kind + "."
    emb = stack_blocks(kind, {k[len(pfx):]: v for k, v in sd.items() if k.startswith(pfx)}, blocks, x, **kw)
    return F.linear(emb, sd["gnn_out.weight"], sd["gnn_out.bias"]), emb


def net_forward(kind: str, sd: Dict[str, Tensor], src, dst, n, fvs, pos_enc=None, **kw):
    """``*Net.forward(g)`` (reference models.py:277-280, 921-925, 1167-1170): head + gnn_out Linear.
    ``sd`` is the full-module state_dict (keys ``gat.*`` / ``gcn.*`` / ``gin.*`` / ``sage.*`` / ``gnn_out.*``)."""
    def sub(pfx):
        return {k[len(pfx):]: v for k, v in sd.items() if k.startswith(pfx)}
    if kind == "gat":
        # under a storage model: the *Net's classifier is joined to the output layer's node and the loss reads the logits only
        emb = (gat_stack(sub("gat."), src, dst, n, fvs, classifier_joined=kw.get("storage") is not None, **kw),)
    elif kind == "spgnn_pel":
        emb = spgnn_pel_stack(sub("gat."), src, dst, n, fvs, pos_enc, **kw)
    elif kind == "spgnn_penl":
        emb = spgnn_penl_stack(sub("gat."), src, dst, n, fvs, pos_enc, **kw)
    elif kind == "gcn":
        emb = (gcn_stack(sub("gcn."), src, dst, n, fvs, **kw),)
    elif kind == "gin":
        emb = (gin_stack(sub("gin."), src, dst, n, fvs, **kw),)
    elif kind == "sage":
        emb = (sage_stack(sub("sage."), src, dst, n, fvs, **kw),)
    else:
        raise KeyError(kind)
    out = F.linear(emb[0], sd["gnn_out.weight"], sd["gnn_out.bias"])
    return (out,) + tuple(emb)


def masked_weighted_ce(logits: Tensor, labels: Tensor, mask: Tensor, class_weight: Tensor) -> Tensor:
    """reference job_runner.py:1900: F.cross_entropy(out[mask], y[mask], weight=w)."""
    return F.cross_entropy(logits[mask], labels[mask], weight=class_weight)
