"""Drop-in graph-convolution layers with DGL's constructor signatures, forward signatures and
state_dict keys, running on libspgnn_hip.so.

These replace ``from dgl.nn.pytorch import GATConv, GraphConv, SAGEConv, GINConv``
(reference models.py:8).  ``graph`` is a :class:`spgnn_amd.graph.TreeGraph`.
"""
from __future__ import annotations

import math
from typing import Callable, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops, ops_bf16
from .graph import TreeGraph

__all__ = ["GATConv", "GraphConv", "GINConv", "SAGEConv", "Identity", "DGLError", "SkinnyLinear"]


class DGLError(Exception):
    """Same role as dgl.DGLError (raised for 0-in-degree nodes / bad options)."""


class SkinnyLinear(nn.Linear):
    """nn.Linear (same parameters / state_dict) for a classifier head with few outputs on ~1e5 rows, e.g. the
    reference's ``gnn_out = nn.Linear(node_embed_dim, out_ch)`` (models.py:1125): streaming HIP kernels on the GPU."""

    def forward(self, x):
        if x.is_cuda and x.dim() == 2 and self.out_features <= 32 and x.dtype == torch.float32:
            return ops.skinny_linear(x, self.weight, self.bias)
        return super().forward(x)


class Identity(nn.Module):
    def forward(self, x):
        return x


def _act_code(fn) -> Optional[int]:
    """Map an activation callable to a fused epilogue code, or None if it must run in Python."""
    if fn is None:
        return ops.ACT_NONE
    if fn in (F.elu,) or (isinstance(fn, nn.ELU) and fn.alpha == 1.0):
        return ops.ACT_ELU
    if fn in (torch.tanh, F.tanh) or isinstance(fn, nn.Tanh):
        return ops.ACT_TANH
    if fn in (F.relu, torch.relu) or isinstance(fn, nn.ReLU):
        return ops.ACT_RELU
    return None


def _act_code_dense(fn) -> Optional[int]:
    """As _act_code, plus LeakyReLU(0.01) - an epilogue of the dense products and of spmm_sum only (not of the GAT kernels)."""
    if isinstance(fn, nn.LeakyReLU) and fn.negative_slope == 0.01:
        return ops.ACT_LRELU
    return _act_code(fn)


def _draw_seed() -> int:
    # host-side draw from torch's CPU generator: reproducible under torch.manual_seed, no device sync
    return int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())


# Layer forms (fixed choices; the parity tests flip AGGREGATE_FIRST / SCORES_FROM_FT to check the forms against each other)
SCORES_FROM_FT = True      # el / er from ft in the projection GEMM's epilogue (DGL's own formulation) vs folded score weights
AGGREGATE_FIRST = True     # inputs narrower than one head's output: aggregate the input rows, then project
LINEAR_MEAN = True         # an output layer without activation whose heads are averaged: ONE product on [z_0 .. z_H-1 | x]
FUSE_CLASSIFIER = True     # the *Net's gnn_out joined to the output layer's autograd node


class GATConv(nn.Module):
    """dgl.nn.pytorch.GATConv (Appendix A.1). Parameters: ``fc.weight`` (H*D, F_in), ``attn_l``,
    ``attn_r`` (1,H,D), ``res_fc.weight`` (H*D, F_in) when residual and F_in != D, ``bias`` (H*D,).

    Reference call sites: models.py:301-314, 425-456, 506-521 (positional args
    ``in_feats, out_feats, num_heads, feat_drop, attn_drop, negative_slope, residual, activation``).
    """

    def __init__(self, in_feats, out_feats, num_heads, feat_drop=0., attn_drop=0., negative_slope=0.2,
                 residual=False, activation=None, allow_zero_in_degree=False, bias=True):
        super().__init__()
        if isinstance(in_feats, (tuple, list)):
            raise DGLError("bipartite (src, dst) feature sizes are not supported")
        self._num_heads, self._in_src_feats, self._in_dst_feats, self._out_feats = num_heads, in_feats, in_feats, out_feats
        self._allow_zero_in_degree = allow_zero_in_degree
        self.fc = nn.Linear(in_feats, out_feats * num_heads, bias=False)
        self.attn_l = nn.Parameter(torch.empty(1, num_heads, out_feats))
        self.attn_r = nn.Parameter(torch.empty(1, num_heads, out_feats))
        self.feat_drop = nn.Dropout(feat_drop)
        self.attn_drop = nn.Dropout(attn_drop)
        self.negative_slope = negative_slope
        self.leaky_relu = nn.LeakyReLU(negative_slope)
        if residual:
            if in_feats != out_feats:
                self.res_fc = nn.Linear(in_feats, num_heads * out_feats, bias=False)
            else:
                self.res_fc = Identity()
        else:
            self.register_buffer("res_fc", None)
        if bias:
            self.bias = nn.Parameter(torch.empty(num_heads * out_feats))
        else:
            self.register_buffer("bias", None)
        self.activation = activation
        self.reset_parameters()

    def reset_parameters(self):
        gain = nn.init.calculate_gain("relu")
        nn.init.xavier_normal_(self.fc.weight, gain=gain)
        nn.init.xavier_normal_(self.attn_l, gain=gain)
        nn.init.xavier_normal_(self.attn_r, gain=gain)
        if isinstance(self.res_fc, nn.Linear):
            nn.init.xavier_normal_(self.res_fc.weight, gain=gain)
        if self.bias is not None:
            nn.init.constant_(self.bias, 0)

    def set_allow_zero_in_degree(self, set_value):
        self._allow_zero_in_degree = set_value

    def can_fuse_out(self, feat: torch.Tensor) -> bool:
        """True when ``forward(..., fuse_out=...)`` is available for this layer and input: the project-first form with
        el / er from the GEMM epilogue (out_feats % 64 == 0, input at least as wide as one head's output), a fusable
        activation and a Linear (or no) residual, fp32 rows on a ROCm device."""
        H, D = self._num_heads, self._out_feats
        if feat.dtype == torch.bfloat16:          # bf16 rows: the project-first layer, single source (total == H*D)
            return (ops_bf16.gat_layer_supported(feat, H, D) and _act_code(self.activation) is not None
                    and not isinstance(self.res_fc, Identity))
        return (feat.is_cuda and feat.dtype == torch.float32 and feat.dim() == 2 and feat.shape[0] > 0 and SCORES_FROM_FT
                and D % 64 == 0 and _act_code(self.activation) is not None and not isinstance(self.res_fc, Identity)
                and not (AGGREGATE_FIRST and feat.shape[1] < D and ops.agg_first_supported(H, feat.shape[1]))
                and ops.GEMM_MODE == "f16x3" and (H * D) % 4 == 0)

    def forward(self, graph: TreeGraph, feat: torch.Tensor, get_attention: bool = False, mean_heads: bool = False,
                feat_dropped: bool = False, classifier: Optional[nn.Linear] = None, fuse_out=None, attn_seed: Optional[int] = None):
        """DGL signature; ``mean_heads=True`` (extension) returns ``rst.mean(1)`` (N, D) with the mean fused
        into the kernel epilogue — what the reference applies to the output layer (models.py:327, 482).
        ``feat_dropped=True`` (extension): the caller already applied this layer's feature dropout while assembling
        ``feat`` (ops.cat_dropout fuses it into the concatenation).
        ``classifier`` (extension, with ``mean_heads``): an ``nn.Linear`` applied to the head mean (the reference's
        ``gnn_out(n_embed)``, models.py:1127-1130); the layer then returns ``(rst, classifier(rst))`` and, in the
        aggregate-first form, back-propagates through both in one node (ops._GATAggFirstFn).
        ``fuse_out`` = (total, p, seed, extra) (extension; see ``can_fuse_out``): the flattened output is written, already
        under the NEXT layer's feature dropout (p, seed), into columns [0, H*D) of a fresh (N, total) buffer - that
        layer's input, completed by ``ops.fill_cols_dropout`` when total > H*D - and ``(buffer, maxima)`` is returned
        (reference models.py:477-481: ``h_s = cat[h_s, h_p]`` then ``GATConv.feat_drop``, without the separate pass).
        ``attn_seed`` (extension): the attention-dropout seed to use instead of drawing one (a caller that draws all seeds
        of a forward pass up front, models.GATPSPGNN)."""
        self._attn_seed = attn_seed
        if fuse_out is not None:
            if classifier is not None or mean_heads or get_attention or not self.can_fuse_out(feat):
                raise ValueError("fuse_out= needs can_fuse_out(feat) and none of classifier / mean_heads / get_attention")
            return self._forward(graph, feat, False, False, feat_dropped, None, fuse_out)
        if classifier is not None:
            if not mean_heads or get_attention:
                raise ValueError("classifier= needs mean_heads=True and get_attention=False")
            return self._forward_with_classifier(graph, feat, feat_dropped, classifier)
        return self._forward(graph, feat, get_attention, mean_heads, feat_dropped, None)

    _attn_seed: Optional[int] = None

    def _forward_with_classifier(self, graph, feat, feat_dropped, classifier):
        res = self._forward(graph, feat, False, True, feat_dropped, classifier)
        if isinstance(res, tuple):
            return res
        return res, classifier(res)

    def _forward(self, graph, feat, get_attention, mean_heads, feat_dropped, classifier, fuse_out=None):
        csc = graph.csc(feat.device)
        if not self._allow_zero_in_degree and csc.min_in_degree == 0:
            raise DGLError("There are 0-in-degree nodes in the graph, output for those nodes will be invalid. "
                           "Add self-loops (g.add_edges(g.nodes(), g.nodes())) or set allow_zero_in_degree.")
        H, D = self._num_heads, self._out_feats
        h = feat if feat_dropped else self.feat_drop(feat)
        has_res = isinstance(self.res_fc, nn.Linear)
        identity_res = isinstance(self.res_fc, Identity)
        act = _act_code(self.activation)
        fuse_epilogue = act is not None and not identity_res
        w_fc = self.fc.weight
        # el = (fc(x) * attn_l).sum(-1) = x @ (attn_l . W_h)^T : fold the score vectors through fc
        p = float(self.attn_drop.p) if self.training else 0.0
        seed = (self._attn_seed if self._attn_seed is not None else _draw_seed()) if p > 0.0 else 0
        self._attn_seed = None
        drop = (p, seed) if p > 0.0 else None
        fuse_mean = mean_heads and fuse_epilogue
        if h.dtype == torch.bfloat16:
            # bf16-storage path (BASELINE config 4): project-first on the bf16 matrix cores, fp32 parameters and scores
            if not ops_bf16.gat_layer_supported(h, H, D):
                raise DGLError(f"bf16 GATConv needs a ROCm device, out_feats % 64 == 0 and in_feats % 4 == 0 (got {tuple(h.shape)} -> {H}x{D})")
            F_in = h.shape[1]
            if (LINEAR_MEAN and fuse_mean and fuse_out is None and act == ops.ACT_NONE and F_in < D and (H + 1) * F_in <= H * D
                    and getattr(csc, "num_dst", None) is None and ops_bf16.linear_mean_supported(h, H, F_in)):
                # output layer without activation, heads averaged: one product on [z_0 .. z_{H-1} | x] (ops._GATAggregateFn)
                w_lr = ops.fold_scores(w_fc, self.attn_l, self.attn_r)
                args = (csc, h, w_fc, self.res_fc.weight if has_res else None, w_lr, self.bias, H, D, float(self.negative_slope), p, seed)
                if FUSE_CLASSIFIER and classifier is not None and classifier.in_features == D:
                    out, attn, logits = ops_bf16.gat_layer_linear_mean(*args, w_cls=classifier.weight, b_cls=classifier.bias)
                    return self._finish(out, attn, csc, h, H, D, True, fuse_epilogue, identity_res, True, False), logits
                out, attn = ops_bf16.gat_layer_linear_mean(*args)
                return self._finish(out, attn, csc, h, H, D, True, fuse_epilogue, identity_res, True, get_attention, drop=drop)
            fuse_mean = fuse_mean and ops.can_fuse_mean(H, D)
            if fuse_out is not None:
                total, fp, fseed, extra = fuse_out
                if total != H * D or extra:
                    raise ValueError("bf16 fuse_out: single source only (total == num_heads * out_feats)")
                out, attn = ops_bf16.gat_layer(csc, h, w_fc, self.res_fc.weight if has_res else None, self.attn_l, self.attn_r,
                                               self.bias, H, D, float(self.negative_slope), act, p, seed, out_drop=(fp, fseed))
                return out, None
            out, attn = ops_bf16.gat_layer(csc, h, w_fc, self.res_fc.weight if has_res else None, self.attn_l, self.attn_r,
                                           self.bias if fuse_epilogue else None, H, D, float(self.negative_slope),
                                           act if fuse_epilogue else ops.ACT_NONE, p, seed, mean=fuse_mean)
            return self._finish(out, attn, csc, h, H, D, fuse_mean, fuse_epilogue, identity_res, mean_heads, get_attention, drop=drop)
        agg_first = (AGGREGATE_FIRST and fuse_epilogue and h.shape[1] < D and ops.GEMM_MODE == "f16x3" and h.shape[0] > 0
                     and ops.agg_first_supported(H, h.shape[1]))
        w_cat = None
        if not agg_first:                               # [W_fc ; W_res]: one projection GEMM reads the input once for both
            if h.is_cuda and ops.GEMM_MODE == "f16x3":
                # one kernel: both row blocks into 16-byte rows (e.g. 1063 -> 1064), the transpose for the input
                # gradient (not for a data input) and the split-GEMM scale
                w_cat = ops.weight_cat(w_fc, self.res_fc.weight if has_res else None, want_t=h.requires_grad)
            else:
                w_cat = torch.cat([w_fc, self.res_fc.weight], dim=0) if has_res else w_fc
                if w_cat.shape[1] % 4:
                    w_cat = F.pad(w_cat, (0, -w_cat.shape[1] % 4))[:, :w_cat.shape[1]]
        if SCORES_FROM_FT and not agg_first and ops.scores_from_ft_supported(h, w_cat, D):
            # el / er from ft in the projection GEMM's epilogue (DGL's own formulation): ops._GATLayerScoresFromFtFn
            if fuse_out is not None:
                total, fp, fseed, extra = fuse_out
                if total % 4 or getattr(csc, "num_dst", None) is not None:
                    raise ValueError("fuse_out: the buffer width must be a multiple of 4 and the graph not a block")
                buf, attn, amax = ops.gat_layer_scores_from_ft(csc, h, w_cat, self.attn_l, self.attn_r, self.bias, H, D, has_res,
                                                               float(self.negative_slope), act, p, seed, fuse=fuse_out)
                if extra == 0:                         # nothing to add: the buffer is complete, its GEMM scale known
                    buf._spgnn_scale = (buf._version, amax)          # the layer's scale block
                return buf, amax
            out, attn = ops.gat_layer_scores_from_ft(csc, h, w_cat, self.attn_l, self.attn_r,
                                                     self.bias if fuse_epilogue else None, H, D, has_res,
                                                     float(self.negative_slope), act if fuse_epilogue else ops.ACT_NONE, p,
                                                     seed, mean=fuse_mean)
            return self._finish(out, attn, csc, h, H, D, fuse_mean, fuse_epilogue, identity_res, mean_heads, get_attention, drop=drop)
        w_lr = ops.fold_scores(w_fc, self.attn_l, self.attn_r)
        if (agg_first and LINEAR_MEAN and fuse_mean and act == ops.ACT_NONE and (H + 1) * h.shape[1] <= H * D
                and getattr(csc, "num_dst", None) is None):
            # no activation between the projection and the head mean: the layer is linear in [z_0 .. z_{H-1} | x]
            # (ops._GATAggregateFn): one product, no per-head (N, H*D) tensors in either direction
            args = (csc, h, w_fc, self.res_fc.weight if has_res else None, w_lr, self.bias, H, D, float(self.negative_slope), p, seed)
            if FUSE_CLASSIFIER and classifier is not None and classifier.in_features == D:
                out, attn, logits = ops.gat_layer_linear_mean(*args, w_cls=classifier.weight, b_cls=classifier.bias)
                return self._finish(out, attn, csc, h, H, D, True, fuse_epilogue, identity_res, True, False), logits
            out, attn = ops.gat_layer_linear_mean(*args)
            return self._finish(out, attn, csc, h, H, D, True, fuse_epilogue, identity_res, True, get_attention, drop=drop)
        if agg_first:
            # input narrower than one head's output: aggregate the input rows, then project (ops._GATAggFirstFn)
            fuse_cls = (FUSE_CLASSIFIER and classifier is not None and fuse_mean and classifier.out_features <= 32
                        and classifier.in_features == D and getattr(csc, "num_dst", None) is None)
            if fuse_cls:
                out, attn, logits = ops.gat_layer_agg_first(csc, h, w_fc, self.res_fc.weight if has_res else None, w_lr,
                                                            self.bias, H, D, float(self.negative_slope), act, p, seed,
                                                            mean=True, w_cls=classifier.weight, b_cls=classifier.bias)
                return self._finish(out, attn, csc, h, H, D, True, fuse_epilogue, identity_res, True, False), logits
            out, attn = ops.gat_layer_agg_first(csc, h, w_fc, self.res_fc.weight if has_res else None, w_lr, self.bias, H, D,
                                                float(self.negative_slope), act, p, seed, mean=fuse_mean)
        else:
            out, attn = ops.gat_layer(csc, h, w_cat, w_lr, self.bias if fuse_epilogue else None, H, D, has_res,
                                      float(self.negative_slope), act if fuse_epilogue else ops.ACT_NONE, p, seed,
                                      mean=fuse_mean)
        return self._finish(out, attn, csc, h, H, D, fuse_mean, fuse_epilogue, identity_res, mean_heads, get_attention, drop=drop)

    def prep_spec(self, want_t: bool, mean_heads: bool = False):
        """This layer's entry for ops.prepared_weights, from its static shape (what forward() will pick for an fp32 input):
        the aggregate-first form with an activation takes [W_fc | W_res] (``"cols"``), the linear-mean form assembles its
        own operand (None), every other layer the row concatenation [W_fc ; W_res]."""
        H, D, F_in = self._num_heads, self._out_feats, self._in_src_feats
        has_res = isinstance(self.res_fc, nn.Linear)
        act = _act_code(self.activation)
        w_b = self.res_fc.weight if has_res else None
        agg_first = (AGGREGATE_FIRST and act is not None and not isinstance(self.res_fc, Identity) and F_in < D
                     and ops.agg_first_supported(H, F_in))
        if not agg_first:
            return (self.fc.weight, w_b, want_t)
        if LINEAR_MEAN and mean_heads and act == ops.ACT_NONE and (H + 1) * F_in <= H * D:
            return None
        if F_in % 4 or D % 4:
            return None
        return (self.fc.weight, w_b, want_t, "cols")

    def _finish(self, out, attn, csc, h, H, D, fuse_mean, fuse_epilogue, identity_res, mean_heads, get_attention, drop=None):
        """``drop`` = (p, seed) of the attention dropout this forward used (training mode, p > 0), else None."""
        rst = out if fuse_mean else out.view(-1, H, D)
        if not fuse_epilogue:
            if identity_res:
                rst = rst + h.view(h.shape[0], -1, D)
            if self.bias is not None:
                rst = rst + self.bias.view(1, H, D)
            if self.activation is not None:
                rst = self.activation(rst)
        if mean_heads and not fuse_mean:
            rst = rst.mean(1)
        nd = getattr(csc, "num_dst", None)             # a Block: the dst nodes are the first rows (graph.Block)
        if nd is not None:
            rst = rst[:nd]
        if get_attention:
            if drop is not None:                       # DGL hands back attn_drop(edge_softmax(e)): what the aggregation used
                attn = attn * ops.attn_dropout_multiplier(attn.shape[0], H, drop[0], drop[1], attn.device)
            a = torch.empty_like(attn)
            a[csc.eid.long()] = attn                   # CSC slot order -> edge id order
            return rst, a.unsqueeze(-1)
        return rst


class GraphConv(nn.Module):
    """dgl.nn.pytorch.GraphConv (Appendix A.2); reference models.py:172-182."""

    def __init__(self, in_feats, out_feats, norm="both", weight=True, bias=True, activation=None,
                 allow_zero_in_degree=False):
        super().__init__()
        if norm not in ("none", "both", "right", "left"):
            raise DGLError(f'Invalid norm value. Must be either "none", "both", "right" or "left". But got "{norm}".')
        self._in_feats, self._out_feats, self._norm = in_feats, out_feats, norm
        self._allow_zero_in_degree = allow_zero_in_degree
        if weight:
            self.weight = nn.Parameter(torch.empty(in_feats, out_feats))
        else:
            self.register_parameter("weight", None)
        if bias:
            self.bias = nn.Parameter(torch.empty(out_feats))
        else:
            self.register_parameter("bias", None)
        self._activation = activation
        self.reset_parameters()

    def reset_parameters(self):
        if self.weight is not None:
            nn.init.xavier_uniform_(self.weight)
        if self.bias is not None:
            nn.init.zeros_(self.bias)

    def set_allow_zero_in_degree(self, set_value):
        self._allow_zero_in_degree = set_value

    def forward(self, graph: TreeGraph, feat: torch.Tensor, weight=None, classifier: Optional[nn.Linear] = None):
        """``classifier`` (extension): the ``*Net``'s ``gnn_out``; returns ``(rst, classifier(rst))`` - folded through the
        layer's product when the layer is linear and projects after aggregating (ops._LinearClassifierFn)."""
        if classifier is None:
            return self._forward(graph, feat, weight, None)
        res = self._forward(graph, feat, weight, classifier)
        return res if isinstance(res, tuple) else (res, classifier(res))

    def _forward(self, graph: TreeGraph, feat: torch.Tensor, weight, classifier):
        csc = graph.csc(feat.device)
        if not self._allow_zero_in_degree and csc.min_in_degree == 0:
            raise DGLError("There are 0-in-degree nodes in the graph.")
        if weight is not None and self.weight is not None:
            raise DGLError("External weight is provided while at the same time the module has defined its own weight.")
        weight = self.weight if weight is None else weight
        w_src = w_dst = None
        if self._norm in ("left", "both"):
            w_src = csc.degree_scale("out", -0.5 if self._norm == "both" else -1.0)
        if self._norm in ("right", "both"):
            w_dst = csc.degree_scale("in", -0.5 if self._norm == "both" else -1.0)
        # bias and activation ride in the epilogue of whichever kernel comes last (FUSE_EPILOGUES): the SpMM when the
        # projection comes first (in_feats > out_feats), else the projection GEMM
        act = _act_code_dense(self._activation)
        fuse = FUSE_EPILOGUES and feat.is_cuda and act is not None and getattr(csc, "num_dst", None) is None and self._out_feats % 4 == 0
        if self._in_feats > self._out_feats:       # mult W first to reduce the aggregated width
            if weight is not None:
                feat = ops.linear(feat, weight.t())
            if fuse:
                return ops.spmm_sum(csc, feat, w_src, w_dst, bias=self.bias, act=act)
            rst = _dst_rows(csc, ops.spmm_sum(csc, feat, w_src, w_dst))
        else:
            rst = _dst_rows(csc, ops.spmm_sum(csc, feat, w_src, w_dst))
            if classifier is not None and rst.is_cuda and getattr(csc, "num_dst", None) is None:
                rst = ops.take_loss_rows(rst, True)      # a loss-rows step: the product and the classifier on the kept rows
            if weight is not None:
                if (fuse and classifier is not None and act == ops.ACT_NONE
                        and ops.linear_classifier_supported(rst, weight.t(), classifier.weight)):
                    # a linear output layer: the classifier folds through the product - no pass over the (N, out) result for
                    # the logits and no (N, out) gradient in the backward pass
                    if ops.FUSE_LINEAR_MEAN_FOLD and rst.shape[1] % 4 == 0:
                        # ... with P = Wc W, c0 and the parameters' gradients from the one-head, no-x-block case of the fold kernels
                        return ops._LinearMeanClassifierFn.apply(rst, weight.t(), None, self.bias, classifier.weight, classifier.bias,
                                                                 1, self._out_feats, False)
                    return ops._LinearClassifierFn.apply(rst, weight.t(), self.bias, classifier.weight, classifier.bias)
                if fuse:
                    return ops.linear(rst, weight.t(), self.bias, act)
                rst = ops.linear(rst, weight.t())
        if self.bias is not None:
            rst = rst + self.bias
        if self._activation is not None:
            rst = self._activation(rst)
        return rst


def _dst_rows(csc, x: torch.Tensor) -> torch.Tensor:
    """Rows of the destination nodes: all of them on a graph, the leading ``num_dst`` on a Block (graph.Block holds
    a block as a square graph over its src nodes, dst nodes first — DGL's ``expand_as_pair`` slice feat[:num_dst])."""
    nd = getattr(csc, "num_dst", None)
    return x if nd is None else x[:nd]


def _hash_dropout(drop: nn.Dropout, x: torch.Tensor) -> torch.Tensor:
    """``drop(x)`` by the counter-hash kernel on the GPU (spgnn_cat_dropout: no mask tensor, the backward regenerates it, and
    the result carries its GEMM operand scale); torch's dropout elsewhere.  fp32 rows of 16-byte-aligned width only."""
    p = float(drop.p) if drop.training else 0.0
    if p == 0.0:
        return x
    if not (FUSE_EPILOGUES and x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and x.shape[1] % 4 == 0 and p < 1.0):
        return drop(x)
    return ops.cat_dropout((x,), p, _draw_seed())


GIN_PROJECT_FIRST = True  # GINConv applies its MLP's first Linear before the aggregation when that narrows the rows
FUSE_EPILOGUES = True     # bias / activation of GraphConv, the GIN MLP and SAGEConv inside the producing kernel's epilogue


def _apply_fast_sequence(mods, x: torch.Tensor) -> torch.Tensor:
    """The modules of an ``nn.Sequential`` applied in order, see _apply_fast_linear."""
    i = 0
    while i < len(mods):
        m = mods[i]
        if type(m) is nn.Linear and FUSE_EPILOGUES and x.is_cuda:
            j = i + 1
            drop = None
            if j < len(mods) and type(mods[j]) is nn.Dropout:
                drop, j = mods[j], j + 1
            code = _act_code_dense(mods[j]) if j < len(mods) and isinstance(mods[j], (nn.LeakyReLU, nn.ReLU)) else None
            if code is not None:
                pd = float(drop.p) if (drop is not None and drop.training) else 0.0
                if 0.0 < pd < 1.0 and ops.linear_drop_supported(x, m.weight):
                    # dropout rides with the product: its mask in the epilogue, its backward and the activation's in one pass
                    x = ops.linear(x, m.weight, m.bias, code, drop=(pd, _draw_seed()))
                else:
                    x = ops.linear(x, m.weight, m.bias, code)
                    if drop is not None:
                        x = _hash_dropout(drop, x)
                i = j + 1
                continue
        x = _apply_fast_linear(m, x)
        i += 1
    return x


def _apply_fast_sequence_classifier(mods, x: torch.Tensor, classifier: nn.Linear):
    """(h, classifier(h)) for h = the modules applied in order; a trailing Linear + activation takes the classifier into its
    autograd node (ops.linear_act_classifier)."""
    k = len(mods) - 1
    code = _act_code_dense(mods[k]) if k >= 1 and isinstance(mods[k], (nn.LeakyReLU, nn.ReLU)) else None
    if code is not None and type(mods[k - 1]) is nn.Linear and FUSE_EPILOGUES and x.is_cuda:
        h = _apply_fast_sequence(mods[:k - 1], x)
        last = mods[k - 1]
        if ops.linear_act_classifier_supported(h, last.weight, classifier.weight):
            return ops.linear_act_classifier(h, last.weight, last.bias, code, classifier.weight, classifier.bias)
        h = ops.linear(h, last.weight, last.bias, code)
        return h, classifier(h)
    h = _apply_fast_sequence(mods, x)
    return h, classifier(h)


def _apply_fast_linear(module: nn.Module, x: torch.Tensor, classifier: Optional[nn.Linear] = None):
    """``module(x)`` with every plain ``nn.Linear`` (also inside an ``nn.Sequential``, e.g. the reference's GIN MLP,
    models.py:236-246: Linear, Dropout, LeakyReLU, Linear, LeakyReLU) evaluated by ops.linear on the matrix-core GEMMs; any
    other module runs as it is.  An activation that follows a Linear - directly or behind a Dropout - goes into that
    product's epilogue: dropout multiplies by 0 or 1 / (1 - p) >= 0 and LeakyReLU / ReLU are positively homogeneous, so
    ``act(dropout(y)) == dropout(act(y))`` (to the last bit but one: the two scalings swap).
    ``classifier`` (a Linear with <= 32 outputs, the *Net's ``gnn_out``): returns ``(module(x), classifier(module(x)))``; when
    the module ends in Linear + activation the classifier joins that product's autograd node (ops.linear_act_classifier)."""
    if classifier is not None:
        return _apply_fast_sequence_classifier(list(module) if type(module) is nn.Sequential else [module], x, classifier)
    if type(module) is nn.Linear:
        return ops.linear(x, module.weight, module.bias)
    if type(module) is nn.Sequential:
        return _apply_fast_sequence(list(module), x)
    return module(x)


class GINConv(nn.Module):
    """dgl.nn.pytorch.GINConv (Appendix A.3); reference models.py:358-383
    (``GINConv(apply_func, "mean", learn_eps=True)``)."""

    def __init__(self, apply_func=None, aggregator_type="sum", init_eps=0, learn_eps=False, activation=None):
        super().__init__()
        if aggregator_type not in ("sum", "max", "mean"):
            raise KeyError(f"Aggregator type {aggregator_type} not recognized.")
        self.apply_func = apply_func
        self._aggregator_type = aggregator_type
        self.activation = activation
        if learn_eps:
            self.eps = nn.Parameter(torch.FloatTensor([init_eps]))
        else:
            self.register_buffer("eps", torch.FloatTensor([init_eps]))

    def forward(self, graph: TreeGraph, feat: torch.Tensor, edge_weight=None, classifier: Optional[nn.Linear] = None):
        """``classifier`` (extension): the *Net's ``gnn_out``; returns ``(rst, classifier(rst))``, joined to the MLP's last
        product when the layer has no activation of its own."""
        if edge_weight is not None:
            raise DGLError("edge_weight is not supported")
        csc = graph.csc(feat.device)
        mods = list(self.apply_func) if type(self.apply_func) is nn.Sequential else [self.apply_func]
        first = mods[0]
        if (GIN_PROJECT_FIRST and FUSE_EPILOGUES and self._aggregator_type != "max" and type(first) is nn.Linear
                and first.in_features > first.out_features and first.out_features % 4 == 0 and feat.is_cuda
                and feat.dtype == torch.float32 and getattr(csc, "num_dst", None) is None):
            # The MLP's first Linear BEFORE the aggregation: both are linear, ((1 + eps) x + A x) W^T = (1 + eps) u + A u with
            # u = x W^T, so the gather, its transpose in the backward pass and eps' gradient run on out_features-wide rows
            # (1024 -> 256 on the first layer), and the input gradient of the first layer's product is never needed.  Bias,
            # activation and dropout go into the aggregation's epilogue.
            j, drop = 1, None
            if j < len(mods) and type(mods[j]) is nn.Dropout:
                drop, j = mods[j], j + 1
            code = _act_code_dense(mods[j]) if j < len(mods) and isinstance(mods[j], (nn.LeakyReLU, nn.ReLU)) else None
            if code is None:
                j, drop, code = 1, None, ops.ACT_NONE            # Linear alone: the rest of the MLP runs as it is
            else:
                j += 1
            u = ops.linear(feat, first.weight, None)
            w_dst = csc.degree_scale("in", -1.0) if self._aggregator_type == "mean" else None
            pd = float(drop.p) if (drop is not None and drop.training) else 0.0
            rst = ops.spmm_sum(csc, u, None, w_dst, self.eps, bias=first.bias, act=code,
                               drop=(pd, _draw_seed()) if 0.0 < pd < 1.0 else None)
            if pd >= 1.0:
                rst = drop(rst)
            if classifier is not None:
                rst = ops.take_loss_rows(rst, True)      # a loss-rows step: the rest of the MLP and the classifier on the kept rows
            rest = mods[j:]
            if classifier is not None and self.activation is None and rest:
                return _apply_fast_sequence_classifier(rest, rst, classifier)
            rst = _apply_fast_sequence(rest, rst)
            if self.activation is not None:
                rst = self.activation(rst)
            return rst if classifier is None else (rst, classifier(rst))
        if self._aggregator_type == "max":
            rst = _dst_rows(csc, (1 + self.eps) * feat + ops.spmm_max(csc, feat))
        else:
            w_dst = csc.degree_scale("in", -1.0) if self._aggregator_type == "mean" else None
            rst = _dst_rows(csc, ops.spmm_sum(csc, feat, None, w_dst, self.eps))     # (1+eps)*x fused into the SpMM
        if classifier is not None and rst.is_cuda and getattr(csc, "num_dst", None) is None:
            rst = ops.take_loss_rows(rst, True)          # a loss-rows step: the MLP and the classifier on the kept rows
        if classifier is not None and self.apply_func is not None and self.activation is None:
            return _apply_fast_linear(self.apply_func, rst, classifier)
        if self.apply_func is not None:
            rst = _apply_fast_linear(self.apply_func, rst)
        if self.activation is not None:
            rst = self.activation(rst)
        return rst if classifier is None else (rst, classifier(rst))


class SAGEConv(nn.Module):
    """dgl.nn.pytorch.SAGEConv (Appendix A.4); reference models.py:668-679 (aggregator 'pool').

    Parameter layout follows DGL 0.6: ``fc_pool`` / ``fc_self`` / ``fc_neigh`` Linear layers, the
    latter two carrying the bias (``bias=True``).  A DGL >= 0.7 checkpoint (bias-free fc_self/fc_neigh
    plus a separate ``bias``) loads through ``_load_from_state_dict`` below.
    """

    def __init__(self, in_feats, out_feats, aggregator_type, feat_drop=0., bias=True, norm=None, activation=None):
        super().__init__()
        if aggregator_type not in ("mean", "pool", "gcn"):
            raise DGLError(f"Invalid aggregator_type {aggregator_type!r}: this build supports 'mean', 'pool', 'gcn'.")
        self._in_src_feats = self._in_dst_feats = in_feats
        self._out_feats, self._aggre_type = out_feats, aggregator_type
        self.norm, self.feat_drop, self.activation = norm, nn.Dropout(feat_drop), activation
        if aggregator_type == "pool":
            self.fc_pool = nn.Linear(in_feats, in_feats)
        if aggregator_type != "gcn":
            self.fc_self = nn.Linear(in_feats, out_feats, bias=bias)
        self.fc_neigh = nn.Linear(in_feats, out_feats, bias=bias)
        self.reset_parameters()

    def reset_parameters(self):
        gain = nn.init.calculate_gain("relu")
        if self._aggre_type == "pool":
            nn.init.xavier_uniform_(self.fc_pool.weight, gain=gain)
        if self._aggre_type != "gcn":
            nn.init.xavier_uniform_(self.fc_self.weight, gain=gain)
        nn.init.xavier_uniform_(self.fc_neigh.weight, gain=gain)

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        key = prefix + "bias"                       # DGL >= 0.7 layout: fold the shared bias into fc_neigh
        if key in state_dict and prefix + "fc_neigh.bias" not in state_dict:
            b = state_dict.pop(key)
            state_dict[prefix + "fc_neigh.bias"] = b
            if hasattr(self, "fc_self") and prefix + "fc_self.bias" not in state_dict:
                state_dict[prefix + "fc_self.bias"] = torch.zeros_like(b)
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def forward(self, graph: TreeGraph, feat: torch.Tensor, edge_weight=None, classifier: Optional[nn.Linear] = None,
                feat_dropped: bool = False, out_drop=None):
        """``classifier`` (extension): the ``*Net``'s ``gnn_out``; returns ``(rst, classifier(rst))`` - joined to this layer's
        product when the layer is linear (no activation, no norm) and takes the K-concatenated form below.
        ``feat_dropped`` / ``out_drop`` = (p, seed) (extensions, as GATConv's): ``feat`` already carries this layer's feature
        dropout / the result is to carry the NEXT layer's (reference models.py:691-696 stacks the layers, each starting with
        ``feat_drop``) - applied by the last product's epilogue where that exists, by the hash-dropout kernel otherwise."""
        if classifier is None:
            return self._out_dropped(self._forward(graph, feat, edge_weight, None, feat_dropped, out_drop), out_drop)
        res = self._forward(graph, feat, edge_weight, classifier, feat_dropped, None)
        return res if isinstance(res, tuple) else (res, classifier(res))

    @staticmethod
    def _out_dropped(res, out_drop):
        rst, done = res if isinstance(res, tuple) else (res, False)
        if out_drop is not None and not done:
            rst = ops.cat_dropout((rst,), out_drop[0], out_drop[1])
        return rst

    def _forward(self, graph: TreeGraph, feat: torch.Tensor, edge_weight, classifier, feat_dropped=False, out_drop=None):
        """-> rst, or (rst, True) when ``out_drop`` was applied by the last product's epilogue."""
        if edge_weight is not None:
            raise DGLError("edge_weight is not supported")
        csc = graph.csc(feat.device)
        h = feat if feat_dropped else _hash_dropout(self.feat_drop, feat)
        act = _act_code_dense(self.activation)
        fuse = FUSE_EPILOGUES and h.is_cuda and act is not None and self._out_feats % 4 == 0
        if self._aggre_type in ("pool", "mean"):
            if self._aggre_type == "pool" and ops.pool_max_supported(csc, h, self.fc_pool.weight):
                neigh = ops.pool_max(csc, h, self.fc_pool.weight, self.fc_pool.bias)       # one node: relu' inside the routing kernel
            elif self._aggre_type == "pool":
                pool = ops.linear(h, self.fc_pool.weight, self.fc_pool.bias, ops.ACT_RELU)
                neigh = _dst_rows(csc, ops.spmm_max(csc, pool))
                tag = getattr(pool, "_spgnn_scale", None)
                if tag is not None and tag[0] == pool._version and neigh.is_cuda:
                    # every element of the neighbourhood maximum is an element of ``pool`` (or 0): pool's operand scale,
                    # which its product's epilogue left behind, bounds it - no absmax pass over ``neigh``
                    neigh._spgnn_scale = (neigh._version, tag[1])
            else:
                neigh = _dst_rows(csc, ops.spmm_sum(csc, h, None, csc.degree_scale("in", -1.0)))
            F_in = self._in_src_feats
            if (fuse and self._out_feats >= 2 * F_in and F_in % 4 == 0 and getattr(csc, "num_dst", None) is None
                    and ops.linear_drop_supported(h, self.fc_neigh.weight) and 2 * F_in >= 32):
                # Output wider than both inputs together (64 -> 1024): ONE product on [neigh | h] with [W_neigh | W_self]
                # (the addend form writes, re-reads and re-writes the (N, out) result: 0.94 GB against 0.35 GB here)
                xc = ops.cat_dropout((neigh, h), 0.0, 0)
                if classifier is not None:
                    xc = ops.take_loss_rows(xc, True)    # a loss-rows step: the product and the classifier on the kept rows
                bs = [b for b in (self.fc_neigh.bias, self.fc_self.bias) if b is not None]
                bc = (bs[0] + bs[1] if len(bs) == 2 else bs[0]) if bs else None
                if (classifier is not None and act == ops.ACT_NONE and self.norm is None and classifier.weight.shape[0] <= 32
                        and classifier.weight.shape[1] == self._out_feats and ops.FUSE_LINEAR_MEAN_FOLD):
                    # linear layer: the classifier folds through the product - no pass over the (N, out) result for the logits,
                    # no (N, out) gradient in the backward pass; [W_neigh | W_self], P, c0 and the way back to the parameters'
                    # gradients are the one-head case of spgnn_linear_mean_fold_fwd / _bwd (two launches)
                    return ops._LinearMeanClassifierFn.apply(xc, self.fc_neigh.weight, self.fc_self.weight, bc, classifier.weight,
                                                             classifier.bias, 1, self._out_feats, True)
                wc = torch.cat([self.fc_neigh.weight, self.fc_self.weight], dim=1)
                if (classifier is not None and act == ops.ACT_NONE and self.norm is None
                        and ops.linear_classifier_supported(xc, wc, classifier.weight)):
                    return ops._LinearClassifierFn.apply(xc, wc, bc, classifier.weight, classifier.bias)
                in_ep = out_drop is not None and self.norm is None and classifier is None
                rst = ops.linear(xc, wc, bc, act, drop=out_drop if in_ep else None)
                return (rst, True) if in_ep else (rst if self.norm is None else self.norm(rst))
            if fuse:        # fc_neigh's product adds fc_self's result and applies the activation in its epilogue
                in_ep = (out_drop is not None and self.norm is None and classifier is None
                         and ops.linear_drop_supported(neigh, self.fc_neigh.weight))
                rst = ops.linear(neigh, self.fc_neigh.weight, self.fc_neigh.bias, act,
                                 addend=ops.linear(_dst_rows(csc, h), self.fc_self.weight, self.fc_self.bias),
                                 drop=out_drop if in_ep else None)
                return (rst, True) if in_ep else (rst if self.norm is None else self.norm(rst))
            rst = (ops.linear(_dst_rows(csc, h), self.fc_self.weight, self.fc_self.bias)
                   + ops.linear(neigh, self.fc_neigh.weight, self.fc_neigh.bias))
        else:  # gcn: (sum_in x_u + x_v) / (deg + 1)
            neigh = _dst_rows(csc, (ops.spmm_sum(csc, h) + h) / (csc.in_degrees_f().unsqueeze(-1) + 1))
            rst = ops.linear(neigh, self.fc_neigh.weight, self.fc_neigh.bias)
        if self.activation is not None:
            rst = self.activation(rst)
        if self.norm is not None:
            rst = self.norm(rst)
        return rst
