"""ORACLE (test infrastructure) — second, independent formulation of the same layers on a
dense (N x N) adjacency mask. Used only to cross-check ``oracle/dgl_cpu.py`` (forward,
attention and all gradients, fp64) since the reference offers nothing to pin against
(PARITY UNPINNED, see dgl_cpu.py header). Small graphs only: memory is O(N^2 H).
"""
from __future__ import annotations

from typing import Callable, Optional

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


def dense_mask(src: Tensor, dst: Tensor, n: int, dtype) -> Tensor:
    """M[v, u] = number of edges u -> v (0/1 for the reference's graphs)."""
    m = torch.zeros(n, n, dtype=dtype)
    m.index_put_((dst, src), torch.ones(src.shape[0], dtype=dtype), accumulate=True)
    return m


def gat_conv_dense(mask: Tensor, feat: Tensor, fc_weight: Tensor, attn_l: Tensor, attn_r: Tensor,
                   res_fc_weight: Optional[Tensor] = None, bias: Optional[Tensor] = None,
                   negative_slope: float = 0.2, activation: Optional[Callable] = None):
    n = feat.shape[0]
    _, H, D = attn_l.shape
    ft = (feat @ fc_weight.t()).view(n, H, D)
    el = torch.einsum("nhd,hd->nh", ft, attn_l[0])
    er = torch.einsum("nhd,hd->nh", ft, attn_r[0])
    score = F.leaky_relu(er.unsqueeze(1) + el.unsqueeze(0), negative_slope)      # [v, u, h]
    score = score.masked_fill(mask.unsqueeze(-1) == 0, float("-inf"))
    alpha = torch.softmax(score, dim=1)                                           # over u
    rst = torch.einsum("vuh,uhd->vhd", alpha, ft)
    if res_fc_weight is not None:
        rst = rst + (feat @ res_fc_weight.t()).view(n, H, D)
    if bias is not None:
        rst = rst + bias.view(1, H, D)
    if activation is not None:
        rst = activation(rst)
    return rst, alpha


def graph_conv_dense(mask: Tensor, feat: Tensor, weight: Tensor, bias: Optional[Tensor], activation=None):
    in_deg = mask.sum(1).clamp(min=1)
    out_deg = mask.sum(0).clamp(min=1)
    a_hat = in_deg.pow(-0.5).unsqueeze(1) * mask * out_deg.pow(-0.5).unsqueeze(0)
    rst = a_hat @ feat @ weight
    if bias is not None:
        rst = rst + bias
    return activation(rst) if activation is not None else rst


def gin_conv_dense(mask: Tensor, feat: Tensor, eps: Tensor, apply_func=None):
    neigh = (mask @ feat) / mask.sum(1).clamp(min=1).unsqueeze(1)
    rst = (1 + eps) * feat + neigh
    return apply_func(rst) if apply_func is not None else rst


def sage_conv_pool_dense(mask: Tensor, feat: Tensor, fc_pool_w, fc_pool_b, fc_self_w, fc_self_b, fc_neigh_w,
                         fc_neigh_b, bias=None, activation=None):
    m = F.relu(feat @ fc_pool_w.t() + fc_pool_b)                                   # (N,F)
    big = m.unsqueeze(0).expand(mask.shape[0], -1, -1).masked_fill(mask.unsqueeze(-1) == 0, float("-inf"))
    neigh = big.max(dim=1).values
    neigh = torch.where(torch.isinf(neigh), torch.zeros_like(neigh), neigh)
    rst = feat @ fc_self_w.t() + neigh @ fc_neigh_w.t()
    if fc_self_b is not None:
        rst = rst + fc_self_b
    if fc_neigh_b is not None:
        rst = rst + fc_neigh_b
    if bias is not None:
        rst = rst + bias
    return activation(rst) if activation is not None else rst
