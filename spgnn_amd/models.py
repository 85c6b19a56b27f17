"""GNN heads and ``*Net`` wrappers with the reference's class names, constructor kwargs
(= the ``MODEL`` dict keys of exp_settings/*.py), methods and state_dict keys, built on
:mod:`spgnn_amd.nn` instead of DGL.

Reference: models.py:160-194 (GCN), 283-340 (GAT), 343-400 (GIN), 403-484 (GATPSPGNN),
487-540 (GATPSPGNNNL), 650-696 (SAGE) and the wrappers at 196-281, 725-822, 824-933,
936-1047, 1050-1174.

The 3-D CNN trunk of the wrappers (``ds_modules``/``bg``/``fc``/``out``, built from the
reference's ``parts.ConvBlock5d``) is outside the hot path and frozen during GNN training
(reference models.py:1127-1130).  It is only instantiated with ``build_trunk=True`` and then
imports ``parts`` from the caller's environment; the GNN-stage entry points
(``forward(g)``, ``forward_emb(g)``, ``set_gcn_only``, ``init``) never touch it.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops, ops_bf16
from .nn import GATConv, GINConv, GraphConv, SAGEConv, SkinnyLinear, _draw_seed
from .ops import cat_padded

__all__ = ["GCN", "GAT", "GIN", "SAGE", "GATPSPGNN", "GATPSPGNNNL", "GCNNet", "GATNet", "GINNet", "SAGENet",
           "GATPositionSPGNNNet", "set_trainable", "set_storage_dtype"]


def _cat_for(layer, a: torch.Tensor, b: torch.Tensor):
    """(cat[a, b] with ``layer``'s feature dropout applied, True) when the fused kernel can be used (GPU), else
    (plain cat, False): the layer then applies its own nn.Dropout."""
    if not a.is_cuda:
        return torch.cat([a, b], dim=1), False
    p = float(layer.feat_drop.p) if layer.training else 0.0
    if a.dtype == torch.bfloat16:
        return ops_bf16.cat_dropout((a, b), p, _draw_seed() if p > 0.0 else 0), True
    return ops.cat_dropout((a, b), p, _draw_seed() if p > 0.0 else 0), True


def _drop_for(layer, x: torch.Tensor, seed: Optional[int] = None):
    """(x with ``layer``'s feature dropout applied by the hash-mask kernel, True) on the GPU in training mode."""
    p = float(layer.feat_drop.p) if layer.training else 0.0
    if not x.is_cuda or p == 0.0:
        return x, False
    seed = _draw_seed() if seed is None else seed
    if x.dtype == torch.bfloat16:
        return ops_bf16.cat_dropout((x,), p, seed), True
    return ops.cat_dropout((x,), p, seed), True


FUSE_OUTPUT_DROPOUT = True     # hidden GATConvs write their rows straight into the next layer's input buffer, already under that
                               # layer's feature dropout (nn.GATConv fuse_out); False: separate concat + dropout pass (tests flip it)


def _fuse_plan(layer, nxt, x: torch.Tensor, extra_width: int, seed: Optional[int] = None):
    """(total, p, seed, extra) for ``layer(g, x, fuse_out=...)`` feeding ``nxt`` together with ``extra_width`` more columns,
    or None when the fused form is not available."""
    if not (FUSE_OUTPUT_DROPOUT and layer.can_fuse_out(x)):
        return None
    total = layer._num_heads * layer._out_feats + extra_width
    if total % 4 or (extra_width and ((layer._num_heads * layer._out_feats) % 4 or x.dtype != torch.float32)):
        return None
    p = float(nxt.feat_drop.p) if nxt.training else 0.0
    return total, p, ((_draw_seed() if seed is None else seed) if p > 0.0 else 0), ops.fused_extra_partials(x.shape[0], extra_width)


def _prep_specs(levels, on_gpu: bool, output=None):
    """The ops.prepared_weights entries of a forward pass, level by level (each level a layer or a tuple of layers; GATConv.
    prep_spec): the input of the first level is node data, so only later levels need the transposed operand (their input
    gradient), and none does without autograd.  ``output``: the last layer, whose heads are averaged."""
    if not on_gpu:
        return []
    specs = []
    for l, layers in enumerate(levels):
        for layer in (layers if isinstance(layers, tuple) else (layers,)):
            specs.append(layer.prep_spec(torch.is_grad_enabled() and l > 0))
    if output is not None:
        specs.append(output.prep_spec(torch.is_grad_enabled(), mean_heads=True))
    return [sp for sp in specs if sp is not None]


def _linear_specs(module: nn.Module, x: torch.Tensor):
    """ops.prepared_weights entries for every ``nn.Linear`` under ``module`` (the GIN MLPs, SAGEConv's fc_pool / fc_self /
    fc_neigh; reference models.py:236-246, 668-679): their padded rows, transposes, pre-split forms and scales come out of
    one spgnn_weight_prep call per forward pass instead of two absmax + two scale launches per Linear."""
    if not (x.is_cuda and x.dtype == torch.float32):
        return []
    want_t = torch.is_grad_enabled()
    return [(m.weight, None, want_t) for m in module.modules() if type(m) is nn.Linear and m.weight.is_cuda]


FUSE_SAGE_DROP = True          # SAGE: a layer's feature dropout applied by the previous layer's last product
FUSE_LSPE = True               # GATPSPGNN: structure + position GATConv of a level in ONE traversal (ops.lspe_level); False: two layers


def _cat(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """torch.cat([a, b], dim=1) (reference models.py:477, 481, 534, 537) with 16-byte-aligned rows on the GPU."""
    return cat_padded((a, b)) if a.is_cuda else torch.cat([a, b], dim=1)


def _ndata_key(g, t: torch.Tensor):
    """What identifies node-data tensor ``t`` in the per-batch cache of ``g``: on a batch arena (spgnn_amd/arena.py: node data
    rewritten IN PLACE by every loader batch, derived tensors refreshed in place by ``g.refresh_derived()``) its ndata name;
    otherwise address, version and shape - a replaced or modified tensor makes a new entry."""
    if getattr(g, "_stable_storage", False):
        for name, v in g.ndata.items():
            if v is t:
                return ("ndata", name)
    return (t.data_ptr(), t._version, tuple(t.shape))


def _derived(g, key, builder):
    """The per-batch constant ``key`` of ``g``, built once per loader batch by ``builder()`` (a graph without the cache:
    built every time).  The builder is kept: an arena graph recomputes every derived tensor into the same storage."""
    cache = getattr(g, "_tensor_cache", None)
    if cache is None:
        return builder()
    hit = cache.get(key)
    if hit is None:
        if len(cache) > 16 and not getattr(g, "_stable_storage", False):
            cache.clear()
            getattr(g, "_derived_builders", {}).clear()
        hit = cache[key] = ops.mark_batch_constant(builder().detach())
        if hasattr(g, "_derived_builders"):
            g._derived_builders[key] = builder
    return hit


def _data_cat(g, a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """cat of two node-DATA tensors (fvs, pos_enc).  The batched graph and its node data are reused for all
    GCN_STEPS = 300 inner steps (reference job_runner.py:1892), so the concatenation is built once per batch
    and kept on the graph (keyed by the tensors' identity and version); the GEMMs keep its scale and its pre-split image
    on it (ops.const_operand)."""
    if a.requires_grad or b.requires_grad or not a.is_cuda or not hasattr(g, "_tensor_cache"):
        return _cat(a, b)
    ka, kb = _ndata_key(g, a), _ndata_key(g, b)
    return _derived(g, ("cat", ka, kb), lambda: _cat(*_ndata_pair(g, ka, kb, a, b)))


def _ndata_pair(g, ka, kb, a, b):
    """The CURRENT tensors behind two cache keys (an arena's builders run again after the node data was rewritten)."""
    return (g.ndata[ka[1]] if ka[0] == "ndata" else a), (g.ndata[kb[1]] if kb[0] == "ndata" else b)


def _data_aligned(g, t: torch.Tensor) -> torch.Tensor:
    """A node-data tensor with 16-byte-aligned rows (e.g. pos_enc, 39 floats wide -> row stride 40), made once
    per batch: the aligned copy is what the MFMA kernels read."""
    already = t.stride(1) == 1 and t.stride(0) % 4 == 0 and t.data_ptr() % 16 == 0
    if t.requires_grad or not t.is_cuda or not hasattr(g, "_tensor_cache"):
        return t
    if already:
        return ops.mark_batch_constant(t) if any(v is t for v in g.ndata.values()) else t
    k = _ndata_key(g, t)
    return _derived(g, ("aligned", k), lambda: cat_padded((g.ndata[k[1]] if k[0] == "ndata" else t,)))


def _data_in(g, t: torch.Tensor, dtype) -> torch.Tensor:
    """A node-DATA tensor in the head's storage dtype: fp32 data -> bf16 rows (16-byte rows, zero padded) once per batch
    (the batched graph and its node data are reused for all GCN_STEPS inner steps, reference job_runner.py:1892)."""
    if dtype is None or t.dtype == dtype:
        if t.is_cuda and not t.requires_grad and t.dtype == torch.float32 and any(v is t for v in g.ndata.values()):
            ops.mark_batch_constant(t)          # fp32 node data feeding the first layer's products as it is
        return t
    if dtype != torch.bfloat16 or t.dtype != torch.float32:
        raise ValueError(f"storage dtype {dtype} from node data of dtype {t.dtype} is not supported")
    if t.requires_grad or not hasattr(g, "_tensor_cache"):
        return ops_bf16.cast_rows(t)
    k = _ndata_key(g, t)
    return _derived(g, ("bf16", k), lambda: ops_bf16.cast_rows(g.ndata[k[1]] if k[0] == "ndata" else t))


def set_storage_dtype(model: nn.Module, dtype) -> nn.Module:
    """Storage dtype of node-feature rows inside the GNN head: None / torch.float32 (the reference's arithmetic, parity
    path) or torch.bfloat16 (rows and their gradients in bf16, fp32 accumulate, fp32 parameters; BASELINE config 4)."""
    if dtype not in (None, torch.float32, torch.bfloat16):
        raise ValueError(f"unsupported storage dtype {dtype}")
    for m in model.modules():
        if hasattr(m, "storage_dtype"):
            m.storage_dtype = None if dtype == torch.float32 else dtype
    return model


def set_trainable(model: nn.Module, trainable: bool) -> None:
    for p in model.parameters():
        p.requires_grad = trainable


# =================================================================================================
# heads
# =================================================================================================
class _InferenceScoped:
    """Module calls made under torch.no_grad() announce it to the ops (ops.inference_scope): small inference batches - the
    reference's one-scan-per-forward test path, job_runner.py:2046-2052 - then take the skinny projection kernel."""

    def __call__(self, *args, **kwargs):
        with ops.inference_scope():
            return super().__call__(*args, **kwargs)


class GCN(_InferenceScoped, nn.Module):
    def __init__(self, num_layers, in_dim, num_hiddens, num_classes, activation):
        super().__init__()
        self.num_layers = num_layers
        dims = [in_dim] + list(num_hiddens[:num_layers])
        self.gcn_layers = nn.ModuleList(
            [GraphConv(dims[i], dims[i + 1], activation=activation) for i in range(num_layers)]
            + [GraphConv(dims[num_layers], num_classes)])

    def reset_parameters(self):
        for layer in self.gcn_layers:
            layer.reset_parameters()

    def forward(self, g, classifier=None):
        """``classifier`` (extension): the ``*Net``'s ``gnn_out``; returns ``(h, classifier(h))``, the classifier joined to the
        (linear) output layer's product."""
        h = _data_in(g, g.ndata["fvs"], None)      # fp32 node data: a batch constant for the first layer's products
        specs = [(l.weight, None, True) for l in self.gcn_layers if l.weight is not None] if (h.is_cuda and h.dtype == torch.float32) else []
        with ops.prepared_weights(specs):       # GraphConv's (in, out) weights: the product reads the TRANSPOSED image
            for layer in self.gcn_layers[:-1]:
                h = layer(g, h)
            return self.gcn_layers[-1](g, h, classifier=classifier)


class GAT(_InferenceScoped, nn.Module):
    def __init__(self, num_layers, in_dim, num_hiddens, out_ch, heads, activation, feat_drop, attn_drop,
                 negative_slope, residual, norm=False):
        super().__init__()
        self.num_layers, self.activation, self.out_ch, self.norm = num_layers, activation, out_ch, norm
        widths = [in_dim] + [num_hiddens[l] * heads[l] for l in range(num_layers)]
        layers = []
        for l in range(num_layers):          # layer 0 runs without dropout, like the reference
            drop = (0.0, 0.0) if l == 0 else (feat_drop, attn_drop)
            layers.append(GATConv(widths[l], num_hiddens[l], heads[l], drop[0], drop[1], negative_slope, residual,
                                  activation))
        layers.append(GATConv(widths[num_layers], out_ch, heads[num_layers], 0.0, 0.0, negative_slope, residual, None))
        self.gat_layers = nn.ModuleList(layers)
        self.storage_dtype = None            # see set_storage_dtype

    def reset_parameters(self):
        for layer in self.gat_layers:
            layer.reset_parameters()

    def _finish(self, h):
        return F.normalize(h, p=2, dim=1) if self.norm else h

    def forward(self, g, classifier=None):
        """``classifier`` (extension): the ``*Net``'s ``gnn_out``; returns ``(h, classifier(h))`` with the classifier
        joined to the output layer's autograd node (not with ``norm``: the normalisation sits in between)."""
        h = _data_in(g, g.ndata["fvs"], self.storage_dtype)
        with ops.prepared_weights(_prep_specs(self.gat_layers[:-1], h.is_cuda and h.dtype == torch.float32, output=self.gat_layers[-1])), \
                ops_bf16.prepared_weights(_prep_specs(self.gat_layers[:-1], h.is_cuda and h.dtype == torch.bfloat16)):
            return self._forward(g, h, classifier)

    def _forward(self, g, h, classifier):
        x, dropped = h, False                        # layer 0 runs without dropout
        for l, layer in enumerate(self.gat_layers[:-1]):
            plan = _fuse_plan(layer, self.gat_layers[l + 1], x, 0)
            if plan is not None:                     # the rows arrive at the next layer already under ITS feature dropout
                x, _ = layer(g, x, feat_dropped=dropped, fuse_out=plan)
                dropped = True
            else:
                h = layer(g, x, feat_dropped=dropped).flatten(1)
                x, dropped = _drop_for(self.gat_layers[l + 1], h)
        if classifier is not None and not self.norm:
            return self.gat_layers[-1](g, x, mean_heads=True, feat_dropped=dropped, classifier=classifier)
        h = self._finish(self.gat_layers[-1](g, x, mean_heads=True, feat_dropped=dropped))
        return h if classifier is None else (h, classifier(h))

    def forward_batch(self, blocks, x):
        h = x
        for layer, block in zip(self.gat_layers[:-1], blocks[:-1]):
            h = layer(block, h).flatten(1)
        return self._finish(self.gat_layers[-1](blocks[-1], h, mean_heads=True))


def _gin_mlp(n_in, n_out):
    return nn.Sequential(nn.Linear(n_in, n_out), nn.Dropout(0.1), nn.LeakyReLU(), nn.Linear(n_out, n_out),
                         nn.LeakyReLU())


class GIN(_InferenceScoped, nn.Module):
    def __init__(self, num_layers, in_dim, num_hiddens, out_ch, norm=False):
        super().__init__()
        self.num_layers, self.in_dim, self.out_ch, self.norm = num_layers, in_dim, out_ch, norm
        dims = [in_dim] + list(num_hiddens[:num_layers]) + [out_ch]
        self.gin_layers = nn.ModuleList(
            [GINConv(_gin_mlp(dims[i], dims[i + 1]), "mean", learn_eps=True) for i in range(num_layers + 1)])

    def forward(self, g, classifier=None):
        """``classifier`` (extension): the ``*Net``'s ``gnn_out``; returns ``(h, classifier(h))``, the classifier joined to the
        last MLP's second product."""
        h = _data_in(g, g.ndata["fvs"], None)      # fp32 node data: a batch constant for the first layer's products
        with ops.prepared_weights(_linear_specs(self, h)):
            for layer in self.gin_layers[:-1]:
                h = layer(g, h)
            if classifier is not None and not self.norm:
                return self.gin_layers[-1](g, h, classifier=classifier)
            h = self.gin_layers[-1](g, h)
        h = F.normalize(h, p=2, dim=1) if self.norm else h
        return h if classifier is None else (h, classifier(h))

    def forward_batch(self, blocks, x):
        h = x
        for layer, block in zip(self.gin_layers, blocks):
            h = layer(block, h)
        return F.normalize(h, p=2, dim=1) if self.norm else h


class GATPSPGNN(_InferenceScoped, nn.Module):
    """SPGNN "PEL": a structure stream on cat[h_s, h_p] and a learnable position stream
    (1-head residual GATConv, tanh) on h_p.  Reference models.py:403-484."""

    def __init__(self, num_layers, in_dim, pos_in_dim, num_hiddens, pos_hiddens, pos_heads, out_ch, heads, activation,
                 feat_drop, attn_drop, negative_slope, residual, norm=False, p_activation=torch.tanh):
        super().__init__()
        self.num_layers, self.activation, self.pos_hiddens, self.out_ch, self.norm = \
            num_layers, activation, pos_hiddens, out_ch, norm
        s_in = [in_dim + pos_in_dim] + [num_hiddens[l] * heads[l] + pos_hiddens[l] * pos_heads[l]
                                        for l in range(num_layers)]
        p_in = [pos_in_dim] + [pos_hiddens[l] * pos_heads[l] for l in range(num_layers - 1)]
        s_layers, p_layers = [], []
        for l in range(num_layers):
            s_drop = (0.0, 0.0) if l == 0 else (feat_drop, attn_drop)
            # position stream: dropout only on its middle layers (not the first, not the last)
            p_drop = (feat_drop, attn_drop) if 0 < l < num_layers - 1 else (0.0, 0.0)
            s_layers.append(GATConv(s_in[l], num_hiddens[l], heads[l], s_drop[0], s_drop[1], negative_slope, residual,
                                    activation))
            p_layers.append(GATConv(p_in[l], pos_hiddens[l], pos_heads[l], p_drop[0], p_drop[1], negative_slope, True,
                                    p_activation))
        s_layers.append(GATConv(s_in[num_layers], out_ch, heads[num_layers], 0.0, 0.0, negative_slope, residual,
                                activation))
        self.gat_layers = nn.ModuleList(s_layers)
        self.pgnn_layers = nn.ModuleList(p_layers)

    def reset_parameters(self):
        for layer in list(self.gat_layers) + list(self.pgnn_layers):
            layer.reset_parameters()

    def _seed_plan(self):
        """Every dropout seed of one forward pass, drawn up front in a fixed order (per level: the next structure layer's
        feature dropout, the structure layer's attention dropout, the position layer's attention dropout, the next position
        layer's feature dropout; only where the rate is non-zero in the current mode) - so the fused and the two-layer form
        of a level use identical masks under one ``torch.manual_seed``."""
        L = self.num_layers
        plan = []
        for l in range(L):
            s_layer, p_layer, nxt_s = self.gat_layers[l], self.pgnn_layers[l], self.gat_layers[l + 1]
            nxt_p = self.pgnn_layers[l + 1] if l + 1 < L else None
            def rate(m, which):
                return float(getattr(m, which).p) if (m is not None and m.training) else 0.0
            d = {"fp": rate(nxt_s, "feat_drop"), "ps": rate(s_layer, "attn_drop"), "pp": rate(p_layer, "attn_drop"),
                 "fp2": rate(nxt_p, "feat_drop")}
            for k, sk in (("fp", "fseed"), ("ps", "seed_s"), ("pp", "seed_p"), ("fp2", "fseed2")):
                d[sk] = _draw_seed() if d[k] > 0.0 else 0
            plan.append(d)
        return plan

    def _lspe_ok(self, g, x: torch.Tensor, h_p: torch.Tensor) -> bool:
        """The fused level needs two structure heads and one position head of the same width, Linear residuals and fusable
        activations in every level (all reference configs), fp32 rows, and a graph the kernels take (ops.lspe_level_supported)."""
        from . import nn as _nn
        from .nn import Identity, _act_code
        if not (FUSE_LSPE and x.is_cuda and _nn.SCORES_FROM_FT and x.dtype == torch.float32):
            return False
        for s_layer, p_layer in zip(self.gat_layers[:-1], self.pgnn_layers):
            if not (s_layer._num_heads == 2 and p_layer._num_heads == 1 and s_layer._out_feats == p_layer._out_feats
                    and isinstance(s_layer.res_fc, nn.Linear) and isinstance(p_layer.res_fc, nn.Linear)
                    and _act_code(s_layer.activation) is not None and _act_code(p_layer.activation) is not None
                    and not isinstance(s_layer.res_fc, Identity)):
                return False
        csc = g.csc(x.device)
        return all(ops.lspe_level_supported(csc, x, h_p, s_layer._out_feats) for s_layer in self.gat_layers[:-1])

    def _forward_lspe(self, g, x, xp, plan):
        """All hidden levels through ops.lspe_level: -> (input of the output layer (N, 3 D_last), h_p (N, D_last))."""
        from .nn import _act_code
        csc = g.csc(x.device)
        for l, (s_layer, p_layer) in enumerate(zip(self.gat_layers[:-1], self.pgnn_layers)):
            d = plan[l]
            w_s = ops.weight_cat(s_layer.fc.weight, s_layer.res_fc.weight, want_t=x.requires_grad)
            w_p = ops.weight_cat(p_layer.fc.weight, p_layer.res_fc.weight, want_t=xp.requires_grad)
            cfg = {"res_s": True, "res_p": True, "act": (_act_code(s_layer.activation), _act_code(p_layer.activation)),
                   "slope": (float(s_layer.negative_slope), float(p_layer.negative_slope)), "p_attn": (d["ps"], d["pp"]),
                   "seed_attn": (d["seed_s"], d["seed_p"]), "fp": d["fp"], "fseed": d["fseed"], "fp2": d["fp2"], "fseed2": d["fseed2"]}
            x, xp = ops.lspe_level(csc, x, xp, w_s, w_p, (s_layer.attn_l, s_layer.attn_r), (p_layer.attn_l, p_layer.attn_r),
                                   s_layer.bias, p_layer.bias, s_layer._out_feats, cfg)
        return x, xp

    def forward(self, g, classifier=None):
        """``classifier`` (extension): the ``*Net``'s ``gnn_out``; returns ``(h_s, h_p, classifier(h_s))`` with the
        classifier joined to the output layer's autograd node."""
        h_p, h_s = g.ndata["pos_enc"], g.ndata["fvs"]
        x, dropped = _data_cat(g, h_s, h_p), False
        with ops.prepared_weights(_prep_specs(zip(self.gat_layers[:-1], self.pgnn_layers), x.is_cuda, output=self.gat_layers[-1])):
            return self._forward(g, x, h_p, dropped, classifier)

    def _forward(self, g, x, h_p, dropped, classifier):
        plan = self._seed_plan() if x.is_cuda else None
        if plan is not None and self._lspe_ok(g, x, h_p):
            # one traversal per level (a graph with 0-in-degree nodes never gets here: the layers below raise DGLError for it)
            x, h_p = self._forward_lspe(g, x, _data_aligned(g, h_p), plan)
            dropped = True
            return self._output(g, x, h_p, dropped, classifier)
        for l, (s_layer, p_layer) in enumerate(zip(self.gat_layers[:-1], self.pgnn_layers)):
            sd = plan[l] if plan is not None else {}
            nxt = self.gat_layers[l + 1]
            w_p = p_layer._num_heads * p_layer._out_feats
            fplan = _fuse_plan(s_layer, nxt, x, w_p, seed=sd.get("fseed"))
            if fplan is not None:      # the structure rows go straight into the next layer's input, under its feature dropout
                buf, amax = s_layer(g, x, feat_dropped=dropped, fuse_out=fplan, attn_seed=sd.get("seed_s"))
            else:
                h_s = s_layer(g, x, feat_dropped=dropped, attn_seed=sd.get("seed_s")).flatten(1)
            xp, dropped_p = (_data_aligned(g, h_p), False) if l == 0 else \
                _drop_for(p_layer, h_p, seed=plan[l - 1]["fseed2"] if plan is not None else None)
            h_p = p_layer(g, xp, feat_dropped=dropped_p, attn_seed=sd.get("seed_p")).flatten(1)
            if fplan is not None:      # ... and the position rows complete it (same seed: one mask over the concatenation)
                total, p, seed, _ = fplan
                x, dropped = ops.fill_cols_dropout(buf, h_p, total - w_p, total, p, seed, amax), True
            else:
                x, dropped = _cat_for(nxt, h_s, h_p)
        return self._output(g, x, h_p, dropped, classifier)

    def _output(self, g, x, h_p, dropped, classifier):
        if classifier is not None:
            h_s, logits = self.gat_layers[-1](g, x, mean_heads=True, feat_dropped=dropped, classifier=classifier)
            return h_s, h_p, logits
        h_s = self.gat_layers[-1](g, x, mean_heads=True, feat_dropped=dropped)
        return h_s, h_p


class GATPSPGNNNL(_InferenceScoped, nn.Module):
    """"PENL" ablation: the static pos_enc is concatenated before every layer.
    Reference models.py:487-540."""

    def __init__(self, num_layers, in_dim, pos_in_dim, num_hiddens, out_ch, heads, activation, feat_drop, attn_drop,
                 negative_slope, residual, norm=False):
        super().__init__()
        self.num_layers, self.activation, self.out_ch, self.norm = num_layers, activation, out_ch, norm
        s_in = [in_dim + pos_in_dim] + [num_hiddens[l] * heads[l] + pos_in_dim for l in range(num_layers)]
        layers = []
        for l in range(num_layers):
            drop = (0.0, 0.0) if l == 0 else (feat_drop, attn_drop)
            layers.append(GATConv(s_in[l], num_hiddens[l], heads[l], drop[0], drop[1], negative_slope, residual,
                                  activation))
        layers.append(GATConv(s_in[num_layers], out_ch, heads[num_layers], 0.0, 0.0, negative_slope, residual,
                              activation))
        self.gat_layers = nn.ModuleList(layers)

    def reset_parameters(self):
        for layer in self.gat_layers:
            layer.reset_parameters()

    def forward(self, g, classifier=None):
        h_p, h_s = g.ndata["pos_enc"], g.ndata["fvs"]
        for l, layer in enumerate(self.gat_layers[:-1]):
            if l == 0:
                x, dropped = _data_cat(g, h_s, h_p), False
            else:
                x, dropped = _cat_for(layer, h_s, h_p)
            h_s = layer(g, x, feat_dropped=dropped).flatten(1)
        x, dropped = _cat_for(self.gat_layers[-1], h_s, h_p)
        if classifier is not None:
            h_s, logits = self.gat_layers[-1](g, x, mean_heads=True, feat_dropped=dropped, classifier=classifier)
            return h_s, h_p, logits
        h_s = self.gat_layers[-1](g, x, mean_heads=True, feat_dropped=dropped)
        return h_s, h_p


class SAGE(_InferenceScoped, nn.Module):
    def __init__(self, num_layers, in_dim, num_hiddens, out_ch, node_ks, node_sample_rate=0.3, activation=F.elu,
                 feat_drop=0.1, aggregator_type="pool", norm=None):
        super().__init__()
        self.num_layers, self.node_ks, self.node_sample_rate, self.out_ch = num_layers, node_ks, node_sample_rate, out_ch
        dims = [in_dim] + list(num_hiddens[:num_layers])
        layers = [SAGEConv(dims[l], dims[l + 1], aggregator_type=aggregator_type,
                           feat_drop=0.0 if l == 0 else feat_drop, activation=activation, norm=norm)
                  for l in range(num_layers)]
        layers.append(SAGEConv(dims[num_layers], out_ch, aggregator_type=aggregator_type))
        self.g_layers = nn.ModuleList(layers)

    def reset_parameters(self):
        for layer in self.g_layers:
            layer.reset_parameters()

    def forward_batch(self, blocks, x):
        h = x
        for layer, block in zip(self.g_layers, blocks):
            h = layer(block, h)
        return h

    def forward(self, g, classifier=None):
        """``classifier`` (extension): the ``*Net``'s ``gnn_out``; returns ``(h, classifier(h))``, the classifier joined to the
        (linear) output layer's product."""
        h = _data_in(g, g.ndata["fvs"], None)      # fp32 node data: a batch constant for the first layer's products
        with ops.prepared_weights(_linear_specs(self, h)):
            dropped = False
            for l, layer in enumerate(self.g_layers[:-1]):
                # the next layer's feature dropout rides in this layer's last product (its epilogue applies the hash mask)
                nxt = self.g_layers[l + 1]
                p = float(nxt.feat_drop.p) if nxt.training else 0.0
                fuse = FUSE_SAGE_DROP and 0.0 < p < 1.0 and h.is_cuda and h.dtype == torch.float32 and layer._out_feats % 4 == 0
                h = layer(g, h, feat_dropped=dropped, out_drop=(p, _draw_seed()) if fuse else None)
                dropped = fuse
            return self.g_layers[-1](g, h, classifier=classifier, feat_dropped=dropped)


# =================================================================================================
# *Net wrappers: [frozen CNN trunk] + GNN head + Linear classifier
# =================================================================================================
class _GraphNetBase(_InferenceScoped, nn.Module):
    """Shared plumbing of the reference's ``*Net`` classes: constructor bookkeeping, optional CNN
    trunk, ``gnn_out``, trainability switches and ``init``."""

    _head_attr = "gat"

    def _setup(self, *, n_layers, in_ch_list, base_ch_list, end_ch_list, checkpoint_layers, kernel_sizes, out_ch,
               padding_list, conv_strides, dropout, spatial_size, fv_dim, num_hiddens, node_embed_dim,
               norm_method, act_method, build_trunk):
        assert len(end_ch_list) == len(base_ch_list) == len(in_ch_list) == len(padding_list)
        self.fv_dim, self.dropout, self.n_layers, self.out_ch = fv_dim, dropout, n_layers, out_ch
        self.in_ch_list, self.base_ch_list, self.end_ch_list = in_ch_list, base_ch_list, end_ch_list
        self.checkpoint_layers, self.conv_strides = checkpoint_layers, conv_strides
        self.spatial_size, self.kernel_sizes = spatial_size, kernel_sizes
        self.num_hiddens, self.node_embed_dim = num_hiddens, node_embed_dim
        self._has_trunk = bool(build_trunk)
        if build_trunk:
            self._build_trunk(padding_list, norm_method, act_method)

    def _build_trunk(self, padding_list, norm_method, act_method):
        try:
            from parts import ConvBlock5d, act_wrapper      # the reference's own CNN blocks
        except ImportError as e:
            raise ImportError("build_trunk=True needs the reference's parts.py on sys.path "
                              "(the 3-D CNN stage is outside this package)") from e
        n, L = self.n_layers, self.n_layers
        self.ds_modules = nn.ModuleList([
            ConvBlock5d([self.in_ch_list[i], self.base_ch_list[i]], [self.base_ch_list[i], self.end_ch_list[i]],
                        self.checkpoint_layers[i], self.kernel_sizes[i], False, padding_list[i],
                        conv_strides=self.conv_strides[i], norm_method=norm_method, act_method=act_method,
                        dropout=self.dropout) for i in range(n)])
        self.bg = ConvBlock5d([self.in_ch_list[L], self.base_ch_list[L]], [self.base_ch_list[L], self.end_ch_list[L]],
                              self.checkpoint_layers[L], self.kernel_sizes[L], False, padding_list[L],
                              dropout=self.dropout, norm_method=norm_method, act_method=act_method)
        self.fc = nn.Sequential(
            nn.Conv3d(self.end_ch_list[L], self.end_ch_list[L], kernel_size=self.spatial_size, padding=0, stride=1),
            nn.Dropout(self.dropout), act_wrapper(act_method),
            nn.Conv3d(self.end_ch_list[L], self.fv_dim, kernel_size=1, padding=0, stride=1), act_wrapper(act_method))
        self.out = nn.Conv3d(self.fv_dim, out_channels=self.out_ch, kernel_size=1, padding=0, bias=True)

    # ---- trainability (reference models.py:1127-1139) -------------------------------------------
    @property
    def _head(self) -> nn.Module:
        return getattr(self, self._head_attr)

    def set_gcn_only(self):
        set_trainable(self, False)
        set_trainable(self._head, True)
        set_trainable(self.gnn_out, True)

    def set_cnn_only(self):
        set_trainable(self, False)
        if self._has_trunk:
            for m in (self.ds_modules, self.bg, self.fc, self.out):
                set_trainable(m, True)

    def set_all(self):
        set_trainable(self, True)

    def init(self, initializer=None):
        """reference models.py:1141-1146: generic initializer over all submodules, then the head's own
        reset_parameters and xavier-normal on gnn_out."""
        if initializer is not None:
            initializer.initialize(self)
        if hasattr(self._head, "reset_parameters"):
            self._head.reset_parameters()
        nn.init.xavier_normal_(self.gnn_out.weight, gain=nn.init.calculate_gain("linear"))
        nn.init.constant_(self.gnn_out.bias, 0.0)

    # ---- CNN stage (only with a trunk) ----------------------------------------------------------
    def _need_trunk(self):
        if not self._has_trunk:
            raise RuntimeError("this model was built without the CNN trunk (build_trunk=False)")

    def extract_feature(self, x):
        self._need_trunk()
        for ds in self.ds_modules:
            x = ds(x)
        return self.fc(self.bg(x))

    def forward_without_gnn(self, x):
        self._need_trunk()
        from torch.utils.checkpoint import checkpoint
        for idx, ds in enumerate(self.ds_modules):
            x = ds(x) if idx == 0 else checkpoint(ds, x)
        xbg = self.fc(self.bg(x))
        return xbg, self.out(xbg)


class GCNNet(_GraphNetBase):
    _head_attr = "gcn"

    def __init__(self, n_layers, num_gcn_layers, in_ch_list, base_ch_list, end_ch_list, checkpoint_layers,
                 kernel_sizes, out_ch, padding_list, conv_strides, dropout, spatial_size, fv_dim, num_hiddens,
                 node_embed_dim, norm_method="bn", act_method="relu", build_trunk=False):
        super().__init__()
        self._setup(n_layers=n_layers, in_ch_list=in_ch_list, base_ch_list=base_ch_list, end_ch_list=end_ch_list,
                    checkpoint_layers=checkpoint_layers, kernel_sizes=kernel_sizes, out_ch=out_ch,
                    padding_list=padding_list, conv_strides=conv_strides, dropout=dropout, spatial_size=spatial_size,
                    fv_dim=fv_dim, num_hiddens=num_hiddens, node_embed_dim=node_embed_dim, norm_method=norm_method,
                    act_method=act_method, build_trunk=build_trunk)
        self.gcn = GCN(num_layers=num_gcn_layers, in_dim=fv_dim, num_hiddens=num_hiddens, num_classes=node_embed_dim,
                       activation=F.elu)
        self.gnn_out = SkinnyLinear(node_embed_dim, out_ch)

    def forward(self, g):
        n_embed, n_out = self.gcn(g, classifier=self.gnn_out)
        return n_out, n_embed


class SAGENet(_GraphNetBase):
    _head_attr = "sage"

    def __init__(self, n_layers, num_layers, in_ch_list, base_ch_list, end_ch_list, checkpoint_layers,
                 kernel_sizes, out_ch, padding_list, conv_strides, dropout, feat_drop, spatial_size, fv_dim,
                 num_hiddens, node_embed_dim, node_ks, node_sample_rate, aggregator_type="pool",
                 norm_method="bn", act_method="relu", build_trunk=False):
        super().__init__()
        self._setup(n_layers=n_layers, in_ch_list=in_ch_list, base_ch_list=base_ch_list, end_ch_list=end_ch_list,
                    checkpoint_layers=checkpoint_layers, kernel_sizes=kernel_sizes, out_ch=out_ch,
                    padding_list=padding_list, conv_strides=conv_strides, dropout=dropout, spatial_size=spatial_size,
                    fv_dim=fv_dim, num_hiddens=num_hiddens, node_embed_dim=node_embed_dim, norm_method=norm_method,
                    act_method=act_method, build_trunk=build_trunk)
        self.node_sample_rate, self.node_ks = node_sample_rate, node_ks
        self.sage = SAGE(num_layers=num_layers, in_dim=fv_dim, num_hiddens=num_hiddens, out_ch=node_embed_dim,
                         activation=F.elu, feat_drop=feat_drop, node_ks=node_ks, aggregator_type=aggregator_type,
                         node_sample_rate=node_sample_rate)
        self.gnn_out = SkinnyLinear(node_embed_dim, out_ch)

    def forward(self, g):
        n_embed, n_out = self.sage(g, classifier=self.gnn_out)
        return n_out, n_embed

    def forward_batch(self, blocks, x):
        n_embed = self.sage.forward_batch(blocks, x)
        return self.gnn_out(n_embed), n_embed


class GATNet(_GraphNetBase):
    _head_attr = "gat"

    def __init__(self, n_layers, num_gat_layers, num_heads, num_out_heads, in_ch_list, base_ch_list, end_ch_list,
                 checkpoint_layers, kernel_sizes, out_ch, padding_list, conv_strides, dropout, feat_drop, attn_drop,
                 negative_slope, spatial_size, fv_dim, num_hiddens, node_embed_dim, res=True,
                 norm_method="bn", act_method="relu", build_trunk=False):
        super().__init__()
        self._setup(n_layers=n_layers, in_ch_list=in_ch_list, base_ch_list=base_ch_list, end_ch_list=end_ch_list,
                    checkpoint_layers=checkpoint_layers, kernel_sizes=kernel_sizes, out_ch=out_ch,
                    padding_list=padding_list, conv_strides=conv_strides, dropout=dropout, spatial_size=spatial_size,
                    fv_dim=fv_dim, num_hiddens=num_hiddens, node_embed_dim=node_embed_dim, norm_method=norm_method,
                    act_method=act_method, build_trunk=build_trunk)
        self.res = res
        heads = [num_heads] * num_gat_layers + [num_out_heads]
        self.gat = GAT(num_layers=num_gat_layers, in_dim=fv_dim, num_hiddens=num_hiddens, out_ch=node_embed_dim,
                       heads=heads, activation=F.elu, feat_drop=feat_drop, attn_drop=attn_drop,
                       negative_slope=negative_slope, residual=res)
        self.gnn_out = SkinnyLinear(node_embed_dim, out_ch)

    def forward(self, g):
        n_embed, n_out = self.gat(g, classifier=self.gnn_out)
        return n_out, n_embed

    def forward_emb(self, g):
        n_embed = self.gat(g)
        return n_embed, n_embed                      # sic: reference models.py:921-923 returns it twice

    def forward_batch(self, blocks, x):
        n_embed = self.gat.forward_batch(blocks, x)
        return self.gnn_out(n_embed), n_embed


class GINNet(_GraphNetBase):
    _head_attr = "gin"

    def __init__(self, n_layers, num_gin_layers, in_ch_list, base_ch_list, end_ch_list, checkpoint_layers,
                 kernel_sizes, out_ch, padding_list, conv_strides, dropout, spatial_size, fv_dim, num_hiddens,
                 node_embed_dim, norm_method="bn", act_method="relu", build_trunk=False):
        super().__init__()
        self._setup(n_layers=n_layers, in_ch_list=in_ch_list, base_ch_list=base_ch_list, end_ch_list=end_ch_list,
                    checkpoint_layers=checkpoint_layers, kernel_sizes=kernel_sizes, out_ch=out_ch,
                    padding_list=padding_list, conv_strides=conv_strides, dropout=dropout, spatial_size=spatial_size,
                    fv_dim=fv_dim, num_hiddens=num_hiddens, node_embed_dim=node_embed_dim, norm_method=norm_method,
                    act_method=act_method, build_trunk=build_trunk)
        self.gin = GIN(num_layers=num_gin_layers, in_dim=fv_dim, num_hiddens=num_hiddens, out_ch=node_embed_dim)
        self.gnn_out = SkinnyLinear(node_embed_dim, out_ch)
        self.gnn_lobe_out = nn.Linear(node_embed_dim, 6)      # auxiliary heads (reference models.py:988-989)
        self.gnn_lung_out = nn.Linear(node_embed_dim, 3)

    def set_gcn_only(self):
        super().set_gcn_only()
        set_trainable(self.gnn_lobe_out, True)
        set_trainable(self.gnn_lung_out, True)

    def forward(self, g):
        n_embed, n_out = self.gin(g, classifier=self.gnn_out)
        return n_out, n_embed

    def forward_batch(self, blocks, x):
        n_embed = self.gin.forward_batch(blocks, x)
        return self.gnn_out(n_embed), n_embed

    def forward_all(self, g):
        n_embed = self.gin(g)
        return self.gnn_out(n_embed), self.gnn_lobe_out(n_embed), self.gnn_lung_out(n_embed), n_embed


class GATPositionSPGNNNet(_GraphNetBase):
    _head_attr = "gat"

    def __init__(self, n_layers, num_gat_layers, num_heads, num_out_heads, in_ch_list, base_ch_list, end_ch_list,
                 checkpoint_layers, kernel_sizes, out_ch, padding_list, conv_strides, dropout, feat_drop, attn_drop,
                 negative_slope, spatial_size, fv_dim, num_hiddens, pos_hiddens, num_pos_heads, node_embed_dim,
                 pos_enc_dim, encodng_merge="cat", norm=False, res=True, norm_method="bn", act_method="relu",
                 p_act="tahn", mode="PEL", build_trunk=False):
        super().__init__()
        self._setup(n_layers=n_layers, in_ch_list=in_ch_list, base_ch_list=base_ch_list, end_ch_list=end_ch_list,
                    checkpoint_layers=checkpoint_layers, kernel_sizes=kernel_sizes, out_ch=out_ch,
                    padding_list=padding_list, conv_strides=conv_strides, dropout=dropout, spatial_size=spatial_size,
                    fv_dim=fv_dim, num_hiddens=num_hiddens, node_embed_dim=node_embed_dim, norm_method=norm_method,
                    act_method=act_method, build_trunk=build_trunk)
        self.pos_enc_dim, self.num_pos_heads, self.encodng_merge = pos_enc_dim, num_pos_heads, encodng_merge
        self.res, self.mode, self.pos_hiddens = res, mode, pos_hiddens
        self.p_act = torch.tanh if p_act == "tahn" else F.elu          # "tahn" [sic], reference models.py:1067
        heads = [num_heads] * num_gat_layers + [num_out_heads]
        pos_heads = [num_pos_heads] * (num_gat_layers + 1)
        if mode == "PEL":
            self.gat = GATPSPGNN(num_layers=num_gat_layers, in_dim=fv_dim, pos_in_dim=pos_enc_dim,
                                 num_hiddens=num_hiddens, pos_hiddens=pos_hiddens, pos_heads=pos_heads,
                                 out_ch=node_embed_dim, heads=heads, activation=F.elu, feat_drop=feat_drop,
                                 attn_drop=attn_drop, negative_slope=negative_slope, residual=res, norm=norm,
                                 p_activation=self.p_act)
        elif mode == "PENL":
            self.gat = GATPSPGNNNL(num_layers=num_gat_layers, in_dim=fv_dim, pos_in_dim=pos_enc_dim,
                                   num_hiddens=num_hiddens, out_ch=node_embed_dim, heads=heads, activation=F.elu,
                                   feat_drop=feat_drop, attn_drop=attn_drop, negative_slope=negative_slope,
                                   residual=res, norm=norm)
        else:
            raise ValueError(f"unknown mode {mode!r} (PEL or PENL)")
        self.gnn_out = SkinnyLinear(node_embed_dim, out_ch)

    def forward(self, g):
        n_embed, n_p_embed, n_out = self.gat(g, classifier=self.gnn_out)
        return n_out, n_embed, n_p_embed

    def forward_emb(self, g):
        return self.gat(g)


def build_model(model_cfg: dict, **extra) -> nn.Module:
    """``cls(**settings.MODEL)`` with the dotted ``method`` popped (reference job_runner.py:217-220);
    class names that the reference's configs misspell are mapped to the evident intent
    (``models.GATPositionLSPENet`` -> GATPositionSPGNNNet(mode="PENL"))."""
    cfg = dict(model_cfg)
    name = cfg.pop("method").split(".")[-1]
    if name == "GATPositionLSPENet":
        name, cfg = "GATPositionSPGNNNet", {**cfg, "mode": "PENL"}
    cls = globals()[name]
    return cls(**cfg, **extra)
