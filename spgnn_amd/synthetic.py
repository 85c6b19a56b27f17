"""Synthetic "random fan-out airway tree" workload.

Every sample follows the cached-embedding schema the reference's CNN stage writes
(reference job_runner.py:796-803): ``adj`` (n, n) uint8 = I + A + A^T, ``fvs`` (n, 1024),
``fvs_out`` (n, 22), ``labels`` (n,) uint8.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import numpy as np
import torch

from . import graph as G
from .posenc import anchors_from_cnn_prediction, distance_pos_enc

__all__ = ["random_tree_parents", "random_tree_adj", "synthetic_tree", "synthetic_trees",
           "batch_from_samples", "make_batch"]


def random_tree_parents(n: int, rng: np.random.Generator, max_children: int = 3, window: int = 32) -> np.ndarray:
    """parent[i] < i; at most ``max_children`` children per node; parents drawn from the last
    ``window`` eligible nodes so depth stays airway-like (Appendix D)."""
    parent = np.full(n, -1, dtype=np.int64)
    nchild = np.zeros(n, dtype=np.int64)
    eligible: List[int] = [0]
    for i in range(1, n):
        lo = max(0, len(eligible) - window)
        k = int(rng.integers(lo, len(eligible)))
        p = eligible[k]
        parent[i] = p
        nchild[p] += 1
        if nchild[p] >= max_children:
            eligible.pop(k)
        eligible.append(i)
    return parent


def random_tree_adj(n: int, rng: np.random.Generator, max_children: int = 3) -> np.ndarray:
    parent = random_tree_parents(n, rng, max_children)
    adj = np.eye(n, dtype=np.uint8)
    c = np.arange(1, n)
    adj[parent[1:], c] = 1
    adj[c, parent[1:]] = 1
    return adj


def synthetic_tree(n: int, rng: np.random.Generator, fv_dim: int = 1024, n_class: int = 22) -> Dict[str, np.ndarray]:
    adj = random_tree_adj(n, rng)
    fvs = np.maximum(rng.standard_normal((n, fv_dim), dtype=np.float32), 0.0)   # post-ReLU CNN features
    fvs_out = rng.standard_normal((n, n_class), dtype=np.float32)
    labels = np.zeros(n, dtype=np.uint8)
    k = min(n_class - 1, n)
    labels[rng.choice(n, size=k, replace=False)] = np.arange(1, k + 1, dtype=np.uint8)
    return dict(adj=adj, fvs=fvs, fvs_out=fvs_out, labels=labels)


def synthetic_trees(num_trees: int, rank: int = 0, n_lo: int = 120, n_hi: int = 180, fixed_n: Optional[int] = None,
                    fv_dim: int = 1024, n_class: int = 22, base_seed: int = 1234) -> List[Dict[str, np.ndarray]]:
    """Tree i of rank r uses ``np.random.default_rng(base_seed + 1000*r + i)``."""
    out = []
    for i in range(num_trees):
        rng = np.random.default_rng(base_seed + 1000 * rank + i)
        n = fixed_n if fixed_n is not None else int(rng.integers(n_lo, n_hi + 1))
        out.append(synthetic_tree(n, rng, fv_dim, n_class))
    return out


def batch_from_samples(samples: List[Dict[str, np.ndarray]], device="cpu", pos_enc_dim: Optional[int] = 39,
                       dtype=torch.float32, device_posenc: Optional[bool] = None) -> G.TreeGraph:
    """List of schema dicts -> one batched graph with ndata fvs / fvs_out / y [/ pos_enc / p],
    the host sequence of reference job_runner.py:1872-1885.  On a ROCm device the distance encoding is
    computed by the HIP kernel (one workgroup per tree) instead of per-tree host BFS (``device_posenc``)."""
    dev = torch.device(device)
    if device_posenc is None:
        device_posenc = dev.type == "cuda"
    graphs, anchors = [], []
    for s in samples:
        g = G.graph_from_adj(s["adj"], device="cpu", add_self_loops=True)
        g.ndata["fvs"] = torch.from_numpy(np.ascontiguousarray(s["fvs"])).to(dtype)
        g.ndata["fvs_out"] = torch.from_numpy(np.ascontiguousarray(s["fvs_out"])).to(dtype)
        g.ndata["y"] = torch.from_numpy(s["labels"].astype(np.int64))
        if pos_enc_dim:
            anc = anchors_from_cnn_prediction(s["fvs_out"], s["adj"], pos_enc_dim)
            anchors.append(anc)
            if not device_posenc:
                pe, _ = distance_pos_enc(s["adj"], anc)
                g.ndata["pos_enc"] = torch.from_numpy(pe).to(dtype)
                g.ndata["p"] = torch.from_numpy(pe).to(dtype)
        graphs.append(g)
    bg = G.batch(graphs).to(device)
    if pos_enc_dim and device_posenc:
        from .posenc import distance_pos_enc_device
        pe, _ = distance_pos_enc_device(bg, anchors)
        bg.ndata["pos_enc"] = pe if dtype == torch.float32 else pe.to(dtype)
        bg.ndata["p"] = bg.ndata["pos_enc"]
    return bg


def make_batch(num_trees: int, rank: int = 0, device="cpu", pos_enc_dim: Optional[int] = 39, **kw) -> G.TreeGraph:
    return batch_from_samples(synthetic_trees(num_trees, rank, **kw), device, pos_enc_dim)
