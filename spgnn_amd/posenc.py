"""Distance positional encodings of an airway tree (host side).

Restates reference job_runner.py:1712-1777 (``add_distal_leafs``,
``get_anchors_from_cnn_prediction``, ``generate_distant_pos_enc``) without networkx:
anchors are picked from the CNN's softmax, 18 "distal leaf" anchors are added, and each
node's encoding is its hop distance to every anchor divided by the tree diameter.
"""
from __future__ import annotations

from collections import deque
from typing import List, Tuple

import numpy as np

__all__ = ["anchors_from_cnn_prediction", "add_distal_leafs", "distance_pos_enc", "bfs_distances"]


def _softmax(x: np.ndarray) -> np.ndarray:
    z = x - x.max(axis=1, keepdims=True)
    e = np.exp(z)
    return e / e.sum(axis=1, keepdims=True)


def _children_lists(adj: np.ndarray) -> List[np.ndarray]:
    """Downstream DAG = strict upper triangle of adj (reference job_runner.py:1713-1715)."""
    up = np.triu(np.asarray(adj), k=1)
    return [np.nonzero(up[i])[0] for i in range(up.shape[0])]


def add_distal_leafs(anchors: List[int], adj: np.ndarray) -> List[int]:
    """For each anchor, its farthest descendant leaf in the downstream DAG (the anchor itself
    if it has no descendant leaf). Reference job_runner.py:1712-1725.

    Ties on distance: the reference sorts ``dict`` items built from a Python ``set`` of
    descendants (insertion = BFS order) with a stable sort and takes the last one; the same
    container sequence is replayed here so ties resolve identically under CPython.
    """
    ch = _children_lists(adj)
    out = []
    for a in anchors:
        desc = set()
        dist = {a: 0}
        q = deque([a])
        while q:                           # nx.bfs_edges order: children ascending
            p = q.popleft()
            for c in ch[p]:
                c = int(c)
                if c not in dist:
                    dist[c] = dist[p] + 1
                    desc.add(c)
                    q.append(c)
        leafs = {n: dist[n] for n in desc if len(ch[n]) == 0}
        if not leafs:
            out.append(int(a))
        else:
            out.append(sorted(leafs.items(), key=lambda x: x[1])[-1][0])
    return out


def anchors_from_cnn_prediction(fvs_out: np.ndarray, adj: np.ndarray, pos_enc_dim: int = 39) -> List[int]:
    """21 anchors = greedy per-label argmax of softmax(fvs_out) over not-yet-taken nodes
    (labels 1..21), plus, for pos_enc_dim == 39, the distal leaf of each of the first 18.
    Reference job_runner.py:1727-1757 (needs n >= 21)."""
    p = _softmax(np.asarray(fvs_out, dtype=np.float32))
    n = p.shape[0]
    if n < 21:
        raise ValueError("anchor rule needs at least 21 nodes")
    mask = np.ones(n, dtype=np.float64)
    anchors = []
    for label in range(1, 22):
        idx = int(np.argmax(p[:, label] * mask))
        mask[idx] = 0.0
        anchors.append(idx)
    if pos_enc_dim == 39:
        extra = add_distal_leafs(anchors[:-3], adj)
    elif pos_enc_dim == 21:
        extra = []
    else:
        raise NotImplementedError(f"pos enc dim : {pos_enc_dim}!")
    return anchors + extra


def bfs_distances(adj: np.ndarray, sources) -> np.ndarray:
    """Hop distances (len(sources), n) on the undirected tree, ignoring self loops."""
    a = np.asarray(adj) != 0
    n = a.shape[0]
    nbrs = [np.nonzero(a[i] | a[:, i])[0] for i in range(n)]
    out = np.full((len(sources), n), -1, dtype=np.int64)
    for k, s in enumerate(sources):
        d = out[k]
        d[s] = 0
        q = deque([int(s)])
        while q:
            u = q.popleft()
            for v in nbrs[u]:
                if d[v] < 0:
                    d[v] = d[u] + 1
                    q.append(int(v))
    if (out < 0).any():
        raise ValueError("graph is not connected")
    return out


def _tree_diameter(adj: np.ndarray) -> int:
    d0 = bfs_distances(adj, [0])[0]
    far = int(np.argmax(d0))
    return int(bfs_distances(adj, [far])[0].max())   # double sweep is exact on trees


def distance_pos_enc(adj: np.ndarray, anchors) -> Tuple[np.ndarray, int]:
    """(n, len(anchors)) float32 = hop distance to each anchor / diameter
    (reference job_runner.py:1759-1771)."""
    diameter = _tree_diameter(adj)
    d = bfs_distances(adj, list(anchors)).T.astype(np.float64)
    return (d / float(diameter)).astype(np.float32), diameter
