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

__all__ = ["anchors_from_cnn_prediction", "add_distal_leafs", "distance_pos_enc", "bfs_distances",
           "distance_pos_enc_device", "anchors_device"]


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


def anchors_device(g, fvs_out, pos_enc_dim: int = 39):
    """Device version of :func:`anchors_from_cnn_prediction` for a whole batched graph (reference job_runner.py:1727-1757,
    1712-1725): softmax of the CNN logits (torch, on the device, as the reference forms it), then one HIP call for the
    greedy per-label argmax and the distal leaves of every tree (spgnn_tree_anchors).  Returns (B, A) int32 GLOBAL node
    ids on the device - what :func:`distance_pos_enc_device` takes.  No per-tree host work."""
    import torch
    from . import _capi
    if pos_enc_dim not in (39, 21):
        raise NotImplementedError(f"pos enc dim : {pos_enc_dim}!")
    csc = g.csc()
    dev = csc.indptr.device
    if dev.type != "cuda":
        raise RuntimeError("anchors_device needs the graph on a ROCm device (use anchors_from_cnn_prediction on the host)")
    nn = np.asarray(g.batch_num_nodes_list, dtype=np.int64)
    if len(nn) and nn.min() < 21:
        raise ValueError("anchor rule needs at least 21 nodes per tree")
    B = len(nn)
    num_labels, num_distal = 21, (18 if pos_enc_dim == 39 else 0)
    tree_ptr = torch.from_numpy(np.concatenate([[0], np.cumsum(nn)])).to(dev)
    prob = torch.softmax(fvs_out.to(device=dev, dtype=torch.float32), dim=1).contiguous()
    anchors = torch.empty((B, num_labels + num_distal), dtype=torch.int32, device=dev)
    lib = _capi.load()
    nmax = int(nn.max()) if B else 0
    ws = torch.empty((int(lib.spgnn_tree_anchors_workspace(B, num_distal, nmax)),), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        _capi.check(lib.spgnn_tree_anchors(prob.data_ptr(), prob.stride(0), csc.out_indptr.data_ptr(), csc.out_indices.data_ptr(),
                                           tree_ptr.data_ptr(), B, int(nn.sum()), nmax, num_labels, num_distal,
                                           anchors.data_ptr(), ws.data_ptr(), torch.cuda.current_stream(dev).cuda_stream),
                    "spgnn_tree_anchors")
    return anchors


def distance_pos_enc_device(g, anchors_per_tree):
    """Device version of :func:`distance_pos_enc` for a whole batched graph.

    ``g``: batched TreeGraph on a ROCm device; ``anchors_per_tree``: list (one per tree) of LOCAL anchor ids, or the
    (B, A) int32 device tensor of GLOBAL ids :func:`anchors_device` returns.
    Returns ``(pos_enc (N, A) float32 on the device, diameters (B,) int32)``; bit-identical to the host/networkx
    path.  One HIP workgroup per tree, BFS distance arrays in LDS (spgnn_tree_distance_encoding)."""
    import torch
    from . import _capi
    csc = g.csc()
    dev = csc.indptr.device
    if dev.type != "cuda":
        raise RuntimeError("distance_pos_enc_device needs the graph on a ROCm device (use distance_pos_enc on the host)")
    nn = np.asarray(g.batch_num_nodes_list, dtype=np.int64)
    tree_ptr = np.concatenate([[0], np.cumsum(nn)])
    tree_ptr_d = torch.from_numpy(tree_ptr).to(dev)
    if torch.is_tensor(anchors_per_tree):
        anc_d = anchors_per_tree.to(device=dev, dtype=torch.int32).contiguous()
        B, A = anc_d.shape
        assert B == len(nn)
    else:
        B, A = len(nn), len(anchors_per_tree[0])
        anc = np.asarray(anchors_per_tree, dtype=np.int64).reshape(B, A) + tree_ptr[:-1, None]
        anc_d = torch.from_numpy(anc.astype(np.int32)).to(dev)
    # rows padded to 16 bytes (39 -> stride 40) so the encoding feeds the vector/MFMA kernels without a copy
    pe = torch.zeros((int(tree_ptr[-1]), (A + 3) // 4 * 4), dtype=torch.float32, device=dev)[:, :A]
    diam = torch.empty((B,), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        _capi.check(_capi.load().spgnn_tree_distance_encoding(
            csc.out_indptr.data_ptr(), csc.out_indices.data_ptr(), tree_ptr_d.data_ptr(), anc_d.data_ptr(), A,
            pe.data_ptr(), pe.stride(0), diam.data_ptr(), B, int(nn.max()) if B else 0,
            torch.cuda.current_stream(dev).cuda_stream), "spgnn_tree_distance_encoding")
    return pe, diam
