"""ORACLE (test infrastructure) — the reference's host-side graph construction and distance
positional encoding, restated with networkx exactly as the reference drives it. Integer
outputs: the product (spgnn_amd.graph / spgnn_amd.posenc) must match bit-exactly.

PARITY UNPINNED for the DGL half: ``DGLGraph(nx_graph)`` and ``dgl.batch`` are restated from
DGL's documented behaviour (edges taken in networkx iteration order; batch = id offsets by
running sums), DGL being absent here. The networkx half is the real library.
"""
from __future__ import annotations

import networkx as nx
import numpy as np


def edges_spgnn(adj: np.ndarray):
    """reference job_runner.py:1779-1800: nx.DiGraph(adj) -> DGLGraph -> remove_self_loop ->
    add_edges(nodes, nodes)."""
    G = nx.DiGraph(np.asarray(adj))
    e = [(u, v) for u, v in G.edges() if u != v]
    n = G.number_of_nodes()
    e += [(i, i) for i in range(n)]
    e = np.asarray(e, dtype=np.int64).reshape(-1, 2)
    return e[:, 0], e[:, 1]


def edges_gcn(adj: np.ndarray, graph_mode: str = "all_connected"):
    """reference job_runner.py:1319-1344 (GCNTrain.from_adj_to_graph)."""
    adj_np = np.asarray(adj)
    upper = np.triu(adj_np)
    if (upper - adj_np).sum() == 0:
        G = nx.DiGraph(adj_np)
    elif graph_mode == "tree_downstream":
        G = nx.DiGraph(upper)
    else:
        G = nx.Graph(adj_np)
    G.remove_edges_from(nx.selfloop_edges(G))
    D = G if G.is_directed() else G.to_directed()
    e = sorted(D.edges()) if not G.is_directed() else list(D.edges())
    n = G.number_of_nodes()
    e = list(e) + [(i, i) for i in range(n)]
    e = np.asarray(e, dtype=np.int64).reshape(-1, 2)
    return e[:, 0], e[:, 1]


def batch_edges(edge_lists, num_nodes):
    """dgl.batch: offsets by running node counts, concatenation in order."""
    srcs, dsts, off = [], [], 0
    for (s, d), n in zip(edge_lists, num_nodes):
        srcs.append(s + off); dsts.append(d + off); off += n
    return np.concatenate(srcs), np.concatenate(dsts)


def csc_stable(src, dst, n):
    """in-neighbour lists in ascending edge id (stable COO->CSC)."""
    indptr = [0]
    indices, eid = [], []
    buckets = [[] for _ in range(n)]
    for k, (s, d) in enumerate(zip(src.tolist(), dst.tolist())):
        buckets[d].append((s, k))
    for b in buckets:
        for s, k in b:
            indices.append(s); eid.append(k)
        indptr.append(len(indices))
    return np.asarray(indptr, np.int32), np.asarray(indices, np.int32), np.asarray(eid, np.int32)


# ---- positional encoding (reference job_runner.py:1712-1777) ---------------------------------
def add_distal_leafs(anchors, adj_np):
    upper = np.triu(adj_np)
    G = nx.DiGraph(upper)
    G.remove_edges_from(nx.selfloop_edges(G))
    adding = []
    for anchor in anchors:
        leafs = {n: nx.shortest_path_length(G, anchor, n)
                 for n in nx.descendants(G, anchor) if G.out_degree(n) == 0}
        if len(leafs) == 0:
            adding.append(anchor)
        else:
            adding.append(sorted(leafs.items(), key=lambda x: x[1])[-1][0])
    return adding


def anchors_from_cnn_prediction(fvs_out, adj_np, pos_enc_dim=39):
    x = np.asarray(fvs_out, dtype=np.float32)
    z = x - x.max(axis=1, keepdims=True)
    p = np.exp(z); p = p / p.sum(axis=1, keepdims=True)
    return anchors_from_probabilities(p, adj_np, pos_enc_dim)


def anchors_from_probabilities(p, adj_np, pos_enc_dim=39):
    """reference job_runner.py:1727-1757 from the softmax onwards (the reference forms ``F.softmax(fvs_out)`` with torch
    on its device and continues on the host with exactly these lines)."""
    p = np.asarray(p)
    mask = np.ones(p.shape[0]) * 1.0
    anchors = []
    for label in range(1, 22):
        index = int(np.argmax(p[:, label] * mask))
        mask[index] = 0.0
        anchors.append(index)
    extra = add_distal_leafs(anchors[:-3], np.asarray(adj_np)) if pos_enc_dim == 39 else []
    return anchors + extra


def distance_pos_enc(adj_np, anchors):
    a = np.asarray(adj_np).copy()
    np.fill_diagonal(a, 0)
    G = nx.DiGraph(a)                                   # both directions present: adj symmetric
    dist = dict(nx.all_pairs_shortest_path_length(G))
    diameter = nx.algorithms.distance_measures.diameter(G)
    n = a.shape[0]
    pe = np.empty((n, len(anchors)), dtype=np.float32)
    for t in range(n):
        pe[t] = np.asarray([dist[t][x] / float(diameter) for x in anchors])
    return pe, diameter
