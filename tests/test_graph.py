"""Host-side graph construction and positional encoding: bit-exact against the networkx
restatement of the reference's rule (oracle/graph_rule_nx.py; reference job_runner.py:1319-1344,
1712-1801, 1882)."""
import networkx as nx
import numpy as np
import pytest
import torch

from oracle import graph_rule_nx as R
from spgnn_amd import graph as G
from spgnn_amd import posenc, synthetic


def _trees(ns, seed=0):
    rng = np.random.default_rng(seed)
    return [synthetic.random_tree_adj(n, rng) for n in ns]


@pytest.mark.parametrize("n", [1, 2, 3, 21, 128, 177])
def test_edge_rule_matches_networkx(n):
    adj = _trees([n], seed=n)[0]
    u, v = G.edges_from_adj(adj)
    ru, rv = R.edges_spgnn(adj)
    assert np.array_equal(u, ru) and np.array_equal(v, rv)
    gu, gv = R.edges_gcn(adj)               # the GCN/GAT runner's path yields the same list
    assert np.array_equal(u, gu) and np.array_equal(v, gv)
    assert u.shape[0] == 3 * n - 2          # E = 3n - 2


def test_dglgraph_from_networkx_then_add_self_loops():
    adj = _trees([40], seed=3)[0]
    Gd = nx.DiGraph(adj)
    g = G.DGLGraph(Gd)
    g = G.remove_self_loop(g)
    g.add_edges(g.nodes(), g.nodes())
    u, v = R.edges_spgnn(adj)
    assert np.array_equal(g._src, u) and np.array_equal(g._dst, v)
    Gu = nx.Graph(adj); Gu.remove_edges_from(nx.selfloop_edges(Gu))
    g2 = G.DGLGraph(Gu); g2.add_edges(g2.nodes(), g2.nodes())
    assert np.array_equal(g2._src, u) and np.array_equal(g2._dst, v)


def test_batch_offsets_and_csc_bit_exact():
    adjs = _trees([5, 1, 33, 150, 2], seed=5)
    graphs = [G.graph_from_adj(a) for a in adjs]
    for i, g in enumerate(graphs):
        g.ndata["fvs"] = torch.full((g.number_of_nodes(), 3), float(i))
    bg = G.batch(graphs)
    rs, rd = R.batch_edges([R.edges_spgnn(a) for a in adjs], [a.shape[0] for a in adjs])
    assert np.array_equal(bg._src, rs) and np.array_equal(bg._dst, rd)
    assert bg.batch_size == 5 and bg.batch_num_nodes_list == [5, 1, 33, 150, 2]
    assert bg.batch_num_edges_list == [3 * n - 2 for n in (5, 1, 33, 150, 2)]
    assert torch.equal(bg.ndata["fvs"][:, 0], torch.cat([torch.full((a.shape[0],), float(i)) for i, a in enumerate(adjs)]))
    csc = G.build_csc_numpy(bg._src, bg._dst, bg.number_of_nodes())
    indptr, indices, eid = R.csc_stable(rs, rd, bg.number_of_nodes())
    assert np.array_equal(csc["indptr"], indptr) and np.array_equal(csc["indices"], indices)
    assert np.array_equal(csc["eid"], eid)
    # CSR side + slot map: edge k of out-list of u is (u -> out_indices[k]) stored at CSC slot out_pos[k]
    for u in range(bg.number_of_nodes()):
        for k in range(csc["out_indptr"][u], csc["out_indptr"][u + 1]):
            slot = csc["out_pos"][k]
            v = csc["out_indices"][k]
            assert csc["indices"][slot] == u and csc["indptr"][v] <= slot < csc["indptr"][v + 1]
    # in-neighbour order of v: ascending tree neighbours, then v itself
    for v in range(bg.number_of_nodes()):
        nb = csc["indices"][csc["indptr"][v]:csc["indptr"][v + 1]]
        assert nb[-1] == v and np.all(np.diff(nb[:-1]) > 0)


def test_unbatch_round_trip_and_edge_cases():
    adjs = _trees([7, 30, 4], seed=6)
    graphs = [G.graph_from_adj(a) for a in adjs]
    for g in graphs:
        g.ndata["y"] = torch.arange(g.number_of_nodes())
    parts = G.unbatch(G.batch(graphs))
    for a, b in zip(graphs, parts):
        assert np.array_equal(a._src, b._src) and np.array_equal(a._dst, b._dst)
        assert torch.equal(a.ndata["y"], b.ndata["y"])
    with pytest.raises(ValueError):
        G.batch([])
    with pytest.raises(ValueError):
        graphs[0].ndata["bad"] = torch.zeros(3)
    with pytest.raises(ValueError):
        graphs[0].add_edges([0], [999])
    e = G.TreeGraph(None, 0)                     # empty graph
    assert e.number_of_nodes() == 0 and e.number_of_edges() == 0
    assert G.build_csc_numpy(e._src, e._dst, 0)["indptr"].tolist() == [0]
    assert int(graphs[1].in_degrees().sum()) == graphs[1].number_of_edges()


def test_to_networkx_and_adjacency():
    adj = _trees([12], seed=8)[0]
    g = G.graph_from_adj(adj, add_self_loops=False)
    Gx = G.to_networkx(g)
    assert sorted(Gx.edges()) == sorted(zip(g._src.tolist(), g._dst.tolist()))
    A = g.adjacency_matrix().to_dense().numpy()
    ref = adj.astype(np.float32) - np.eye(12, dtype=np.float32)
    assert np.array_equal(A, ref)
    assert np.array_equal(g.adjacency_matrix(scipy_fmt="csr").toarray(), ref)


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_positional_encoding_matches_networkx(seed):
    rng = np.random.default_rng(100 + seed)
    s = synthetic.synthetic_tree(int(rng.integers(30, 160)), rng)
    a1 = posenc.anchors_from_cnn_prediction(s["fvs_out"], s["adj"], 39)
    a2 = R.anchors_from_cnn_prediction(s["fvs_out"], s["adj"], 39)
    assert a1 == a2 and len(a1) == 39
    pe1, d1 = posenc.distance_pos_enc(s["adj"], a1)
    pe2, d2 = R.distance_pos_enc(s["adj"], a2)
    assert d1 == d2
    assert np.array_equal(pe1, pe2) and pe1.dtype == np.float32 and pe1.shape == (s["adj"].shape[0], 39)
    assert posenc.anchors_from_cnn_prediction(s["fvs_out"], s["adj"], 21) == a1[:21]


def test_synthetic_workload_shape():
    samples = synthetic.synthetic_trees(4, rank=0)
    again = synthetic.synthetic_trees(4, rank=0)
    other = synthetic.synthetic_trees(4, rank=1)
    for a, b in zip(samples, again):
        assert all(np.array_equal(a[k], b[k]) for k in a)              # seeded, reproducible
    assert not np.array_equal(samples[0]["fvs"][:50], other[0]["fvs"][:50])
    for s in samples:
        n = s["adj"].shape[0]
        assert 120 <= n <= 180 and s["fvs"].shape == (n, 1024) and s["fvs_out"].shape == (n, 22)
        assert (s["fvs"] >= 0).all() and s["adj"].dtype == np.uint8
        up = np.triu(s["adj"], 1)
        assert (up.sum(0)[1:] == 1).all() and up.sum(1).max() <= 3      # one parent < child, fan-out <= 3
        assert sorted(s["labels"][s["labels"] > 0].tolist()) == list(range(1, 22))
    bg = synthetic.batch_from_samples(samples)
    assert bg.ndata["pos_enc"].shape == (bg.number_of_nodes(), 39)
    assert bg.number_of_edges() == 3 * bg.number_of_nodes() - 2 * 4


def test_tree_downstream_graph_mode_follows_the_reference_rule():
    """GRAPH_MODE == "tree_downstream" (reference job_runner.py:1334-1336: nx.DiGraph(np.triu(adj)), self loops removed, then
    g.add_edges(nodes, nodes)): parent -> child edges only.  Bit-exact against the networkx restatement, for symmetric trees
    and for an adjacency matrix that is already upper triangular (job_runner.py:1329-1332: the same directed graph in either
    mode); no config of the reference selects it, the boundary still has it."""
    from oracle import graph_rule_nx as R
    rng = np.random.default_rng(3)
    for n in (1, 2, 7, 33, 150):
        adj = synthetic.random_tree_adj(n, rng)
        u, v = G.edges_from_adj(adj, graph_mode="tree_downstream")
        ru, rv = R.edges_gcn(adj, "tree_downstream")
        assert np.array_equal(u, ru) and np.array_equal(v, rv)
        assert u.shape[0] == 2 * n - 1 and (u <= v).all()                  # n - 1 parent -> child edges + n self loops
        tri = np.triu(adj)
        for mode in G.GRAPH_MODES:
            u2, v2 = G.edges_from_adj(tri, graph_mode=mode)
            ru2, rv2 = R.edges_gcn(tri, mode)
            assert np.array_equal(u2, ru2) and np.array_equal(v2, rv2) and np.array_equal(u2, u)
    with pytest.raises(ValueError):
        G.edges_from_adj(np.eye(3), graph_mode="upstream")
    g = G.graph_from_adj(synthetic.random_tree_adj(20, rng), graph_mode="tree_downstream")
    assert g.number_of_edges() == 39
