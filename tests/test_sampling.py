"""Neighbour sampling and message-flow blocks (reference job_runner.py:1484-1506).

Host-side index work: everything here runs on the CPU.  DGL's random stream is not reproducible outside DGL, so the
checks are the properties ``dgl.sampling.sample_neighbors`` / ``dgl.to_block`` / ``NodeDataLoader`` guarantee."""
import numpy as np
import pytest
import torch

from spgnn_amd import dataloading, dgl_compat as dgl
from spgnn_amd.graph import Block, TreeGraph, to_block

from tests.util import tree_batch_edges


def _graph(ns=(40, 23, 57), seed=0):
    src, dst, n = tree_batch_edges(ns, seed)
    g = TreeGraph((src, dst), n)
    g.ndata["fvs"] = torch.arange(n * 3, dtype=torch.float32).view(n, 3)
    g.ndata["y"] = torch.arange(n) % 5
    return g


def _in_edges(g, v):
    m = g._dst == v
    return set(zip(g._src[m].tolist(), np.nonzero(m)[0].tolist()))


def test_sample_neighbors_subset_counts_and_order():
    g = _graph()
    deg = np.bincount(g._dst, minlength=g.number_of_nodes())
    seeds = np.array([5, 0, 17, 99, 64])
    for fanout in (1, 2, 3, 10, None, -1):
        f = dataloading.sample_neighbors(g, seeds, fanout, generator=np.random.default_rng(1))
        assert f.number_of_nodes() == g.number_of_nodes()
        eid = f.edata["_ID"].numpy()
        assert len(set(eid.tolist())) == eid.size                      # without replacement
        np.testing.assert_array_equal(g._src[eid], f._src)             # the parent's edges, by id
        np.testing.assert_array_equal(g._dst[eid], f._dst)
        want = deg[seeds] if fanout in (None, -1) else np.minimum(deg[seeds], fanout)
        got = np.array([(f._dst == v).sum() for v in seeds])
        np.testing.assert_array_equal(got, want)
        assert set(f._dst.tolist()) <= set(seeds.tolist())
        # grouped by seed in seed order, ascending parent edge id inside a seed
        pos = {v: i for i, v in enumerate(seeds.tolist())}
        key = np.array([pos[v] for v in f._dst.tolist()])
        assert (np.diff(key) >= 0).all()
        for v in seeds:
            assert (np.diff(eid[f._dst == v]) > 0).all()


def test_sample_neighbors_is_uniform():
    g = _graph((64,), seed=3)
    deg = np.bincount(g._dst, minlength=g.number_of_nodes())
    v = int(np.argmax(deg)); d = int(deg[v])
    assert d >= 3
    rng = np.random.default_rng(7)
    trials, k = 6000, 2
    hits = {}
    for _ in range(trials):
        f = dataloading.sample_neighbors(g, [v], k, generator=rng)
        for e in f.edata["_ID"].tolist():
            hits[e] = hits.get(e, 0) + 1
    assert len(hits) == d
    p = k / d
    sigma = (trials * p * (1 - p)) ** 0.5
    for c in hits.values():                                            # each in-edge kept with probability k/d
        assert abs(c - trials * p) < 5 * sigma


def test_sample_neighbors_errors_and_empty():
    g = _graph()
    with pytest.raises(ValueError):
        dataloading.sample_neighbors(g, [g.number_of_nodes()], 2)
    with pytest.raises(ValueError):
        dataloading.sample_neighbors(g, [0], 2, edge_dir="out")
    with pytest.raises(ValueError):
        dataloading.sample_neighbors(g, [0], 2, replace=True)
    f = dataloading.sample_neighbors(g, np.zeros(0, np.int64), 2)
    assert f.number_of_edges() == 0
    f = dataloading.sample_neighbors(g, [1, 2], 0)
    assert f.number_of_edges() == 0
    b = to_block(f, [1, 2])
    assert (b.number_of_src_nodes(), b.number_of_dst_nodes(), b.number_of_edges()) == (2, 2, 0)


def test_to_block_layout():
    g = _graph()
    seeds = np.array([30, 2, 77])
    f = dataloading.in_subgraph(g, seeds)
    b = dgl.to_block(f, seeds)
    assert isinstance(b, Block) and b.is_block
    nid = b.srcdata[dgl.NID].numpy()
    np.testing.assert_array_equal(nid[: seeds.size], seeds)           # dst nodes first, in the given order
    np.testing.assert_array_equal(b.dstdata[dgl.NID].numpy(), seeds)
    assert len(set(nid.tolist())) == nid.size
    np.testing.assert_array_equal(nid[b._src], f._src)                 # edges kept in frontier order, relabelled
    np.testing.assert_array_equal(nid[b._dst], f._dst)
    assert b._dst.max() < b.number_of_dst_nodes()
    extra = nid[seeds.size:]                                           # the rest in order of first appearance
    firsts = [int(np.nonzero(f._src == v)[0][0]) for v in extra]
    assert firsts == sorted(firsts)
    assert b.in_degrees().shape[0] == seeds.size
    c = b.csc("cpu")
    assert c.num_dst == seeds.size and c.min_in_degree >= 1 and c.num_nodes == nid.size
    with pytest.raises(ValueError):
        to_block(f, [30, 2])                                           # an edge ends outside dst_nodes
    with pytest.raises(ValueError):
        to_block(f, [30, 2, 77, 2])                                    # duplicate dst
    with pytest.raises(ValueError):
        b.add_edges([0], [0])
    with pytest.raises(ValueError):
        b.dstdata["y"] = torch.zeros(b.number_of_src_nodes() + 1)


def test_block_sampler_chains_layers():
    g = _graph()
    dgl.seed(11)
    sampler = dgl.dataloading.MultiLayerNeighborSampler([2, 3, None])
    seeds = np.array([4, 100, 50, 51])
    blocks = sampler.sample_blocks(g, seeds)
    assert len(blocks) == 3
    np.testing.assert_array_equal(blocks[-1].dstdata["_ID"].numpy(), seeds)
    for lo, hi in zip(blocks[:-1], blocks[1:]):                        # a block's dst nodes are the next one's src nodes
        np.testing.assert_array_equal(lo.dstdata["_ID"].numpy(), hi.srcdata["_ID"].numpy())
        assert lo.number_of_dst_nodes() == hi.number_of_src_nodes()
    deg = np.bincount(g._dst, minlength=g.number_of_nodes())
    for b, k in zip(blocks, [2, 3, None]):
        d = np.bincount(b._dst, minlength=b.number_of_dst_nodes())
        want = deg[b.dstdata["_ID"].numpy()]
        np.testing.assert_array_equal(d, want if k is None else np.minimum(want, k))
    full = dgl.dataloading.MultiLayerFullNeighborSampler(2).sample_blocks(g, seeds)
    for b in full:
        np.testing.assert_array_equal(np.bincount(b._dst, minlength=b.number_of_dst_nodes()),
                                      deg[b.dstdata["_ID"].numpy()])
    dgl.seed(11)
    again = sampler.sample_blocks(g, seeds)
    for a, b in zip(blocks, again):                                    # reseeding reproduces the sample
        np.testing.assert_array_equal(a._src, b._src)
        np.testing.assert_array_equal(a.srcdata["_ID"].numpy(), b.srcdata["_ID"].numpy())


@pytest.mark.parametrize("workers", [0, 2])
@pytest.mark.parametrize("drop_last", [False, True])
def test_node_dataloader_covers_nids_and_gathers_data(workers, drop_last):
    g = _graph()
    torch.manual_seed(0)
    nids = list(range(0, g.number_of_nodes(), 2))
    dl = dgl.dataloading.NodeDataLoader(g, nids, dgl.dataloading.MultiLayerNeighborSampler([2, 2]), device="cpu",
                                        batch_size=16, shuffle=True, drop_last=drop_last, num_workers=workers)
    assert len(dl) == (len(nids) // 16 if drop_last else -(-len(nids) // 16))
    seen = []
    for input_nodes, seeds, blocks in dl:
        assert len(blocks) == 2
        torch.testing.assert_close(blocks[0].srcdata["fvs"], g.ndata["fvs"][input_nodes])
        torch.testing.assert_close(blocks[-1].dstdata["y"], g.ndata["y"][seeds])
        assert blocks[0].srcdata["fvs"].shape[0] == blocks[0].number_of_src_nodes()
        seen += seeds.tolist()
    if drop_last:
        assert len(seen) == (len(nids) // 16) * 16 and set(seen) <= set(nids)
    else:
        assert sorted(seen) == nids                                    # every seed exactly once per epoch
    assert seen != nids                                                # shuffled
    early = iter(dl)                                                   # abandoning an iterator must not hang
    next(early)
    del early


def test_node_dataloader_propagates_worker_errors():
    g = _graph()

    class Bad(dataloading.BlockSampler):
        def sample_frontier(self, block_id, g, seed_nodes):
            raise RuntimeError("boom")
    dl = dataloading.NodeDataLoader(g, [0, 1, 2], Bad(1), batch_size=2, num_workers=1)
    with pytest.raises(RuntimeError, match="boom"):
        list(dl)
    with pytest.raises(ValueError):
        dataloading.NodeDataLoader(g, [0], Bad(1), batch_size=0)
