"""Cached-embedding reader and batch assembler against the synthetic generator."""
import numpy as np
import pytest
import torch

from spgnn_amd import data, synthetic


def _write(tmp_path, n=5):
    samples = synthetic.synthetic_trees(n, rank=9, n_lo=21, n_hi=60)
    uids = [f"1.2.840.{i}" for i in range(n)]
    for uid, s in zip(uids, samples):
        data.write_embedding(str(tmp_path), uid, s)
    return samples, uids


def test_dataset_reads_reference_schema_and_collates(tmp_path):
    samples, uids = _write(tmp_path)
    ds = data.ConvEmbeddingDataset(str(tmp_path), uids)
    assert len(ds) == 5
    item = ds[2]
    assert item["fvs"].dtype == np.float64 and item["adj"].dtype == np.uint8 and item["labels"].dtype == np.uint8
    assert np.array_equal(item["adj"], samples[2]["adj"]) and np.allclose(item["fvs"], samples[2]["fvs"])
    loader = torch.utils.data.DataLoader(ds, batch_size=3, collate_fn=data.collate_native, num_workers=0)
    batches = list(loader)
    assert [len(b["adj"]) for b in batches] == [3, 2] and batches[0]["meta"]["uid"] == uids[:3]
    with open(tmp_path / "derived" / "conv_embedding" / "bad.pkl", "wb") as fp:
        import pickle; pickle.dump({"fvs": 1}, fp)
    with pytest.raises(KeyError):
        data.ConvEmbeddingDataset(str(tmp_path), ["bad"])[0]


def test_assemble_batch_on_cpu_matches_generator(tmp_path):
    samples, uids = _write(tmp_path)
    ds = data.ConvEmbeddingDataset(str(tmp_path), uids)
    g = data.assemble_batch(data.collate_native([ds[i] for i in range(5)]), device="cpu")
    ref = synthetic.batch_from_samples(samples, "cpu", 39)
    assert np.array_equal(g._src, ref._src) and np.array_equal(g._dst, ref._dst)
    assert g.batch_num_nodes_list == ref.batch_num_nodes_list and g.batch_num_edges_list == ref.batch_num_edges_list
    for k in ("fvs", "fvs_out", "y", "pos_enc"):
        assert torch.equal(g.ndata[k], ref.ndata[k]), k


def _write_reference_state(tmp_path, samples, uids):
    """Pickles with the reference's exact ``save_state`` dict (job_runner.py:796-805): float64 fvs / fvs_out, uint8 adj /
    labels, plus the keys the GNN stage never reads (ref, all_airway, branch_info, meta)."""
    import pickle
    d = tmp_path / "derived" / "conv_embedding"
    d.mkdir(parents=True, exist_ok=True)
    rng = np.random.default_rng(0)
    for uid, s in zip(uids, samples):
        n = s["adj"].shape[0]
        state = {"fvs": s["fvs"].astype(np.float64), "adj": s["adj"].astype(np.uint8), "labels": s["labels"].astype(np.uint8),
                 "fvs_out": s["fvs_out"].astype(np.float64), "ref": rng.integers(0, 2, (4, 4, 4)).astype(np.uint8),
                 "all_airway": rng.integers(0, n, (4, 4, 4)).astype(np.int16),
                 "branch_info": {i: {"len": float(i)} for i in range(n)}, "meta": {"uid": [uid], "spacing": [np.ones(3)]}}
        with open(d / f"{uid}.pkl", "wb") as fp:
            pickle.dump(state, fp)


@pytest.mark.gpu
def test_assemble_batch_on_gpu_matches_networkx_oracle(tmp_path):
    """SURVEY.md §8f-2 end to end on the device, checked against the ORACLE (real networkx driven as the reference drives
    it: oracle/graph_rule_nx.py), not against this package's own host path: edge list and batching, CSC, node data,
    anchors (device kernel) and the distance encoding (device kernel), all bit-exact."""
    from oracle import graph_rule_nx as R
    n_trees = 24
    samples = synthetic.synthetic_trees(n_trees, rank=3, n_lo=21, n_hi=200)
    uids = [f"1.2.840.{i}" for i in range(n_trees)]
    _write_reference_state(tmp_path, samples, uids)
    ds = data.ConvEmbeddingDataset(str(tmp_path), uids)
    loader = torch.utils.data.DataLoader(ds, batch_size=n_trees, collate_fn=data.collate_native, num_workers=0)
    g = data.assemble_batch(next(iter(loader)), device="cuda")
    # graph rule + dgl.batch
    edges = [R.edges_spgnn(s["adj"]) for s in samples]
    ns = [s["adj"].shape[0] for s in samples]
    src, dst = R.batch_edges(edges, ns)
    assert np.array_equal(g._src, src) and np.array_equal(g._dst, dst)
    indptr, indices, eid = R.csc_stable(src, dst, sum(ns))
    csc = g.csc()
    assert np.array_equal(csc.indptr.cpu().numpy(), indptr) and np.array_equal(csc.indices.cpu().numpy(), indices)
    assert np.array_equal(csc.eid.cpu().numpy(), eid)
    # node data: float64 -> float32 casts of the pickled arrays
    assert torch.equal(g.ndata["fvs"].cpu(), torch.from_numpy(np.concatenate([s["fvs"] for s in samples]).astype(np.float32)))
    assert torch.equal(g.ndata["fvs_out"].cpu(), torch.from_numpy(np.concatenate([s["fvs_out"] for s in samples]).astype(np.float32)))
    assert torch.equal(g.ndata["y"].cpu(), torch.from_numpy(np.concatenate([s["labels"] for s in samples]).astype(np.int64)))
    # anchors + distance encoding: the reference takes softmax(fvs_out) with torch on its device and continues on the host
    prob = torch.softmax(g.ndata["fvs_out"], dim=1).cpu().numpy()
    off, pes = 0, []
    for s, n in zip(samples, ns):
        anc = R.anchors_from_probabilities(prob[off:off + n], s["adj"], 39)
        pes.append(R.distance_pos_enc(s["adj"], anc)[0])
        off += n
    assert torch.equal(g.ndata["pos_enc"].cpu(), torch.from_numpy(np.concatenate(pes)))
    assert g.ndata["pos_enc"].stride(0) % 4 == 0           # rows ready for the vector / MFMA kernels


def _tree_from_parents(parent):
    n = len(parent)
    adj = np.eye(n, dtype=np.uint8)
    c = np.arange(1, n)
    adj[parent[1:], c] = 1; adj[c, parent[1:]] = 1
    return adj


@pytest.mark.gpu
def test_device_anchor_selection_matches_networkx_oracle_including_ties():
    """spgnn_tree_anchors vs the reference's own host code on real networkx + real Python sets, on trees built to tie:
    complete binary / ternary trees (all leaves at one depth), brooms, paths, random trees up to 700 nodes (ids beyond
    the emulated set's table mask, several table growths), with exactly tied probabilities for the greedy argmax too."""
    from oracle import graph_rule_nx as R
    from spgnn_amd import graph as G
    from spgnn_amd.posenc import anchors_device
    rng = np.random.default_rng(5)
    adjs = []
    for k in (2, 3):                                       # complete k-ary trees, breadth-first numbering
        for n in (40, 121, 364, 700):
            adjs.append(_tree_from_parents(np.array([-1] + [(i - 1) // k for i in range(1, n)])))
    adjs.append(_tree_from_parents(np.array([-1] + list(range(0, 59)))))                       # a path
    adjs.append(_tree_from_parents(np.array([-1] + list(range(0, 30)) + [30] * 40)))           # a broom: 40 tied leaves
    for n in (21, 33, 150, 299, 511, 640):
        adjs.append(synthetic.random_tree_adj(n, rng))
    graphs, probs = [], []
    for adj in adjs:
        n = adj.shape[0]
        lg = rng.standard_normal((n, 22)).astype(np.float32)
        lg[rng.integers(0, n, 8)] = lg[rng.integers(0, n)]                                      # identical rows: exact argmax ties
        g = G.graph_from_adj(adj, device="cpu", add_self_loops=True)
        g.ndata["fvs_out"] = torch.from_numpy(lg)
        graphs.append(g)
    bg = G.batch(graphs).to("cuda")
    anc = anchors_device(bg, bg.ndata["fvs_out"], 39).cpu().numpy()
    prob = torch.softmax(bg.ndata["fvs_out"], dim=1).cpu().numpy()
    off = 0
    for t, adj in enumerate(adjs):
        n = adj.shape[0]
        ref = R.anchors_from_probabilities(prob[off:off + n], adj, 39)
        assert (anc[t] - off).tolist() == [int(x) for x in ref], (t, n)
        off += n
    anc21 = anchors_device(bg, bg.ndata["fvs_out"], 21).cpu().numpy()
    assert anc21.shape[1] == 21 and np.array_equal(anc21, anc[:, :21])


@pytest.mark.gpu
def test_build_csc_device_bit_exact_at_512_trees():
    """the batch's edge list, CSC, CSR and slot map built on the device (spgnn_build_csc) from the packed
    adjacency matrices - bit-exact against the networkx oracle of the reference rule (oracle/graph_rule_nx.py) at the
    bench's batch size, and against the host builder for matrices the reference never produces (asymmetric, zero diagonal,
    a single-node tree): the rule is `non-zero off-diagonal entries in row-major order, then the self loops`."""
    import time
    from oracle import graph_rule_nx as R
    from spgnn_amd import graph as G
    samples = synthetic.synthetic_trees(512, rank=1)
    adjs = [s["adj"] for s in samples]
    ns = [a.shape[0] for a in adjs]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    src, dst, csc, nn_, ne_ = G.build_csc_device(adjs, "cuda")
    torch.cuda.synchronize()
    t_dev = time.perf_counter() - t0
    e_src, e_dst = R.batch_edges([R.edges_spgnn(a) for a in adjs], ns)
    assert np.array_equal(src.cpu().numpy(), e_src) and np.array_equal(dst.cpu().numpy(), e_dst)
    assert nn_ == ns and ne_ == [3 * n - 2 for n in ns]            # trees: 2 (n - 1) directed edges + n self loops
    indptr, indices, eid = R.csc_stable(e_src, e_dst, sum(ns))
    assert np.array_equal(csc.indptr.cpu().numpy(), indptr) and np.array_equal(csc.indices.cpu().numpy(), indices)
    assert np.array_equal(csc.eid.cpu().numpy(), eid)
    t0 = time.perf_counter()
    ref = G.build_csc_numpy(e_src, e_dst, sum(ns))
    t_host = time.perf_counter() - t0
    for k in ("indptr", "indices", "eid", "out_indptr", "out_indices", "out_pos"):
        assert np.array_equal(getattr(csc, k).cpu().numpy(), ref[k]), k
    assert (csc.min_in_degree, csc.max_in_degree) == (int(np.diff(ref["indptr"]).min()), int(np.diff(ref["indptr"]).max()))
    assert (csc.min_out_degree, csc.max_out_degree) == (int(np.diff(ref["out_indptr"]).min()), int(np.diff(ref["out_indptr"]).max()))
    print(f"build_csc_device 512 trees (N={sum(ns)}, E={src.numel()}): {t_dev * 1e3:.1f} ms incl. upload; host CSC alone {t_host * 1e3:.1f} ms")
    # general matrices
    rng = np.random.default_rng(0)
    odd = [(rng.random((n, n)) < 0.15).astype(np.uint8) for n in (1, 2, 7, 33, 300)]
    odd.append(np.triu(adjs[0]))                                    # the reference's unused `tree_downstream` form
    odd.append(np.zeros((5, 5), dtype=np.uint8))                    # no edges at all: only the self loops
    src, dst, csc, nn_, ne_ = G.build_csc_device(odd, "cuda")
    us, vs, off = [], [], 0
    for a in odd:
        u, v = G.edges_from_adj(a)
        us.append(u + off); vs.append(v + off); off += a.shape[0]
    e_src, e_dst = np.concatenate(us), np.concatenate(vs)
    assert np.array_equal(src.cpu().numpy(), e_src) and np.array_equal(dst.cpu().numpy(), e_dst)
    ref = G.build_csc_numpy(e_src, e_dst, off)
    for k in ("indptr", "indices", "eid", "out_indptr", "out_indices", "out_pos"):
        assert np.array_equal(getattr(csc, k).cpu().numpy(), ref[k]), k
    assert ne_ == [int(x.shape[0]) for x in us]
    g = G.TreeGraph.from_device(src, dst, off, csc, nn_, ne_)
    assert g.csc("cuda") is csc and g.number_of_edges() == e_src.shape[0] and np.array_equal(g._src, e_src)
    assert torch.equal(g.edges()[0].cpu(), torch.from_numpy(e_src))


@pytest.mark.gpu
def test_tree_downstream_batches_on_the_device_and_through_a_model():
    """GRAPH_MODE == "tree_downstream" (reference job_runner.py:1334-1336) end to end: the device builder gives the networkx
    rule's edge list, and a GAT head run on the directed batch (every node: its parent's message and its own) equals the CPU
    oracle on the same edge list."""
    from oracle import dgl_cpu as O
    from oracle import graph_rule_nx as R
    from spgnn_amd import graph as G, models
    from spgnn_amd.configs import get_config
    from spgnn_amd.data import assemble_batch
    samples = synthetic.synthetic_trees(6, rank=3)
    adjs = [s["adj"] for s in samples]
    ns = [a.shape[0] for a in adjs]
    src, dst, csc, nn_, ne_ = G.build_csc_device(adjs, "cuda", graph_mode="tree_downstream")
    e_src, e_dst = R.batch_edges([R.edges_gcn(a, "tree_downstream") for a in adjs], ns)
    assert np.array_equal(src.cpu().numpy(), e_src) and np.array_equal(dst.cpu().numpy(), e_dst)
    assert ne_ == [2 * n - 1 for n in ns] and csc.min_in_degree == 1 and csc.min_out_degree == 1
    g = assemble_batch(samples, "cuda", pos_enc_dim=None, graph_mode="tree_downstream")
    assert g.number_of_edges() == e_src.shape[0]
    cfg = get_config("st_gat_3")
    torch.manual_seed(0)
    model = models.build_model(cfg.MODEL).cuda()
    model.init(None); model.set_gcn_only(); model.eval()
    with torch.no_grad():
        logits = model(g)[0]
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    ref = O.net_forward(cfg.KIND, sd, torch.from_numpy(e_src), torch.from_numpy(e_dst), sum(ns), g.ndata["fvs"].cpu(), None)[0]
    err = float((logits.cpu() - ref).abs().max() / ref.abs().max())
    assert err < 1e-5, err
