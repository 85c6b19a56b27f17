"""Neighbour-sampled training on message-flow blocks: the HIP layers on sampled blocks against
the oracle's bipartite restatement (oracle/dgl_cpu.py stack_blocks), layer by layer and through
``*Net.forward_batch`` (reference models.py:331-340, 394-400, 685-689, 814-817; job_runner.py:1484-1506)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import dgl_cpu as O
from spgnn_amd import dgl_compat as dgl, models, nn as snn, synthetic
from spgnn_amd.configs import get_config
from spgnn_amd.dataloading import sample_block_device
from spgnn_amd.graph import DeviceBlock, build_csc_numpy
from tests.util import rel_err, sample_neighbors_host

pytestmark = pytest.mark.gpu
TOL = 1e-5     # same bar as the full-graph path: fp32, relative


def _blocks_host(blocks):
    return [(torch.from_numpy(b._src), torch.from_numpy(b._dst), b.number_of_src_nodes(), b.number_of_dst_nodes())
            for b in blocks]


def _sample(n_trees, fanouts, rate=0.3, seed=0, pos_enc_dim=0, where="cuda"):
    """Blocks on the GPU, sampled either there (HIP sampler) or on the host (numpy sampler) and uploaded."""
    g = synthetic.make_batch(n_trees, rank=11, device="cuda", pos_enc_dim=pos_enc_dim)
    dgl.seed(seed)
    rng = np.random.default_rng(seed)
    nids = rng.choice(g.number_of_nodes(), int(g.number_of_nodes() * rate), replace=False)
    blocks = dgl.dataloading.MultiLayerNeighborSampler(fanouts).sample_blocks(g if where == "cuda" else g.cpu(), nids)
    assert all(isinstance(b, DeviceBlock) == (where == "cuda") for b in blocks)
    _, _, blocks = dgl.dataloading.NodeCollator(g, nids, None).attach(blocks, "cuda")
    return g, blocks


@pytest.mark.parametrize("where", ["cuda", "cpu"])
@pytest.mark.parametrize("fanout", [1, 2, None])
def test_layers_on_a_block_match_oracle(fanout, where):
    g, (b,) = _sample(3, [fanout], seed=2, where=where)
    src, dst, ns, nd = _blocks_host([b])[0]
    assert nd < ns
    torch.manual_seed(0)
    x = torch.randn(ns, 96, device="cuda")
    xc = x.cpu()

    sage = snn.SAGEConv(96, 64, "pool", activation=F.elu).cuda()
    ref = O.sage_conv_pool_block(src, dst, ns, nd, xc, *[t.detach().cpu() for t in (
        sage.fc_pool.weight, sage.fc_pool.bias, sage.fc_self.weight, sage.fc_self.bias, sage.fc_neigh.weight,
        sage.fc_neigh.bias)], None, F.elu)
    out = sage(b, x)
    assert out.shape == (nd, 64) and rel_err(out, ref) < TOL

    gat = snn.GATConv(96, 32, 4, residual=True, activation=F.elu).cuda()
    out, att = gat(b, x, get_attention=True)
    ref, ra = O.gat_conv_block(src, dst, ns, nd, xc, gat.fc.weight.detach().cpu(), gat.attn_l.detach().cpu(),
                               gat.attn_r.detach().cpu(), gat.res_fc.weight.detach().cpu(), gat.bias.detach().cpu(),
                               0.2, F.elu)
    assert out.shape == (nd, 4, 32) and rel_err(out, ref) < TOL
    assert att.shape == (b.number_of_edges(), 4, 1) and rel_err(att.squeeze(-1), ra) < TOL

    mlp = torch.nn.Sequential(torch.nn.Linear(96, 48), torch.nn.LeakyReLU()).cuda()
    gin = snn.GINConv(mlp, "mean", learn_eps=True).cuda()
    with torch.no_grad():
        gin.eps.fill_(0.3)
    ref = O.gin_conv_block(src, dst, ns, nd, xc, gin.eps.detach().cpu(),
                           lambda t: F.leaky_relu(F.linear(t, mlp[0].weight.detach().cpu(), mlp[0].bias.detach().cpu())))
    out = gin(b, x)
    assert out.shape == (nd, 48) and rel_err(out, ref) < TOL

    for fo in (48, 128):                                   # both multiplication orders of GraphConv
        gc = snn.GraphConv(96, fo, activation=F.elu).cuda()
        with torch.no_grad():
            gc.bias.normal_(0, 0.1)
        ref = O.graph_conv_block(src, dst, ns, nd, xc, gc.weight.detach().cpu(), gc.bias.detach().cpu(), F.elu)
        out = gc(b, x)
        assert out.shape == (nd, fo) and rel_err(out, ref) < TOL


@pytest.mark.parametrize("fanout", [0, 1, 2, 3, 100, None])
def test_device_sampler_is_bit_exact_against_its_host_copy(fanout):
    g = synthetic.make_batch(5, rank=2, device="cuda")
    N = g.number_of_nodes()
    c = g.csc("cuda")
    indptr, indices, eid = (t.cpu().numpy().astype(np.int64) for t in (c.indptr, c.indices, c.eid))
    seeds = np.random.default_rng(1).permutation(N)[: N // 3]
    for rng_seed in (0, 12345, 2 ** 61 + 7):
        b = sample_block_device(g, torch.from_numpy(seeds), fanout, rng_seed=rng_seed)
        src_p, eids, cnt = sample_neighbors_host(indptr, indices, eid, seeds, fanout, rng_seed)
        S, E = seeds.size, int(cnt.sum())
        assert (b.number_of_dst_nodes(), b.number_of_edges()) == (S, E)
        extra = np.setdiff1d(src_p, seeds)                              # ascending parent id
        nid = b.srcdata["_ID"].cpu().numpy()
        np.testing.assert_array_equal(nid, np.concatenate([seeds, extra]))
        np.testing.assert_array_equal(b.dstdata["_ID"].cpu().numpy(), seeds)
        np.testing.assert_array_equal(b.edata["_ID"].cpu().numpy(), eids)
        src_l, dst_l = b._src, b._dst                                   # host view of the device arrays
        np.testing.assert_array_equal(nid[src_l], src_p)
        np.testing.assert_array_equal(dst_l, np.repeat(np.arange(S), cnt))
        want = build_csc_numpy(src_l, dst_l, nid.size)                  # the block's CSC / CSR, as the host builds them
        bc = b.csc()
        for k, v in want.items():
            np.testing.assert_array_equal(getattr(bc, k).cpu().numpy(), v, err_msg=k)
        assert bc.num_dst == S and bc.num_nodes == nid.size
        assert bc.min_in_degree == (int(cnt.min()) if S else 0)
    if fanout == 2:                                                     # a different seed draws a different sample
        a = sample_block_device(g, torch.from_numpy(seeds), 2, rng_seed=1).edata["_ID"]
        assert not torch.equal(a, sample_block_device(g, torch.from_numpy(seeds), 2, rng_seed=2).edata["_ID"])


def test_device_sampler_is_uniform_and_validates_seeds():
    g = synthetic.make_batch(1, rank=4, device="cuda")
    c = g.csc("cuda")
    deg = (c.indptr[1:] - c.indptr[:-1]).cpu().numpy()
    v = int(np.argmax(deg)); d = int(deg[v]); k = 2
    assert d >= 3
    trials = 4000
    hits = torch.zeros(g.number_of_edges(), dtype=torch.int64, device="cuda")
    for t in range(trials):
        b = sample_block_device(g, torch.tensor([v]), k, rng_seed=1000 + t)
        hits[b.edata["_ID"]] += 1
    hits = hits.cpu().numpy()
    cand = c.eid[int(c.indptr[v]): int(c.indptr[v + 1])].cpu().numpy()
    assert hits.sum() == trials * k and np.count_nonzero(hits) == d
    p = k / d
    assert (np.abs(hits[cand] - trials * p) < 5 * (trials * p * (1 - p)) ** 0.5).all()
    for bad in ([0, 0], [-1], [g.number_of_nodes()]):                  # reported, never dereferenced
        with pytest.raises(ValueError):
            sample_block_device(g, torch.tensor(bad), 2)
    b = sample_block_device(g, torch.zeros(0, dtype=torch.int64), 2)
    assert (b.number_of_src_nodes(), b.number_of_dst_nodes(), b.number_of_edges()) == (0, 0, 0)


def test_zero_in_degree_dst_node_raises_like_dgl():
    g, (b,) = _sample(2, [0], seed=1)
    gat = snn.GATConv(16, 8, 2).cuda()
    with pytest.raises(snn.DGLError):
        gat(b, torch.randn(b.number_of_src_nodes(), 16, device="cuda"))
    sage = snn.SAGEConv(16, 8, "pool").cuda()             # SAGEConv has no such check: neigh = 0 for an isolated node
    x = torch.randn(b.number_of_src_nodes(), 16, device="cuda")
    out = sage(b, x)
    ref = F.linear(x, sage.fc_self.weight, sage.fc_self.bias) + sage.fc_neigh.bias
    assert rel_err(out, ref[: b.number_of_dst_nodes()]) < TOL


def _net(name, seed=0):
    cfg = get_config(name)
    torch.manual_seed(seed)
    model = models.build_model(cfg.MODEL).cuda()
    model.init(None)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith("bias") or n.endswith("eps"):
                p.normal_(0, 0.05)
    model.set_gcn_only()
    return cfg, model


@pytest.mark.parametrize("name,fanouts", [("st_sage_3", [3, 2, 2, 1]), ("st_gat_3", [2, 2, 2, 2]),
                                          ("st_gin_3", [2, 3, 1, 2])])
def test_forward_batch_matches_oracle_with_gradients(name, fanouts):
    cfg, model = _net(name)
    n_layers = {"sage": lambda m: len(m.sage.g_layers), "gat": lambda m: len(m.gat.gat_layers),
                "gin": lambda m: len(m.gin.gin_layers)}[cfg.KIND](model)
    fanouts = fanouts[:n_layers]
    assert len(fanouts) == n_layers
    g, blocks = _sample(4, fanouts, seed=3)
    model.eval()                                          # dropout off; gradients still flow
    x, y = blocks[0].srcdata["fvs"], blocks[-1].dstdata["y"]
    out, emb = model.forward_batch(blocks, x)
    assert out.shape == (blocks[-1].number_of_dst_nodes(), 22)
    sd = {k: v.detach().cpu().clone().requires_grad_(v.dtype.is_floating_point) for k, v in model.state_dict().items()}
    ref, remb = O.net_forward_batch(cfg.KIND, sd, _blocks_host(blocks), x.cpu())
    assert rel_err(out, ref) < TOL and rel_err(emb, remb) < TOL
    w = torch.linspace(0.5, 1.5, 22)
    F.cross_entropy(out, y, weight=w.cuda()).backward()   # the reference's sampled-step loss (job_runner.py:1504)
    F.cross_entropy(ref, y.cpu(), weight=w).backward()
    # gradients are judged as in test_hip_models.py: within 1e-4 of the fp32 oracle, or no further from an fp64
    # evaluation of the oracle than 5x the fp32 oracle's own distance from it.  Two layer types have gradients that
    # are discontinuous in the activations and get an L2 bound when that fails: SAGE's max-pool routing (ties, see
    # test_hip_models.py) and GIN's LeakyReLU MLP - a pre-activation within an ulp of zero takes slope 1 on one side
    # and 0.01 on the other (measured on this very sample: the fp64 oracle's first-layer gradients move by
    # 1.6e-4 / 3.3e-4 / 1.9e-4 when the inputs are perturbed by 2e-7 relative, the same amounts the HIP path differs by).
    sd64 = {k: v.detach().cpu().double().requires_grad_(v.dtype.is_floating_point) for k, v in model.state_dict().items()}
    ref64, _ = O.net_forward_batch(cfg.KIND, sd64, _blocks_host(blocks), x.cpu().double())
    F.cross_entropy(ref64, y.cpu(), weight=w.double()).backward()
    checked = 0
    for n, p in model.named_parameters():
        if p.grad is None and sd[n].grad is None:
            continue
        assert p.grad is not None and sd[n].grad is not None, n
        d = p.grad.cpu().double() - sd[n].grad.double()
        e32 = rel_err(p.grad, sd[n].grad)
        ok = e32 < 1e-4 or rel_err(p.grad, sd64[n].grad) < 5 * rel_err(sd[n].grad, sd64[n].grad) + 1e-6
        if not ok and cfg.KIND in ("sage", "gin"):
            ok = d.norm() / sd[n].grad.double().norm() < 5e-3
        assert ok, (n, e32)
        checked += 1
    assert checked >= 2 * n_layers


def test_sampled_training_loop_reduces_loss():
    """The reference's sampled GraphSAGE epoch (job_runner.py:1484-1506) end to end: NodeDataLoader over a 30 % node
    sample with model.node_ks fanouts, forward_batch, weighted CE, SGD."""
    cfg, model = _net("st_sage_3", seed=4)
    g = synthetic.make_batch(8, rank=5, device="cuda")
    model.train()
    opt = torch.optim.SGD(model.parameters(), lr=cfg.OPTIMIZER["lr"], momentum=cfg.OPTIMIZER["momentum"])
    ks = list(model.node_ks)
    assert len(ks) == len(model.sage.g_layers)
    rng = np.random.default_rng(0)
    nids = rng.choice(g.number_of_nodes(), int(g.number_of_nodes() * model.node_sample_rate), replace=False)
    dgl.seed(0); torch.manual_seed(0)
    losses = []
    for epoch in range(6):
        dl = dgl.dataloading.NodeDataLoader(g, nids, dgl.dataloading.MultiLayerNeighborSampler(ks), device="cuda",
                                            batch_size=256, shuffle=True, drop_last=False, num_workers=1)
        tot = 0.0
        for input_nodes, seeds, blocks in dl:
            assert blocks[0].srcdata["fvs"].is_cuda and blocks[0].number_of_src_nodes() == input_nodes.shape[0]
            opt.zero_grad()
            out, _ = model.forward_batch(blocks, blocks[0].srcdata["fvs"])
            loss = F.cross_entropy(out, blocks[-1].dstdata["y"])
            loss.backward()
            opt.step()
            tot += float(loss.detach()) * seeds.shape[0]
        losses.append(tot / len(nids))
    assert all(np.isfinite(losses)) and losses[-1] < 0.8 * losses[0], losses
