"""CPU-only checks: the C-ABI library loads and exports every symbol include/spgnn_hip.h declares,
the embedded configs equal the reference's exp_settings, the loss/mask host logic equals the
reference formulation, and the N>1 data-parallel step (gloo, world_size 2) equals the 1-rank step."""
import json
import os
import re
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(lib):
    header = open(os.path.join(ROOT, "include", "spgnn_hip.h")).read()
    declared = set(re.findall(r"\b(spgnn_[a-z0-9_]+)\s*\(", header))
    from spgnn_amd import _capi
    assert declared == set(_capi.SIGNATURES), declared ^ set(_capi.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.spgnn_abi_version() == int(re.search(r"#define SPGNN_ABI_VERSION (\d+)", header).group(1))
    # argument validation happens before any device work, so it can be exercised without a GPU
    assert lib.spgnn_gat_fwd(0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 4, 4, 2, 4, 0.2, 0, 0.0, 0, 0, 0.0, 0, 0, 0, 0, 0) == -1
    assert b"null pointer" in lib.spgnn_last_error()
    assert lib.spgnn_spmm_sum(0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, -3, 0, 8, 0, 0) == -2
    assert lib.spgnn_gat_bwd_src(0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 2, 4, 0.0, 0, 0, 0) == 0    # N == 0: no-op


def test_integration_stub_names_the_current_abi_version():
    """INTEGRATION.md's binding stub is what a maintainer pastes first: its version assert must be the header's."""
    header = open(os.path.join(ROOT, "include", "spgnn_hip.h")).read()
    ver = int(re.search(r"#define SPGNN_ABI_VERSION (\d+)", header).group(1))
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    stub = [int(v) for v in re.findall(r"spgnn_abi_version\(\)\s*==\s*(\d+)", doc)]
    assert stub and all(v == ver for v in stub), (stub, ver)
    from spgnn_amd import _capi
    assert _capi.ABI_VERSION == ver


def test_prep_cache_never_evicts_what_a_graph_recorded(monkeypatch):
    """ops._prep_lookup: LRU over loose entries; an entry that ran under stream capture is pinned and is handed
    to the capturing step's reference list."""
    from spgnn_amd import ops

    class P:
        pass
    cache = {}
    capturing = {"on": False}
    monkeypatch.setattr(torch.cuda, "is_current_stream_capturing", lambda: capturing["on"])
    monkeypatch.setattr(ops, "CAPTURE_REFS", [])
    capturing["on"] = True
    held = ops._prep_lookup(cache, "captured", P)
    capturing["on"] = False
    assert held.pinned and ops.CAPTURE_REFS == [held]
    for i in range(3 * ops._PREP_CACHE_MAX):
        ops._prep_lookup(cache, ("loose", i), P)
    assert cache["captured"] is held
    loose = [k for k in cache if k != "captured"]
    assert len(loose) == ops._PREP_CACHE_MAX and loose[-1] == ("loose", 3 * ops._PREP_CACHE_MAX - 1)
    first = loose[0]
    assert ops._prep_lookup(cache, first, P) is not None and list(cache)[-1] == first      # a hit becomes the most recent
    ops._prep_lookup(cache, "new", P)
    assert first in cache and loose[1] not in cache


def test_isa_fence_finds_a_packed_write_feeding_a_cross_lane_read(monkeypatch):
    """csrc/build.py packed_crosslane_hazards: the build fails when a DPP / bpermute / readlane operand was
    last written by a packed fp32 op; a plain move in between (single_pass()) clears it; registers of a pair are tracked."""
    from spgnn_amd.csrc import build as b
    isa = """
0000000000001000 <bad_kernel>:
	v_pk_fma_f32 v[4:5], v[0:1], v[2:3], v[4:5] op_sel_hi:[0,1,1]  // 000000001000: 00000000
	v_add_f32_dpp v6, v5, v5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf  // 000000001008: 00000000
	s_endpgm
0000000000002000 <fenced_kernel>:
	v_pk_mul_f32 v[4:5], v[0:1], v[2:3]  // 000000002000: 00000000
	v_mov_b32_e32 v5, v5  // 000000002008: 00000000
	v_add_f32_dpp v6, v5, v5 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf  // 000000002010: 00000000
	ds_bpermute_b32 v7, v8, v4  // 000000002018: 00000000
	s_endpgm
0000000000003000 <far_kernel>:
	v_pk_add_f32 v[10:11], v[0:1], v[2:3]  // 000000003000: 00000000
""" + "\tv_add_f32_e32 v20, v21, v22  // 0: 0\n" * (b.HAZARD_WINDOW + 1) + """	v_readlane_b32 s4, v10, 3  // 000000003100: 00000000
	s_endpgm
"""
    monkeypatch.setattr(b, "device_isa", lambda obj: isa)
    hz = b.packed_crosslane_hazards("x.o")
    fns = [h[0] for h in hz]
    assert fns.count("bad_kernel") == 1 and "v5" in hz[0][1]
    assert fns.count("fenced_kernel") == 1 and "ds_bpermute" in [h for h in hz if h[0] == "fenced_kernel"][0][1]     # v4 of the pair, unfenced
    assert "far_kernel" not in fns                                     # beyond the window: the two passes are long done


def test_ops_refuse_cpu_tensors():
    from spgnn_amd import nn as snn
    from spgnn_amd.graph import TreeGraph
    g = TreeGraph((np.array([0, 1, 0, 1]), np.array([1, 0, 0, 1])), 2)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        snn.GATConv(4, 4, 2)(g, torch.randn(2, 4))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        snn.GraphConv(4, 2)(g, torch.randn(2, 4))


@pytest.mark.skipif(not os.path.isdir("/root/reference/exp_settings"), reason="reference checkout not present")
def test_embedded_configs_equal_reference_settings():
    from spgnn_amd import configs as c
    for name, mine in c.CONFIGS.items():
        s = c.load_settings(f"/root/reference/exp_settings/{name}.py")
        ref_model = dict(s.MODEL)
        if name == "st_pgat_spgnnnl_3":      # the reference names a class that does not exist
            assert ref_model.pop("method") == "models.GATPositionLSPENet"
            ref_model["method"] = "models.GATPositionSPGNNNet"
        assert ref_model == mine["MODEL"], name
        assert s.SAMPLING_RATE == mine["SAMPLING_RATE"] and getattr(s, "POS_ENC_DIM", None) == mine["POS_ENC_DIM"]
        assert s.CLASS_WEIGHTS == mine["CLASS_WEIGHTS"] and s.OPTIMIZER == mine["OPTIMIZER"]
        assert s.GCN_STEPS == mine["GCN_STEPS"] and s.SCHEDULER == mine["SCHEDULER"]


def test_models_build_from_every_config_on_cpu():
    from spgnn_amd import configs as c, models
    for name in c.CONFIGS:
        m = models.build_model(c.get_config(name).MODEL)
        m.init(None); m.set_gcn_only()
        assert all(p.requires_grad for p in m.gnn_out.parameters())
        m.set_cnn_only()
        assert not any(p.requires_grad for p in m.parameters())      # no trunk built: nothing left trainable
        m.set_all()
        with pytest.raises(RuntimeError):
            m.extract_feature(torch.zeros(1))


def test_masked_weighted_ce_equals_reference_formulation():
    from spgnn_amd import configs as c, train
    torch.manual_seed(0)
    n = 500
    logits, y = torch.randn(n, 22), torch.randint(0, 22, (n,))
    y[torch.rand(n) < 0.7] = 0
    w = torch.tensor(c.class_weight_list(c.CLASS_WEIGHTS))
    p = train.sampling_probabilities(y, 0.15)
    assert set(p.unique().tolist()) == {0.15000000596046448, 1.0} and bool((p[y != 0] == 1).all())
    draws = torch.rand(n)
    mask = train.mask_from_draws(draws, p)
    assert bool(mask[y != 0].all())                                  # labelled nodes always kept (job_runner.py:1897)
    ref = F.cross_entropy(logits[mask], y[mask], weight=w)           # job_runner.py:1900
    assert torch.allclose(train.masked_weighted_ce(logits, y, mask, w), ref, rtol=1e-6, atol=0)


# ---- data parallel over gloo ----------------------------------------------------------------------
class _TinyNet(torch.nn.Module):
    """CPU stand-in for the GNN (the HIP ops need a GPU): per-node MLP on g.ndata['fvs']."""
    def __init__(self):
        super().__init__()
        self.a, self.b = torch.nn.Linear(8, 16), torch.nn.Linear(16, 22)

    def forward(self, g):
        return (self.b(torch.tanh(self.a(g.ndata["fvs"]))),)


def _cpu_update(self, inv):   # test-side stand-in for the HIP SGD kernel: same arithmetic in torch
    b = self.bucket
    g = b.flat_grad * inv + self.weight_decay * b.flat_param
    g[b.numel:] = 0
    b.flat_mom.copy_(g if b.steps == 0 else self.momentum * b.flat_mom + g)
    b.flat_param.sub_(self.lr * b.flat_mom)


def _dp_worker(rank, world, port, ret):
    from spgnn_amd import configs as c, synthetic, train
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    train.TrainStep._apply_update = _cpu_update
    torch.manual_seed(0)
    model = _TinyNet()
    samples = synthetic.synthetic_trees(4, rank=0, fv_dim=8, n_lo=21, n_hi=30)
    mine = samples[rank * 2:(rank + 1) * 2]                         # rank r gets trees [r*B/W, (r+1)*B/W)
    g = synthetic.batch_from_samples(mine, "cpu", None)
    full = synthetic.batch_from_samples(samples, "cpu", None)
    offs = np.cumsum([0] + full.batch_num_nodes_list)
    draws_full = torch.rand(3, full.number_of_nodes(), generator=torch.Generator().manual_seed(9))
    ts = train.TrainStep(model, c.class_weight_list(c.CLASS_WEIGHTS), 0.15, 0.05, 0.9)
    losses = [float(ts.step(g, draws_full[i, offs[rank * 2]:offs[rank * 2 + 2]])) for i in range(3)]
    ret[rank] = (losses, ts.bucket.flat_param[:ts.bucket.numel].clone())
    dist.destroy_process_group()


def test_data_parallel_step_equals_single_rank_step():
    from spgnn_amd import configs as c, synthetic, train
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mgr = mp.Manager(); ret = mgr.dict()
    mp.spawn(_dp_worker, args=(2, port, ret), nprocs=2, join=True)
    # the same three steps on one rank with all four trees
    orig = train.TrainStep._apply_update
    train.TrainStep._apply_update = _cpu_update
    try:
        torch.manual_seed(0)
        model = _TinyNet()
        full = synthetic.batch_from_samples(synthetic.synthetic_trees(4, rank=0, fv_dim=8, n_lo=21, n_hi=30), "cpu", None)
        draws_full = torch.rand(3, full.number_of_nodes(), generator=torch.Generator().manual_seed(9))
        ts = train.TrainStep(model, c.class_weight_list(c.CLASS_WEIGHTS), 0.15, 0.05, 0.9)
        losses = [float(ts.step(full, draws_full[i])) for i in range(3)]
        flat = ts.bucket.flat_param[:ts.bucket.numel]
    finally:
        train.TrainStep._apply_update = orig
    for r in range(2):
        assert np.allclose(ret[r][0], losses, rtol=1e-5)            # global class-weighted mean, not a mean of means
        assert torch.allclose(ret[r][1], flat, rtol=1e-5, atol=1e-7)
    assert torch.equal(ret[0][1], ret[1][1])                        # replicas stay identical


def _uneven_setup():
    """Five trees: rank 0 gets the four larger ones (>= 3 x rank 1's node count), rank 1 one small tree WITHOUT labels whose
    draws all miss the sampling rate - its mask keeps no node, its local weight sum and loss numerator are 0."""
    from spgnn_amd import synthetic
    big = synthetic.synthetic_trees(4, rank=0, fv_dim=8, n_lo=40, n_hi=60)
    small = synthetic.synthetic_trees(1, rank=1, fv_dim=8, n_lo=21, n_hi=24)
    small[0]["labels"] = np.zeros_like(small[0]["labels"])
    return big, small


def _dp_worker_uneven(rank, world, port, ret):
    from spgnn_amd import configs as c, synthetic, train
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    train.TrainStep._apply_update = _cpu_update
    torch.manual_seed(0)
    model = _TinyNet()
    big, small = _uneven_setup()
    g = synthetic.batch_from_samples(big if rank == 0 else small, "cpu", None)
    n_big = sum(s_["fvs"].shape[0] for s_ in big)
    n_all = n_big + small[0]["fvs"].shape[0]
    draws_full = torch.rand(3, n_all, generator=torch.Generator().manual_seed(9))
    draws_full[:, n_big:] = 0.99                                    # rank 1: every draw misses SAMPLING_RATE = 0.15
    ts = train.TrainStep(model, c.class_weight_list(c.CLASS_WEIGHTS), 0.15, 0.05, 0.9)
    mine = draws_full[:, :n_big] if rank == 0 else draws_full[:, n_big:]
    losses, wsums = [], []
    for i in range(3):
        ts._front(g, mine[i])
        wsums.append(float(ts.bucket.wsum_slot))                   # the LOCAL class-weight sum, before the exchange
        losses.append(float(ts._back(ts._reduce(ts.bucket.loss_slot))))
    ret[rank] = (losses, ts.bucket.flat_param[:ts.bucket.numel].clone(), wsums, g.number_of_nodes())
    dist.destroy_process_group()


def test_data_parallel_uneven_ranks_and_empty_mask():
    """ranks with clearly unequal node counts, and a rank whose mask keeps no node (local weight sum 0): the
    step must still equal the one-rank step on all trees - the reciprocal is taken of the GLOBAL weight sum."""
    from spgnn_amd import configs as c, synthetic, train
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mgr = mp.Manager(); ret = mgr.dict()
    mp.spawn(_dp_worker_uneven, args=(2, port, ret), nprocs=2, join=True)
    orig = train.TrainStep._apply_update
    train.TrainStep._apply_update = _cpu_update
    try:
        torch.manual_seed(0)
        model = _TinyNet()
        big, small = _uneven_setup()
        full = synthetic.batch_from_samples(big + small, "cpu", None)
        n_big = sum(s_["fvs"].shape[0] for s_ in big)
        draws_full = torch.rand(3, full.number_of_nodes(), generator=torch.Generator().manual_seed(9))
        draws_full[:, n_big:] = 0.99
        ts = train.TrainStep(model, c.class_weight_list(c.CLASS_WEIGHTS), 0.15, 0.05, 0.9)
        losses = [float(ts.step(full, draws_full[i])) for i in range(3)]
        flat = ts.bucket.flat_param[:ts.bucket.numel]
    finally:
        train.TrainStep._apply_update = orig
    assert ret[0][3] >= 3 * ret[1][3]                               # node counts differ by >= 3 x
    assert ret[1][2] == [0.0, 0.0, 0.0] and all(w > 0 for w in ret[0][2])   # rank 1 contributed nothing, rank 0 did
    for r in range(2):
        assert np.all(np.isfinite(ret[r][0])) and np.allclose(ret[r][0], losses, rtol=1e-5)
        assert torch.allclose(ret[r][1], flat, rtol=1e-5, atol=1e-7)
    assert torch.equal(ret[0][1], ret[1][1])


def test_train_step_state_dict_round_trip_matches_torch_sgd_layout():
    """TrainStep.state_dict() has torch.optim.SGD's layout (the reference's ``optimizer_dict``,
    job_runner.py:336-343); save -> reload -> step continues exactly; a torch.optim.SGD state loads too."""
    from spgnn_amd import configs as c, synthetic, train
    orig = train.TrainStep._apply_update
    train.TrainStep._apply_update = _cpu_update
    try:
        g = synthetic.batch_from_samples(synthetic.synthetic_trees(3, rank=0, fv_dim=8, n_lo=21, n_hi=30), "cpu", None)
        draws = torch.rand(6, g.number_of_nodes(), generator=torch.Generator().manual_seed(3))
        w = c.class_weight_list(c.CLASS_WEIGHTS)

        def fresh():
            torch.manual_seed(0)
            return _TinyNet()
        a = train.TrainStep(fresh(), w, 0.15, 0.05, 0.9)
        for i in range(3):
            a.step(g, draws[i])
        sd = a.state_dict()
        model_sd = {k: v.clone() for k, v in a.model.state_dict().items()}
        # layout: what torch.optim.SGD over the same parameters writes
        opt = torch.optim.SGD(list(fresh().parameters()), lr=0.05, momentum=0.9)
        assert set(sd["param_groups"][0]) >= set(opt.state_dict()["param_groups"][0])
        assert sorted(sd["state"]) == list(range(4)) and all("momentum_buffer" in v for v in sd["state"].values())
        for i in range(3, 6):
            a.step(g, draws[i])
        # resume in a new TrainStep
        m2 = fresh(); m2.load_state_dict(model_sd)
        b = train.TrainStep(m2, w, 0.15, 0.01, 0.5)                 # wrong lr / momentum on purpose: the state overrides them
        b.load_state_dict(sd)
        assert b.bucket.steps == 3 and b.lr == 0.05 and b.momentum == 0.9
        for i in range(3, 6):
            b.step(g, draws[i])
        assert torch.equal(a.bucket.flat_param[:a.bucket.numel], b.bucket.flat_param[:b.bucket.numel])
        # a reference-side optimizer state (torch.optim.SGD after real steps) loads, and a torch optimizer loads ours
        m3 = fresh()
        opt3 = torch.optim.SGD(list(m3.parameters()), lr=0.05, momentum=0.9)
        m3(g)[0].sum().backward(); opt3.step()
        c3 = train.TrainStep(fresh(), w, 0.15, 0.05, 0.9)
        c3.load_state_dict(opt3.state_dict())
        assert c3.bucket.steps == 1
        for p, off in zip(m3.parameters(), c3.bucket.offsets):
            assert torch.equal(c3.bucket.flat_mom[off:off + p.numel()].view_as(p), opt3.state[p]["momentum_buffer"])
        opt4 = torch.optim.SGD(list(fresh().parameters()), lr=0.05, momentum=0.9)
        opt4.load_state_dict({k: v for k, v in sd.items() if k != "spgnn"})
        with pytest.raises(ValueError):
            c3.load_state_dict({"state": {}, "param_groups": [{"params": [0, 1]}]})
        # the bucket's tail slots (weight sum, loss numerator) are not part of the optimizer's arithmetic
        assert float(a.bucket.flat_mom[a.bucket.numel:].abs().sum()) == 0.0
    finally:
        train.TrainStep._apply_update = orig


class _Odd:                                                         # a class torch's weights-only unpickler does not allow
    def __init__(self):
        self.v = 3


def test_checkpoint_load_is_weights_only_unless_trusted(tmp_path):
    """load_pretrained_model unpickles tensors and plain containers only; a file with other objects needs trusted=True."""
    from spgnn_amd import checkpoint
    net = torch.nn.Linear(3, 2)
    path = str(tmp_path / "c.pt")
    torch.save({"model_dict": net.state_dict(), "metric": _Odd()}, path)
    with pytest.raises(RuntimeError, match="trusted=True"):
        checkpoint.load_pretrained_model(path, [net], ["model_dict"], device="cpu")
    states = checkpoint.load_pretrained_model(path, [net], ["model_dict"], device="cpu", trusted=True)
    assert states["metric"].v == 3


def test_balanced_tree_partition_balances_node_counts():
    """shard trees over ranks by node count.  4096 trees of U[120,180] nodes over 8 ranks (BASELINE config 5):
    per-rank node sums within 1 % of each other; every tree exactly once; deterministic."""
    from spgnn_amd.train import balanced_tree_partition
    rng = np.random.default_rng(0)
    for trees, world in ((4096, 8), (512, 4), (64, 2), (7, 3)):
        n = rng.integers(120, 181, size=trees)
        parts = balanced_tree_partition(n.tolist(), world)
        assert sorted(i for p in parts for i in p) == list(range(trees))
        assert all(p == sorted(p) for p in parts)
        sums = np.array([n[p].sum() for p in parts])
        if trees >= 64:
            assert (sums.max() - sums.min()) / sums.mean() < 0.01, sums
        assert parts == balanced_tree_partition(n.tolist(), world)
    assert balanced_tree_partition([5, 5, 5], 5)[3:] == [[], []]
    with pytest.raises(ValueError):
        balanced_tree_partition([1], 0)


def test_checkpoint_filter_follows_reference_rule(tmp_path):
    """reference job_runner.py:85-123: keep a saved entry iff its key exists in the live state_dict, it is not ignored and
    its tensor size matches; "metric" objects are overwritten as they are."""
    from spgnn_amd import checkpoint
    net = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.Linear(3, 2))
    saved = {"0.weight": torch.ones(3, 4), "0.bias": torch.ones(5), "1.weight": torch.full((2, 3), 2.0), "9.weight": torch.ones(1)}
    before = {k: v.clone() for k, v in net.state_dict().items()}
    taken = checkpoint.reload_state(net, saved, ignored_keys=["1.weight"])
    assert taken == ["0.weight"]
    after = net.state_dict()
    assert torch.equal(after["0.weight"], torch.ones(3, 4)) and torch.equal(after["0.bias"], before["0.bias"])
    assert torch.equal(after["1.weight"], before["1.weight"])
    opt = torch.optim.SGD(net.parameters(), lr=0.1, momentum=0.9)
    sched = torch.optim.lr_scheduler.ExponentialLR(opt, gamma=0.9)
    path = os.path.join(tmp_path, "3.pth")
    checkpoint.save_states(path, checkpoint.make_states(net, opt, sched, iteration=3, epoch_n=1))
    net2 = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.Linear(3, 2))
    opt2 = torch.optim.SGD(net2.parameters(), lr=0.5, momentum=0.9)
    sched2 = torch.optim.lr_scheduler.ExponentialLR(opt2, gamma=0.9)
    states = checkpoint.load_pretrained_model(path, [net2, opt2, sched2], ["model_dict", "optimizer_dict", "scheduler_dict"], device="cpu")
    assert states["iteration"] == 3 and states["epoch_n"] == 1
    assert all(torch.equal(a, b) for a, b in zip(net.state_dict().values(), net2.state_dict().values()))


def test_attention_dropout_multiplier_is_the_kernels_hash():
    """ops.attn_dropout_multiplier (what get_attention=True applies in training mode) restates the kernels' counter hash in
    int64 tensor arithmetic; tests/util.keep_scale_host is its bit-exact numpy copy (checked against the kernels on the GPU)."""
    import numpy as np
    import torch
    from spgnn_amd import ops
    from tests.util import keep_scale_host
    for seed, p in ((0, 0.1), (192837465, 0.5), (2 ** 62 - 3, 0.25), (2 ** 63 + 12345, 0.1)):
        E, H = 777, 3
        got = ops.attn_dropout_multiplier(E, H, p, seed, torch.device("cpu")).numpy()
        want = keep_scale_host(seed % 2 ** 64, np.arange(E * H), p).reshape(E, H)
        assert np.array_equal(got, want), (seed, p)


def test_bench_accounting_knows_every_kernel_key():
    """bench.py prices every (kernel, shape) key the ops record: a new kernel whose key it cannot price must not cost the
    driver its JSON line (the accounting is wrapped, and this test names the forms)."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    N, E = 1000, 2990
    keys = [("gat_fwd", N, E, 2, 64, 1, 0, 1), ("gat_bwd_dst", N, E, 2, 64, 1, 0), ("gat_bwd_src", N, E, 2, 64),
            ("lspe_fwd", N, E, 64), ("lspe_bwd_dst", N, E, 64, 1), ("lspe_bwd_src", N, E, 64),
            ("gat_agg_fwd", N, E, 2, 192, 1), ("gat_agg_bwd_dst", N, E, 2, 192), ("gat_agg_bwd_src", N, E, 2, 192),
            ("scores_fwd", N, 1024, 22), ("scores_bwd_w", N, 1024, 22), ("scores_bwd_w_pair", N, 512, 4, 256, 2), ("scores_bwd_x", N, 1024, 22),
            ("act_bwd", N, 1, 64, 4, 0), ("act_bwd_proj", N, 2, 1024, 1, 22), ("masked_ce", N, 22),
            ("spmm_sum", N, E, 64), ("spmm_max_fwd", N, E, 64), ("spmm_max_bwd", N, E, 64),
            ("gat_fwd_bf16", N, E, 2, 64, 1, 0, 1)]
    for k in keys:
        assert bench.algorithmic_bytes(k) > 0, k
    single = bench.gemm_flops(("gemm_nt", N, 512, 768)) + bench.gemm_flops(("gemm_nt", N, 256, 256))
    assert bench.gemm_flops(("gemm_nt_pair", N, 512, 768, N, 256, 256)) == single
    assert bench.gemm_bytes(("gemm_tn_pair", N, 512, 768, N, 256, 256)) == \
        bench.gemm_bytes(("gemm_tn", N, 512, 768)) + bench.gemm_bytes(("gemm_tn", N, 256, 256))
    assert set(bench.GEMM_NAMES) >= {"gemm_nt", "gemm_tn", "gemm_nt_pair", "gemm_tn_pair"}
    assert len(bench.SECONDARY_LEGS) == 3


def _run_bench(argv, env_extra, timeout=180):
    import subprocess
    import sys
    env = dict(os.environ, **env_extra)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=env, capture_output=True, text=True, timeout=timeout)


def test_bench_self_launches_its_ranks_without_a_launcher():
    """`python3 bench.py --gpus 2` with no WORLD_SIZE in the environment (the shape of the driver's N = 1 command) must start
    the two ranks itself, rendezvous on 127.0.0.1, print ONE JSON line from rank 0 and exit 0.  Dry mode:
    the same launch / rendezvous / one-line protocol over gloo on CPU tensors - the GPU rehearsal of the whole step is
    tests/test_hip_models.py::test_bench_two_rank_rehearsal_through_self_launch."""
    import json
    r = _run_bench(["--gpus", "2", "--steps", "3", "--warmup", "1"], {"SPGNN_BENCH_DRY": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1
    assert out["config"]["comm_world_size"] == 2                    # counted by the all-reduce itself
    assert out["config"]["launched_by"] == "bench.py self_launch"


def test_bench_under_torch_distributed_run_keeps_working():
    """The driver's documented N > 1 form: python -m torch.distributed.run ... bench.py --gpus N (dry mode on the CPU)."""
    import json
    import subprocess
    import sys
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, SPGNN_BENCH_DRY="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["comm_world_size"] == 2 and out["config"]["launched_by"] == "torch.distributed.run"


def test_bench_self_launch_returns_a_failed_ranks_exit_code():
    """Without a GPU the real ranks stop at "needs a ROCm GPU": the launcher must come back with a non-zero code instead of
    hanging on the surviving ranks (here every rank fails; the one-rank-dies case is the same poll loop)."""
    if torch.cuda.is_available():
        pytest.skip("needs a box without a GPU")
    r = _run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0"], {"SPGNN_BENCH_REHEARSAL": "1"}, timeout=120)
    assert r.returncode != 0
    assert "needs a ROCm GPU" in r.stderr


def test_bench_flat_scalars_for_the_drivers_record():
    """The driver keeps scalar members of `roofline` / `config` only: flatten_for_driver must repeat the nested K1-K3 figure."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    out = {"config": {}, "roofline": {"bound": "mfma", "frac": 0.1, "hbm": {"frac": 0.6, "own_frac": 0.55, "ms_per_step": 1.1, "survey_bytes_per_step": 5.2e9,
                                                                            "own_bytes_per_step": 4.9e9, "achieved": 4700.0, "traffic": 5.0e9,
                                                                            "layer_edges_per_s": 1.4e9, "launches_per_step": 12}},
           "gemm": {"ms_per_step": 3.3}, "message_passing": {"ms_per_step": 1.7}, "step_ms": {"median": 5.2}, "copy_bandwidth": {"GBps": 4900.0}}
    bench.flatten_for_driver(out)
    r, c = out["roofline"], out["config"]
    assert r["hbm_frac"] == 0.6 and r["hbm_ms_per_step"] == 1.1 and r["hbm_survey_bytes"] == 5.2e9 and r["hbm_traffic"] == 5.0e9
    assert c["gemm_ms_per_step"] == 3.3 and c["message_passing_ms_per_step"] == 1.7 and c["step_ms_median"] == 5.2
    assert all(not isinstance(v, (dict, list)) for k, v in r.items() if k.startswith("hbm_"))


def _free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _load_bench():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    return bench


def _strict_loads(txt):
    def refuse(tok):
        raise ValueError("non-finite constant in the line: " + tok)
    return json.loads(txt, parse_constant=refuse)


def test_bench_driver_line_is_short_flat_and_strict():
    """The ONE stdout line the driver parses (a 31 KB line left BENCH_r05.parsed null).  Built here from the
    recorded full object of a real default run (tests/golden/bench_full_object_r05e.json: headline + five secondary legs +
    batch cycle + single-tree leg): one line, far below 12 KB, strict JSON, the contract's keys, and `config` / `roofline` /
    `cpu_baseline` hold scalars only - no prose members, no nested legs."""
    bench = _load_bench()
    full = json.load(open(os.path.join(ROOT, "tests", "golden", "bench_full_object_r05e.json")))
    assert len(json.dumps(full)) > 25000                       # the object that broke the driver's parser
    # what the new legs add to a run's object
    full["roofline"]["hbm"]["in_step_frac"] = 0.40
    full["roofline"]["hbm"]["in_step_ms_per_step"] = 1.63
    bench.flatten_for_driver(full)
    full["loss"] = float("nan")                                # a diverged run must still give a strict line
    full["config"]["capture_error"] = "x" * 300
    txt = bench.driver_line(full)
    assert "\n" not in txt and len(txt) < bench.LINE_LIMIT == 12288 and len(txt) < 6000, len(txt)
    line = _strict_loads(txt)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["loss"] is None and line["vs_baseline"] is None and line["higher_is_better"] is True
    assert line["value"] == pytest.approx(full["value"], rel=1e-8) and line["ms_per_step"] == pytest.approx(full["ms_per_step"], rel=1e-8)
    for part in ("config", "roofline", "cpu_baseline"):
        for k, v in line[part].items():
            assert not isinstance(v, (dict, list)), (part, k)
            assert not isinstance(v, str) or len(v) <= 118, (part, k, len(v))
            assert k not in ("what", "note", "how", "measured_in", "traffic_source"), (part, k)
    assert "secondary" not in line and "composite" not in line and "gemm" not in line
    r, c, b = line["roofline"], line["config"], line["cpu_baseline"]
    assert r["bound"] == "mfma" and r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-4)
    assert r["hbm_frac"] == pytest.approx(0.5859, abs=1e-3) and r["hbm_in_step_frac"] == 0.40 and r["traffic"] > 0
    assert c["sec_st_pgat_spgnn_3_f32_64_ms"] > 0 and c["sec_single_tree_forward_captured_us"] > 0 and c["batch_cycle_over_steady"] > 0
    assert c["workload"].startswith("st_pgat_spgnn_3") and c["step_ms_median"] > 0
    assert b["kind"] == "port" and b["cores"] >= 1 and b["value"] > 0 and isinstance(b["sample"], str)


def test_bench_driver_line_sheds_optional_scalars_before_it_outgrows_the_limit():
    bench = _load_bench()
    full = json.load(open(os.path.join(ROOT, "tests", "golden", "bench_full_object_r05e.json")))
    for i in range(600):                                       # a future builder adds legs without thinking of the driver
        full["config"][f"sec_future_leg_{i:04d}_ms"] = 1.0 + i
    txt = bench.driver_line(full)
    assert len(txt) <= bench.LINE_LIMIT
    line = _strict_loads(txt)
    assert "value" in line and "frac" in line["roofline"] and "workload" in line["config"] and "value" in line["cpu_baseline"]


def test_bench_prints_exactly_one_stdout_line_and_writes_the_detail_file(tmp_path):
    """End to end through main() in dry mode (no GPU): stdout is ONE line of strict JSON and nothing after it."""
    r = _run_bench(["--gpus", "2", "--steps", "3", "--warmup", "1"], {"SPGNN_BENCH_DRY": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and len(lines[0]) < 12288
    _strict_loads(lines[0])
    # write_detail falls back instead of failing the run
    bench = _load_bench()
    got = bench.write_detail({"a": 1}, str(tmp_path / "no_such_dir" / "x.json"))
    assert got is not None and os.path.exists(got)
    os.remove(got)
    got = bench.write_detail({"a": 1}, str(tmp_path / "d.json"))
    assert got == str(tmp_path / "d.json") and json.load(open(got)) == {"a": 1}


def test_bench_launcher_never_touches_torch_cuda_and_runs_forced_with_one_rank():
    """the launcher parent counts GPUs from the KFD topology in /sys, not through torch.cuda (whose
    device_count() can fall back to hipGetDeviceCount and bring up the runtime in a process that then starts its ranks).
    And SPGNN_BENCH_FORCE_LAUNCH=1 takes the launcher path with ONE rank (dry mode here; RCCL on the GPU box:
    tests/test_hip_models.py::test_bench_real_launcher_and_rccl_with_one_rank)."""
    import inspect
    bench = _load_bench()
    for fn in (bench.self_launch, bench.visible_gpus):
        src = inspect.getsource(fn)
        assert "torch.cuda" not in src.split('"""')[2] and "hipGetDeviceCount" not in src.split('"""')[2], fn.__name__
    n = bench.visible_gpus()
    assert n is None or (isinstance(n, int) and n >= 0)
    if not torch.cuda.is_available():
        assert n in (None, 0)
    r = _run_bench(["--gpus", "1", "--steps", "2", "--warmup", "1"], {"SPGNN_BENCH_DRY": "1", "SPGNN_BENCH_FORCE_LAUNCH": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    out = _strict_loads(lines[0])
    assert out["n_gpus"] == 1 and out["config"]["comm_world_size"] == 1 and out["config"]["launched_by"] == "bench.py self_launch"


def test_train_step_always_exchange_issues_the_collective_with_one_rank():
    """TrainStep(always_exchange=True) in a ONE-rank process group: _reduce calls dist.all_reduce on the flat bucket (a sum over
    one rank: the identity), so the multi-rank step runs - and gives what the plain step gives."""
    from spgnn_amd.train import FlatBucket, TrainStep
    port = _free_port()
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    try:
        calls = []
        real = dist.all_reduce

        def spy(t, *a, **k):
            calls.append(t.numel())
            return real(t, *a, **k)
        lin = torch.nn.Linear(6, 3)
        ts = TrainStep.__new__(TrainStep)
        ts.bucket, ts.pg, ts.world = FlatBucket(list(lin.parameters())), None, 1
        ts.bucket.flat_grad.normal_()
        before = ts.bucket.flat_grad.clone()
        dist.all_reduce = spy
        try:
            ts.exchange = False
            ts._reduce(ts.bucket.loss_slot)
            assert calls == []
            ts.exchange = True
            ts._reduce(ts.bucket.loss_slot)
        finally:
            dist.all_reduce = real
        assert calls == [ts.bucket.flat_grad.numel()] and torch.equal(ts.bucket.flat_grad, before)
    finally:
        dist.destroy_process_group()


def test_loss_rows_capacity_rule():
    """train.TrainStep._loss_rows_cap (host arithmetic): mean + 8 sigma of the kept count, 256-row tiles, 15 % headroom on a batch
    arena, and 0 (= run dense) when the list would not be clearly shorter than the node count."""
    import types

    import torch

    from spgnn_amd.train import TrainStep
    ts = TrainStep.__new__(TrainStep)
    ts.sampling_rate, ts._rows_headroom, ts._rows_gen = 0.15, 1.0, 0
    big = types.SimpleNamespace()
    p = torch.full((76410,), 0.15)
    p[::7] = 1.0                                                   # ~14 % labelled
    p[-300:] = -1.0                                                # pad rows never count
    mu = float(p.clamp(min=0).sum())
    var = float((p.clamp(min=0) * (1 - p.clamp(min=0))).sum())
    cap = ts._loss_rows_cap(big, p)
    assert cap % 256 == 0 and mu + 8 * var ** 0.5 <= cap < mu + 8 * var ** 0.5 + 32 + 256 and cap < 0.35 * 76410
    assert ts._loss_rows_cap(big, p * 0 + 1.0) == cap              # cached on the graph: one host read per graph
    arena = types.SimpleNamespace(_stable_storage=True)
    assert ts._loss_rows_cap(arena, p) > cap                       # headroom for the other batches of the size class
    small = types.SimpleNamespace()
    assert ts._loss_rows_cap(small, torch.full((150,), 0.15)) == 0   # 256 slots for 150 nodes: no gain, run dense
    dense = types.SimpleNamespace()
    assert ts._loss_rows_cap(dense, torch.ones(50000)) == 0        # every node labelled: the list would be the node set
    ts._rows_headroom, ts._rows_gen = 1.5, 1                       # after an overflow (check_loss_rows): recomputed, with more room
    assert ts._loss_rows_cap(big, p) > 1.3 * cap
