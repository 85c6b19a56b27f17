"""Batch arenas (spgnn_amd/arena.py): the reference's loader-batch cycle - GCN_STEPS optimizer steps on every freshly
built batch (job_runner.py:1870-1920) - as replays of ONE captured step per size class.  CPU tests pin the pad structure
against the reference's edge rule; GPU tests run two different batches through one captured step and compare with eager
steps on the same padded graphs and on the unpadded batches."""
import copy

import numpy as np
import pytest
import torch

from spgnn_amd import graph as G
from spgnn_amd.arena import _pad_graph, size_class
from tests.util import rel_err


@pytest.mark.parametrize("n_pad,m", [(1, 0), (5, 0), (5, 2), (7, 6), (300, 299), (300, 17)])
def test_pad_component_follows_the_reference_edge_rule(n_pad, m):
    """The pad's index arrays equal what the reference's rule (edges_from_adj: off-diagonal entries by (u, v), then the
    self loops; job_runner.py:1779-1801) gives on the pad's adjacency matrix, through the same stable COO -> CSC / CSR."""
    adj = np.eye(n_pad, dtype=np.uint8)
    for j in range(m):
        adj[j, j + 1] = adj[j + 1, j] = 1
    src, dst = G.edges_from_adj(adj, add_self_loops=True)
    pad = _pad_graph(n_pad, m)
    assert np.array_equal(pad["src"], src) and np.array_equal(pad["dst"], dst)
    ref = G.build_csc_numpy(src, dst, n_pad)
    for k, v in ref.items():
        assert np.array_equal(pad[k], v), k
    deg = pad["indptr"][1:] - pad["indptr"][:-1]
    assert deg.min() >= 1 and deg.max() <= 3 and src.shape[0] == n_pad + 2 * m


def test_size_class_arithmetic():
    """Tree batches (E = 3N - 2B) of one tree count share a class per node granule; the pad always fits: its edge count has
    the right parity and at most n_pad - 1 path edges.  Other graphs get a class from their own edge count."""
    rng = np.random.default_rng(0)
    for _ in range(200):
        B = int(rng.integers(1, 80))
        N = int(rng.integers(B * 21, B * 300))
        E = 3 * N - 2 * B
        n_cap, e_cap = size_class(N, E, B, 256)
        n_pad = n_cap - N
        assert 1 <= n_pad <= 256 and n_cap % 256 == 0
        assert e_cap == 3 * n_cap - 2 * (B + 1)                       # a function of (n_cap, B) only: the class is shared
        twice_m = e_cap - E - n_pad
        assert twice_m >= 0 and twice_m % 2 == 0 and twice_m // 2 == n_pad - 1
    # a graph that is not a forest of B trees (extra edges): still a valid pad
    n_cap, e_cap = size_class(1000, 3 * 1000 - 2 * 4 + 10, 4, 256)
    n_pad = n_cap - 1000
    twice_m = e_cap - (3 * 1000 - 8 + 10) - n_pad
    assert twice_m % 2 == 0 and 0 <= twice_m // 2 <= n_pad - 1


# ---- GPU ---------------------------------------------------------------------------------------------------------------

def _build(name, seed=0):
    from spgnn_amd import models
    from spgnn_amd.configs import get_config
    cfg = get_config(name)
    torch.manual_seed(seed)
    model = models.build_model(cfg.MODEL).cuda()
    model.init(None)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith("bias"):
                p.normal_(0, 0.05)
    model.set_gcn_only()
    return cfg, model


@pytest.mark.gpu
@pytest.mark.parametrize("name,bf16", [("st_pgat_spgnn_3", False), ("st_gat_3", False), ("st_gcn_3", False), ("st_gin_3", False),
                                       ("st_sage_3", False), ("st_gat_6", True)])
def test_two_batches_through_one_captured_step_equal_eager_runs(name, bf16):
    """batch A is captured (3 warm-up steps + capture), batch B - other trees, other N and E, same size
    class - is a copy into the arena plus replays of the SAME two HIP graphs.  Parameters after 4 + 4 steps equal (i) eager
    steps on the same padded graphs (separate arena objects, separate storage; same arithmetic: 1e-6) and (ii) eager steps
    on the unpadded batches (padding moves only fp32 summation order and power-of-two scales: 1e-5)."""
    from spgnn_amd import ops, synthetic
    from spgnn_amd.arena import BatchArena
    from spgnn_amd.configs import class_weight_list
    from spgnn_amd.train import TrainStep
    from spgnn_amd import models as _models
    cfg, model = _build(name, seed=5)
    if bf16:
        _models.set_storage_dtype(model, torch.bfloat16)
    model.eval()                                                       # deterministic arithmetic (no dropout masks)
    w = class_weight_list(cfg.CLASS_WEIGHTS)
    ga = synthetic.make_batch(6, rank=3, device="cuda", pos_enc_dim=getattr(cfg, "POS_ENC_DIM", None))
    gb = synthetic.make_batch(6, rank=4, device="cuda", pos_enc_dim=getattr(cfg, "POS_ENC_DIM", None))
    assert ga.number_of_nodes() != gb.number_of_nodes()
    GRAN = 2048                                                        # both batches (~900 nodes) in one class
    assert BatchArena.class_key(ga, GRAN) == BatchArena.class_key(gb, GRAN)
    m_pad, m_raw = copy.deepcopy(model), copy.deepcopy(model)
    ts = TrainStep(model, w, 1.0, 1e-3, 0.9)                           # sampling rate 1: the mask does not depend on the RNG
    la = ts.run_batch(ga, 4, granule=GRAN)
    lb = ts.run_batch(gb, 4, granule=GRAN)
    assert len(ts._captures) == 1 and len(ts._arenas) == 1 and next(iter(ts._arenas.values())).loads == 2
    n = ts.bucket.numel
    # (i) eager on the same padded graphs
    ts_p = TrainStep(m_pad, w, 1.0, 1e-3, 0.9)
    pa, pb = BatchArena(ga, GRAN).load(ga), BatchArena(gb, GRAN).load(gb)
    for _ in range(4):
        lpa = ts_p.step(pa)
    for _ in range(4):
        lpb = ts_p.step(pb)
    assert rel_err(la, lpa) < 1e-6 and rel_err(lb, lpb) < 1e-6
    assert rel_err(ts.bucket.flat_param[:n], ts_p.bucket.flat_param[:n]) < 1e-6
    # (ii) eager on the unpadded batches
    ts_r = TrainStep(m_raw, w, 1.0, 1e-3, 0.9)
    for _ in range(4):
        lra = ts_r.step(ga)
    for _ in range(4):
        lrb = ts_r.step(gb)
    # (bf16 rows: an fp32 summation-order change can move a stored value across a rounding boundary - 2^-8 relative; SAGE's
    # max-pool routing can flip on near-ties: both looser, as in tests/test_hip_bf16.py / test_hip_models.py)
    tol = 2e-2 if bf16 else 5e-3 if cfg.KIND == "sage" else 1e-5
    assert rel_err(la, lra) < tol and rel_err(lb, lrb) < tol
    assert rel_err(ts.bucket.flat_param[:n], ts_r.bucket.flat_param[:n]) < tol
    # the arena's real rows give the unpadded forward; pad rows never reach the loss
    ag = ts.arena_graph(gb, GRAN)
    with torch.no_grad():
        lo_pad, lo_raw = model(ag)[0], model(gb)[0]
    assert rel_err(lo_pad[:gb.number_of_nodes()], lo_raw) < (2e-2 if bf16 else 1e-6 if cfg.KIND != "sage" else 1e-5)
    p = ts._sampling(ag)
    assert bool((p[gb.number_of_nodes():] == -1).all()) and bool((p[:gb.number_of_nodes()] == 1).all())
    ops.DROPOUT_SEED_OFFSET = None


@pytest.mark.gpu
def test_batch_cycle_with_dropout_and_two_size_classes():
    """Training mode (dropout and node sampling on), three loader batches over two size classes: one capture per class,
    finite losses, the loss normaliser counts real nodes only, and a learning-rate change reaches every capture."""
    from spgnn_amd import ops, synthetic
    from spgnn_amd.configs import class_weight_list
    from spgnn_amd.train import TrainStep
    cfg, model = _build("st_pgat_spgnn_3", seed=6)
    model.train()
    w = class_weight_list(cfg.CLASS_WEIGHTS)
    ts = TrainStep(model, w, 1.0, 1e-3, 0.9, seed=3)
    g1 = synthetic.make_batch(4, rank=0, device="cuda", pos_enc_dim=cfg.POS_ENC_DIM)
    g2 = synthetic.make_batch(4, rank=1, device="cuda", pos_enc_dim=cfg.POS_ENC_DIM)
    g3 = synthetic.make_batch(7, rank=2, device="cuda", pos_enc_dim=cfg.POS_ENC_DIM)      # more trees: another class
    losses = []
    for g in (g1, g2, g3, g1):
        losses.append(float(ts.run_batch(g, 5, granule=1024)))
        wt = torch.tensor(w, device="cuda")[g.ndata["y"]].sum()                            # rate 1: every real node is kept
        assert rel_err(ts.bucket.wsum_slot, wt.reshape(1)) < 1e-6
    assert all(np.isfinite(losses)) and len(ts._captures) == 2
    ts.set_lr(0.0)                                                                          # frozen: parameters stay put under replay
    before = ts.bucket.flat_param.clone()
    ts.run_batch(g2, 3, granule=1024)
    ts.run_batch(g3, 3, granule=1024)
    assert torch.equal(before[:ts.bucket.numel] + 0, ts.bucket.flat_param[:ts.bucket.numel]) or \
        rel_err(ts.bucket.flat_param[:ts.bucket.numel], before[:ts.bucket.numel]) < 1e-7   # momentum tail with lr 0 moves nothing
    ts.max_arenas = 1                                                                       # LRU over size classes: a NEW class evicts the others
    g4 = synthetic.make_batch(5, rank=3, device="cuda", pos_enc_dim=cfg.POS_ENC_DIM)
    ts.run_batch(g4, 4, granule=1024)
    assert len(ts._arenas) == 1 and len(ts._captures) == 1
    assert np.isfinite(float(ts.run_batch(g1, 4, granule=1024)))                           # its class is simply built again
    ops.DROPOUT_SEED_OFFSET = None


@pytest.mark.gpu
def test_random_batch_sequence_through_arenas_tracks_unpadded_training():
    """Twenty loader batches of random tree counts and sizes through run_batch (size classes come and go under a small LRU,
    captures are reused and rebuilt) against plain eager steps on the unpadded batches: the same training trajectory."""
    from spgnn_amd import ops, synthetic
    from spgnn_amd.configs import class_weight_list
    from spgnn_amd.train import TrainStep
    cfg, model = _build("st_pgat_spgnn_3", seed=21)
    model.eval()
    w = class_weight_list(cfg.CLASS_WEIGHTS)
    ref = copy.deepcopy(model)
    ts, ts_ref = TrainStep(model, w, 1.0, 1e-3, 0.9), TrainStep(ref, w, 1.0, 1e-3, 0.9)
    ts.max_arenas = 3
    rng = np.random.default_rng(5)
    classes = set()
    for b in range(20):
        trees = int(rng.integers(2, 6))
        samples = synthetic.synthetic_trees(trees, rank=50 + b, n_lo=40, n_hi=90)
        g = synthetic.batch_from_samples(samples, "cuda", cfg.POS_ENC_DIM)
        steps = int(rng.integers(4, 7))
        la = ts.run_batch(g, steps, granule=128)
        for _ in range(steps):
            lr_ = ts_ref.step(g)
        classes.add((g.batch_size, (g.number_of_nodes() // 128 + 1) * 128))
        assert rel_err(la, lr_) < 1e-4, (b, float(la), float(lr_))
    n = ts.bucket.numel
    assert len(classes) > 3 and len(ts._arenas) <= 3
    assert rel_err(ts.bucket.flat_param[:n], ts_ref.bucket.flat_param[:n]) < 1e-5
    ops.DROPOUT_SEED_OFFSET = None


@pytest.mark.gpu
def test_run_batches_prefetches_the_next_assembly_and_equals_run_batch():
    """The whole loader loop (TrainStep.run_batches: batch i + 1 assembled on a side stream under batch i's replays) gives
    the parameters of the same batches through run_batch one after the other - the side stream moves WHEN the assembly runs, never
    what it computes - also when the staged blocks are reused by the next assembly."""
    from spgnn_amd import data, ops, synthetic
    from spgnn_amd.configs import class_weight_list
    from spgnn_amd.train import TrainStep
    cfg, model = _build("st_pgat_spgnn_3", seed=8)
    model.train()
    w = class_weight_list(cfg.CLASS_WEIGHTS)
    ref = copy.deepcopy(model)
    batches = [synthetic.synthetic_trees(5, rank=70 + b, n_lo=60, n_hi=120) for b in range(7)]
    asm = lambda s_: data.assemble_batch(s_, "cuda", cfg.POS_ENC_DIM)
    ts = TrainStep(model, w, cfg.SAMPLING_RATE, 1e-3, 0.9, seed=4)
    torch.manual_seed(77)                                              # the dropout seeds a capture bakes in are host draws
    losses = ts.run_batches(batches, 6, asm, granule=1024)
    ts_ref = TrainStep(ref, w, cfg.SAMPLING_RATE, 1e-3, 0.9, seed=4)
    torch.manual_seed(77)
    losses_ref = [ts_ref.run_batch(asm(b), 6, granule=1024).clone() for b in batches]
    torch.cuda.synchronize()
    errs = [rel_err(a, b) for a, b in zip(losses, losses_ref)]
    n = ts.bucket.numel
    assert len(losses) == 7 and max(errs) < 1e-6, errs                 # (atomic float adds in the backward: not bit for bit)
    assert rel_err(ts.bucket.flat_param[:n], ts_ref.bucket.flat_param[:n]) < 1e-6
    assert ts.run_batches([], 3, asm) == []
    ops.DROPOUT_SEED_OFFSET = None
