"""train.TrainStep(loss_rows_only=True): the part of a training step behind the last aggregation - output-layer projection, head
mean, classifier, their backward products - on the rows the mask keeps only (reference job_runner.py:1896-1900:
``F.cross_entropy(pre[mask], y[mask], weight=w)``; no other row reaches the loss or a gradient).  The list kernels against torch,
the step against the dense step (same draws: same loss and parameters up to fp32 summation order) and against the oracle."""
import copy
import warnings

import numpy as np
import pytest
import torch

from spgnn_amd import _capi, models, ops, synthetic
from spgnn_amd.configs import class_weight_list, get_config
from spgnn_amd.train import TrainStep
from tests.util import rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N,rate,cap", [(1, 1.0, 256), (255, 0.3, 256), (256, 0.0, 256), (5000, 0.15, 1024), (76410, 0.27, 22016),
                                        (3000, 0.9, 512)])
def test_row_list_is_the_mask_in_node_order(N, rate, cap):
    g = torch.Generator(device="cuda").manual_seed(N)
    p = torch.full((N,), rate, device="cuda")
    p[::7] = 1.0                                                   # "labelled" nodes: always kept
    p[3::11] = -1.0                                                # pad nodes of an arena: never
    draws = torch.rand(N, device="cuda", generator=g)
    want = torch.nonzero(draws < p).flatten().to(torch.int32)
    r = ops.loss_rows(p, draws, 0, cap)
    cnt, over = (int(v) for v in r.cnt.tolist())
    assert over == (1 if want.numel() > cap else 0)
    k = min(want.numel(), cap)
    assert cnt == k and torch.equal(r.idx[:k], want[:k]) and bool((r.idx[k:] == 0).all())
    inv = torch.full((N,), -1, dtype=torch.int32, device="cuda")
    inv[want[:k].long()] = torch.arange(k, dtype=torch.int32, device="cuda")
    assert torch.equal(r.inv[:N], inv)
    x = torch.randn(N, 8, device="cuda")
    xc = ops.gather_rows(x, r)
    assert xc.shape == (cap, 8) and torch.equal(xc[:k], x[want[:k].long()]) and bool((xc[k:] == 0).all())
    back = ops.expand_rows(xc, r)
    ref = torch.zeros_like(x)
    ref[want[:k].long()] = x[want[:k].long()]
    assert torch.equal(back, ref)


def test_row_list_from_the_counter_hash_is_the_loss_kernels_mask():
    """``draws`` None: the list kernel and spgnn_masked_ce_step draw from the same hash - the loss over the listed rows equals the
    dense loss with the kernel's own mask, and so do the logit gradients."""
    N, C = 4000, 22
    torch.manual_seed(3)
    logits = torch.randn(N, C, device="cuda")
    y = torch.randint(0, C, (N,), device="cuda")
    y[torch.rand(N, device="cuda") < 0.85] = 0                    # most nodes unlabelled, as in the airway trees
    p = torch.full((N,), 0.15, device="cuda")
    p[y > 0] = 1.0
    w = torch.rand(C, device="cuda") + 0.5
    dense_in = logits.clone().requires_grad_(True)
    num, den = ops.masked_ce_sums(dense_in, y, None, p, w, draw_seed=12345)
    num.backward()
    r = ops.loss_rows(p, None, 12345, 2048)
    k = int(r.cnt[0])
    assert 0 < k < 2048 and int(r.cnt[1]) == 0
    lc = ops.gather_rows(torch.cat([logits, logits.new_zeros(N, 2)], 1), r)[:, :C].contiguous().requires_grad_(True)
    num_r, den_r = ops.masked_ce_sums(lc, y, None, p, w, draw_seed=12345, rows=r)
    num_r.backward()
    assert rel_err(num_r, num) < 1e-6 and rel_err(den_r, den) < 1e-6
    kept = r.idx[:k].long()
    assert torch.equal(lc.grad[:k], dense_in.grad[kept]) and bool((lc.grad[k:] == 0).all())
    others = torch.ones(N, dtype=torch.bool, device="cuda")
    others[kept] = False
    assert bool((dense_in.grad[others] == 0).all())               # the rows outside the list carry no gradient at all


def test_overflow_raises_the_flag_and_poisons_the_loss():
    N, C = 1000, 22
    p = torch.ones(N, device="cuda")
    r = ops.loss_rows(p, torch.zeros(N, device="cuda"), 0, 256)
    assert r.cnt.tolist() == [256, 1]
    logits = torch.randn(256, C, device="cuda")
    num, den = ops.masked_ce_sums(logits, torch.zeros(N, dtype=torch.int64, device="cuda"), None, p, torch.ones(C, device="cuda"), rows=r)
    assert bool(torch.isnan(num)) and bool(torch.isnan(den))


def _model(name, seed):
    cfg = get_config(name)
    torch.manual_seed(seed)
    m = models.build_model(cfg.MODEL).cuda()
    m.init(None)
    m.set_gcn_only()
    return cfg, m


@pytest.mark.parametrize("mode", [True, "backward"])
@pytest.mark.parametrize("name,trees,train_mode", [("st_pgat_spgnn_3", 6, False), ("st_pgat_spgnn_3", 64, True), ("st_pgat_spgnnnl_3", 6, False),
                                                   ("st_gat_3", 6, False)])
def test_loss_rows_step_equals_the_dense_step(name, trees, train_mode, mode):
    """Four optimizer steps with and without the switch from the same parameters, the same mask stream and (train mode) the same
    hash dropout masks: losses and parameters agree to fp32 summation order.  st_gat_3's head (no activation on the output layer:
    the linear-mean form) takes the list through a differentiable gather behind its aggregation (ops.take_loss_rows) when the
    forward pass may use it, and runs unchanged under ``mode`` "backward": dense forward (the step's model call returns every
    row), the list only in the backward products of the heads that have them."""
    cfg, model = _model(name, 21)
    model.train(train_mode)
    dense = copy.deepcopy(model)
    w = class_weight_list(cfg.CLASS_WEIGHTS)
    g = synthetic.make_batch(trees, rank=2, device="cuda", pos_enc_dim=getattr(cfg, "POS_ENC_DIM", None))
    ts_r = TrainStep(model, w, cfg.SAMPLING_RATE, 1e-3, 0.9, seed=5, loss_rows_only=mode)
    ts_d = TrainStep(dense, w, cfg.SAMPLING_RATE, 1e-3, 0.9, seed=5)
    if train_mode:                                 # the counter-hash streams (mask and dropout): what a captured step uses
        for ts in (ts_r, ts_d):
            ts._seed_ctr = torch.zeros((1,), dtype=torch.int64, device="cuda")
            ts._use_default_rng = True
    for i in range(4):
        torch.manual_seed(100 + i)                 # the layers draw their dropout seeds from torch's CPU generator: same for both
        lr_ = ts_r.step(g)
        torch.manual_seed(100 + i)
        ld_ = ts_d.step(g)
        assert rel_err(lr_, ld_) < 2e-6, (float(lr_), float(ld_))
    takes = mode is True or name != "st_gat_3"
    assert (ts_r._rows_cnt is not None and int(ts_r._rows_cnt[0]) > 0) if takes else True
    n = ts_r.bucket.numel
    assert rel_err(ts_r.bucket.flat_param[:n], ts_d.bucket.flat_param[:n]) < 2e-6
    ts_r.check_loss_rows()
    if takes:
        N = g.number_of_nodes()
        assert int(ts_r._rows_cnt[0]) < 0.5 * N                   # the list is what makes the step cheaper: well under half the nodes


@pytest.mark.parametrize("mode", [True, "backward"])
def test_captured_loss_rows_step_replays_with_fresh_masks(mode):
    """run_batch (arena + HIP-graph replays) with the switch: the replays draw a new mask each (the list kernel reads the step
    counter from device memory like the loss kernel), the losses follow the dense captured step's."""
    cfg, model = _model("st_pgat_spgnn_3", 4)
    model.train(True)
    dense = copy.deepcopy(model)
    w = class_weight_list(cfg.CLASS_WEIGHTS)
    g = synthetic.make_batch(8, rank=1, device="cuda", pos_enc_dim=cfg.POS_ENC_DIM)
    ts_r = TrainStep(model, w, cfg.SAMPLING_RATE, 1e-3, 0.9, seed=9, loss_rows_only=mode)
    ts_d = TrainStep(dense, w, cfg.SAMPLING_RATE, 1e-3, 0.9, seed=9)
    counts = []
    for i, k in enumerate((4, 1, 1, 1)):
        torch.manual_seed(200 + i)                 # (the dropout seeds a capture freezes come from torch's CPU generator)
        lr_ = ts_r.run_batch(g, k, granule=512)
        torch.manual_seed(200 + i)
        ld_ = ts_d.run_batch(g, k, granule=512)
        assert rel_err(lr_, ld_) < 1e-5, (float(lr_), float(ld_))
        counts.append(int(ts_r._rows_cnt[0]))
    assert len(set(counts)) > 1                                    # a fresh mask per replay
    n = ts_r.bucket.numel
    assert rel_err(ts_r.bucket.flat_param[:n], ts_d.bucket.flat_param[:n]) < 1e-5
    ts_r.check_loss_rows()


@pytest.mark.parametrize("mode,trees", [(False, 64), (True, 64), ("backward", 64), ("backward", 512)])
def test_loss_rows_step_matches_the_oracle_at_64_trees(mode, trees):
    """The TRAINING STEP's own front half (train.TrainStep._front: the path the bench times - fused loss head, deferred sums, side
    stream) against the CPU oracle: the loss and every parameter gradient at TRAIN_BATCH_SIZE, dense (mode False) and with the
    row list, against ``F.cross_entropy(pre[mask], y[mask], weight=w)`` over the full forward (reference
    job_runner.py:1896-1900), fp32 and fp64, with the gradient rule of tests/test_hip_parity_at_size.py.
    ("backward", 512): VERDICT r5 item 7 - the mode INTEGRATION.md recommends, at the headline batch (fp32 oracle only)."""
    from oracle import dgl_cpu as O
    from tests.test_hip_parity_at_size import _assert_gradients, _build, _oracle
    cfg, model = _build("st_pgat_spgnn_3", seed=11)
    g = synthetic.make_batch(trees, rank=0, device="cuda", pos_enc_dim=cfg.POS_ENC_DIM)
    N = g.number_of_nodes()
    w = torch.tensor(class_weight_list(cfg.CLASS_WEIGHTS))
    y = g.ndata["y"]
    draws = torch.rand(N, generator=torch.Generator().manual_seed(5))
    mask = draws < torch.where(y.cpu() != 0, torch.tensor(1.0), torch.tensor(cfg.SAMPLING_RATE))
    ts = TrainStep(model, w.tolist(), cfg.SAMPLING_RATE, 1e-3, 0.9, loss_rows_only=mode)
    num = ts._front(g, draws.cuda()).clone()              # gradient SUMS of the numerator in p.grad, the weight sum in the bucket
    den = ts.bucket.wsum_slot.clone()
    if mode:
        assert int(ts._rows_cnt[0]) == int(mask.sum()) and int(ts._rows_cnt[1]) == 0
    for p in model.parameters():
        if p.grad is not None:
            p.grad = p.grad / den
    refs, sd = _oracle(cfg, model, g, grad=True)
    ref_loss = O.masked_weighted_ce(refs[0], y.cpu(), mask, w)
    ref_loss.backward()
    if trees <= 64:
        refs64, sd64 = _oracle(cfg, model, g, dtype=torch.float64, grad=True)
        O.masked_weighted_ce(refs64[0], y.cpu(), mask, w.double()).backward()
    else:
        sd64 = sd                                        # no fp64 leg at this size: the plain rule (1e-4, or within 1e-7 of the largest gradient)
    assert rel_err(num / den, ref_loss) < 1e-5
    worst = _assert_gradients(model, sd, sd64, 1e-4)
    print(f"training step front half (loss_rows_only={mode}), {trees} trees: loss {rel_err(num / den, ref_loss):.2e}, worst non-tiny gradient normwise error vs the fp32 oracle {worst:.2e}")


@pytest.mark.parametrize("mode", [True, "backward"])
def test_list_overflow_loses_the_step_not_the_parameters(mode):
    """A draw beyond the list's capacity: the step's loss is NaN (the dense loss of a "backward" step included), the guarded
    optimizer kernel does not apply it - parameters and momentum as before - and check_loss_rows() warns, enlarges the
    capacities and lets training go on."""
    cfg, model = _model("st_pgat_spgnn_3", 2)
    model.eval()
    w = class_weight_list(cfg.CLASS_WEIGHTS)
    g = synthetic.make_batch(6, rank=2, device="cuda", pos_enc_dim=cfg.POS_ENC_DIM)
    ts = TrainStep(model, w, cfg.SAMPLING_RATE, 1e-3, 0.9, seed=5, loss_rows_only=mode)
    g.__dict__["_loss_rows_cap"] = {(cfg.SAMPLING_RATE, 0): 512}          # (forced: the ~27 % of ~900 nodes fit)
    l0 = ts.step(g)                                                        # a good step first (it also passes the first-step check)
    assert bool(torch.isfinite(l0)) and ts.check_loss_rows() == 0
    n = ts.bucket.numel
    before, mom = ts.bucket.flat_param[:n].clone(), ts.bucket.flat_mom[:n].clone()
    p = ts._sampling(g)
    keep = p.clone()
    p.fill_(1.0)                                                           # every node kept: 900 rows into 512 slots
    l1 = ts.step(g)
    assert bool(torch.isnan(l1))
    assert torch.equal(ts.bucket.flat_param[:n], before) and torch.equal(ts.bucket.flat_mom[:n], mom)
    with pytest.warns(RuntimeWarning, match="row list"):
        assert ts.check_loss_rows() == 1
    assert ts.check_loss_rows() == 1 and int(ts._rows_cnt[1]) == 0         # flag cleared, the counter stays
    p.copy_(keep)
    l2 = ts.step(g)                                                        # capacities recomputed (1.5 x the headroom): training goes on
    assert bool(torch.isfinite(l2)) and not torch.equal(ts.bucket.flat_param[:n], before)


def test_list_aware_traversals_equal_the_expanded_copy(monkeypatch):
    """The output layer's two backward traversals reading the listed g_z rows through ``inv`` (zero rows skipped) against the same
    traversals on the expanded node-order copy: the same sums in the same order - bit-identical parameters after two steps."""
    cfg, model = _model("st_pgat_spgnn_3", 8)
    model.eval()
    other = copy.deepcopy(model)
    w = class_weight_list(cfg.CLASS_WEIGHTS)
    g = synthetic.make_batch(12, rank=3, device="cuda", pos_enc_dim=cfg.POS_ENC_DIM)
    res = []
    for aware, m in ((True, model), (False, other)):
        monkeypatch.setattr(ops, "LIST_AWARE_TRAVERSALS", aware)
        ts = TrainStep(m, w, cfg.SAMPLING_RATE, 1e-3, 0.9, seed=5, loss_rows_only=True)
        for _ in range(2):
            ts.step(g)
        res.append(ts.bucket.flat_param[:ts.bucket.numel].clone())
    assert torch.equal(res[0], res[1])


@pytest.mark.parametrize("mode", [True, "backward"])
def test_loss_rows_through_an_arena_follow_each_loaded_batch(mode):
    """Two loader batches of one size class through run_batch: the second is a copy into the arena plus replays of the graph
    captured on the first - the list kernel reads the sampling probabilities the arena load rewrote (other labelled nodes, other
    pad rows), with the capacity fixed at capture (15 % headroom for exactly this)."""
    cfg, model = _model("st_pgat_spgnn_3", 6)
    model.train(True)
    dense = copy.deepcopy(model)
    w = class_weight_list(cfg.CLASS_WEIGHTS)
    ga = synthetic.make_batch(6, rank=3, device="cuda", pos_enc_dim=cfg.POS_ENC_DIM)
    gb = synthetic.make_batch(6, rank=4, device="cuda", pos_enc_dim=cfg.POS_ENC_DIM)
    assert ga.number_of_nodes() != gb.number_of_nodes()
    ts_r = TrainStep(model, w, cfg.SAMPLING_RATE, 1e-3, 0.9, seed=9, loss_rows_only=mode)      # "backward": what INTEGRATION.md recommends
    ts_d = TrainStep(dense, w, cfg.SAMPLING_RATE, 1e-3, 0.9, seed=9)
    for i, g in enumerate((ga, gb, ga)):
        torch.manual_seed(300 + i)
        lr_ = ts_r.run_batch(g, 4, granule=2048)
        torch.manual_seed(300 + i)
        ld_ = ts_d.run_batch(g, 4, granule=2048)
        assert rel_err(lr_, ld_) < 1e-5, (i, float(lr_), float(ld_))
        y = g.ndata["y"]
        assert int((y != 0).sum()) <= int(ts_r._rows_cnt[0]) < 0.5 * g.number_of_nodes()      # this batch's labelled nodes are all listed
    assert len(ts_r._captures) == 1 and len(ts_r._arenas) == 1
    n = ts_r.bucket.numel
    assert rel_err(ts_r.bucket.flat_param[:n], ts_d.bucket.flat_param[:n]) < 1e-5
    ts_r.check_loss_rows()


def test_arena_batch_with_more_labelled_nodes_than_the_captured_capacity_is_recaptured_not_lost():
    """the row-list capacity of a captured step came from the FIRST batch of its size class.  A later batch of the
    class with many more labelled nodes (always kept) exceeds it on every inner step - before the fix all of that loader
    batch's steps were NaN and skipped.  Now the arena load recomputes mean + 8 sigma from the batch's own probabilities, grows
    the capacity and drops the old capture: no step is lost, and the losses equal the dense step's.  (Eval mode: a re-capture
    freezes NEW dropout seeds, so with dropout on the two step objects would draw different masks from the second batch on.)"""
    cfg, model = _model("st_pgat_spgnn_3", 6)
    model.train(False)
    dense = copy.deepcopy(model)
    w = class_weight_list(cfg.CLASS_WEIGHTS)
    ga = synthetic.make_batch(6, rank=3, device="cuda", pos_enc_dim=cfg.POS_ENC_DIM)
    gb = synthetic.make_batch(6, rank=4, device="cuda", pos_enc_dim=cfg.POS_ENC_DIM)
    y = gb.ndata["y"]
    torch.manual_seed(1)
    more = torch.rand(y.shape, device=y.device) < 0.6
    y[more & (y == 0)] = 3                                                  # 60 % of the second batch labelled
    ts_r = TrainStep(model, w, cfg.SAMPLING_RATE, 1e-3, 0.9, seed=9, loss_rows_only=True)
    ts_d = TrainStep(dense, w, cfg.SAMPLING_RATE, 1e-3, 0.9, seed=9)
    caps = []
    for i, g in enumerate((ga, gb, ga)):
        torch.manual_seed(300 + i)
        lr_ = ts_r.run_batch(g, 4, granule=2048)
        torch.manual_seed(300 + i)
        ld_ = ts_d.run_batch(g, 4, granule=2048)
        assert bool(torch.isfinite(lr_)) and rel_err(lr_, ld_) < 1e-5, (i, float(lr_), float(ld_))
        ag = next(iter(ts_r._arenas.values())).graph
        caps.append(ag.__dict__["_loss_rows_cap"][(cfg.SAMPLING_RATE, 0)])
        assert int((g.ndata["y"] != 0).sum()) <= int(ts_r._rows_cnt[0]) <= caps[-1]
        assert int(ts_r._rows_cnt[1]) == 0
    assert caps[1] > caps[0] and caps[2] == caps[1]                          # grown once, kept (capacities never shrink)
    assert int((gb.ndata["y"] != 0).sum()) > caps[0]                        # ... and it had to: the labelled nodes alone overflow the old list
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        assert ts_r.check_loss_rows() == 0                                  # no step skipped, nothing to warn about
    n = ts_r.bucket.numel
    assert rel_err(ts_r.bucket.flat_param[:n], ts_d.bucket.flat_param[:n]) < 1e-5


def test_steps_skipped_without_an_overflow_are_reported():
    """the guarded optimizer kernel skips every step whose loss is not finite.  Without a row-list overflow that is
    divergence (the reference would carry the NaN on): check_loss_rows() says so instead of counting silently."""
    cfg, model = _model("st_pgat_spgnn_3", 2)
    model.eval()
    w = class_weight_list(cfg.CLASS_WEIGHTS)
    g = synthetic.make_batch(6, rank=2, device="cuda", pos_enc_dim=cfg.POS_ENC_DIM)
    ts = TrainStep(model, w, cfg.SAMPLING_RATE, 1e-3, 0.9, seed=5, loss_rows_only=True)
    assert bool(torch.isfinite(ts.step(g))) and ts.check_loss_rows() == 0
    p = ts._sampling(g)
    keep = p.clone()
    p.fill_(-1.0)                                                           # no node kept: weight sum 0, loss 0 / 0 - and no overflow
    assert not bool(torch.isfinite(ts.step(g)))
    with pytest.warns(RuntimeWarning, match="WITHOUT a row-list overflow"):
        assert ts.check_loss_rows() == 1
    p.copy_(keep)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        assert ts.check_loss_rows() == 1                                    # reported once


@pytest.mark.parametrize("name,trees", [("st_gat_6", 6), ("st_gat_6", 64)])
def test_bf16_storage_loss_rows_step_equals_the_dense_step(name, trees):
    """BASELINE config 4's model (bf16 rows, linear-mean head) with the list: rows of the output product, the head mean and the
    classifier for the kept nodes only.  Row-wise arithmetic is the dense step's per row and weight-gradient sums differ by fp32
    summation order only, so the FIRST step's loss agrees to 1e-6; from the second on, parameters that differ in the 7th digit
    move stored bf16 activations across rounding boundaries (2^-8 each) - the envelope of tests/test_arena.py's bf16 case."""
    cfg, model = _model(name, 13)
    models.set_storage_dtype(model, torch.bfloat16)
    model.eval()
    dense = copy.deepcopy(model)
    w = class_weight_list(cfg.CLASS_WEIGHTS)
    g = synthetic.make_batch(trees, rank=2, device="cuda", pos_enc_dim=None)
    ts_r = TrainStep(model, w, cfg.SAMPLING_RATE, 1e-3, 0.9, seed=5, loss_rows_only=True)
    ts_d = TrainStep(dense, w, cfg.SAMPLING_RATE, 1e-3, 0.9, seed=5)
    for i in range(3):
        lr_, ld_ = ts_r.step(g), ts_d.step(g)
        assert rel_err(lr_, ld_) < (1e-6 if i == 0 else 2e-2), (i, float(lr_), float(ld_))
        if i == 0:
            n = ts_r.bucket.numel
            # (round 6: the dense step's logits now come from spgnn_classifier_ce_bf16, the listed step's from the separate
            # launches - the same fp32 arithmetic in another summation order; a logit gradient that differs in its last bit rounds
            # a few bf16 elements of g_Zx the other way (2^-8 each): the first step's gradients agree to 1e-3 normwise, measured 1e-4)
            assert rel_err(ts_r.bucket.flat_grad[:n], ts_d.bucket.flat_grad[:n]) < 1e-3
    assert ts_r._rows_cnt is not None and 0 < int(ts_r._rows_cnt[0]) < 0.5 * g.number_of_nodes()
    assert rel_err(ts_r.bucket.flat_param[:n], ts_d.bucket.flat_param[:n]) < 2e-2


@pytest.mark.parametrize("name", ["st_gcn_3", "st_gin_3", "st_sage_3"])
def test_loss_rows_for_the_spmm_heads(name):
    """GraphConv / GINConv / SAGEConv nets: the differentiable gather sits behind the LAST aggregation (ops.take_loss_rows in
    nn.GraphConv / GINConv / SAGEConv with ``classifier=``), everything after it - st_gin_3's two 1024-wide products included -
    runs on the kept rows.  Same losses and parameters as the dense step."""
    cfg, model = _model(name, 17)
    model.eval()
    dense = copy.deepcopy(model)
    w = class_weight_list(cfg.CLASS_WEIGHTS)
    g = synthetic.make_batch(16, rank=2, device="cuda", pos_enc_dim=None)
    ts_r = TrainStep(model, w, cfg.SAMPLING_RATE, 1e-3, 0.9, seed=5, loss_rows_only=True)
    ts_d = TrainStep(dense, w, cfg.SAMPLING_RATE, 1e-3, 0.9, seed=5)
    for i in range(3):
        lr_, ld_ = ts_r.step(g), ts_d.step(g)
        assert rel_err(lr_, ld_) < 2e-6, (i, float(lr_), float(ld_))
    assert ts_r._rows_cnt is not None and 0 < int(ts_r._rows_cnt[0]) < 0.5 * g.number_of_nodes()
    n = ts_r.bucket.numel
    # (SAGE's max-pool routing and GIN's LeakyReLU branches are decided on rows that both steps compute with the same arithmetic)
    assert rel_err(ts_r.bucket.flat_param[:n], ts_d.bucket.flat_param[:n]) < 5e-6


@pytest.mark.parametrize("mode", [True, "backward"])
def test_loss_rows_steps_repeat_bit_for_bit(mode):
    """The list is made by counts and ordered writes (no atomics, no arrival order), the products on it are the dense step's
    kernels: two runs from the same state and seeds end in bit-identical parameters."""
    cfg, model = _model("st_pgat_spgnn_3", 31)
    model.train(True)
    twin = copy.deepcopy(model)
    w = class_weight_list(cfg.CLASS_WEIGHTS)
    g = synthetic.make_batch(24, rank=5, device="cuda", pos_enc_dim=cfg.POS_ENC_DIM)
    ends = []
    for m in (model, twin):
        ts = TrainStep(m, w, cfg.SAMPLING_RATE, 1e-3, 0.9, seed=77, loss_rows_only=mode)
        ts._seed_ctr = torch.zeros((1,), dtype=torch.int64, device="cuda")
        ts._use_default_rng = True
        for i in range(3):
            torch.manual_seed(400 + i)
            ts.step(g)
        ends.append(ts.bucket.flat_param[:ts.bucket.numel].clone())
    assert torch.equal(ends[0], ends[1])


def _dp_rows_worker(rank, world, port, ret):
    import os
    import warnings

    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = get_config("st_pgat_spgnn_3")
    w = class_weight_list(cfg.CLASS_WEIGHTS)
    out = {}
    for mode in (False, True):
        torch.manual_seed(0)
        model = models.build_model(cfg.MODEL).cuda()
        model.init(None); model.set_gcn_only(); model.eval()
        g = synthetic.make_batch(6 + rank, rank=rank, device="cuda", pos_enc_dim=cfg.POS_ENC_DIM)       # other trees, other counts per rank
        ts = TrainStep(model, w, cfg.SAMPLING_RATE, 0.02, 0.9, seed=5, loss_rows_only=mode)
        losses = [float(ts.step(g)) for _ in range(3)]
        n = ts.bucket.numel
        params = ts.bucket.flat_param[:n].detach().cpu().clone()
        extra = None
        if mode:
            if rank == 1:                                            # the list overflows on ONE rank only
                ts._sampling(g).fill_(1.0)
                g.__dict__["_loss_rows_cap"][(cfg.SAMPLING_RATE, 0)] = 256
            lost = float(ts.step(g))
            same = torch.equal(ts.bucket.flat_param[:n].detach().cpu(), params)
            with warnings.catch_warnings(record=True) as caught:
                warnings.simplefilter("always")
                skipped = ts.check_loss_rows()
            extra = (lost, same, skipped, len(caught))
        out[mode] = (losses, params, extra)
    ret[rank] = out
    dist.destroy_process_group()


def test_two_ranks_loss_rows_equal_dense_and_skip_an_overflowed_step_together():
    """world_size 2 (two processes on the one GPU, gloo carrying the tensors): each rank lists its own rows, the bucket's
    all-reduce is the dense step's; losses and parameters equal the dense 2-rank run.  Then the list overflows on rank 1 only:
    the all-reduced loss is NaN on BOTH ranks, the guarded optimizer kernel skips the step on both - the replicas stay identical -
    and only rank 1 has a flag to clear."""
    import socket

    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mgr = mp.Manager(); ret = mgr.dict()
    mp.spawn(_dp_rows_worker, args=(2, port, ret), nprocs=2, join=True)
    for r in (0, 1):
        ld, pd, _ = ret[r][False]
        lr_, pr, extra = ret[r][True]
        assert np.allclose(ld, lr_, rtol=2e-6), (ld, lr_)
        assert rel_err(pr, pd) < 5e-6
        lost, same, skipped, warned = extra
        assert np.isnan(lost) and same and skipped == 1
        assert warned == (1 if r == 1 else 0)
    assert torch.equal(ret[0][True][1], ret[1][True][1])              # replicas identical
