"""The training step's tail in one pass (ABI 61, spgnn_classifier_ce): the skinny classifier (reference models.py:1125,
1167-1170: n_out = gnn_out(n_embed)), the masked class-weighted cross entropy (job_runner.py:1896-1900) and the classifier's
weight / bias gradient - against plain torch arithmetic in fp64, against the three launches it replaces, and inside the
training step (ops.FUSED_LOSS_HEAD on / off: same losses, same parameters)."""
import copy

import numpy as np
import pytest
import torch

from spgnn_amd import _capi, models, ops, synthetic
from spgnn_amd.configs import class_weight_list, get_config
from spgnn_amd.train import TrainStep
from tests.util import rel_err

pytestmark = pytest.mark.gpu


def _inputs(N, K, J, seed=0, labelled=0.2):
    gen = torch.Generator().manual_seed(seed)
    x = torch.randn(N, K, generator=gen).cuda()
    w = (torch.randn(J, K, generator=gen) / K ** 0.5).cuda()
    b = (0.1 * torch.randn(J, generator=gen)).cuda()
    y = torch.randint(0, J, (N,), generator=gen)
    y[torch.rand(N, generator=gen) > labelled] = 0
    p = torch.where(y != 0, torch.tensor(1.0), torch.tensor(0.3)).cuda()
    draws = torch.rand(N, generator=gen).cuda()
    cw = (0.5 + torch.rand(J, generator=gen)).cuda()
    return x, w, b, y.cuda(), p, draws, cw


def _reference(x, w, b, y, p, draws, cw):
    xd, wd = x.double(), w.double()
    logits = xd @ wd.t() + (b.double() if b is not None else 0.0)
    m = (draws < p).double()
    wy = m * cw.double()[y]
    lp = torch.log_softmax(logits, 1)
    num = -(wy * lp.gather(1, y[:, None])[:, 0]).sum()
    den = wy.sum()
    g = wy[:, None] * (torch.softmax(logits, 1) - torch.nn.functional.one_hot(y, logits.shape[1]).double())
    return logits, g, num, den, g.t() @ xd, g.sum(0)


@pytest.mark.parametrize("N,K,J,bias", [(76410, 1024, 22, True), (9859, 1024, 22, True), (1, 1024, 22, True), (17, 128, 3, False),
                                        (1000, 384, 22, True), (4097, 256, 16, True), (333, 1024, 32, False), (65, 640, 8, True),
                                        (5000, 1024, 2, True), (76410, 384, 22, True), (2000, 256, 22, True), (999, 128, 20, False),
                                        (3001, 512, 24, True)])
def test_classifier_ce_matches_fp64_arithmetic(N, K, J, bias):
    x, w, b, y, p, draws, cw = _inputs(N, K, J, seed=N + K + J)
    if not bias:
        b = None
    sums = torch.zeros(2, device="cuda")
    head = ops.LossHead(y, p, draws, 0, cw, sums)
    assert ops.classifier_ce_supported(x, w)
    logits, g, wpart, colsum = ops.classifier_ce(x, w, b, head)
    rl, rg, rnum, rden, rgw, rgb = _reference(x, w, b, y, p, draws, cw)
    assert rel_err(logits, rl) < 2e-6 and rel_err(g, rg) < 2e-6
    assert abs(float(sums[0]) - float(rnum)) <= 2e-6 * abs(float(rnum)) + 1e-6 and abs(float(sums[1]) - float(rden)) <= 1e-6 * float(rden) + 1e-6
    rps = _capi.load().spgnn_classifier_ce_rows_per_block(N)
    groups = (2 if K > 256 else 4) if (17 <= J <= 24 and K <= 512) else 1
    assert wpart.shape == ((N + rps - 1) // rps * groups, J, (K + 15) // 16 * 16)
    assert wpart.shape[0] == _capi.load().spgnn_classifier_ce_partial_slices(N, K, J)
    assert rel_err(wpart.sum(0)[:, :K], rgw) < 3e-6 and rel_err(colsum, rgb) < 3e-6
    # rows outside the mask: exact zeros (the backward products rely on it)
    out = ~(draws < p)
    assert (N < 100 or out.any()) and (not out.any() or float(g[out].abs().max()) == 0.0)
    # deterministic: a second launch gives the same bits
    sums2 = torch.zeros(2, device="cuda")
    l2, g2, wp2, cs2 = ops.classifier_ce(x, w, b, ops.LossHead(y, p, draws, 0, cw, sums2))
    assert torch.equal(l2, logits) and torch.equal(g2, g) and torch.equal(wp2, wpart) and torch.equal(cs2, colsum) and torch.equal(sums2, sums)


def test_classifier_ce_equals_the_three_launches_it_replaces():
    """spgnn_scores_fwd + spgnn_masked_ce_step + spgnn_scores_bwd_w on the same inputs: same logits to rounding, the same mask
    bit for bit (kernel-drawn: seed + step counter), sums / gradients to fp32 summation order."""
    N, K, J = 20000, 1024, 22
    x, w, b, y, p, _draws, cw = _inputs(N, K, J, seed=5)
    ctr = torch.full((1,), 7, dtype=torch.int64, device="cuda")
    prev, ops.DROPOUT_SEED_OFFSET = ops.DROPOUT_SEED_OFFSET, ctr
    try:
        sums = torch.zeros(2, device="cuda")
        logits, g, wpart, colsum = ops.classifier_ce(x, w, b, ops.LossHead(y, p, None, 12345, cw, sums))
        l_old = ops.scores_fwd(x, w, bias=b).requires_grad_(True)
        num, den = ops.masked_ce_sums(l_old, y, None, p, cw, draw_seed=12345, unit_grad=True)
        num.backward()
        g_old = l_old.grad
        gw_old = ops.scores_bwd_w(g_old, x)
    finally:
        ops.DROPOUT_SEED_OFFSET = prev
    assert rel_err(logits, l_old) < 1e-6
    assert torch.equal(g == 0, g_old == 0)                     # the same nodes kept
    assert rel_err(g, g_old) < 2e-6 and rel_err(wpart.sum(0)[:, :K], gw_old) < 3e-6
    assert abs(float(sums[0]) - float(num)) < 2e-6 * abs(float(num)) and abs(float(sums[1]) - float(den)) < 1e-6 * float(den)
    assert rel_err(colsum, g_old.sum(0)) < 3e-6


def test_classifier_ce_poisons_like_the_loss_kernel():
    """A label outside [0, J) gives a NaN weight (F.cross_entropy raises; no out-of-bounds read here), a raised overflow flag
    turns every weight into NaN - as spgnn_masked_ce_step / _flagged do."""
    x, w, b, y, p, draws, cw = _inputs(500, 1024, 22, seed=9)
    y2 = y.clone(); y2[3] = 40
    p2 = p.clone(); p2[3] = 1.0
    sums = torch.zeros(2, device="cuda")
    ops.classifier_ce(x, w, b, ops.LossHead(y2, p2, draws, 0, cw, sums))
    assert bool(torch.isnan(sums).all())
    flag = torch.tensor([0, 1], dtype=torch.int32, device="cuda")
    sums = torch.zeros(2, device="cuda")
    ops.classifier_ce(x, w, b, ops.LossHead(y, p, draws, 0, cw, sums, flag=flag))
    assert bool(torch.isnan(sums).all())
    flag.zero_()
    ops.classifier_ce(x, w, b, ops.LossHead(y, p, draws, 0, cw, sums, flag=flag))
    assert bool(torch.isfinite(sums).all())


def _model(name, seed):
    cfg = get_config(name)
    torch.manual_seed(seed)
    model = models.build_model(cfg.MODEL).cuda()
    model.init(None)
    model.set_gcn_only()
    return cfg, model


@pytest.mark.parametrize("name,trees,train_mode", [("st_pgat_spgnn_3", 5, False), ("st_pgat_spgnn_3", 40, True), ("st_gat_3", 5, False),
                                                   ("st_gat_6", 40, True), ("st_sage_3", 40, False)])
def test_training_step_with_the_fused_loss_head_equals_the_separate_launches(name, trees, train_mode, monkeypatch):
    cfg, model = _model(name, 4)
    model.train(train_mode)
    other = copy.deepcopy(model)
    w = class_weight_list(cfg.CLASS_WEIGHTS)
    g = synthetic.make_batch(trees, rank=2, device="cuda", pos_enc_dim=getattr(cfg, "POS_ENC_DIM", None))
    res = {}
    for fused, m in ((True, model), (False, other)):
        monkeypatch.setattr(ops, "FUSED_LOSS_HEAD", fused)
        ts = TrainStep(m, w, cfg.SAMPLING_RATE, 1e-3, 0.9, seed=5)
        ops.KernelTimer.start()
        torch.manual_seed(11)
        losses = [float(ts.step(g)) for _ in range(4)]
        used = {k[0] for k in ops.KernelTimer.stop()}
        assert ("classifier_ce" in used) == fused and ("masked_ce" in used) != fused, used
        res[fused] = (losses, ts.bucket.flat_param[:ts.bucket.numel].clone())
    assert np.allclose(res[True][0], res[False][0], rtol=2e-6, atol=0), (res[True][0], res[False][0])
    assert rel_err(res[True][1], res[False][1]) < 2e-6


def test_captured_step_with_the_fused_loss_head_replays_with_fresh_masks():
    cfg, model = _model("st_pgat_spgnn_3", 6)
    model.train(True)
    w = class_weight_list(cfg.CLASS_WEIGHTS)
    g = synthetic.make_batch(8, rank=3, device="cuda", pos_enc_dim=cfg.POS_ENC_DIM)
    ts = TrainStep(model, w, cfg.SAMPLING_RATE, 1e-3, 0.9, seed=5)
    ts.capture(g)
    losses = [float(ts.replay()) for _ in range(6)]
    assert np.isfinite(losses).all() and len(set(losses)) == 6           # a fresh node mask (and dropout) on every replay
    assert losses[-1] < losses[0] * 1.5


@pytest.mark.parametrize("name", ["st_pgat_spgnnnl_3", "st_gcn_3"])
def test_heads_that_cannot_take_the_loss_keep_the_separate_launches(name):
    """A head whose classifier input is not one of the fused nodes' (the PENL ablation projects first: 167-wide rows; GCN's
    folded classifier reads 64-wide rows, below the kernel's 128-column granule) leaves the LossHead unused and the step falls
    back to spgnn_masked_ce_step."""
    cfg, model = _model(name, 4)
    model.eval()
    w = class_weight_list(cfg.CLASS_WEIGHTS)
    g = synthetic.make_batch(5, rank=2, device="cuda", pos_enc_dim=getattr(cfg, "POS_ENC_DIM", None))
    ts = TrainStep(model, w, cfg.SAMPLING_RATE, 1e-3, 0.9, seed=5)
    ops.KernelTimer.start()
    loss = float(ts.step(g))
    used = {k[0] for k in ops.KernelTimer.stop()}
    assert np.isfinite(loss) and "masked_ce" in used and "classifier_ce" not in used


@pytest.mark.parametrize("N,K,J", [(76410, 384, 22), (3000, 1024, 22), (130, 128, 5)])
def test_classifier_ce_on_bf16_rows(N, K, J):
    """ABI 63: the same pass over bf16 rows (config 4's folded classifier reads the bf16 rows [z_0 .. z_{H-1} | x]): the rows
    are widened exactly, so the reference is the fp64 arithmetic on the bf16-rounded values."""
    x, w, b, y, p, draws, cw = _inputs(N, K, J, seed=N + J)
    xb = x.to(torch.bfloat16)
    sums = torch.zeros(2, device="cuda")
    assert ops.classifier_ce_supported(xb, w)
    logits, g, wpart, colsum = ops.classifier_ce(xb, w, b, ops.LossHead(y, p, draws, 0, cw, sums))
    rl, rg, rnum, rden, rgw, rgb = _reference(xb.float(), w, b, y, p, draws, cw)
    assert rel_err(logits, rl) < 2e-6 and rel_err(g, rg) < 2e-6
    assert abs(float(sums[0]) - float(rnum)) <= 2e-6 * abs(float(rnum)) + 1e-6 and abs(float(sums[1]) - float(rden)) <= 1e-6 * float(rden) + 1e-6
    assert rel_err(wpart.sum(0)[:, :K], rgw) < 3e-6 and rel_err(colsum, rgb) < 3e-6


def test_bf16_training_step_with_the_fused_loss_head_equals_the_separate_launches(monkeypatch):
    cfg, model = _model("st_gat_6", 4)
    models.set_storage_dtype(model, torch.bfloat16)
    model.eval()
    other = copy.deepcopy(model)
    w = class_weight_list(cfg.CLASS_WEIGHTS)
    g = synthetic.make_batch(40, rank=2, device="cuda", pos_enc_dim=getattr(cfg, "POS_ENC_DIM", None))
    res = {}
    for fused, m in ((True, model), (False, other)):
        monkeypatch.setattr(ops, "FUSED_LOSS_HEAD", fused)
        ts = TrainStep(m, w, cfg.SAMPLING_RATE, 1e-3, 0.9, seed=5)
        ops.KernelTimer.start()
        torch.manual_seed(11)
        losses = [float(ts.step(g))]
        first = ts.bucket.flat_param[:ts.bucket.numel].clone()
        losses += [float(ts.step(g)) for _ in range(3)]
        used = {k[0] for k in ops.KernelTimer.stop()}
        assert ("classifier_ce_bf16" in used) == fused and ("masked_ce" in used) != fused, used
        res[fused] = (losses, first)
    # one step: the same loss and update to fp32 summation order; later steps drift apart through bf16 rounding boundaries
    # (a 1e-7 difference upstream flips a stored bf16 value now and then), so they are held to 1e-3 only
    assert abs(res[True][0][0] - res[False][0][0]) <= 5e-6 * abs(res[False][0][0])
    assert rel_err(res[True][1], res[False][1]) < 5e-6
    assert np.allclose(res[True][0], res[False][0], rtol=1e-3, atol=0), (res[True][0], res[False][0])
