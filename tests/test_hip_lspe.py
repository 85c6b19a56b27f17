"""The fused SPGNN level (spgnn_lspe_fwd / _bwd_dst / _bwd_src: structure GATConv + position GATConv of reference
models.py:472-484 in ONE traversal) against the two-layer form it replaces and against the oracle.

* forward: bit for bit equal to the two-layer form (same arithmetic per element, same masks from one seed plan), in eval mode
  and with every dropout on;
* backward: the per-edge dots are summed over a different team geometry (3 heads per team instead of 2 and 1), so gradients
  agree to fp32 rounding: <= 2e-6 normwise here (tolerance written below), and they match the oracle like every other
  configuration (tests/test_hip_models.py runs st_pgat_spgnn_3 through this path);
* graphs the kernels do not take (a node of degree > 8) fall back to the two-layer form.
"""
import numpy as np
import pytest
import torch

from oracle import dgl_cpu as O
from spgnn_amd import models, ops, synthetic
from spgnn_amd.configs import class_weight_list, get_config
from spgnn_amd.graph import TreeGraph
from spgnn_amd.train import masked_weighted_ce
from tests.util import rel_err

pytestmark = pytest.mark.gpu
GRAD_TOL = 2e-6      # fused vs two-layer gradients, normwise relative (fp32 summation order only)


def _model(seed=0):
    cfg = get_config("st_pgat_spgnn_3")
    torch.manual_seed(seed)
    model = models.build_model(cfg.MODEL).cuda()
    model.init(None)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith("bias"):
                p.normal_(0, 0.05)
    model.set_gcn_only()
    return cfg, model


def _run(model, g, cfg, fused, train, monkeypatch, seed=77):
    monkeypatch.setattr(models, "FUSE_LSPE", fused)
    model.train(train)
    for p in model.parameters():
        p.grad = None
    torch.manual_seed(seed)                       # the seed plan draws from torch's CPU generator
    outs = model(g)
    w = torch.tensor(class_weight_list(cfg.CLASS_WEIGHTS), device="cuda")
    y = g.ndata["y"]
    mask = (torch.rand(y.shape[0], generator=torch.Generator().manual_seed(5)) < 0.5).cuda()
    # the position output h_p enters the loss too, so both consumers of every level's position rows carry a gradient
    loss = masked_weighted_ce(outs[0], y, mask, w) + 1e-3 * outs[2].square().sum() + 1e-4 * outs[1].square().mean()
    loss.backward()
    grads = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
    return [o.detach().clone() for o in outs], grads, float(loss.detach())


@pytest.mark.parametrize("train", [False, True])
def test_fused_level_equals_two_layer_form(train, monkeypatch):
    cfg, model = _model()
    g = synthetic.make_batch(5, rank=2, device="cuda", pos_enc_dim=cfg.POS_ENC_DIM)
    assert model.gat._lspe_ok(g, torch.empty(1, 8, device="cuda"), g.ndata["pos_enc"])
    timed = {}
    ops.KernelTimer.start()
    o_f, g_f, l_f = _run(model, g, cfg, True, train, monkeypatch)
    timed["fused"] = {k[0] for k in ops.KernelTimer.stop()}
    ops.KernelTimer.start()
    o_u, g_u, l_u = _run(model, g, cfg, False, train, monkeypatch)
    timed["two"] = {k[0] for k in ops.KernelTimer.stop()}
    assert {"lspe_fwd", "lspe_bwd_dst", "lspe_bwd_src"} <= timed["fused"] and "lspe_fwd" not in timed["two"]
    for a, b in zip(o_f, o_u):                    # logits, node embedding, position embedding: bit for bit
        assert torch.equal(a, b)
    assert l_f == l_u
    assert set(g_f) == set(g_u)
    gmax = max(float(v.abs().max()) for v in g_u.values())
    for n in g_u:
        tiny = float((g_f[n] - g_u[n]).abs().max()) < 1e-7 * gmax
        # with feature dropout on, the two-layer form keeps the position rows un-dropped for tanh' while the fused form
        # recovers them as stored * (1 - p): one more rounding per element
        assert rel_err(g_f[n], g_u[n]) < (4 * GRAD_TOL if train else GRAD_TOL) or tiny, (n, rel_err(g_f[n], g_u[n]))


def test_pair_launches_leave_the_step_bitwise_unchanged(monkeypatch):
    """ops.PAIR_GEMMS: a level's two projections / weight gradients / input gradients as one launch each - outputs and every
    gradient equal the separate launches bit for bit, and the paired step has the fewer launches."""
    cfg, model = _model(3)
    g = synthetic.make_batch(4, rank=1, device="cuda", pos_enc_dim=cfg.POS_ENC_DIM)
    res, names = [], []
    for pair in (True, False):
        monkeypatch.setattr(ops, "PAIR_GEMMS", pair)
        monkeypatch.setattr(ops, "PAIR_SCORE_GRADS", pair)          # ... and the two attention-vector gradient passes
        ops.KernelTimer.start()
        res.append(_run(model, g, cfg, True, True, monkeypatch))
        names.append([k[0] for k in ops.KernelTimer.sequence])
        ops.KernelTimer.stop()
    assert "gemm_nt_pair" in names[0] and "gemm_tn_pair" in names[0] and "gemm_nt_pair" not in names[1]
    assert "scores_bwd_w_pair" in names[0] and "scores_bwd_w_pair" not in names[1]
    for a, b in zip(res[0][0], res[1][0]):
        assert torch.equal(a, b)
    assert set(res[0][1]) == set(res[1][1])
    for n in res[0][1]:
        assert torch.equal(res[0][1][n], res[1][1][n]), n


def test_fused_level_matches_oracle_gradients(monkeypatch):
    """fwd + all parameter gradients of the fused path against the CPU oracle (eval-mode arithmetic), with the position
    output inside the loss so that the second gradient path of every level is exercised."""
    cfg, model = _model(seed=3)
    g = synthetic.make_batch(3, rank=4, device="cuda", pos_enc_dim=cfg.POS_ENC_DIM)
    outs, grads, _ = _run(model, g, cfg, True, False, monkeypatch)
    src, dst = g.cpu().edges()
    sd = {k: v.detach().cpu().double().requires_grad_(v.dtype.is_floating_point) for k, v in model.state_dict().items()}
    ref = O.net_forward(cfg.KIND, sd, src, dst, g.number_of_nodes(), g.ndata["fvs"].cpu().double(), g.ndata["pos_enc"].cpu().double())
    w = torch.tensor(class_weight_list(cfg.CLASS_WEIGHTS)).double()
    y = g.ndata["y"].cpu()
    mask = torch.rand(y.shape[0], generator=torch.Generator().manual_seed(5)) < 0.5
    (O.masked_weighted_ce(ref[0], y, mask, w) + 1e-3 * ref[2].square().sum() + 1e-4 * ref[1].square().mean()).backward()
    for o, r in zip(outs, ref):
        assert rel_err(o, r) < 1e-5
    gmax = max(float(v.grad.abs().max()) for v in sd.values() if v.grad is not None)
    for n, gr in grads.items():
        tiny = float((gr.cpu().double() - sd[n].grad).abs().max()) < 1e-7 * gmax
        assert rel_err(gr, sd[n].grad) < 1e-4 or tiny, (n, rel_err(gr, sd[n].grad))


def test_high_degree_graph_takes_the_two_layer_form(monkeypatch):
    """A node with more than 8 neighbours: the fused kernels are not used (they assume degree <= 8) and the model still
    matches the oracle."""
    cfg, model = _model(seed=1)
    n = 40
    src = np.concatenate([np.zeros(n - 1, dtype=np.int64), np.arange(1, n), np.arange(n)])     # a star + self loops
    dst = np.concatenate([np.arange(1, n), np.zeros(n - 1, dtype=np.int64), np.arange(n)])
    g = TreeGraph((src, dst), n).to("cuda")
    torch.manual_seed(0)
    g.ndata["fvs"] = torch.relu(torch.randn(n, 1024, device="cuda"))
    g.ndata["pos_enc"] = torch.rand(n, cfg.POS_ENC_DIM, device="cuda")
    model.eval()
    assert not model.gat._lspe_ok(g, torch.empty(1, 8, device="cuda"), g.ndata["pos_enc"])
    ops.KernelTimer.start()
    with torch.no_grad():
        outs = model(g)
    assert "lspe_fwd" not in {k[0] for k in ops.KernelTimer.stop()}
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    ref = O.net_forward(cfg.KIND, sd, torch.from_numpy(src), torch.from_numpy(dst), n, g.ndata["fvs"].cpu(), g.ndata["pos_enc"].cpu())
    for o, r in zip(outs, ref):
        assert rel_err(o, r) < 1e-5
