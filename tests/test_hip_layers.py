"""HIP kernels vs the CPU oracle, layer level, through the C ABI (spgnn_amd.nn -> ops -> ctypes).
Tolerance: 1e-5 normwise relative on forward outputs in fp32 (BASELINE.json north_star); gradients
5e-5 (same arithmetic, longer reduction chains)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import dgl_cpu as O
from spgnn_amd import nn as snn
from spgnn_amd import ops
from spgnn_amd.graph import TreeGraph
from tests.util import keep4_scale_host, keep_scale_host, rel_err, tree_batch_edges

pytestmark = pytest.mark.gpu
FWD_TOL, GRAD_TOL = 1e-5, 5e-5


def _graph(ns, seed=0):
    s, d, n = tree_batch_edges(ns, seed)
    return TreeGraph((s, d), n).to("cuda"), torch.from_numpy(s), torch.from_numpy(d), n


def _oracle_gat(layer, src, dst, n, x, act, dtype=torch.float32, attn_keep=None):
    sd = {k: v.detach().cpu().to(dtype).requires_grad_(True) for k, v in layer.state_dict().items()}
    xo = x.detach().cpu().to(dtype).requires_grad_(True)
    rst, a = O.gat_conv(src, dst, n, xo, sd["fc.weight"], sd["attn_l"], sd["attn_r"], sd.get("res_fc.weight"),
                        sd.get("bias"), layer.negative_slope, act, attn_keep,
                        residual_identity=isinstance(layer.res_fc, snn.Identity))
    return rst, a, xo, sd


# every GATConv shape of the configs + head sweep + odd shapes (scalar fallback)
GAT_CASES = [
    (1063, 2, 256, True, F.elu), (39, 1, 256, True, torch.tanh), (768, 2, 128, True, F.elu),
    (256, 1, 128, True, torch.tanh), (384, 2, 64, True, F.elu), (128, 1, 64, True, torch.tanh),
    (192, 2, 1024, True, F.elu), (1024, 2, 256, True, F.elu), (128, 2, 1024, True, None),
    (512, 2, 128, False, F.elu), (64, 8, 64, True, F.elu), (48, 8, 256, False, None), (40, 4, 16, True, F.relu),
    (7, 3, 5, True, F.elu), (9, 1, 6, False, None), (20, 2, 12, True, torch.tanh), (33, 1, 1024, True, None),
    (16, 4, 8, True, None), (16, 1, 4, False, F.elu), (24, 2, 512, True, F.elu),
]


@pytest.mark.parametrize("fin,H,D,res,act", GAT_CASES)
def test_gatconv_forward_backward(fin, H, D, res, act):
    torch.manual_seed(fin * 7 + H * 3 + D)
    g, src, dst, n = _graph([23, 150, 1, 64], seed=fin)
    layer = snn.GATConv(fin, D, H, 0.0, 0.0, 0.2, res, act).cuda()
    with torch.no_grad():
        layer.bias.normal_(0, 0.1)
    x = torch.randn(n, fin, device="cuda").clamp_(min=-0.5).requires_grad_(True)
    out, attn = layer(g, x, get_attention=True)
    ref, a_ref, xo, sd = _oracle_gat(layer, src, dst, n, x, act)
    assert out.shape == (n, H, D)
    assert rel_err(out, ref) < FWD_TOL
    assert rel_err(attn.squeeze(-1), a_ref) < FWD_TOL            # attention returned in edge-id order
    cot = torch.randn(n, H, D)
    (out * cot.cuda()).sum().backward()
    (ref * cot).sum().backward()
    assert rel_err(x.grad, xo.grad) < GRAD_TOL
    for name, p in layer.named_parameters():
        assert rel_err(p.grad, sd[name].grad) < GRAD_TOL, name
    # the fp32 HIP path should sit as close to an fp64 evaluation as the fp32 CPU oracle does (x4 slack)
    ref64 = _oracle_gat(layer, src, dst, n, x, act, torch.float64)[0]
    assert rel_err(out, ref64) < max(4 * rel_err(ref, ref64), 2e-6)


@pytest.mark.parametrize("fin,H,D,act", [(192, 2, 1024, F.elu), (128, 2, 1024, None), (64, 2, 256, torch.tanh),
                                          (32, 8, 256, F.elu), (24, 2, 64, F.elu), (10, 3, 5, None), (16, 1, 128, F.elu)])
def test_gatconv_fused_head_mean(fin, H, D, act):
    """mean_heads=True == rst.mean(1) (reference models.py:327, 482), forward and gradients; covers the fused
    epilogue (head >= team), the unfused geometries and the scalar fallback."""
    torch.manual_seed(H * 100 + D)
    g, src, dst, n = _graph([40, 150, 7], seed=D)
    layer = snn.GATConv(fin, D, H, 0.0, 0.0, 0.2, True, act).cuda()
    with torch.no_grad():
        layer.bias.normal_(0, 0.1)
    x = torch.randn(n, fin, device="cuda", requires_grad=True)
    out = layer(g, x, mean_heads=True)
    ref, _, xo, sd = _oracle_gat(layer, src, dst, n, x, act)
    ref = ref.mean(1)
    assert out.shape == (n, D) and rel_err(out, ref) < FWD_TOL
    cot = torch.randn(n, D)
    (out * cot.cuda()).sum().backward(); (ref * cot).sum().backward()
    assert rel_err(x.grad, xo.grad) < GRAD_TOL
    for name, p in layer.named_parameters():
        assert rel_err(p.grad, sd[name].grad) < GRAD_TOL, name


def test_gatconv_high_degree_and_no_self_loops():
    """Star graph with 70 children (degree loops far beyond the airway 2..5) and a ring without
    self loops (non-symmetric in/out lists)."""
    torch.manual_seed(0)
    n = 71
    s = np.concatenate([np.zeros(70, np.int64), np.arange(1, 71), np.arange(71)])
    d = np.concatenate([np.arange(1, 71), np.zeros(70, np.int64), np.arange(71)])
    ring_s = np.arange(10, dtype=np.int64) + n
    ring_d = (np.arange(10, dtype=np.int64) + 1) % 10 + n
    s, d = np.concatenate([s, ring_s, ring_s]), np.concatenate([d, ring_d, (ring_d - n + 1) % 10 + n])
    n += 10
    g = TreeGraph((s, d), n).to("cuda")
    for fin, H, D in [(12, 2, 64), (12, 2, 256), (5, 3, 7)]:
        layer = snn.GATConv(fin, D, H, 0.0, 0.0, 0.2, True, F.elu).cuda()
        x = torch.randn(n, fin, device="cuda", requires_grad=True)
        out = layer(g, x)
        ref, _, xo, sd = _oracle_gat(layer, torch.from_numpy(s), torch.from_numpy(d), n, x, F.elu)
        assert rel_err(out, ref) < FWD_TOL
        out.pow(2).sum().backward(); ref.pow(2).sum().backward()
        assert rel_err(x.grad, xo.grad) < GRAD_TOL
        for name, p in layer.named_parameters():
            assert rel_err(p.grad, sd[name].grad) < GRAD_TOL, name


def test_gatconv_edge_cases():
    layer = snn.GATConv(8, 4, 2).cuda()
    g0 = TreeGraph((np.array([0, 1]), np.array([1, 1])), 3).to("cuda")       # nodes 0 and 2 have no in-edge
    with pytest.raises(snn.DGLError):
        layer(g0, torch.zeros(3, 8, device="cuda"))
    layer.set_allow_zero_in_degree(True)
    out = layer(g0, torch.randn(3, 8, device="cuda"))
    assert torch.equal(out[0], layer.bias.view(2, 4)) and torch.isfinite(out).all()   # empty sum + bias
    g1 = TreeGraph((np.array([0]), np.array([0])), 1).to("cuda")              # single node, self loop
    x = torch.randn(1, 8, device="cuda")
    assert rel_err(layer(g1, x), layer.fc(x).view(1, 2, 4) + layer.bias.view(1, 2, 4)) < FWD_TOL
    ge = TreeGraph(None, 0).to("cuda")                                        # empty graph
    assert layer(ge, torch.zeros(0, 8, device="cuda")).shape == (0, 2, 4)
    with pytest.raises(RuntimeError):                                         # no CPU fallback
        layer(TreeGraph((np.array([0]), np.array([0])), 1), torch.randn(1, 8))
    # identity residual (in_feats == out_feats): broadcast over heads, DGL 0.6-0.8 semantics
    idl = snn.GATConv(4, 4, 3, residual=True, activation=F.elu).cuda()
    assert isinstance(idl.res_fc, snn.Identity)
    s, d, n = tree_batch_edges([9], 1)
    g = TreeGraph((s, d), n).to("cuda")
    x = torch.randn(n, 4, device="cuda")
    sd = {k: v.cpu() for k, v in idl.state_dict().items()}
    ref = O.gat_conv(torch.from_numpy(s), torch.from_numpy(d), n, x.cpu(), sd["fc.weight"], sd["attn_l"], sd["attn_r"],
                     None, sd["bias"], 0.2, F.elu, residual_identity=True)[0]
    assert rel_err(idl(g, x), ref) < FWD_TOL


def test_attention_dropout_matches_oracle_with_same_mask(monkeypatch):
    """attn_drop > 0: the kernel's counter-based mask is reproduced on the host and fed to the
    oracle; forward and every gradient must then agree, and the keep rate must be ~1-p."""
    torch.manual_seed(3)
    g, src, dst, n = _graph([150, 140, 160], seed=9)
    H, D, fin, p, seed = 2, 128, 96, 0.1, 123456789
    csc = g.csc()
    layer = snn.GATConv(fin, D, H, 0.0, p, 0.2, True, F.elu).cuda().train()
    monkeypatch.setattr(snn, "_draw_seed", lambda: seed)
    x = torch.randn(n, fin, device="cuda", requires_grad=True)
    out = layer(g, x)
    E = csc.num_edges
    keep_slot = keep_scale_host(seed, np.arange(E * H), p).reshape(E, H)      # CSC slot order
    keep_edge = np.empty_like(keep_slot); keep_edge[csc.eid.cpu().numpy()] = keep_slot
    assert abs((keep_slot > 0).mean() - (1 - p)) < 0.01
    ref, _, xo, sd = _oracle_gat(layer, src, dst, n, x, F.elu, attn_keep=torch.from_numpy(keep_edge))
    assert rel_err(out, ref) < FWD_TOL
    out.pow(2).sum().backward(); ref.pow(2).sum().backward()
    assert rel_err(x.grad, xo.grad) < GRAD_TOL
    for name, prm in layer.named_parameters():
        assert rel_err(prm.grad, sd[name].grad) < GRAD_TOL, name
    # get_attention=True in training mode returns attn_drop(edge_softmax(e)) per edge id, as DGL does (what was aggregated)
    _, a_ret = layer(g, x.detach(), get_attention=True)
    a_ref = _oracle_gat(layer, src, dst, n, x, F.elu, attn_keep=torch.from_numpy(keep_edge))[1] * torch.from_numpy(keep_edge)
    assert a_ret.shape == (E, H, 1) and rel_err(a_ret.squeeze(-1), a_ref) < FWD_TOL
    assert ((a_ret.squeeze(-1) == 0).cpu() == torch.from_numpy(keep_edge == 0)).all()
    monkeypatch.undo()
    # fresh seeds per call in train mode -> different masks; eval mode -> no dropout, deterministic
    a = layer(g, x.detach()); b = layer(g, x.detach())
    assert not torch.equal(a, b)
    layer.eval(); c = layer(g, x.detach()); d = layer(g, x.detach())
    assert torch.equal(c, d)


@pytest.mark.parametrize("fin,H,D,res,act,mean", [(192, 2, 1024, True, F.elu, True), (600, 4, 640, True, F.elu, False),
                                                  (300, 1, 512, False, torch.tanh, False), (36, 2, 64, False, None, True),
                                                  (256, 4, 260, True, F.elu, True), (1024, 1, 2048, True, F.elu, False),
                                                  (100, 2, 128, True, F.elu, False), (128, 4, 256, False, F.elu, True)])   # 16-lane teams, two chunks per lane
def test_aggregate_first_form_matches_project_first_and_oracle(monkeypatch, fin, H, D, res, act, mean):
    """Input narrower than a head: the layer aggregates input rows first and projects afterwards (ops._GATAggFirstFn).
    Same function as the project-first form (A/B through nn.AGGREGATE_FIRST) and as the oracle, with attention
    dropout on (same counter-based mask in both forms).  (Smooth activations only: with ReLU a pre-activation of
    -1.8e-7 came out +2.7e-7 in one form, and that single flipped derivative moves the gradients by 1 %; the ReLU
    epilogue and derivative kernels are checked on their own below.)"""
    torch.manual_seed(fin + H + D)
    g, src, dst, n = _graph([33, 150, 2, 64], seed=fin)
    p, seed = 0.1, 987654321
    layer = snn.GATConv(fin, D, H, 0.0, p, 0.2, res, act).cuda().train()
    monkeypatch.setattr(snn, "_draw_seed", lambda: seed)
    with torch.no_grad():
        layer.bias.normal_(0, 0.1)
    x = torch.randn(n, fin, device="cuda")
    cot = torch.randn(n, D, device="cuda") if mean else torch.randn(n, H, D, device="cuda")
    results = {}
    for form in (True, False):
        monkeypatch.setattr(snn, "AGGREGATE_FIRST", form)
        ops.KernelTimer.start()
        xg = x.clone().requires_grad_(True)
        layer.zero_grad()
        out = layer(g, xg, mean_heads=mean)
        (out * cot).sum().backward()
        used = {k[0] for k in ops.KernelTimer.stop()}
        assert ("gat_agg_fwd" in used) == form and ("gat_fwd" in used) != form
        results[form] = [out.detach(), xg.grad] + [q.grad.clone() for q in layer.parameters()]
    csc = g.csc()
    E = csc.num_edges
    keep_slot = keep_scale_host(seed, np.arange(E * H), p).reshape(E, H)
    keep_edge = np.empty_like(keep_slot); keep_edge[csc.eid.cpu().numpy()] = keep_slot
    ref, _, xo, sd = _oracle_gat(layer, src, dst, n, x, act, attn_keep=torch.from_numpy(keep_edge))
    ref = ref.mean(1) if mean else ref
    (ref * cot.cpu()).sum().backward()
    names = ["out", "x"] + [k for k, _ in layer.named_parameters()]
    want = [ref, xo.grad] + [sd[k].grad for k in names[2:]]
    for form in (True, False):
        for name, got, w in zip(names, results[form], want):
            assert rel_err(got, w) < (FWD_TOL if name == "out" else GRAD_TOL), (form, name)
    for name, a, b in zip(names, results[True], results[False]):
        assert rel_err(a, b) < GRAD_TOL, name

@pytest.mark.parametrize("fin,H,D,res,p", [(128, 2, 1024, True, 0.0), (36, 2, 64, False, 0.1), (192, 4, 256, True, 0.1), (64, 1, 512, True, 0.0)])
def test_linear_mean_output_layer_matches_per_head_form_and_oracle(monkeypatch, fin, H, D, res, p):
    """An output GATConv without activation whose heads are averaged (reference models.py:320-327, GAT's last layer) is
    linear in [z_0 .. z_{H-1} | x]: ops.gat_layer_linear_mean evaluates it as ONE product (nn.LINEAR_MEAN).  Same
    function as the per-head aggregate-first form and as the oracle, forward and every gradient, attention dropout on."""
    torch.manual_seed(fin + H + D)
    g, src, dst, n = _graph([150, 33, 2, 64], seed=fin)
    seed = 192837465
    layer = snn.GATConv(fin, D, H, 0.0, p, 0.2, res, None).cuda().train()
    monkeypatch.setattr(snn, "_draw_seed", lambda: seed)
    with torch.no_grad():
        layer.bias.normal_(0, 0.1)
    x = torch.randn(n, fin, device="cuda")
    cot = torch.randn(n, D, device="cuda")
    results = {}
    for form in (True, False):
        monkeypatch.setattr(snn, "LINEAR_MEAN", form)
        xg = x.clone().requires_grad_(True)
        layer.zero_grad()
        out = layer(g, xg, mean_heads=True)
        (out * cot).sum().backward()
        results[form] = [out.detach(), xg.grad] + [q.grad.clone() for q in layer.parameters()]
    csc = g.csc()
    E = csc.num_edges
    keep_edge = None
    if p > 0:
        keep_slot = keep_scale_host(seed, np.arange(E * H), p).reshape(E, H)
        keep_edge = np.empty_like(keep_slot); keep_edge[csc.eid.cpu().numpy()] = keep_slot
        keep_edge = torch.from_numpy(keep_edge)
    ref, _, xo, sd = _oracle_gat(layer, src, dst, n, x, None, attn_keep=keep_edge)
    ref = ref.mean(1)
    (ref * cot.cpu()).sum().backward()
    names = ["out", "x"] + [k for k, _ in layer.named_parameters()]
    want = [ref, xo.grad] + [sd[k].grad for k in names[2:]]
    for form in (True, False):
        for name, got, w in zip(names, results[form], want):
            assert rel_err(got, w) < (FWD_TOL if name == "out" else GRAD_TOL), (form, name)


@pytest.mark.parametrize("emb_in_loss", [False, True])
def test_linear_with_joined_classifier_both_routes(emb_in_loss):
    """ops._LinearClassifierFn (the *Net's gnn_out joined to the linear-mean output product, reference models.py:921-933):
    y = x W^T + b and logits = y Wc^T + bc against fp64, forward and every gradient - by the folded route (only the
    logits in the loss: no (N, C) gradient is formed) and by the ordinary one (the embedding in the loss as well)."""
    torch.manual_seed(3 + emb_in_loss)
    N, K, C, J = 700, 384, 1024, 22
    x = torch.randn(N, K, device="cuda", requires_grad=True)
    w = (torch.randn(C, K, device="cuda") / 16).requires_grad_()
    b = torch.randn(C, device="cuda", requires_grad=True)
    wc = (torch.randn(J, C, device="cuda") / 32).requires_grad_()
    bc = torch.randn(J, device="cuda", requires_grad=True)
    assert ops.linear_classifier_supported(x, w, wc)
    y, logits = ops._LinearClassifierFn.apply(x, w, b, wc, bc)
    cl, cy = torch.randn(N, J, device="cuda"), torch.randn(N, C, device="cuda")
    loss = (logits * cl).sum() + ((y * cy).sum() if emb_in_loss else 0.0)
    loss.backward()
    leaves = [x, w, b, wc, bc]
    ref = [t.detach().double().cpu().requires_grad_() for t in leaves]
    ry = ref[0] @ ref[1].t() + ref[2]
    rl = ry @ ref[3].t() + ref[4]
    ((rl * cl.double().cpu()).sum() + ((ry * cy.double().cpu()).sum() if emb_in_loss else 0.0)).backward()
    assert rel_err(y, ry) < FWD_TOL and rel_err(logits, rl) < FWD_TOL
    for name, got, want in zip(["x", "w", "b", "w_cls", "b_cls"], leaves, ref):
        assert rel_err(got.grad, want.grad) < GRAD_TOL, name


@pytest.mark.parametrize("act", [ops.ACT_NONE, ops.ACT_ELU, ops.ACT_TANH, ops.ACT_RELU])
def test_epilogue_and_derivative_kernels_of_the_aggregate_first_form(act):
    """spgnn_gemm_nt's bias + activation epilogue (on column-slice outputs), spgnn_head_mean and spgnn_act_bwd."""
    torch.manual_seed(act)
    M, K, Nn, H = 249, 512, 260, 4
    a = torch.randn(M, K, device="cuda"); b = torch.randn(Nn, K, device="cuda") / 16; bias = torch.randn(Nn, device="cuda")
    big = torch.full((M, H * Nn), float("nan"), device="cuda")
    for h in range(H):
        ops.gemm_nt(a, b, ops.pow2_scale(a), ops.pow2_scale(b), out=big[:, h * Nn:(h + 1) * Nn], bias=bias, act=act)
    pre = a.double() @ b.double().t() + bias.double()
    f = {ops.ACT_NONE: lambda t: t, ops.ACT_ELU: F.elu, ops.ACT_TANH: torch.tanh, ops.ACT_RELU: torch.relu}[act]
    ref = f(pre).float()
    for h in range(H):
        assert rel_err(big[:, h * Nn:(h + 1) * Nn], ref) < 4e-6      # tanhf: 2.1e-6 against fp64
    out = f(torch.randn(M, H * Nn, device="cuda"))
    assert rel_err(ops.head_mean(out, H, Nn), out.view(M, H, Nn).mean(1)) < 1e-6
    for mean in (False, True):
        g = torch.randn(M, Nn if mean else H * Nn, device="cuda")
        gp, amax = ops.act_bwd(g, out if act != ops.ACT_NONE else None, H, Nn, act, mean)
        d = {ops.ACT_NONE: torch.ones_like(out), ops.ACT_ELU: torch.where(out > 0, torch.ones_like(out), out + 1),
             ops.ACT_TANH: 1 - out * out, ops.ACT_RELU: (out > 0).float()}[act]
        want = (g / H).repeat(1, H) * d if mean else g * d
        assert rel_err(gp, want) < 1e-6
        # the scale block: its largest slot is max |g_pre| exactly (a maximum of stored values), and the scale derived from it
        assert amax.numel() == ops.SCALE_HEADER + ops.SCALE_SLOTS and float(amax[0]) == -ops.SCALE_SLOTS
        assert float(amax[ops.SCALE_HEADER:].max()) == float(gp.abs().max())
        assert ops.scale_value(amax) == float(ops.pow2_scale(gp.contiguous()))


def test_cat_dropout_is_cat_then_dropout_with_a_regenerated_mask():
    torch.manual_seed(0)
    N, p, seed = 777, 0.25, 4242
    a = torch.randn(N, 130, device="cuda", requires_grad=True); b = torch.randn(N, 39, device="cuda", requires_grad=True)
    y0 = ops.cat_dropout((a, b), 0.0)
    assert y0.stride(0) % 4 == 0 and torch.equal(y0, torch.cat([a, b], 1))
    y = ops.cat_dropout((a, b), p, seed)
    keep = torch.from_numpy(keep4_scale_host(seed, N, (130, 39), p)).cuda()
    assert abs(float((keep > 0).float().mean()) - (1 - p)) < 0.01
    assert torch.equal(y, torch.cat([a, b], 1) * keep)
    cot = torch.randn(N, 169, device="cuda")
    (y * cot).sum().backward()
    assert torch.equal(a.grad, (cot * keep)[:, :130]) and torch.equal(b.grad, (cot * keep)[:, 130:])
    assert torch.equal(ops.cat_dropout((a, b), p, seed), y) and not torch.equal(ops.cat_dropout((a, b), p, seed + 1), y)
    c = torch.randn(N, 512, device="cuda"); d = torch.randn(N, 256, device="cuda")      # the 16-byte vector path
    keep2 = torch.from_numpy(keep4_scale_host(seed, N, (512, 256), p)).cuda()
    assert torch.equal(ops.cat_dropout((c, d), p, seed), torch.cat([c, d], 1) * keep2)
    # p = 0: the gradients are column blocks of the incoming one (views when 16-byte aligned, copies otherwise)
    for wa, wb in ((512, 64), (130, 39)):
        e = torch.randn(N, wa, device="cuda", requires_grad=True); f = torch.randn(N, wb, device="cuda", requires_grad=True)
        cot = torch.randn(N, wa + wb, device="cuda")
        (ops.cat_dropout((e, f), 0.0) * cot).sum().backward()
        assert torch.equal(e.grad, cot[:, :wa]) and torch.equal(f.grad, cot[:, wa:])


SPMM_F = [64, 128, 256, 1024, 22, 7, 192]


@pytest.mark.parametrize("F_", SPMM_F)
def test_spmm_sum_and_max_vs_oracle(F_):
    torch.manual_seed(F_)
    g, src, dst, n = _graph([150, 33, 1, 90], seed=F_)
    csc = g.csc()
    x = torch.randn(n, F_, device="cuda", requires_grad=True)
    xo = x.detach().cpu().requires_grad_(True)
    ws, wd = csc.out_degrees_f().pow(-0.5), csc.in_degrees_f().pow(-0.5)
    eps = torch.tensor([0.25], device="cuda", requires_grad=True)
    epo = eps.detach().cpu().requires_grad_(True)
    out = ops.spmm_sum(csc, x, ws, wd, eps)
    ref = (1 + epo) * xo + O.spmm_sum(src, dst, xo * ws.cpu().unsqueeze(1), n) * wd.cpu().unsqueeze(1)
    assert rel_err(out, ref) < FWD_TOL
    cot = torch.randn(n, F_)
    (out * cot.cuda()).sum().backward(); (ref * cot).sum().backward()
    assert rel_err(x.grad, xo.grad) < GRAD_TOL and rel_err(eps.grad, epo.grad) < GRAD_TOL
    x.grad = None; xo.grad = None
    outm = ops.spmm_max(csc, x); refm = O.spmm_max(src, dst, xo, n)
    assert torch.equal(outm.cpu(), refm.detach())                                # max is exact
    (outm * cot.cuda()).sum().backward(); (refm * cot).sum().backward()
    assert rel_err(x.grad, xo.grad) < GRAD_TOL


@pytest.mark.parametrize("F_", [64, 128, 256, 1024, 32])        # 32: eight float4 per row, no team shape - slot form
def test_spmm_max_compact_argmax_equals_the_slot_form(F_, monkeypatch):
    """The one-byte argmax (position inside the in-edge list, spgnn_spmm_max_fwd_u8 / _bwd_u8) routes every gradient exactly as
    the 32-bit CSC-slot form: values and gradients bit for bit - on trees, on a 70-child star (degree loop), with every edge
    tied, and with a node of in-degree 300 (no compact form: the op falls back by itself)."""
    torch.manual_seed(F_)
    star = lambda k: (np.concatenate([np.zeros(k, np.int64), np.arange(1, k + 1), np.arange(k + 1)]),
                      np.concatenate([np.arange(1, k + 1), np.zeros(k, np.int64), np.arange(k + 1)]), k + 1)
    s70, d70, n70 = star(70)
    s300, d300, n300 = star(300)
    graphs = [_graph([150, 33, 1, 90], seed=F_)[0], TreeGraph((s70, d70), n70).to("cuda"), TreeGraph((s300, d300), n300).to("cuda")]
    for gi, g in enumerate(graphs):
        csc = g.csc()
        n = csc.num_nodes
        for tied in (False, True):
            x0 = torch.ones(n, F_, device="cuda") if tied else torch.randn(n, F_, device="cuda").round(decimals=1)   # some ties too
            cot = torch.randn(n, F_, device="cuda")
            res = []
            for compact in (True, False):
                monkeypatch.setattr(ops, "COMPACT_MAX_ARG", compact)
                x = x0.clone().requires_grad_(True)
                y = ops.spmm_max(csc, x)
                assert y.grad_fn.u8 == bool(compact and gi < 2 and ops._capi.load().spgnn_spmm_max_u8_supported(F_))
                (y * cot).sum().backward()
                res.append((y.detach().clone(), x.grad.clone()))
            assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])


@pytest.mark.parametrize("fi,fo", [(1024, 256), (128, 64), (64, 1024)])
def test_sage_pool_front_as_one_node_equals_three(fi, fo, monkeypatch):
    """SAGEConv 'pool': fc_pool + ReLU + max aggregation as one autograd node whose routing kernel applies relu' itself
    (ops.pool_max, spgnn_spmm_max_bwd_u8_relu) - output and every gradient bit for bit those of linear -> spmm_max with the
    activation-backward pass in between."""
    torch.manual_seed(fi + fo)
    g, _, _, n = _graph([150, 170, 130, 160, 140], seed=4)
    layer = snn.SAGEConv(fi, fo, "pool", activation=F.elu).cuda()
    with torch.no_grad():
        layer.fc_pool.bias.normal_(0, 0.2)
    x0 = torch.randn(n, fi, device="cuda")
    res = {}
    for fused in (True, False):
        monkeypatch.setattr(ops, "POOL_MAX_FUSED", fused)
        x = x0.clone().requires_grad_(True)
        for p_ in layer.parameters():
            p_.grad = None
        y = layer(g, x)
        (y * torch.linspace(-1, 1, fo, device="cuda")).sum().backward()
        res[fused] = (y.detach().clone(), x.grad.clone(), {n_: p_.grad.clone() for n_, p_ in layer.named_parameters()})
    assert torch.equal(res[True][0], res[False][0]) and torch.equal(res[True][1], res[False][1])
    for n_ in res[True][2]:
        assert torch.equal(res[True][2][n_], res[False][2][n_]), n_
    assert float(res[True][2]["fc_pool.weight"].abs().max()) > 0


def test_spmm_max_ties_go_to_one_edge():
    g, src, dst, n = _graph([40], seed=2)
    x = torch.ones(n, 64, device="cuda", requires_grad=True)                     # every in-edge ties
    ops.spmm_max(g.csc(), x).sum().backward()
    assert torch.equal(x.grad.sum(0).cpu(), torch.full((64,), float(n)))        # each dst routes its grad once


@pytest.mark.parametrize("fi,fo", [(1024, 256), (64, 1024), (10, 10), (24, 7)])
@pytest.mark.parametrize("sizes", [[60, 150, 21], [150, 170, 130, 160, 140]])      # 231 nodes: torch products; 750: matrix-core ops.linear
def test_graphconv_gin_sage_modules(fi, fo, sizes):
    torch.manual_seed(fi + fo)
    g, src, dst, n = _graph(sizes, seed=fi)
    x = torch.randn(n, fi, device="cuda")
    gc = snn.GraphConv(fi, fo, activation=F.elu).cuda()
    with torch.no_grad():
        gc.bias.normal_(0, 0.1)
    sd = {k: v.cpu() for k, v in gc.state_dict().items()}
    assert rel_err(gc(g, x), O.graph_conv(src, dst, n, x.cpu(), sd["weight"], sd["bias"], F.elu)) < FWD_TOL
    mlp = torch.nn.Sequential(torch.nn.Linear(fi, fo), torch.nn.LeakyReLU()).cuda()
    gin = snn.GINConv(mlp, "mean", learn_eps=True).cuda()
    with torch.no_grad():
        gin.eps.fill_(0.3)
    ref = O.gin_conv(src, dst, n, x.cpu(), gin.eps.detach().cpu(), mlp.cpu(), "mean")
    mlp.cuda()
    assert rel_err(gin(g, x), ref) < FWD_TOL
    lin = torch.nn.Linear(fi, fo).cuda()                                    # bare nn.Linear: routed through ops.linear
    gin2 = snn.GINConv(lin, "mean", learn_eps=True).cuda()
    ref2 = O.gin_conv(src, dst, n, x.cpu(), gin2.eps.detach().cpu(), lin.cpu(), "mean")
    lin.cuda()
    assert rel_err(gin2(g, x), ref2) < FWD_TOL
    sg = snn.SAGEConv(fi, fo, "pool", activation=F.elu).cuda()
    sd = {k: v.cpu() for k, v in sg.state_dict().items()}
    ref = O.sage_conv_pool(src, dst, n, x.cpu(), sd["fc_pool.weight"], sd["fc_pool.bias"], sd["fc_self.weight"],
                           sd["fc_self.bias"], sd["fc_neigh.weight"], sd["fc_neigh.bias"], None, F.elu)
    assert rel_err(sg(g, x), ref) < FWD_TOL


def test_sgd_momentum_step_matches_torch():
    torch.manual_seed(0)
    p = torch.randn(100003, device="cuda"); ref = p.clone().requires_grad_(True)
    opt = torch.optim.SGD([ref], lr=0.05, momentum=0.9, weight_decay=1e-4)
    buf = torch.zeros_like(p)
    scale = torch.tensor([0.5], device="cuda")
    for step in range(4):
        gr = torch.randn_like(p)
        ref.grad = gr * 0.5
        opt.step()
        ops.sgd_momentum_step_(p, gr, buf, 0.05, 0.9, 1e-4, first_step=(step == 0), grad_scale=scale)
        assert rel_err(p, ref) < 1e-6


def test_sgd_mean_form_and_step_begin():
    """spgnn_sgd_momentum_step_mean (gradient divided by a device weight sum, loss scalar written by the same launch) against
    torch.optim.SGD; spgnn_step_begin re-arms the scale pool and advances the counter in one launch."""
    torch.manual_seed(1)
    p = torch.randn(50001, device="cuda"); ref = p.clone().requires_grad_(True)
    opt = torch.optim.SGD([ref], lr=0.05, momentum=0.9, weight_decay=1e-4)
    buf = torch.zeros_like(p)
    wsum, num, loss = torch.tensor([7.25], device="cuda"), torch.tensor([3.5], device="cuda"), torch.zeros(1, device="cuda")
    for step in range(3):
        gr = torch.randn_like(p)
        ref.grad = gr / 7.25
        opt.step()
        ops.sgd_momentum_step_(p, gr, buf, 0.05, 0.9, 1e-4, first_step=(step == 0), weight_sum=wsum, loss_num=num, loss_out=loss)
        assert rel_err(p, ref) < 1e-6
    assert abs(float(loss) - 3.5 / 7.25) < 1e-7
    pool = ops.scale_pool("cuda")
    ctr = torch.tensor([41], dtype=torch.int64, device="cuda")
    pool.buf.fill_(3.0)
    pool.begin(counter=ctr)
    try:
        assert int(ctr) == 42
        assert torch.equal(pool.buf, ops._scale_template(pool.device, pool.capacity))
        blk = ops.new_scale_block("cuda")
        assert blk.data_ptr() == pool.buf.data_ptr()
    finally:
        pool.end()


@pytest.mark.parametrize("N,K,C,act", [(3000, 96, 128, "LRELU"), (2500, 64, 1024, "LRELU"), (1111, 100, 132, "RELU"),
                                        (4100, 256, 512, "NONE"), (900, 64, 64, "ELU")])
def test_linear_with_dropout_equals_linear_then_hash_dropout(N, K, C, act):
    """ops.linear(..., drop=(p, seed)): the product's epilogue applies the hash dropout itself and the node undoes dropout +
    activation in one backward pass from the dropped result alone (spgnn_act_bwd_dropped) - values equal linear followed by
    ops.cat_dropout bit for bit, gradients too (ELU: to one rounding of the recovered value), interior and ragged tiles,
    every NT kernel."""
    torch.manual_seed(4)
    act = getattr(ops, "ACT_" + act)
    x0 = torch.randn(N, K, device="cuda")
    w0, b0 = torch.randn(C, K, device="cuda") * 0.1, torch.randn(C, device="cuda") * 0.1
    gout = torch.randn(N, C, device="cuda")
    res = []
    for fused in (True, False):
        x, w, b = x0.clone().requires_grad_(True), w0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
        if fused:
            y = ops.linear(x, w, b, act, drop=(0.1, 777))
        else:
            y = ops.cat_dropout((ops.linear(x, w, b, act),), 0.1, 777)
        assert getattr(y, "_spgnn_scale", None) is not None
        (y * gout).sum().backward()
        res.append((y.detach().clone(), x.grad.clone(), w.grad.clone(), b.grad.clone(), ops.scale_value(y._spgnn_scale[1])))
    assert torch.equal(res[0][0], res[1][0])
    for a, b_ in zip(res[0][1:4], res[1][1:4]):
        if act == ops.ACT_ELU:      # the derivative y + 1 comes from the dropped value times (1 - p): y to one rounding
            assert rel_err(a, b_) < 1e-6
        else:
            assert torch.equal(a, b_)
    assert res[0][4] == res[1][4]
    if act in (ops.ACT_LRELU, ops.ACT_NONE, ops.ACT_ELU):          # (ReLU zeroes half of the values by itself)
        kept = float((res[0][0] != 0).float().mean())
        assert 0.85 < kept < 0.95


@pytest.mark.parametrize("N,K,C,J,act", [(3000, 96, 128, 22, "LRELU"), (2049, 64, 1024, 22, "LRELU"), (777, 128, 64, 6, "RELU"),
                                          (1500, 64, 256, 3, "NONE")])
def test_linear_act_classifier_equals_the_three_nodes(N, K, C, J, act):
    """ops.linear_act_classifier (Linear + activation + skinny classifier as one node; backward through spgnn_act_bwd_proj when
    only the logits carry a gradient): same values as ops.linear -> ops.skinny_linear bit for bit, gradients to the last bits
    (g_logits Wc is summed in another order), and against fp64; a gradient into y as well takes the general route."""
    torch.manual_seed(11)
    act_c = getattr(ops, "ACT_" + act)
    x0 = torch.randn(N, K, device="cuda")
    w0, b0 = torch.randn(C, K, device="cuda") * 0.1, torch.randn(C, device="cuda") * 0.1
    wc0, bc0 = torch.randn(J, C, device="cuda") * 0.2, torch.randn(J, device="cuda") * 0.1
    gl, gy = torch.randn(N, J, device="cuda"), torch.randn(N, C, device="cuda") * 0.05
    ref_act = {"LRELU": lambda t: torch.nn.functional.leaky_relu(t, 0.01), "RELU": torch.relu, "NONE": lambda t: t}[act]
    for with_y in (False, True):
        res = []
        for form in ("fused", "split", "fp64"):
            dt = torch.float64 if form == "fp64" else torch.float32
            ps = [t.clone().to(dt).requires_grad_(True) for t in (x0, w0, b0, wc0, bc0)]
            x, w, b, wc, bc = ps
            if form == "fused":
                y, lg = ops.linear_act_classifier(x, w, b, act_c, wc, bc)
                assert getattr(y, "_spgnn_scale", None) is not None
            elif form == "split":
                y = ops.linear(x, w, b, act_c)
                lg = ops.skinny_linear(y, wc, bc)
            else:
                y = ref_act(torch.nn.functional.linear(x, w, b))
                lg = torch.nn.functional.linear(y, wc, bc)
            loss = (lg * gl.to(dt)).sum()
            if with_y:
                loss = loss + (y * gy.to(dt)).sum()
            loss.backward()
            res.append([y.detach(), lg.detach()] + [t.grad for t in ps])
        fused, split, ref = res
        assert torch.equal(fused[0], split[0]) and torch.equal(fused[1], split[1])
        for i, (a, b_, r) in enumerate(zip(fused, split, ref)):
            scale = float(r.abs().max()) + 1e-30
            e_f, e_s = float((a.double() - r).abs().max()) / scale, float((b_.double() - r).abs().max()) / scale
            assert e_f < 2e-5 and e_f <= 2.0 * e_s + 2e-6, (i, e_f, e_s)


@pytest.mark.parametrize("F_,act", [(256, "LRELU"), (64, "RELU"), (128, "NONE"), (1024, "LRELU")])
def test_spmm_sum_with_dropout_equals_spmm_sum_then_hash_dropout(F_, act):
    """ops.spmm_sum(..., drop=(p, seed)): the aggregation's epilogue applies bias, activation and the hash dropout, its node
    undoes dropout + activation in one backward pass from the dropped rows - bit for bit the two-node form."""
    torch.manual_seed(F_)
    act = getattr(ops, "ACT_" + act)
    g, _, _, n = _graph([150, 170, 130, 160], seed=5)
    csc = g.csc("cuda")
    x0, b0 = torch.randn(n, F_, device="cuda"), torch.randn(F_, device="cuda") * 0.1
    eps0 = torch.tensor([0.25], device="cuda")
    gout = torch.randn(n, F_, device="cuda")
    res = []
    for fused in (True, False):
        x, b, eps = x0.clone().requires_grad_(True), b0.clone().requires_grad_(True), eps0.clone().requires_grad_(True)
        w_dst = csc.degree_scale("in", -1.0)
        if fused:
            y = ops.spmm_sum(csc, x, None, w_dst, eps, bias=b, act=act, drop=(0.1, 4242))
        else:
            y = ops.cat_dropout((ops.spmm_sum(csc, x, None, w_dst, eps, bias=b, act=act),), 0.1, 4242)
        (y * gout).sum().backward()
        res.append((y.detach().clone(), x.grad.clone(), eps.grad.clone(), b.grad.clone(), ops.scale_value(y._spgnn_scale[1])))
    for a, b_ in zip(res[0][:2], res[1][:2]):
        assert torch.equal(a, b_)
    # the bias gradient (a column sum) and eps' gradient (a dot product) are summed in a different order by the two forms
    # (inside spgnn_act_bwd_colsum / by torch)
    assert float((res[0][3] - res[1][3]).abs().max()) <= 2e-6 * float(gout.abs().sum(0).max())
    assert abs(float(res[0][2]) - float(res[1][2])) <= 2e-6 * float((gout.abs() * x0.abs()).sum())
    assert res[0][4] == res[1][4]


@pytest.mark.parametrize("fi,fo,training", [(1024, 256, True), (256, 128, True), (128, 64, False)])
def test_ginconv_first_linear_before_the_aggregation(fi, fo, training, monkeypatch):
    """GINConv with in_feats > out_feats applies its MLP's first Linear before the aggregation (nn.GIN_PROJECT_FIRST; both are
    linear).  Same dropout seed -> same mask: values and every gradient equal the aggregate-first form within fp32 rounding,
    with and without the classifier joined to the last product."""
    from spgnn_amd import models
    torch.manual_seed(fi)
    g, _, _, n = _graph([150, 170, 130, 160, 140], seed=2)
    layer = snn.GINConv(models._gin_mlp(fi, fo), "mean", learn_eps=True).cuda()
    layer.train(training)
    cls = snn.SkinnyLinear(fo, 22).cuda()
    with torch.no_grad():
        layer.eps.fill_(0.2)
    x0 = torch.randn(n, fi, device="cuda")
    gl = torch.randn(n, 22, device="cuda")
    for with_cls in (False, True):
        res = {}
        for first in (True, False):
            monkeypatch.setattr(snn, "GIN_PROJECT_FIRST", first)
            monkeypatch.setattr(snn, "_draw_seed", lambda: 991)
            x = x0.clone().requires_grad_(True)
            for p_ in list(layer.parameters()) + list(cls.parameters()):
                p_.grad = None
            if with_cls:
                y, lg = layer(g, x, classifier=cls)
                (lg * gl).sum().backward()
            else:
                y = layer(g, x)
                (y * torch.linspace(-1, 1, fo, device="cuda")).sum().backward()
            res[first] = (y.detach(), x.grad.detach(),
                          {n_: p_.grad.detach().clone() for n_, p_ in list(layer.named_parameters()) + list(cls.named_parameters())
                           if p_.grad is not None})
        y1, gx1, gp1 = res[True]; y0, gx0, gp0 = res[False]
        assert rel_err(y1, y0) < 2e-6 and rel_err(gx1, gx0) < 5e-6
        assert set(gp1) == set(gp0) and "eps" in gp1
        for n_ in gp0:
            assert rel_err(gp1[n_], gp0[n_]) < 5e-6, n_


@pytest.mark.parametrize("N,W,act,p", [(5000, 256, "ELU", 0.0), (777, 64, "LRELU", 0.1), (3001, 1024, "RELU", 0.0), (130, 128, "TANH", 0.0),
                                        (9, 4, "LRELU", 0.3)])
def test_act_bwd_colsum_is_act_bwd_plus_column_sums(N, W, act, p):
    """spgnn_act_bwd_colsum: g_pre bit for bit as spgnn_act_bwd_dropout, plus per-block column sums whose ordered sum
    (spgnn_sum_partials) is the bias gradient - against an fp64 sum, and identical from run to run."""
    torch.manual_seed(N + W)
    lib = ops._capi.load()
    act_c = getattr(ops, "ACT_" + act)
    g = torch.randn(N, W, device="cuda")
    out = torch.randn(N, W, device="cuda").clamp_(-0.9, 0.9)
    ref = torch.empty_like(g)
    assert lib.spgnn_act_bwd_dropout(g.data_ptr(), W, out.data_ptr(), W, ref.data_ptr(), W, 0, N, W, act_c, p, 99, 0, 0) == 0
    nb = lib.spgnn_act_bwd_colsum_blocks(N, W)
    assert nb > 0 and lib.spgnn_act_bwd_colsum_blocks(N, 24) == 0          # 6 float4 per row do not divide 256
    xd = torch.randn(N, W, device="cuda")
    sums = []
    for rep in range(3):
        with_dot = rep < 2
        g_pre = torch.empty_like(g)
        P = W + 4 if with_dot else W
        part = torch.full((nb, P), float("nan"), device="cuda")
        blk = ops.new_scale_block(g.device)
        assert lib.spgnn_act_bwd_colsum(g.data_ptr(), W, out.data_ptr(), W, g_pre.data_ptr(), W, blk.data_ptr(), part.data_ptr(), N, W,
                                        act_c, p, 99, 0, xd.data_ptr() if with_dot else 0, W if with_dot else 0, 0) == 0
        assert torch.equal(g_pre, ref)
        assert ops.scale_value(blk) == float(ops.pow2_scale(ref))
        sums.append(ops.sum_partials(part))
    assert torch.equal(sums[0], sums[1]) and torch.equal(sums[0][:W], sums[2])
    want = ref.double().sum(0)
    assert float((sums[0][:W].double() - want).abs().max()) <= 1e-5 * float(ref.abs().double().sum(0).max())
    dot = float((ref.double() * xd.double()).sum())                      # the eps gradient riding along
    assert abs(float(sums[0][W]) - dot) <= 1e-5 * float((ref.double() * xd.double()).abs().sum())
    assert float(sums[0][W + 1:].abs().max()) == 0.0


def test_emitted_scales_equal_an_absmax_pass():
    """The scale block a product's epilogue (absmax_out) or spgnn_spmm_sum leaves behind gives exactly the power-of-two scale
    an absmax pass over the result gives - interior and ragged tiles, with and without an activation."""
    torch.manual_seed(9)
    for N, K, C, act in [(3000, 96, 128, ops.ACT_NONE), (2049, 200, 260, ops.ACT_LRELU), (700, 64, 1024, ops.ACT_ELU)]:
        x = torch.randn(N, K, device="cuda").requires_grad_(True)
        w, b = torch.randn(C, K, device="cuda") * 0.2, torch.randn(C, device="cuda")
        y = ops.linear(x, w, b, act)
        tag = getattr(y, "_spgnn_scale", None)
        assert tag is not None and tag[0] == y._version
        assert ops.scale_value(tag[1]) == float(ops.pow2_scale(y.detach()))
    g, _, _, n = _graph([41, 57, 33], seed=3)
    csc = g.csc("cuda")
    xs = (torch.randn(n, 64, device="cuda") * 3).requires_grad_(True)
    ys = ops.spmm_sum(csc, xs, None, csc.degree_scale("in", -1.0))
    assert ops.scale_value(ys._spgnn_scale[1]) == float(ops.pow2_scale(ys.detach()))


def test_c_abi_argument_errors_are_reported():
    from spgnn_amd import _capi
    lib = _capi.load()
    assert lib.spgnn_gat_fwd(0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 5, 5, 2, 4, 0.2, 0, 0.0, 0, 0, 0.0, 0, 0, 0, 0, 0) == -1   # null pointers
    assert b"null" in lib.spgnn_last_error()
    assert lib.spgnn_gat_fwd(0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, -1, 0, 2, 4, 0.2, 0, 0.0, 0, 0, 0.0, 0, 0, 0, 0, 0) == -2  # bad shape
    assert lib.spgnn_spmm_sum(0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 8, 0, 0) == 0                                   # N == 0 is a no-op


@pytest.mark.parametrize("K,J", [(1063, 4), (1064, 4), (39, 2), (768, 4), (192, 4), (256, 2), (64, 16), (100, 8), (17, 4), (600, 22), (520, 4), (1024, 22)])
def test_score_projection_kernels(K, J):
    """spgnn_scores_fwd / _bwd_w / _bwd_x against plain matmuls (ragged K tails, padded row strides)."""
    torch.manual_seed(K + J)
    N = 2531
    buf = torch.randn(N, (K + 3) // 4 * 4 + 4, device="cuda")
    x = buf[:, :K]                                         # 16-byte-aligned rows, stride > K
    w = torch.randn(J, K, device="cuda")
    s = ops.scores_fwd(x, w)
    assert rel_err(s, x.double() @ w.double().t()) < 2e-6
    s2, scale = ops.scores_fwd(x, w, want_scale=True)      # the absmax the kernel collects on the way (K >= 512: four waves per row group)
    assert torch.equal(s2, s) and ops.scale_value(scale) == float(ops.pow2_scale(x))
    bias = torch.randn(J, device="cuda")                   # the optional bias rides in the store: s + bias exactly
    assert torch.equal(ops.scores_fwd(x, w, bias=bias), s + bias)
    gs = torch.randn(N, J, device="cuda")
    assert rel_err(ops.scores_bwd_w(gs, x), gs.double().t() @ x.double()) < 2e-6
    gbuf = torch.randn(N, (K + 3) // 4 * 4, device="cuda")
    gx = gbuf[:, :K]
    ref = gx.double() + gs.double() @ w.double()
    ops.scores_bwd_x_(gx, gs, w)
    assert rel_err(gx, ref) < 2e-6
    if gbuf.shape[1] > K:
        assert torch.isfinite(gbuf).all()
    xu = torch.randn(N, K, device="cuda")                  # unaligned rows (K odd) fall back to rocBLAS
    assert rel_err(ops.scores_fwd(xu, w), xu.double() @ w.double().t()) < 2e-6


@pytest.mark.parametrize("K,J", [(1024, 22), (600, 22), (100, 17), (384, 32)])
def test_score_projection_forward_two_row_groups_per_wave(K, J):
    """spgnn_scores_fwd with J > 16 on >= 32 768 rows takes the form with two 16-row groups per wave (W fragments reused):
    ragged row count, ragged K, the absmax by-product, fp32 and bf16 rows."""
    torch.manual_seed(K * J)
    N = 33000 + 13
    x = torch.randn(N, (K + 3) // 4 * 4 + 4, device="cuda")[:, :K]
    w = torch.randn(J, K, device="cuda")
    s, scale = ops.scores_fwd(x, w, want_scale=True)
    assert rel_err(s, x.double() @ w.double().t()) < 2e-6 and ops.scale_value(scale) == float(ops.pow2_scale(x))
    assert float(scale[ops.SCALE_HEADER:].max()) == float(x.abs().max())
    from spgnn_amd import ops_bf16
    xb = ops_bf16.cast_rows(x.contiguous())
    assert rel_err(ops_bf16.scores_fwd(xb, w), xb.double() @ w.double().t()) < 2e-6
    bias = torch.randn(J, device="cuda")
    assert torch.equal(ops.scores_fwd(x, w, bias=bias), s + bias)
    assert torch.equal(ops_bf16.scores_fwd(xb, w, bias=bias), ops_bf16.scores_fwd(xb, w) + bias)


@pytest.mark.parametrize("K,J", [(1024, 22), (100, 17), (64, 32), (1024, 6)])
def test_skinny_linear_matches_nn_linear(K, J):
    """The classifier head gnn_out = Linear(1024, 22) on the streaming kernels: same parameters, same results."""
    torch.manual_seed(J)
    lin = snn.SkinnyLinear(K, J).cuda()
    ref = torch.nn.Linear(K, J).cuda(); ref.load_state_dict(lin.state_dict())
    x = torch.randn(3001, K, device="cuda", requires_grad=True)
    xr = x.detach().clone().requires_grad_(True)
    y, yr = lin(x), ref(xr)
    assert rel_err(y, yr) < 2e-6
    cot = torch.randn_like(y)
    (y * cot).sum().backward(); (yr * cot).sum().backward()
    assert rel_err(x.grad, xr.grad) < 2e-6 and rel_err(lin.weight.grad, ref.weight.grad) < 5e-6
    assert rel_err(lin.bias.grad, ref.bias.grad) < 2e-6


def test_device_positional_encoding_is_bit_exact():
    """spgnn_tree_distance_encoding == host restatement == networkx (reference job_runner.py:1759-1777)."""
    from oracle import graph_rule_nx as R
    from spgnn_amd import posenc, synthetic
    samples = synthetic.synthetic_trees(24, rank=5, n_lo=21, n_hi=300)
    g = synthetic.batch_from_samples(samples, "cuda", 39, device_posenc=False)
    anchors = [posenc.anchors_from_cnn_prediction(s["fvs_out"], s["adj"], 39) for s in samples]
    pe, diam = posenc.distance_pos_enc_device(g, anchors)
    assert torch.equal(pe.cpu(), g.ndata["pos_enc"].cpu())                       # vs the host path (bitwise)
    for i in (0, 7, 23):
        ref, d = R.distance_pos_enc(samples[i]["adj"], anchors[i])              # vs networkx
        off = sum(s["adj"].shape[0] for s in samples[:i])
        assert np.array_equal(pe[off:off + ref.shape[0]].cpu().numpy(), ref) and int(diam[i]) == d
    # a path graph (diameter n-1) and a star, single anchor
    from spgnn_amd.graph import graph_from_adj, batch
    n = 50
    path = np.eye(n, dtype=np.uint8); idx = np.arange(n - 1); path[idx, idx + 1] = 1; path[idx + 1, idx] = 1
    star = np.eye(n, dtype=np.uint8); star[0, 1:] = 1; star[1:, 0] = 1
    gb = batch([graph_from_adj(path), graph_from_adj(star)]).to("cuda")
    pe2, d2 = posenc.distance_pos_enc_device(gb, [[0, n - 1], [0, 5]])
    assert d2.tolist() == [n - 1, 2]
    assert torch.equal(pe2[:n, 0].cpu(), torch.tensor([(i / (n - 1)) for i in range(n)], dtype=torch.float64).float())
    assert pe2[n:, 1].cpu().tolist() == [0.5] + [1.0] * 4 + [0.0] + [1.0] * (n - 6)


@pytest.mark.parametrize("N,H,D,J,act", [(1000, 2, 1024, 22, "elu"), (257, 2, 1024, 22, "none"), (300, 1, 64, 3, "tanh"),
                                         (129, 4, 256, 32, "relu"), (5, 2, 1000, 9, "elu")])
def test_act_bwd_with_the_classifier_gradient_formed_on_the_fly(N, H, D, J, act):
    """spgnn_act_bwd_proj = spgnn_scores_bwd_x (g_mean = g_logits W) followed by spgnn_act_bwd, without g_mean in memory."""
    from spgnn_amd import ops
    code = {"none": ops.ACT_NONE, "elu": ops.ACT_ELU, "tanh": ops.ACT_TANH, "relu": ops.ACT_RELU}[act]
    g_s = torch.randn(N, J, device="cuda")
    w = torch.randn(J, D, device="cuda") * 0.1
    pre = torch.randn(N, H * D, device="cuda")
    out = {"none": pre, "elu": torch.nn.functional.elu(pre), "tanh": torch.tanh(pre), "relu": torch.relu(pre)}[act]
    g_pre, part = ops.act_bwd_proj(g_s, w, out if code != ops.ACT_NONE else None, H, D, code)
    g_mean = (g_s.double() @ w.double()) / H
    dact = {"none": torch.ones_like(pre), "elu": torch.where(pre > 0, torch.ones_like(pre), torch.exp(pre)),
            "tanh": 1 - torch.tanh(pre) ** 2, "relu": (pre > 0).float()}[act].double()
    ref = g_mean.repeat(1, H) * dact
    assert rel_err(g_pre, ref) < 2e-6
    assert abs(float(part.max()) - float(g_pre.abs().max())) <= 1e-6 * float(g_pre.abs().max())


@pytest.mark.parametrize("kind", ["gcn_hidden", "gcn_out", "gin", "sage_pool", "sage_mean"])
def test_fused_epilogues_equal_separate_passes(kind, monkeypatch):
    """bias / activation of GraphConv inside spgnn_spmm_sum's or the GEMM's epilogue, the GIN MLP's
    LeakyReLUs inside its two products (SPGNN_ACT_LRELU; across the Dropout: act(dropout(y)) == dropout(act(y))), SAGEConv's
    fc_self + fc_neigh with the addition and the activation in the second product's epilogue - against the same layers with
    the epilogues as separate torch passes (nn.FUSE_EPILOGUES = False): forward and every gradient within fp32 rounding."""
    from spgnn_amd import models, synthetic
    torch.manual_seed(11)
    g = synthetic.make_batch(6, rank=1, device="cuda", pos_enc_dim=None, fv_dim=8)
    N = g.number_of_nodes()
    if kind == "gcn_hidden":
        layer, fin = snn.GraphConv(256, 128, activation=F.elu).cuda(), 256
    elif kind == "gcn_out":
        layer, fin = snn.GraphConv(64, 256, activation=torch.tanh).cuda(), 64
    elif kind == "gin":
        layer, fin = snn.GINConv(models._gin_mlp(128, 256), "mean", learn_eps=True).cuda(), 128
        layer.eval()                                            # (the Dropout inside the MLP draws from torch's generator)
    else:
        layer, fin = snn.SAGEConv(128, 64, aggregator_type=kind.split("_")[1], activation=F.elu).cuda(), 128
    with torch.no_grad():
        for n_, p in layer.named_parameters():
            if n_.endswith("bias"):
                p.normal_(0, 0.1)
    x0 = torch.randn(N, fin, device="cuda")
    res = {}
    for fused in (True, False):
        monkeypatch.setattr(snn, "FUSE_EPILOGUES", fused)
        x = x0.clone().requires_grad_(True)
        for p in layer.parameters():
            p.grad = None
        y = layer(g, x)
        (y * torch.linspace(-1, 1, y.shape[1], device="cuda")).sum().backward()
        res[fused] = (y.detach(), x.grad.detach(), {n_: p.grad.detach().clone() for n_, p in layer.named_parameters() if p.grad is not None})
    y1, gx1, gp1 = res[True]; y0, gx0, gp0 = res[False]
    assert rel_err(y1, y0) < 2e-6 and rel_err(gx1, gx0) < 5e-6
    assert set(gp1) == set(gp0)
    for n_ in gp0:
        assert rel_err(gp1[n_], gp0[n_]) < 5e-6, n_
