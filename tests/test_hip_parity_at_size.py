"""Oracle parity AT THE BASELINE BATCH SIZES: the model-level comparisons of test_hip_models.py run on
2-5 trees; here the same comparisons run on 64 trees (TRAIN_BATCH_SIZE, reference exp_settings/st_pgat_spgnn_3.py:29) -
logits, loss and every parameter gradient against the fp32 and the fp64 oracle - and on 512 trees (BASELINE.json's headline
batch) - logits against the fp32 oracle.  Both the normwise bar (max|a-b| / max|b| <= 1e-5, BASELINE.json north_star) and an
ELEMENTWISE mixed bound (|a-b| <= 1e-5 |b| + 1e-5 rms(b)) are asserted; the measured values are printed (-s) so the metric is
on record.  Forward reference: models.py:472-484 (GATPSPGNN.forward), 321-329 (GAT.forward)."""
import pytest
import torch

from oracle import dgl_cpu as O
from spgnn_amd import models, synthetic
from spgnn_amd.configs import class_weight_list, get_config
from spgnn_amd.train import masked_weighted_ce
from tests.util import mixed_err, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _build(name, seed=0):
    cfg = get_config(name)
    torch.manual_seed(seed)
    model = models.build_model(cfg.MODEL).cuda()
    model.init(None)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith("bias"):
                p.normal_(0, 0.05)
    model.set_gcn_only()
    model.eval()                                   # dropout off: the oracle has no mask stream to share
    return cfg, model


def _oracle(cfg, model, g, dtype=torch.float32, grad=False):
    src, dst = g.cpu().edges()
    sd = {k: v.detach().cpu().to(dtype if v.dtype.is_floating_point else v.dtype).requires_grad_(grad and v.dtype.is_floating_point)
          for k, v in model.state_dict().items()}
    pe = g.ndata["pos_enc"].cpu().to(dtype) if "pos_enc" in g.ndata else None
    return O.net_forward(cfg.KIND, sd, src, dst, g.number_of_nodes(), g.ndata["fvs"].cpu().to(dtype), pe), sd


@pytest.mark.parametrize("name", ["st_gat_3", "st_pgat_spgnn_3", "st_gcn_3"])
def test_logits_loss_and_gradients_match_the_oracle_at_64_trees(name):
    cfg, model = _build(name, seed=11)
    g = synthetic.make_batch(64, rank=0, device="cuda", pos_enc_dim=getattr(cfg, "POS_ENC_DIM", None))
    assert g.number_of_nodes() > 9000
    w = torch.tensor(class_weight_list(cfg.CLASS_WEIGHTS))
    y = g.ndata["y"]
    mask = torch.rand(y.shape[0], generator=torch.Generator().manual_seed(5)) < torch.where(y.cpu() != 0, torch.tensor(1.0), torch.tensor(cfg.SAMPLING_RATE))
    outs = model(g)
    loss = masked_weighted_ce(outs[0], y, mask.cuda(), w.cuda())
    loss.backward()
    refs, sd = _oracle(cfg, model, g, grad=True)
    ref_loss = O.masked_weighted_ce(refs[0], y.cpu(), mask, w)
    ref_loss.backward()
    refs64, sd64 = _oracle(cfg, model, g, dtype=torch.float64, grad=True)
    O.masked_weighted_ce(refs64[0], y.cpu(), mask, w.double()).backward()
    # forward: every output of the head (logits, embedding[, position embedding])
    for o, r, r64 in zip(outs, refs, refs64):
        e_n, e_m = rel_err(o, r), mixed_err(o, r)
        print(f"{name} 64 trees {tuple(o.shape)}: normwise {e_n:.2e}, elementwise mixed {e_m:.2e}; vs fp64 {rel_err(o, r64):.2e} "
              f"(fp32 oracle vs fp64 {rel_err(r, r64):.2e})")
        assert e_n < TOL and e_m < TOL
        assert rel_err(o, r64) < TOL
    assert rel_err(loss, ref_loss) < TOL
    # every parameter gradient (rule of test_hip_models.test_config_loss_gradients_match_oracle): 1e-4 normwise against the
    # fp32 oracle, or no further from fp64 than 5 x the fp32 oracle's own distance.  GIN has its own test below.
    worst = _assert_gradients(model, sd, sd64, 1e-4)
    print(f"{name} 64 trees: worst non-tiny gradient normwise error vs the fp32 oracle {worst:.2e}")


def _assert_gradients(model, sd, sd64, gtol, only=None):
    """Every trainable parameter's gradient against the fp32 oracle's (``sd``) at ``gtol`` normwise - or within 1e-7 of the
    largest gradient, or no further from the fp64 oracle's (``sd64``) than 5 x the fp32 oracle is.  ``only``: a predicate on
    the parameter name.  -> the worst non-tiny normwise error."""
    gmax = max(float(v.grad.abs().max()) for v in sd64.values() if v.grad is not None)
    worst = 0.0
    for n, p in model.named_parameters():
        if not p.requires_grad or (p.grad is None and sd[n].grad is None) or (only is not None and not only(n)):
            continue
        assert p.grad is not None and sd[n].grad is not None, n
        e32 = rel_err(p.grad, sd[n].grad)
        tiny = (p.grad.cpu().double() - sd64[n].grad).abs().max() < 1e-7 * gmax
        ok = e32 < gtol or tiny or rel_err(p.grad, sd64[n].grad) < 5 * rel_err(sd[n].grad, sd64[n].grad) + 1e-6
        worst = max(worst, 0.0 if tiny else e32)
        assert ok, (n, e32, rel_err(p.grad, sd64[n].grad), rel_err(sd[n].grad, sd64[n].grad))
    return worst


def _gin_flipped_units(model, g):
    """st_gin_3: per GINConv layer, the output units whose LeakyReLU branch in the HIP forward differs from an fp64 evaluation
    of the same layer stack (plain torch on the GPU; reference models.py:358-383: mean aggregation incl. the self loop,
    (1 + eps) x + agg, Linear -> LeakyReLU -> Linear -> LeakyReLU in eval mode) -> [(layer, count, max |pre64| / rms)]."""
    layers = list(model.gin.gin_layers)
    outs = []
    hooks = [l.register_forward_hook(lambda m, i, o: outs.append((o[0] if isinstance(o, tuple) else o).detach())) for l in layers]
    with torch.no_grad():
        model(g)
    for h in hooks:
        h.remove()
    src, dst = g.edges()
    src, dst = src.long(), dst.long()
    x = g.ndata["fvs"].double()
    deg = torch.zeros(x.shape[0], dtype=torch.float64, device=x.device).index_add_(0, dst, torch.ones_like(dst, dtype=torch.float64))
    found = []
    for li, (l, o) in enumerate(zip(layers, outs)):
        f = l.apply_func
        agg = torch.zeros_like(x).index_add_(0, dst, x[src]) / deg.clamp(min=1)[:, None]
        h = (1 + l.eps.double()) * x + agg
        h = torch.nn.functional.leaky_relu(h @ f[0].weight.double().t() + f[0].bias.double(), 0.01)
        pre = h @ f[3].weight.double().t() + f[3].bias.double()
        flips = (o > 0) != (pre > 0)
        n = int(flips.sum())
        found.append((li, n, float(pre[flips].abs().max() / pre.pow(2).mean().sqrt()) if n else 0.0))
        x = torch.nn.functional.leaky_relu(pre, 0.01)
    return found


def test_gin_gradients_at_64_trees_hold_the_bar_outside_the_leaky_relu_kink(monkeypatch):
    """VERDICT r4 item 6 / ADVICE r4: GIN's gradients were bounded at 1e-2 at this size because LeakyReLU's derivative jumps at
    zero and one or two of the 2.5 M hidden units of a 64-tree batch sit within 1e-8 of it - a bound that would also hide a
    real indexing error.  Now: (1) the flipped units are FOUND (HIP output sign vs an fp64 evaluation), there are at most a
    handful and each has |pre| < 1e-6 rms - so a flip is conditioning, not a kernel error; (2) every parameter the flipped
    units cannot reach - the layers AFTER the last flipped one and the classifier: their gradients depend on the forward
    values and on the signal from the loss only - holds 1e-4 in the default form; (3) in the products' wide-range form
    (ops.GEMM_WIDE; same traversal, activation and reduction kernels, another rounding of the products), which keeps these
    units' branches, EVERY gradient holds 1e-4 (measured <= 8e-6)."""
    from spgnn_amd import ops
    name = "st_gin_3"
    cfg, model = _build(name, seed=11)
    g = synthetic.make_batch(64, rank=0, device="cuda", pos_enc_dim=None)
    w = torch.tensor(class_weight_list(cfg.CLASS_WEIGHTS))
    y = g.ndata["y"]
    mask = torch.rand(y.shape[0], generator=torch.Generator().manual_seed(5)) < torch.where(y.cpu() != 0, torch.tensor(1.0), torch.tensor(cfg.SAMPLING_RATE))
    refs, sd = _oracle(cfg, model, g, grad=True)
    O.masked_weighted_ce(refs[0], y.cpu(), mask, w).backward()
    refs64, sd64 = _oracle(cfg, model, g, dtype=torch.float64, grad=True)
    O.masked_weighted_ce(refs64[0], y.cpu(), mask, w.double()).backward()

    def run():
        model.zero_grad(set_to_none=True)
        outs = model(g)
        loss = masked_weighted_ce(outs[0], y, mask.cuda(), w.cuda())
        loss.backward()
        return outs, loss

    # default (narrow) form
    flips = _gin_flipped_units(model, g)
    outs, loss = run()
    assert rel_err(outs[0], refs[0]) < TOL and rel_err(loss, O.masked_weighted_ce(refs[0], y.cpu(), mask, w)) < TOL
    total = sum(n for _, n, _ in flips)
    print(f"st_gin_3 64 trees, default form: LeakyReLU branch flips per layer {[(li, n, f'{r:.1e}') for li, n, r in flips]}")
    assert total <= 8 and all(r < 1e-6 for _, _, r in flips), flips
    last = max([li for li, n, _ in flips if n], default=-1)
    clean = lambda pn: not pn.startswith("gin.gin_layers.") or int(pn.split(".")[2]) > last
    worst_clean = _assert_gradients(model, sd, sd64, 1e-4, only=clean)
    worst_all = _assert_gradients(model, sd, sd64, 1e-2)              # behind a flipped unit: the old bound, now with its reason on record
    print(f"   parameters no flipped unit reaches (layers > {last} + classifier): worst {worst_clean:.2e}; all: {worst_all:.2e}")
    # wide-range form
    monkeypatch.setattr(ops, "GEMM_WIDE", True)
    flips_w = _gin_flipped_units(model, g)
    outs, loss = run()
    assert rel_err(outs[0], refs[0]) < TOL
    print(f"st_gin_3 64 trees, wide-range form: flips {[(li, n) for li, n, _ in flips_w]}")
    if sum(n for _, n, _ in flips_w) == 0:
        worst_w = _assert_gradients(model, sd, sd64, 1e-4)
        print(f"   every gradient, wide-range form: worst {worst_w:.2e}")
    else:                                                             # another box / library build may round these units the other way:
        last_w = max(li for li, n, _ in flips_w if n)                 # then the same reach argument applies to this form
        _assert_gradients(model, sd, sd64, 1e-4, only=lambda pn: not pn.startswith("gin.gin_layers.") or int(pn.split(".")[2]) > last_w)


@pytest.mark.parametrize("name", ["st_gcn_3", "st_gin_3", "st_sage_3", "st_pgat_spgnnnl_3", "st_gat_6", "st_gat_1"])
def test_every_other_head_matches_the_oracle_at_64_trees(name):
    """The remaining model families of north_star (GCN / GIN / SAGE, the PENL ablation, the deep and the one-layer GAT) at the
    reference's TRAIN_BATCH_SIZE: every output of the head against the fp32 oracle, normwise and elementwise."""
    cfg, model = _build(name, seed=13)
    g = synthetic.make_batch(64, rank=1, device="cuda", pos_enc_dim=getattr(cfg, "POS_ENC_DIM", None))
    assert g.number_of_nodes() > 9000
    with torch.no_grad():
        outs = model(g)
        refs, _ = _oracle(cfg, model, g)
    outs = outs if isinstance(outs, (tuple, list)) else (outs,)
    refs = refs if isinstance(refs, (tuple, list)) else (refs,)
    for o, r in zip(outs, refs):
        e_n, e_m = rel_err(o, r), mixed_err(o, r)
        print(f"{name} 64 trees {tuple(o.shape)}: normwise {e_n:.2e}, elementwise mixed {e_m:.2e}")
        assert o.shape == r.shape and e_n < TOL and e_m < 2 * TOL


@pytest.mark.parametrize("name", ["st_pgat_spgnn_3", "st_gat_6"])
def test_logits_match_the_oracle_at_512_trees(name):
    cfg, model = _build(name, seed=12)
    g = synthetic.make_batch(512, rank=0, device="cuda", pos_enc_dim=cfg.POS_ENC_DIM)
    assert g.number_of_nodes() > 70000
    with torch.no_grad():
        outs = model(g)
        refs, _ = _oracle(cfg, model, g)
    for i, (o, r) in enumerate(zip(outs, refs)):
        e_n, e_m = rel_err(o, r), mixed_err(o, r)
        print(f"{name} 512 trees {tuple(o.shape)}: normwise {e_n:.2e}, elementwise mixed {e_m:.2e}")
        # the logits (output 0: BASELINE.json's bar) hold the elementwise bound at 1e-5 too; the 78 M elements of the
        # (N, 1024) embedding behind seven layers reach 1.25e-5 in their worst element against the fp32 oracle (measured,
        # MI355X; normwise 2.3e-6) - two fp32 evaluations in different summation orders - and are bounded at 2e-5
        assert o.shape == r.shape and e_n < TOL and e_m < (TOL if i == 0 else 2 * TOL)


def test_loss_and_gradients_match_the_oracle_at_512_trees():
    """the headline batch (st_pgat_spgnn_3, 512 trees, N = 76 410) with loss and EVERY parameter gradient
    against the fp32 oracle (about a minute of CPU work; the fp64 evaluation of the 64-tree test is left out here, so the bar
    is the plain one: 1e-4 normwise per tensor, or within 1e-7 of the largest gradient)."""
    name = "st_pgat_spgnn_3"
    cfg, model = _build(name, seed=12)
    g = synthetic.make_batch(512, rank=0, device="cuda", pos_enc_dim=cfg.POS_ENC_DIM)
    assert g.number_of_nodes() > 70000
    w = torch.tensor(class_weight_list(cfg.CLASS_WEIGHTS))
    y = g.ndata["y"]
    mask = torch.rand(y.shape[0], generator=torch.Generator().manual_seed(5)) < torch.where(y.cpu() != 0, torch.tensor(1.0), torch.tensor(cfg.SAMPLING_RATE))
    outs = model(g)
    loss = masked_weighted_ce(outs[0], y, mask.cuda(), w.cuda())
    loss.backward()
    refs, sd = _oracle(cfg, model, g, grad=True)
    ref_loss = O.masked_weighted_ce(refs[0], y.cpu(), mask, w)
    ref_loss.backward()
    assert rel_err(outs[0], refs[0]) < TOL and mixed_err(outs[0], refs[0]) < TOL
    e_loss = rel_err(loss, ref_loss)
    gmax = max(float(v.grad.abs().max()) for v in sd.values() if v.grad is not None)
    worst, worst_name = 0.0, ""
    for n, p in model.named_parameters():
        if not p.requires_grad or (p.grad is None and sd[n].grad is None):
            continue
        assert p.grad is not None and sd[n].grad is not None, n
        e = rel_err(p.grad, sd[n].grad)
        tiny = float((p.grad.cpu() - sd[n].grad).abs().max()) < 1e-7 * gmax
        if not tiny and e > worst:
            worst, worst_name = e, n
        assert e < 1e-4 or tiny, (n, e)
    print(f"{name} 512 trees: loss {e_loss:.2e}, worst non-tiny gradient normwise error vs the fp32 oracle {worst:.2e} ({worst_name})")
    assert e_loss < TOL
