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


class _Gin64:
    """st_gin_3's head + classifier + loss in fp64 on the GPU in plain torch, with the branch of chosen LeakyReLU units
    TOGGLED (reference models.py:343-400, 358-383: GINConv 'mean' incl. the self loop, (1 + eps) x + agg, Linear -> [Dropout,
    eval: identity] -> LeakyReLU -> Linear -> LeakyReLU; gnn_out models.py:988; loss job_runner.py:1400).  Test infrastructure:
    pinned to oracle/dgl_cpu.py by comparing its untoggled gradients with the CPU oracle's in the test."""

    def __init__(self, model, g, y, mask, w):
        self.names = [n for n, p in model.named_parameters() if p.requires_grad and (n.startswith("gin.") or n.startswith("gnn_out."))]
        self.p = {n: q.detach().double() for n, q in model.named_parameters() if n in self.names}
        src, dst = g.edges()
        self.src, self.dst = src.long(), dst.long()
        self.x0 = g.ndata["fvs"].double()
        self.deg = torch.zeros(self.x0.shape[0], dtype=torch.float64, device=self.x0.device).index_add_(
            0, self.dst, torch.ones_like(self.dst, dtype=torch.float64)).clamp(min=1)[:, None]
        self.y, self.mask, self.w = y.cuda(), mask.cuda(), w.double().cuda()
        self.layers = len(model.gin.gin_layers)

    def run(self, toggles=()):
        """-> ({name: grad}, [(layer, stage, pre-activation)]); ``toggles``: (layer, stage, flat unit index) whose branch is
        the OTHER one than the sign of its fp64 pre-activation."""
        P = {n: q.clone().requires_grad_(True) for n, q in self.p.items()}
        tog = {}
        for (l, st, idx) in toggles:
            tog.setdefault((l, st), []).append(idx)
        pres = []

        def lrelu(v, key):
            pos = v > 0
            if key in tog:
                flat = pos.clone().view(-1)
                ii = torch.tensor(tog[key], device=v.device)
                flat[ii] = ~flat[ii]
                pos = flat.view_as(pos)
            pres.append((key[0], key[1], v.detach()))
            return torch.where(pos, v, 0.01 * v)
        x = self.x0
        for l in range(self.layers):
            pre = f"gin.gin_layers.{l}."
            agg = torch.zeros_like(x).index_add_(0, self.dst, x[self.src]) / self.deg
            h = (1 + P[pre + "eps"]) * x + agg
            h = lrelu(h @ P[pre + "apply_func.0.weight"].t() + P[pre + "apply_func.0.bias"], (l, 0))
            x = lrelu(h @ P[pre + "apply_func.3.weight"].t() + P[pre + "apply_func.3.bias"], (l, 1))
        logits = x @ P["gnn_out.weight"].t() + P["gnn_out.bias"]
        lp = torch.log_softmax(logits, 1)
        wy = self.w[self.y] * self.mask.double()
        loss = -(wy * lp.gather(1, self.y[:, None])[:, 0]).sum() / wy.sum()
        grads = torch.autograd.grad(loss, [P[n] for n in self.names])
        return {n: q.detach() for n, q in zip(self.names, grads)}, pres


def test_gin_gradients_at_64_trees_equal_the_oracle_up_to_identified_leaky_relu_branches(monkeypatch):
    """VERDICT r4 item 6 / ADVICE r4.  GIN's gradients were bounded at 1e-2 at this size: LeakyReLU's derivative jumps at zero,
    a 64-tree batch has 22 M LeakyReLU units, a handful of them sit within 1e-6 rms of zero, and an fp32 evaluation in another
    summation order takes the other branch there - each such unit moves the gradients behind it by up to a few 1e-3.  A 1e-2
    bound would also hide a real indexing error, so the comparison is now EXACT about the branches: the ambiguous units are
    enumerated from the fp64 pre-activations (|pre| < 3e-6 rms: a few dozen), the gradient change of toggling each ONE is
    computed in fp64 (a unit's effect is additive to second order), the subset the HIP evaluation took is fitted greedily, and
    every parameter's gradient must then agree at 1e-4 - in the default form of the split products and in the wide-range one.
    Nothing but the branch of those specific units can be explained away: an indexing error is not in their span."""
    from spgnn_amd import ops
    cfg, model = _build("st_gin_3", seed=11)
    g = synthetic.make_batch(64, rank=0, device="cuda", pos_enc_dim=None)
    w = torch.tensor(class_weight_list(cfg.CLASS_WEIGHTS))
    y = g.ndata["y"]
    mask = torch.rand(y.shape[0], generator=torch.Generator().manual_seed(5)) < torch.where(y.cpu() != 0, torch.tensor(1.0), torch.tensor(cfg.SAMPLING_RATE))
    refs, sd = _oracle(cfg, model, g, grad=True)
    O.masked_weighted_ce(refs[0], y.cpu(), mask, w).backward()
    refs64, sd64 = _oracle(cfg, model, g, dtype=torch.float64, grad=True)
    O.masked_weighted_ce(refs64[0], y.cpu(), mask, w.double()).backward()
    # the fp64 restatement used for the toggles IS the oracle: untoggled, its gradients equal the CPU fp64 oracle's
    net = _Gin64(model, g, y, mask, w)
    g0, pres = net.run()
    for n in net.names:
        assert rel_err(g0[n], sd64[n].grad) < 1e-9, n
    # ambiguous units: within 3e-6 rms of the kink (HIP's fp32 pre-activations are ~1e-7 relative off the fp64 ones)
    amb = []
    for (l, st, pre) in pres:
        rms = pre.pow(2).mean().sqrt()
        idx = torch.nonzero((pre.abs() < 3e-6 * rms).view(-1))[:, 0].tolist()
        amb += [(l, st, i) for i in idx]
    print(f"st_gin_3 64 trees: {sum(p_[2].numel() for p_ in pres) / 1e6:.1f} M LeakyReLU units, {len(amb)} within 3e-6 rms of zero "
          f"(per layer/stage: {sorted({(l, st): sum(1 for a in amb if a[:2] == (l, st)) for (l, st, _) in amb}.items())})")
    assert len(amb) <= 400
    deltas = []
    for a in amb:
        gi, _ = net.run([a])
        deltas.append({n: gi[n] - g0[n] for n in net.names})
    scale = {n: float(g0[n].abs().max()) + 1e-300 for n in net.names}

    def fit_and_check(tag):
        model.zero_grad(set_to_none=True)
        outs = model(g)
        masked_weighted_ce(outs[0], y, mask.cuda(), w.cuda()).backward()
        assert rel_err(outs[0], refs[0]) < TOL
        got = {n: dict(model.named_parameters())[n].grad.double() for n in net.names}
        res = {n: got[n] - g0[n] for n in net.names}
        norm = lambda r: sum(float((r[n] / scale[n]).pow(2).sum()) for n in net.names)
        raw = max(float(res[n].abs().max()) / scale[n] for n in net.names)
        taken = []
        for _ in range(3):                               # greedy, a few sweeps: toggle a unit whenever that shrinks the residual
            changed = False
            for i, d in enumerate(deltas):
                sign = -1.0 if i in taken else 1.0
                trial = {n: res[n] - sign * d[n] for n in net.names}
                if norm(trial) < norm(res) * (1 - 1e-9):
                    res = trial
                    taken.remove(i) if i in taken else taken.append(i)
                    changed = True
            if not changed:
                break
        worst = max(float(res[n].abs().max()) / scale[n] for n in net.names)
        print(f"   {tag}: worst gradient error vs fp64 {raw:.2e} before, {worst:.2e} after fitting the branches of {len(taken)} unit(s) "
              f"{[amb[i][:2] for i in taken]}")
        for n in net.names:
            e = float(res[n].abs().max()) / scale[n]
            assert e < 1e-4, (tag, n, e)
        # every other trainable parameter (none for this model) keeps the plain rule
        _assert_gradients(model, sd, sd64, 1e-4, only=lambda pn: pn not in net.names)
        return raw

    fit_and_check("default form")
    monkeypatch.setattr(ops, "GEMM_WIDE", True)
    fit_and_check("wide-range form")


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


def _spgnn_host_masks(plan, model, g):
    """The dropout multipliers of ONE training-mode forward of GATPSPGNN, rebuilt on the host from the forward's own seed plan
    (spgnn_amd.models.GATPSPGNN._seed_plan: per level the next structure layer's feature dropout over cat[h_s, h_p], the
    structure / position layer's attention dropout, the next position layer's feature dropout) with the bit-exact host
    copies of the kernels' counter hashes (tests/util.py) - in the oracle's layout: {(kind, stream, layer): multiplier}."""
    import numpy as np

    from tests.util import keep4_scale_host, keep_scale_host
    csc = g.csc()
    N, E = g.number_of_nodes(), csc.num_edges
    eid = csc.eid.cpu().numpy()
    head = model.gat
    drop = {}

    def attn(seed, H, p):
        slot = keep_scale_host(seed, np.arange(E * H), p).reshape(E, H)          # CSC slot order -> edge-id order
        edge = np.empty_like(slot)
        edge[eid] = slot
        return torch.from_numpy(edge)
    for l, d in enumerate(plan):
        s_layer, p_layer = head.gat_layers[l], head.pgnn_layers[l]
        w_s, w_p = s_layer._num_heads * s_layer._out_feats, p_layer._num_heads * p_layer._out_feats
        if d["fp"] > 0:
            drop[("feat", "s", l + 1)] = torch.from_numpy(keep4_scale_host(d["fseed"], N, (w_s, w_p), d["fp"]))
        if d["ps"] > 0:
            drop[("attn", "s", l)] = attn(d["seed_s"], s_layer._num_heads, d["ps"])
        if d["pp"] > 0:
            drop[("attn", "p", l)] = attn(d["seed_p"], p_layer._num_heads, d["pp"])
        if d["fp2"] > 0:
            drop[("feat", "p", l + 1)] = torch.from_numpy(keep4_scale_host(d["fseed2"], N, (w_p,), d["fp2"]))
    return drop


@pytest.mark.parametrize("trees", [3, 64])
def test_training_mode_step_with_dropout_on_matches_the_oracle(trees):
    """the TIMED workload's arithmetic - st_pgat_spgnn_3 in TRAINING mode, feature and attention dropout
    on (reference models.py:431-434, 449-456: rate 0.1 on structure layers 1, 2 and position layer 1) - end to end against the
    oracle at TRAIN_BATCH_SIZE = 64 trees.  The HIP forward draws its masks from counter hashes of per-forward seeds; the test
    records the forward's seed plan, rebuilds every mask on the host bit for bit and hands them to the oracle as multipliers
    (oracle.dgl_cpu.spgnn_pel_stack drop=).  Logits and loss 1e-5, every gradient 1e-4 (rule of _assert_gradients)."""
    cfg, model = _build("st_pgat_spgnn_3", seed=12)
    model.train(True)
    g = synthetic.make_batch(trees, rank=1, device="cuda", pos_enc_dim=cfg.POS_ENC_DIM)
    w = torch.tensor(class_weight_list(cfg.CLASS_WEIGHTS))
    y = g.ndata["y"]
    mask = torch.rand(y.shape[0], generator=torch.Generator().manual_seed(6)) < torch.where(y.cpu() != 0, torch.tensor(1.0), torch.tensor(cfg.SAMPLING_RATE))
    plans = []
    orig = model.gat._seed_plan

    def recording_plan():
        plans.append(orig())
        return plans[-1]
    model.gat._seed_plan = recording_plan
    torch.manual_seed(77)
    outs = model(g)
    loss = masked_weighted_ce(outs[0], y, mask.cuda(), w.cuda())
    loss.backward()
    assert len(plans) == 1
    plan = plans[0]
    rates = [(d["fp"], d["ps"], d["pp"], d["fp2"]) for d in plan]
    assert rates == [(0.1, 0.0, 0.0, 0.1), (0.1, 0.1, 0.1, 0.0), (0.0, 0.1, 0.0, 0.0)], rates     # the reference's placement
    drop = _spgnn_host_masks(plan, model, g)
    assert sorted(drop) == [("attn", "p", 1), ("attn", "s", 1), ("attn", "s", 2), ("feat", "p", 1), ("feat", "s", 1), ("feat", "s", 2)]
    for k, m in drop.items():
        kept = float((m > 0).float().mean())
        assert abs(kept - 0.9) < (0.05 if trees < 10 else 0.01), (k, kept)
    src, dst = g.cpu().edges()
    res = {}
    for dtype in (torch.float32, torch.float64):
        sd = {k: v.detach().cpu().to(dtype if v.dtype.is_floating_point else v.dtype).requires_grad_(v.dtype.is_floating_point)
              for k, v in model.state_dict().items()}
        refs = O.net_forward(cfg.KIND, sd, src, dst, g.number_of_nodes(), g.ndata["fvs"].cpu().to(dtype), g.ndata["pos_enc"].cpu().to(dtype),
                             drop={k: m.to(dtype) for k, m in drop.items()})
        ref_loss = O.masked_weighted_ce(refs[0], y.cpu(), mask, w.to(dtype))
        ref_loss.backward()
        res[dtype] = (refs, sd, ref_loss)
    refs, sd, ref_loss = res[torch.float32]
    refs64, sd64, _ = res[torch.float64]
    # dropout really is on: the eval-mode oracle is far away
    ev = O.net_forward(cfg.KIND, {k: v.detach() for k, v in sd.items()}, src, dst, g.number_of_nodes(), g.ndata["fvs"].cpu(), g.ndata["pos_enc"].cpu())
    assert rel_err(outs[0], ev[0]) > 1e-2
    for o, r, r64 in zip(outs, refs, refs64):
        e_n, e_m = rel_err(o, r), mixed_err(o, r)
        print(f"st_pgat_spgnn_3 TRAIN mode {trees} trees {tuple(o.shape)}: normwise {e_n:.2e}, elementwise mixed {e_m:.2e}; vs fp64 {rel_err(o, r64):.2e}")
        assert e_n < TOL and e_m < TOL and rel_err(o, r64) < TOL
    assert rel_err(loss, ref_loss) < TOL
    worst = _assert_gradients(model, sd, sd64, 1e-4)
    print(f"st_pgat_spgnn_3 TRAIN mode {trees} trees: loss {rel_err(loss, ref_loss):.2e}, worst non-tiny gradient normwise error vs the fp32 oracle {worst:.2e}")
