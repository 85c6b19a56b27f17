"""Drop-in proof (SURVEY.md §8b level 1): model code written the way the reference's models.py writes
it - DGL layer constructors called positionally, ``torch.cat``, ``.flatten(1)``, ``.mean(1)``, a separate ``nn.Linear``
classifier, none of this package's extension keywords - runs on ``spgnn_amd.nn`` and equals both ``spgnn_amd.models``
(the fused path the package ships) and the oracle.  Plus the harness items a caller depends on: ExponentialLR under
HIP-graph replay (``TrainStep.set_lr``), a DGL-0.6-layout checkpoint through the reference's key + size filter, and the
2-rank x B trees == 1-rank x 2B trees equivalence with the real flagship model."""
import copy
import os
import socket

import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from oracle import dgl_cpu as O
from spgnn_amd import checkpoint, models, synthetic
from spgnn_amd.configs import class_weight_list, get_config
from spgnn_amd.nn import GATConv                      # the import swap: `from dgl.nn.pytorch import GATConv`
from spgnn_amd.train import TrainStep, masked_weighted_ce
from tests.util import rel_err

pytestmark = pytest.mark.gpu


class CallerGAT(nn.Module):
    """A caller's GAT head in the reference's style (cf. reference models.py:283-329): positional GATConv arguments,
    hidden layers flattened, output layer averaged over heads."""

    def __init__(self, num_layers, in_dim, num_hiddens, out_ch, heads, activation, feat_drop, attn_drop, negative_slope, residual):
        super().__init__()
        self.num_layers = num_layers
        self.gat_layers = nn.ModuleList()
        self.gat_layers.append(GATConv(in_dim, num_hiddens[0], heads[0], 0.0, 0.0, negative_slope, residual, activation))
        for l in range(1, num_layers):
            self.gat_layers.append(GATConv(num_hiddens[l - 1] * heads[l - 1], num_hiddens[l], heads[l], feat_drop, attn_drop,
                                           negative_slope, residual, activation))
        self.gat_layers.append(GATConv(num_hiddens[-1] * heads[num_layers - 1], out_ch, heads[num_layers], 0.0, 0.0,
                                       negative_slope, residual, None))

    def forward(self, g):
        h = g.ndata["fvs"]
        for l in range(self.num_layers):
            h = self.gat_layers[l](g, h).flatten(1)
        return self.gat_layers[-1](g, h).mean(1)


class CallerSPGNN(nn.Module):
    """A caller's position-aware head in the reference's style (cf. reference models.py:403-484)."""

    def __init__(self, num_layers, in_dim, pos_in_dim, num_hiddens, pos_hiddens, pos_heads, out_ch, heads, activation,
                 feat_drop, attn_drop, negative_slope, residual, p_activation=torch.tanh):
        super().__init__()
        self.num_layers = num_layers
        self.gat_layers, self.pgnn_layers = nn.ModuleList(), nn.ModuleList()
        width = in_dim + pos_in_dim
        for l in range(num_layers):
            drop = (0.0, 0.0) if l == 0 else (feat_drop, attn_drop)
            self.gat_layers.append(GATConv(width, num_hiddens[l], heads[l], drop[0], drop[1], negative_slope, residual, activation))
            width = num_hiddens[l] * heads[l] + pos_hiddens[l] * pos_heads[l]
        self.gat_layers.append(GATConv(width, out_ch, heads[num_layers], 0.0, 0.0, negative_slope, residual, activation))
        p_width = pos_in_dim
        for l in range(num_layers):
            drop = (feat_drop, attn_drop) if 0 < l < num_layers - 1 else (0.0, 0.0)
            self.pgnn_layers.append(GATConv(p_width, pos_hiddens[l], pos_heads[l], drop[0], drop[1], negative_slope, True, p_activation))
            p_width = pos_hiddens[l] * pos_heads[l]

    def forward(self, g):
        h_p, h_s = g.ndata["pos_enc"], g.ndata["fvs"]
        for l in range(self.num_layers):
            h_s = torch.cat([h_s, h_p], dim=1)
            h_s = self.gat_layers[l](g, h_s).flatten(1)
            h_p = self.pgnn_layers[l](g, h_p).flatten(1)
        h_s = torch.cat([h_s, h_p], dim=1)
        return self.gat_layers[-1](g, h_s).mean(1), h_p


def _packaged(name):
    cfg = get_config(name)
    torch.manual_seed(0)
    model = models.build_model(cfg.MODEL).cuda()
    model.init(None)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith("bias"):
                p.normal_(0, 0.05)
    model.set_gcn_only()
    model.eval()
    return cfg, model


@pytest.mark.parametrize("name", ["st_gat_3", "st_pgat_spgnn_3"])
def test_reference_style_stack_equals_packaged_model_and_oracle(name):
    cfg, model = _packaged(name)
    M = cfg.MODEL
    L = M["num_gat_layers"]
    heads = [M["num_heads"]] * L + [M["num_out_heads"]]
    if name == "st_gat_3":
        head = CallerGAT(L, M["fv_dim"], M["num_hiddens"], M["node_embed_dim"], heads, F.elu, M["feat_drop"], M["attn_drop"],
                         M["negative_slope"], True)
    else:
        head = CallerSPGNN(L, M["fv_dim"], M["pos_enc_dim"], M["num_hiddens"], M["pos_hiddens"], [M["num_pos_heads"]] * (L + 1),
                           M["node_embed_dim"], heads, F.elu, M["feat_drop"], M["attn_drop"], M["negative_slope"], True)
    head = head.cuda().eval()
    classifier = nn.Linear(M["node_embed_dim"], M["out_ch"]).cuda()           # the reference's gnn_out (models.py:1125)
    # same parameters: the caller-style head has exactly the packaged head's state_dict keys and shapes
    assert set(head.state_dict()) == set(model.gat.state_dict())
    head.load_state_dict(model.gat.state_dict())
    classifier.load_state_dict(model.gnn_out.state_dict())
    g = synthetic.make_batch(3, rank=2, device="cuda", pos_enc_dim=cfg.POS_ENC_DIM)
    w = torch.tensor(class_weight_list(cfg.CLASS_WEIGHTS))
    y = g.ndata["y"]
    mask = torch.rand(y.shape[0], generator=torch.Generator().manual_seed(5)) < 0.5

    emb = head(g)
    emb = emb[0] if isinstance(emb, tuple) else emb
    logits = classifier(emb)
    masked_weighted_ce(logits, y, mask.cuda(), w.cuda()).backward()
    outs = model(g)
    masked_weighted_ce(outs[0], y, mask.cuda(), w.cuda()).backward()
    # (1) == the packaged (fused) model: same function, different kernels and summation order (the packaged output layer
    # aggregates before it projects; here it projects first): fp32 noise between two evaluations, a few 1e-6 - both are
    # held to 1e-5 against the oracle below
    assert rel_err(logits, outs[0]) < 4e-6 and rel_err(emb, outs[1]) < 4e-6
    ref_grads = dict(model.gat.named_parameters())
    gmax = max(float(p.grad.abs().max()) for p in head.parameters())
    for n, p in head.named_parameters():
        # (the position stream's score-vector gradients are ~1e-11 here, total cancellations far below the fp32 resolution
        # of the sums they come from: see tests/test_hip_models.py)
        tiny = float((p.grad - ref_grads[n].grad).abs().max()) < 1e-7 * gmax
        assert rel_err(p.grad, ref_grads[n].grad) < 2e-5 or tiny, n
    assert rel_err(classifier.weight.grad, model.gnn_out.weight.grad) < 2e-5
    # (2) == the oracle (the DGL-CPU-equivalent restatement), BASELINE's 1e-5 on the forward pass
    src, dst = g.cpu().edges()
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    pe = g.ndata["pos_enc"].cpu() if "pos_enc" in g.ndata else None
    ref = O.net_forward(cfg.KIND, sd, src, dst, g.number_of_nodes(), g.ndata["fvs"].cpu(), pe)
    assert rel_err(logits, ref[0]) < 1e-5 and rel_err(emb, ref[1]) < 1e-5


def test_exponential_lr_under_graph_replay_equals_torch_sgd():
    """reference job_runner.py:254-259, 1350-1366: SGD(momentum) + ExponentialLR(gamma) stepped once per epoch.  The
    captured step reads its learning rate from a device scalar (TrainStep.set_lr), so the schedule works under replay."""
    cfg, model = _packaged("st_gat_3")
    ref_model = copy.deepcopy(model)
    g = synthetic.make_batch(3, rank=4, device="cuda", pos_enc_dim=cfg.POS_ENC_DIM)
    w = class_weight_list(cfg.CLASS_WEIGHTS)
    lr0, gamma, steps_per_epoch, epochs = 2e-3, 0.9, 3, 3
    # reference flow: torch.optim.SGD + ExponentialLR on the same (eval-mode, all-nodes-masked-in) loss
    opt = torch.optim.SGD([p for p in ref_model.parameters() if p.requires_grad], lr=lr0, momentum=0.9)
    sched = torch.optim.lr_scheduler.ExponentialLR(opt, gamma=gamma)
    y = g.ndata["y"]
    all_in = torch.ones_like(y, dtype=torch.bool)
    ts = TrainStep(model, w, 1.0, lr0, 0.9)                # sampling rate 1: every node in the mask, no RNG dependence
    ts.capture(g, warmup=1)                                # one eager step ...
    opt.zero_grad(); masked_weighted_ce(ref_model(g)[0], y, all_in, torch.tensor(w).cuda()).backward(); opt.step()
    done = 1
    for e in range(epochs):
        while done < (e + 1) * steps_per_epoch:            # ... then replays
            ts.replay()
            opt.zero_grad(); masked_weighted_ce(ref_model(g)[0], y, all_in, torch.tensor(w).cuda()).backward(); opt.step()
            done += 1
        sched.step()
        ts.set_lr(sched.get_last_lr()[0])
    got = dict(model.named_parameters())
    for n, p in ref_model.named_parameters():
        if p.requires_grad:
            assert torch.isfinite(p).all() and rel_err(got[n], p) < 1e-5, n
    assert abs(ts.lr - lr0 * gamma ** epochs) < 1e-12


def test_dgl06_layout_checkpoint_loads_through_reference_filter(tmp_path):
    """A GNN-stage checkpoint in the reference's file layout written under DGL 0.6 (GATConv has no ``bias``), reloaded by
    the reference's rule (job_runner.py:85-123): unknown keys and size mismatches are skipped, the rest is loaded."""
    cfg, model = _packaged("st_pgat_spgnn_3")
    donor = copy.deepcopy(model)
    with torch.no_grad():
        for p in donor.parameters():
            p.add_(0.25)
    sd = {k: v.clone() for k, v in donor.state_dict().items()}
    dgl06 = {k: v for k, v in sd.items() if not (k.startswith("gat.") and k.endswith(".bias"))}     # no GATConv bias in 0.6
    dgl06["gat.gat_layers.0.fc.weight"] = torch.zeros(7, 7)                 # a size mismatch: skipped, not an error
    dgl06["ds_modules.0.conv.weight"] = torch.zeros(3)                      # CNN-trunk key unknown to this build: skipped
    path = os.path.join(tmp_path, "100.pth")
    checkpoint.save_states(path, {"iteration": 100, "epoch_n": 7, "model_dict": dgl06})
    before = {k: v.clone() for k, v in model.state_dict().items()}
    states = checkpoint.load_pretrained_model(path, [model], ["model_dict"], device="cuda")
    assert states["iteration"] == 100 and states["epoch_n"] == 7
    after = model.state_dict()
    for k in after:
        if k.startswith("gat.") and k.endswith(".bias"):
            assert torch.equal(after[k], before[k]), k                     # absent from the checkpoint: untouched
        elif k == "gat.gat_layers.0.fc.weight":
            assert torch.equal(after[k], before[k])                        # size mismatch: untouched
        else:
            assert torch.equal(after[k].cpu(), sd[k].cpu()), k
    # and the model still runs on the loaded weights
    g = synthetic.make_batch(2, rank=1, device="cuda", pos_enc_dim=cfg.POS_ENC_DIM)
    assert torch.isfinite(model(g)[0]).all()


# ---- 2 ranks x B trees == 1 rank x 2B trees with the real flagship model -------------------------------
def _dp_real_worker(rank, world, port, ret):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)     # two ranks on ONE GPU: RCCL refuses that, gloo moves the CUDA tensors
    cfg = get_config("st_pgat_spgnn_3")
    torch.manual_seed(0)
    model = models.build_model(cfg.MODEL).cuda()
    model.init(None); model.set_gcn_only(); model.eval()
    samples = synthetic.synthetic_trees(6, rank=0)
    from spgnn_amd.train import balanced_tree_partition
    parts = balanced_tree_partition([s["fvs"].shape[0] for s in samples], world)
    g = synthetic.batch_from_samples([samples[i] for i in parts[rank]], "cuda", cfg.POS_ENC_DIM)
    ts = TrainStep(model, class_weight_list(cfg.CLASS_WEIGHTS), 1.0, 0.05, 0.9, seed=5)
    losses = [float(ts.step(g)) for _ in range(3)]
    ret[rank] = (losses, ts.bucket.flat_param[:ts.bucket.numel].detach().cpu().clone(), parts)
    dist.destroy_process_group()


def test_two_ranks_equal_one_rank_on_the_union_with_the_real_model():
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mgr = mp.Manager(); ret = mgr.dict()
    mp.spawn(_dp_real_worker, args=(2, port, ret), nprocs=2, join=True)
    cfg = get_config("st_pgat_spgnn_3")
    torch.manual_seed(0)
    model = models.build_model(cfg.MODEL).cuda()
    model.init(None); model.set_gcn_only(); model.eval()
    g = synthetic.batch_from_samples(synthetic.synthetic_trees(6, rank=0), "cuda", cfg.POS_ENC_DIM)
    ts = TrainStep(model, class_weight_list(cfg.CLASS_WEIGHTS), 1.0, 0.05, 0.9, seed=5)
    losses = [float(ts.step(g)) for _ in range(3)]
    flat = ts.bucket.flat_param[:ts.bucket.numel].detach().cpu()
    for r in (0, 1):
        assert np.allclose(ret[r][0], losses, rtol=1e-5), (ret[r][0], losses)       # global class-weighted mean, not a mean of means
        assert rel_err(ret[r][1], flat) < 1e-6
    assert torch.equal(ret[0][1], ret[1][1])                                        # replicas stay identical
    assert sorted(ret[0][2][0] + ret[0][2][1]) == list(range(6))
