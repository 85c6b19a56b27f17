"""Generates the committed golden vectors under tests/golden/ from the CPU oracle
(oracle/dgl_cpu.py).  The reference itself cannot run here (DGL absent: SURVEY.md §8c), so these
pin the ORACLE's outputs (and through the parity tests, the HIP path's) against silent drift; they
are not reference outputs ("parity unpinned").

    python tests/golden/make_golden.py        # rewrites the .npz files (deterministic)
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import dgl_cpu as O  # noqa: E402
from spgnn_amd import synthetic  # noqa: E402
from spgnn_amd.graph import edges_from_adj  # noqa: E402


def gat_layer_case(name, ns, fin, H, D, res, act, seed):
    rng = np.random.default_rng(seed)
    gen = torch.Generator().manual_seed(seed)
    srcs, dsts, off = [], [], 0
    for n in ns:
        u, v = edges_from_adj(synthetic.random_tree_adj(n, rng))
        srcs.append(u + off); dsts.append(v + off); off += n
    src, dst = torch.from_numpy(np.concatenate(srcs)), torch.from_numpy(np.concatenate(dsts))
    mk = lambda *s: (torch.randn(*s, generator=gen) * 0.5).requires_grad_(True)
    x, w, al, ar, b = mk(off, fin), mk(H * D, fin), mk(1, H, D), mk(1, H, D), mk(H * D)
    wr = mk(H * D, fin) if res else None
    actf = {"elu": F.elu, "tanh": torch.tanh, "none": None}[act]
    rst, a = O.gat_conv(src, dst, off, x, w, al, ar, wr, b, 0.2, actf)
    cot = torch.randn(rst.shape, generator=gen)
    leaves = [t for t in (x, w, al, ar, wr, b) if t is not None]
    grads = torch.autograd.grad((rst * cot).sum(), leaves)
    out = dict(src=src.numpy(), dst=dst.numpy(), num_nodes=off, H=H, D=D, act=act, x=x.detach().numpy(),
               fc_weight=w.detach().numpy(), attn_l=al.detach().numpy(), attn_r=ar.detach().numpy(),
               bias=b.detach().numpy(), rst=rst.detach().numpy(), attn=a.detach().numpy(), cot=cot.numpy())
    if res:
        out["res_fc_weight"] = wr.detach().numpy()
    for nm, gr in zip([n for n, t in zip(["x", "fc_weight", "attn_l", "attn_r", "res_fc_weight", "bias"],
                                         (x, w, al, ar, wr, b)) if t is not None], grads):
        out["grad_" + nm] = gr.numpy()
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)


def spmm_case(name, ns, F_, seed):
    rng = np.random.default_rng(seed)
    gen = torch.Generator().manual_seed(seed)
    srcs, dsts, off = [], [], 0
    for n in ns:
        u, v = edges_from_adj(synthetic.random_tree_adj(n, rng))
        srcs.append(u + off); dsts.append(v + off); off += n
    src, dst = torch.from_numpy(np.concatenate(srcs)), torch.from_numpy(np.concatenate(dsts))
    x = torch.randn(off, F_, generator=gen)
    w = torch.randn(F_, 6, generator=gen)
    gcn = O.graph_conv(src, dst, off, x, w, torch.zeros(6), F.elu)
    gin = O.gin_conv(src, dst, off, x, torch.tensor([0.3]), None, "mean")
    mx = O.spmm_max(src, dst, x, off)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), src=src.numpy(), dst=dst.numpy(), num_nodes=off, x=x.numpy(),
                        weight=w.numpy(), gcn=gcn.numpy(), gin_eps03=gin.numpy(), max=mx.numpy())


def bf16_storage_case(name, ns, fin, H, D, act, mean, seed):
    """The build's bf16-STORAGE path (BASELINE config 4) has no reference result; what is pinned is the oracle's storage
    model (oracle.dgl_cpu.Bf16Storage, fp64 arithmetic with bf16 rounding at the product's storage points): a hidden layer
    (project-first form) or, with ``mean`` and no activation, the output layer in its linear-mean form."""
    rng = np.random.default_rng(seed)
    gen = torch.Generator().manual_seed(seed)
    srcs, dsts, off = [], [], 0
    for n in ns:
        u, v = edges_from_adj(synthetic.random_tree_adj(n, rng))
        srcs.append(u + off); dsts.append(v + off); off += n
    src, dst = torch.from_numpy(np.concatenate(srcs)), torch.from_numpy(np.concatenate(dsts))
    mk = lambda *s_: (torch.randn(*s_, generator=gen, dtype=torch.float64) * 0.5)
    x = O._rb(mk(off, fin)).requires_grad_(True)                    # the input rows are bf16 values
    w, al, ar, b, wr = (mk(H * D, fin) * 0.3).requires_grad_(True), mk(1, H, D).requires_grad_(True), mk(1, H, D).requires_grad_(True), \
        (mk(H * D) * 0.1).requires_grad_(True), (mk(H * D, fin) * 0.3).requires_grad_(True)
    actf = {"elu": F.elu, "none": None}[act]
    if mean:
        assert act == "none" and O.linear_mean_form(H, D, fin, True)
        rst = O.gat_conv_linear_mean(src, dst, off, x, w, al, ar, wr, b, 0.2, storage=O.Bf16Storage)[0]
    else:
        rst = O.gat_conv(src, dst, off, x, w, al, ar, wr, b, 0.2, actf, storage=O.Bf16Storage)[0].flatten(1)
    cot = torch.randn(rst.shape, generator=gen, dtype=torch.float64)
    if not mean:
        cot = O._rb(cot)                                            # the gradient of stored rows arrives as bf16
    leaves = dict(x=x, fc_weight=w, attn_l=al, attn_r=ar, res_fc_weight=wr, bias=b)
    grads = torch.autograd.grad((rst * cot).sum(), list(leaves.values()))
    out = dict(src=src.numpy(), dst=dst.numpy(), num_nodes=off, H=H, D=D, act=act, mean=int(mean), rst=rst.detach().numpy(),
               cot=cot.numpy(), **{k: v.detach().numpy() for k, v in leaves.items()})
    for k, gr in zip(leaves, grads):
        out["grad_" + k] = (O._rb(gr) if k == "x" else gr).numpy()   # g_x is stored as bf16 rows
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)


def config1_case():
    """BASELINE.json configs[0]: st_gcn_3 forward on one 128-node synthetic tree.  Inputs and weights are
    regenerated from seeds (torch CPU generator); the expected logits are stored."""
    from spgnn_amd import models
    from spgnn_amd.configs import get_config
    cfg = get_config("st_gcn_3")
    torch.manual_seed(0)
    m = models.build_model(cfg.MODEL); m.init(None)
    s = synthetic.synthetic_trees(1, rank=0, fixed_n=128)[0]
    u, v = edges_from_adj(s["adj"])
    sd = {k: t.detach() for k, t in m.state_dict().items()}
    out, emb = O.net_forward("gcn", sd, torch.from_numpy(u), torch.from_numpy(v), 128, torch.from_numpy(s["fvs"]))
    np.savez_compressed(os.path.join(HERE, "config1_st_gcn_3_n128.npz"), logits=out.detach().numpy(),
                        embed_rowsum=emb.detach().sum(1).numpy(), src=u, dst=v,
                        weight0_checksum=float(sd["gcn.gcn_layers.0.weight"].double().sum()))


if __name__ == "__main__":
    gat_layer_case("gat_layer_h2_d8_res_elu", [6, 9, 1], 7, 2, 8, True, "elu", 1)
    gat_layer_case("gat_layer_h1_d64_res_tanh", [12, 30], 39, 1, 64, True, "tanh", 2)
    gat_layer_case("gat_layer_h2_d64_nores_none", [25], 16, 2, 64, False, "none", 3)
    spmm_case("spmm_f64", [10, 21, 3], 64, 4)
    bf16_storage_case("bf16_storage_hidden_h2_d64_elu", [14, 23], 32, 2, 64, "elu", False, 5)
    bf16_storage_case("bf16_storage_output_h2_d128_linear_mean", [9, 30], 32, 2, 128, "none", True, 6)
    config1_case()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))
