"""Committed golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py from the oracle):
the oracle must keep reproducing them on CPU; the HIP path must match them on the GPU."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import dgl_cpu as O

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
GAT_FILES = ["gat_layer_h2_d8_res_elu", "gat_layer_h1_d64_res_tanh", "gat_layer_h2_d64_nores_none"]
ACT = {"elu": F.elu, "tanh": torch.tanh, "none": None}


def _load(name):
    z = np.load(os.path.join(HERE, name + ".npz"))
    return {k: z[k] for k in z.files}


def _t(a, grad=False):
    return torch.from_numpy(np.asarray(a)).requires_grad_(grad)


@pytest.mark.parametrize("name", GAT_FILES)
def test_oracle_reproduces_golden_gat(name):
    z = _load(name)
    src, dst, n = _t(z["src"]), _t(z["dst"]), int(z["num_nodes"])
    leaves = {k: _t(z[k], True) for k in ("x", "fc_weight", "attn_l", "attn_r", "bias")}
    wr = _t(z["res_fc_weight"], True) if "res_fc_weight" in z else None
    rst, a = O.gat_conv(src, dst, n, leaves["x"], leaves["fc_weight"], leaves["attn_l"], leaves["attn_r"], wr,
                        leaves["bias"], 0.2, ACT[str(z["act"])])
    assert np.allclose(rst.detach().numpy(), z["rst"], rtol=1e-5, atol=1e-6)
    assert np.allclose(a.detach().numpy(), z["attn"], rtol=1e-5, atol=1e-7)
    (rst * _t(z["cot"])).sum().backward()
    for k, t in leaves.items():
        assert np.allclose(t.grad.numpy(), z["grad_" + k], rtol=1e-4, atol=1e-5), k


def test_oracle_reproduces_golden_spmm_and_config1():
    z = _load("spmm_f64")
    src, dst, n, x = _t(z["src"]), _t(z["dst"]), int(z["num_nodes"]), _t(z["x"])
    assert np.allclose(O.graph_conv(src, dst, n, x, _t(z["weight"]), torch.zeros(6), F.elu).numpy(), z["gcn"], rtol=1e-5, atol=1e-6)
    assert np.allclose(O.gin_conv(src, dst, n, x, torch.tensor([0.3]), None, "mean").numpy(), z["gin_eps03"], rtol=1e-5, atol=1e-6)
    assert np.array_equal(O.spmm_max(src, dst, x, n).numpy(), z["max"])
    # BASELINE configs[0]: st_gcn_3 forward on one 128-node tree, CPU plumbing (no GPU involved)
    from spgnn_amd import models, synthetic
    from spgnn_amd.configs import get_config
    from spgnn_amd.graph import edges_from_adj
    z = _load("config1_st_gcn_3_n128")
    torch.manual_seed(0)
    m = models.build_model(get_config("st_gcn_3").MODEL); m.init(None)
    sd = {k: t.detach() for k, t in m.state_dict().items()}
    assert abs(float(sd["gcn.gcn_layers.0.weight"].double().sum()) - float(z["weight0_checksum"])) < 1e-6
    s = synthetic.synthetic_trees(1, rank=0, fixed_n=128)[0]
    u, v = edges_from_adj(s["adj"])
    assert np.array_equal(u, z["src"]) and np.array_equal(v, z["dst"]) and u.shape[0] == 382
    out, emb = O.net_forward("gcn", sd, _t(u), _t(v), 128, _t(s["fvs"]))
    assert np.allclose(out.numpy(), z["logits"], rtol=1e-4, atol=1e-5)
    assert np.allclose(emb.sum(1).numpy(), z["embed_rowsum"], rtol=1e-4, atol=1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("name", GAT_FILES)
def test_hip_matches_golden_gat(name):
    from spgnn_amd import nn as snn
    from spgnn_amd.graph import TreeGraph
    z = _load(name)
    H, D, fin = int(z["H"]), int(z["D"]), z["x"].shape[1]
    g = TreeGraph((z["src"], z["dst"]), int(z["num_nodes"])).to("cuda")
    layer = snn.GATConv(fin, D, H, 0.0, 0.0, 0.2, "res_fc_weight" in z, ACT[str(z["act"])]).cuda()
    sd = {"fc.weight": z["fc_weight"], "attn_l": z["attn_l"], "attn_r": z["attn_r"], "bias": z["bias"]}
    if "res_fc_weight" in z:
        sd["res_fc.weight"] = z["res_fc_weight"]
    layer.load_state_dict({k: _t(v) for k, v in sd.items()})
    x = _t(z["x"]).cuda().requires_grad_(True)
    rst, attn = layer(g, x, get_attention=True)
    scale = np.abs(z["rst"]).max()
    assert np.abs(rst.detach().cpu().numpy() - z["rst"]).max() / scale < 1e-5
    assert np.abs(attn.squeeze(-1).cpu().numpy() - z["attn"]).max() < 1e-6
    (rst * _t(z["cot"]).cuda()).sum().backward()
    got = {"x": x.grad, "fc_weight": layer.fc.weight.grad, "attn_l": layer.attn_l.grad, "attn_r": layer.attn_r.grad,
           "bias": layer.bias.grad}
    if "res_fc_weight" in z:
        got["res_fc_weight"] = layer.res_fc.weight.grad
    for k, t in got.items():
        ref = z["grad_" + k]
        assert np.abs(t.cpu().numpy() - ref).max() / np.abs(ref).max() < 5e-5, k


@pytest.mark.gpu
def test_hip_matches_golden_spmm_and_config1():
    from spgnn_amd import models, nn as snn, ops, synthetic
    from spgnn_amd.configs import get_config
    from spgnn_amd.graph import TreeGraph
    z = _load("spmm_f64")
    g = TreeGraph((z["src"], z["dst"]), int(z["num_nodes"])).to("cuda")
    x = _t(z["x"]).cuda()
    gc = snn.GraphConv(64, 6, activation=F.elu).cuda()
    gc.load_state_dict({"weight": _t(z["weight"]), "bias": torch.zeros(6)})
    assert np.allclose(gc(g, x).detach().cpu().numpy(), z["gcn"], rtol=1e-5, atol=2e-6)
    gin = snn.GINConv(None, "mean", init_eps=0.3).cuda()
    assert np.allclose(gin(g, x).cpu().numpy(), z["gin_eps03"], rtol=1e-5, atol=2e-6)
    assert np.array_equal(ops.spmm_max(g.csc(), x).cpu().numpy(), z["max"])
    z = _load("config1_st_gcn_3_n128")
    torch.manual_seed(0)
    m = models.build_model(get_config("st_gcn_3").MODEL); m.init(None)
    m = m.cuda().eval()
    gb = synthetic.batch_from_samples(synthetic.synthetic_trees(1, rank=0, fixed_n=128), "cuda", None)
    with torch.no_grad():
        out, emb = m(gb)
    assert np.abs(out.cpu().numpy() - z["logits"]).max() / np.abs(z["logits"]).max() < 1e-5


BF16_FILES = ["bf16_storage_hidden_h2_d64_elu", "bf16_storage_output_h2_d128_linear_mean"]


def _bf16_model(z):
    src, dst, n = _t(z["src"]), _t(z["dst"]), int(z["num_nodes"])
    leaves = {k: _t(z[k], True) for k in ("x", "fc_weight", "attn_l", "attn_r", "res_fc_weight", "bias")}
    if int(z["mean"]):
        rst = O.gat_conv_linear_mean(src, dst, n, leaves["x"], leaves["fc_weight"], leaves["attn_l"], leaves["attn_r"],
                                     leaves["res_fc_weight"], leaves["bias"], 0.2, storage=O.Bf16Storage)[0]
    else:
        rst = O.gat_conv(src, dst, n, leaves["x"], leaves["fc_weight"], leaves["attn_l"], leaves["attn_r"], leaves["res_fc_weight"],
                         leaves["bias"], 0.2, ACT[str(z["act"])], storage=O.Bf16Storage)[0].flatten(1)
    return rst, leaves


@pytest.mark.parametrize("name", BF16_FILES)
def test_oracle_storage_model_reproduces_golden(name):
    """BASELINE config 4 (bf16 storage): the oracle's storage model keeps reproducing its committed vectors."""
    z = _load(name)
    rst, leaves = _bf16_model(z)
    assert np.allclose(rst.detach().numpy(), z["rst"], rtol=1e-12, atol=1e-12)
    (rst * _t(z["cot"])).sum().backward()
    for k, t in leaves.items():
        got = O._rb(t.grad) if k == "x" else t.grad
        assert np.allclose(got.numpy(), z["grad_" + k], rtol=1e-10, atol=1e-12), k


@pytest.mark.gpu
@pytest.mark.parametrize("name", BF16_FILES)
def test_hip_bf16_matches_golden_storage_model(name):
    """The HIP bf16-storage layer against the committed storage-model vectors: within two bf16 ulps (2^-7) normwise - the
    path may differ from the model only where fp32 accumulation order moves a value across a rounding boundary."""
    from spgnn_amd import nn as snn, ops_bf16
    from spgnn_amd.graph import TreeGraph
    z = _load(name)
    H, D, fin, mean = int(z["H"]), int(z["D"]), z["x"].shape[1], bool(int(z["mean"]))
    g = TreeGraph((z["src"], z["dst"]), int(z["num_nodes"])).to("cuda")
    layer = snn.GATConv(fin, D, H, 0.0, 0.0, 0.2, True, ACT[str(z["act"])]).cuda()
    layer.load_state_dict({"fc.weight": _t(z["fc_weight"]).float(), "attn_l": _t(z["attn_l"]).float(), "attn_r": _t(z["attn_r"]).float(),
                           "bias": _t(z["bias"]).float(), "res_fc.weight": _t(z["res_fc_weight"]).float()})
    x = ops_bf16.cast_rows(_t(z["x"]).float().cuda()).requires_grad_(True)
    out = layer(g, x, mean_heads=mean)
    out = out if mean else out.flatten(1)
    assert out.dtype == (torch.float32 if mean else torch.bfloat16)
    ulp2 = 2.0 ** -7
    assert np.abs(out.detach().float().cpu().numpy() - z["rst"]).max() <= ulp2 * np.abs(z["rst"]).max()
    out.backward(_t(z["cot"]).to(out.dtype).cuda())
    got = {"x": x.grad.float(), "fc_weight": layer.fc.weight.grad, "attn_l": layer.attn_l.grad, "attn_r": layer.attn_r.grad,
           "res_fc_weight": layer.res_fc.weight.grad, "bias": layer.bias.grad}
    for k, t in got.items():
        ref = z["grad_" + k]
        assert np.abs(t.cpu().numpy().reshape(ref.shape) - ref).max() <= ulp2 * np.abs(ref).max(), k
