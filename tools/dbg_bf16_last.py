"""Debug aid: isolate the bf16 output layer of st_gat_3 - feed the storage model the HIP path's own input rows and
incoming gradient, compare the layer's parameter gradients."""
import sys
import torch
sys.path.insert(0, ".")
from tests import test_hip_bf16 as T
from tests.util import rel_err
from oracle import dgl_cpu as O
from spgnn_amd import synthetic, nn as snn
from spgnn_amd.configs import class_weight_list
from spgnn_amd.train import masked_weighted_ce

name = sys.argv[1] if len(sys.argv) > 1 else "st_gat_3"
cfg, model = T._build(name)
g = synthetic.make_batch(3, rank=5, device="cuda", pos_enc_dim=cfg.POS_ENC_DIM)
model.eval()
layer = model.gat.gat_layers[-1]
cap = {}
orig = layer._forward
def wrapped(graph, feat, *a, **k):
    cap["x"] = feat.detach().float().cpu()
    out = orig(graph, feat, *a, **k)
    t = out[0] if isinstance(out, tuple) else out
    t.register_hook(lambda gr: cap.__setitem__("g", gr.detach().float().cpu()))
    cap["out"] = t.detach().float().cpu()
    return out
layer._forward = wrapped
w = torch.tensor(class_weight_list(cfg.CLASS_WEIGHTS))
y = g.ndata["y"]
mask = torch.rand(y.shape[0], generator=torch.Generator().manual_seed(5)) < 0.5
logits, emb = model(g)
masked_weighted_ce(logits, y, mask.cuda(), w.cuda()).backward()
src, dst = g.cpu().edges()
n = g.number_of_nodes()
sd = {k: v.detach().cpu().double().requires_grad_() for k, v in layer.state_dict().items()}
x64 = cap["x"].double().requires_grad_()
for label, storage in (("model", O.Bf16Storage), ("true", None)):
    for p in sd.values():
        p.grad = None
    x64.grad = None
    r = O.gat_conv_linear_mean(src, dst, n, x64, sd["fc.weight"], sd["attn_l"], sd["attn_r"], sd.get("res_fc.weight"), sd["bias"], 0.2,
                               storage=storage)[0]
    r.backward(cap["g"].double())
    print(label, "out", rel_err(cap["out"], r), "g_x skipped")
    for k, p in layer.named_parameters():
        print(f"   {k:16s} hip-vs-{label} {rel_err(p.grad, sd[k].grad):.5f}  max|g| {float(sd[k].grad.abs().max()):.3e}")

# ---- the full storage-model stack: capture ITS input rows / incoming gradient of the output layer
mcap = {}
orig_lm = O.gat_conv_linear_mean
def lm(src_, dst_, n_, feat, *a, **k):
    mcap["x"] = feat.detach()
    out = orig_lm(src_, dst_, n_, feat, *a, **k)
    out[0].register_hook(lambda gr: mcap.__setitem__("g", gr.detach()))
    return out
O.gat_conv_linear_mean = lm
(m_logits, m_emb), sd_m = T._oracle_logits(cfg, model, g, torch.float64, O.Bf16Storage, grad=True)
O.masked_weighted_ce(m_logits, y.cpu(), mask, w.double()).backward()
O.gat_conv_linear_mean = orig_lm
xm, gm = mcap["x"], mcap["g"]
xh, gh = cap["x"].double(), cap["g"].double()
print("input rows hip vs model: rel", rel_err(xh, xm), "fraction differing", float((xh != xm).double().mean()),
      "max |d| / max|x|", float((xh - xm).abs().max() / xm.abs().max()))
print("incoming grad hip vs model: rel", rel_err(gh, gm))
def grads(x, gout):
    sd2 = {k: v.detach().cpu().double().requires_grad_() for k, v in layer.state_dict().items()}
    r = orig_lm(src, dst, n, x.clone().requires_grad_(), sd2["fc.weight"], sd2["attn_l"], sd2["attn_r"], sd2.get("res_fc.weight"),
                sd2["bias"], 0.2, storage=O.Bf16Storage)[0]
    r.backward(gout)
    return {k: v.grad for k, v in sd2.items()}
base = grads(xm, gm)
for label, (x_, g_) in {"hip x, model g": (xh, gm), "model x, hip g": (xm, gh), "hip x, hip g": (xh, gh)}.items():
    o = grads(x_, g_)
    print(label, {k: round(rel_err(o[k], base[k]), 5) for k in ("attn_l", "attn_r", "fc.weight")})
