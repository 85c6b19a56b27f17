"""st_gin_3 at 64 trees: the layer outputs in the split products' standard and wide-range form against an fp64 evaluation in plain torch, with the
units whose LeakyReLU branch differs (the reason GIN's gradients are compared at 1e-2 in tests/test_hip_parity_at_size.py)."""
import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import dgl_cpu as O
from spgnn_amd import models, synthetic, ops, nn as snn
from spgnn_amd.configs import class_weight_list, get_config
from tests.util import rel_err
from tests.test_hip_parity_at_size import _build, _oracle
cfg, model = _build("st_gin_3", seed=11)
g = synthetic.make_batch(64, rank=0, device="cuda", pos_enc_dim=None)
layers = list(model.gin.gin_layers)
def run(wide):
    ops.GEMM_WIDE = wide
    outs = []
    hs = [l.register_forward_hook(lambda m, i, o, outs=outs: outs.append((o[0] if isinstance(o, tuple) else o).detach().clone())) for l in layers]
    with torch.no_grad(): model(g)
    for h in hs: h.remove()
    return outs
a = run(False); b = run(True)
# fp64 reference of every layer on the GPU in plain torch
src, dst = g.edges(); src, dst = src.long(), dst.long()
x = g.ndata["fvs"].double()
refs = []
for l in layers:
    f = l.apply_func
    agg = torch.zeros_like(x).index_add_(0, dst, x[src])
    deg = torch.zeros(x.shape[0], dtype=torch.float64, device=x.device).index_add_(0, dst, torch.ones_like(dst, dtype=torch.float64))
    agg = agg / deg.clamp(min=1)[:, None]
    h = (1 + l.eps.double()) * x + agg
    h = torch.nn.functional.leaky_relu(h @ f[0].weight.double().t() + f[0].bias.double(), 0.01)
    h = torch.nn.functional.leaky_relu(h @ f[3].weight.double().t() + f[3].bias.double(), 0.01)
    refs.append(h.detach()); x = h.detach()
for i, (p, q, r) in enumerate(zip(a, b, refs)):
    d_n = (p.double() - r).abs(); flips = (p > 0) != (r > 0)
    print("   |pre| of the flipped units (ref):", [f"{float(v):.1e}" for v in r[flips].abs().flatten()[:8]], "narrow-vs-wide flips", int(((p > 0) != (q > 0)).sum()))
    print(i, tuple(p.shape), "narrow vs ref", rel_err(p, r), "wide vs ref", rel_err(q, r), "sign flips narrow", int(((p > 0) != (r > 0)).sum()), "wide", int(((q > 0) != (r > 0)).sum()),
          "max", float(r.abs().max()), "rms", float(r.pow(2).mean().sqrt()))
