"""Bitwise run-to-run check of a full forward + backward at bench size (dropout off): no atomics and no
order-dependent reductions anywhere, so every gradient must repeat exactly.  (This is the check that exposed the
packed-op / cross-lane hazard documented in DESIGN.md §4.1.)  usage: determinism_check.py [config] [trees] [reps]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spgnn_amd import models, synthetic
from spgnn_amd.configs import class_weight_list, get_config
from spgnn_amd.train import masked_weighted_ce

name = sys.argv[1] if len(sys.argv) > 1 else "st_pgat_spgnn_3"
trees = int(sys.argv[2]) if len(sys.argv) > 2 else 512
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
cfg = get_config(name)
torch.manual_seed(0)
model = models.build_model(cfg.MODEL).cuda().eval()
g = synthetic.make_batch(trees, rank=0, device="cuda", pos_enc_dim=cfg.POS_ENC_DIM)
w = torch.tensor(class_weight_list(cfg.CLASS_WEIGHTS), device="cuda")
y = g.ndata["y"]
mask = torch.rand(y.shape[0], device="cuda") < 0.5
ref = None
bad = 0
for r in range(reps):
    model.zero_grad(set_to_none=True)
    out = model(g)[0]
    loss = masked_weighted_ce(out, y, mask, w)
    loss.backward()
    cur = {"logits": out.detach().clone(), **{n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}}
    if ref is None:
        ref = cur
    else:
        for k in ref:
            if not torch.equal(ref[k], cur[k]):
                d = (ref[k] - cur[k]).abs()
                print(f"rep {r}: {k} differs: max |diff| {float(d.max()):.3e}, {int((d > 0).sum())} elements")
                bad += 1
print(name, trees, "trees:", "bitwise reproducible" if bad == 0 else f"{bad} tensors differed", f"over {reps} runs")
sys.exit(1 if bad else 0)
