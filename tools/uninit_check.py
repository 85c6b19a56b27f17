"""Every torch.empty / empty_like float tensor is pre-filled with NaN (ints with a large value): a kernel that reads memory it
was never given shows up as NaN in the loss / gradients.  usage: dbg_uninit.py [config] [dtype]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
_e, _el = torch.empty, torch.empty_like
def _fill(t):
    if t.is_cuda and t.numel():
        if t.is_floating_point(): t.fill_(float("nan"))
        elif t.dtype in (torch.int32, torch.int64): t.fill_(0)
    return t
torch.empty = lambda *a, **k: _fill(_e(*a, **k))
torch.empty_like = lambda *a, **k: _fill(_el(*a, **k))
from spgnn_amd import models, synthetic
from spgnn_amd.configs import class_weight_list, get_config
from spgnn_amd.train import TrainStep
name = sys.argv[1] if len(sys.argv) > 1 else "st_pgat_spgnn_3"
cfg = get_config(name)
torch.manual_seed(0)
model = models.build_model(cfg.MODEL).cuda()
model.init(None); model.set_gcn_only()
if len(sys.argv) > 2 and sys.argv[2] == "bf16":
    models.set_storage_dtype(model, torch.bfloat16)
for train in (False, True):
    model.train(train)
    g = synthetic.make_batch(3, rank=0, device="cuda", pos_enc_dim=cfg.POS_ENC_DIM)
    ts = TrainStep(model, class_weight_list(cfg.CLASS_WEIGHTS), 1.0, 0.01, 0.9, seed=5)
    for i in range(2):
        loss = float(ts.step(g))
        gnan = int(torch.isnan(ts.bucket.flat_grad).sum())
        print(name, "train" if train else "eval", "step", i, "loss", loss, "NaNs in grads:", gnan, "in params:", int(torch.isnan(ts.bucket.flat_param).sum()), flush=True)
        if gnan:
            off = 0
            for n, p in model.named_parameters():
                if p.requires_grad:
                    k = p.numel(); c = int(torch.isnan(ts.bucket.flat_grad[off:off + k]).sum()); off += (k + 3) // 4 * 4
                    if c: print("   ", n, c, "/", k)
            sys.exit(1)
