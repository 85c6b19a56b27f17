"""Debug aid: per-parameter gradient distances of the bf16 path (HIP) to the storage model and to the fp64 oracle."""
import sys
import torch
sys.path.insert(0, ".")
from tests import test_hip_bf16 as T
from tests.util import rel_err
from oracle import dgl_cpu as O
from spgnn_amd import synthetic
from spgnn_amd.configs import class_weight_list
from spgnn_amd.train import masked_weighted_ce

for name in sys.argv[1:] or ["st_gat_3", "st_gat_6"]:
    cfg, model = T._build(name)
    g = synthetic.make_batch(3, rank=5, device="cuda", pos_enc_dim=cfg.POS_ENC_DIM)
    model.eval()
    w = torch.tensor(class_weight_list(cfg.CLASS_WEIGHTS))
    y = g.ndata["y"]
    mask = torch.rand(y.shape[0], generator=torch.Generator().manual_seed(5)) < 0.5
    logits, emb = model(g)
    masked_weighted_ce(logits, y, mask.cuda(), w.cuda()).backward()
    (m_logits, m_emb), sd_m = T._oracle_logits(cfg, model, g, torch.float64, O.Bf16Storage, grad=True)
    O.masked_weighted_ce(m_logits, y.cpu(), mask, w.double()).backward()
    (t_logits, t_emb), sd_t = T._oracle_logits(cfg, model, g, torch.float64, None, grad=True)
    O.masked_weighted_ce(t_logits, y.cpu(), mask, w.double()).backward()
    print(name, "logits", rel_err(logits, m_logits), rel_err(logits, t_logits), rel_err(m_logits, t_logits))
    for n, p in model.named_parameters():
        if p.requires_grad:
            print(f"  {n:40s} hip-model {rel_err(p.grad, sd_m[n].grad):.4f} hip-true {rel_err(p.grad, sd_t[n].grad):.4f} "
                  f"model-true {rel_err(sd_m[n].grad, sd_t[n].grad):.4f} |g| {float(sd_t[n].grad.abs().max()):.3e}")
