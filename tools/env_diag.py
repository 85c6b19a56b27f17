import os, sys, torch
sys.path.insert(0, "/root/repo")
import tests.test_hip_bf16 as T
from oracle import dgl_cpu as O
from spgnn_amd import synthetic
from spgnn_amd.configs import class_weight_list
from spgnn_amd.train import masked_weighted_ce
from tests.util import rel_err
cfg, model = T._build("st_gat_3")
g = synthetic.make_batch(3, rank=5, device="cuda", pos_enc_dim=cfg.POS_ENC_DIM)
model.eval()
w = torch.tensor(class_weight_list(cfg.CLASS_WEIGHTS)); y = g.ndata["y"]
mask = torch.rand(y.shape[0], generator=torch.Generator().manual_seed(5)) < 0.5
logits, emb = model(g)
masked_weighted_ce(logits, y, mask.cuda(), w.cuda()).backward()
(m_logits, _), sd_m = T._oracle_logits(cfg, model, g, torch.float64, O.Bf16Storage, grad=True)
O.masked_weighted_ce(m_logits, y.cpu(), mask, w.double()).backward()
names = [n for n, p in model.named_parameters() if p.requires_grad and ".attn_" in n]
hip = {n: rel_err(dict(model.named_parameters())[n].grad, sd_m[n].grad) for n in names}
for rate in (0.01, 0.04):
    T.FLIP_RATE, T.FLIP_TRIALS = rate, 1
    devs = {n: [] for n in names}
    for k in range(12):
        orig_seed = 1234 + k
        import types
        # one trial with its own seed
        gen_env = T._flip_envelope
        # patch the generator seed by re-implementing the single trial
        orig = O._rb; gen = torch.Generator().manual_seed(orig_seed)
        def flipping(x):
            r = orig(x)
            m = torch.rand(r.shape, generator=gen) < rate
            up = torch.rand(r.shape, generator=gen) < 0.5
            ulp1 = torch.ldexp(torch.ones_like(r), torch.frexp(r)[1] - 8)
            return torch.where(m & (r != 0), orig(r + torch.where(up, ulp1, -ulp1)), r)
        try:
            O._rb = flipping
            (lg, _e), sd = T._oracle_logits(cfg, model, g, torch.float64, O.Bf16Storage, grad=True)
            O.masked_weighted_ce(lg, y.cpu(), mask, w.double()).backward()
        finally:
            O._rb = orig
        for n in names: devs[n].append(rel_err(sd[n].grad, sd_m[n].grad))
    for n in names:
        v = sorted(devs[n])
        print(f"rate {rate} {n:30s} hip {hip[n]:.3f}  trials min {v[0]:.3f} med {v[len(v)//2]:.3f} max {v[-1]:.3f}", flush=True)
