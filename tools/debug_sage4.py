import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from oracle import dgl_cpu as O
from spgnn_amd import synthetic, models
from spgnn_amd.configs import class_weight_list, get_config
from spgnn_amd.train import masked_weighted_ce
from tests.util import rel_err
cfg = get_config("st_sage_3")
g = synthetic.make_batch(2, rank=3, device="cuda", pos_enc_dim=None)
src, dst = g.cpu().edges(); n = g.number_of_nodes()
w = torch.tensor(class_weight_list(cfg.CLASS_WEIGHTS)); y = g.ndata["y"]
mask = torch.rand(y.shape[0], generator=torch.Generator().manual_seed(5)) < 0.5
for bias_mode in ["default", "normal_all", "normal_pool_only", "normal_not_pool"]:
  for loss_mode in ["cot", "ce"]:
    torch.manual_seed(1)
    model = models.build_model(cfg.MODEL).cuda(); model.init(None)
    with torch.no_grad():
        for nme, p in model.named_parameters():
            if nme.endswith("bias"):
                if bias_mode == "normal_all" or (bias_mode == "normal_pool_only" and "fc_pool" in nme) or (bias_mode == "normal_not_pool" and "fc_pool" not in nme):
                    p.normal_(0, 0.05)
    model.set_gcn_only(); model.eval()
    sd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    out = model(g)[0]; ref = O.net_forward("sage", sd, src, dst, n, g.ndata["fvs"].cpu())[0]
    if loss_mode == "cot":
        cot = torch.randn(n, 22, generator=torch.Generator().manual_seed(3))
        (out * cot.cuda()).sum().backward(); (ref * cot).sum().backward()
    else:
        masked_weighted_ce(out, y, mask.cuda(), w.cuda()).backward(); O.masked_weighted_ce(ref, y.cpu(), mask, w).backward()
    errs = {k: rel_err(p.grad, sd[k].grad) for k, p in model.named_parameters() if p.grad is not None}
    worst = max(errs, key=errs.get)
    d = (dict(model.named_parameters())[worst].grad.cpu() - sd[worst].grad)
    print(bias_mode, loss_mode, "fwd", f"{rel_err(out, ref):.2e}", "worst", worst, f"{errs[worst]:.2e}", "L2", f"{float(d.norm()/sd[worst].grad.norm()):.2e}",
          "frac>1e-4max", float((d.abs() > 1e-4 * sd[worst].grad.abs().max()).float().mean()))
