import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from oracle import dgl_cpu as O
from spgnn_amd import models, synthetic
from spgnn_amd.configs import get_config, class_weight_list
from spgnn_amd.train import masked_weighted_ce
from tests.util import rel_err
cfg = get_config("st_sage_3")
torch.manual_seed(1)
model = models.build_model(cfg.MODEL).cuda(); model.init(None); model.set_gcn_only(); model.eval()
g = synthetic.make_batch(2, rank=3, device="cuda", pos_enc_dim=None)
src, dst = g.cpu().edges(); n = g.number_of_nodes()
sd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in model.state_dict().items()}
out = model(g)[0]; ref = O.net_forward("sage", sd, src, dst, n, g.ndata["fvs"].cpu())[0]
print("fwd", rel_err(out, ref))
cot = torch.randn(n, 22)
(out * cot.cuda()).sum().backward(); (ref * cot).sum().backward()
for k, p in model.named_parameters():
    if p.grad is not None: print(k, rel_err(p.grad, sd[k].grad))
# tie statistics in layer 0
x = g.ndata["fvs"].cpu()
m = F.relu(F.linear(x, sd["sage.g_layers.0.fc_pool.weight"], sd["sage.g_layers.0.fc_pool.bias"])).detach()
mm = m[src]
mx = O.spmm_max(src, dst, m, n)
ties = (mm == mx[dst]).float()
cnt = torch.zeros(n, m.shape[1]).index_add_(0, dst, ties)
print("positions with ties>1 and max>0:", int(((cnt > 1) & (mx > 0)).sum()), "of", cnt.numel())
