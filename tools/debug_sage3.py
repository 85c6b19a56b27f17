import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from oracle import dgl_cpu as O
from spgnn_amd import synthetic
from tests.test_hip_models import _build, _oracle
from spgnn_amd.configs import class_weight_list
from spgnn_amd.train import masked_weighted_ce
from tests.util import rel_err
cfg, model = _build("st_sage_3", seed=1)
g = synthetic.make_batch(2, rank=3, device="cuda", pos_enc_dim=None)
model.eval()
w = torch.tensor(class_weight_list(cfg.CLASS_WEIGHTS)); y = g.ndata["y"]
mask = torch.rand(y.shape[0], generator=torch.Generator().manual_seed(5)) < 0.5
loss = masked_weighted_ce(model(g)[0], y, mask.cuda(), w.cuda()); loss.backward()
refs, sd = _oracle(cfg, model, g, grad=True)
O.masked_weighted_ce(refs[0], y.cpu(), mask, w).backward()
for k, p in model.named_parameters():
    if p.grad is not None:
        e = rel_err(p.grad, sd[k].grad)
        if e > 1e-5: print(k, e)
src, dst = g.cpu().edges(); n = g.number_of_nodes()
h = g.ndata["fvs"].cpu()
for l in range(4):
    p = f"sage.g_layers.{l}."
    m = F.relu(F.linear(h, sd[p+"fc_pool.weight"], sd[p+"fc_pool.bias"])).detach()
    mx = O.spmm_max(src, dst, m, n)
    cnt = torch.zeros(n, m.shape[1]).index_add_(0, dst, (m[src] == mx[dst]).float())
    print("layer", l, "ties>1 & max>0:", int(((cnt > 1) & (mx > 0)).sum()), "/", cnt.numel(), " frac zero m:", float((m == 0).float().mean()))
    h = O.sage_conv_pool(src, dst, n, h, sd[p+"fc_pool.weight"], sd[p+"fc_pool.bias"], sd[p+"fc_self.weight"], sd[p+"fc_self.bias"], sd[p+"fc_neigh.weight"], sd[p+"fc_neigh.bias"], None, F.elu if l < 3 else None).detach()
