import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from oracle import dgl_cpu as O
from spgnn_amd import nn as snn, ops, synthetic
from spgnn_amd.graph import TreeGraph
from tests.util import tree_batch_edges, rel_err
torch.manual_seed(0)
s, d, n = tree_batch_edges([150, 140], 3)
g = TreeGraph((s, d), n).to("cuda"); src, dst = torch.from_numpy(s), torch.from_numpy(d)
for kind in ["randn", "relu"]:
    x = torch.randn(n, 256)
    if kind == "relu": x = x.clamp(min=0)
    xg = x.cuda().requires_grad_(True); xc = x.clone().requires_grad_(True)
    cot = torch.randn(n, 256)
    o = ops.spmm_max(g.csc(), xg); r = O.spmm_max(src, dst, xc, n)
    (o * cot.cuda()).sum().backward(); (r * cot).sum().backward()
    print(kind, "fwd", rel_err(o, r), "grad x", rel_err(xg.grad, xc.grad))
    # through relu of a linear
    lin = torch.nn.Linear(256, 256)
    xg = x.cuda().requires_grad_(True); xc = x.clone().requires_grad_(True)
    ling = torch.nn.Linear(256, 256).cuda(); ling.load_state_dict(lin.state_dict())
    o = ops.spmm_max(g.csc(), F.relu(ling(xg))); r = O.spmm_max(src, dst, F.relu(lin(xc)), n)
    (o * cot.cuda()).sum().backward(); (r * cot).sum().backward()
    print(kind, "relu(lin) fwd", rel_err(o, r), "grad x", rel_err(xg.grad, xc.grad), "grad W", rel_err(ling.weight.grad, lin.weight.grad))
sg = snn.SAGEConv(256, 64, "pool", activation=F.elu).cuda()
sd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in sg.state_dict().items()}
x = torch.randn(n, 256).clamp(min=0)
xg = x.cuda().requires_grad_(True)
o = sg(g, xg)
r = O.sage_conv_pool(src, dst, n, x, sd["fc_pool.weight"], sd["fc_pool.bias"], sd["fc_self.weight"], sd["fc_self.bias"], sd["fc_neigh.weight"], sd["fc_neigh.bias"], None, F.elu)
cot = torch.randn(n, 64)
(o * cot.cuda()).sum().backward(); (r * cot).sum().backward()
for k, p in sg.named_parameters():
    print(k, rel_err(p.grad, sd[k].grad))
