import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import dgl_cpu as O
from spgnn_amd import ops
from spgnn_amd.graph import TreeGraph
from tests.util import tree_batch_edges
s, d, n = tree_batch_edges([23, 150, 1, 64], 5)
g = TreeGraph((s, d), n).to("cuda"); csc = g.csc()
src, dst = torch.from_numpy(s), torch.from_numpy(d)
for (H, D) in [(2,256),(8,64),(4,16),(8,256),(2,64),(4,64),(16,16)]:
    torch.manual_seed(0)
    ft = torch.randn(n, H*D, device="cuda"); el = torch.randn(n, H, device="cuda"); er = torch.randn(n, H, device="cuda")
    out, _, attn = ops.gat_fwd_raw(csc, ft, el, er, None, None, H, D, 0.2, 0)
    e = torch.nn.functional.leaky_relu(el.cpu()[src] + er.cpu()[dst], 0.2)
    a = O.edge_softmax(dst, e, n)
    ref = O.spmm_sum(src, dst, ft.cpu().view(n, H, D), n, a.unsqueeze(-1))
    err = (out.cpu().view(n, H, D) - ref).abs().amax(dim=(0, 2))
    a_hip = torch.empty_like(attn.cpu()); a_hip[csc.eid.cpu().long()] = attn.cpu()
    print(H, D, "out err per head", err.tolist(), "attn err", (a_hip - a).abs().amax(0).tolist())
