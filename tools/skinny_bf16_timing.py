"""Timings of the skinny (J <= 32 column) products on fp32 and bf16 rows at N = 76 410: scores_fwd / scores_bwd_w / scores_bwd_x."""
import torch, sys
sys.path.insert(0,'.')
from spgnn_amd import ops, ops_bf16
N=76410
def t_once(fn, iters=20):
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    for _ in range(3): fn()
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
for K,J in ((384,22),(128,4),(512,4)):
    x=torch.randn(N,K,device='cuda'); xb=ops_bf16.cast_rows(x); gs=torch.randn(N,J,device='cuda')
    w=torch.randn(J,K,device='cuda')
    print(K,J,'bwd_w f32 %.0f us  bf16 %.0f us | fwd f32 %.0f bf16 %.0f | bwd_x bf16 %.0f' % (t_once(lambda: ops.scores_bwd_w(gs,x)), t_once(lambda: ops_bf16.scores_bwd_w(gs,xb)), t_once(lambda: ops.scores_fwd(x,w)), t_once(lambda: ops_bf16.scores_fwd(xb,w)), t_once(lambda: ops_bf16.scores_bwd_x(gs,w,K))))
