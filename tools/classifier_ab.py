"""Classifier head gnn_out (1024 -> 22 on 76 410 rows), forward + backward: the streaming skinny kernels
(ops.skinny_linear) against the matrix-core GEMM path (ops._LinearFn), one process, medians."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spgnn_amd import ops
dev = "cuda"; N = 76410
torch.manual_seed(0)
x = torch.randn(N, 1024, device=dev, requires_grad=True)
w = (torch.randn(22, 1024, device=dev) * 0.03).requires_grad_(True); b = torch.zeros(22, device=dev, requires_grad=True)
go = torch.randn(N, 22, device=dev)
def run(fn):
    y = fn()
    gx, gw, gb = torch.autograd.grad(y, (x, w, b), go)
    return y, gx, gw, gb
fa = lambda: ops.skinny_linear(x, w, b)
fb = lambda: ops._LinearFn.apply(x, w, b, ops.ACT_NONE)
ra, rb = run(fa), run(fb)
ref = torch.nn.functional.linear(x.double(), w.double(), b.double())
for name, r in (("skinny", ra), ("mfma", rb)):
    print(name, "fwd err %.2e" % float((r[0].double() - ref).abs().max() / ref.abs().max()),
          "gx err %.2e" % float((r[1].double() - go.double() @ w.double()).abs().max() / (go.double() @ w.double()).abs().max()),
          "gw err %.2e" % float((r[2].double() - go.double().t() @ x.double()).abs().max() / (go.double().t() @ x.double()).abs().max()))
def t(fn, it=10):
    a = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): run(fn)
    e.record(); torch.cuda.synchronize(); return a.elapsed_time(e) / it
res = {"skinny": [], "mfma": []}
for _ in range(5):
    res["skinny"].append(t(fa)); res["mfma"].append(t(fb))
for k, v in res.items(): print(k, "fwd+bwd median %.3f ms" % sorted(v)[2])
