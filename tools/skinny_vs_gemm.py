"""The classifier's skinny products (J = 22) on the dedicated kernels vs the split-fp16 matrix-core GEMMs, N = 76 410."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spgnn_amd import ops
R, K, J = int(os.environ.get("ROWS", "76410")), 1024, 22
x = torch.randn(R, K, device="cuda"); w = torch.randn(J, K, device="cuda") * 0.05; g = torch.randn(R, 24, device="cuda")[:, :J]
def t(fn, it=20):
    for _ in range(3): fn()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(5):
        a.record()
        for _ in range(it): fn()
        b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) / it * 1e3)
    return sorted(ts)[2]
sx, sw = ops.pow2_scale(x), ops.pow2_scale(w)
wps = ops.presplit(w, scale=sw)[0]
out = torch.empty(R, J + 2, device="cuda")[:, :J]
print("scores_fwd        %.1f us" % t(lambda: ops.scores_fwd(x, w)))
for tile in (2, 4):
    print("gemm_nt tile%d     %.1f us" % (tile, t(lambda: ops.gemm_nt(x, wps, sx, sw, out=out, tile=tile, b_presplit=True))))
ref = x.double() @ w.double().t()
print("  err scores_fwd %.2e  gemm %.2e" % (float((ops.scores_fwd(x, w).double() - ref).abs().max() / ref.abs().max()),
                                             float((ops.gemm_nt(x, wps, sx, sw, b_presplit=True).double() - ref).abs().max() / ref.abs().max())))
print("scores_bwd_w      %.1f us" % t(lambda: ops.scores_bwd_w(g, x)))
gp = torch.zeros(R, 24, device="cuda"); gp[:, :J] = g
sg = ops.pow2_scale(gp)
print("gemm_tn (24 cols) %.1f us" % t(lambda: ops.gemm_tn(gp, x, sg, sx)))
