"""fp32-operand GEMM vs split-once + planes GEMM at the st_pgat_spgnn_3 shapes (one process, interleaved)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spgnn_amd import ops
dev = "cuda"; M = int(sys.argv[1]) if len(sys.argv) > 1 else 76410
def t_once(fn, iters=6):
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters
for (K, N) in [(1063, 1024), (768, 512), (384, 1024), (384, 256), (512, 768), (1024, 384), (256, 384)]:
    Kp4 = (K + 3) // 4 * 4
    x = torch.randn(M, Kp4, device=dev)[:, :K]; w = (torch.randn(N, Kp4, device=dev) * 0.05)[:, :K]
    sx, sw = ops.pow2_scale(x), ops.pow2_scale(w)
    out1 = torch.empty(M, N, device=dev); out2 = torch.empty(M, N, device=dev)
    f_old = lambda: ops.gemm_nt(x, w, sx, sw, out=out1)
    xp, wp = ops.split_rows(x, sx), ops.split_rows(w, sw)
    f_new = lambda: ops.gemm_nt_planes(xp, wp, out=out2)
    f_split = lambda: ops.split_rows(x, sx)
    f_old(); f_new(); torch.cuda.synchronize()
    same = torch.equal(out1, out2)
    err = float((out1 - out2).abs().max())
    r = {k: [] for k in ("old", "new", "split")}
    for _ in range(5):
        r["old"].append(t_once(f_old)); r["new"].append(t_once(f_new)); r["split"].append(t_once(f_split))
    med = {k: sorted(v)[2] * 1e3 for k, v in r.items()}
    fl = 2.0 * M * N * K
    print(f"K={K} N={N}: fp32-operand {med['old']:.0f} us ({fl/med['old']/1e6:.0f} TF)  planes {med['new']:.0f} us ({fl/med['new']/1e6:.0f} TF)  "
          f"split(A) {med['split']:.0f} us  bit-equal {same} maxdiff {err:.2e}", flush=True)
