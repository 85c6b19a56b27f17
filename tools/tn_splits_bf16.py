"""Split-K count of spgnn_gemm_tn_bf16 (weight gradients on bf16 rows) at the st_gat_6 shapes, incl. the partial-sum kernel."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spgnn_amd import ops_bf16
R = 76410
def t_once(fn, iters=10):
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters
for (M, N) in [(1024, 1024), (512, 512), (256, 256), (256, 128), (128, 128)]:
    g = ops_bf16.cast_rows(torch.randn(R, M, device="cuda") * 1e-3)
    x = ops_bf16.cast_rows(torch.randn(R, N, device="cuda"))
    res = {}
    for sp in (16, 32, 64, 96, 128, 192, 256):
        fn = lambda: ops_bf16.gemm_tn(g, x, splits=sp)
        fn(); fn()
        ts = sorted(t_once(fn) for _ in range(5))
        res[sp] = ts[2]
    d = ops_bf16._tn_splits(R, M, N)
    print(f"M={M} N={N} default {d}: " + "  ".join(f"{sp}: {t*1e3:.0f}us" for sp, t in res.items()), flush=True)
