"""spgnn_gemm_nt tile shapes per product: variant 3 = 128x128 tiles, 4 = 256x128, 5 = 256x256, 2 = the library's choice."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spgnn_amd import _capi, ops
lib = _capi.load(); dev = "cuda"; M = 76410
def t_once(fn, iters=6):
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters
for (K, N) in [(1024, 384), (256, 256), (384, 256), (256, 384), (128, 128), (768, 512), (512, 768), (384, 1024), (1063, 1024)]:
    Kp4 = (K + 3) // 4 * 4
    x = torch.randn(M, Kp4, device=dev)[:, :K]; w = (torch.randn(N, Kp4, device=dev) * 0.05)[:, :K]
    sx, sw = ops.pow2_scale(x), ops.pow2_scale(w)
    out = torch.empty(M, N, device=dev)
    fn = lambda: ops.gemm_nt(x, w, sx, sw, out=out)
    res = {}
    for r in range(5):
        for v in (2, 3, 4, 5):
            lib.spgnn_gemm_set_variant(v); fn(); torch.cuda.synchronize()
            res.setdefault(v, []).append(t_once(fn))
    lib.spgnn_gemm_set_variant(2)
    print(f"K={K} N={N}: " + " | ".join(f"v{v} {sorted(t)[2]*1e3:.0f}us" for v, t in res.items()), flush=True)
