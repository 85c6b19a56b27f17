import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spgnn_amd import ops, _capi
dev = "cuda"; M = 76410
def t_once(fn, iters=6):
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters
lib = _capi.load()
for (K, N, ldc, act) in [(384, 1024, 2048, ops.ACT_ELU), (384, 1024, 1024, ops.ACT_NONE), (384, 1024, 2048, ops.ACT_NONE), (384, 1024, 1024, ops.ACT_ELU), (384, 1024, 1024, ops.ACT_RELU), (1024, 384, 768, ops.ACT_NONE), (768, 512, 512, 0), (384, 256, 256, 0), (256, 256, 256, 0), (512, 768, 768, 0), (256, 384, 384, 0), (128, 256, 256, 0)]:
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.05
    sx, sw = ops.pow2_scale(x), ops.pow2_scale(w)
    big = torch.empty(M, ldc, device=dev); bias = torch.randn(N, device=dev)
    u = torch.randn(M, 4, device=dev); v = torch.randn(4, (N + 15) // 16 * 16, device=dev)
    useJ = os.environ.get("WMJ") == "1"
    fn = lambda: ops.gemm_nt(x, w, sx, sw, out=big[:, :N], bias=bias if act else None, act=act, upd_u=u if useJ else None, upd_v=v if useJ else None)
    res = {}
    for v in (3, 4):
        lib.spgnn_gemm_set_variant(v); fn()
    torch.cuda.synchronize()
    for r in range(5):
        for v in (3, 4):
            lib.spgnn_gemm_set_variant(v); res.setdefault(v, []).append(t_once(fn))
    lib.spgnn_gemm_set_variant(2)
    fl = 2.0 * M * N * K
    print(f"K={K} N={N} ldc={ldc} act={act}: " + " | ".join(f"WM={2 if v == 3 else 4} {sorted(t)[2]*1e3:.0f}us ({fl/sorted(t)[2]/1e9:.0f} TF)" for v, t in res.items()), flush=True)
