"""A/B the NT GEMM kernel generations in one process (interleaved rounds)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spgnn_amd import ops, _capi
lib = _capi.load()
dev = "cuda"
def t_once(fn, iters=8):
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters
variants = [1, 2, 3]
for v in variants:   # correctness first
    lib.spgnn_gemm_set_variant(v)
    for (M, N, K) in [(1, 1, 4), (33, 17, 36), (257, 129, 39), (300, 200, 1063), (5000, 512, 768), (4100, 130, 64), (4096, 128, 32)]:
        Kp = (K + 3) // 4 * 4
        a = torch.randint(-8, 9, (M, Kp), device=dev).float()[:, :K]; b = torch.randint(-8, 9, (N, Kp), device=dev).float()[:, :K]
        ok = torch.equal(ops.gemm_nt(a, b), a @ b.t())
        a2 = torch.randn(M, Kp, device=dev)[:, :K]; b2 = torch.randn(N, Kp, device=dev)[:, :K]
        ref = a2.double() @ b2.double().t()
        e = ((ops.gemm_nt(a2, b2, ops.pow2_scale(a2), ops.pow2_scale(b2)).double() - ref).abs().max() / ref.abs().max()).item()
        print(f"variant {v} M={M} N={N} K={K}: integer exact {ok} random err {e:.2e}", flush=True)
N = 76410
for (K, C) in [(1063, 1024), (768, 512), (384, 256), (192, 4096), (39, 512), (256, 256), (128, 128), (1024, 1063), (512, 768), (4096, 192)]:
    Kp = (K + 3) // 4 * 4
    x = torch.randn(N, Kp, device=dev)[:, :K]; w = (torch.randn(C, Kp, device=dev) * 0.05)[:, :K]
    sx, sw = ops.pow2_scale(x), ops.pow2_scale(w)
    out = torch.empty(N, C, device=dev)
    fn = lambda: ops.gemm_nt(x, w, sx, sw, out=out)
    res = {v: [] for v in variants}
    for v in variants:
        lib.spgnn_gemm_set_variant(v); fn()
    torch.cuda.synchronize()
    for r in range(5):
        for v in variants:
            lib.spgnn_gemm_set_variant(v); res[v].append(t_once(fn))
    fl = 2 * N * K * C / 1e9
    print(f"K={K} C={C}: " + " | ".join(f"v{v} {sorted(t)[2]:.3f} ms ({fl/sorted(t)[2]:.0f} TF-eq, {3*fl/sorted(t)[2]:.0f} raw)" for v, t in res.items()), flush=True)
lib.spgnn_gemm_set_variant(2)
