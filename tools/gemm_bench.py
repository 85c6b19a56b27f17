"""Correctness (vs fp64) and speed of spgnn_gemm_nt against rocBLAS fp32 at the model's projection shapes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spgnn_amd import ops
def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters
dev = "cuda"
# small odd shapes first: exactness of tails
for (M, N, K) in [(1, 1, 4), (33, 17, 36), (130, 257, 100), (128, 128, 32), (300, 200, 1063), (257, 129, 39)]:
    Kp = (K + 3) // 4 * 4
    a = torch.randn(M, Kp, device=dev)[:, :K]; b = torch.randn(N, Kp, device=dev)[:, :K]
    c = ops.gemm_nt(a, b, ops.pow2_scale(a), ops.pow2_scale(b))
    ref = a.double() @ b.double().t()
    c32 = a @ b.t()
    print(f"M={M} N={N} K={K}: err f16x3 {((c.double()-ref).abs().max()/ref.abs().max()).item():.2e}  fp32 {((c32.double()-ref).abs().max()/ref.abs().max()).item():.2e}", flush=True)
# exact integer data catches layout bugs
a = torch.randint(-8, 9, (256, 64), device=dev).float(); b = torch.randint(-8, 9, (192, 64), device=dev).float()
assert torch.equal(ops.gemm_nt(a, b), a @ b.t()), "integer GEMM mismatch"
print("integer exact OK")
N = 76410
for (K, C) in [(1063, 1024), (768, 512), (384, 256), (192, 4096), (39, 512), (256, 256), (128, 128), (1024, 1063), (512, 768), (4096, 192)]:
    Kp = (K + 3) // 4 * 4
    x = torch.randn(N, Kp, device=dev)[:, :K]; w = (torch.randn(C, Kp, device=dev) * 0.05)[:, :K]
    sx, sw = ops.pow2_scale(x), ops.pow2_scale(w)
    out = torch.empty(N, C, device=dev)
    t16 = timeit(lambda: ops.gemm_nt(x, w, sx, sw, out=out))
    t32 = timeit(lambda: torch.mm(x, w.t()))
    tsc = timeit(lambda: ops.pow2_scale(x))
    ref = x[:4096].double() @ w.double().t()
    e16 = ((out[:4096].double() - ref).abs().max() / ref.abs().max()).item()
    e32 = (((x[:4096] @ w.t()).double() - ref).abs().max() / ref.abs().max()).item()
    fl = 2 * N * K * C / 1e9
    print(f"K={K} C={C}: f16x3 {t16:.3f} ms ({fl/t16:.0f} TF-equiv, {3*fl/t16:.0f} raw) err {e16:.2e} | rocBLAS fp32 {t32:.3f} ms ({fl/t32:.0f} TF) err {e32:.2e} | absmax pass {tsc:.3f} ms", flush=True)

print("---- TN (weight gradient) ----")
for (R, M, N_) in [(40, 7, 5), (100, 130, 33), (1000, 256, 128), (513, 64, 300)]:
    Mp, Np = (M + 3) // 4 * 4, (N_ + 3) // 4 * 4
    a = torch.randint(-8, 9, (R, Mp), device=dev).float()[:, :M]; b = torch.randint(-8, 9, (R, Np), device=dev).float()[:, :N_]
    c = ops.gemm_tn(a, b)
    print(f"R={R} M={M} N={N_}: integer exact {torch.equal(c, a.t() @ b)}", flush=True)
for (K, C) in [(1063, 1024), (768, 512), (384, 256), (192, 4096), (39, 512), (256, 256), (128, 128)]:
    Kp = (K + 3) // 4 * 4
    x = torch.randn(N, Kp, device=dev)[:, :K]; g = (torch.randn(N, C, device=dev) * 1e-5)
    sx, sg = ops.pow2_scale(x), ops.pow2_scale(g)
    t16 = timeit(lambda: ops.gemm_tn(g, x, sg, sx))
    t32 = timeit(lambda: ops._dw_gemm(g, x))
    out = ops.gemm_tn(g, x, sg, sx)
    ref = g.double().t() @ x.double()
    e16 = ((out.double() - ref).abs().max() / ref.abs().max()).item()
    e32 = ((ops._dw_gemm(g, x).double() - ref).abs().max() / ref.abs().max()).item()
    fl = 2 * N * K * C / 1e9
    print(f"dW K={K} C={C}: f16x3 {t16:.3f} ms ({fl/t16:.0f} TF-equiv) err {e16:.2e} | rocBLAS splitK fp32 {t32:.3f} ms ({fl/t32:.0f} TF) err {e32:.2e}", flush=True)
