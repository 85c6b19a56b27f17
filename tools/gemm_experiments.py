"""Which rocBLAS/hipBLASLt formulation is fastest for the weight-gradient and skinny GEMMs?"""
import os, sys, torch
def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters
N = 76410
dev = "cuda"
for lib in ["default", "hipblaslt", "rocblas"]:
    if lib != "default":
        try:
            torch.backends.cuda.preferred_blas_library(lib)
        except Exception as e:
            print("cannot select", lib, e); continue
    print("== blas:", lib, flush=True)
    for (K, C) in [(1063, 1024), (1064, 1024), (768, 512), (384, 256), (192, 4096), (256, 256), (39, 512)]:
        x = torch.randn(N, K, device=dev); gy = torch.randn(N, C, device=dev); w = torch.randn(C, K, device=dev)
        fl = 2 * N * K * C / 1e9
        t1 = timeit(lambda: torch.mm(gy.t(), x))
        t2 = timeit(lambda: torch.mm(x.t(), gy))
        t3 = timeit(lambda: torch.mm(x, w.t()))
        t4 = timeit(lambda: torch.mm(gy, w))
        # manual split-K in 4 chunks via bmm
        xs = x[: N // 4 * 4].view(4, N // 4, K); gs = gy[: N // 4 * 4].view(4, N // 4, C)
        t5 = timeit(lambda: torch.bmm(gs.transpose(1, 2), xs).sum(0))
        print(f"K={K} C={C}: dW gy^T x {t1:.3f} ms ({fl/t1:.0f} TF) | x^T gy {t2:.3f} ({fl/t2:.0f}) | splitK4 {t5:.3f} ({fl/t5:.0f}) | fwd {t3:.3f} ({fl/t3:.0f}) | dX {t4:.3f} ({fl/t4:.0f})", flush=True)
        del x, gy, w
    # skinny
    for (K, C) in [(1063, 4), (768, 4), (192, 4), (256, 2)]:
        x = torch.randn(N, K, device=dev); w = torch.randn(C, K, device=dev); gs = torch.randn(N, C, device=dev)
        gx = torch.randn(N, K, device=dev)
        print(f"skinny K={K} C={C}: fwd {timeit(lambda: torch.mm(x, w.t())):.3f} dW {timeit(lambda: torch.mm(gs.t(), x)):.3f} "
              f"dX-addmm {timeit(lambda: gx.addmm_(gs, w)):.3f} ms ; bytes-bound fwd ~{N*K*4/5e9:.3f} ms", flush=True)
