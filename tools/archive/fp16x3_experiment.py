"""Does a K-concatenated fp16 GEMM with fp32 output (hipBLASLt via torch.mm out_dtype) beat the fp32 GEMM,
and how accurate is the 3-product fp16 split [Ah Ah Al].[Bh Bl Bh]^T ?"""
import torch, sys
def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters
dev = "cuda"; N = 76410
def split(x):
    m = x.abs().max().clamp(min=1e-30)
    s = torch.exp2(14 - torch.ceil(torch.log2(m)))
    xs = x * s
    h = xs.half(); l = (xs - h.float()).half()
    return h, l, s
for (K, C) in [(1064, 1024), (768, 512), (192, 4096), (384, 256)]:
    x = torch.randn(N, K, device=dev); w = torch.randn(C, K, device=dev) * 0.05
    ref64 = (x.double() @ w.double().t())
    y32 = x @ w.t()
    xh, xl, sx = split(x); wh, wl, sw = split(w)
    A = torch.cat([xh, xh, xl], 1).contiguous(); B = torch.cat([wh, wl, wh], 1).contiguous()
    try:
        y = torch.mm(A, B.t(), out_dtype=torch.float32) / (sx * sw)
    except Exception as e:
        print("out_dtype unsupported:", repr(e)[:300]); sys.exit(0)
    e32 = ((y32.double() - ref64).abs().max() / ref64.abs().max()).item()
    e16 = ((y.double() - ref64).abs().max() / ref64.abs().max()).item()
    t32 = timeit(lambda: x @ w.t())
    t16 = timeit(lambda: torch.mm(A, B.t(), out_dtype=torch.float32))
    t16plain = timeit(lambda: torch.mm(xh, wh.t(), out_dtype=torch.float32))
    tsplit = timeit(lambda: torch.cat([*split(x)[:2]], 1))
    fl = 2 * N * K * C / 1e9
    print(f"K={K} C={C}: fp32 {t32:.3f} ms ({fl/t32:.0f} TF) err {e32:.2e} | fp16x3 Kcat {t16:.3f} ms ({fl/t16:.0f} TF-equiv, raw {3*fl/t16:.0f} TF) err {e16:.2e} | plain fp16 {t16plain:.3f} ms (raw {fl/t16plain:.0f} TF) | torch split pass {tsplit:.3f} ms", flush=True)
    # dW-like: (C x N) . (N x K) with N as the reduction dim, operands stored transposed (C x 3N), (K x 3N)
    gy = torch.randn(N, C, device=dev) * 1e-5
    gh, gl, sg = split(gy)
    At = torch.cat([gh, gh, gl], 0).t().contiguous(); Bt = torch.cat([xh, xl, xh], 0).t().contiguous()
    refw = gy.double().t() @ x.double()
    yw = torch.mm(At, Bt.t(), out_dtype=torch.float32) / (sg * sx)
    yw32 = gy.t() @ x
    tw16 = timeit(lambda: torch.mm(At, Bt.t(), out_dtype=torch.float32))
    tw32 = timeit(lambda: gy.t() @ x)
    print(f"   dW: fp32 {tw32:.3f} ms err {((yw32.double()-refw).abs().max()/refw.abs().max()).item():.2e} | fp16x3 {tw16:.3f} ms ({fl/tw16:.0f} TF-equiv) err {((yw.double()-refw).abs().max()/refw.abs().max()).item():.2e}", flush=True)
    del x, w, A, B, At, Bt, gy
