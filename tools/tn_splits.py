"""Weight-gradient GEMM (spgnn_gemm_tn) time against the number of K-splits, at the bench shapes.
A split count that is a multiple of 8 puts every split on one XCD (by_xcd placement)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spgnn_amd import _capi, ops
dev = "cuda"; R = 76410
lib = _capi.load()
def run(a, b, sa, sb, splits):
    M, N = a.shape[1], b.shape[1]
    ldc = (N + 3) // 4 * 4
    part = torch.empty((splits, M, ldc), dtype=torch.float32, device=dev)
    def fn():
        _capi.check(lib.spgnn_gemm_tn(a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), part.data_ptr(), ldc, M * ldc, splits, R, M, N,
                                      sa.data_ptr(), sb.data_ptr(), 0, ldc, M * ldc, torch.cuda.current_stream().cuda_stream), "tn")
        return ops.sum_partials(part)
    for _ in range(2): fn()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(5):
        e0.record()
        for _ in range(4): fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 4)
    return sorted(ts)[2]
R = int(sys.argv[1]) if len(sys.argv) > 1 else R
CAND = {72: [7, 14, 21, 28, 32, 35], 24: [21, 32, 42, 63, 64], 6: [42, 64, 85, 128, 170], 4: [64, 96, 128, 192, 256], 1: [64, 128, 256, 512]}
for (M, N) in [(1024, 1063), (1024, 384), (512, 768), (256, 384), (256, 256), (512, 39), (128, 128)]:
    Np = (N + 3) // 4 * 4
    a = torch.randn(R, M, device=dev); b = torch.randn(R, Np, device=dev)[:, :N]
    sa, sb = ops.pow2_scale(a), ops.pow2_scale(b)
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    cur = max(1, min(64, 512 // tiles, R // 256))
    cand = sorted(set([cur] + [c for c in CAND.get(tiles, [16, 32, 64]) if c <= R // 64]))
    print(f"M={M} N={N} tiles={tiles} current={cur}: " + " | ".join(f"{s}:{run(a, b, sa, sb, s)*1e3:.0f}us" for s in cand), flush=True)
