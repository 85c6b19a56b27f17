"""bf16 NT product: 256 x 256 tile on v_mfma_f32_32x32x16_bf16 (tile 5) vs v_mfma_f32_16x16x32_bf16 (tile 6), interleaved
rounds in one process, random operands; also checks tile 6 against tile 5 (fp32 rounding) and exactly on integer data."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spgnn_amd import ops, ops_bf16

R = 76410
dev = "cuda"

def t_once(fn, iters=10):
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters

# correctness
ai = torch.randint(-3, 4, (1000, 200), device=dev).float(); bi = torch.randint(-3, 4, (512, 200), device=dev).float()
ra, rb = ops_bf16.cast_rows(ai), ops_bf16.cast_rows(bi)
o6 = ops_bf16.gemm_nt(ra, rb, out_f32=True, tile=6)
print("tile 6 exact on integers:", torch.equal(o6, ai @ bi.t()), flush=True)
for (C, K) in [(1024, 1024), (1024, 384), (512, 768), (2048, 1024), (1024, 4096)]:
    x = ops_bf16.cast_rows(torch.randn(R, K, device=dev))
    w = ops_bf16.cast_rows(torch.randn(C, K, device=dev) * 0.05)
    out = ops_bf16.empty_rows(R, C, dev)
    res = {5: [], 6: []}
    o5 = ops_bf16.gemm_nt(x, w, out_f32=True, tile=5); o6 = ops_bf16.gemm_nt(x, w, out_f32=True, tile=6)
    err = float((o5 - o6).abs().max() / o5.abs().max())
    for t in (5, 6):
        fn = lambda: ops_bf16.gemm_nt(x, w, out=out, tile=t)
        fn(); fn()
    torch.cuda.synchronize()
    for _ in range(9):
        for t in (5, 6):
            res[t].append(t_once(lambda: ops_bf16.gemm_nt(x, w, out=out, tile=t)))
    fl = 2.0 * R * C * K
    med = {t: sorted(v)[len(v) // 2] for t, v in res.items()}
    print(f"bf16 nt C={C} K={K}: 32x32x16 {med[5]*1e3:.0f} us ({fl/med[5]/1e9:.0f} TF)  16x16x32 {med[6]*1e3:.0f} us ({fl/med[6]/1e9:.0f} TF)  "
          f"ratio {med[5]/med[6]:.3f}  max rel diff {err:.2e}", flush=True)
