"""Weight-gradient GEMM: 128-row against 256-row block tiles over split counts, at the step's shapes (one process).
usage: tn_tiles.py [R]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spgnn_amd import ops
R = int(sys.argv[1]) if len(sys.argv) > 1 else 76410
def run(g, x, sg, sx, tile, splits, ps=None):
    def fn():
        return ops.gemm_tn(g, ps if ps is not None else x, sg, sx, tile=tile, splits=splits, b_presplit=ps is not None)
    for _ in range(2): fn()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(5):
        e0.record()
        for _ in range(4): fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 4)
    return sorted(ts)[2] * 1e3
for (M, N) in [(1024, 1063), (1024, 384), (512, 768), (256, 384), (256, 256)]:
    Np = (N + 3) // 4 * 4
    g = torch.randn(R, M, device="cuda"); x = torch.randn(R, Np, device="cuda")[:, :N]
    sg, sx = ops.pow2_scale(g), ops.pow2_scale(x)
    ps = ops.presplit(x, scale=sx)[0] if N == 1063 else None
    line = [f"M={M} N={N}"]
    for tile in (128, 256):
        cur = ops.TnProblem(g, x, sg, sx, tile=tile).splits
        cands = sorted({cur, 16, 21, 32, 43, 64} if M * N >= 1024 * 384 else {cur, 32, 64, 85, 128})
        line.append(f"tile {tile} (default {cur}): " + " ".join(f"{s_}:{run(g, x, sg, sx, tile, s_, ps):.0f}" for s_ in cands))
    print(" | ".join(line), flush=True)
