import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spgnn_amd import ops
dev = "cuda"; N = 76410
for (K, C) in [(1063, 1024), (768, 512), (192, 4096), (384, 1024)]:
    Kp = (K + 3) // 4 * 4
    x = torch.randn(N, Kp, device=dev)[:, :K]; w = (torch.randn(C, Kp, device=dev) * 0.05)[:, :K]
    g = torch.randn(N, C, device=dev)
    for _ in range(2):
        ops.gemm_nt(x, w); ops.gemm_tn(g, x)
    torch.cuda.synchronize()
print("done")
