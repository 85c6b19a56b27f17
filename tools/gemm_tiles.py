"""Every NT product shape of the st_pgat_spgnn_3 step at 512 trees under each block tile (2 = 128x128, 4 = 256x128,
5 = 256x256; 0 = the library's own choice), weights pre-split, interleaved rounds in one process.  [act = ELU epilogue]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spgnn_amd import ops
R = int(os.environ.get("ROWS", "76410"))
shapes = [(1024, 1063, 0), (1024, 384, 1), (384, 1024, 0), (512, 768, 0), (768, 512, 0), (256, 256, 0), (256, 384, 0), (384, 256, 0),
          (512, 39, 0), (128, 128, 0)]
def t_once(fn, iters=10):
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters
for (C, K, act) in shapes:
    Kp = (K + 3) // 4 * 4
    x = torch.randn(R, Kp, device="cuda")[:, :K]
    w = (torch.randn(C, Kp, device="cuda") * 0.05)[:, :K]
    sx, sw = ops.pow2_scale(x), ops.pow2_scale(w)
    wps = ops.presplit(w, scale=sw)[0]
    bias = torch.randn(C, device="cuda") if act else None
    out = torch.empty(R, C, device="cuda")
    res = {}
    for tile in (0, 2, 4, 5):
        fn = lambda: ops.gemm_nt(x, wps, sx, sw, out=out, bias=bias, act=ops.ACT_ELU if act else 0, tile=tile, b_presplit=True)
        fn(); fn()
        res[tile] = []
    torch.cuda.synchronize()
    for _ in range(7):
        for tile in res:
            res[tile].append(t_once(lambda: ops.gemm_nt(x, wps, sx, sw, out=out, bias=bias, act=ops.ACT_ELU if act else 0, tile=tile, b_presplit=True)))
    fl = 2.0 * R * C * K
    print(f"nt N={C} K={K} act={act}: " + "  ".join(f"tile{t} {sorted(v)[3]*1e3:.0f}us({fl/sorted(v)[3]/1e9:.0f}TF)" for t, v in res.items()), flush=True)
