import glob, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spgnn_amd import _capi, ops
paths = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "..", "build", "variants", "g_ablate*.so")))
libs = {}
for p in paths:
    _capi._lib = None; _capi.LIB_PATH = p
    libs[os.path.basename(p)] = _capi.load()
dev = "cuda"; N = 76410
def t_once(fn, iters=8):
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters
for (K, C) in [(1063, 1024), (768, 512), (192, 4096)]:
    Kp = (K + 3) // 4 * 4
    x = torch.randn(N, Kp, device=dev)[:, :K]; w = (torch.randn(C, Kp, device=dev) * 0.05)[:, :K]
    out = torch.empty(N, C, device=dev)
    fn = lambda: ops.gemm_nt(x, w, None, None, out=out)
    res = {}
    for k, lib in libs.items():
        _capi._lib = lib; fn()
    torch.cuda.synchronize()
    for r in range(5):
        for k, lib in libs.items():
            _capi._lib = lib; res.setdefault(k, []).append(t_once(fn))
    print(f"K={K} C={C}: " + " | ".join(f"{k[8:-3]} {sorted(v)[2]*1e3:.0f}us" for k, v in res.items()), flush=True)
