import glob, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spgnn_amd import _capi, ops
paths = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "..", "build", "variants", "t_*.so")))
libs = {}
for p in paths:
    _capi._lib = None; _capi.LIB_PATH = p
    libs[os.path.basename(p)] = _capi.load()
dev = "cuda"; M = 76410
def t_once(fn, iters=6):
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters
for (K, N) in [(1063, 1024), (1024, 1063), (768, 512), (384, 1024), (384, 256), (512, 768), (1024, 384), (256, 384)]:
    Kp4 = (K + 3) // 4 * 4
    x = torch.randn(M, Kp4, device=dev)[:, :K]; w = (torch.randn(N, Kp4, device=dev) * 0.05)[:, :K]
    sx, sw = ops.pow2_scale(x), ops.pow2_scale(w)
    out = torch.empty(M, N, device=dev)
    fn = lambda: ops.gemm_nt(x, w, sx, sw, out=out)
    res = {}; outs = {}
    for k, lib in libs.items():
        _capi._lib = lib; fn(); outs[k] = out.clone()
    torch.cuda.synchronize()
    for r in range(5):
        for k, lib in libs.items():
            _capi._lib = lib; res.setdefault(k, []).append(t_once(fn))
    ks = list(libs); fl = 2.0 * M * N * K
    ref = (x.double() @ w.double().t())
    print(f"K={K} N={N}: " + " | ".join(f"{k[2:-3]} {sorted(v)[2]*1e3:.0f}us ({fl/sorted(v)[2]/1e9:.0f} TF) err {float((outs[k]-ref).abs().max()/ref.abs().max()):.1e}" for k, v in res.items()), flush=True)
