import glob, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spgnn_amd import _capi, ops
paths = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "..", "build", "variants", "t_*.so")))
libs = {}
for p in paths:
    _capi._lib = None; _capi.LIB_PATH = p
    libs[os.path.basename(p)] = _capi.load()
dev = "cuda"; R = 76410
def t_once(fn, iters=6):
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters
for (M, N) in [(1024, 1064), (512, 768), (1024, 384), (256, 384), (512, 256)]:
    g = torch.randn(R, M, device=dev); x = torch.randn(R, N, device=dev)
    sg, sx = ops.pow2_scale(g), ops.pow2_scale(x)
    fn = lambda: ops.gemm_tn(g, x, sg, sx, want_colsum=True)
    res = {}; outs = {}
    for k, lib in libs.items():
        _capi._lib = lib; outs[k] = fn()[0].clone()
    torch.cuda.synchronize()
    for r in range(5):
        for k, lib in libs.items():
            _capi._lib = lib; res.setdefault(k, []).append(t_once(fn))
    ks = list(libs)
    fl = 2.0 * R * M * N
    print(f"M={M} N={N}: " + " | ".join(f"{k[2:-3]} {sorted(v)[2]*1e3:.0f}us ({fl/sorted(v)[2]/1e9:.0f} TF)" for k, v in res.items()),
          "maxrel", max(float((outs[ks[0]] - outs[k]).abs().max() / outs[ks[0]].abs().max()) for k in ks), flush=True)
