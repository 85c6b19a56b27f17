import glob, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spgnn_amd import _capi, ops
paths = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "..", "build", "variants", "s_*.so")))
libs = {}
for p in paths:
    _capi._lib = None; _capi.LIB_PATH = p
    libs[os.path.basename(p)] = _capi.load()
dev = "cuda"; N = 76410
def t_once(fn, iters=10):
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters
for (K, J) in [(1024, 22), (192, 4), (384, 22), (1063, 4), (256, 2)]:
    Kp4 = (K + 3) // 4 * 4
    x = torch.randn(N, Kp4, device=dev)[:, :K]; w = torch.randn(J, K, device=dev); gs = torch.randn(N, J, device=dev)
    gx = torch.empty(N, Kp4, device=dev)[:, :K]
    fns = {"fwd": lambda: ops.scores_fwd(x, w, want_scale=(J <= 16))}
    if os.environ.get("ALL"):
        fns.update({"bwd_w": lambda: ops.scores_bwd_w(gs, x), "bwd_x": lambda: ops.scores_bwd_x_(gx, gs, w, accumulate=False)})
    for name, fn in fns.items():
        res = {}
        for k, lib in libs.items():
            _capi._lib = lib; fn()
        torch.cuda.synchronize()
        for r in range(5):
            for k, lib in libs.items():
                _capi._lib = lib; res.setdefault(k, []).append(t_once(fn))
        print(f"K={K} J={J} {name}: " + " | ".join(f"{k[2:-3]} {sorted(v)[2]*1e3:.0f}us" for k, v in res.items()), flush=True)
