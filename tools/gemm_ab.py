"""A/B of GEMM kernel variants (build/variants/*.so) in ONE process, interleaved rounds, median.
usage: gemm_ab.py [nt|tn] [M,N,K ...]   (nt: C[M,N] = A[M,K] B[N,K]^T with M = 76410 rows; tn: R = 76410)"""
import glob, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spgnn_amd import _capi, ops

kind = sys.argv[1] if len(sys.argv) > 1 else "nt"
shapes = [tuple(int(x) for x in s.split(",")) for s in sys.argv[2:]] or [(1024, 1063), (1024, 384), (384, 1024), (512, 768)]
paths = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "..", "build", "variants", "*.so")))
libs = {}
for p in paths:
    _capi._lib = None; _capi.LIB_PATH = p
    libs[os.path.basename(p)[:-3]] = _capi.load()
dev = "cuda"
R = 76410

def t_once(fn, iters=10):
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters

for (C, K) in shapes:
    Kp = (K + 3) // 4 * 4
    x = torch.randn(R, Kp, device=dev)[:, :K]
    w = (torch.randn(C, Kp, device=dev) * 0.05)[:, :K]
    g = torch.randn(R, C, device=dev) * 1e-3
    sx, sw, sg = ops.pow2_scale(x), ops.pow2_scale(w), ops.pow2_scale(g)
    out = torch.empty(R, C, device=dev)
    fn = (lambda: ops.gemm_nt(x, w, sx, sw, out=out)) if kind == "nt" else (lambda: ops.gemm_tn(g, x, sg, sx))
    res = {n: [] for n in libs}
    for n, lib in libs.items():
        _capi._lib = lib
        fn(); fn()
    torch.cuda.synchronize()
    for _ in range(7):
        for n, lib in libs.items():
            _capi._lib = lib
            res[n].append(t_once(fn))
    fl = 2.0 * R * C * K
    print(f"{kind} C={C} K={K}: " + "  ".join(f"{n} {sorted(v)[len(v)//2]*1e3:.0f} us ({3*fl/sorted(v)[len(v)//2]/1e9:.0f} TF exec)" for n, v in res.items()), flush=True)
