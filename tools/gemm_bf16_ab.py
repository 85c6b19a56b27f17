"""A/B of bf16 GEMM kernel variants (build/variants/*.so) in ONE process, interleaved rounds, medians.
usage: gemm_bf16_ab.py [nt|tn] [N,K ...]   (nt: C[M,N] = A[M,K] B[N,K]^T, M = 76410 bf16 rows; tn: C[N,K] = G[R,N]^T X[R,K])"""
import glob, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spgnn_amd import _capi, ops_bf16

kind = sys.argv[1] if len(sys.argv) > 1 else "nt"
shapes = [tuple(int(x) for x in s.split(",")) for s in sys.argv[2:]] or [(512, 1024), (256, 512), (256, 256), (256, 128), (1024, 384)]
paths = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "..", "build", "variants", "*.so")))
libs = {}
for p in paths:
    _capi._lib = None; _capi.LIB_PATH = p
    libs[os.path.basename(p)[:-3]] = _capi.load()
R = 76410


def t_once(fn, iters=10):
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


for (C, K) in shapes:
    x = ops_bf16.cast_rows(torch.randn(R, K, device="cuda"))
    w = ops_bf16.cast_rows(torch.randn(C, K, device="cuda") * 0.05)
    g = ops_bf16.cast_rows(torch.randn(R, C, device="cuda") * 1e-3)
    fn = (lambda: ops_bf16.gemm_nt(x, w)) if kind == "nt" else (lambda: ops_bf16.gemm_tn(g, x))
    res = {n: [] for n in libs}
    outs = {}
    for n, lib in libs.items():
        _capi._lib = lib
        fn(); outs[n] = fn()
    torch.cuda.synchronize()
    first = next(iter(outs.values()))
    same = all(torch.equal(first if not isinstance(first, tuple) else first[0], o if not isinstance(o, tuple) else o[0]) for o in outs.values())
    for _ in range(7):
        for n, lib in libs.items():
            _capi._lib = lib
            res[n].append(t_once(fn))
    fl = 2.0 * R * C * K
    if kind == "nt":                                   # yardstick only (the library's product of the same operands; never shipped)
        xt, wt = x[:, :K], w[:, :K]
        lib = sorted(t_once(lambda: torch.mm(xt, wt.t())) for _ in range(7))[3]
        print(f"   torch.mm (hipBLASLt/rocBLAS) {lib*1e3:.0f} us ({fl/lib/1e9:.0f} TF)", flush=True)
    else:
        gt, xt = g[:, :C], x[:, :K]
        lib = sorted(t_once(lambda: torch.mm(gt.t(), xt)) for _ in range(7))[3]
        print(f"   torch.mm (hipBLASLt/rocBLAS, bf16 out) {lib*1e3:.0f} us ({fl/lib/1e9:.0f} TF)", flush=True)
    print(f"{kind} C={C} K={K} identical={same}: " + "  ".join(f"{n} {sorted(v)[len(v)//2]*1e3:.0f} us ({fl/sorted(v)[len(v)//2]/1e9:.0f} TF)" for n, v in res.items()), flush=True)
