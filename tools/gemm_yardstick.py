"""The step's projection products beside the vendor library's products of the same operands (a yardstick for what these
tall-and-shallow shapes - 76 410 rows, K and N of a few hundred to a thousand - reach on this chip; the library is never on
the product path).  Per shape: the shipped kernel (fp32 in / out, three fp16 MFMA products per fp32 product), torch.mm in fp32
(the library's fp32 path) and torch.mm on operands cast to fp16 beforehand, fp16 out (ONE MFMA product, half the bytes).
usage: gemm_yardstick.py [nt|tn] [C,K ...] > profiles/...json"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spgnn_amd import ops

kind = sys.argv[1] if len(sys.argv) > 1 else "nt"
shapes = [tuple(int(x) for x in s.split(",")) for s in sys.argv[2:]] or [(1024, 1063), (1024, 384), (384, 1024), (512, 768), (768, 512), (384, 256), (256, 384)]
R = 76410


def t_med(fn, iters=10, rounds=7):
    ts = []
    for _ in range(rounds):
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters): fn()
        b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / iters)
    return sorted(ts)[len(ts) // 2] * 1e3            # us


rows = []
for (C, K) in shapes:
    Kp = (K + 7) // 8 * 8
    x = torch.randn(R, Kp, device="cuda")[:, :K]
    w = (torch.randn(C, Kp, device="cuda") * 0.05)[:, :K]
    g = torch.randn(R, C, device="cuda") * 1e-3
    sx, sw, sg = ops.pow2_scale(x), ops.pow2_scale(w), ops.pow2_scale(g)
    x16, w16, g16 = x.half(), w.half(), g.half()
    xc, wc = x.contiguous(), w.contiguous()
    if kind == "nt":
        out = torch.empty(R, C, device="cuda")
        mine = lambda: ops.gemm_nt(x, w, sx, sw, out=out)
        lib32 = lambda: torch.mm(xc, wc.t())
        lib16 = lambda: torch.mm(x16, w16.t())
    else:
        mine = lambda: ops.gemm_tn(g, x, sg, sx)
        lib32 = lambda: torch.mm(g.t(), xc)
        lib16 = lambda: torch.mm(g16.t(), x16)
    for f in (mine, lib32, lib16): f(); f()
    torch.cuda.synchronize()
    tm, t32, t16 = t_med(mine), t_med(lib32), t_med(lib16)
    fl = 2.0 * R * C * K
    rows.append({"kind": kind, "rows": R, "C": C, "K": K, "shipped_us": round(tm, 1), "library_fp32_us": round(t32, 1), "library_fp16_us": round(t16, 1),
                 "shipped_executed_mfma_TFLOPs": round(3 * fl / tm / 1e6, 0), "shipped_algorithmic_TFLOPs": round(fl / tm / 1e6, 0),
                 "library_fp32_TFLOPs": round(fl / t32 / 1e6, 0), "library_fp16_TFLOPs": round(fl / t16 / 1e6, 0)})
    print(rows[-1], file=sys.stderr, flush=True)
print(json.dumps({"note": "library = torch.mm (hipBLASLt / rocBLAS as this image ships them); the shipped kernel runs THREE fp16 MFMA products per "
                          "fp32 product and reads / writes fp32 rows: its executed rate is the one to hold against the library's fp16 rate",
                  "products": rows}, indent=1))
