"""What would a backward level's input-gradient (NT) and weight-gradient (TN) products gain from sharing ONE launch at 64 trees
(9 641 rows: every launch is a round or two of tiles and its ramp and tail are a large part of it)?  Upper bound without writing the
kernel: A = NT product, B = TN product of a level; 10 x (A then B) on one stream against 10 x A on one stream beside 10 x B on another
(no joins in between).  usage: quad_probe.py [rows]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spgnn_amd import ops
R = int(sys.argv[1]) if len(sys.argv) > 1 else 9641
dev = "cuda"
side = torch.cuda.Stream()
for (C, K) in [(1024, 384), (768, 512), (384, 256), (1024, 1063)]:
    Kp = (K + 7) // 8 * 8
    g = torch.randn(R, C, device=dev) * 1e-3
    x = torch.randn(R, Kp, device=dev)[:, :K]
    w = (torch.randn(K, C, device=dev) * 0.05)                 # g_x = g @ w^T: (R, C) x (K, C)^T
    sg, sx, sw = ops.pow2_scale(g), ops.pow2_scale(x), ops.pow2_scale(w)
    gx = torch.empty(R, K, device=dev)
    A = lambda: ops.gemm_nt(g, w, sg, sw, out=gx)
    B = lambda: ops.gemm_tn(g, x, sg, sx)
    for _ in range(3): A(); B()
    torch.cuda.synchronize()

    def timed(fn):
        ts = []
        for _ in range(7):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); a.record(); fn(); b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) * 1e3 / 10)
        return sorted(ts)[3]

    def serial():
        for _ in range(10): A(); B()

    def only_a():
        for _ in range(10): A()

    def only_b():
        for _ in range(10): B()

    def beside():
        cur = torch.cuda.current_stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            for _ in range(10): B()
        for _ in range(10): A()
        cur.wait_stream(side)

    # captured, so that the host does not pace the launches
    res = {}
    for name, fn in (("A", only_a), ("B", only_b), ("A;B", serial), ("A||B", beside)):
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            fn()
        res[name] = timed(gr.replay)
    print(f"rows {R} g({C}) x({K}):  NT {res['A']:.1f} us  TN {res['B']:.1f} us  one stream {res['A;B']:.1f} us  two streams {res['A||B']:.1f} us", flush=True)
