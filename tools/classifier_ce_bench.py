"""spgnn_classifier_ce against the three launches it replaces (scores_fwd + masked_ce_step + scores_bwd_w [+ the partial sums]),
each alone on the stream, HIP events, medians.  usage: classifier_ce_bench.py [N ...]"""
import os, sys, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spgnn_amd import ops

def timed(fn, n=30):
    for _ in range(5): fn()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    return statistics.median(ts)

for N in [int(a) for a in sys.argv[1:]] or [76410, 9859]:
    K, J = 1024, 22
    g = torch.Generator().manual_seed(0)
    x = torch.randn(N, K, generator=g).cuda(); w = (torch.randn(J, K, generator=g) / 32).cuda(); b = torch.zeros(J).cuda()
    y = torch.randint(0, J, (N,), generator=g).cuda(); p = torch.full((N,), 0.3).cuda(); cw = torch.ones(J).cuda()
    draws = torch.rand(N, generator=g).cuda(); sums = torch.zeros(2).cuda()
    res = {}
    res["classifier_ce"] = timed(lambda: ops.classifier_ce(x, w, b, ops.LossHead(y, p, draws, 0, cw, sums)))
    out = ops.classifier_ce(x, w, b, ops.LossHead(y, p, draws, 0, cw, sums))
    res["  + sum of its partials"] = timed(lambda: ops.sum_partials(out[2]))
    res["scores_fwd"] = timed(lambda: ops.scores_fwd(x, w, bias=b))
    lg = ops.scores_fwd(x, w, bias=b).requires_grad_(True)
    res["masked_ce"] = timed(lambda: ops.masked_ce_sums(lg, y, draws, p, cw, unit_grad=True))
    gl = out[1]
    res["scores_bwd_w (+ its sum)"] = timed(lambda: ops.scores_bwd_w(gl, x))
    print(f"N={N} K={K} J={J}: " + " | ".join(f"{k} {v:.1f} us" for k, v in res.items()), flush=True)
