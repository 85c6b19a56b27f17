"""what a stream-K tail could at most gain.  Every NT launch of the st_pgat_spgnn_3 step at 512 trees
(M = 76 410 rows) is timed at its real M and over a sweep of neighbouring row counts (whole 256-row tiles, 280 .. 320 tiles);
the cheapest time PER ROW seen in the sweep, scaled to 76 410 rows, is what a launch without any round quantisation would
take - a bound no K-split of the last partial round can beat, since it adds a fix-up pass on top.  Pair launches are timed as
pairs (ops.gemm_nt_pair), as the step issues them.  One process, interleaved rounds, medians.
usage: python3 tools/gemm_round_quantisation.py > profiles/r06_gemm_streamk_ab.txt"""
import os, sys, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spgnn_amd import ops

M0 = int(os.environ.get("ROWS", "76410"))
# (launches per step, [(C, K), ...]) - a pair launch = two products in one grid (bench_detail.json: gemm.per_shape)
LAUNCHES = [(2, [(1024, 384)]), (1, [(1024, 1063), (512, 39)]), (1, [(384, 1024), (384, 1024)]), (1, [(512, 768), (256, 256)]),
            (1, [(768, 512), (256, 256)]), (1, [(384, 256), (128, 128)]), (1, [(256, 384), (128, 128)])]
MMAX = 256 * 321


def operands(C, K):
    Kp = (K + 3) // 4 * 4
    x = torch.randn(MMAX, Kp, device="cuda")[:, :K]
    w = (torch.randn(C, Kp, device="cuda") * 0.05)[:, :K]
    sx, sw = ops.pow2_scale(x), ops.pow2_scale(w)
    return x, ops.presplit(w, scale=sw)[0], sx, sw, torch.empty(MMAX, C, device="cuda")


def timed(fn, iters=8):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


total_now = total_ideal = 0.0
for n_launch, probs in LAUNCHES:
    ops_ = [operands(C, K) for C, K in probs]

    def run(M):
        if len(ops_) == 1:
            x, wps, sx, sw, out = ops_[0]
            ops.gemm_nt(x[:M], wps, sx, sw, out=out[:M], b_presplit=True)
        else:
            ps = [ops.NtProblem(x[:M], wps, sx, sw, out=out[:M], b_presplit=True) for x, wps, sx, sw, out in ops_]
            ops.gemm_nt_pair(ps[0], ps[1])
    sweep = [M0] + [256 * k for k in range(280, 321, 2)]
    for M in sweep:
        run(M)
    torch.cuda.synchronize()
    res = {M: [] for M in sweep}
    for _ in range(5):
        for M in sweep:
            res[M].append(timed(lambda: run(M)))
    med = {M: statistics.median(v) for M, v in res.items()}
    per_row = min(med[M] / M for M in sweep if M != M0)
    best_M = min((M for M in sweep if M != M0), key=lambda M: med[M] / M)
    ideal = per_row * M0
    total_now += n_launch * med[M0]
    total_ideal += n_launch * min(ideal, med[M0])
    name = " + ".join(f"{C}x{K}" for C, K in probs)
    print(f"{n_launch} x NT {name:28s} at M = {M0}: {med[M0]:7.1f} us; cheapest per row in the sweep at M = {best_M} ({med[best_M]:.1f} us) "
          f"-> {ideal:7.1f} us without quantisation; at most {med[M0] - ideal:6.1f} us per launch to gain", flush=True)
print(f"\nall NT launches of a step: {total_now:.0f} us now, {total_ideal:.0f} us with every launch at its sweep's best time per row: "
      f"an upper bound of {total_now - total_ideal:.0f} us per step for ANY treatment of the last partial round (stream-K adds its fix-up "
      f"pass on top).  Adoption threshold (VERDICT r5 item 8): 80 us.")
