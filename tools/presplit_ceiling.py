"""VERDICT r4 item 3(a), its ceiling without the DMA ring: every NT product of one st_pgat_spgnn_3 step at 512 trees with its
A operand (the activation) as fp32 rows (split in the K loop, as shipped) against the same product with A PRE-SPLIT by a
separate, uncounted pass (the weights are pre-split either way).  One process, interleaved, medians of 7 x 10 launches.
The TN products likewise with their B operand (the layer input) pre-split.  -> what producer-emitted pre-split operands
could save per step at most, before paying for their production."""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spgnn_amd import ops
M = 76410
dev = torch.device("cuda")
def t_once(fn, iters=10):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
def med(fns):
    for f in fns.values(): f()
    torch.cuda.synchronize()
    r = {k: [] for k in fns}
    for _ in range(7):
        for k, f in fns.items(): r[k].append(t_once(f))
    return {k: sorted(v)[len(v) // 2] for k, v in r.items()}
rows = []
# (N, K, launches per step): forward and input-gradient products whose A operand is an ACTIVATION (level 0's is node data: done)
nt_shapes = [(512, 768, 1), (256, 256, 1), (256, 384, 1), (128, 128, 1), (1024, 384, 2), (384, 1024, 2), (768, 512, 1), (256, 256, 1), (384, 256, 1), (128, 128, 1)]
tot = {"fp32_A": 0.0, "presplit_A": 0.0}
for (N, K, n) in nt_shapes:
    Kp = (K + 3) // 4 * 4
    a = torch.randn(M, Kp, device=dev)[:, :K]; w = (torch.randn(N, Kp, device=dev) * 0.1)[:, :K]
    sa, sw = ops.pow2_scale(a), ops.pow2_scale(w)
    a_ps = ops.presplit(a, scale=sa)[0]; w_ps = ops.presplit(w, scale=sw)[0]
    out = torch.empty(M, N, device=dev)
    m = med({"fp32_A": lambda: ops.gemm_nt(a, w_ps, sa, sw, out=out, b_presplit=True),
             "presplit_A": lambda: ops.gemm_nt(a_ps, w_ps, sa, sw, out=out, b_presplit=True, a_presplit=True)})
    rows.append({"kind": "nt", "N": N, "K": K, "per_step": n, **{k: round(v, 1) for k, v in m.items()}})
    for k in tot: tot[k] += n * m[k]
    print(rows[-1], flush=True)
    del a, w, a_ps, w_ps, out
# weight gradients (R = M rows): A = g_Y (Mo columns), B = the layer input X (K columns); B pre-split vs fp32
tn_shapes = [(1024, 1063), (512, 768), (256, 384), (1024, 384), (1024, 384)]
tot_tn = {"fp32_B": 0.0, "presplit_B": 0.0}
for (Mo, K) in tn_shapes:
    Kp = (K + 3) // 4 * 4
    g = torch.randn(M, Mo, device=dev); x = torch.randn(M, Kp, device=dev)[:, :K]
    sg, sx = ops.pow2_scale(g), ops.pow2_scale(x)
    x_ps = ops.presplit(x, scale=sx)[0]
    m = med({"fp32_B": lambda: ops.gemm_tn(g, x, sg, sx), "presplit_B": lambda: ops.gemm_tn(g, x_ps, sg, sx, b_presplit=True)})
    rows.append({"kind": "tn", "M": Mo, "K": K, **{k: round(v, 1) for k, v in m.items()}})
    for k in tot_tn: tot_tn[k] += m[k]
    print(rows[-1], flush=True)
    del g, x, x_ps
print(json.dumps({"rows": rows, "nt_us_per_step": {k: round(v, 1) for k, v in tot.items()}, "tn_us_per_step": {k: round(v, 1) for k, v in tot_tn.items()}}))
