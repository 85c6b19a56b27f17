"""Does an MFMA-bound weight-gradient product overlap with HBM-bound traversal kernels when they are issued on two streams?
One process: A = spgnn_gemm_tn (76 410 x 1024)^T (76 410 x 1064); B = the three GAT traversals at 2 x 256 fp32, three times.
Prints the time of A, of B, of A then B on one stream, and of A || B on two streams (medians of 9)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spgnn_amd import ops, synthetic
dev = torch.device("cuda")
g = synthetic.make_batch(512, pos_enc_dim=None, fv_dim=8).to(dev)
csc = g.csc(); N, E = csc.num_nodes, csc.num_edges
H, D = 2, 256; HD = H * D
gy = torch.randn(N, 1024, device=dev); x = torch.randn(N, 1064, device=dev)
y = torch.randn(N, 2 * HD, device=dev); s = torch.randn(N, 2 * H, device=dev); bias = torch.zeros(HD, device=dev)
g_out = torch.randn(N, HD, device=dev); g_y = torch.empty_like(y); g_s = torch.empty_like(s)
out, _, attn = ops.gat_fwd_raw(csc, y[:, :HD], s[:, :H], s[:, H:], y[:, HD:], bias, H, D, 0.2, ops.ACT_ELU, 0.1, 7)

def A():
    ops.gemm_tn(gy, x)
def B():
    for _ in range(3):
        ops.gat_fwd_raw(csc, y[:, :HD], s[:, :H], s[:, H:], y[:, HD:], bias, H, D, 0.2, ops.ACT_ELU, 0.1, 7, out=out)
        ops.gat_bwd_raw(csc, y[:, :HD], s[:, :H], s[:, H:], attn, g_out, out, H, D, 0.2, ops.ACT_ELU, 0.1, 7, g_y[:, HD:], g_y[:, :HD], g_s[:, :H], g_s[:, H:])
side = torch.cuda.Stream()
def timed(fn, n=5):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
def both_seq(): A(); B()
def both_par():
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side): A()
    B()
    main.wait_stream(side)
for f in (A, B, both_seq, both_par): f()
res = {k: [] for k in ("A", "B", "seq", "par")}
for r in range(9):
    res["A"].append(timed(A)); res["B"].append(timed(B)); res["seq"].append(timed(both_seq)); res["par"].append(timed(both_par))
print({k: round(sorted(v)[len(v) // 2], 1) for k, v in res.items()}, "us")
