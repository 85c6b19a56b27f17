"""Launch only the message-passing kernels at the st_pgat_spgnn_3 layer shapes (for rocprofv3 --pmc passes)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spgnn_amd import ops, synthetic

trees = int(sys.argv[1]) if len(sys.argv) > 1 else 512
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda")
manifest = []                            # launch order of the three GAT kernels with bench.py's (kernel, shape) keys
g = synthetic.make_batch(trees, pos_enc_dim=None, fv_dim=8).to(dev)
csc = g.csc()
N, E = csc.num_nodes, csc.num_edges
for (H, D) in [(2, 1024), (2, 256), (2, 128), (2, 64), (1, 256), (1, 128), (1, 64)]:
    HD = H * D
    y = torch.randn(N, 2 * HD, device=dev); s = torch.randn(N, 2 * H, device=dev); bias = torch.zeros(HD, device=dev)
    g_out = torch.randn(N, HD, device=dev); g_y = torch.empty_like(y); g_s = torch.empty_like(s)
    mean = D == 1024                      # the output layer runs with the fused head mean (as in the model)
    if mean:
        g_out = torch.randn(N, D, device=dev)
    amax = torch.empty(2 * N, device=dev)
    ops.KernelTimer.start()
    for _ in range(reps):
        out, _, attn = ops.gat_fwd_raw(csc, y[:, :HD], s[:, :H], s[:, H:], y[:, HD:], bias, H, D, 0.2, ops.ACT_ELU, mean=mean)
        ops.gat_bwd_raw(csc, y[:, :HD], s[:, :H], s[:, H:], attn, g_out, out, H, D, 0.2, ops.ACT_ELU, 0.0, 0,
                        g_y[:, HD:], g_y[:, :HD], g_s[:, :H], g_s[:, H:], mean=mean, absmax=amax)
    keys = {k[0]: "_".join(str(x) for x in k) for k in ops.KernelTimer.stop()}
    manifest += [["gat_fwd", keys["gat_fwd"]], ["gat_bwd_dst", keys["gat_bwd_dst"]], ["gat_bwd_src", keys["gat_bwd_src"]]] * reps
    del y, s, g_out, g_y, g_s, out, attn
import json
os.makedirs("gpurun_out", exist_ok=True)
json.dump(manifest, open("gpurun_out/mp_manifest.json", "w"))
print("done", N, E)
