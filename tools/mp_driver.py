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
# aggregate-first output layer (192 -> 2 x 1024): its three kernels, the derivative pass and the head mean
H, D, F_ = 2, 1024, 192
x = torch.randn(N, F_, device=dev); s = torch.randn(N, 2 * H, device=dev)
out = torch.randn(N, H * D, device=dev); g_mean = torch.randn(N, D, device=dev)
w_lr = torch.randn(2 * H, F_, device=dev)
g_logits = torch.randn(N, 22, device=dev); w_cls = torch.randn(22, D, device=dev) * 0.03
from spgnn_amd import _capi
lib = _capi.load()
st = torch.cuda.current_stream().cuda_stream
ops.KernelTimer.start()
for _ in range(reps):
    z, attn, amax = ops.gat_agg_fwd_raw(csc, x, s[:, :H], s[:, H:], H, 0.2, 0.0, 0, True)
    om = ops.head_mean(out, H, D)
    g_pre, amax_g = ops.act_bwd(g_mean, out, H, D, ops.ACT_ELU, True)
    g_pre2, amax_g2 = ops.act_bwd_proj(g_logits, w_cls, out, H, D, ops.ACT_ELU)
    g_z = torch.randn_like(z); g_e = torch.empty(E, H, device=dev); g_s = torch.empty_like(s); g_x = torch.empty_like(x)
    with ops._timed("gat_agg_bwd_dst", (N, E, H, F_)):
        _capi.check(lib.spgnn_gat_agg_bwd_dst(csc.indptr.data_ptr(), csc.indices.data_ptr(), x.data_ptr(), x.stride(0), s.data_ptr(),
            s[:, H:].data_ptr(), s.stride(0), attn.data_ptr(), g_z.data_ptr(), g_z.stride(0), 2 * F_, g_e.data_ptr(),
            g_s[:, H:].data_ptr(), g_s.stride(0), N, E, H, F_, 0.2, 0.0, 0, 0, st), "dst")
    with ops._timed("gat_agg_bwd_src", (N, E, H, F_)):
        _capi.check(lib.spgnn_gat_agg_bwd_src(csc.out_indptr.data_ptr(), csc.out_indices.data_ptr(), csc.out_pos.data_ptr(),
            attn.data_ptr(), g_e.data_ptr(), g_z.data_ptr(), g_z.stride(0), 2 * F_, F_, g_s[:, H:].data_ptr(), w_lr.data_ptr(),
            w_lr.stride(0), g_x.data_ptr(), g_x.stride(0), g_s.data_ptr(), g_s.stride(0), N, E, H, F_, 0.0, 0, 0, st), "src")
keys = {k[0]: "_".join(str(v) for v in k) for k in ops.KernelTimer.stop()}
manifest += [["gat_agg_fwd", keys["gat_agg_fwd"]], ["head_mean", keys["head_mean"]], ["act_bwd", keys["act_bwd"]],
             ["act_bwd_proj", keys["act_bwd_proj"]],
             ["gat_agg_bwd_dst", keys["gat_agg_bwd_dst"]], ["gat_agg_bwd_src", keys["gat_agg_bwd_src"]]] * reps
import json
os.makedirs("gpurun_out", exist_ok=True)
json.dump(manifest, open("gpurun_out/mp_manifest.json", "w"))
print("done", N, E)
