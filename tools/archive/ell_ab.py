"""The three GAT kernels with and without the padded (N, 8) neighbour rows (ops.USE_ELL), one process, interleaved rounds."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spgnn_amd import ops, synthetic

dev = torch.device("cuda")
g = synthetic.make_batch(int(sys.argv[1]) if len(sys.argv) > 1 else 512, pos_enc_dim=None, fv_dim=8).to(dev)
csc = g.csc(); N, E = csc.num_nodes, csc.num_edges
csc.ell()

def t_once(fn, iters=10):
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3

tot = {False: 0.0, True: 0.0}
for (H, D, mean) in [(2, 1024, True), (2, 256, False), (2, 128, False), (2, 64, False), (1, 256, False), (1, 128, False), (1, 64, False)]:
    HD = H * D
    y = torch.randn(N, 2 * HD, device=dev); s = torch.randn(N, 2 * H, device=dev); bias = torch.zeros(HD, device=dev)
    g_out = torch.randn(N, D if mean else HD, device=dev); g_y = torch.empty_like(y); g_s = torch.empty_like(s)
    out, om, attn = ops.gat_fwd_raw(csc, y[:, :HD], s[:, :H], s[:, H:], y[:, HD:], bias, H, D, 0.2, ops.ACT_ELU, mean=mean)
    fwd = lambda: ops.gat_fwd_raw(csc, y[:, :HD], s[:, :H], s[:, H:], y[:, HD:], bias, H, D, 0.2, ops.ACT_ELU, out=out, mean=mean)
    bwd = lambda: ops.gat_bwd_raw(csc, y[:, :HD], s[:, :H], s[:, H:], attn, g_out, out, H, D, 0.2, ops.ACT_ELU, 0.0, 0,
                                  g_y[:, HD:], g_y[:, :HD], g_s[:, :H], g_s[:, H:], mean=mean)
    for name, fn in (("fwd", fwd), ("bwd", bwd)):
        r = {False: [], True: []}
        for _ in range(7):
            for flag in (False, True):
                ops.USE_ELL = flag
                fn(); r[flag].append(t_once(fn))
        med = {k: sorted(v)[len(v) // 2] for k, v in r.items()}
        for k in med: tot[k] += med[k]
        print(f"{H}x{D} {name}: csc {med[False]:.1f} us  ell {med[True]:.1f} us  ({med[True] / med[False]:.3f})", flush=True)
    del y, s, g_out, g_y, g_s, out, attn
print("total us: csc %.1f ell %.1f" % (tot[False], tot[True]))
