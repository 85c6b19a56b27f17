"""Bitwise repeatability of the three GAT kernels under GPU contention: a child process keeps the GPU busy while the parent
re-runs every kernel on fixed inputs and compares every output with the first run.  usage: stress_determinism.py [trees] [reps]"""
import os, sys, subprocess, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "--hog":
    import torch
    a = torch.randn(4096, 4096, device="cuda"); b = torch.randn(4096, 4096, device="cuda")
    t0 = time.time()
    while time.time() - t0 < float(sys.argv[2]):
        for _ in range(20):
            c = a @ b
            d = torch.relu(c) + 1
        torch.cuda.synchronize()
    sys.exit(0)
import torch
from spgnn_amd import _capi, ops, synthetic
if os.environ.get("LIBV"):
    _capi.LIB_PATH = os.environ["LIBV"]
trees = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = torch.device("cuda")
g = synthetic.make_batch(trees, pos_enc_dim=None, fv_dim=8).to(dev)
csc = g.csc(); N, E = csc.num_nodes, csc.num_edges
hog = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--hog", "60"])
time.sleep(8)
P = float(os.environ.get("P", "0.1")); ACT = int(os.environ.get("ACT", "1"))
bad = 0
for (H, D, mean) in [(2, 1024, True), (2, 256, False), (2, 128, False), (2, 64, False), (1, 256, False), (1, 128, False), (1, 64, False)]:
    HD = H * D
    torch.manual_seed(H * D)
    y = torch.randn(N, 2 * HD, device=dev); s = torch.randn(N, 2 * H, device=dev); bias = torch.randn(HD, device=dev)
    g_out = torch.randn(N, D if mean else HD, device=dev)
    ref = None
    for r in range(reps):
        g_y = torch.zeros_like(y); g_s = torch.zeros_like(s)
        out, om, attn = ops.gat_fwd_raw(csc, y[:, :HD], s[:, :H], s[:, H:], y[:, HD:], bias, H, D, 0.2, ACT, P, 7, mean=mean)
        amax = torch.zeros(2 * N, device=dev)
        al_, ar_ = torch.randn(HD, device=dev), torch.randn(HD, device=dev)
        g_e = ops.gat_bwd_raw(csc, y[:, :HD], s[:, :H], s[:, H:], attn, g_out, out, H, D, 0.2, ACT, P, 7,
                              g_y[:, HD:], g_y[:, :HD], g_s[:, :H], g_s[:, H:], mean=mean, absmax=amax,
                              score_l=None if mean else al_ * 0 + 0.5, score_r=None if mean else ar_ * 0 + 0.25)
        cur = dict(amax=amax, out=out, attn=attn, g_pre=g_y[:, HD:].clone(), g_ft=g_y[:, :HD].clone(), g_e=g_e, g_el=g_s[:, :H].clone(), g_er=g_s[:, H:].clone())
        if om is not None: cur["mean"] = om
        if ref is None:
            ref = {k: v.clone() for k, v in cur.items()}
        else:
            for k in ref:
                if not torch.equal(ref[k], cur[k]):
                    d = (ref[k] - cur[k]).abs()
                    idx = (d > 0).nonzero()[:3].tolist()
                    print(f"{H}x{D} rep {r}: {k} differs in {int((d > 0).sum())} elements, max {float(d.max()):.3e}, first {idx}", flush=True)
                    bad += 1
    print(f"{H}x{D}: done", flush=True)
hog.kill()
print("BAD" if bad else "all bitwise repeatable", bad)
