"""A/B kernel variants (build/variants/*.so) in ONE process, interleaved rounds (median over rounds)."""
import glob, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spgnn_amd import _capi, ops, synthetic

paths = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "..", "build", "variants", "*.so")))
libs = {}
for p in paths:
    _capi._lib = None; _capi.LIB_PATH = p
    libs[os.path.basename(p)] = _capi.load()
dev = torch.device("cuda")
g = synthetic.make_batch(int(sys.argv[1]) if len(sys.argv) > 1 else 512, pos_enc_dim=None, fv_dim=8).to(dev)
csc = g.csc(); N, E = csc.num_nodes, csc.num_edges

def t_once(fn, iters=10):
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters

only = os.environ.get("AB_SHAPES")        # e.g. "2x1024" to restrict the sweep
shapes = [(2, 1024, True), (2, 256, False), (2, 128, False), (2, 64, False), (1, 256, False), (1, 128, False), (1, 64, False)]
if only:
    shapes = [s_ for s_ in shapes if f"{s_[0]}x{s_[1]}" in only.split(",")]
kinds = os.environ.get("AB_KINDS", "fwd,dst,src").split(",")
P = float(os.environ.get("AB_P", "0"))     # attention dropout (0.1 in the training configs)
res = {}
for (H, D, mean) in shapes:
    HD = H * D
    y = torch.randn(N, 2 * HD, device=dev); s = torch.randn(N, 2 * H, device=dev); bias = torch.zeros(HD, device=dev)
    g_out = torch.randn(N, D if mean else HD, device=dev); g_y = torch.empty_like(y); g_s = torch.empty_like(s)
    _capi._lib = libs[list(libs)[0]]
    out, om, attn = ops.gat_fwd_raw(csc, y[:, :HD], s[:, :H], s[:, H:], y[:, HD:], bias, H, D, 0.2, ops.ACT_ELU, P, 7, mean=mean)
    fwd = lambda: ops.gat_fwd_raw(csc, y[:, :HD], s[:, :H], s[:, H:], y[:, HD:], bias, H, D, 0.2, ops.ACT_ELU, P, 7, out=out, mean=mean)
    st = torch.cuda.current_stream().cuda_stream
    g_e = torch.empty(E, H, device=dev)
    def dst():
        _capi.check(_capi._lib.spgnn_gat_bwd_dst(csc.indptr.data_ptr(), csc.indices.data_ptr(), ops._ell(csc)[0], y.data_ptr(), y.stride(0),
            s.data_ptr(), s[:, H:].data_ptr(), s.stride(0), attn.data_ptr(), g_out.data_ptr(), g_out.stride(0), int(mean),
            out.data_ptr(), out.stride(0), g_y[:, HD:].data_ptr(), g_y.stride(0), g_e.data_ptr(), g_s[:, H:].data_ptr(),
            g_s.stride(0), 0, N, E, H, D, 0.2, ops.ACT_ELU, P, 7, 0, 0.0, 0, 0, 0, st), "dst")
    def src():
        _capi.check(_capi._lib.spgnn_gat_bwd_src(csc.out_indptr.data_ptr(), csc.out_indices.data_ptr(), csc.out_pos.data_ptr(), ops._ell(csc)[1], ops._ell(csc)[2],
            attn.data_ptr(), g_e.data_ptr(), g_y[:, HD:].data_ptr(), g_y.stride(0), g_y.data_ptr(), g_y.stride(0),
            g_s.data_ptr(), g_s.stride(0), 0, 0, 0, 0, N, E, H, D, P, 7, 0, st), "src")
    for name, fn in (("fwd", fwd), ("dst", dst), ("src", src)):
        if name not in kinds:
            continue
        for lib in libs.values():
            _capi._lib = lib; fn()
        torch.cuda.synchronize()
        rounds = {k: [] for k in libs}
        for r in range(7):
            for k, lib in libs.items():
                _capi._lib = lib
                rounds[k].append(t_once(fn))
        res[(H, D, name)] = {k: sorted(v)[len(v) // 2] for k, v in rounds.items()}
        print(H, D, name, " ".join(f"{k}={v*1e3:.1f}us" for k, v in res[(H, D, name)].items()), flush=True)
    del y, s, g_out, g_y, g_s, out, attn, g_e
tot = {k: {n: sum(v[k] for kk, v in res.items() if kk[2] == n) for n in kinds} for k in libs}
for k, v in tot.items():
    print(k, {n: round(x * 1e3, 1) for n, x in v.items()}, "total us", round(sum(v.values()) * 1e3, 1))
