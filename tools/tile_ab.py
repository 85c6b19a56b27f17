"""LDS tile kernels (csrc/spgnn_tile.hip) against the row kernels, one process, interleaved rounds: per (dtype, H, D) the
three GATConv traversals timed back to back (10 launches per sample, median of 7 samples) with ops.TILE_KERNELS off / on.
  python tools/tile_ab.py [trees] [cap,cap,...]      -> a table + one JSON line (profiles/r05_tile_ab_*.json)"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spgnn_amd import ops, synthetic

trees = int(sys.argv[1]) if len(sys.argv) > 1 else 512
caps = [int(c) for c in sys.argv[2].split(",")] if len(sys.argv) > 2 else [192]
dev = torch.device("cuda")
g = synthetic.make_batch(trees, pos_enc_dim=None, fv_dim=8).to(dev)
csc = g.csc(); N, E = csc.num_nodes, csc.num_edges
BF = torch.bfloat16
import spgnn_amd._capi as _capi
lib = _capi.load()

def t_once(fn, iters=10):
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3

def bytes_of(kind, H, D, s):
    HD = H * D
    if kind == "fwd": return s * 3 * N * HD + 4 * (2 * N * H + E * H) + 4 * (N + 1 + E)
    if kind == "dst": return s * 4 * N * HD + 4 * (3 * N * H + 2 * E * H) + 4 * (N + 1 + E)
    return s * 2 * N * HD + 4 * (N * H + 2 * E * H) + 4 * (N + 1 + 2 * E)

out_rows = []
for dtype, shapes in ((BF, [(2, 256), (2, 128), (2, 64)]), (torch.float32, [(2, 128), (2, 64), (1, 128), (1, 64)])):
    s_b = 2 if dtype == BF else 4
    for (H, D) in shapes:
        HD = H * D
        y = torch.randn(N, 2 * HD, device=dev).to(dtype); s = torch.randn(N, 2 * H, device=dev); bias = torch.zeros(HD, device=dev)
        g_out = torch.randn(N, HD, device=dev).to(dtype); g_y = torch.empty_like(y); g_s = torch.empty_like(s)
        out = torch.empty(N, HD, device=dev, dtype=dtype); attn = torch.empty(E, H, device=dev); g_e = torch.empty(E, H, device=dev)
        al = torch.randn(HD, device=dev); ar = torch.randn(HD, device=dev)
        st = torch.cuda.current_stream().cuda_stream
        ell = ops._ell(csc)
        sfx = "_bf16" if dtype == BF else ""
        def row(kind):
            if kind == "fwd":
                args = [csc.indptr.data_ptr(), csc.indices.data_ptr(), ell[0], y.data_ptr(), y.stride(0), s.data_ptr(), s[:, H:].data_ptr(), s.stride(0),
                        y[:, HD:].data_ptr(), y.stride(0), bias.data_ptr(), out.data_ptr(), out.stride(0), 0, 0, attn.data_ptr(), N, E, H, D, 0.2, ops.ACT_ELU,
                        0.1, 7, 0, 0.1, 9, HD, 0]
                args += ([0, st] if dtype != BF else [st])
                return lambda: _capi.check(getattr(lib, "spgnn_gat_fwd" + sfx)(*args), "fwd")
            if kind == "dst":
                args = [csc.indptr.data_ptr(), csc.indices.data_ptr(), ell[0], y.data_ptr(), y.stride(0), s.data_ptr(), s[:, H:].data_ptr(), s.stride(0),
                        attn.data_ptr(), g_out.data_ptr(), g_out.stride(0), 0, out.data_ptr(), out.stride(0), g_y[:, HD:].data_ptr(), g_y.stride(0),
                        g_e.data_ptr(), g_s[:, H:].data_ptr(), g_s.stride(0)]
                args += ([0] if dtype != BF else [])
                args += [N, E, H, D, 0.2, ops.ACT_ELU, 0.1, 7, 0, 0.1, 9, HD, 0, st]
                return lambda: _capi.check(getattr(lib, "spgnn_gat_bwd_dst" + sfx)(*args), "dst")
            args = [csc.out_indptr.data_ptr(), csc.out_indices.data_ptr(), csc.out_pos.data_ptr(), ell[1], ell[2], attn.data_ptr(), g_e.data_ptr(),
                    g_y[:, HD:].data_ptr(), g_y.stride(0), g_y.data_ptr(), g_y.stride(0), g_s.data_ptr(), g_s.stride(0)]
            args += ([0] if dtype != BF else [])
            args += [al.data_ptr(), ar.data_ptr(), g_s[:, H:].data_ptr(), N, E, H, D, 0.1, 7, 0, st]
            return lambda: _capi.check(getattr(lib, "spgnn_gat_bwd_src" + sfx)(*args), "src")
        def tile(kind, cap):
            t, n = csc.tiles(cap)
            if kind == "fwd":
                args = [t.data_ptr(), n, cap, csc.indptr.data_ptr(), ell[0], y.data_ptr(), y.stride(0), s.data_ptr(), s[:, H:].data_ptr(), s.stride(0),
                        y[:, HD:].data_ptr(), y.stride(0), bias.data_ptr(), out.data_ptr(), out.stride(0), attn.data_ptr(), 0, N, H, D, 0.2, ops.ACT_ELU,
                        0.1, 7, 0, 0.1, 9, HD, 0, st]
                return lambda: _capi.check(getattr(lib, "spgnn_gat_fwd_tile" + sfx)(*args), "fwd")
            if kind == "dst":
                args = [t.data_ptr(), n, cap, csc.indptr.data_ptr(), ell[0], y.data_ptr(), y.stride(0), s.data_ptr(), s[:, H:].data_ptr(), s.stride(0),
                        attn.data_ptr(), g_out.data_ptr(), g_out.stride(0), out.data_ptr(), out.stride(0), g_y[:, HD:].data_ptr(), g_y.stride(0),
                        g_e.data_ptr(), g_s[:, H:].data_ptr(), g_s.stride(0), 0, N, H, D, 0.2, ops.ACT_ELU, 0.1, 7, 0, 0.1, 9, HD, 0, st]
                return lambda: _capi.check(getattr(lib, "spgnn_gat_bwd_dst_tile" + sfx)(*args), "dst")
            args = [t.data_ptr(), n, cap, csc.indptr.data_ptr(), csc.out_indptr.data_ptr(), ell[1], ell[2], attn.data_ptr(), g_e.data_ptr(),
                    g_y[:, HD:].data_ptr(), g_y.stride(0), g_y.data_ptr(), g_y.stride(0), g_s.data_ptr(), g_s.stride(0), 0, al.data_ptr(), ar.data_ptr(),
                    g_s[:, H:].data_ptr(), N, H, D, 0.1, 7, 0, st]
            return lambda: _capi.check(getattr(lib, "spgnn_gat_bwd_src_tile" + sfx)(*args), "src")
        for kind in ("fwd", "dst", "src"):
            fns = {"row": row(kind)}
            for cap in caps:
                if lib.spgnn_gat_tile_supported(H, D, s_b, cap):
                    fns[f"tile{cap}"] = tile(kind, cap)
            for f in fns.values(): f()
            torch.cuda.synchronize()
            rounds = {k: [] for k in fns}
            for r in range(7):
                for k, f in fns.items(): rounds[k].append(t_once(f))
            med = {k: sorted(v)[len(v) // 2] for k, v in rounds.items()}
            b = bytes_of(kind, H, D, s_b)
            rec = {"dtype": "bf16" if dtype == BF else "f32", "H": H, "D": D, "kind": kind, "algorithmic_MB": round(b / 1e6, 1),
                   **{k + "_us": round(v, 1) for k, v in med.items()}, **{k + "_TBps": round(b / (v * 1e-6) / 1e12, 2) for k, v in med.items()}}
            out_rows.append(rec)
            print(rec, flush=True)
print(json.dumps({"trees": trees, "N": N, "E": E, "rows": out_rows}))
