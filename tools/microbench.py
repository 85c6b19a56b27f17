"""Per-kernel microbenchmark on one MI355X: message-passing kernels at the config layer shapes
(achieved GB/s against the algorithmic byte counts of SURVEY.md §8d) and the rocBLAS fp32 GEMMs
they sit between.  Usage: python tools/microbench.py [--trees 512]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spgnn_amd import ops, synthetic  # noqa: E402


def timeit(fn, iters=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trees", type=int, default=512)
    ap.add_argument("--out", default="gpurun_out/microbench.json")
    args = ap.parse_args()
    dev = torch.device("cuda")
    g = synthetic.make_batch(args.trees, pos_enc_dim=None, fv_dim=8).to(dev)
    csc = g.csc()
    N, E = csc.num_nodes, csc.num_edges
    res = {"N": N, "E": E, "trees": args.trees}
    big = torch.empty(1 << 28, device=dev)           # 1 GiB fp32
    t = timeit(lambda: big.clone())
    res["copy_GBps"] = 2 * big.numel() * 4 / t / 1e6
    del big
    print(f"N={N} E={E} copy {res['copy_GBps']:.0f} GB/s", flush=True)

    rows = []
    for (H, D) in [(2, 256), (2, 128), (2, 64), (2, 1024), (1, 256), (1, 128), (1, 64)]:
        HD = H * D
        y = torch.randn(N, 2 * HD, device=dev)
        s = torch.randn(N, 2 * H, device=dev)
        bias = torch.zeros(HD, device=dev)
        out, _, attn = ops.gat_fwd_raw(csc, y[:, :HD], s[:, :H], s[:, H:], y[:, HD:], bias, H, D, 0.2, ops.ACT_ELU)
        t_f = timeit(lambda: ops.gat_fwd_raw(csc, y[:, :HD], s[:, :H], s[:, H:], y[:, HD:], bias, H, D, 0.2, ops.ACT_ELU, out=out))
        g_out = torch.randn(N, HD, device=dev)
        g_y = torch.empty_like(y); g_s = torch.empty_like(s)
        lib_call = lambda: ops.gat_bwd_raw(csc, y[:, :HD], s[:, :H], s[:, H:], attn, g_out, out, H, D, 0.2, ops.ACT_ELU,
                                           0.0, 0, g_y[:, HD:], g_y[:, :HD], g_s[:, :H], g_s[:, H:])
        t_b = timeit(lambda: lib_call())
        b_f = 4 * (3 * N * HD) + 4 * (2 * N * H + E * H) + 4 * (N + 1 + E)          # + res read (fused here)
        b_b = 4 * (5 * N * HD) + 4 * (2 * N * H + 3 * E * H) + 4 * 2 * (N + 1 + E)  # g_out,out,ft in; g_pre,g_ft out
        rows.append(dict(H=H, D=D, fwd_ms=t_f, bwd_ms=t_b, fwd_GBps=b_f / t_f / 1e6, bwd_GBps=b_b / t_b / 1e6))
        print(rows[-1], flush=True)
        del y, s, out, attn, g_out, g_y, g_s
    res["gat"] = rows

    rows = []
    for F_ in [64, 128, 256, 1024]:
        x = torch.randn(N, F_, device=dev)
        w = torch.rand(N, device=dev)
        t_s = timeit(lambda: ops.spmm_sum_raw(csc.indptr, csc.indices, x, w, w, None, N, E))
        rows.append(dict(F=F_, sum_ms=t_s, sum_GBps=(2 * N * F_ * 4 + 4 * (N + 1 + E)) / t_s / 1e6))
        print(rows[-1], flush=True)
    res["spmm"] = rows

    rows = []
    for (K, C) in [(1063, 1024), (768, 512), (384, 256), (192, 4096), (39, 512), (256, 256), (128, 128), (1024, 1024)]:
        x = torch.randn(N, K, device=dev); w = torch.randn(C, K, device=dev); gy = torch.randn(N, C, device=dev)
        t_f = timeit(lambda: torch.mm(x, w.t()), 10, 3)
        t_dw = timeit(lambda: torch.mm(gy.t(), x), 10, 3)
        t_dx = timeit(lambda: torch.mm(gy, w), 10, 3)
        fl = 2 * N * K * C
        rows.append(dict(K=K, C=C, fwd_ms=t_f, dW_ms=t_dw, dX_ms=t_dx, fwd_TF=fl / t_f / 1e9, dW_TF=fl / t_dw / 1e9,
                         dX_TF=fl / t_dx / 1e9))
        print(rows[-1], flush=True)
        del x, w, gy
    res["gemm_fp32"] = rows
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(res, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
