"""Child process of tests/test_two_process.py (never collected by pytest: no ``test_`` prefix).

Runs REPS forward + backward repetitions of one model on the GPU it shares with its sibling process and compares every
repetition's logits and parameter gradients BITWISE with the first one.  Same seeds every repetition (dropout masks repeat),
so any difference is a kernel that is not run-to-run repeatable under GPU sharing - the hazard of DESIGN.md section 4.1
(packed fp32 ops next to cross-lane reads), which only ever showed with a second process on the device.

usage: two_proc_worker.py <config> <f32|bf16> <trees> <reps> <sync dir> <my id> <n procs>   -> one JSON line on stdout
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def rendezvous(sync_dir: str, me: int, n: int, tag: str, timeout: float = 180.0) -> bool:
    """All ``n`` processes reach ``tag`` (files in a directory: no process group, the processes stay independent)."""
    open(os.path.join(sync_dir, f"{tag}.{me}"), "w").close()
    t0 = time.time()
    while time.time() - t0 < timeout:
        if all(os.path.exists(os.path.join(sync_dir, f"{tag}.{i}")) for i in range(n)):
            return True
        time.sleep(0.01)
    return False


def gemm_loop(reps: int, sync_dir: str, me: int, n: int):
    """The matrix-core kernels on their own: every epilogue that reads across lanes - the 16-lane score
    partial sums (row16_sum), the absmax wave reduction, the rank-J update, the head mean - and the weight-gradient kernel with
    its column sums, in all three NT tile shapes, pre-split and not, repeated and compared bitwise with the first run."""
    import torch
    from spgnn_amd import _capi, ops
    _capi.load()
    torch.manual_seed(7 + me)
    M, N, K = 5000, 512, 1063
    a = torch.randn(M, K + 1, device="cuda")[:, :K]
    b = (torch.randn(N, K + 1, device="cuda") / 5)[:, :K]
    sa, sb = ops.pow2_scale(a), ops.pow2_scale(b)
    a_ps, b_ps = ops.presplit(a, scale=sa)[0], ops.presplit(b, scale=sb)[0]
    sl, sr = torch.randn(N, device="cuda"), torch.randn(N, device="cuda")
    u, v = torch.randn(M, 4, device="cuda"), torch.randn(4, N, device="cuda")
    gy = torch.randn(M, 384, device="cuda")
    sg = ops.pow2_scale(gy)
    bw = b.detach().clone().requires_grad_(True)                 # a weight that needs a gradient: linear() keeps the result's scale block

    def one():
        outs = []
        for tile in (2, 4, 5):
            parts = torch.empty((M, N // 64, 2), device="cuda")
            outs += [ops.gemm_nt(a, b_ps, sa, sb, score_l=sl, score_r=sr, score_out=parts, tile=tile, b_presplit=True), parts]
            outs.append(ops.gemm_nt(a_ps, b_ps, sa, sb, tile=tile, b_presplit=True, a_presplit=True))
            outs.append(ops.gemm_nt(a, b, sa, sb, upd_u=u, upd_v=v, tile=tile))
        y = ops.linear(a, bw, torch.zeros(N, device="cuda"), ops.ACT_ELU)          # absmax_out: the wave reduction + slots
        tag = getattr(y, "_spgnn_scale", None)
        outs += [y.detach()] + ([tag[1].clone()] if tag is not None else [])
        gw, cs = ops.gemm_tn(gy, a, sg, sa, want_colsum=True)
        outs += [gw, cs, ops.gemm_tn(gy, a_ps, sg, sa, b_presplit=True)]
        return [t.clone() for t in outs]

    ref = one()
    torch.cuda.synchronize()
    together = rendezvous(sync_dir, me, n, "start")
    bad, first_bad, t0 = 0, None, time.time()
    for r in range(reps):
        same = [torch.equal(x, y) for x, y in zip(ref, one())]
        if not all(same):
            bad += 1
            first_bad = first_bad or {"rep": r, "outputs": [i for i, s_ in enumerate(same) if not s_][:6]}
    torch.cuda.synchronize()
    rendezvous(sync_dir, me, n, "done", timeout=120.0)
    print(json.dumps({"proc": me, "config": "gemm_kernels", "dtype": "f32", "reps": reps, "bad": bad, "first_bad": first_bad,
                      "overlapped": together, "seconds": round(time.time() - t0, 2), "finite": bool(all(torch.isfinite(t).all() for t in ref))}),
          flush=True)
    sys.exit(1 if bad else 0)


def main():
    config, dtype, trees, reps, sync_dir, me, n = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5], \
        int(sys.argv[6]), int(sys.argv[7])
    if config == "gemm_kernels":
        return gemm_loop(reps, sync_dir, me, n)
    import torch
    from spgnn_amd import _capi, models, ops, synthetic
    from spgnn_amd.configs import class_weight_list, get_config
    _capi.load()
    dev = torch.device("cuda", 0)
    cfg = get_config(config)
    torch.manual_seed(0)
    model = models.build_model(cfg.MODEL).to(dev)
    model.init(None)
    model.set_gcn_only()
    model.train()                                            # dropout on: the hash-mask paths of every kernel run
    if dtype == "bf16":
        models.set_storage_dtype(model, torch.bfloat16)
    g = synthetic.make_batch(trees, rank=me, device=dev, pos_enc_dim=getattr(cfg, "POS_ENC_DIM", None))
    y = g.ndata["y"]
    w = torch.tensor(class_weight_list(cfg.CLASS_WEIGHTS), dtype=torch.float32, device=dev)
    p_keep = torch.where(y != 0, torch.tensor(1.0, device=dev), torch.tensor(float(cfg.SAMPLING_RATE), device=dev))
    draws = torch.rand(y.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(11))
    params = [p for p in model.parameters() if p.requires_grad]

    def one():
        torch.manual_seed(123)                               # the dropout seeds are drawn from torch's CPU generator
        for p in params:
            p.grad = None
        logits = model(g)[0]
        num, den = ops.masked_ce_sums(logits, y, draws, p_keep, w)
        (num / den).backward()
        return [logits.detach().clone()] + [p.grad.detach().clone() for p in params]

    ref = one()
    torch.cuda.synchronize()
    together = rendezvous(sync_dir, me, n, "start")          # both processes warmed up: now they overlap on the device
    bad, first_bad = 0, None
    t0 = time.time()
    for r in range(reps):
        cur = one()
        same = [torch.equal(a, b) for a, b in zip(ref, cur)]
        if not all(same):
            bad += 1
            if first_bad is None:
                names = ["logits"] + [n_ for n_, p in model.named_parameters() if p.requires_grad]
                first_bad = {"rep": r, "tensors": [names[i] for i, s_ in enumerate(same) if not s_][:6]}
    torch.cuda.synchronize()
    dt = time.time() - t0
    rendezvous(sync_dir, me, n, "done", timeout=120.0)       # keep the device shared until the sibling has finished too
    print(json.dumps({"proc": me, "config": config, "dtype": dtype, "reps": reps, "bad": bad, "first_bad": first_bad,
                      "overlapped": together, "seconds": round(dt, 2), "finite": bool(all(torch.isfinite(t).all() for t in ref))}),
          flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
