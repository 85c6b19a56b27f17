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


def main():
    config, dtype, trees, reps, sync_dir, me, n = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5], \
        int(sys.argv[6]), int(sys.argv[7])
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
