"""Two INDEPENDENT single-rank processes on one GPU (no process group, no gloo): each runs five eager steps, five eager steps
again and 2 eager + 3 replayed steps from the same seeds and compares its own parameters bit for bit.  Separates "the kernels
are not repeatable when another process shares the GPU" from "the 2-rank gloo harness races" for SPGNN_DIST_DST
(tools/two_rank_determinism.py).  env: TRIALS, LIBV (variant .so), TREES, CONFIG, DTYPE=bf16."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.multiprocessing as mp


def worker(rank, ret):
    from spgnn_amd import _capi
    if os.environ.get("LIBV"):
        _capi.LIB_PATH = os.environ["LIBV"]
    from spgnn_amd import models, synthetic
    from spgnn_amd.configs import class_weight_list, get_config
    from spgnn_amd.train import TrainStep
    cfg = get_config(os.environ.get("CONFIG", "st_pgat_spgnn_3"))
    out = {}
    for mode in ("eager", "eager2", "graph"):
        torch.manual_seed(0)
        model = models.build_model(cfg.MODEL).cuda()
        model.init(None); model.set_gcn_only(); model.eval()
        if os.environ.get("DTYPE") == "bf16":
            models.set_storage_dtype(model, torch.bfloat16)
        g = synthetic.make_batch(int(os.environ.get("TREES", "3")), rank=rank, device="cuda", pos_enc_dim=cfg.POS_ENC_DIM)
        ts = TrainStep(model, class_weight_list(cfg.CLASS_WEIGHTS), 1.0, float(os.environ.get("LR", "0.05")), 0.9, seed=5)
        if mode != "graph":
            [float(ts.step(g)) for _ in range(5)]
        else:
            ts.capture(g, warmup=2)
            [float(ts.replay()) for _ in range(3)]
        out[mode] = ts.bucket.flat_param[:ts.bucket.numel].detach().cpu().clone()
    ret[rank] = out


if __name__ == "__main__":
    for trial in range(int(os.environ.get("TRIALS", "5"))):
        mgr = mp.Manager(); ret = mgr.dict()
        mp.spawn(worker, args=(ret,), nprocs=2, join=True)
        for r in (0, 1):
            e, e2, gph = ret[r]["eager"], ret[r]["eager2"], ret[r]["graph"]
            print("trial", trial, "proc", r, "eager-eager2 %.3e" % float((e - e2).abs().max()),
                  "eager-graph %.3e" % float((e - gph).abs().max()), flush=True)
