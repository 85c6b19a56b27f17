"""Two ranks on ONE GPU (gloo moves the tensors): five eager steps, five eager steps again, and 2 eager + 3 graph-replayed
steps from the same seeds; the parameters must agree bit for bit, trial after trial.  Found the run-to-run nondeterminism of
the entry-per-lane dst-major kernel (spgnn_kernels.hip SPGNN_DIST_DST).  env: TRIALS, FUSE, ELL, LIBV (variant .so)."""
import os, sys, socket
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import torch.multiprocessing as mp

def worker(rank, world, port, ret, fuse, ell):
    import torch.distributed as dist
    from spgnn_amd import _capi
    if os.environ.get('LIBV'):
        _capi.LIB_PATH = os.environ['LIBV']
    from spgnn_amd import models, synthetic, ops
    from spgnn_amd.configs import class_weight_list, get_config
    from spgnn_amd.train import TrainStep
    models.FUSE_OUTPUT_DROPOUT = fuse
    ops.USE_ELL = ell
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = get_config("st_pgat_spgnn_3")
    out = {}
    for mode in ("eager", "eager2", "graph"):
        torch.manual_seed(0)
        model = models.build_model(cfg.MODEL).cuda()
        model.init(None); model.set_gcn_only(); model.eval()
        g = synthetic.make_batch(3, rank=rank, device="cuda", pos_enc_dim=cfg.POS_ENC_DIM)
        ts = TrainStep(model, class_weight_list(cfg.CLASS_WEIGHTS), 1.0, 0.05, 0.9, seed=5)
        if mode != "graph":
            losses = [float(ts.step(g)) for _ in range(5)]
        else:
            ts.capture(g, warmup=2)
            losses = [float(ts.replay()) for _ in range(3)]
        out[mode] = (losses, ts.bucket.flat_param[:ts.bucket.numel].detach().cpu().clone())
    ret[rank] = out
    dist.destroy_process_group()

if __name__ == "__main__":
    fuse = os.environ.get("FUSE", "1") == "1"; ell = os.environ.get("ELL", "1") == "1"
    for trial in range(int(os.environ.get("TRIALS", "5"))):
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        mgr = mp.Manager(); ret = mgr.dict()
        mp.spawn(worker, args=(2, port, ret, fuse, ell), nprocs=2, join=True)
        for r in (0, 1):
            e, e2, gph = ret[r]["eager"][1], ret[r]["eager2"][1], ret[r]["graph"][1]
            print("trial", trial, "rank", r, "fuse", fuse, "ell", ell, "eager-eager2 %.3e" % float((e - e2).abs().max()),
                  "eager-graph %.3e" % float((e - gph).abs().max()), "n>1e-7:", int(((e - gph).abs() > 1e-7).sum()), flush=True)
        print(" ranks equal:", bool(torch.equal(ret[0]["graph"][1], ret[1]["graph"][1])), bool(torch.equal(ret[0]["eager"][1], ret[1]["eager"][1])), flush=True)
