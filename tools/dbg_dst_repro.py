"""Debug aid for SPGNN_DIST_DST: two independent processes on one GPU repeat the same forward + backward pass and compare
every output of every gat_bwd_raw call with the first repetition, bit for bit; prints the first tensor that differs.
env: LIBV (variant .so), REPS, TREES."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.multiprocessing as mp


def worker(rank, ret):
    from spgnn_amd import _capi
    if os.environ.get("LIBV"):
        _capi.LIB_PATH = os.environ["LIBV"]
    from spgnn_amd import models, ops, synthetic
    from spgnn_amd.configs import class_weight_list, get_config
    from spgnn_amd.train import masked_weighted_ce
    ops.USE_ELL = os.environ.get("ELL", "1") == "1"
    cfg = get_config("st_pgat_spgnn_3")
    torch.manual_seed(0)
    model = models.build_model(cfg.MODEL).cuda()
    model.init(None); model.set_gcn_only(); model.eval()
    g = synthetic.make_batch(int(os.environ.get("TREES", "3")), rank=rank, device="cuda", pos_enc_dim=cfg.POS_ENC_DIM)
    w = torch.tensor(class_weight_list(cfg.CLASS_WEIGHTS)).cuda()
    y = g.ndata["y"]; mask = torch.ones_like(y, dtype=torch.bool)
    rec = []
    inner = ops.gat_bwd_raw

    def spy(csc, ft, el, er, attn, g_out, out, H, D, slope, act, p_drop, seed, g_pre, g_ft, g_el, g_er, **kw):
        g_e = inner(csc, ft, el, er, attn, g_out, out, H, D, slope, act, p_drop, seed, g_pre, g_ft, g_el, g_er, **kw)
        rec.append(dict(H=H, D=D, g_out=g_out.clone(), ft=ft.clone(), attn=attn.clone(), g_pre=g_pre.clone(), g_e=g_e.clone(),
                        g_er=g_er.clone(), g_ft=g_ft.clone(), g_el=g_el.clone()))
        return g_e
    ops.gat_bwd_raw = spy
    first, bad = None, []
    for rep in range(int(os.environ.get("REPS", "60"))):
        rec.clear()
        model.zero_grad(set_to_none=True)
        logits = model(g)[0]
        masked_weighted_ce(logits, y, mask, w).backward()
        torch.cuda.synchronize()
        if first is None:
            first = [dict(r) for r in rec]
            continue
        for i, (a, b) in enumerate(zip(first, rec)):
            done = False
            for k in ("g_out", "ft", "attn", "g_pre", "g_e", "g_er", "g_ft", "g_el"):
                if not torch.equal(a[k], b[k]):
                    d = (a[k] != b[k])
                    idx = d.nonzero()[:4].tolist()
                    msg = (f"proc {rank} rep {rep} call {i} (H={a['H']} D={a['D']}) {k}: {int(d.sum())} of {d.numel()} differ, first at {idx}, "
                           f"{[float(a[k][tuple(j)]) for j in idx[:2]]} vs {[float(b[k][tuple(j)]) for j in idx[:2]]}")
                    if k == "g_e":                       # which run is right, and which dot is off?
                        csc = g.csc()
                        slot, hh = idx[0]
                        ip = csc.indptr.cpu()
                        v = int(torch.searchsorted(ip, torch.tensor(slot), right=True)) - 1
                        beg, end = int(ip[v]), int(ip[v + 1])
                        us = csc.indices[beg:end].long()
                        H, D = a["H"], a["D"]
                        gp = b["g_pre"][v].double().view(H, D)[hh]
                        ga = (b["ft"][us].double().view(-1, H, D)[:, hh] * gp).sum(-1)
                        al = b["attn"][beg:end, hh].double()
                        S = (al * ga).sum()
                        ge_ref = al * (ga - S)           # before the LeakyReLU factor
                        r0 = a["g_e"][beg:end, hh].double() / ge_ref
                        r1 = b["g_e"][beg:end, hh].double() / ge_ref
                        msg += (f" | node {v} deg {end - beg} slots {beg}..{end - 1} head {hh}: first-run/ref {[round(float(t), 4) for t in r0]} "
                                f"this-run/ref {[round(float(t), 4) for t in r1]} ga {[float('%.3e' % t) for t in ga]} al {[round(float(t), 3) for t in al]}"
                                f" g_er equal {bool(torch.equal(a['g_er'], b['g_er']))}")
                    bad.append(msg)
                    done = True
                    break
            if done:
                break
    ret[rank] = bad


if __name__ == "__main__":
    mgr = mp.Manager(); ret = mgr.dict()
    mp.spawn(worker, args=(ret,), nprocs=2, join=True)
    for r in (0, 1):
        print("proc", r, len(ret[r]), "repetitions with a difference")
        for line in ret[r][:12]:
            print("  ", line)
