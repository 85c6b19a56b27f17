"""Whole training step (HIP-graph replays) under different module-level switches, ONE process, interleaved rounds, medians.
usage: step_toggle_ab.py <trees> name=module.ATTR:value[,module.ATTR:value] ...   (e.g. base= lspe_off=models.FUSE_LSPE:0)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spgnn_amd import models, ops, ops_bf16, nn as snn, synthetic, train
from spgnn_amd.configs import class_weight_list, get_config
mods = {"models": models, "ops": ops, "ops_bf16": ops_bf16, "nn": snn, "train": train}
trees = int(sys.argv[1])
cfgname = os.environ.get("CONFIG", "st_pgat_spgnn_3")
variants = []
for spec in sys.argv[2:]:
    name, _, rest = spec.partition("=")
    sets = []
    for item in filter(None, rest.split(",")):
        path, _, val = item.partition(":")
        m, attr = path.split(".")
        sets.append((mods[m], attr, type(getattr(mods[m], attr))(int(val)) if isinstance(getattr(mods[m], attr), bool) else int(val)))
    variants.append((name, sets))
cfg = get_config(cfgname)
g = synthetic.make_batch(trees, rank=0, device="cuda", pos_enc_dim=getattr(cfg, "POS_ENC_DIM", None))
steps = {}
for name, sets in variants:
    old = [(m, a, getattr(m, a)) for m, a, _ in sets]
    for m, a, v in sets: setattr(m, a, v)
    torch.manual_seed(0)
    model = models.build_model(cfg.MODEL).cuda(); model.init(None); model.set_gcn_only(); model.train(True)
    if os.environ.get("DTYPE") == "bf16": models.set_storage_dtype(model, torch.bfloat16)
    st = train.TrainStep(model, class_weight_list(cfg.CLASS_WEIGHTS), cfg.SAMPLING_RATE, 1e-4, 0.9)
    st.capture(g)
    steps[name] = st
    for m, a, v in old: setattr(m, a, v)
def timed(st, n=20):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): st.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
for st in steps.values(): timed(st, 5)
res = {k: [] for k in steps}
for r in range(7):
    for k, st in steps.items(): res[k].append(timed(st))
print(f"{cfgname} {trees} trees: " + " | ".join(f"{k} {sorted(v)[len(v)//2]:.3f} ms" for k, v in res.items()), flush=True)
