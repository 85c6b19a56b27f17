"""Per-kernel times inside the training step for each build under build/variants/t_*.so (one process, HIP events
around every hand-written launch of eagerly issued steps; interleaved rounds, medians).  Prints the keys that differ."""
import glob, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spgnn_amd import _capi, models, ops, synthetic
from spgnn_amd.configs import class_weight_list, get_config
from spgnn_amd.train import TrainStep
paths = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "..", "build", "variants", "t_*.so")))
libs = {}
for p in paths:
    _capi._lib = None; _capi.LIB_PATH = p
    libs[os.path.basename(p)] = _capi.load()
cfg = get_config(sys.argv[1] if len(sys.argv) > 1 else "st_pgat_spgnn_3")
only = sys.argv[2] if len(sys.argv) > 2 else "gemm"
torch.manual_seed(0)
g = synthetic.make_batch(512, rank=0, device="cuda", pos_enc_dim=cfg.POS_ENC_DIM)
model = models.build_model(cfg.MODEL).cuda()
model.init(None); model.set_gcn_only(); model.train(True)
st = TrainStep(model, class_weight_list(cfg.CLASS_WEIGHTS), cfg.SAMPLING_RATE, 1e-4, 0.9)
for k, lib in libs.items():
    _capi._lib = lib
    for _ in range(4): st.step(g)
acc = {k: {} for k in libs}
for r in range(5):
    for k, lib in libs.items():
        _capi._lib = lib
        ops.KernelTimer.start()
        st.step(g)
        for key, v in ops.KernelTimer.stop().items():
            acc[k].setdefault(key, []).append(sum(v))
keys = sorted(acc[next(iter(libs))], key=lambda q: -sorted(acc[next(iter(libs))][q])[2])
tot = {k: 0.0 for k in libs}
for key in keys:
    med = {k: sorted(acc[k][key])[len(acc[k][key]) // 2] for k in libs if key in acc[k]}
    for k, v in med.items(): tot[k] += v
    if only in key[0]:
        print("_".join(str(x) for x in key), " | ".join(f"{k[2:-3]} {v*1e3:.0f}us" for k, v in med.items()))
print("sum of all instrumented kernels:", " | ".join(f"{k[2:-3]} {v:.3f} ms" for k, v in tot.items()))
