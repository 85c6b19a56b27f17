"""Whole training step with each build under build/variants/t_*.so, interleaved in one process (medians).
Steps are HIP-graph replays (TrainStep.capture), one capture per build: eagerly issued steps are host-paced."""
import glob, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spgnn_amd import _capi, models, synthetic
from spgnn_amd.configs import class_weight_list, get_config
from spgnn_amd.train import TrainStep
paths = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "..", "build", "variants", "t_*.so")))
libs = {}
for p in paths:
    _capi._lib = None; _capi.LIB_PATH = p
    libs[os.path.basename(p)] = _capi.load()
cfg = get_config(sys.argv[1] if len(sys.argv) > 1 else "st_pgat_spgnn_3")
torch.manual_seed(0)
g = synthetic.make_batch(int(os.environ.get("TREES", "512")), rank=0, device="cuda", pos_enc_dim=getattr(cfg, "POS_ENC_DIM", None))
steps = {}
for k, lib in libs.items():
    _capi._lib = lib
    torch.manual_seed(0)
    model = models.build_model(cfg.MODEL).cuda()
    model.init(None); model.set_gcn_only(); model.train(True)
    if os.environ.get("DTYPE") == "bf16": models.set_storage_dtype(model, torch.bfloat16)
    st = TrainStep(model, class_weight_list(cfg.CLASS_WEIGHTS), cfg.SAMPLING_RATE, 1e-4, 0.9)
    for _ in range(5): st.step(g)
    st.capture(g)
    steps[k] = st
torch.cuda.synchronize()
res = {k: [] for k in libs}
for r in range(7):
    for k, st in steps.items():
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): st.replay()
        torch.cuda.synchronize(); res[k].append((time.perf_counter() - t0) / 20 * 1e3)
for k, v in res.items():
    print(k, "median %.3f ms/step" % sorted(v)[len(v) // 2], ["%.2f" % x for x in v])
