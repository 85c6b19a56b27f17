"""Does ordinary training trip the GEMM range monitor?  50 steps of each config (dropout on), then the device counter."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spgnn_amd import models, ops, synthetic, train
from spgnn_amd.configs import class_weight_list, get_config
trees = int(sys.argv[1]) if len(sys.argv) > 1 else 64
for name in ("st_pgat_spgnn_3", "st_gat_3", "st_gat_6", "st_gcn_3", "st_gin_3", "st_sage_3"):
    cfg = get_config(name)
    g = synthetic.make_batch(trees, rank=0, device="cuda", pos_enc_dim=getattr(cfg, "POS_ENC_DIM", None))
    torch.manual_seed(0)
    model = models.build_model(cfg.MODEL).cuda(); model.init(None); model.set_gcn_only(); model.train(True)
    st = train.TrainStep(model, class_weight_list(cfg.CLASS_WEIGHTS), cfg.SAMPLING_RATE, cfg.OPTIMIZER["lr"], cfg.OPTIMIZER["momentum"])
    before = st.range_violations()
    for _ in range(50):
        loss = st.step(g)
    st.step(g)                                   # the flags of step 50 are counted when step 51 begins
    print(f"{name} {trees} trees: loss {float(loss):.4f}, range violations in 50 steps: {st.range_violations() - before}", flush=True)
