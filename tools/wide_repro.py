"""ops.GEMM_WIDE smoke: eager steps, then capture + replays, at the given tree count (stderr tells what failed)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spgnn_amd import models, ops, synthetic, train
from spgnn_amd.configs import class_weight_list, get_config
trees = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ops.GEMM_WIDE = True
cfg = get_config("st_pgat_spgnn_3")
g = synthetic.make_batch(trees, rank=0, device="cuda", pos_enc_dim=cfg.POS_ENC_DIM)
torch.manual_seed(0)
model = models.build_model(cfg.MODEL).cuda(); model.init(None); model.set_gcn_only(); model.train(True)
st = train.TrainStep(model, class_weight_list(cfg.CLASS_WEIGHTS), cfg.SAMPLING_RATE, 1e-4, 0.9)
for i in range(3):
    l = st.step(g); torch.cuda.synchronize(); print("eager", i, float(l), flush=True)
st.capture(g); torch.cuda.synchronize(); print("captured", flush=True)
for i in range(3):
    l = st.replay(); torch.cuda.synchronize(); print("replay", i, float(l), flush=True)
