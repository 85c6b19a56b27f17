"""Which torch ops launch the small fill / copy kernels of a training step (torch.profiler, CPU-side op names with their
python stacks, two eagerly issued steps)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spgnn_amd import models, synthetic
from spgnn_amd.configs import class_weight_list, get_config
from spgnn_amd.train import TrainStep
from torch.profiler import profile, ProfilerActivity
cfg = get_config("st_pgat_spgnn_3")
torch.manual_seed(0)
model = models.build_model(cfg.MODEL).cuda(); model.init(None); model.set_gcn_only(); model.train(True)
g = synthetic.make_batch(int(sys.argv[1]) if len(sys.argv) > 1 else 64, rank=0, device="cuda", pos_enc_dim=cfg.POS_ENC_DIM)
st = TrainStep(model, class_weight_list(cfg.CLASS_WEIGHTS), cfg.SAMPLING_RATE, 1e-4, 0.9)
for _ in range(4): st.step(g)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    st.step(g)
    torch.cuda.synchronize()
want = ("aten::zeros", "aten::zero_", "aten::fill_", "aten::copy_", "aten::clone", "aten::contiguous", "aten::add", "aten::add_", "aten::mul",
        "aten::sum", "aten::cat", "aten::stack", "aten::zeros_like", "aten::reciprocal", "aten::_to_copy")
rows = {}
for e in prof.events():
    if e.name in want and e.device_type == torch.autograd.DeviceType.CPU:
        chain, q = [], e.cpu_parent
        while q is not None and len(chain) < 4:
            chain.append(q.name[:60]); q = q.cpu_parent
        shp = ""
        key = (e.name, tuple(chain))
        rows[key] = rows.get(key, 0) + 1
for (name, chain), n in sorted(rows.items(), key=lambda kv: -kv[1])[:70]:
    print(n, name, " <- ", " <- ".join(chain))
