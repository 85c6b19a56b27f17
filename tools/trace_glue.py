"""Which Python lines launch the small torch kernels (fill / copy / reduce / cat) of a training step."""
import os, sys, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torch.profiler import profile, ProfilerActivity
from spgnn_amd import models, synthetic
from spgnn_amd.configs import class_weight_list, get_config
from spgnn_amd.train import TrainStep
cfg = get_config(os.environ.get("CONFIG", "st_pgat_spgnn_3"))
model = models.build_model(cfg.MODEL).cuda()
g = synthetic.make_batch(64, rank=0, device="cuda", pos_enc_dim=getattr(cfg, "POS_ENC_DIM", None))
step = TrainStep(model, class_weight_list(cfg.CLASS_WEIGHTS), cfg.SAMPLING_RATE, 1e-4, 0.9)
for _ in range(3): step.step(g)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=False) as prof:
    step.step(g)
torch.cuda.synchronize()
want = ("aten::fill_", "aten::zero_", "aten::copy_", "aten::sum", "aten::cat", "aten::add", "aten::add_", "aten::mul",
        "aten::pad", "aten::stack", "aten::mm", "aten::bmm", "aten::addmm", "aten::index", "aten::index_put_")
rows = []
for ev in prof.key_averages(group_by_stack_n=12):
    if ev.key in want:
        frames = [f for f in ev.stack if "/spgnn_amd/" in f]
        where = frames[0].split("/spgnn_amd/")[-1] if frames else "(autograd engine / other): " + (ev.stack[0][-60:] if ev.stack else "?")
        rows.append((ev.count, ev.key, where))
for c, name, where in sorted(rows, key=lambda r: -r[0])[:70]:
    print(f"{c:3d}  {name:18s} {where}")
