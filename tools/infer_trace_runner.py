"""(runner variant) Kernel sequence of ONE ForwardRunner call: arena load + graph replay.  Kernel sequence of ONE eval-mode forward on a single tree (per-scan inference): names and durations from torch.profiler."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spgnn_amd import models, ops, synthetic
from spgnn_amd.configs import get_config
from torch.profiler import ProfilerActivity, profile
name = sys.argv[1] if len(sys.argv) > 1 else "st_pgat_spgnn_3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 150
cfg = get_config(name)
torch.manual_seed(0)
model = models.build_model(cfg.MODEL).cuda(); model.init(None); model.set_gcn_only(); model.eval()
g = synthetic.make_batch(1, rank=700, device="cuda", pos_enc_dim=getattr(cfg, "POS_ENC_DIM", None), fixed_n=n)
from spgnn_amd.infer import ForwardRunner
runner = ForwardRunner(model, granule=64)
with torch.no_grad():
    for _ in range(3): runner(g)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
        runner(g); torch.cuda.synchronize()
evs = [e for e in prof.events() if str(getattr(e, "device_type", "")).endswith("CUDA")]
evs.sort(key=lambda e: e.time_range.start)
tot = 0.0
for e in evs:
    d = e.time_range.end - e.time_range.start
    tot += d
    print(f"{d:8.1f} us  {e.name[:150]}")
print(f"{len(evs)} kernels, {tot:.1f} us of kernel time")
