"""Where the host time of one per-scan call goes: cProfile over ForwardRunner.__call__ on 150-node scans.  usage: infer_host_profile.py [config]"""
import cProfile, io, os, pstats, sys, time, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spgnn_amd import models, synthetic
from spgnn_amd.configs import get_config
from spgnn_amd.infer import ForwardRunner
name = sys.argv[1] if len(sys.argv) > 1 else "st_pgat_spgnn_3"
cfg = get_config(name)
torch.manual_seed(0)
model = models.build_model(cfg.MODEL).cuda(); model.init(None); model.set_gcn_only(); model.eval()
runner = ForwardRunner(model, granule=64)
scans = [synthetic.make_batch(1, rank=700 + i, device="cuda", pos_enc_dim=getattr(cfg, "POS_ENC_DIM", None), fixed_n=150) for i in range(4)]
for g in scans:
    g.csc("cuda"); runner(g)
torch.cuda.synchronize()
ts = []
for i in range(300):
    g = scans[i % 4]
    torch.cuda.synchronize(); t0 = time.perf_counter(); runner(g); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e6)
print(f"{name}: per scan {statistics.median(ts):.1f} us (p90 {sorted(ts)[270]:.1f})")
# host-only time: no synchronisation inside the loop
t0 = time.perf_counter()
for i in range(300): runner(scans[i % 4])
t1 = time.perf_counter(); torch.cuda.synchronize()
print(f"host issue time per scan {(t1 - t0) / 300 * 1e6:.1f} us")
pr = cProfile.Profile(); pr.enable()
for i in range(300): runner(scans[i % 4])
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:6000])
