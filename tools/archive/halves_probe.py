"""Would two half-batches pipelined on two streams beat one full batch?  Two independent TrainSteps (model copies) on 256 trees
each, captured separately; replays back to back on one stream vs concurrently on two streams vs ONE step on the 512 trees."""
import os, sys, copy
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spgnn_amd import models, ops, synthetic, train
from spgnn_amd.configs import class_weight_list, get_config
cfgname = os.environ.get("CONFIG", "st_pgat_spgnn_3")
cfg = get_config(cfgname)
pe = getattr(cfg, "POS_ENC_DIM", None)
samples = synthetic.synthetic_trees(512, rank=0)
g_full = synthetic.batch_from_samples(samples, "cuda", pe)
g_a = synthetic.batch_from_samples(samples[:256], "cuda", pe)
g_b = synthetic.batch_from_samples(samples[256:], "cuda", pe)
def mk(g):
    torch.manual_seed(0)
    m = models.build_model(cfg.MODEL).cuda(); m.init(None); m.set_gcn_only(); m.train(True)
    st = train.TrainStep(m, class_weight_list(cfg.CLASS_WEIGHTS), cfg.SAMPLING_RATE, 1e-4, 0.9)
    st.capture(g)
    return st
full, a, b = mk(g_full), mk(g_a), mk(g_b)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def timed(fn, n=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
def seq(): a.replay(); b.replay()
def par():
    main = torch.cuda.current_stream()
    s1.wait_stream(main); s2.wait_stream(main)
    with torch.cuda.stream(s1): a.replay()
    with torch.cuda.stream(s2): b.replay()
    main.wait_stream(s1); main.wait_stream(s2)
for f in (full.replay, seq, par): timed(f, 5)
res = {"full512": [], "halves_seq": [], "halves_par": []}
for r in range(7):
    res["full512"].append(timed(full.replay)); res["halves_seq"].append(timed(seq)); res["halves_par"].append(timed(par))
print(cfgname, {k: round(sorted(v)[len(v) // 2], 3) for k, v in res.items()}, "ms")
