"""One eagerly issued training step for rocprofv3 --pmc passes (HBM traffic per launch of the GAT and GEMM kernels).

usage (under the profiler, program directly after `--`):
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <dir> -- python3 tools/pmc_step.py <config> <f32|bf16> <trees> <manifest.json>

Runs two warm-up steps and ONE instrumented step of the real model (same construction as bench.py) and writes the
instrumented step's launch-order manifest [[kernel stem, bench key], ...] for the kernels tools/pmc_merge.py matches."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spgnn_amd import models, ops, synthetic
from spgnn_amd.configs import class_weight_list, get_config
from spgnn_amd.train import TrainStep

STEMS = ("gat_fwd", "gat_bwd_dst", "gat_bwd_src", "gat_agg_fwd", "gat_agg_bwd_dst", "gat_agg_bwd_src", "lspe_fwd", "lspe_bwd_dst",
         "lspe_bwd_src", "gemm_nt", "gemm_tn", "gemm_nt_pair", "gemm_tn_pair")

name, dtype, trees, mpath = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
cfg = get_config(name)
torch.manual_seed(0)
dev = torch.device("cuda:0")
model = models.build_model(cfg.MODEL).to(dev)
model.init(None)
model.set_gcn_only()
if dtype == "bf16":
    models.set_storage_dtype(model, torch.bfloat16)
model.train()
g = synthetic.batch_from_samples(synthetic.synthetic_trees(trees, rank=0), dev, cfg.POS_ENC_DIM)
step = TrainStep(model, class_weight_list(cfg.CLASS_WEIGHTS), cfg.SAMPLING_RATE, cfg.OPTIMIZER["lr"], cfg.OPTIMIZER["momentum"], seed=1234)
for _ in range(2):
    step.step(g)
torch.cuda.synchronize()
ops.KernelTimer.start()
step.step(g)
ops.KernelTimer.stop()
seq = []
for k in ops.KernelTimer.sequence:
    base = k[0][:-5] if k[0].endswith("_bf16") else k[0]
    if base in STEMS:
        seq.append([base, "_".join(str(x) for x in k)])
json.dump(seq, open(mpath, "w"))
print(len(seq), "matched launches in the instrumented step")
