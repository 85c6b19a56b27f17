import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spgnn_amd import ops_bf16
torch.manual_seed(0)
for (M, N, K) in [(128, 128, 64), (128, 128, 32), (128, 128, 8), (1, 4, 8), (256, 256, 128)]:
    a = torch.ones(M, K); b = torch.ones(N, K)
    out = ops_bf16.gemm_nt(ops_bf16.cast_rows(a.cuda()), ops_bf16.cast_rows(b.cuda()), out_f32=True)
    o = out.cpu()
    print(M, N, K, "nan:", int(torch.isnan(o).sum()), "min/max", float(o[~torch.isnan(o)].min()) if (~torch.isnan(o)).any() else None,
          float(o[~torch.isnan(o)].max()) if (~torch.isnan(o)).any() else None, "expected", K)
    a = torch.randint(-3, 4, (M, K)).float(); b = torch.randint(-3, 4, (N, K)).float()
    out = ops_bf16.gemm_nt(ops_bf16.cast_rows(a.cuda()), ops_bf16.cast_rows(b.cuda()), out_f32=True).cpu()
    ref = a @ b.t()
    bad = (out != ref)
    print("  rand: mismatches", int(bad.sum()), "of", out.numel(), "first bad idx", bad.nonzero()[:4].tolist())
