"""Timing of the neighbour-sampled GraphSAGE step (reference job_runner.py:1484-1506) on one GPU:
block sampling (on the device when the graph lives there, SPGNN_SAMPLE_ON=cpu forces the host sampler), feature
gather and the device fwd+bwd+SGD, per mini-batch.

    python tools/sampled_bench.py [trees] [node_batch] [workers] [iters]
"""
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
from spgnn_amd import dgl_compat as dgl, models, synthetic  # noqa: E402
from spgnn_amd.configs import get_config  # noqa: E402


def main():
    trees = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    node_batch = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    workers = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    iters = int(sys.argv[4]) if len(sys.argv) > 4 else 40
    cfg = get_config("st_sage_3")
    torch.manual_seed(0)
    model = models.build_model(cfg.MODEL).cuda()
    model.init(None); model.set_gcn_only(); model.train()
    opt = torch.optim.SGD(model.parameters(), lr=1e-4, momentum=0.9)
    g = synthetic.make_batch(trees, rank=0, device="cuda")
    if os.environ.get("SPGNN_SAMPLE_ON", "cuda") == "cpu":
        g = g.cpu()
    n = g.number_of_nodes()
    rng = np.random.default_rng(0)
    nids = rng.choice(n, int(n * model.node_sample_rate), replace=False)
    sampler = dgl.dataloading.MultiLayerNeighborSampler(list(model.node_ks))
    coll = dgl.dataloading.NodeCollator(g, nids, sampler)

    # leg timings on one fixed mini-batch
    seeds = nids[:node_batch]
    t0 = time.perf_counter()
    for _ in range(iters):
        blocks_h = coll.sample(seeds)
    t_sample = (time.perf_counter() - t0) / iters
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters):
        _, _, blocks = coll.attach(blocks_h, "cuda")
    torch.cuda.synchronize(); t_attach = (time.perf_counter() - t0) / iters

    def step(blocks):
        opt.zero_grad()
        out, _ = model.forward_batch(blocks, blocks[0].srcdata["fvs"])
        F.cross_entropy(out, blocks[-1].dstdata["y"]).backward()
        opt.step()
    for _ in range(5):
        step(blocks)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters):
        step(blocks)
    torch.cuda.synchronize(); t_step = (time.perf_counter() - t0) / iters
    edges = sum(b.number_of_edges() for b in blocks)
    sizes = [(b.number_of_src_nodes(), b.number_of_dst_nodes(), b.number_of_edges()) for b in blocks]

    # the loop as the reference runs it
    dl = dgl.dataloading.NodeDataLoader(g, nids, sampler, device="cuda", batch_size=node_batch, shuffle=True,
                                        drop_last=False, num_workers=workers)
    for _ in dl:
        break
    torch.cuda.synchronize(); t0 = time.perf_counter(); nb = 0; ne = 0
    for epoch in range(max(1, iters // max(1, len(dl)))):
        for _, _, bl in dl:
            step(bl); nb += 1; ne += sum(b.number_of_edges() for b in bl)
    torch.cuda.synchronize(); t_loop = (time.perf_counter() - t0) / nb
    print(f"trees={trees} nodes={n} seeds/batch={node_batch} workers={workers} blocks(src,dst,E)={sizes}")
    print(f"  sample 4 blocks {t_sample*1e3:.3f} ms | attach features {t_attach*1e3:.3f} ms | device fwd+bwd+sgd {t_step*1e3:.3f} ms "
          f"({edges / t_step / 1e6:.2f} M layer-edges/s)")
    print(f"  dataloader loop: {t_loop*1e3:.3f} ms per mini-batch over {nb} batches ({ne / nb / t_loop / 1e6:.2f} M layer-edges/s)")


if __name__ == "__main__":
    main()
