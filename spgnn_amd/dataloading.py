"""Neighbour-sampled mini-batches over one batched tree graph.

Replaces the slice of ``dgl.dataloading`` / ``dgl.sampling`` the reference's sampled GraphSAGE training touches
(job_runner.py:1484-1506):

    sampler = dgl.dataloading.MultiLayerNeighborSampler(self.model.node_ks)
    dataloader = dgl.dataloading.NodeDataLoader(batch_g, nids, sampler, device=..., batch_size=NODE_BATCH_SIZE,
                                                shuffle=True, drop_last=False, num_workers=NUM_WORKERS)
    for input_nodes, seeds, blocks in dataloader:
        batch_inputs = blocks[0].srcdata['fvs'];  batch_labels = blocks[-1].dstdata['y']
        batch_outputs, _ = self.model.forward_batch(blocks, batch_inputs)

Sampling is index work on the host, as it is in DGL (its samplers run in the DataLoader's CPU workers); the blocks
it produces are the :class:`spgnn_amd.graph.Block` layout the HIP layers run on unchanged.  The algorithm follows
DGL 0.6's ``BlockSampler.sample_blocks``: walking the layers from the output side, sample at most ``fanout`` in-edges
of every seed uniformly without replacement (all of them when the in-degree is not larger, or when the fanout is
``None`` / -1), compact the frontier with ``to_block``, and let the block's source nodes seed the next layer down.
DGL's own random stream is not reproduced (it is a C++ generator seeded by ``dgl.seed``); the parity contract is on
the block a sample defines — layer outputs on a block equal the oracle's on the same edge list — and on the
sampler's distributional properties (tests/test_sampling.py).
"""
from __future__ import annotations

import queue
import threading
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import _capi
from .graph import Block, DeviceBlock, DeviceCSC, TreeGraph, build_csc_numpy, to_block, _as_np_i64

__all__ = ["seed", "sample_neighbors", "in_subgraph", "BlockSampler", "MultiLayerNeighborSampler",
           "MultiLayerFullNeighborSampler", "NodeCollator", "NodeDataLoader"]

_rng: Optional[np.random.Generator] = None


def seed(val: int) -> None:
    """``dgl.seed`` / ``dgl.random.seed``: reseed the sampler's generator."""
    global _rng
    _rng = np.random.default_rng(int(val))


def _generator() -> np.random.Generator:
    global _rng
    if _rng is None:                                  # follow torch.manual_seed unless seeded explicitly
        _rng = np.random.default_rng(torch.initial_seed() % (2 ** 63))
    return _rng


def _host_in_csc(g: TreeGraph):
    """(indptr, indices, eid) int64 host arrays of the in-edges of ``g``, cached on the graph."""
    cached = getattr(g, "_host_in_csc", None)
    if cached is not None and cached[0] == g.number_of_edges():
        return cached[1]
    a = build_csc_numpy(g._src, g._dst, g.number_of_nodes())
    arrays = (a["indptr"].astype(np.int64), a["indices"].astype(np.int64), a["eid"].astype(np.int64))
    g._host_in_csc = (g.number_of_edges(), arrays)
    return arrays


def _frontier(g: TreeGraph, src, dst, eid) -> TreeGraph:
    f = TreeGraph((src, dst), g.number_of_nodes(), "cpu")
    f.edata = {"_ID": torch.from_numpy(eid)}
    return f


def sample_neighbors(g: TreeGraph, nodes, fanout, edge_dir: str = "in", prob=None, replace: bool = False,
                     generator: Optional[np.random.Generator] = None) -> TreeGraph:
    """``dgl.sampling.sample_neighbors``: a graph over the nodes of ``g`` holding, for every node in ``nodes``, at most
    ``fanout`` of its in-edges chosen uniformly without replacement (all of them if it has no more than that, or if
    ``fanout`` is None / -1).  Edges come out grouped by seed in the order of ``nodes``, in ascending parent edge id
    within a seed; ``edata['_ID']`` holds the parent edge ids."""
    if edge_dir != "in":
        raise ValueError("sample_neighbors: only edge_dir='in' is supported")
    if prob is not None or replace:
        raise ValueError("sample_neighbors: only uniform sampling without replacement is supported")
    nodes = _as_np_i64(nodes)
    n = g.number_of_nodes()
    if nodes.size and (nodes.min() < 0 or nodes.max() >= n):
        raise ValueError("sample_neighbors: seed node id out of range")
    indptr, indices, eid = _host_in_csc(g)
    start = indptr[nodes]
    deg = indptr[nodes + 1] - start
    total = int(deg.sum())
    seg = np.repeat(np.arange(nodes.size, dtype=np.int64), deg)
    first = np.cumsum(deg) - deg                                      # offset of each seed's segment
    pos = np.repeat(start - first, deg) + np.arange(total, dtype=np.int64)   # CSC slot of every candidate edge
    take_all = fanout is None or int(fanout) < 0
    if not take_all and int(fanout) == 0:
        pos = pos[:0]; seg = seg[:0]
    elif not take_all and total and int(deg.max()) > int(fanout):
        rng = generator if generator is not None else _generator()
        keys = rng.random(total)
        order = np.lexsort((keys, seg))                               # by seed, random within a seed
        rank = np.arange(total, dtype=np.int64) - np.repeat(first, deg)
        chosen = np.sort(order[rank < int(fanout)])                   # back to ascending slot order
        pos, seg = pos[chosen], seg[chosen]
    return _frontier(g, indices[pos], nodes[seg], eid[pos])


def sample_block_device(g: TreeGraph, seeds: torch.Tensor, fanout, rng_seed: Optional[int] = None) -> DeviceBlock:
    """``to_block(sample_neighbors(g, seeds, fanout), seeds)`` entirely on the device holding ``g`` (a CUDA graph):
    the HIP sampler (include/spgnn_hip.h spgnn_sample_neighbors / spgnn_block_relabel) reads the graph's resident
    CSC, and the block's own CSC / CSR are assembled from its output with device scans and one sort; the host reads
    back two integers (edge and source counts) per block.  The block's dst nodes are ``seeds`` in order; its other
    source nodes follow in ascending parent id (DGL's order among them is an artefact of its hash map; no layer
    depends on it).  Sampled edges keep CSC (ascending parent edge id) order within a seed, like the host sampler."""
    if g.device.type != "cuda":
        raise ValueError("sample_block_device: the graph must live on a GPU (use sample_neighbors + to_block on the host)")
    dev = g.device
    lib = _capi.load()
    csc = g.csc(dev)
    N = csc.num_nodes
    seeds = torch.as_tensor(seeds, dtype=torch.int64, device=dev).contiguous()
    S = int(seeds.shape[0])
    fan = -1 if fanout is None or int(fanout) < 0 else int(fanout)
    if rng_seed is None:
        rng_seed = int(_generator().integers(0, 2 ** 62))
    stream = torch.cuda.current_stream(dev).cuda_stream
    i32 = dict(dtype=torch.int32, device=dev)
    if S and N == 0:
        raise ValueError("sample_block_device: seeds given for an empty graph")
    safe = seeds.clamp(0, max(N - 1, 0))                            # out-of-range ids are reported below, never dereferenced
    deg = csc.indptr[safe + 1] - csc.indptr[safe]
    cnt = deg if fan < 0 else deg.clamp(max=fan)
    out_indptr = torch.zeros(S + 1, **i32)
    torch.cumsum(cnt, 0, dtype=torch.int32, out=out_indptr[1:])
    # room for every edge the kernel can emit: fanout per seed, or (all in-edges) the exact count, read back first
    bound = S * fan if fan >= 0 else int(out_indptr[-1])
    out_src = torch.empty(max(bound, 1), **i32)                     # (never a null pointer across the C ABI)
    out_eid = torch.empty(max(bound, 1), **i32)
    local = torch.full((N,), -1, **i32)
    flag = torch.zeros(N, **i32)
    _capi.check(lib.spgnn_sample_neighbors(csc.indptr.data_ptr(), csc.indices.data_ptr(), csc.eid.data_ptr(), N, safe.data_ptr(),
                                           S, fan, out_indptr.data_ptr(), rng_seed, local.data_ptr(), out_src.data_ptr(),
                                           out_eid.data_ptr(), flag.data_ptr(), stream), "spgnn_sample_neighbors")
    rank = torch.cumsum(flag, 0, dtype=torch.int32)
    ar = torch.arange(S, **i32)
    bad = (local[safe] != ar).any() if S else torch.zeros((), dtype=torch.bool, device=dev)
    oob = (safe != seeds).any() if S else bad
    E, n_extra, bad, min_in = torch.stack([out_indptr[-1], rank[-1] if N else out_indptr[-1], (bad | oob).to(torch.int32),
                                           cnt.min() if S else out_indptr[-1]]).tolist()       # the one host read
    if bad:
        raise ValueError("sample_block_device: seed nodes must be distinct ids of the graph")
    extra_buf = torch.empty(max(n_extra, 1), dtype=torch.int64, device=dev)
    extra = extra_buf[:n_extra]
    src_local = torch.empty(E, **i32)
    _capi.check(lib.spgnn_block_relabel(flag.data_ptr(), rank.data_ptr(), local.data_ptr(), N, S, out_src.data_ptr(), E,
                                        extra_buf.data_ptr(), src_local.data_ptr(), stream), "spgnn_block_relabel")
    num_src = S + n_extra
    dst_local = torch.repeat_interleave(ar, cnt.long(), output_size=E)
    order = torch.sort(src_local, stable=True)                      # src-major view for the backward kernels
    tensors = dict(
        indptr=torch.cat([out_indptr, out_indptr[-1:].expand(n_extra)]).contiguous(),
        indices=src_local,
        eid=torch.arange(E, **i32),
        out_indptr=torch.searchsorted(order.values, torch.arange(num_src + 1, **i32), out_int32=True),
        out_indices=dst_local[order.indices],
        out_pos=order.indices.to(torch.int32),
    )
    dcsc = DeviceCSC.from_tensors(tensors, num_src, E, min_in_degree=min_in if S else 0)
    return DeviceBlock(dcsc, S, torch.cat([seeds, extra]), seeds, out_eid[:E].long())


def in_subgraph(g: TreeGraph, nodes) -> TreeGraph:
    """``dgl.in_subgraph``: all in-edges of ``nodes``."""
    return sample_neighbors(g, nodes, None)


class BlockSampler:
    """``dgl.dataloading.BlockSampler``: ``sample_blocks`` builds the list of blocks from ``sample_frontier``."""

    def __init__(self, num_layers: int, return_eids: bool = False):
        self.num_layers = int(num_layers)
        self.return_eids = return_eids

    def sample_frontier(self, block_id: int, g: TreeGraph, seed_nodes) -> TreeGraph:
        raise NotImplementedError

    def sample_blocks(self, g: TreeGraph, seed_nodes) -> List[Block]:
        blocks: List[Block] = []
        if g.device.type == "cuda" and type(self).sample_frontier is MultiLayerNeighborSampler.sample_frontier:
            # graph resident on a GPU: sample there (sample_block_device); the seeds never leave the device
            for block_id in reversed(range(self.num_layers)):
                block = sample_block_device(g, seed_nodes, self.fanouts[block_id])
                seed_nodes = block.srcdata["_ID"]
                blocks.insert(0, block)
            return blocks
        seed_nodes = _as_np_i64(seed_nodes)
        for block_id in reversed(range(self.num_layers)):
            frontier = self.sample_frontier(block_id, g, seed_nodes)
            block = to_block(frontier, seed_nodes)
            seed_nodes = block.srcdata["_ID"].numpy()
            blocks.insert(0, block)
        return blocks


class MultiLayerNeighborSampler(BlockSampler):
    """``dgl.dataloading.MultiLayerNeighborSampler(fanouts)`` — ``fanouts[i]`` in-edges per node for layer ``i``
    (the reference passes ``model.node_ks``, models.py:665)."""

    def __init__(self, fanouts: Sequence[Optional[int]], replace: bool = False, return_eids: bool = False):
        super().__init__(len(fanouts), return_eids)
        if replace:
            raise ValueError("MultiLayerNeighborSampler: sampling with replacement is not supported")
        self.fanouts = list(fanouts)

    def sample_frontier(self, block_id, g, seed_nodes):
        return sample_neighbors(g, seed_nodes, self.fanouts[block_id])


class MultiLayerFullNeighborSampler(MultiLayerNeighborSampler):
    """``dgl.dataloading.MultiLayerFullNeighborSampler(n_layers)``: every in-edge at every layer."""

    def __init__(self, n_layers: int, return_eids: bool = False):
        super().__init__([None] * int(n_layers), return_eids=return_eids)


class NodeCollator:
    """``dgl.dataloading.NodeCollator``: ``collate(seeds)`` -> (input_nodes, output_nodes, blocks) with the parent's
    node data gathered into ``blocks[0].srcdata`` and ``blocks[-1].dstdata``."""

    def __init__(self, g: TreeGraph, nids, block_sampler: BlockSampler):
        self.g, self.nids, self.block_sampler = g, _as_np_i64(nids), block_sampler

    @property
    def dataset(self):
        return self.nids

    def sample(self, seeds) -> List[Block]:
        return self.block_sampler.sample_blocks(self.g, seeds)

    def attach(self, blocks: List[Block], device=None):
        """Move the blocks to ``device`` and gather the parent's node data for the outermost src / dst nodes."""
        g = self.g
        device = torch.device(device) if device is not None else g.device
        input_nodes = blocks[0].srcdata["_ID"]
        output_nodes = blocks[-1].dstdata["_ID"]
        blocks = [b.to(device) for b in blocks]
        in_idx, out_idx = input_nodes.to(g.device), output_nodes.to(g.device)
        for k, v in g.ndata.items():
            blocks[0].srcdata[k] = v.index_select(0, in_idx).to(device)
            blocks[-1].dstdata[k] = v.index_select(0, out_idx).to(device)
        return blocks[0].srcdata["_ID"], blocks[-1].dstdata["_ID"], blocks

    def collate(self, seeds, device=None):
        return self.attach(self.sample(seeds), device)


class NodeDataLoader:
    """``dgl.dataloading.NodeDataLoader(g, nids, block_sampler, device=, batch_size=, shuffle=, drop_last=,
    num_workers=)``: iterating yields ``(input_nodes, output_nodes, blocks)`` per mini-batch of seed nodes.

    ``num_workers > 0`` samples ahead on a host thread (numpy index work) while the device runs the previous
    mini-batch; graph transfer and feature gathers stay on the iterating thread's stream."""

    def __init__(self, g: TreeGraph, nids, block_sampler: BlockSampler, device="cpu", batch_size: int = 1,
                 shuffle: bool = False, drop_last: bool = False, num_workers: int = 0, **kwargs):
        if kwargs:
            raise TypeError(f"NodeDataLoader: unsupported arguments {sorted(kwargs)}")
        if int(batch_size) <= 0:
            raise ValueError("batch_size should be a positive integer value")
        self.collator = NodeCollator(g, nids, block_sampler)
        self.device = device
        self.batch_size, self.shuffle, self.drop_last = int(batch_size), bool(shuffle), bool(drop_last)
        self.num_workers = int(num_workers)

    def __len__(self) -> int:
        n = self.collator.nids.size
        return n // self.batch_size if self.drop_last else -(-n // self.batch_size)

    def _seed_batches(self):
        nids = self.collator.nids
        if self.shuffle:
            nids = nids[torch.randperm(nids.size).numpy()]          # torch's generator, like torch's DataLoader
        for i in range(len(self)):
            yield nids[i * self.batch_size: (i + 1) * self.batch_size]

    def _iter_device(self):
        """Graph resident on a GPU: the blocks of mini-batch i+1 are sampled on a side stream while the caller's
        stream still runs mini-batch i, so the sampler's per-block host read (two integers) waits for the sampling
        kernels only, never for the training step queued ahead of it."""
        g = self.collator.g
        main = torch.cuda.current_stream(g.device)
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream(g.device)
        side = self._side
        side.wait_stream(main)                                       # the graph's CSC may have just been uploaded

        def launch(seeds):
            with torch.cuda.stream(side):
                blocks = self.collator.sample(seeds)
                done = torch.cuda.Event()
                done.record(side)
            return blocks, done

        batches = self._seed_batches()
        pending = None
        for seeds in batches:
            ahead = launch(seeds)
            if pending is not None:
                yield self._hand_over(pending, main)
            pending = ahead
        if pending is not None:
            yield self._hand_over(pending, main)

    def _hand_over(self, pending, main):
        blocks, done = pending
        main.wait_event(done)
        for b in blocks:
            if hasattr(b, "record_stream"):
                b.record_stream(main)
        return self.collator.attach(blocks, self.device)

    def __iter__(self):
        if self.collator.g.device.type == "cuda" and isinstance(self.collator.block_sampler, MultiLayerNeighborSampler):
            yield from self._iter_device()
            return
        if self.num_workers <= 0:
            for seeds in self._seed_batches():
                yield self.collator.collate(seeds, self.device)
            return
        q: "queue.Queue" = queue.Queue(maxsize=2 * self.num_workers)
        stop = threading.Event()

        def produce():
            try:
                for seeds in self._seed_batches():
                    if stop.is_set():
                        return
                    q.put(("ok", self.collator.sample(seeds)))
                q.put(("end", None))
            except BaseException as e:                               # surfaced on the consumer side
                q.put(("err", e))

        t = threading.Thread(target=produce, daemon=True)
        t.start()
        try:
            while True:
                kind, item = q.get()
                if kind == "end":
                    break
                if kind == "err":
                    raise item
                yield self.collator.attach(item, self.device)
        finally:
            stop.set()
            while t.is_alive():                                      # unblock a producer waiting on a full queue
                try:
                    q.get_nowait()
                except queue.Empty:
                    pass
                t.join(timeout=0.01)
