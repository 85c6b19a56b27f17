"""Cached-embedding reader and batch assembler: the step in front of the hot path.

The reference's CNN stage writes one pickle per scan, ``<DB_PATH>/derived/conv_embedding/<uid>.pkl``
(job_runner.py:796-805): ``fvs`` float64 (n, 1024), ``adj`` uint8 (n, n), ``labels`` uint8 (n,),
``fvs_out`` float64 (n, 22), plus ``ref``, ``all_airway``, ``branch_info``, ``meta`` (unused by the GNN stage).
``ConvEmbeddingDataset`` (dataset.py:24-49) loads them, ``collate_func_nativa`` (utils.py:76-84) turns a list of
dicts into a dict of lists, and the runner uploads every tree separately and builds one DGL graph per tree
(job_runner.py:1872-1882).  Here the batch is assembled once: node arrays are concatenated into pinned host
buffers (one asynchronous copy per array), the edge list / CSC / CSR are built for the whole batch, and the
distance encoding runs on the device.
"""
from __future__ import annotations

import os
import pickle
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import graph as G
from .posenc import anchors_device, anchors_from_cnn_prediction, distance_pos_enc, distance_pos_enc_device

__all__ = ["ConvEmbeddingDataset", "collate_native", "assemble_batch", "write_embedding"]

SCHEMA_KEYS = ("fvs", "adj", "labels", "fvs_out")


class ConvEmbeddingDataset(torch.utils.data.Dataset):
    """Same constructor and item contract as the reference's ConvEmbeddingDataset (dataset.py:24-49)."""

    def __init__(self, archive_path: str, series_uids: Sequence[str], transforms=None, keep_sorted: bool = True):
        import random
        self.keep_sorted = keep_sorted
        self.series_uids = list(series_uids) if keep_sorted else random.sample(list(series_uids), len(series_uids))
        self.uid_indice_map = {uid: idx for idx, uid in enumerate(self.series_uids)}
        self.fe_path = os.path.join(archive_path, "derived", "conv_embedding")
        self.transforms = transforms

    def __len__(self) -> int:
        return len(self.series_uids)

    def __getitem__(self, scan_index: int) -> Dict:
        with open(os.path.join(self.fe_path, f"{self.series_uids[scan_index]}.pkl"), "rb") as fp:
            fe = pickle.load(fp)
        missing = [k for k in SCHEMA_KEYS if k not in fe]
        if missing:
            raise KeyError(f"{self.series_uids[scan_index]}.pkl lacks {missing}")
        return fe


def write_embedding(archive_path: str, uid: str, sample: Dict[str, np.ndarray], meta: Optional[dict] = None) -> str:
    """Write one sample in the reference's on-disk schema (dtypes as job_runner.py:796-803)."""
    d = os.path.join(archive_path, "derived", "conv_embedding")
    os.makedirs(d, exist_ok=True)
    state = {"fvs": np.asarray(sample["fvs"]).astype(np.float64), "adj": np.asarray(sample["adj"]).astype(np.uint8),
             "labels": np.asarray(sample["labels"]).astype(np.uint8),
             "fvs_out": np.asarray(sample["fvs_out"]).astype(np.float64), "meta": meta or {"uid": uid}}
    path = os.path.join(d, f"{uid}.pkl")
    with open(path, "wb") as fp:
        pickle.dump(state, fp)
    return path


def collate_native(batch: List[Dict]) -> Dict:
    """List of dicts -> dict of lists, nested dicts merged the same way (reference utils.py:76-84)."""
    out = {}
    for k in batch[0].keys():
        if isinstance(batch[0][k], dict):
            out[k] = collate_native([b[k] for b in batch])
        else:
            out[k] = [b[k] for b in batch]
    return out


def _pinned(shape, dtype, pin: bool) -> torch.Tensor:
    t = torch.empty(shape, dtype=dtype)
    return t.pin_memory() if pin else t


def assemble_batch(batch, device="cuda", pos_enc_dim: Optional[int] = 39, pin: Optional[bool] = None,
                   graph_mode: str = "all_connected") -> G.TreeGraph:
    """``batch``: a collated dict of lists (``collate_native``) or a list of sample dicts.  Returns the batched
    graph on ``device`` with ndata fvs / fvs_out / y [/ pos_enc / p] and its CSC/CSR already built.  ``graph_mode``: the
    reference's GRAPH_MODE (graph.edges_from_adj; "tree_downstream" = parent -> child edges only, job_runner.py:1334-1336;
    anchors and distances of the positional encoding always follow the undirected tree, as job_runner.py:1759-1777 does)."""
    if isinstance(batch, dict):
        samples = [{k: batch[k][i] for k in SCHEMA_KEYS} for i in range(len(batch["adj"]))]
    else:
        samples = list(batch)
    dev = torch.device(device)
    on_gpu = dev.type == "cuda"
    pin = on_gpu if pin is None else pin
    ns = [int(np.asarray(s["adj"]).shape[0]) for s in samples]
    N = int(sum(ns))
    fv_dim = int(np.asarray(samples[0]["fvs"]).shape[1]); n_cls = int(np.asarray(samples[0]["fvs_out"]).shape[1])
    fvs_h, out_h, y_h = _pinned((N, fv_dim), torch.float32, pin), _pinned((N, n_cls), torch.float32, pin), _pinned((N,), torch.int64, pin)
    # node arrays: one concatenation each, converting float64 -> float32 / uint8 -> int64 while filling the pinned buffers
    np.concatenate([np.asarray(s["fvs"]) for s in samples], axis=0, out=fvs_h.numpy(), casting="same_kind")
    np.concatenate([np.asarray(s["fvs_out"]) for s in samples], axis=0, out=out_h.numpy(), casting="same_kind")
    np.concatenate([np.asarray(s["labels"]).reshape(-1) for s in samples], axis=0, out=y_h.numpy(), casting="unsafe")
    pes = []
    if on_gpu:
        # edge list, CSC and CSR of the whole batch on the device, from the packed adjacency matrices (spgnn_build_csc)
        src, dst, csc, nn_, ne_ = G.build_csc_device([s["adj"] for s in samples], dev, pin=pin, graph_mode=graph_mode)
        g = G.TreeGraph.from_device(src, dst, N, csc, nn_, ne_)
    else:
        srcs, dsts, off = [], [], 0
        for s, n in zip(samples, ns):
            adj = np.asarray(s["adj"])
            u, v = G.edges_from_adj(adj, add_self_loops=True, graph_mode=graph_mode)
            srcs.append(u + off); dsts.append(v + off)
            if pos_enc_dim:                                  # host path (CPU graphs): per-tree anchors + BFS in Python
                anc = anchors_from_cnn_prediction(np.asarray(s["fvs_out"], dtype=np.float32), adj, pos_enc_dim)
                pes.append(distance_pos_enc(adj, anc)[0])
            off += n
        g = G.TreeGraph((np.concatenate(srcs), np.concatenate(dsts)), N, dev)
        g.batch_num_nodes_list = ns
        g.batch_num_edges_list = [int(x.shape[0]) for x in srcs]
    g.ndata["fvs"] = fvs_h.to(dev, non_blocking=True)
    g.ndata["fvs_out"] = out_h.to(dev, non_blocking=True)
    g.ndata["y"] = y_h.to(dev, non_blocking=True)
    g.csc(dev)
    if pos_enc_dim:
        if on_gpu:                                           # anchors and distances for the whole batch on the device:
            pe, _ = distance_pos_enc_device(g, anchors_device(g, g.ndata["fvs_out"], pos_enc_dim))   # two HIP calls, no per-tree BFS
        else:
            pe = torch.from_numpy(np.concatenate(pes))
        g.ndata["pos_enc"] = pe
        g.ndata["p"] = pe
    return g
