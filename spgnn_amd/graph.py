"""Batched airway-tree graph container (host side of the hot path).

Replaces the slice of the DGL graph API the reference touches:
``DGLGraph(nx_graph)``, ``add_edges``, ``remove_self_loop``, ``batch``/``unbatch``,
``ndata``, ``in_degrees``, ``adjacency_matrix``, ``to_networkx``.

The edge list follows the reference's construction rule bit-exactly
(reference job_runner.py:1319-1344 and 1779-1801, batching at 1390/1882):

* every directed pair (u, v), u != v, adj[u, v] != 0, sorted by (u, v);
* then n self loops (i, i) appended last;
* ``batch`` offsets node ids by the running node count and concatenates the
  per-tree edge lists in order.

For the HIP kernels the edge list is turned (once per loader batch, reused for the
300 inner steps, reference job_runner.py:1892) into int32 device arrays:

``indptr[N+1], indices[E], eid[E]``
    dst-major CSC; the in-edges of v are ``indices[indptr[v]:indptr[v+1]]`` in
    ascending edge id (stable), which is the summation order DGL's COO->CSC gives.
``out_indptr[N+1], out_indices[E], out_pos[E]``
    src-major CSR; ``out_pos`` is the CSC slot of the same edge, so per-edge
    arrays kept in CSC order (attention, its gradient) are addressed from
    either side without a second copy.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

__all__ = [
    "edges_from_adj", "TreeGraph", "DGLGraph", "batch", "unbatch", "remove_self_loop",
    "to_networkx", "graph_from_adj", "DeviceCSC", "build_csc_numpy", "build_csc_device", "Block", "DeviceBlock", "to_block",
]


GRAPH_MODES = ("all_connected", "tree_downstream")


def edges_from_adj(adj: np.ndarray, add_self_loops: bool = True, graph_mode: str = "all_connected"):
    """Directed edge list of one tree by the reference rule (job_runner.py:1779-1801).

    ``nx.DiGraph(adj)`` enumerates non-zero entries row by row, i.e. sorted by (u, v);
    ``dgl.remove_self_loop`` drops the diagonal keeping order; ``g.add_edges(nodes, nodes)``
    appends (i, i) for i = 0..n-1.

    ``graph_mode`` = the reference's ``GRAPH_MODE`` setting (job_runner.py:1329-1339, GCNTrain.from_adj_to_graph):
    "all_connected" (every config of the reference) keeps both directions of every tree edge; "tree_downstream" keeps the
    upper triangle only - ``nx.DiGraph(np.triu(adj))``: parent -> child edges, the parent having the smaller index.  An
    ``adj`` that is already upper triangular gives the same directed graph in either mode (job_runner.py:1329-1332).
    """
    if graph_mode not in GRAPH_MODES:
        raise ValueError(f"graph_mode must be one of {GRAPH_MODES}, got {graph_mode!r}")
    a = np.asarray(adj)
    if a.ndim != 2 or a.shape[0] != a.shape[1]:
        raise ValueError(f"adj must be square, got {a.shape}")
    if graph_mode == "tree_downstream":
        a = np.triu(a)
    n = a.shape[0]
    u, v = np.nonzero(a)                 # row-major == sorted by (u, v)
    keep = u != v
    u, v = u[keep], v[keep]
    if add_self_loops:
        loops = np.arange(n, dtype=u.dtype)
        u = np.concatenate([u, loops])
        v = np.concatenate([v, loops])
    return u.astype(np.int64), v.astype(np.int64)


def build_csc_numpy(src: np.ndarray, dst: np.ndarray, num_nodes: int):
    """Stable COO -> CSC/CSR in numpy. Returns dict of int32 arrays."""
    E = int(src.shape[0])
    if E >= 2 ** 31 or num_nodes >= 2 ** 31:
        raise ValueError("graph too large for int32 indexing")
    order = np.argsort(dst, kind="stable")
    indptr = np.zeros(num_nodes + 1, dtype=np.int64)
    np.add.at(indptr, dst + 1, 1)
    indptr = np.cumsum(indptr)
    indices = src[order]
    eid = order
    csc_pos_of_edge = np.empty(E, dtype=np.int64)
    csc_pos_of_edge[order] = np.arange(E)
    oorder = np.argsort(src, kind="stable")
    out_indptr = np.zeros(num_nodes + 1, dtype=np.int64)
    np.add.at(out_indptr, src + 1, 1)
    out_indptr = np.cumsum(out_indptr)
    out_indices = dst[oorder]
    out_pos = csc_pos_of_edge[oorder]
    i32 = lambda x: np.ascontiguousarray(x, dtype=np.int32)
    return dict(indptr=i32(indptr), indices=i32(indices), eid=i32(eid),
                out_indptr=i32(out_indptr), out_indices=i32(out_indices), out_pos=i32(out_pos))


class DeviceCSC:
    """int32 CSC/CSR arrays of a batched graph resident on one device."""

    def __init__(self, arrays: Dict[str, np.ndarray], num_nodes: int, num_edges: int, device):
        self.num_nodes = int(num_nodes)
        self.num_edges = int(num_edges)
        self.device = torch.device(device)
        for k, a in arrays.items():
            setattr(self, k, torch.from_numpy(a).to(self.device))
        ind = arrays["indptr"].astype(np.int64)
        oind = arrays["out_indptr"].astype(np.int64)
        self.max_in_degree = int((ind[1:] - ind[:-1]).max()) if num_nodes else 0
        self.max_out_degree = int((oind[1:] - oind[:-1]).max()) if num_nodes else 0
        self.min_in_degree = int((ind[1:] - ind[:-1]).min()) if num_nodes else 0
        self.min_out_degree = int((oind[1:] - oind[:-1]).min()) if num_nodes else 0
        # degree-derived edge weights are computed lazily by ops (GraphConv/GIN)
        self._cache: Dict[str, torch.Tensor] = {}

    @classmethod
    def from_tensors(cls, tensors: Dict[str, torch.Tensor], num_nodes: int, num_edges: int, min_in_degree: int = 1,
                     max_in_degree: int = 0, max_out_degree: int = 0) -> "DeviceCSC":
        """Wrap int32 index arrays that were built on the device (dataloading's device sampler); the degree bounds
        are whatever the builder knows without a host read."""
        self = cls.__new__(cls)
        self.num_nodes, self.num_edges = int(num_nodes), int(num_edges)
        for k in ("indptr", "indices", "eid", "out_indptr", "out_indices", "out_pos"):
            t = tensors[k]
            if t.dtype != torch.int32 or not t.is_contiguous():
                raise ValueError(f"DeviceCSC.from_tensors: {k} must be contiguous int32")
            setattr(self, k, t)
        if self.indptr.shape[0] != self.num_nodes + 1 or self.out_indptr.shape[0] != self.num_nodes + 1:
            raise ValueError("DeviceCSC.from_tensors: indptr length != num_nodes + 1")
        for k in ("indices", "eid", "out_indices", "out_pos"):
            if getattr(self, k).shape[0] != self.num_edges:
                raise ValueError(f"DeviceCSC.from_tensors: {k} length != num_edges")
        self.device = self.indptr.device
        self.min_in_degree, self.max_in_degree, self.max_out_degree = int(min_in_degree), int(max_in_degree), int(max_out_degree)
        self.min_out_degree = 0                 # unknown without a host read (blocks: sources without out-edges exist)
        self._cache = {}
        return self

    def ell(self):
        """(nbr8, out_nbr8, out_pos8): the in- / out-neighbour lists once more as padded (N, 8) int32 rows,
        ``nbr8[v, k] = indices[indptr[v] + min(k, deg(v) - 1)]`` (and the same over out_indices / out_pos), built on the
        device once per batch.  The GAT kernels fetch them with indptr[v] instead of after it (include/spgnn_hip.h,
        spgnn_gat_fwd); rows of nodes with more than 8 edges are never used."""
        if "ell" not in self._cache:
            def rows(ptr, *arrays):
                if self.num_nodes == 0 or self.num_edges == 0:
                    return tuple(torch.zeros((self.num_nodes, 8), dtype=torch.int32, device=ptr.device) for _ in arrays)
                if ptr.is_cuda and all(a.dtype == torch.int32 and a.is_contiguous() for a in (ptr,) + arrays):
                    from . import _capi                                  # one launch (spgnn_ell_rows)
                    outs = tuple(torch.empty((self.num_nodes, 8), dtype=torch.int32, device=ptr.device) for _ in arrays)
                    with torch.cuda.device(ptr.device):
                        _capi.check(_capi.load().spgnn_ell_rows(ptr.data_ptr(), arrays[0].data_ptr(),
                                                                arrays[1].data_ptr() if len(arrays) > 1 else 0, self.num_nodes,
                                                                self.num_edges, outs[0].data_ptr(),
                                                                outs[1].data_ptr() if len(arrays) > 1 else 0,
                                                                torch.cuda.current_stream(ptr.device).cuda_stream), "spgnn_ell_rows")
                    return outs
                deg = (ptr[1:] - ptr[:-1]).to(torch.int64)         # (host graphs: the same rows by torch indexing)
                k = torch.arange(8, device=ptr.device, dtype=torch.int64)
                pos = ptr[:-1].to(torch.int64)[:, None] + torch.minimum(k[None, :], (deg - 1).clamp(min=0)[:, None])
                pos.clamp_(max=self.num_edges - 1)
                return tuple(a[pos].contiguous() for a in arrays)
            (nbr8,) = rows(self.indptr, self.indices)
            out_nbr8, out_pos8 = rows(self.out_indptr, self.out_indices, self.out_pos)
            self._cache["ell"] = (nbr8, out_nbr8, out_pos8)
        return self._cache["ell"]

    def tiles(self, cap: int, rebuild: bool = False):
        """Node tiles for the tree-resident LDS kernels (csrc/spgnn_tile.hip): -> (tile_ptr int32 device tensor, n_tiles).
        Runs of consecutive nodes of at most ``cap`` nodes each, cut at the batch's tree boundaries (``self.segments``: nodes
        per tree, attached by TreeGraph.csc; a tree is closed under neighbours - dgl.batch, reference job_runner.py:1882) so
        that in the normal case every gather of a tile is served from LDS; a tree larger than ``cap`` is split (the kernels
        load out-of-tile neighbours from global memory), a graph without boundaries is cut uniformly.  The tensor has the
        FIXED length 2 ceil(N / cap) + 3 - greedy packing never needs more tiles, since two consecutive tiles together
        exceed ``cap`` - with empty tiles at the end, so a batch arena can rewrite it in place under a captured step."""
        key = ("tiles", int(cap))
        if key in self._cache and not rebuild:
            return self._cache[key]
        N = self.num_nodes
        n_max = 2 * ((N + cap - 1) // cap) + 2
        bounds, cur, pos = [0], 0, 0
        segs = getattr(self, "segments", None)
        if not segs or sum(segs) != N:
            segs = [N]
        for n in segs:
            n = int(n)
            while n > 0:
                take = min(n, cap)
                if cur + take > cap:
                    bounds.append(pos)
                    cur = 0
                cur += take
                pos += take
                n -= take
        if pos > bounds[-1]:
            bounds.append(pos)
        n_tiles = len(bounds) - 1
        assert n_tiles <= n_max and all(b1 - b0 <= cap for b0, b1 in zip(bounds, bounds[1:]))
        bounds += [N] * (n_max + 1 - len(bounds))
        t = torch.tensor(bounds, dtype=torch.int32).to(self.device)
        old = self._cache.get(key)
        if old is not None and old[0].shape == t.shape and rebuild:
            old[0].copy_(t)                      # same address: what a captured step reads
            t = old[0]
        # the launch grid always covers n_max tiles when the storage may be rewritten under a capture; else the used ones
        self._cache[key] = (t, n_max if getattr(self, "_fixed_tile_count", False) else n_tiles)
        return self._cache[key]

    def in_degrees_f(self) -> torch.Tensor:
        if "in_deg" not in self._cache:
            self._cache["in_deg"] = (self.indptr[1:] - self.indptr[:-1]).to(torch.float32)
        return self._cache["in_deg"]

    def out_degrees_f(self) -> torch.Tensor:
        if "out_deg" not in self._cache:
            self._cache["out_deg"] = (self.out_indptr[1:] - self.out_indptr[:-1]).to(torch.float32)
        return self._cache["out_deg"]

    def degree_scale(self, which: str, power: float) -> torch.Tensor:
        """clamp(in- or out-degree, min=1) ** power, cached: the degree normalisations of GraphConv ('both': -0.5, 'right' /
        'left': -1), GIN-mean and SAGE-mean are properties of the graph, not of the step (three launches each, every layer)."""
        key = ("deg_scale", which, float(power))
        if key not in self._cache:
            d = (self.in_degrees_f() if which == "in" else self.out_degrees_f()).clamp(min=1)
            self._cache[key] = d.pow(power) if power != -1.0 else 1.0 / d
        return self._cache[key]


def build_csc_device(adjs: Sequence[np.ndarray], device, pin: bool = True, graph_mode: str = "all_connected"):
    """Edge list + CSC + CSR of a loader batch built ON THE DEVICE from the trees' adjacency matrices (reference rule:
    job_runner.py:1779-1801 + dgl.batch, restated in csrc/spgnn_graph.hip): the n x n uint8 matrices are packed into one
    (pinned) buffer, uploaded once, and two kernels + two prefix sums produce every index array; the host reads back five
    integers (E and the degree bounds).  -> (src, dst int32 device tensors, DeviceCSC, nodes per tree, edges per tree)."""
    from . import _capi
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError("build_csc_device needs a ROCm device (host graphs: edges_from_adj + build_csc_numpy)")
    lib = _capi.load()
    # any non-zero entry is an edge (nx.DiGraph(adj) in the reference, edges_from_adj here): a plain uint8 cast would turn
    # 0.5 or 256 into "no edge" and the device path would disagree with the host path
    # (the kernels test bytes against zero, so uint8 / bool matrices go up as they are)
    if graph_mode not in GRAPH_MODES:
        raise ValueError(f"graph_mode must be one of {GRAPH_MODES}, got {graph_mode!r}")
    if graph_mode == "tree_downstream":      # the reference's nx.DiGraph(np.triu(adj)) (job_runner.py:1334-1336): the kernels see the triangle
        adjs = [np.triu(np.asarray(a)) for a in adjs]
    mats = [np.ascontiguousarray(a).view(np.uint8) if (isinstance(a, np.ndarray) and a.dtype in (np.uint8, np.bool_))
            else np.ascontiguousarray(np.asarray(a) != 0).view(np.uint8) for a in adjs]
    for a in mats:
        if a.ndim != 2 or a.shape[0] != a.shape[1]:
            raise ValueError(f"adj must be square, got {a.shape}")
    ns = np.array([a.shape[0] for a in mats], dtype=np.int64)
    T, N = len(mats), int(ns.sum())
    tree_ptr = np.zeros(T + 1, dtype=np.int64); np.cumsum(ns, out=tree_ptr[1:])
    adj_ptr = np.zeros(T + 1, dtype=np.int64); np.cumsum(ns * ns, out=adj_ptr[1:])
    packed = torch.empty((int(adj_ptr[-1]),), dtype=torch.uint8)
    if pin and packed.numel():
        packed = packed.pin_memory()
    if T:
        np.concatenate([a.reshape(-1) for a in mats], out=packed.numpy())
    with torch.cuda.device(dev):
        st = torch.cuda.current_stream(dev).cuda_stream
        adj_d = packed.to(dev, non_blocking=True)
        tp_d, ap_d = torch.from_numpy(tree_ptr).to(dev), torch.from_numpy(adj_ptr).to(dev)
        rc = torch.empty((N,), dtype=torch.int32, device=dev)
        cc = torch.empty((N,), dtype=torch.int32, device=dev)
        _capi.check(lib.spgnn_build_csc_count(adj_d.data_ptr(), ap_d.data_ptr(), tp_d.data_ptr(), T, rc.data_ptr(), cc.data_ptr(), st),
                    "spgnn_build_csc_count")
        zero = torch.zeros((1,), dtype=torch.int64, device=dev)
        row_start = torch.cat([zero, torch.cumsum(rc, 0, dtype=torch.int64)])
        col_start = torch.cat([zero, torch.cumsum(cc, 0, dtype=torch.int64)])
        if N:
            stats = torch.stack([row_start[-1], col_start[-1], cc.min().long(), cc.max().long(), rc.min().long(), rc.max().long()]).tolist()
        else:
            stats = [0, 0, -1, -1, -1, -1]
        if stats[0] != stats[1]:
            raise RuntimeError("build_csc_device: row and column counts disagree (internal error)")
        E = int(stats[0]) + N
        if E >= 2 ** 31:
            raise ValueError("graph too large for int32 indexing")
        i32 = lambda n: torch.empty((n,), dtype=torch.int32, device=dev)
        src, dst = i32(E), i32(E)
        t = dict(indptr=i32(N + 1), indices=i32(E), eid=i32(E), out_indptr=i32(N + 1), out_indices=i32(E), out_pos=i32(E))
        if T == 0:
            t["indptr"].zero_(); t["out_indptr"].zero_()
        _capi.check(lib.spgnn_build_csc(adj_d.data_ptr(), ap_d.data_ptr(), tp_d.data_ptr(), T, row_start.data_ptr(), col_start.data_ptr(),
                                        src.data_ptr(), dst.data_ptr(), t["indptr"].data_ptr(), t["indices"].data_ptr(), t["eid"].data_ptr(),
                                        t["out_indptr"].data_ptr(), t["out_indices"].data_ptr(), t["out_pos"].data_ptr(), N, E, st),
                    "spgnn_build_csc")
    csc = DeviceCSC.from_tensors(t, N, E, min_in_degree=stats[2] + 1, max_in_degree=stats[3] + 1, max_out_degree=stats[5] + 1)
    csc.min_out_degree = stats[4] + 1
    # per-tree edge counts: off-diagonal non-zeros + one self loop per node (host arithmetic on T + 1 numbers read back once)
    rs = row_start[torch.from_numpy(tree_ptr).to(dev)].cpu().numpy() if T else np.zeros(1, dtype=np.int64)
    edges_per_tree = (rs[1:] - rs[:-1] + ns).tolist()
    return src, dst, csc, ns.tolist(), edges_per_tree


class _NData(dict):
    """``g.ndata`` — a dict of node tensors that checks the leading dimension."""

    def __init__(self, graph):
        super().__init__()
        self._g = graph

    def __setitem__(self, key, value):
        if not torch.is_tensor(value):
            value = torch.as_tensor(value)
        if value.shape[0] != self._g.number_of_nodes():
            raise ValueError(f"ndata['{key}'] has {value.shape[0]} rows, graph has "
                             f"{self._g.number_of_nodes()} nodes")
        if value.device != self._g.device:
            value = value.to(self._g.device)
        super().__setitem__(key, value)
        self._invalidate()

    def __delitem__(self, key):
        super().__delitem__(key)
        self._invalidate()

    def pop(self, key, *default):
        self._invalidate()
        return super().pop(key, *default)

    def update(self, *args, **kwargs):
        for k, v in dict(*args, **kwargs).items():
            self[k] = v

    def _invalidate(self):
        # per-batch constants derived from node data (models._data_cat / _data_aligned / _data_in) are keyed by the
        # source tensors' address and version: a replaced tensor may be handed the freed block's address again, so
        # every (re)assignment drops them
        cache = getattr(self._g, "_tensor_cache", None)
        if cache:
            cache.clear()
        builders = getattr(self._g, "_derived_builders", None)
        if builders:
            builders.clear()


class TreeGraph:
    """A (batched) directed graph in DGL edge order with node data.

    Mirrors the ``DGLGraph`` calls the reference makes.
    """

    def __init__(self, data=None, num_nodes: Optional[int] = None, device="cpu"):
        self.device = torch.device(device)
        if data is None:
            src = np.zeros(0, np.int64); dst = np.zeros(0, np.int64); n = num_nodes or 0
        elif isinstance(data, tuple):
            src = np.asarray(data[0], dtype=np.int64); dst = np.asarray(data[1], dtype=np.int64)
            n = num_nodes if num_nodes is not None else (int(max(src.max(initial=-1), dst.max(initial=-1))) + 1)
        else:
            src, dst, n = _edges_from_networkx(data)
        self._src_np, self._dst_np, self._n = src, dst, int(n)
        self._edges_dev = None                                 # (src, dst) int32 device tensors of a graph built on the device
        self._num_edges = int(src.shape[0])
        self.batch_num_nodes_list: List[int] = [self._n]
        self.batch_num_edges_list: List[int] = [int(src.shape[0])]
        self.ndata = _NData(self)
        self._csc: Dict[str, DeviceCSC] = {}
        self._tensor_cache: Dict[tuple, torch.Tensor] = {}     # per-batch constants derived from node data
        self._derived_builders: Dict[tuple, object] = {}       # ... and how each is rebuilt (models._derived)
        self._stable_storage = False                           # True on a batch arena's graph (arena.BatchArena)

    is_block = False      # Block (below) is the bipartite message-flow graph of neighbour-sampled training

    # host copies of the edge list: a graph assembled on the device (build_csc_device) downloads them on first use only
    def _host_edges(self):
        if self._src_np is None:
            if self._edges_dev is None:
                raise RuntimeError("this graph's edge list was not kept (a per-scan inference arena, BatchArena.keep_edges = False): "
                                   "its CSC / CSR index arrays are current, src / dst in edge-id order are not")
            s_, d_ = self._edges_dev
            self._src_np, self._dst_np = s_.cpu().numpy().astype(np.int64), d_.cpu().numpy().astype(np.int64)
        return self._src_np, self._dst_np

    @property
    def _src(self):
        return self._host_edges()[0]

    @_src.setter
    def _src(self, v):
        self._src_np = v
        self._num_edges = int(v.shape[0])
        self._edges_dev = None

    @property
    def _dst(self):
        return self._host_edges()[1]

    @_dst.setter
    def _dst(self, v):
        self._dst_np = v
        self._edges_dev = None

    @classmethod
    def from_device(cls, src: torch.Tensor, dst: torch.Tensor, num_nodes: int, csc: "DeviceCSC", batch_num_nodes: Sequence[int],
                    batch_num_edges: Sequence[int]) -> "TreeGraph":
        """A batched graph whose edge list and index structures were built on the device (:func:`build_csc_device`)."""
        g = cls(None, num_nodes, src.device)
        g._src_np = g._dst_np = None
        g._edges_dev = (src, dst)
        g._num_edges = int(src.shape[0])
        g.batch_num_nodes_list = [int(x) for x in batch_num_nodes]
        g.batch_num_edges_list = [int(x) for x in batch_num_edges]
        dev = src.device
        if dev.type == "cuda" and dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        g._csc[str(dev)] = csc
        return g

    # ---- structure -------------------------------------------------------------------
    def number_of_nodes(self) -> int:
        return self._n

    num_nodes = number_of_nodes
    number_of_src_nodes = num_src_nodes = number_of_nodes
    number_of_dst_nodes = num_dst_nodes = number_of_nodes

    def number_of_edges(self) -> int:
        return self._num_edges

    num_edges = number_of_edges

    def nodes(self) -> torch.Tensor:
        return torch.arange(self._n, dtype=torch.int64, device=self.device)

    def edges(self):
        if self._edges_dev is not None and self._edges_dev[0].device == self.device:
            return self._edges_dev[0].long(), self._edges_dev[1].long()
        return (torch.from_numpy(self._src).to(self.device), torch.from_numpy(self._dst).to(self.device))

    def add_edges(self, u, v) -> None:
        """Append edges in place (reference: ``g.add_edges(g.nodes(), g.nodes())``)."""
        u = _as_np_i64(u); v = _as_np_i64(v)
        if u.shape != v.shape:
            raise ValueError("add_edges: u and v differ in length")
        if u.size and (max(u.max(), v.max()) >= self._n or min(u.min(), v.min()) < 0):
            raise ValueError("add_edges: node id out of range")
        self._src = np.concatenate([self._src, u]); self._dst = np.concatenate([self._dst, v])
        if len(self.batch_num_edges_list) == 1:
            self.batch_num_edges_list = [int(self._src.shape[0])]
        else:  # DGL resets batch info on mutation of a batched graph
            self.batch_num_nodes_list = [self._n]; self.batch_num_edges_list = [int(self._src.shape[0])]
        self._csc.clear()

    @property
    def batch_size(self) -> int:
        return len(self.batch_num_nodes_list)

    def batch_num_nodes(self) -> torch.Tensor:
        return torch.tensor(self.batch_num_nodes_list, dtype=torch.int64)

    def batch_num_edges(self) -> torch.Tensor:
        return torch.tensor(self.batch_num_edges_list, dtype=torch.int64)

    def in_degrees(self) -> torch.Tensor:
        return torch.from_numpy(np.bincount(self._dst, minlength=self._n).astype(np.int64)).to(self.device)

    def out_degrees(self) -> torch.Tensor:
        return torch.from_numpy(np.bincount(self._src, minlength=self._n).astype(np.int64)).to(self.device)

    # ---- device ----------------------------------------------------------------------
    def to(self, device) -> "TreeGraph":
        device = torch.device(device)
        if device.type == "cuda" and device.index is None:
            device = torch.device("cuda", torch.cuda.current_device() if torch.cuda.is_available() else 0)
        if device == self.device:
            return self
        g = TreeGraph((self._src, self._dst), self._n, device)
        g.batch_num_nodes_list = list(self.batch_num_nodes_list)
        g.batch_num_edges_list = list(self.batch_num_edges_list)
        for k, v in self.ndata.items():
            g.ndata[k] = v.to(device)
        return g

    def cpu(self) -> "TreeGraph":
        return self.to("cpu")

    def csc(self, device=None) -> DeviceCSC:
        """int32 CSC + CSR of the current edge list on ``device`` (built once, cached)."""
        device = torch.device(device) if device is not None else self.device
        if device.type == "cuda" and device.index is None:        # "cuda" and "cuda:<current>" are one cache entry
            device = torch.device("cuda", torch.cuda.current_device() if torch.cuda.is_available() else 0)
        key = str(device)
        if key not in self._csc:
            arrays = build_csc_numpy(self._src, self._dst, self._n)
            self._csc[key] = DeviceCSC(arrays, self._n, self.number_of_edges(), device)
        self._csc[key].segments = self.batch_num_nodes_list          # tree boundaries: DeviceCSC.tiles cuts there
        return self._csc[key]

    # ---- conversions -----------------------------------------------------------------
    def adjacency_matrix(self, scipy_fmt: Optional[str] = None):
        """Row = dst, col = src ... DGL's ``adjacency_matrix()`` puts src on rows for
        ``transpose=False`` in 0.5+; the reference only uses it on symmetric graphs
        (job_runner.py:1742, 1814), so orientation does not matter there. Rows = src here."""
        import scipy.sparse as sp
        m = sp.coo_matrix((np.ones(self.number_of_edges(), np.float32), (self._src, self._dst)),
                          shape=(self._n, self._n))
        if scipy_fmt is not None:
            return m.asformat(scipy_fmt)
        idx = torch.from_numpy(np.stack([self._src, self._dst]))
        return torch.sparse_coo_tensor(idx, torch.ones(self.number_of_edges()), (self._n, self._n))

    def to_networkx(self):
        return to_networkx(self)

    def __repr__(self):
        return (f"TreeGraph(num_nodes={self._n}, num_edges={self.number_of_edges()}, "
                f"batch_size={self.batch_size}, ndata={list(self.ndata.keys())}, device={self.device})")


DGLGraph = TreeGraph  # the name the reference constructs (job_runner.py:1341,1783)


class _Rows:
    """The (row count, device) pair ``_NData`` checks against, for a block's dst-side data."""

    def __init__(self, n, device):
        self._n, self.device = int(n), device

    def number_of_nodes(self):
        return self._n


class Block(TreeGraph):
    """One message-flow graph of neighbour-sampled training (``dgl.to_block``; the ``blocks`` the reference hands to
    ``forward_batch``, job_runner.py:1499-1503, models.py:331-340, 394-400, 685-689).

    Layout: the block is held as a *square* graph over its ``num_src`` source nodes whose first ``num_dst`` nodes
    are the destination nodes (DGL puts the dst nodes first among the src nodes, ``include_dst_in_src=True``) and are
    the only ones with in-edges.  Every aggregation kernel then runs on it unchanged with N = num_src, and a layer
    keeps rows ``[:num_dst]`` of the result, which is what DGL's ``expand_as_pair(feat, block)`` formulation computes.
    ``srcdata`` (num_src rows) is ``ndata``; ``dstdata`` has num_dst rows.
    """

    is_block = True

    def __init__(self, data, num_src: int, num_dst: int, device="cpu"):
        super().__init__(data, num_src, device)
        self._num_dst = int(num_dst)
        if not 0 <= self._num_dst <= self._n:
            raise ValueError(f"block with {num_dst} dst nodes but {num_src} src nodes (dst nodes come first among src)")
        if self._dst.size and int(self._dst.max()) >= self._num_dst:
            raise ValueError("block edge points at a node outside the dst range")
        self.srcdata = self.ndata
        self.dstdata = _NData(_Rows(self._num_dst, self.device))

    def number_of_dst_nodes(self) -> int:
        return self._num_dst

    num_dst_nodes = number_of_dst_nodes

    def number_of_src_nodes(self) -> int:
        return self._n

    num_src_nodes = number_of_src_nodes

    def dstnodes(self) -> torch.Tensor:
        return torch.arange(self._num_dst, dtype=torch.int64, device=self.device)

    srcnodes = TreeGraph.nodes

    def in_degrees(self) -> torch.Tensor:           # of the dst nodes, as DGL reports for a block
        return super().in_degrees()[: self._num_dst]

    def add_edges(self, u, v) -> None:
        raise ValueError("a Block is immutable")

    def int(self) -> "Block":
        return self

    long = int

    def to(self, device) -> "Block":
        device = torch.device(device)
        if device.type == "cuda" and device.index is None:
            device = torch.device("cuda", torch.cuda.current_device() if torch.cuda.is_available() else 0)
        if device == self.device:
            return self
        b = Block((self._src, self._dst), self._n, self._num_dst, device)
        for k, v in self.srcdata.items():
            b.srcdata[k] = v.to(device)
        for k, v in self.dstdata.items():
            b.dstdata[k] = v.to(device)
        return b

    def csc(self, device=None) -> DeviceCSC:
        device = torch.device(device) if device is not None else self.device
        if device.type == "cuda" and device.index is None:        # "cuda" and "cuda:<current>" are one cache entry
            device = torch.device("cuda", torch.cuda.current_device() if torch.cuda.is_available() else 0)
        key = str(device)
        if key not in self._csc:
            c = super().csc(device)
            # the zero-in-degree check of GATConv / GraphConv is about the dst nodes only
            ind = np.bincount(self._dst, minlength=self._n)[: self._num_dst]
            c.min_in_degree = int(ind.min()) if self._num_dst else 0
            c.num_dst = self._num_dst
        return self._csc[key]

    def __repr__(self):
        return (f"Block(num_src_nodes={self._n}, num_dst_nodes={self._num_dst}, num_edges={self.number_of_edges()}, "
                f"device={self.device})")


class DeviceBlock(Block):
    """A Block whose index arrays were built on the device (dataloading's device sampler): it is born with its
    DeviceCSC; the host edge list is only materialised if somebody asks for it (``edges()``, the oracle in tests)."""

    def __init__(self, csc: DeviceCSC, num_dst: int, src_nodes: torch.Tensor, dst_nodes: torch.Tensor,
                 edge_ids: Optional[torch.Tensor] = None):
        self._host_edges = None
        self._device_csc = csc
        TreeGraph.__init__(self, None, csc.num_nodes, csc.device)
        self.batch_num_edges_list = [csc.num_edges]
        self._num_dst = int(num_dst)
        self.srcdata = self.ndata
        self.dstdata = _NData(_Rows(self._num_dst, self.device))
        self.srcdata["_ID"] = src_nodes
        self.dstdata["_ID"] = dst_nodes
        self.edata = {} if edge_ids is None else {"_ID": edge_ids}
        csc.num_dst = self._num_dst
        self._csc[str(self.device)] = csc

    def record_stream(self, stream) -> None:
        """The block was built on another stream than the one that will use it: tell the caching allocator."""
        c = self._device_csc
        for t in (c.indptr, c.indices, c.eid, c.out_indptr, c.out_indices, c.out_pos, *self.srcdata.values(),
                  *self.dstdata.values(), *self.edata.values()):
            if t.is_cuda:
                t.record_stream(stream)

    def _edges_host(self):
        if self._host_edges is None:
            c = self._device_csc
            indptr = c.indptr.cpu().numpy().astype(np.int64)
            dst = np.repeat(np.arange(c.num_nodes, dtype=np.int64), np.diff(indptr))
            src = c.indices.cpu().numpy().astype(np.int64)
            order = np.argsort(c.eid.cpu().numpy(), kind="stable")       # CSC slot order -> edge id order
            self._host_edges = (src[order], dst[order])
        return self._host_edges

    # TreeGraph keeps the edge list in _src/_dst; here they are views of the device arrays, fetched on demand
    @property
    def _src(self):
        return self._edges_host()[0]

    @_src.setter
    def _src(self, value):
        pass

    @property
    def _dst(self):
        return self._edges_host()[1]

    @_dst.setter
    def _dst(self, value):
        pass

    def number_of_edges(self) -> int:
        return self._device_csc.num_edges

    num_edges = number_of_edges

    def to(self, device) -> "Block":
        device = torch.device(device)
        if device.type == "cuda" and device.index is None:
            device = torch.device("cuda", torch.cuda.current_device() if torch.cuda.is_available() else 0)
        if device == self.device:
            return self
        b = Block((self._src, self._dst), self._n, self._num_dst, device)
        for k, v in self.srcdata.items():
            b.srcdata[k] = v.to(device)
        for k, v in self.dstdata.items():
            b.dstdata[k] = v.to(device)
        return b

    def csc(self, device=None) -> DeviceCSC:
        device = torch.device(device) if device is not None else self.device
        if device == self.device:
            return self._device_csc
        return self.to(device).csc(device)


def to_block(frontier: "TreeGraph", dst_nodes, include_dst_in_src: bool = True) -> Block:
    """``dgl.to_block(frontier, dst_nodes)``: compact the edges of ``frontier`` (a graph over the parent's node ids)
    into a Block.  The dst nodes are ``dst_nodes`` in the given order; the src nodes are the dst nodes followed by
    every other edge source in order of first appearance.  ``srcdata['_ID']`` / ``dstdata['_ID']`` hold the parent
    node ids (``dgl.NID``), ``edata['_ID']`` the parent edge ids when the frontier carries them."""
    if not include_dst_in_src:
        raise ValueError("to_block: only include_dst_in_src=True is supported (the layers slice feat[:num_dst])")
    dst_nodes = _as_np_i64(dst_nodes)
    n = frontier.number_of_nodes()
    local = np.full(n, -1, dtype=np.int64)
    if dst_nodes.size:
        if dst_nodes.min() < 0 or dst_nodes.max() >= n:
            raise ValueError("to_block: dst node id out of range")
        local[dst_nodes] = np.arange(dst_nodes.size)
        if np.count_nonzero(local >= 0) != dst_nodes.size:
            raise ValueError("to_block: dst_nodes must be unique")
    src, dst = frontier._src, frontier._dst
    if dst.size and (local[dst] < 0).any():
        raise ValueError("to_block: an edge of the frontier ends outside dst_nodes")
    extra = src[local[src] < 0]
    if extra.size:
        uniq, first = np.unique(extra, return_index=True)
        extra = uniq[np.argsort(first, kind="stable")]          # order of first appearance
        local[extra] = dst_nodes.size + np.arange(extra.size)
    src_nodes = np.concatenate([dst_nodes, extra]) if extra.size else dst_nodes.copy()
    b = Block((local[src], local[dst]), int(src_nodes.size), int(dst_nodes.size), frontier.device)
    b.srcdata["_ID"] = torch.from_numpy(src_nodes)
    b.dstdata["_ID"] = torch.from_numpy(dst_nodes.copy())
    b.edata = {}
    if getattr(frontier, "edata", None) and "_ID" in frontier.edata:
        b.edata["_ID"] = frontier.edata["_ID"]
    return b


def _as_np_i64(x) -> np.ndarray:
    if torch.is_tensor(x):
        return x.detach().cpu().numpy().astype(np.int64)
    return np.asarray(x, dtype=np.int64)


def _edges_from_networkx(G):
    """``DGLGraph(nx_graph)``: nodes must be 0..n-1; a DiGraph contributes its edges in
    iteration order, an undirected Graph both directions (DGL converts via ``to_directed()``,
    whose iteration is per source node in node order == sorted by (u, v) for graphs built
    from a dense adjacency, which is all the reference does: job_runner.py:1331-1341)."""
    import networkx as nx
    n = G.number_of_nodes()
    if sorted(G.nodes()) != list(range(n)):
        raise ValueError("networkx graph nodes must be 0..n-1")
    D = G if G.is_directed() else G.to_directed()
    e = np.asarray(list(D.edges()), dtype=np.int64).reshape(-1, 2)
    if not G.is_directed():   # to_directed() iterates adjacency in insertion order; sort by (u, v)
        e = e[np.lexsort((e[:, 1], e[:, 0]))]
    return e[:, 0].copy(), e[:, 1].copy(), n


def remove_self_loop(g: TreeGraph) -> TreeGraph:
    keep = g._src != g._dst
    out = TreeGraph((g._src[keep], g._dst[keep]), g._n, g.device)
    for k, v in g.ndata.items():
        out.ndata[k] = v
    return out


def batch(graphs: Sequence[TreeGraph]) -> TreeGraph:
    """Block-diagonal union (``dgl.batch``): node/edge ids offset by running sums; node data
    present in every member is concatenated."""
    if len(graphs) == 0:
        raise ValueError("batch of zero graphs")
    device = graphs[0].device
    srcs, dsts, nn, ne, off = [], [], [], [], 0
    for g in graphs:
        if g.device != device:
            raise ValueError("all graphs in a batch must be on one device")
        srcs.append(g._src + off); dsts.append(g._dst + off)
        nn.extend(g.batch_num_nodes_list); ne.extend(g.batch_num_edges_list)
        off += g._n
    out = TreeGraph((np.concatenate(srcs), np.concatenate(dsts)), off, device)
    out.batch_num_nodes_list, out.batch_num_edges_list = nn, ne
    keys = set(graphs[0].ndata.keys())
    for g in graphs[1:]:
        keys &= set(g.ndata.keys())
    for k in sorted(keys):
        out.ndata[k] = torch.cat([g.ndata[k] for g in graphs], dim=0)
    return out


def unbatch(g: TreeGraph) -> List[TreeGraph]:
    outs, no, eo = [], 0, 0
    for n, e in zip(g.batch_num_nodes_list, g.batch_num_edges_list):
        s = g._src[eo:eo + e] - no; d = g._dst[eo:eo + e] - no
        sub = TreeGraph((s, d), n, g.device)
        for k, v in g.ndata.items():
            sub.ndata[k] = v[no:no + n]
        outs.append(sub); no += n; eo += e
    return outs


def to_networkx(g: TreeGraph):
    import networkx as nx
    G = nx.MultiDiGraph()
    G.add_nodes_from(range(g._n))
    G.add_edges_from(zip(g._src.tolist(), g._dst.tolist()))
    return G


def graph_from_adj(adj, device="cpu", add_self_loops: bool = True, graph_mode: str = "all_connected") -> TreeGraph:
    """One tree's graph straight from the dense ``adj`` of the cached-embedding schema
    (reference job_runner.py:796-803), equivalent to the nx.DiGraph -> DGLGraph ->
    remove_self_loop -> add_edges(nodes, nodes) sequence of job_runner.py:1779-1800; ``graph_mode``: see
    :func:`edges_from_adj` (the reference's GRAPH_MODE)."""
    a = adj.detach().cpu().numpy() if torch.is_tensor(adj) else np.asarray(adj)
    u, v = edges_from_adj(a, add_self_loops, graph_mode)
    return TreeGraph((u, v), a.shape[0], device)
