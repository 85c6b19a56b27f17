"""Batch arenas: loader batches of one size class share ONE set of device buffers, so a training step captured on the
first batch of the class (train.TrainStep.capture: one HIP graph, two when ranks exchange) serves every later one - a new batch is a handful of
device copies into fixed addresses plus ``replay()``, not a warm-up, a capture and an instantiation.

Reference loop being served (job_runner.py:1870-1920, exp_settings/st_pgat_spgnn_3.py:29,34): for every loader batch of
TRAIN_BATCH_SIZE = 64 trees, build the batched graph, then take GCN_STEPS = 300 optimizer steps on it.  The batches
differ in node and edge count (trees of 100-300 branches), and a HIP graph bakes in every kernel's grid size, every
N / E argument and every pointer.  So a batch is PADDED to its size class: N up to the next multiple of ``granule``
nodes, E up to the edge count of that many nodes, by appending pad nodes that

  * form components of their own (one path + isolated nodes; every pad node has its self loop, as every real node does:
    ``g.add_edges(g.nodes(), g.nodes())``, job_runner.py:1800) - no edge joins them to a tree, so no real node's output
    changes (the adjacency stays block diagonal, the same fact tree-sharded data parallelism rests on);
  * carry zero features and positional encodings and label 0;
  * are never sampled into the loss: their sampling probability is -1 (train.TrainStep._sampling), so the node mask of
    job_runner.py:1896 excludes them, their logit gradient is zero and they contribute nothing to any weight gradient.

What padding does change: per-tensor GEMM scales see the pad rows' activations (bounded by the bias terms), and split-K
row ranges move - i.e. fp32 summation order, within the parity tolerance, never an index (tests/test_arena.py compares
replays on an arena with eager steps on the same padded graph bit for bit, and with the unpadded batch to 1e-5).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from . import graph as G
from . import ops

__all__ = ["BatchArena", "size_class"]


def size_class(num_nodes: int, num_edges: int, num_trees: int, granule: int = 256) -> Tuple[int, int]:
    """(n_cap, e_cap) of the arena a batch of ``num_nodes`` / ``num_edges`` (self loops included) goes to.  n_cap: the next
    multiple of ``granule`` strictly above N (at least one pad node).  e_cap: for tree batches E = 3N - 2B, so the edge
    count of n_cap nodes in B + 1 trees - the pad is then ONE path and every batch of B trees in this node class shares the
    class; a batch that does not fit that (not trees, or another B) gets the class of its own edge count."""
    n_cap = (num_nodes // granule + 1) * granule
    n_pad = n_cap - num_nodes
    e_cap = 3 * n_cap - 2 * (num_trees + 1)
    twice_m = e_cap - num_edges - n_pad                   # pad edges beyond the self loops: 2 per undirected path edge
    if twice_m < 0 or twice_m % 2 or twice_m // 2 > n_pad - 1:
        e_cap = num_edges + n_pad + 2 * (n_pad - 1)       # the whole pad as one path
    return n_cap, e_cap


def _pad_graph(n_pad: int, m: int) -> Dict[str, np.ndarray]:
    """Index arrays of the pad component(s): nodes 0..m a path (m undirected edges), nodes m+1.. isolated, a self loop on
    every node; edge ids in the reference's order (off-diagonal entries by (u, v), then the self loops,
    job_runner.py:1779-1801).  Local node / edge / slot numbers: the arena offsets them."""
    j = np.arange(m + 1, dtype=np.int64)
    u = np.concatenate([j[1:], j[:-1]])
    v = np.concatenate([j[1:] - 1, j[:-1] + 1])
    order = np.lexsort((v, u))
    loops = np.arange(n_pad, dtype=np.int64)
    src, dst = np.concatenate([u[order], loops]), np.concatenate([v[order], loops])
    arrays = G.build_csc_numpy(src, dst, n_pad)
    arrays["src"], arrays["dst"] = src.astype(np.int32), dst.astype(np.int32)
    return arrays


_PAD_CACHE: Dict[tuple, Dict[str, torch.Tensor]] = {}     # (device, n_pad, m) -> the pad's index arrays on the device


def _pad_pieces(n_pad: int, m: int, dev) -> Dict[str, torch.Tensor]:
    """:func:`_pad_graph` on the device, ONE upload per (n_pad, m): the pad depends on nothing else, and a stream of scans or
    loader batches revisits the same few shapes (per-scan inference paid ~0.2 ms of numpy + upload per scan for it)."""
    key = (str(dev), int(n_pad), int(m))
    hit = _PAD_CACHE.pop(key, None)
    if hit is None:
        pad = _pad_graph(n_pad, m)
        order = ("indptr", "out_indptr", "indices", "out_indices", "eid", "out_pos", "src", "dst")
        host = np.concatenate([pad[k].astype(np.int32, copy=False) for k in order])
        up = torch.from_numpy(host).to(dev)
        off, hit = 0, {}
        for k in order:
            hit[k] = up[off:off + pad[k].shape[0]]
            off += pad[k].shape[0]
        while len(_PAD_CACHE) >= 256:
            _PAD_CACHE.pop(next(iter(_PAD_CACHE)))
    _PAD_CACHE[key] = hit
    return hit


class BatchArena:
    """Fixed-address storage for the batches of one size class on one device."""

    INDEX_KEYS = ("indptr", "indices", "eid", "out_indptr", "out_indices", "out_pos")

    def __init__(self, first: G.TreeGraph, granule: int = 256):
        dev = first.device
        if dev.type != "cuda":
            raise RuntimeError("BatchArena serves captured HIP graphs: it needs a ROCm device")
        N, E, B = first.number_of_nodes(), first.number_of_edges(), first.batch_size
        self.granule, self.device = granule, dev
        self.n_cap, self.e_cap = size_class(N, E, B, granule)
        self.key = self.class_key(first, granule)
        i32 = lambda n: torch.zeros((n,), dtype=torch.int32, device=dev)
        t = dict(indptr=i32(self.n_cap + 1), indices=i32(self.e_cap), eid=i32(self.e_cap), out_indptr=i32(self.n_cap + 1),
                 out_indices=i32(self.e_cap), out_pos=i32(self.e_cap))
        self.src, self.dst = i32(self.e_cap), i32(self.e_cap)
        self.keep_edges = True          # False (infer.ForwardRunner): load() skips the edge-id-order copy of the edge list
        csc = G.DeviceCSC.from_tensors(t, self.n_cap, self.e_cap)
        csc._fixed_tile_count = True             # DeviceCSC.tiles: the launch grids cover the fixed-length tile table (see _refresh)
        g = self.graph = G.TreeGraph.from_device(self.src, self.dst, self.n_cap, csc, [self.n_cap], [self.e_cap])
        g._stable_storage = True
        g._refresh_hooks = []
        g.arena = self
        # node data: same names, widths and dtypes as the first batch; aliases (ndata['p'] is ndata['pos_enc']) stay aliases
        made: Dict[int, torch.Tensor] = {}
        for k, v in first.ndata.items():
            buf = made.get(id(v))
            if buf is None:
                buf = made[id(v)] = torch.zeros((self.n_cap,) + tuple(v.shape[1:]), dtype=v.dtype, device=dev)
            dict.__setitem__(g.ndata, k, buf)            # (plain insert: _NData.__setitem__ would clear the derived cache)
        self.loads = 0
        self._rows_dirty = 0                      # node-data rows [0, _rows_dirty) may be non-zero (buffers start zeroed)
        self.refresh_constants = True             # False: the GEMM scales / pre-split images of node data are not refreshed per load
                                                  # (per-scan inference on the skinny products, which take neither: infer.ForwardRunner)

    @staticmethod
    def class_key(g: G.TreeGraph, granule: int = 256):
        csc = g.csc(g.device)
        n_cap, e_cap = size_class(g.number_of_nodes(), g.number_of_edges(), g.batch_size, granule)
        nd = tuple(sorted((k, tuple(v.shape[1:]), str(v.dtype)) for k, v in g.ndata.items()))
        # the kernels choose their forms from the degree bounds (<= 8 in- / out-edges: the straight-line paths): a batch
        # on the other side of that line must not replay a graph captured on this side
        #  ... and so must a batch with a sink or a source-less node (no self loop somewhere): the fused LSPE level
        # (ops.lspe_supported) and GATConv's zero-in-degree error both look at the minimum degrees
        return (str(g.device), n_cap, e_cap, csc.max_in_degree <= 8, csc.max_out_degree <= 8,
                csc.min_in_degree >= 1, int(getattr(csc, "min_out_degree", 0) or 0) >= 1, nd)

    def load(self, g: G.TreeGraph, key=None) -> G.TreeGraph:
        """Copy batch ``g`` (a device graph with its node data; e.g. data.assemble_batch) into the arena, pad it to the class
        and refresh everything derived from it, in place.  -> the arena's graph (always the same object).  ``key``: the
        batch's class key when the caller has just computed it (per-scan inference: no second pass over the node data dict)."""
        if (key if key is not None else self.class_key(g, self.granule)) != self.key:
            raise ValueError("batch does not belong to this arena's size class")
        ag, dev = self.graph, self.device
        csc, acsc = g.csc(dev), ag.csc(dev)
        N, E = g.number_of_nodes(), g.number_of_edges()
        n_pad = self.n_cap - N
        m = (self.e_cap - E - n_pad) // 2
        pieces = _pad_pieces(n_pad, m, dev)                                 # the pad's index arrays on the device (cached per shape)
        with torch.no_grad():
            # the six index arrays, every node-data tensor and the tensors derived from node data in ONE launch (spgnn_arena_load,
            # ABI 62; before: spgnn_copy_pad_i32 + a copy and a zero fill per tensor + a builder call and a copy per derived tensor,
            # ~17 launches per inference scan).  Index arrays: the batch's array, then the pad component's shifted by the batch's
            # edge / node count (slot offsets: real edges fill slots [0, E), pad nodes follow the real ones)
            import ctypes
            from . import _capi
            jobs = _capi.CopyPadJobs()
            q = 0
            for k, n_real, shift, skip in (("indptr", N + 1, E, 1), ("out_indptr", N + 1, E, 1), ("indices", E, N, 0),
                                           ("out_indices", E, N, 0), ("eid", E, E, 0), ("out_pos", E, E, 0)):
                dst, src_, pad = getattr(acsc, k), getattr(csc, k), pieces[k]
                if src_.dtype != torch.int32 or not src_.is_contiguous():
                    src_ = src_.to(torch.int32).contiguous()
                n_tail = dst.shape[0] - n_real                             # the pad component's share of this array
                assert pad.dtype == torch.int32 and pad.shape[0] - skip == n_tail and src_.shape[0] == n_real and dst.dtype == torch.int32
                jobs.job[q] = _capi.CopyPadJob(dst.data_ptr(), src_.data_ptr(), pad.data_ptr() + 4 * skip, n_real, n_tail, shift, 0)
                q += 1
            jobs.n_jobs = q
            rows, slow, derived_done = self._row_jobs(g, N)
            with torch.cuda.device(dev):
                _capi.check(_capi.load().spgnn_arena_load(ctypes.addressof(jobs), ctypes.addressof(rows) if rows.n_jobs else None,
                                                         torch.cuda.current_stream(dev).cuda_stream), "spgnn_arena_load")
            if self.keep_edges:                                             # src / dst in edge-id order (graph.edges(); no kernel reads them)
                ed = getattr(g, "_edges_dev", None)
                s_, d_ = ed if (ed is not None and ed[0].device == self.src.device) else g.edges()
                self.src[:E].copy_(s_); self.dst[:E].copy_(d_)
                torch.add(pieces["src"], N, out=self.src[E:]); torch.add(pieces["dst"], N, out=self.dst[E:])
                ag._edges_dev = (self.src, self.dst)
            else:
                ag._edges_dev = None                                        # asking for them raises (graph._host_edges)
            for buf, v in slow:                          # (tensors the row-copy jobs cannot express: odd byte widths, strided rows)
                buf[:N].copy_(v)
                if self._rows_dirty > N:                 # pad rows a LARGER earlier batch filled: back to zero (else they still are)
                    buf[N:self._rows_dirty].zero_()
        self._derived_done = derived_done
        self._rows_dirty = N
        acsc.min_in_degree = min(csc.min_in_degree, 1)
        acsc.max_in_degree = max(csc.max_in_degree, 3 if m > 1 else (2 if m == 1 else 1))
        acsc.max_out_degree = max(csc.max_out_degree, 3 if m > 1 else (2 if m == 1 else 1))
        acsc.min_out_degree = min(int(getattr(csc, "min_out_degree", 0) or 0), 1)    # unknown counts as 0: never claims an edge that may not be there
        ag.num_real_nodes, ag.num_real_edges = N, E
        ag.batch_num_nodes_list = list(g.batch_num_nodes_list) + [n_pad]
        ag.batch_num_edges_list = list(g.batch_num_edges_list) + [self.e_cap - E]
        ag._src_np = ag._dst_np = None
        acsc.segments = ag.batch_num_nodes_list          # this batch's tree boundaries (+ the pad): the LDS tiles are cut there
        self._refresh(acsc)
        self.loads += 1
        return ag

    @staticmethod
    def _words(t: torch.Tensor):
        """(words per row, row stride in words) of a tensor whose rows can be copied as 4-byte words, else None."""
        es = t.element_size()
        if t.dim() == 0 or es % 4 or t.data_ptr() % 4:
            return None
        if t.dim() == 1:
            return (es // 4, es // 4) if t.stride(0) == 1 else None
        inner = 1
        for d in range(t.dim() - 1, 0, -1):              # dims 1.. must be dense; dim 0 may carry a padded stride
            if t.stride(d) != inner:
                return None
            inner *= t.shape[d]
        if t.stride(0) < inner or inner == 0:
            return None
        return inner * es // 4, t.stride(0) * es // 4

    def _row_jobs(self, g: G.TreeGraph, N: int):
        """The row-copy jobs of one load: every node-data tensor of ``g`` into its arena buffer, and the derived tensors whose
        recipe is a plain concatenation / aligned copy of node data (models._data_cat / _data_aligned) straight from ``g``'s
        tensors.  -> (jobs, [(buffer, tensor)] left to torch copies, the derived keys the jobs cover).
        The job table is PLANNED once per layout of the incoming node data (names, widths, dtypes, strides) and set of derived
        tensors; a load then only fills in the source pointers and the two row counts (per-scan inference: the planning was
        a third of the host time of a scan)."""
        from . import _capi
        ag = self.graph
        sig = (tuple((k, tuple(v.shape[1:]), v.dtype, v.stride()) for k, v in g.ndata.items()), tuple(ag._derived_builders))
        plan = self._plan if getattr(self, "_plan_sig", None) == sig else None
        if plan is None:
            entries, slow_names, done = [], [], set()       # entries: (dst tensor, dst stride, dst col, width, source name, src stride)

            def add(dst, dst_col_w, src, name):
                ws, wd = self._words(src), self._words(dst)
                if ws is None or wd is None or len(entries) >= 16 or src.dtype != dst.dtype or dst_col_w + ws[0] > wd[1]:
                    return False
                entries.append((dst, wd[1], dst_col_w, ws[0], name, ws[1]))
                return True
            seen = set()
            for k, v in g.ndata.items():
                buf = ag.ndata[k]
                if id(buf) in seen:
                    continue
                seen.add(id(buf))
                if not (tuple(buf.shape[1:]) == tuple(v.shape[1:]) and add(buf, 0, v, k)):
                    slow_names.append(k)
            for key in list(ag._derived_builders):
                held = ag._tensor_cache[key]
                names = None
                if key[0] == "cat" and all(isinstance(k_, tuple) and k_[0] == "ndata" for k_ in key[1:]):
                    names = [k_[1] for k_ in key[1:]]
                elif key[0] == "aligned" and isinstance(key[1], tuple) and key[1][0] == "ndata":
                    names = [key[1][1]]
                parts = [g.ndata.get(n_) for n_ in names] if names else None
                if not parts or any(p_ is None or p_.dim() != 2 or p_.dtype != held.dtype for p_ in parts) or held.dim() != 2 \
                        or sum(p_.shape[1] for p_ in parts) != held.shape[1] or len(entries) + len(parts) > 16:
                    continue
                n0, col, ok = len(entries), 0, True
                for n_, p_ in zip(names, parts):
                    ok = ok and add(held, col * held.element_size() // 4, p_, n_)
                    col += p_.shape[1]
                if ok:
                    done.add(key)
                else:
                    del entries[n0:]
            jobs = _capi.RowCopyJobs()
            for q, (dst, dstr, dcol, w, _name, sstr) in enumerate(entries):
                jobs.job[q] = _capi.RowCopyJob(dst.data_ptr(), 0, dstr, sstr, 0, 0, dcol, w)
            jobs.n_jobs = len(entries)
            plan = self._plan = (jobs, entries, slow_names, done)
            self._plan_sig = sig
        jobs, entries, slow_names, done = plan
        total = max(N, self._rows_dirty)
        nd = g.ndata
        bump = torch.autograd.graph.increment_version
        for q, e in enumerate(entries):
            j = jobs.job[q]
            j.src, j.rows_copy, j.rows_total = nd[e[4]].data_ptr(), N, total
            # the kernel writes behind torch's back: what torch's own copy_ did for the caches keyed on a tensor's version
            # (TrainStep._sampling recomputes the sampling probabilities when ndata['y'] changed; ops.operand_scale)
            bump(e[0])
        return jobs, [(ag.ndata[k], nd[k]) for k in slow_names], done

    def _refresh(self, acsc: G.DeviceCSC) -> None:
        """Everything computed FROM the batch, recomputed into the storage a captured step already addresses: the padded
        neighbour rows and degree vectors of the graph, the cached concatenations / aligned copies of node data with their
        GEMM scales and pre-split images, and whatever a training step registered (its sampling probabilities)."""
        ag = self.graph
        with torch.no_grad():
            old = dict(acsc._cache)
            acsc._cache = {}
            try:
                for key, val in old.items():
                    if key == "ell":
                        if acsc.indptr.is_cuda and acsc.num_edges > 0:       # both directions in one launch, into the cached rows
                            from . import _capi
                            with torch.cuda.device(self.device):
                                _capi.check(_capi.load().spgnn_ell_rows_both(
                                    acsc.indptr.data_ptr(), acsc.indices.data_ptr(), acsc.out_indptr.data_ptr(), acsc.out_indices.data_ptr(),
                                    acsc.out_pos.data_ptr(), acsc.num_nodes, acsc.num_edges, val[0].data_ptr(), val[1].data_ptr(),
                                    val[2].data_ptr(), torch.cuda.current_stream(self.device).cuda_stream), "spgnn_ell_rows_both")
                            continue
                        new = acsc.ell()
                    elif key == "in_deg":
                        new = acsc.in_degrees_f()
                    elif key == "out_deg":
                        new = acsc.out_degrees_f()
                    elif isinstance(key, tuple) and key[0] == "deg_scale":
                        new = acsc.degree_scale(key[1], key[2])
                    elif isinstance(key, tuple) and key[0] == "tiles":
                        acsc._cache = {key: val}             # rewritten IN PLACE (same length for every batch of the class)
                        acsc.tiles(key[1], rebuild=True)
                        acsc._cache = {}
                        continue
                    else:
                        continue
                    for o, n_ in zip(val if isinstance(val, tuple) else (val,), new if isinstance(new, tuple) else (new,)):
                        o.copy_(n_)
            finally:
                acsc._cache = old                    # the tensors a captured step addresses stay the cached ones, whatever happened
            done = getattr(self, "_derived_done", ())
            for key, builder in list(ag._derived_builders.items()):
                held = ag._tensor_cache[key]
                if key not in done:                      # (the load's row-copy jobs wrote the plain concatenations already)
                    held.copy_(builder())
                if self.refresh_constants:
                    ops.refresh_batch_constant(held)
            seen = set()
            for v in ag.ndata.values():
                # every node-data tensor that carries a GEMM scale or a pre-split image, marked constant or not: a captured
                # step would otherwise keep the previous batch's scale
                if self.refresh_constants and id(v) not in seen and (getattr(v, "_spgnn_const", False) or hasattr(v, "_spgnn_scale")
                                                                      or hasattr(v, "_spgnn_aps")):
                    ops.refresh_batch_constant(v)
                seen.add(id(v))
            for fn in ag._refresh_hooks:
                fn()
