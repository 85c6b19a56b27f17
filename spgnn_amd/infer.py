"""Per-scan inference: the reference's other call pattern on the hot path.

``GCNTestSPGNN.run`` / ``GCNTest.run`` / ``SPGNNE2ETest`` build ONE graph per scan, ``dgl.batch([g])`` it and call
``model.forward(g)`` once (reference job_runner.py:1601-1610, 2046-2052, 2261-2263): a forward pass over a single airway tree
of 100-300 branches, latency- not throughput-bound.  Issued eagerly such a pass is ~40 launches of a few microseconds each and
the host paces it; :class:`ForwardRunner` captures the eval-mode forward once per SIZE CLASS of the scan (a batch arena,
spgnn_amd/arena.py: the tree is copied into fixed buffers and padded with pad nodes that form components of their own, so no
real node's value changes) and replays it for every later scan of the class: a scan = a dozen small device copies + one HIP
graph launch.  No gradients, no dropout (``model.eval()``), outputs sliced back to the scan's real nodes.

The parameters are FROZEN for a runner: what the forward derives from them alone (the projection operands of
spgnn_weight_prep, the folded score vectors) is computed once and kept out of the captured graph, so a replay starts at the
first kernel that touches node data.  A change of the parameters is noticed at the next call (their version counters and
the update epoch ``train.TrainStep`` keeps on the model) and drops the captures; ``runner.reset()`` does the same by hand.
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch

from .arena import BatchArena

__all__ = ["ForwardRunner"]


class ForwardRunner:
    """``runner(g)`` == ``model(g)`` in eval mode under ``torch.no_grad()`` for a device graph ``g`` (one scan or a batch),
    as a HIP-graph replay per size class.  Outputs are views of buffers the next call of the same class overwrites:
    ``clone()`` what must outlive it (the reference consumes them at once: softmax + argmax, job_runner.py:2053-2055)."""

    def __init__(self, model: torch.nn.Module, granule: int = 64, max_classes: int = 16):
        self.model, self.granule, self.max_classes = model, granule, max_classes
        self._classes: Dict[tuple, Tuple[BatchArena, torch.cuda.CUDAGraph, tuple]] = {}
        self._frozen: dict = {}                    # ops.FROZEN_WEIGHTS of this runner: what the forward derives from the parameters alone
        self._stamp = self._fingerprint()

    def _forward(self, ag):
        """The model's forward with this runner's frozen-weight cache installed and the step's scale blocks pooled (one re-arm
        launch instead of a clone per block; inference never reads them back)."""
        from . import ops
        pool = ops.scale_pool(ag.device)
        prev, ops.FROZEN_WEIGHTS = ops.FROZEN_WEIGHTS, self._frozen
        pool.begin()
        try:
            return self.model(ag)
        finally:
            pool.end()
            ops.FROZEN_WEIGHTS = prev

    def _capture(self, arena: BatchArena):
        ag = arena.graph
        dev = arena.device
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(2):                     # lazy initialisations, allocator pools; fills the frozen-weight cache:
                self._forward(ag)                  # projection operands and folded score vectors of the CURRENT parameters
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        from . import ops
        prev_refs, ops.CAPTURE_REFS = ops.CAPTURE_REFS, []
        try:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, capture_error_mode="thread_local"), torch.no_grad():
                outs = self._forward(ag)           # recorded WITHOUT the weight-only launches: a replay starts at the node data
            refs = ops.CAPTURE_REFS
        finally:
            ops.CAPTURE_REFS = prev_refs
        outs = outs if isinstance(outs, (tuple, list)) else (outs,)
        return graph, tuple(outs), refs

    def reset(self) -> None:
        """Drop every capture and the frozen-weight cache: call after the model's parameters changed (another checkpoint)."""
        self._classes.clear()
        self._frozen = {}
        self._stamp = self._fingerprint()

    def _fingerprint(self):
        """What tells that the parameters changed since the captures were made: every parameter's version counter
        (load_state_dict, optimizer.step, any in-place update through torch bump it) and the model's update epoch, which
        train.TrainStep advances on every step and replay - its fused optimizer kernel writes the flat bucket through raw
        pointers, which no version counter sees.  A few microseconds per call; a change drops the captures (reset())."""
        ps = self.__dict__.get("_params")
        if ps is None:                             # (the module tree is walked once: ~40 us for 60 parameters, per scan otherwise)
            ps = self._params = list(self.model.parameters())
        return ([p._version for p in ps], getattr(self.model, "_spgnn_param_epoch", 0))

    def __call__(self, g):
        if self.model.training:
            raise RuntimeError("ForwardRunner replays an eval-mode forward: call model.eval() first")
        if self._classes and self._fingerprint() != self._stamp:
            self.reset()                           # the parameters moved under the frozen operands: capture again with the new ones
        key = BatchArena.class_key(g, self.granule)
        hit = self._classes.pop(key, None)
        if hit is None:
            while len(self._classes) >= self.max_classes:
                self._classes.pop(next(iter(self._classes)))
            arena = BatchArena(g, self.granule)
            from . import ops
            # scans of this class run their projections on the skinny kernel (fp32, no operand scales, no pre-split images):
            # nothing to refresh per scan beyond the data itself
            # (only below MIN_GEMM_ROWS: from there on some layer forms - fused output layers, pooled SAGE products - may still
            # take a matrix-core product whose operand scale is cached on the node data and must follow every scan)
            arena.refresh_constants = not (ops.SKINNY_GEMM and arena.n_cap <= ops.SKINNY_ROWS and arena.n_cap < ops.MIN_GEMM_ROWS)
            arena.keep_edges = False               # the forward pass reads CSC / CSR only: no src / dst copy per scan
            arena.load(g, key)
            graph, outs, refs = self._capture(arena)
            hit = (arena, graph, outs, refs)
        else:
            hit[0].load(g, key)
        self._classes[key] = hit                   # re-inserted last: dict order = recency
        hit[1].replay()
        n = g.number_of_nodes()
        return tuple(o[:n] if (o is not None and o.dim() >= 1 and o.shape[0] == hit[0].n_cap) else o for o in hit[2])
